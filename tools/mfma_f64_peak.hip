// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 and v_fma_f64 rates on gfx950, to calibrate the FP64 "peak" the
// likelihood kernels' roofline fractions are quoted against (MI355X_MICROARCH.md has no f64 MFMA row).
//   hipcc --offload-arch=gfx950 -O3 -o tools/mfma_f64_peak tools/mfma_f64_peak.hip && tools/mfma_f64_peak
//
// Round 6: the round-2 version re-used ONE (a, b) pair for every accumulator and measured 47 - 49
// TFLOP/s, while score_mfma_kernel executes 62 in a real sweep.  This one varies what could have held the loop back:
//   * NACC independent accumulators, each with its OWN a and b operand registers (no operand shared between two
//     instructions in flight),
//   * the operands rotated every iteration (a fresh register pair per issue, as a kernel that streams tiles has),
//   * 1 / 2 / 4 / 8 wavefronts per SIMD,
//   * the loop timed by HIP events AND by s_memtime inside the kernel (cycles per MFMA per SIMD, clock-independent),
// and prints TFLOP/s = issued MFMAs x 2048 flop / time for each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

// (inline asm with the accumulator pinned to VGPRs: through the builtin the compiler keeps the loop-carried accumulators
//  in VGPRs but feeds the instruction from AGPRs -- 8 v_accvgpr_write + 8 v_accvgpr_read per MFMA and an s_nop 11 per
//  iteration in the ISA of this very loop, which is what the round-2 figure of 47 - 49 TFLOP/s measured)
#define MFMA(ACC, A, B) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B));
// MODE 0: one shared (a, b);  1: own (a_i, b_i) per accumulator;  2: own operands, rotated through a ring of 2 NACC
template <int NACC, int MODE>
__global__ __launch_bounds__(256) void mfma_loop(double *out, unsigned long long *cyc, int iters) {
    v4d acc[NACC];
    double a[2 * NACC], b[2 * NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 2 * NACC; ++i) { a[i] = threadIdx.x * 1e-3 + i; b[i] = 1.0 + threadIdx.x * 1e-6 * (i + 1); }
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();  // constant 100 MHz
    const unsigned long long t0 = __builtin_readcyclecounter();      // s_memtime: shader clock
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            const int k = MODE == 0 ? 0 : i;
            MFMA(acc[i], a[k], b[k])
        }
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            const int k = MODE == 0 ? 0 : (MODE == 1 ? i : NACC + i);
            MFMA(acc[i], a[k], b[k])
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        cyc[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = t1 - t0;
        cyc[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0;
    }
}

// the round-2 loop verbatim (builtin, one shared operand pair, four accumulators): kept so that the two are measured side by side
__global__ __launch_bounds__(256) void mfma_loop_r2(double *out, int iters) {
    v4d acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void fma_loop(double *out, int iters) {
    double acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = i;
    double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = fma(acc[i], a, b);
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
static float time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms;
}

static double *out;
static unsigned long long *cyc;
static int n_cu = 256;

template <int NACC, int MODE>
static void run(int bpc, int iters) {
    const int grid = n_cu * bpc;
    float ms = time_ms([&] { hipLaunchKernelGGL((mfma_loop<NACC, MODE>), dim3(grid), dim3(256), 0, 0, out, cyc, iters); });
    std::vector<unsigned long long> h(grid * 8);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * grid * 8, hipMemcpyDeviceToHost);
    double mean = 0, real = 0;
    for (size_t i = 0; i < h.size(); i += 2) { mean += (double)h[i]; real += (double)h[i + 1]; }
    mean /= (double)(h.size() / 2);
    real /= (double)(h.size() / 2);
    const double n_issue = (double)iters * NACC;                     // per wavefront
    const double flops = (double)grid * 4 * n_issue * 2048.0;
    // s_memtime counts shader clocks, s_memrealtime the constant 100 MHz reference: their ratio is the clock the SIMD ran
    // at INSIDE the loop; shader clocks per MFMA per SIMD = clocks of one wavefront's loop / (its issues x wavefronts
    // sharing the SIMD) -- 64 if the pipe is the limit (16 passes of 4 clocks).
    const double us = ms * 1e3;
    printf("acc=%2d operands=%-7s waves/SIMD=%d : %8.3f ms  %6.1f TFLOP/s   %.2f ns, %.1f shader clocks per MFMA per SIMD, clock in the loop %.0f MHz\n",
           NACC, MODE == 0 ? "shared" : (MODE == 1 ? "own" : "rotated"), bpc, ms, flops / ms / 1e9,
           us * 1e3 / (n_issue * bpc), mean / (n_issue * bpc), mean / real * 100.0);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    n_cu = prop.multiProcessorCount;
    printf("%s: %d CUs, clock %d MHz (nominal)\n", prop.gcnArchName, n_cu, prop.clockRate / 1000);
    hipMalloc(&out, sizeof(double) * 256 * 256 * 64);
    hipMalloc(&cyc, sizeof(unsigned long long) * 256 * 8 * 8);
    const int iters = 20000;
    for (int bpc : {1, 2, 4}) {
        const int grid = n_cu * bpc;
        float ms = time_ms([&] { hipLaunchKernelGGL(mfma_loop_r2, dim3(grid), dim3(256), 0, 0, out, iters); });
        printf("round-2 loop (builtin, acc=4, shared operands) waves/SIMD=%d : %.3f ms  %.1f TFLOP/s\n", bpc, ms,
               (double)grid * 4 * iters * 4 * 2048.0 / ms / 1e9);
    }
    for (int bpc : {1, 2, 4, 8}) run<4, 0>(bpc, iters);              // the same configuration, accumulators pinned to VGPRs
    for (int bpc : {1, 2, 4}) run<4, 1>(bpc, iters);
    for (int bpc : {1, 2, 4}) run<8, 1>(bpc, iters);
    for (int bpc : {1, 2}) run<16, 1>(bpc, iters);
    for (int bpc : {1, 2, 4}) run<8, 2>(bpc, iters);
    for (int bpc : {1, 2}) run<16, 2>(bpc, iters);
    run<1, 0>(1, iters);             // ONE dependent chain alone on a SIMD
    run<1, 0>(2, iters);
    run<2, 1>(1, iters);
    for (int bpc : {2, 8}) {
        const int grid = n_cu * bpc;
        float ms = time_ms([&] { hipLaunchKernelGGL(fma_loop, dim3(grid), dim3(256), 0, 0, out, iters); });
        double flops = (double)grid * 256 * iters * 8 * 2.0;
        printf("v_fma_f64 waves/SIMD=%d : %.2f ms  %.1f TFLOP/s\n", bpc, ms, flops / ms / 1e9);
    }
    return 0;
}

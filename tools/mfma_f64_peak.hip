// Microbenchmark: sustained v_mfma_f64_16x16x4_f64 and v_fma_f64 rates on gfx950, to calibrate
// the FP64 "peak" the likelihood kernel's roofline fraction is quoted against
// (MI355X_MICROARCH.md has no f64 MFMA row).   hipcc --offload-arch=gfx950 -O3 -o mfma_f64_peak ...
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(double *out, int iters) {
    v4d acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (v4d){0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-6;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
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
    return ms;
}

int main() {
    double *out; hipMalloc(&out, sizeof(double) * 256 * 256 * 64);
    const int iters = 20000;
    for (int bpc : {1, 2, 4, 8}) {           // blocks per CU -> waves per SIMD
        const int grid = 256 * bpc;
        float ms = time_ms([&] { hipLaunchKernelGGL(mfma_loop<4>, dim3(grid), dim3(256), 0, 0, out, iters); });
        double flops = (double)grid * 4 /*waves*/ * iters * 4 /*acc*/ * 2048.0;
        printf("mfma_f64_16x16x4 acc=4 waves/SIMD=%d : %.2f ms  %.1f TFLOP/s\n", bpc, ms, flops / ms / 1e9);
    }
    {
        const int grid = 256 * 2;
        float ms = time_ms([&] { hipLaunchKernelGGL(mfma_loop<1>, dim3(grid), dim3(256), 0, 0, out, iters); });
        double flops = (double)grid * 4 * iters * 1 * 2048.0;
        printf("mfma_f64_16x16x4 acc=1 (dependent chain) waves/SIMD=2 : %.2f ms  %.1f TFLOP/s\n", ms, flops / ms / 1e9);
        ms = time_ms([&] { hipLaunchKernelGGL(mfma_loop<2>, dim3(256), dim3(256), 0, 0, out, iters); });
        flops = (double)256 * 4 * iters * 2 * 2048.0;
        printf("mfma_f64_16x16x4 acc=2 waves/SIMD=1 : %.2f ms  %.1f TFLOP/s\n", ms, flops / ms / 1e9);
    }
    for (int bpc : {2, 8}) {
        const int grid = 256 * bpc;
        float ms = time_ms([&] { hipLaunchKernelGGL(fma_loop, dim3(grid), dim3(256), 0, 0, out, iters); });
        double flops = (double)grid * 256 * iters * 8 * 2.0;
        printf("v_fma_f64 waves/SIMD=%d : %.2f ms  %.1f TFLOP/s\n", bpc, ms, flops / ms / 1e9);
    }
    return 0;
}

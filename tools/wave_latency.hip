// Single-wavefront instruction costs on gfx950 (cycles, s_memtime): what bounds the one-workgroup kernels whose
// cost is the length of one dependent chain.   hipcc --offload-arch=gfx950 -O3 tools/wave_latency.hip -o /tmp/wl && /tmp/wl
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../pybgmm_amd/csrc/fast_math.h"
#define N 256
__global__ void k(long long *out, double *sink, const double *gsrc, int nwaves_active) {
    __shared__ double lds[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = (double)((i * 7 + 1) & 4095);
    __syncthreads();
    if (wave != 0 && nwaves_active > 4) {      // neighbours that COMPUTE (the resolver's update wavefronts): the same arithmetic
        if (wave < nwaves_active - 4) {
            double a = 1.0 + lane * 1e-9;
            const long long t0 = clock64();
#pragma unroll 1
            for (int i = 0; i < 4000; ++i) {
                const double acc = a * 1e-3, cdv = 0.7, Dt = 1.0 + acc;
                const double invD = fm_div(1.0, Dt);
                const double qv = fma(-(acc * acc), invD, cdv) - 0.01;
                const double f = qv * 0.05;
                const double rcf = 1.0 * fm_rsqrt(fabs(Dt));
                const double l1 = fm_log1p_small(f);
                a = fm_exp((2.0 - 1.5) - 30.0 * l1) * rcf;
                asm volatile("" : "+v"(a));
            }
            const long long t1 = clock64();
            if (lane == 0) out[40 + wave] = (t1 - t0) / 4000;
            sink[threadIdx.x] = a;
        }
        return;
    }
    if (wave != 0) {                 // optional neighbours on the other SIMDs: spin on LDS like the helpers do
        if (wave < nwaves_active) { volatile double *p = lds; double s = 0; for (int i = 0; i < 20000; ++i) { s += p[lane]; __builtin_amdgcn_s_sleep(2); } sink[threadIdx.x] = s; }
        return;
    }
    double a = 1.0 + lane * 1e-9, b = 1.0000001, c = 0.5;
    long long t0, t1; int idx = 0; int x = lane;
#define TIC asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "+v"(a), "+v"(x) :: "memory");
#define TOC asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+v"(a), "+v"(x) :: "memory"); if (lane == 0) out[idx] = t1 - t0; ++idx;
    // 0: empty
    TIC TOC
    // 1: dependent f64 fma chain
    TIC
#pragma unroll
    for (int i = 0; i < N; ++i) a = __builtin_fma(a, b, c);
    TOC
    // 2: 4 independent f64 fma chains (N each)
    double a1 = a + 1, a2 = a + 2, a3 = a + 3;
    TIC
#pragma unroll
    for (int i = 0; i < N; ++i) { a = __builtin_fma(a, b, c); a1 = __builtin_fma(a1, b, c); a2 = __builtin_fma(a2, b, c); a3 = __builtin_fma(a3, b, c); }
    a += a1 + a2 + a3;
    TOC
    // 3: dependent f32 fma chain
    float f = (float)a, g = 1.0000001f, h = 0.5f;
    TIC
#pragma unroll
    for (int i = 0; i < N; ++i) f = __builtin_fmaf(f, g, h);
    a += f;
    TOC
    // 4: dependent int add chain
    TIC
#pragma unroll
    for (int i = 0; i < N; ++i) x = x * 3 + i;
    TOC
    a += x;
    // 5: dependent v_rcp_f64 chain
    TIC
#pragma unroll
    for (int i = 0; i < 64; ++i) a = __builtin_amdgcn_rcp(a) + 1.0;
    TOC
    // 6: readlane -> use chain (double)
    TIC
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        const int lo = __builtin_amdgcn_readlane(__double2loint(a), i & 63), hi = __builtin_amdgcn_readlane(__double2hiint(a), i & 63);
        a = a + __hiloint2double(hi, lo);
    }
    TOC
    // 7: dependent LDS read chain (pointer chase)
    int p = lane;
    TIC
#pragma unroll
    for (int i = 0; i < 64; ++i) p = (int)lds[p & 4095];
    x += p;
    TOC
    // 8: DPP row_shr add chain (f64)
    TIC
#pragma unroll
    for (int i = 0; i < 64; ++i) {
        int lo = __builtin_amdgcn_update_dpp(0, __double2loint(a), 0x111, 0xF, 0xF, false), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(a), 0x111, 0xF, 0xF, false);
        a += __hiloint2double(hi, lo);
    }
    TOC
    // 9: taken uniform branches (a loop that is not unrolled: one backward branch per iteration, 2 VALU inside)
    TIC
#pragma unroll 1
    for (int i = 0; i < N; ++i) { a = __builtin_fma(a, b, c); asm volatile("" : "+v"(a)); }
    TOC
    // 10: dependent global load chain (pointer chase through a 64 MB array: far memory), 32 hops
    long long q = lane * 8;
    TIC
#pragma unroll 1
    for (int i = 0; i < 32; ++i) q = (long long)gsrc[q & ((1 << 23) - 1)];
    x += (int)q;
    TOC
    // 11: same addresses again (now cached in L2 / L1)
    q = lane * 8;
    TIC
#pragma unroll 1
    for (int i = 0; i < 32; ++i) q = (long long)gsrc[q & ((1 << 23) - 1)];
    x += (int)q;
    TOC
    // 12: v_cndmask + compare chains (f64 select): per iteration cmp + 2 cndmask
    TIC
#pragma unroll
    for (int i = 0; i < N; ++i) a = (a > 1.5) ? a * 0.5 : a + 0.75;
    TOC
    // 13: exec-mask divergent if (s_and_saveexec): per iteration
    TIC
#pragma unroll
    for (int i = 0; i < 64; ++i) { if ((lane + i) & 1) a = __builtin_fma(a, b, c); asm volatile("" : "+v"(a)); }
    TOC
    // 14: ballot + ffs + readlane
    TIC
#pragma unroll
    for (int i = 0; i < 64; ++i) { unsigned long long m = __ballot(a > (double)i); int fl = m ? __ffsll((long long)m) - 1 : 0; a += (double)__builtin_amdgcn_readlane(x, fl); }
    TOC
    // 15: sqrt/rsq f64 chain
    TIC
#pragma unroll
    for (int i = 0; i < 64; ++i) a = __builtin_amdgcn_rsq(a) + 1.5;
    TOC
    // 16: ds_write + ds_read same address round trip x64
    TIC
#pragma unroll
    for (int i = 0; i < 64; ++i) { lds[lane + 64 * (i & 7)] = a; asm volatile("" ::: "memory"); a += lds[(lane ^ 1) + 64 * (i & 7)]; }
    TOC
    // 17: 64 independent LDS reads (throughput)
    TIC
    { double s = 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) s += lds[lane + 64 * (i & 31)];
    a += s; }
    TOC
    // 18: the resolver's column update arithmetic, one column: D, 1/D, q, log1p, exp (x16, each feeding the next)
    TIC
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
        const double acc = a * 1e-3, cdv = 0.7, Dt = 1.0 + acc;
        const double invD = fm_div(1.0, Dt);
        const double qv = fma(-(acc * acc), invD, cdv) - 0.01;
        const double f = qv * 0.05;
        const double rcf = 1.0 * fm_rsqrt(fabs(Dt));
        const double l1 = fm_log1p_small(f);
        a = fm_exp((2.0 - 1.5) - 30.0 * l1) * rcf;
        asm volatile("" : "+v"(a));
    }
    TOC
    // 19: two columns side by side (x16)
    { double a2 = a + 0.5;
    TIC
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
        double r2[2] = {a, a2};
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const double acc = r2[k] * 1e-3, cdv = 0.7, Dt = 1.0 + acc;
            const double invD = fm_div(1.0, Dt);
            const double qv = fma(-(acc * acc), invD, cdv) - 0.01;
            const double f = qv * 0.05;
            const double rcf = 1.0 * fm_rsqrt(fabs(Dt));
            const double l1 = fm_log1p_small(f);
            r2[k] = fm_exp((2.0 - 1.5) - 30.0 * l1) * rcf;
        }
        a = r2[0]; a2 = r2[1];
        asm volatile("" : "+v"(a), "+v"(a2));
    }
    a += a2;
    TOC }
    // 20: libm log + exp (x16)
    TIC
#pragma unroll 1
    for (int i = 0; i < 16; ++i) { a = exp(0.5 - 30.0 * log(1.0 + a * 1e-3)); asm volatile("" : "+v"(a)); }
    TOC
    // 21: s_barrier round trip with 4 waves?  (only wave 0 here: n/a) -- wave_scan-like: 4 dpp adds + 3 readlane pairs
    TIC
#pragma unroll 1
    for (int i = 0; i < 16; ++i) {
        double v = a;
#pragma unroll
        for (int sft = 0; sft < 4; ++sft) {
            int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x111, 0xF, 0xF, false), hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x111, 0xF, 0xF, false);
            v += __hiloint2double(hi, lo);
        }
        const double t0 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 15), __builtin_amdgcn_readlane(__double2loint(v), 15));
        const double t1 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 31), __builtin_amdgcn_readlane(__double2loint(v), 31));
        const double t2 = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 47), __builtin_amdgcn_readlane(__double2loint(v), 47));
        a = v + (lane < 16 ? 0.0 : (lane < 32 ? t0 : (lane < 48 ? t0 + t1 : t0 + t1 + t2))) * 1e-9;
        asm volatile("" : "+v"(a));
    }
    TOC
    sink[threadIdx.x] = a;
}
int main() {
    long long *out; double *sink, *g;
    hipMalloc(&out, 64 * 8); hipMalloc(&sink, 1024 * 8); hipMalloc(&g, (size_t)(1 << 23) * 8);
    // pointer chase: g[i] = (i * 1048583 + 12345) mod 2^23, as double
    double *h = (double *)malloc((size_t)(1 << 23) * 8);
    for (long long i = 0; i < (1 << 23); ++i) h[i] = (double)((i * 1048583ll + 12345ll) & ((1 << 23) - 1));
    hipMemcpy(g, h, (size_t)(1 << 23) * 8, hipMemcpyHostToDevice);
    const char *names[] = {"empty (s_memtime pair)", "dependent f64 fma x256", "4 independent f64 fma chains x256 (1024 instr)", "dependent f32 fma x256",
                           "dependent int mad x256", "dependent rcp_f64 + add x64", "readlane(2) + f64 add x64", "dependent LDS read (f64 -> int) x64",
                           "dpp(2) + f64 add x64", "loop with backward branch x256 (fma inside)", "dependent far global load x32", "dependent cached global load x32",
                           "f64 cmp + mul + add + 2 cndmask x256", "divergent if around one fma x64", "ballot + ffs + readlane + cvt + add x64", "dependent rsq_f64 + add x64", "ds_write -> ds_read (other lane) + add x64", "64 independent LDS reads + adds", "column update arithmetic, one column x16", "column update arithmetic, two columns side by side x16", "libm exp(.. log(..)) x16", "wave scan (4 dpp adds + 3 readlane pairs) x16"};
    for (int nw : {1, 4, 6, 7, 8}) {
        hipMemset(out, 0, 64 * 8);
        hipLaunchKernelGGL(k, dim3(1), dim3(256), 0, 0, out, sink, g, nw);
        hipDeviceSynchronize();
        long long r[64]; hipMemcpy(r, out, 64 * 8, hipMemcpyDeviceToHost);
        printf("== %d wavefront(s) active in the workgroup\n", nw);
        if (nw > 4) {
            printf("== %d wavefronts of the workgroup computing beside wavefront 0: column update arithmetic x16 on wavefront 0: %lld cycles (one at a time: see above); "
                   "per iteration on the neighbours: %lld %lld %lld\n", nw - 5, r[18], r[41], r[42], r[43]);
            continue;
        }
        for (int i = 0; i < 22; ++i) printf("%2d %-52s %8lld cycles\n", i, names[i], r[i]);
    }
    return 0;
}

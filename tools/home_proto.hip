// Development probe for home_kernel (kernels_home.hip): how fast can 1e6 rows of X, gathered by an index that is
// grouped by home, be pushed through ONE triangular quadratic form each (2 NJ (NJ + 1) v_mfma_f64_16x16x4 per 16
// rows, factor fragments in LDS) -- with the rows staged through an LDS ring by LDS-DMA (global_load_lds_dwordx4,
// whole 512-byte rows, XOR-swizzled on the source side) instead of straight into A-operand registers.
//   hipcc --offload-arch=gfx950 -O3 -o tools/home_proto tools/home_proto.hip && tools/home_proto [N] [variant ...]
// Variants: reg (the shipped kernel's scheme: NS = 2 register slots per wavefront, 2 workgroups of 4 waves per CU),
//           dma<NW,R> (one workgroup of NW waves per CU, R ring slots of 8 KB per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define LDS_AS __attribute__((address_space(3)))
typedef double v4d __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));
typedef double d2a __attribute__((ext_vector_type(2)));

constexpr int D = 64, NJ = 4, NF = 2 * NJ * (NJ + 1), NKK = 16, NJ8 = 8;

template <int CTRL>
__device__ __forceinline__ double dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_sum4(const double (&a)[4], int lane) {
    const bool p = lane & 1, q = lane & 2;
    const double k0 = p ? a[1] : a[0], s0 = p ? a[0] : a[1];
    const double k1 = p ? a[3] : a[2], s1 = p ? a[2] : a[3];
    const double b0 = k0 + dpp<0xB1>(s0);
    const double b1 = k1 + dpp<0xB1>(s1);
    const double kk = q ? b1 : b0, ss = q ? b0 : b1;
    double c = kk + dpp<0x4E>(ss);
    c += dpp<0x124>(c);
    c += dpp<0x128>(c);
    return c;
}

// the matrix work of one tile: xf = the lane's 16 A entries, Bf = fragments in LDS, returns the row sums in lanes lr < 4
__device__ __forceinline__ double tile_forms(const double (&xf)[NKK], LDS_AS const double *wf, int lane) {
    constexpr int LRING = 4;
    double ringk[LRING];
#pragma unroll
    for (int i = 0; i < LRING; ++i) ringk[i] = wf[i * 64];
    double qp[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int J = 0; J < NJ; ++J) {
        v4d acc = (v4d){0.5, 0.5, 0.5, 0.5};
#pragma unroll
        for (int kk = 0; kk < 4 * (J + 1); ++kk) {
            const int f = 2 * J * (J + 1) + kk;
            const double bfr = ringk[f % LRING];
            if (f + LRING < NF) ringk[f % LRING] = wf[(f + LRING) * 64];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xf[kk], bfr, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) qp[r] = fma(acc[r], acc[r], qp[r]);
    }
    return row_sum4(qp, lane);
}

// ---------------------------------------------------------------------------------------------------------------
// reg: rows straight into A-operand registers, two slots
template <int TAIL, int TAILF32>
__global__ __launch_bounds__(256, 2) void reg_kernel(const double *__restrict__ X, const int *__restrict__ idx, long long nrows,
                                                     const double *__restrict__ Bg, double *__restrict__ out) {
    __shared__ __attribute__((aligned(16))) double lds[NF * 64 + 4 * 64];
    LDS_AS double *const Bf = (LDS_AS double *)lds;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    LDS_AS double *const sideQ = Bf + NF * 64 + w * 64;
    for (int e = tid; e < NF * 64; e += 256) Bf[e] = Bg[e];
    __syncthreads();
    const long long nblocks = (nrows + 255) >> 8;
    const int nb = (int)gridDim.x, b = (int)blockIdx.x;
    const int per = nb >> 3, rem = nb & 7, xcd = b & 7;
    const int lb = xcd * per + (xcd < rem ? xcd : rem) + (b >> 3);
    const long long bpw = (nblocks + nb - 1) / nb;
    const long long b0 = (long long)lb * bpw, b1 = b0 + bpw < nblocks ? b0 + bpw : nblocks;
    if (b0 >= b1) return;
    const int lr = lane & 15, lk = lane >> 4;
    auto row_of = [&](long long blk) { const long long k = blk * 256 + w * 64 + lane; return (blk < b1 && k < nrows) ? idx[k] : 0; };
    long long xo_cur = (long long)row_of(b0) * D, xo_next = (long long)row_of(b0 + 1) * D;
    double xt[2][NKK];
#define ISSUE(DST, XO, TN)                                                                  \
    {                                                                                       \
        const int src_ = 16 * (TN) + lr;                                                    \
        const long long o_ = ((long long)__shfl((int)((XO) >> 32), src_) << 32) | (unsigned int)__shfl((int)(XO), src_); \
        const d2 *__restrict__ xrow = (const d2 *)(X + o_ + 2 * lk);                        \
        _Pragma("unroll") for (int j = 0; j < NJ8; ++j) {                                   \
            const d2 v_ = __builtin_nontemporal_load(xrow + 4 * j);                         \
            DST[2 * j] = v_.x; DST[2 * j + 1] = v_.y;                                       \
        }                                                                                   \
    }
    ISSUE(xt[0], xo_cur, 0)
#pragma unroll 1
    for (long long blk = b0; blk < b1; ++blk) {
        const long long i_after = (long long)row_of(blk + 2) * D;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            asm volatile("" ::: "memory");
            const double (&xf)[NKK] = xt[t % 2];
            if (t + 1 < 4) { ISSUE(xt[(t + 1) % 2], xo_cur, t + 1) } else { ISSUE(xt[(t + 1) % 2], xo_next, 0) }
            const double v = tile_forms(xf, Bf + lane, lane);
            if (lr < 4) sideQ[16 * t + lk + 4 * lr] = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const long long k = blk * 256 + w * 64 + lane;
        double q = sideQ[lane];
        // a stand-in for the scalar tail: TAIL dependent FP64 FMAs (lane = row), TAILF32 FP32 ones
#pragma unroll
        for (int e = 0; e < TAIL; ++e) q = fma(q, 1.0000001, 1e-9);
        float qf = (float)q;
#pragma unroll
        for (int e = 0; e < TAILF32; ++e) qf = fmaf(qf, 1.0000001f, 1e-9f);
        q += (double)qf;
        if (k < nrows && q < 0.0) out[k] = q;            // (never true: the tail of the real kernel stores for movers only)
        xo_cur = xo_next; xo_next = i_after;
    }
#undef ISSUE
}

// ---------------------------------------------------------------------------------------------------------------
// dma: every wavefront owns a ring of R tile slots (8 KB each) in LDS and fills it by LDS-DMA; no ordinary global load
// while DMAs are in flight (hipcc would wait vmcnt(0) for it): the row indices come by DMA too.
__device__ __forceinline__ void glds16(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void glds4(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(N) : "memory"); }

template <int NW, int R, bool NT>
__global__ __launch_bounds__(64 * NW, 1) void dma_kernel(const double *__restrict__ X, const int *__restrict__ idx, long long nrows,
                                                         const double *__restrict__ Bg, double *__restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) double lds_dyn[];
    LDS_AS double *const L = (LDS_AS double *)lds_dyn;
    LDS_AS double *const Bf = L;                                      // [NF][64]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    LDS_AS double *const ring = Bf + NF * 64 + w * (R * 1024);        // [R][1024]: tile slot = 16 rows x 512 bytes, swizzled
    LDS_AS double *const sideQ = Bf + NF * 64 + NW * (R * 1024) + w * 64;
    LDS_AS int *const idxb = (LDS_AS int *)(Bf + NF * 64 + NW * (R * 1024) + NW * 64) + w * (4 * 64);   // [4][64] row indices of a group
    for (int e = tid; e < NF * 64; e += 64 * NW) Bf[e] = Bg[e];
    __syncthreads();
    // the workgroup's run of 64-row groups; wave w takes groups g0 + w, g0 + w + NW, ...
    const long long ngroups = (nrows + 63) >> 6;
    const int nb = (int)gridDim.x, b = (int)blockIdx.x;
    const int per = nb >> 3, rem = nb & 7, xcd = b & 7;
    const int lb = xcd * per + (xcd < rem ? xcd : rem) + (b >> 3);
    const long long gpw = (ngroups + nb - 1) / nb;
    const long long g0 = (long long)lb * gpw, g1 = g0 + gpw < ngroups ? g0 + gpw : ngroups;
    const long long my_groups = g0 + w < g1 ? (g1 - g0 - w + NW - 1) / NW : 0;
    if (my_groups == 0) return;
    const int lr = lane & 15, lk = lane >> 4;
    const unsigned ring_addr = (unsigned)(size_t)ring, idx_addr = (unsigned)(size_t)idxb;
    auto group_of = [&](long long m) { return g0 + w + m * NW; };          // the wave's m-th group
    // row indices of the wave's m-th group -> idxb[m & 3] (one DMA instruction; clamped beyond the end)
    auto issue_idx = [&](long long m) {
        long long k = group_of(m < my_groups ? m : my_groups - 1) * 64 + lane;
        if (k >= nrows) k = nrows - 1;
        glds4(idx + k, idx_addr + (unsigned)((m & 3) * 256));
    };
    // tile n (group n / 4, rows 16 (n % 4) ..) -> ring slot n % R: 8 instructions, two whole rows each
    auto issue_tile = [&](long long n) {
        const long long m = n >> 2;
        const int t = (int)(n & 3);
        LDS_AS const int *ib = idxb + (m & 3) * 64 + 16 * t;
        const unsigned dst = ring_addr + (unsigned)((n % R) * 8192);
        const int half = lane >> 5, s = lane & 31;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int r = 2 * j + half;                       // row of the tile
            const long long i = ib[r];
            const int p = s ^ r;                              // the 16-byte piece of that row this lane fetches
            glds16(X + i * D + 2 * p, dst + (unsigned)(j * 1024));
        }
    };
    // prologue: indices of the first groups, then the first R tiles
    issue_idx(0); issue_idx(1); issue_idx(2);
    wait_vm<0>();
    for (int n = 0; n < R; ++n) issue_tile(n);
    const long long ntiles = my_groups * 4;
#pragma unroll 1
    for (long long n = 0; n < ntiles; ++n) {
        const int t = (int)(n & 3);
        wait_vm<8 * (R - 1)>();
        double xf[NKK];
        {
            LDS_AS const double *slot = ring + (n % R) * 1024 + lr * 64;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) {
                const int p = (4 * jj + lk) ^ lr;
                const d2a v = *(LDS_AS const d2a *)(slot + 2 * p);
                xf[2 * jj] = v.x; xf[2 * jj + 1] = v.y;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (t == 0) issue_idx((n >> 2) + 3);                  // (three groups ahead; its slot's last reader was group - 1)
        issue_tile(n + R < ntiles ? n + R : ntiles - 1 + 0 * n);   // (beyond the end: the last tile again, harmless)
        const double v = tile_forms(xf, Bf + lane, lane);
        if (lr < 4) sideQ[16 * t + lk + 4 * lr] = v;
        if (t == 3) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const long long k = group_of(n >> 2) * 64 + lane;
            const double q = sideQ[lane];
            if (k < nrows && q < 0.0) out[k] = q;
        }
    }
    wait_vm<0>();
}

// check kernel: the same forms, written out (for comparing the variants' arithmetic)
template <int NW, int R>
__global__ __launch_bounds__(64 * NW, 1) void dma_check(const double *X, const int *idx, long long nrows, const double *Bg, double *out);

template <typename F>
static double time_us(F f, int reps = 20) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    f(); f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return 1e3 * ms / reps;
}


struct Variant { const char *name; double (*run)(const double *, const int *, long long, const double *, double *, int); };

template <int NW, int R, bool NT>
static double run_dma(const double *X, const int *idx, long long N, const double *Bg, double *out, int cus) {
    const int lds = (NF * 64 + NW * R * 1024 + NW * 64) * 8 + NW * 4 * 64 * 4;
    CK(hipFuncSetAttribute((const void *)dma_kernel<NW, R, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    return time_us([&] { hipLaunchKernelGGL((dma_kernel<NW, R, NT>), dim3(cus), dim3(64 * NW), lds, 0, X, idx, N, Bg, out); }, 100);
}
template <int TAIL, int TAILF32>
static double run_reg(const double *X, const int *idx, long long N, const double *Bg, double *out, int cus) {
    return time_us([&] { hipLaunchKernelGGL((reg_kernel<TAIL, TAILF32>), dim3(2 * cus), dim3(256), 0, 0, X, idx, N, Bg, out); }, 100);
}

int main(int argc, char **argv) {
    const long long N = argc > 1 ? atoll(argv[1]) : 1000000;
    double *X, *out, *Bg;
    int *idx_s;
    CK(hipMalloc(&X, (size_t)N * D * 8));
    CK(hipMalloc(&out, (size_t)N * 8));
    CK(hipMalloc(&Bg, NF * 64 * 8));
    std::mt19937 rng(1);
    {
        std::vector<double> h((size_t)N * D);
        std::normal_distribution<double> nd(0.0, 1.0);
        for (auto &v : h) v = nd(rng);
        CK(hipMemcpy(X, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        std::vector<double> bh(NF * 64);
        for (auto &v : bh) v = 0.05 * nd(rng);
        CK(hipMemcpy(Bg, bh.data(), bh.size() * 8, hipMemcpyHostToDevice));
    }
    std::vector<int> lab(N), ord(N);
    for (long long i = 0; i < N; ++i) lab[i] = (int)(rng() % 200);
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return lab[a] < lab[b]; });
    CK(hipMalloc(&idx_s, N * 4)); CK(hipMemcpy(idx_s, ord.data(), N * 4, hipMemcpyHostToDevice));
    int cus = 256;
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    printf("N = %lld, D = %d, %d CUs; algorithmic row bytes %.1f MB; 100 launches back to back per figure\n", N, D, cus, (double)N * D * 8 / 1e6);
    const Variant vs[] = {
        {"reg  tail 0", run_reg<0, 0>},
        {"reg  tail 100 f64 fma", run_reg<100, 0>},
        {"reg  tail 300 f64 fma", run_reg<300, 0>},
        {"reg  tail 300 f32 fma", run_reg<0, 300>},
        {"dma  NW=12 R=1", run_dma<12, 1, false>},
    };
    for (int round = 0; round < 4; ++round)
        for (const Variant &v : vs) {
            const double us = v.run(X, idx_s, N, Bg, out, cus);
            printf("round %d  %-30s : %8.1f us  %6.2f TB/s\n", round, v.name, us, (double)N * D * 8 / us / 1e6);
        }
    return 0;
}

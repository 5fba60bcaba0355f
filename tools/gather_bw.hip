// Development probe: what a gather of whole rows of X (N x D doubles, row = D * 8 bytes contiguous) can reach
// on this box, against a plain stream of the same bytes.  Every wavefront reads R rows per step, each row one
// wave-wide load (lane l: entries l, l + 64, ...), sums them (so nothing is optimised away).
//   hipcc --offload-arch=gfx950 -O3 -o tools/gather_bw tools/gather_bw.hip && tools/gather_bw [N] [D]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int R, int NP, bool NT>
__global__ __launch_bounds__(256) void gather_kernel(const double *__restrict__ X, const int *__restrict__ idx, long long nrows,
                                                     int D, double *out, int persistent) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long ngroups = (nrows + R - 1) / R;
    long long g0 = wave, g1 = wave + 1, gs = 1;
    if (persistent) { const long long per = (ngroups + nwaves - 1) / nwaves; g0 = wave * per; g1 = g0 + per < ngroups ? g0 + per : ngroups; }
    double acc = 0.0;
    for (long long g = g0; g < g1; g += gs) {
        if (g >= ngroups) break;
        int my = 0;
        if (lane < R && g * R + lane < nrows) my = idx[g * R + lane];
        double v[R][NP];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int i = __builtin_amdgcn_readlane(my, r);
            const double *row = X + (long long)i * D;
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                const unsigned l = p * 64 + lane;
                v[r][p] = NT ? __builtin_nontemporal_load(row + l) : row[l];
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int p = 0; p < NP; ++p) acc += v[r][p];
    }
    if (acc == 1.2345e-300) out[0] = acc;
}

// The same rows fetched straight into MFMA A-fragment order: lane (lr = lane % 16, lk = lane / 16) takes 16 bytes
// of row lr at column 8 j + 2 lk, j = 0 .. D/8 - 1 (64 contiguous bytes per row and instruction); T tiles of 16
// rows in flight per wavefront.
template <int T, int NJ8, bool QUAD>
__global__ __launch_bounds__(256) void frag_kernel(const double *__restrict__ X, const int *__restrict__ idx, long long nrows,
                                                   int D, double *out) {
    // QUAD: four consecutive lanes share a row (64 contiguous bytes per quad) instead of lanes l, l + 16, l + 32, l + 48
    const int lane = threadIdx.x & 63, lr = QUAD ? lane >> 2 : lane & 15, lk = QUAD ? lane & 3 : lane >> 4;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nwaves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long ntiles = (nrows + 15) / 16;
    const long long per = (ntiles + nwaves - 1) / nwaves;
    const long long t0 = wave * per, t1 = t0 + per < ntiles ? t0 + per : ntiles;
    double acc = 0.0;
    for (long long t = t0; t < t1; t += T) {
        typedef double d2v __attribute__((ext_vector_type(2)));
        d2v v[T][NJ8];
#pragma unroll
        for (int tt = 0; tt < T; ++tt) {
            const long long r = (t + tt) * 16 + lr;
            const int i = (t + tt < t1 && r < nrows) ? idx[r] : 0;
            const d2v *row = (const d2v *)(X + (long long)i * D) + lk;
#pragma unroll
            for (int j = 0; j < NJ8; ++j) v[tt][j] = __builtin_nontemporal_load(row + 4 * j);
        }
#pragma unroll
        for (int tt = 0; tt < T; ++tt)
#pragma unroll
            for (int j = 0; j < NJ8; ++j) acc += v[tt][j].x + v[tt][j].y;
    }
    if (acc == 1.2345e-300) out[0] = acc;
}

__global__ __launch_bounds__(256) void stream_kernel(const double2 *__restrict__ X, long long n2, double *out) {
    double acc = 0.0;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n2; k += (long long)gridDim.x * blockDim.x) {
        const double2 v = X[k];
        acc += v.x + v.y;
    }
    if (acc == 1.2345e-300) out[0] = acc;
}

template <typename F>
static double time_ms(F f, int reps = 10) {
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
    return ms / reps;
}

template <int R, int NP, bool NT>
static void run(const double *X, const int *idx, long long N, int D, double *out, const char *what) {
    const double bytes = (double)N * D * 8.0;
    // one group per wave
    {
        const long long ngroups = (N + R - 1) / R;
        const unsigned gx = (unsigned)((ngroups + 3) / 4);
        const double ms = time_ms([&] { hipLaunchKernelGGL((gather_kernel<R, NP, NT>), dim3(gx), dim3(256), 0, 0, X, idx, N, D, out, 0); });
        printf("%-10s R=%2d nt=%d one group per wave      : %8.1f us  %6.2f TB/s\n", what, R, (int)NT, ms * 1e3, bytes / ms / 1e9);
    }
    for (int wg : {256, 512, 1024, 2048}) {
        const double ms = time_ms([&] { hipLaunchKernelGGL((gather_kernel<R, NP, NT>), dim3(wg), dim3(256), 0, 0, X, idx, N, D, out, 1); });
        printf("%-10s R=%2d nt=%d persistent %4d workgroups : %8.1f us  %6.2f TB/s\n", what, R, (int)NT, wg, ms * 1e3, bytes / ms / 1e9);
    }
}

int main(int argc, char **argv) {
    const long long N = argc > 1 ? atoll(argv[1]) : 1000000;
    const int D = argc > 2 ? atoi(argv[2]) : 64;
    double *X, *out;
    int *idx_r, *idx_s, *idx_i;
    CK(hipMalloc(&X, (size_t)N * D * 8));
    CK(hipMemset(X, 0, (size_t)N * D * 8));
    CK(hipMalloc(&out, 8));
    std::vector<int> id(N);
    std::iota(id.begin(), id.end(), 0);
    CK(hipMalloc(&idx_i, N * 4)); CK(hipMemcpy(idx_i, id.data(), N * 4, hipMemcpyHostToDevice));
    std::mt19937 rng(1);
    std::shuffle(id.begin(), id.end(), rng);
    CK(hipMalloc(&idx_r, N * 4)); CK(hipMemcpy(idx_r, id.data(), N * 4, hipMemcpyHostToDevice));
    // "sorted by home": K = 200 homes, labels random per point: the gather order is the stable sort of a random label
    std::vector<int> lab(N), ord(N);
    for (long long i = 0; i < N; ++i) lab[i] = (int)(rng() % 200);
    std::iota(ord.begin(), ord.end(), 0);
    std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return lab[a] < lab[b]; });
    CK(hipMalloc(&idx_s, N * 4)); CK(hipMemcpy(idx_s, ord.data(), N * 4, hipMemcpyHostToDevice));
    const double bytes = (double)N * D * 8.0;
    for (int wg : {1024, 2048, 4096, 16384}) {
        const double ms = time_ms([&] { hipLaunchKernelGGL(stream_kernel, dim3(wg), dim3(256), 0, 0, (const double2 *)X, N * D / 2, out); });
        printf("stream double2, %5d workgroups            : %8.1f us  %6.2f TB/s\n", wg, ms * 1e3, bytes / ms / 1e9);
    }
    if (D == 64) {
        run<16, 1, true>(X, idx_i, N, D, out, "identity");
        run<16, 1, true>(X, idx_r, N, D, out, "random");
        run<16, 1, false>(X, idx_r, N, D, out, "random");
        run<32, 1, true>(X, idx_r, N, D, out, "random");
        run<8, 1, true>(X, idx_r, N, D, out, "random");
        run<16, 1, true>(X, idx_s, N, D, out, "by-home");
        for (int wg : {512, 768, 1024, 2048}) {
            double ms = time_ms([&] { hipLaunchKernelGGL((frag_kernel<1, 8, false>), dim3(wg), dim3(256), 0, 0, X, idx_s, N, D, out); });
            printf("by-home fragment order T=1, %4d workgroups  : %8.1f us  %6.2f TB/s\n", wg, ms * 1e3, bytes / ms / 1e9);
            ms = time_ms([&] { hipLaunchKernelGGL((frag_kernel<2, 8, false>), dim3(wg), dim3(256), 0, 0, X, idx_s, N, D, out); });
            printf("by-home fragment order T=2, %4d workgroups  : %8.1f us  %6.2f TB/s\n", wg, ms * 1e3, bytes / ms / 1e9);
            ms = time_ms([&] { hipLaunchKernelGGL((frag_kernel<1, 8, true>), dim3(wg), dim3(256), 0, 0, X, idx_s, N, D, out); });
            printf("by-home quad order     T=1, %4d workgroups  : %8.1f us  %6.2f TB/s\n", wg, ms * 1e3, bytes / ms / 1e9);
        }
    } else if (D == 128) {
        run<16, 2, true>(X, idx_r, N, D, out, "random");
        run<16, 2, true>(X, idx_s, N, D, out, "by-home");
        for (int wg : {256, 512, 1024}) {
            double ms = time_ms([&] { hipLaunchKernelGGL((frag_kernel<1, 16, false>), dim3(wg), dim3(256), 0, 0, X, idx_s, N, D, out); });
            printf("by-home fragment order T=1, %4d workgroups  : %8.1f us  %6.2f TB/s\n", wg, ms * 1e3, bytes / ms / 1e9);
        }
    } else if (D == 16) {
        printf("(D = 16: a row is 128 bytes; lanes 16.. idle in this probe)\n");
    }
    return 0;
}

// The caller's uniform stream, generated on the device.
//
// The reference draws ONE random.random() per visit (utils/utils.py:13) from CPython's Mersenne
// Twister: MT19937, two 32-bit outputs a, b per double, (a >> 5) * 2^26 + (b >> 6)) / 2^53
// (genrand_res53).  At rest a sweep is shorter than producing and uploading its 8 MB of uniforms on
// the host, so the library can continue the caller's generator itself: the host hands over the 624
// state words and the position (random.getstate()), the device produces exactly the N doubles that N
// calls of random.random() would return, and hands back the state those calls would leave behind
// (random.setstate()).  Bit-identical by construction; tests/test_gpu_parity.py compares with the host.
//
// The recurrence x[k+624] = x[k+397] ^ twist(x[k], x[k+1]) is a chain through k -> k + 227: word k of
// the next block needs word k - 227 of the SAME block when k >= 227.  One workgroup regenerates a
// whole 624-word block per barrier: lane l < 227 computes the words l, l + 227 and l + 454 of the new
// block one after the other -- each is the (k - 227) term of the next, a register -- from the old
// block in LDS (double buffered); the last word also needs the new word 0, which its lane (169)
// recomputes from the old block instead of waiting for lane 0.  Tempered words go straight to global
// memory; a second kernel pairs them into doubles.
#include "bgmm_device.h"

__device__ __forceinline__ unsigned mt_temper(unsigned y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

__device__ __forceinline__ unsigned mt_twist(unsigned y0, unsigned y1) {
    const unsigned y = (y0 & 0x80000000u) | (y1 & 0x7fffffffu);
    return (y >> 1) ^ ((y1 & 1u) ? 0x9908b0dfu : 0u);
}

// key_io: 624 state words in / out;  pos_io: position in / out;  words: n_words tempered outputs.
__global__ __launch_bounds__(256) void mt19937_words_kernel(unsigned *__restrict__ key_io, int *__restrict__ pos_io,
                                                            unsigned *__restrict__ words, long long n_words) {
    __shared__ unsigned blk[2][624];
    const int tid = threadIdx.x;
    for (int k = tid; k < 624; k += 256) blk[0][k] = key_io[k];
    const long long pos = pos_io[0];
    __syncthreads();
    const long long E = pos + n_words;                        // one past the last consumed index of x
    // what is left of the current block
    for (long long k = pos + tid; k < 624 && k < E; k += 256) words[k - pos] = mt_temper(blk[0][k]);
    if (E <= 624) {
        if (tid == 0) pos_io[0] = (int)E;
        return;
    }
    const long long nblocks = (E - 1) / 624;                  // blocks to generate; the generator ends in the last
    const int l = tid < 227 ? tid : 0;
    const bool has_c = l <= 169;                              // word l + 454 exists
    const int ic0 = has_c ? l + 454 : 623, ic1 = (has_c && l < 169) ? l + 455 : 623;
    int cur = 0;
    unsigned *wp = words + (624 - pos) + l;                   // word l of block 1 in the output stream
    // every block but the last lies entirely inside the requested range: no bounds checks there
#define MT_BLOCK(CHECKED)                                                                          \
    {                                                                                              \
        const unsigned *__restrict__ old = blk[cur];                                               \
        unsigned *__restrict__ nw = blk[cur ^ 1];                                                  \
        const unsigned a0 = old[l], a1 = old[l + 1], a397 = old[l + 397];                          \
        const unsigned b0 = old[l + 227], b1 = old[l + 228];                                       \
        const unsigned c0 = old[ic0], c1_old = old[ic1];                                           \
        const unsigned z0 = old[0], z1 = old[1], z397 = old[397];                                  \
        const unsigned nA = a397 ^ mt_twist(a0, a1);                                               \
        const unsigned nB = nA ^ mt_twist(b0, b1);                                                 \
        const unsigned n0 = z397 ^ mt_twist(z0, z1); /* new word 0: lane 169's x[k+1] at k = 623 */ \
        const unsigned nC = nB ^ mt_twist(c0, l == 169 ? n0 : c1_old);                             \
        if (tid < 227) {                                                                           \
            nw[l] = nA;                                                                            \
            nw[l + 227] = nB;                                                                      \
            if (has_c) nw[l + 454] = nC;                                                           \
            if (!(CHECKED) || wp < wend) wp[0] = mt_temper(nA);                                    \
            if (!(CHECKED) || wp + 227 < wend) wp[227] = mt_temper(nB);                            \
            if (has_c && (!(CHECKED) || wp + 454 < wend)) wp[454] = mt_temper(nC);                 \
        }                                                                                          \
        wp += 624;                                                                                 \
        cur ^= 1;                                                                                  \
        /* publish the LDS words only: the global stores need not have landed */                   \
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");                            \
    }
    const unsigned *wend = words + n_words;
    for (long long b = 1; b < nblocks; ++b) MT_BLOCK(false)
    MT_BLOCK(true)
#undef MT_BLOCK
    __syncthreads();
    for (int k = tid; k < 624; k += 256) key_io[k] = blk[cur][k];
    if (tid == 0) pos_io[0] = (int)(E - 624 * nblocks);
}

__global__ void mt19937_doubles_kernel(const unsigned *__restrict__ words, double *__restrict__ u, long long n,
                                       int *__restrict__ zero_flag) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned a = words[2 * i] >> 5, b = words[2 * i + 1] >> 6;
    const double v = ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
    u[i] = v;
    if (v == 0.0) atomicOr(zero_flag, 1);
}

void launch_mt19937(unsigned *key_io, int *pos_io, unsigned *words, double *u, long long n, int *zero_flag,
                    hipStream_t st) {
    hipLaunchKernelGGL(mt19937_words_kernel, dim3(1), dim3(256), 0, st, key_io, pos_io, words, 2 * n);
    hipLaunchKernelGGL(mt19937_doubles_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, words, u, n,
                       zero_flag);
}

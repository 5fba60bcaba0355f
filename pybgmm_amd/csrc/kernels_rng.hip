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
// The recurrence x[k+624] = x[k+397] ^ twist(x[k], x[k+1]) yields 227 new words from the previous
// 624 independently of each other: one workgroup slides over the sequence, 227 words and one barrier
// per step, tempering and storing them as it goes; a second kernel pairs the words into doubles.
#include "bgmm_device.h"

static constexpr int kMtRing = 2048;

__device__ __forceinline__ unsigned mt_temper(unsigned y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// key_io: 624 state words in / out;  pos_io: position in / out;  words: 2 n tempered outputs.
// Thread t of a step computes x[P + t], P = words known so far.  Its x[.-227] term is its own output
// of the previous step (a register); the two x[.-624], x[.-623] terms were written at least two steps
// ago, so they are fetched from LDS one step ahead, and the only thing on the critical path of a step
// is the barrier that publishes its 227 new words.
__global__ __launch_bounds__(256) void mt19937_words_kernel(unsigned *__restrict__ key_io, int *__restrict__ pos_io,
                                                            unsigned *__restrict__ words, long long n_words) {
    __shared__ unsigned ring[kMtRing];
    const int tid = threadIdx.x;
    for (int k = tid; k < 624; k += 256) ring[k] = key_io[k];
    const long long pos = pos_io[0];
    __syncthreads();
    const long long E = pos + n_words;                        // one past the last consumed index of x
    // what is left of the current block
    for (long long k = pos + tid; k < 624 && k < E; k += 256) words[k - pos] = mt_temper(ring[k]);
    if (E <= 624) {
        if (tid == 0) pos_io[0] = (int)E;
        return;
    }
    const long long b = (E - 1) / 624;                        // block the generator ends in
    const long long target = 624 * (b + 1);
    const int t = tid < 227 ? tid : 0;
    unsigned prev = ring[397 + t];                            // x[P - 227 + t] at P = 624
    unsigned o0 = ring[t], o1 = ring[t + 1];                  // x[P - 624 + t], x[P - 623 + t]
    for (long long produced = 624; produced < target; produced += 227) {
        const long long left = target - produced;
        const int n = left < 227 ? (int)left : 227;
        const unsigned y = (o0 & 0x80000000u) | (o1 & 0x7fffffffu);
        const unsigned xn = prev ^ (y >> 1) ^ ((o1 & 1u) ? 0x9908b0dfu : 0u);
        // the old terms of the NEXT step: indices < produced - 169, published by earlier barriers
        const long long kn = produced + 227 - 624 + t;
        o0 = ring[kn & (kMtRing - 1)];
        o1 = ring[(kn + 1) & (kMtRing - 1)];
        if (tid < n) {
            const long long idx = produced + tid;
            ring[idx & (kMtRing - 1)] = xn;
            if (idx < E) words[idx - pos] = mt_temper(xn);
        }
        prev = xn;
        // publish the LDS words only: the global stores above need not have landed (a plain
        // __syncthreads() would also wait for them, every step)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    __syncthreads();
    for (int k = tid; k < 624; k += 256) key_io[k] = ring[(624 * b + k) & (kMtRing - 1)];
    if (tid == 0) pos_io[0] = (int)(E - 624 * b);
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

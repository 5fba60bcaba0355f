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
// memory; a second kernel pairs them into doubles.  A long request is cut into chains of chain_blocks() blocks
// that run side by side from jumped-ahead states (below).
#include "bgmm_device.h"

#include <mutex>
#include <vector>

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

// (n_per: doubles per sweep of a request that spans several -- zero_flag[j] tells whether sweep j's hold an exact 0)
__global__ void mt19937_doubles_kernel(const unsigned *__restrict__ words, double *__restrict__ u, long long n, long long n_per,
                                       int *__restrict__ zero_flag) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned a = words[2 * i] >> 5, b = words[2 * i + 1] >> 6;
    const double v = ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
    u[i] = v;
    if (v == 0.0) atomicOr(zero_flag + i / n_per, 1);
}

// ------------------------------------------------------------------------------------------
// More than one workgroup on the caller's stream: jump-ahead.
//
// The generator's state s_k = (top bit of x[k], x[k+1 .. k+623]) moves by a linear map F over GF(2) whose minimal
// polynomial phi has degree 19937.  With g_J = t^J mod phi:  s_J = g_J(F) s_0 = sum_i g_J[i] s_i, i.e.
//     x[J + w] = XOR over the set coefficients i of g_J of x[i + w]        (w = 1 .. 623 in full, w = 0: top bit)
// -- the state J words ahead is a GF(2) convolution of the first 19937 + 623 words of the stream with the
// coefficient bits of g_J.  One workgroup produces those words (they are the first 33 blocks of the sweep's own
// uniforms), mt19937_jump_kernel forms the states cb, 2 cb, ... blocks ahead (cb = chain_blocks()) for all chains at
// once, and every chain regenerates its cb blocks as before.  The low 31 bits of a jumped state's word 0
// are not determined -- nor needed: that block is the LAST one of the chain in front, which emits it from the
// recurrence; the jumped copy only seeds the blocks behind it.  phi comes from Berlekamp-Massey on one output bit
// (checked: degree 19937), the g_J from shift-and-reduce / multiply-and-reduce on the host, once per process.
// ------------------------------------------------------------------------------------------
// 624-word blocks per chain (>= 33: the first chain emits the convolution's input).  The jump costs 19 937 x 624 word
// operations per chain -- at 64 blocks a chain it was the generator's largest kernel (190 us per four sweeps of C4, on every
// compute unit); a chain's step is ~0.9 us, so the chains must still finish inside the sweeps they run beside.  Measured at C4
// on one box (profiles/r06/perm_rounds_experiments.txt, last paragraph), sweeps/s and home_kernel's time beside the generator:
// 64 blocks, batches of 4e6 doubles: 5 973 - 5 999, 0.128 ms;  128 / 4e6: 6 326 - 6 397, 0.133 - 0.134;  128 / 8e6: 6 296 -
// 6 352, 0.126 - 0.128;  256 / 8e6: 6 463 - 6 557, 0.130 - 0.131;  512: the generation no longer keeps up.
// BGMM_DEV_OPTIONS="mt_chain_blocks=..." (read once per process).
static int chain_blocks() {
    static const int v = [] { int b = bgmm_dev_option("mt_chain_blocks", 256); return b < 33 ? 33 : (b > 4096 ? 4096 : b); }();
    return v;
}
static constexpr int kPhiDeg = 19937;
static constexpr int kPolyWords = 624;                   // 32-bit words of a coefficient vector (19968 bits)
static constexpr int kRawWords = 33 * 624;               // untempered words the convolution reads: x[0 .. 20591]
int mt19937_raw_words() { return kRawWords; }
int mt19937_chain_blocks() { return chain_blocks(); }
static constexpr int kJumpSplits = 16, kJumpTargets = 4; // coefficient range per workgroup / chains per workgroup

// chain `p` of a sweep: seed = block p * cb of the stream (key_in for p = 0, a jumped state otherwise);
// emits the blocks (p cb, (p + 1) cb] that the request [pos, E) reaches into -- chain 0 also what is
// left of block 0 -- and, if the request ends in its range, the generator state the caller gets back.
// `mids`: a request that spans several sweeps (the look-ahead, api_inputs.hip) also wants the generator state at every
// sweep boundary inside it: mids.nb[j] / mids.pos[j] = the block that boundary lies in and the position in it; the chain
// that regenerates that block leaves it in key_mid[j] (blocks >= 1 only: such requests are long).
__global__ __launch_bounds__(256) void mt19937_chain_kernel(const unsigned *__restrict__ key_in, const unsigned *__restrict__ seeds,
                                                            int p_first, int pos, long long E, unsigned *__restrict__ words,
                                                            unsigned *__restrict__ raw, unsigned *__restrict__ key_out,
                                                            int *__restrict__ pos_out, unsigned *__restrict__ seed_next,
                                                            MtMids mids, unsigned *__restrict__ key_mid, int *__restrict__ pos_mid,
                                                            int untempered, int cb) {
    __shared__ unsigned blk[2][624];
    const int tid = threadIdx.x;
    const int p = p_first + (int)blockIdx.x;
    const unsigned *__restrict__ src = p == 0 ? key_in : seeds + (long long)p * 624;
    for (int k = tid; k < 624; k += 256) blk[0][k] = src[k];
    __syncthreads();
    const long long nb = (E - 1) / 624;                      // the block the request ends in
    if (p == 0) {
        for (long long k = pos + tid; k < 624 && k < E; k += 256) words[k - pos] = untempered ? blk[0][k] : mt_temper(blk[0][k]);
        if (raw) for (int k = tid; k < 624; k += 256) raw[k] = blk[0][k];
        if (nb == 0) {
            for (int k = tid; k < 624; k += 256) key_out[k] = blk[0][k];
            if (tid == 0) pos_out[0] = (int)E;
            return;
        }
    }
    const long long b_lo = (long long)p * cb + 1;
    long long b_hi = b_lo + cb - 1;
    if (b_hi > nb) b_hi = nb;
    if (b_lo > b_hi) return;
    const int l = tid < 227 ? tid : 0;
    const bool has_c = l <= 169;                              // word l + 454 exists
    const int ic0 = has_c ? l + 454 : 623, ic1 = (has_c && l < 169) ? l + 455 : 623;
    int cur = 0;
    const long long n_words = E - pos;
    for (long long b = b_lo; b <= b_hi; ++b) {
        const unsigned *__restrict__ old = blk[cur];
        unsigned *__restrict__ nw = blk[cur ^ 1];
        const unsigned a0 = old[l], a1 = old[l + 1], a397 = old[l + 397];
        const unsigned b0 = old[l + 227], b1 = old[l + 228];
        const unsigned c0 = old[ic0], c1_old = old[ic1];
        const unsigned z0 = old[0], z1 = old[1], z397 = old[397];
        const unsigned nA = a397 ^ mt_twist(a0, a1);
        const unsigned nB = nA ^ mt_twist(b0, b1);
        const unsigned n0 = z397 ^ mt_twist(z0, z1);     // new word 0: lane 169's x[k+1] at k = 623
        const unsigned nC = nB ^ mt_twist(c0, l == 169 ? n0 : c1_old);
        if (tid < 227) {
            nw[l] = nA;
            nw[l + 227] = nB;
            if (has_c) nw[l + 454] = nC;
            const long long o = 624 * b - pos + l;           // word l of block b in the output stream
            // (untempered: the words as the state holds them -- the permutation kernels temper on the fly and read the
            // generator state behind the words they consumed straight out of this array)
            if (o < n_words) words[o] = untempered ? nA : mt_temper(nA);
            if (o + 227 < n_words) words[o + 227] = untempered ? nB : mt_temper(nB);
            if (has_c && o + 454 < n_words) words[o + 454] = untempered ? nC : mt_temper(nC);
            if (raw && b < 33) {
                raw[624 * b + l] = nA; raw[624 * b + l + 227] = nB;
                if (has_c) raw[624 * b + l + 454] = nC;
            }
            for (int j = 0; j < mids.m; ++j)
                if (mids.nb[j] == b) {                       // (uniform: a sweep boundary lies in this block)
                    unsigned *__restrict__ km = key_mid + (long long)j * 624;
                    km[l] = nA; km[l + 227] = nB;
                    if (has_c) km[l + 454] = nC;
                    if (tid == 0) pos_mid[j] = mids.pos[j];
                }
        }
        cur ^= 1;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // (the LDS words only: global stores need not have landed)
    }
    if (b_hi == nb) {
        for (int k = tid; k < 624; k += 256) key_out[k] = blk[cur][k];
        if (tid == 0) pos_out[0] = (int)(E - 624 * nb);
    } else if (seed_next) {
        // (no jump polynomials: the chains run one after the other, each handing its last block to the next)
        for (int k = tid; k < 624; k += 256) seed_next[(long long)(p + 1) * 624 + k] = blk[cur][k];
    }
}

// seeds[p][w] ^= XOR over i in this workgroup's coefficient range with coef[p][i] set of raw[i + w]; thread = w,
// kJumpTargets chains per workgroup (blockIdx.y), coefficient range blockIdx.x.  seeds is zeroed beforehand.
__global__ __launch_bounds__(640) void mt19937_jump_kernel(const unsigned *__restrict__ raw, const unsigned *__restrict__ coef,
                                                           int n_chains, unsigned *__restrict__ seeds) {
    constexpr int IL = (kPhiDeg + kJumpSplits - 1) / kJumpSplits;     // coefficients per workgroup
    constexpr int ILp = (IL + 31) / 32 * 32;
    __shared__ unsigned xs[ILp + 640];
    const int i0 = (int)blockIdx.x * ILp;                    // (ranges aligned to coefficient words)
    const int w = threadIdx.x;
    for (int k = w; k < ILp + 640; k += 640) xs[k] = i0 + k < kRawWords ? raw[i0 + k] : 0u;
    __syncthreads();
    const int p0 = 1 + (int)blockIdx.y * kJumpTargets;
    unsigned acc[kJumpTargets];
#pragma unroll
    for (int t = 0; t < kJumpTargets; ++t) acc[t] = 0u;
    for (int c = 0; c < ILp / 32; ++c) {
        const int iw = (i0 >> 5) + c;
        if (iw >= kPolyWords) break;
        unsigned cw[kJumpTargets];
#pragma unroll
        for (int t = 0; t < kJumpTargets; ++t) cw[t] = p0 + t < n_chains ? coef[(long long)(p0 + t) * kPolyWords + iw] : 0u;
#pragma unroll
        for (int b = 0; b < 32; ++b) {
            const unsigned xv = xs[32 * c + b + w];
#pragma unroll
            for (int t = 0; t < kJumpTargets; ++t)          // acc ^= xv & mask in ONE v_bitop3_b32 (truth table a ^ (b & c)); the
                acc[t] = __builtin_amdgcn_bitop3_b32(acc[t], xv, 0u - ((cw[t] >> b) & 1u), 0x78);   // mask is scalar (s_bfe_i32)
        }
    }
    if (w < 624) {
#pragma unroll
        for (int t = 0; t < kJumpTargets; ++t)
            if (p0 + t < n_chains && acc[t]) atomicXor(&seeds[(long long)(p0 + t) * 624 + w], acc[t]);
    }
}

// ---- host: phi and the jump polynomials --------------------------------------------------------
namespace {
struct MtJumpTables {
    std::mutex mu;
    bool ready = false, failed = false;
    std::vector<int> phi_low;                        // exponents of phi's terms below the leading one
    std::vector<std::vector<uint64_t>> g;            // g[p] = t^(p * chain_blocks() * 624) mod phi, 312 x 64 bits; g[0] unused
};
MtJumpTables g_mtj;

constexpr int kW64 = 312;                            // 64-bit words of a residue (19968 bits)

void host_mt_block(unsigned *mt) {                   // one regeneration of the 624-word block (reference recurrence)
    for (int k = 0; k < 624; ++k) {
        const unsigned y = (mt[k] & 0x80000000u) | (mt[(k + 1) % 624] & 0x7fffffffu);
        mt[k] = mt[(k + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
}

// Berlekamp-Massey over GF(2) on bit 0 of the untempered stream: the minimal polynomial of F
bool build_phi(std::vector<int> &low) {
    const int n = kPhiDeg, len = 2 * n + 64;
    std::vector<unsigned char> sq((size_t)len);
    {
        unsigned mt[624];
        unsigned v = 19650218u;
        for (int k = 0; k < 624; ++k) { mt[k] = v; v = 1812433253u * (v ^ (v >> 30)) + (unsigned)(k + 1); }
        int have = 0;
        while (have < len) {
            host_mt_block(mt);
            for (int k = 0; k < 624 && have < len; ++k) sq[(size_t)have++] = (unsigned char)(mt[k] & 1u);
        }
    }
    const int W = (n + 64) / 64 + 2;
    std::vector<uint64_t> C((size_t)W, 0), B((size_t)W, 0), T((size_t)W, 0), Wn((size_t)W, 0);
    C[0] = 1; B[0] = 1;
    int L = 0, m = 1;
    for (int N = 0; N < 2 * n; ++N) {
        // window: bit i = s[N - i]
        for (int k = W - 1; k > 0; --k) Wn[(size_t)k] = (Wn[(size_t)k] << 1) | (Wn[(size_t)k - 1] >> 63);
        Wn[0] = (Wn[0] << 1) | (uint64_t)sq[(size_t)N];
        uint64_t acc = 0;
        const int lw = L / 64 + 1;
        for (int k = 0; k < lw && k < W; ++k) acc ^= C[(size_t)k] & Wn[(size_t)k];
        if (!(__builtin_popcountll(acc) & 1)) { ++m; continue; }
        const bool grow = 2 * L <= N;
        if (grow) T = C;
        {   // C ^= B << m
            const int ws = m / 64, bs = m % 64;
            for (int k = W - 1; k >= ws; --k) {
                uint64_t v = B[(size_t)(k - ws)] << bs;
                if (bs && k - ws - 1 >= 0) v |= B[(size_t)(k - ws - 1)] >> (64 - bs);
                C[(size_t)k] ^= v;
            }
        }
        if (grow) { L = N + 1 - L; B = T; m = 1; } else ++m;
    }
    if (L != n) return false;
    // phi(t) = t^L C(1/t): phi_j = c_{L - j}
    low.clear();
    for (int j = 0; j < n; ++j) {
        const int i = L - j;
        if ((C[(size_t)(i / 64)] >> (i % 64)) & 1ull) low.push_back(j);
    }
    return ((C[0] & 1ull) != 0);                     // (leading term of phi)
}

inline void flip_bit(std::vector<uint64_t> &v, int i) { v[(size_t)(i >> 6)] ^= 1ull << (i & 63); }
inline bool get_bit(const std::vector<uint64_t> &v, int i) { return (v[(size_t)(i >> 6)] >> (i & 63)) & 1ull; }

// v (degree < 2 * 19937) modulo phi, in place; the result sits in the low 312 words
void reduce_phi(std::vector<uint64_t> &v, const std::vector<int> &low) {
    for (int k = 2 * kPhiDeg; k >= kPhiDeg; --k) {
        if (!get_bit(v, k)) continue;
        flip_bit(v, k);
        const int sh = k - kPhiDeg;
        for (int e : low) flip_bit(v, sh + e);
    }
}

void mulmod_phi(const std::vector<uint64_t> &a, const std::vector<uint64_t> &b, const std::vector<int> &low,
                std::vector<uint64_t> &out) {
    // 64 aligned copies of b, then one XOR run per set bit of a
    static thread_local std::vector<uint64_t> sh;
    sh.assign((size_t)64 * (kW64 + 1), 0);
    for (int s = 0; s < 64; ++s)
        for (int k = 0; k <= kW64; ++k) {
            uint64_t v = k < kW64 ? b[(size_t)k] << s : 0;
            if (s && k > 0) v |= b[(size_t)k - 1] >> (64 - s);
            sh[(size_t)s * (kW64 + 1) + k] = v;
        }
    std::vector<uint64_t> prod((size_t)2 * kW64 + 2, 0);
    for (int wa = 0; wa < kW64; ++wa) {
        uint64_t m = a[(size_t)wa];
        while (m) {
            const int s = __builtin_ctzll(m);
            m &= m - 1;
            const uint64_t *src = &sh[(size_t)s * (kW64 + 1)];
            uint64_t *dst = &prod[(size_t)wa];
            for (int k = 0; k <= kW64; ++k) dst[k] ^= src[k];
        }
    }
    reduce_phi(prod, low);
    out.assign(prod.begin(), prod.begin() + kW64);
}

// coefficient vectors of t^(p J1) mod phi for p = 1 .. n_chains - 1 (J1 = chain_blocks() * 624 words); grows on demand
bool ensure_jump_polys(int n_chains) {
    std::lock_guard<std::mutex> guard(g_mtj.mu);
    if (g_mtj.failed) return false;
    if (!g_mtj.ready) {
        if (!build_phi(g_mtj.phi_low)) { g_mtj.failed = true; return false; }
        // g_1 = t^J1 mod phi: from t^19936, one multiplication by t (shift, conditional reduction) at a time
        std::vector<uint64_t> v((size_t)kW64 + 1, 0);
        flip_bit(v, kPhiDeg - 1);
        for (int step = kPhiDeg - 1; step < chain_blocks() * 624; ++step) {
            for (int k = kW64; k > 0; --k) v[(size_t)k] = (v[(size_t)k] << 1) | (v[(size_t)k - 1] >> 63);
            v[0] <<= 1;
            if (get_bit(v, kPhiDeg)) {
                flip_bit(v, kPhiDeg);
                for (int e : g_mtj.phi_low) flip_bit(v, e);
            }
        }
        v.resize((size_t)kW64);
        g_mtj.g.clear();
        g_mtj.g.push_back(std::vector<uint64_t>());
        g_mtj.g.push_back(v);
        g_mtj.ready = true;
    }
    while ((int)g_mtj.g.size() < n_chains) {
        std::vector<uint64_t> nx;
        mulmod_phi(g_mtj.g.back(), g_mtj.g[1], g_mtj.phi_low, nx);
        g_mtj.g.push_back(nx);
    }
    return true;
}
}  // namespace

int mt19937_chains_for(long long pos, long long n) {
    const long long E = pos + 2 * n;
    const long long nb = E > 0 ? (E - 1) / 624 : 0;
    return (int)((nb + chain_blocks() - 1) / chain_blocks());          // chains that have a block to emit (>= 1 once nb >= 1)
}

// coefficient words of chains 1 .. n_chains - 1 as the device wants them: [n_chains][624] uint32 (row 0 unused).
// False when the polynomial could not be established (the one-workgroup route carries on).
bool mt19937_jump_coefficients(int n_chains, std::vector<unsigned> &out) {
    if (!ensure_jump_polys(n_chains)) return false;
    out.assign((size_t)n_chains * kPolyWords, 0u);
    std::lock_guard<std::mutex> guard(g_mtj.mu);
    for (int p = 1; p < n_chains; ++p)
        for (int k = 0; k < kW64; ++k) {
            out[(size_t)p * kPolyWords + 2 * k] = (unsigned)(g_mtj.g[(size_t)p][(size_t)k] & 0xffffffffull);
            out[(size_t)p * kPolyWords + 2 * k + 1] = (unsigned)(g_mtj.g[(size_t)p][(size_t)k] >> 32);
        }
    return true;
}

// key_io / pos_io: the caller's generator (device copies), advanced by 2 n words; words: 2 n tempered outputs; u: the n
// doubles.  coef_dev == nullptr (or a single chain): one workgroup walks the whole stream.  Otherwise the first chain
// emits its blocks and the convolution's input (raw), the jump kernel seeds the other chains (seeds: zeroed here), and
// they run side by side.  key_out / pos_out: where the advanced state is left (may not alias key_io).
// n_sweeps > 1: the request is n_sweeps consecutive sweeps of n / n_sweeps doubles each; key_mid [n_sweeps][624] /
// pos_mid [n_sweeps] receive the generator state behind each of them, zero_flag has n_sweeps entries.
void launch_mt19937(const unsigned *key_in, int pos, unsigned *key_out, int *pos_out, unsigned *words, double *u, long long n,
                    int *zero_flag, const unsigned *coef_dev, int n_chains, unsigned *raw, unsigned *seeds, hipStream_t st,
                    int n_sweeps, unsigned *key_mid, int *pos_mid) {
    const long long E = (long long)pos + 2 * n;
    MtMids mids;
    mids.m = 0;
    const long long n_per = n_sweeps > 1 ? n / n_sweeps : n;
    if (n_sweeps > 1 && key_mid && pos_mid)
        for (int j = 0; j < n_sweeps && j < kMtMaxMids; ++j) {
            const long long Ej = (long long)pos + 2 * n_per * (j + 1);
            mids.nb[j] = (Ej - 1) / 624;
            mids.pos[j] = (int)(Ej - 624 * mids.nb[j]);
            mids.m = j + 1;
        }
    if (!coef_dev || n_chains < 2) {
        // (one chain per launch, in stream order: each leaves the state the next one starts from)
        const int chains = mt19937_chains_for(pos, n);
        for (int p = 0; p < (chains > 1 ? chains : 1); ++p)
            hipLaunchKernelGGL(mt19937_chain_kernel, dim3(1), dim3(256), 0, st, key_in, (const unsigned *)seeds, p, pos, E,
                               words, (unsigned *)nullptr, key_out, pos_out, seeds, mids, key_mid, pos_mid, 0, chain_blocks());
    } else {
        (void)hipMemsetAsync(seeds, 0, sizeof(unsigned) * 624 * (size_t)n_chains, st);
        hipLaunchKernelGGL(mt19937_chain_kernel, dim3(1), dim3(256), 0, st, key_in, (const unsigned *)seeds, 0, pos, E, words, raw,
                           key_out, pos_out, (unsigned *)nullptr, mids, key_mid, pos_mid, 0, chain_blocks());
        hipLaunchKernelGGL(mt19937_jump_kernel, dim3(kJumpSplits, (unsigned)((n_chains - 1 + kJumpTargets - 1) / kJumpTargets)),
                           dim3(640), 0, st, raw, coef_dev, n_chains, seeds);
        hipLaunchKernelGGL(mt19937_chain_kernel, dim3((unsigned)(n_chains - 1)), dim3(256), 0, st, key_in, (const unsigned *)seeds, 1,
                           pos, E, words, (unsigned *)nullptr, key_out, pos_out, (unsigned *)nullptr, mids, key_mid, pos_mid, 0, chain_blocks());
    }
    hipLaunchKernelGGL(mt19937_doubles_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, words, u, n, n_per, zero_flag);
}

// n_words UNTEMPERED words behind (key_in, pos) into `words` (the permutation kernels' input): the same chains, no doubles.
// spare_key / spare_pos: scratch for the state the chains leave (not used by the caller).
void launch_mt19937_raw(const unsigned *key_in, int pos, unsigned *words, long long n_words, const unsigned *coef_dev, int n_chains,
                        unsigned *raw, unsigned *seeds, unsigned *spare_key, int *spare_pos, hipStream_t st) {
    const long long E = (long long)pos + n_words;
    MtMids mids;
    mids.m = 0;
    if (!coef_dev || n_chains < 2) {
        const long long nb = E > 0 ? (E - 1) / 624 : 0;
        const int chains = (int)((nb + chain_blocks() - 1) / chain_blocks());
        for (int p = 0; p < (chains > 1 ? chains : 1); ++p)
            hipLaunchKernelGGL(mt19937_chain_kernel, dim3(1), dim3(256), 0, st, key_in, (const unsigned *)seeds, p, pos, E,
                               words, (unsigned *)nullptr, spare_key, spare_pos, seeds, mids, (unsigned *)nullptr, (int *)nullptr, 1, chain_blocks());
    } else {
        (void)hipMemsetAsync(seeds, 0, sizeof(unsigned) * 624 * (size_t)n_chains, st);
        hipLaunchKernelGGL(mt19937_chain_kernel, dim3(1), dim3(256), 0, st, key_in, (const unsigned *)seeds, 0, pos, E, words, raw,
                           spare_key, spare_pos, (unsigned *)nullptr, mids, (unsigned *)nullptr, (int *)nullptr, 1, chain_blocks());
        hipLaunchKernelGGL(mt19937_jump_kernel, dim3(kJumpSplits, (unsigned)((n_chains - 1 + kJumpTargets - 1) / kJumpTargets)),
                           dim3(640), 0, st, raw, coef_dev, n_chains, seeds);
        hipLaunchKernelGGL(mt19937_chain_kernel, dim3((unsigned)(n_chains - 1)), dim3(256), 0, st, key_in, (const unsigned *)seeds, 1,
                           pos, E, words, (unsigned *)nullptr, spare_key, spare_pos, (unsigned *)nullptr, mids, (unsigned *)nullptr,
                           (int *)nullptr, 1, chain_blocks());
    }
}

// chains a request of n_words words from position pos is cut into
int mt19937_chains_for_words(long long pos, long long n_words) {
    const long long E = pos + n_words;
    const long long nb = E > 0 ? (E - 1) / 624 : 0;
    return (int)((nb + chain_blocks() - 1) / chain_blocks());
}

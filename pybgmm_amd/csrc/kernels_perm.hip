// The visiting order of a pCRP sweep, drawn on the device.
//
// PCRPMM.collapsed_gibbs_sampler visits the data in `np.random.permutation(range(N))` (pcrpmm.py:86-91), a fresh one every
// sweep, drawn from the caller's legacy numpy stream.  On the host that is 9 ms per 1e6 points (17 ms for C5's 2e6) in
// front of a sweep of 0.5 - 1 ms.  This file produces the SAME permutation from the SAME stream on the GPU, bit for bit,
// and hands back the generator state numpy would be left in.
//
// What numpy does (legacy RandomState.permutation(n) = shuffle(arange(n)), mtrand.pyx / distributions.c):
//     for i = n-1 down to 1:   j = random_interval(i);   swap(x[i], x[j])
//     random_interval(max):    mask = smallest 2^k - 1 >= max;   repeat v = next_uint32() & mask until v <= max
// Two sequential chains -- which 32-bit words step i consumes depends on every rejection before it, and the swaps act on
// one array -- both restated here in parallel form:
//
// 1. DRAWS (perm_draw_kernel).  Inside a wavefront: with c = the number of words accepted so far in a run of 64 words,
//    word l is accepted iff (w_l & mask) <= i - c_l.  Starting from "every word with (w & mask) <= i is accepted" the wave
//    iterates acc <- ballot((w & mask) <= i - popcount(acc below me)); the iterates bracket the true set from above and
//    below in turn and lane l is exact once the lanes below it are, so the fixed point -- usually reached in two or three
//    rounds, because a word's fate depends on c only when it lies within 64 of the threshold -- is the sequential answer.
//    A run is cut behind the step that ends a mask's range (i = 2^(k-1)): the words behind it meet the next mask.
//    Across wavefronts the same idea once more: the stream is cut into segments of 1024 words, one wavefront each, and
//    the step a segment starts at (n - 1 minus what the segments in front of it accept) is taken from the previous
//    ROUND's counts; the rounds settle because a segment's count depends only weakly on where it starts.
//
// 2. SWAPS.  Position i is final once step i has run, and what it receives is what position J[i] held just before:
//       final[i] = (J[i] == i) ? V(i) : (pred(i) exists ? V(pred(i)) : J[i])
//    where V(i) = the value at position i just before step i, pred(i) = the next larger step with the same target as i
//    (the most recent earlier swap into that position), and V(i) = V(predV(i)) with predV(i) = the smallest step > i whose
//    target is i (or i itself, untouched, if there is none).  pred / predV fall out of a stable sort of the steps by
//    target (rocPRIM radix sort); the chains behind V are a handful of links long and every position walks its own.
// tests/test_perm_formulation.py holds the same formulation in numpy, checked against np.random on the CPU.
#include "bgmm_device.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <rocprim/device/device_radix_sort.hpp>

__device__ __forceinline__ unsigned perm_temper(unsigned y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

static constexpr int kPermRoundsDefault = 30;   // rounds of draws queued at a time (those behind the write pass return at once)
// (BGMM_DEV_OPTIONS perm_rounds, 3 .. 56: for the test that drives the "not settled yet, more rounds" repair)
static int perm_rounds_now() {
    static const int r = [] {
        const int v = bgmm_dev_option("perm_rounds", kPermRoundsDefault);
        return v < 3 ? 3 : (v > 56 ? 56 : v);
    }();
    return r;
}
// steps below this are served by ONE wavefront behind the rounds (perm_tail_kernel); BGMM_DEV_OPTIONS perm_tail_log2 overrides (10 .. 16)
static const int kPermTailLow = [] { const int v = bgmm_dev_option("perm_tail_log2", 14);
                                     return 1 << (v < 10 ? 10 : (v > 16 ? 16 : v)); }();
static constexpr int kPermSeg = 1024;      // words per segment (one wavefront, 4 KB of LDS)
int perm_segments(long long n_avail) { return (int)((n_avail + kPermSeg - 1) / kPermSeg); }
int perm_rounds() { return perm_rounds_now(); }
// the rounds serve the steps n - 1 .. perm_low(n), the tail the rest (all of them when n is small)
static int perm_low(int n) { return n - 1 >= kPermTailLow ? kPermTailLow : n; }

// raw: untempered MT19937 words following the caller's position, n_avail of them, cut into segments of kPermSeg words;
// one wavefront per segment, its words in LDS.  A round takes the step a segment starts at from the counts the segments in
// front of it hold at that moment and writes its own: segment 0 is exact at once, segment t once those in front of it are, and
// a segment's count depends only weakly on where it starts (a word's fate changes only if its value lies between the two
// thresholds), so the rounds settle -- a dozen or two of them, launched blindly; a segment whose start has not changed
// since it last ran keeps its count without running again.  The round in which no count changed had every segment at its
// true start; the round behind it writes the targets J[i] (i = 1 .. n-1) -- numpy's -- and out[0] = words consumed,
// out[1] = 0 (by the segment in which step 1 is served), and the rounds behind that return at once.
__global__ __launch_bounds__(256) void perm_draw_kernel(const unsigned *__restrict__ raw, long long n_avail, int n, int low, int round,
                                                        int rounds, int *cnt, int *__restrict__ seen, int *__restrict__ J,
                                                        long long *__restrict__ out, int *__restrict__ flags) {
    __shared__ unsigned ws[4][kPermSeg];
    // flags[r] = some count changed in round r.  The first round behind a round that changed nothing is the WRITE pass:
    // every segment runs from its (now true) start once more and leaves its targets in J; the rounds behind it return.
    const bool settled = round >= 2 && flags[round - 1] == 0;
    if (settled && round >= 3 && flags[round - 2] == 0) {
        if (blockIdx.x == 0 && threadIdx.x == 0) flags[round] = 0;
        return;
    }
    const bool write_pass = settled;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int t = (int)blockIdx.x * 4 + wv;
    const long long seg0 = (long long)t * kPermSeg;
    if (seg0 >= n_avail) return;
    const int len = (int)(seg0 + kPermSeg < n_avail ? kPermSeg : n_avail - seg0);
    // the step this segment starts at
    // (the counts are updated IN PLACE, read past the L1: a segment in front of this one that has already run in this
    // round is seen with its new count -- the rounds are chaotic iterations, which settle in about half as many launches
    // as rounds that only see the previous launch's counts; the test for "settled" -- a round in which no count changed,
    // hence every read saw the final value -- is unaffected)
    long long before = 0;
    for (int k = lane; k < t; k += 64) before += __hip_atomic_load(cnt + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
    const long long i0l = (long long)(n - 1) - before;
    const int i0 = i0l > 0 ? (int)i0l : 0;
    // seen[3 t ..]: the start this segment last ran from, what it accepted then, where it served step `low` (-1: not here).
    // From the same start it would do exactly the same again: its count and the targets it wrote stand.
    int *__restrict__ mine_seen = seen + 3 * (long long)t;
    int accepted = 0, end = -1;
    if (round >= 2 && !write_pass && mine_seen[0] == i0) {
        accepted = mine_seen[1];
        end = mine_seen[2];
    } else if (i0 >= low) {
        unsigned *__restrict__ W = ws[wv];
        for (int k = lane; k < len; k += 64) W[k] = perm_temper(raw[seg0 + k]);     // (all of the segment's loads in flight together)
        const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
        int i = i0;
        int p = 0;
        while (i >= low && p < len) {
            const unsigned mask = 0xffffffffu >> __builtin_clz((unsigned)i);
            const int lowi = (int)(mask >> 1) + 1;
            const int a = (p + lane < len) ? (int)(W[p + lane] & mask) : 0x7fffffff;
            unsigned long long acc = __ballot(a <= i);
            for (;;) {
                const int c = __builtin_popcountll(acc & below);
                const unsigned long long acc2 = __ballot(a <= i - c);
                if (acc2 == acc) break;
                acc = acc2;
            }
            const int c = __builtin_popcountll(acc & below);
            const int s = i - c;                                         // the step this word serves, if accepted
            const bool mine = (acc >> lane) & 1ull;
            const unsigned long long last = __ballot(mine && s == lowi);   // the step that ends this mask's range
            const int cut = last ? (int)__builtin_ctzll(last) + 1 : 64;
            if (write_pass && mine && lane < cut) J[s] = a;
            const unsigned long long used = cut == 64 ? acc : (acc & (~0ull >> (64 - cut)));
            const int k = (int)__builtin_popcountll(used);
            i -= k;
            accepted += k;
            p += cut;
        }
        if (i < low) end = p < len ? p : len;            // step `low` has been served (a mask's range ends there: the run was cut)
    }
    if (lane == 0) {
        mine_seen[0] = i0; mine_seen[1] = accepted; mine_seen[2] = end;
        const int was = __hip_atomic_load(cnt + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (was != accepted) {
            __hip_atomic_store(cnt + t, accepted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (!write_pass) flags[round] = 1;
        }
        // (exactly one segment sees the last of the rounds' n - low steps served inside it: the tail starts behind it)
        if (write_pass && accepted > 0 && before + accepted == (long long)(n - low) && end >= 0) { out[2] = seg0 + end; out[3] = 0; }
        if (write_pass && t == 0) flags[rounds + 1] = 1;                // "the targets have been written"
    }
}

// flags cleared (flags[0] = 1: "round 0 changed everything"), out = {0, "the words ran out"} until a segment says otherwise
__global__ void perm_reflag_kernel(int rounds, int *__restrict__ flags, long long *__restrict__ out) {
    if (threadIdx.x != 0) return;
    for (int r = 0; r <= rounds + 1; ++r) flags[r] = r == 0 ? 1 : 0;      // (flags[rounds + 1]: the write pass has run)
    out[0] = 0; out[1] = 1;          // the whole stream: words consumed, "ran out" until the tail says otherwise
    out[2] = 0; out[3] = 1;          // the rounds: where the tail starts, "not reached"
}

// The steps below `low` (the masks of at most 14 bits: 23 k words or so), one wavefront, strictly in order behind the rounds:
// every one of these short ranges starts where the one before it ended, so rounds would need one launch per range.
__global__ __launch_bounds__(64) void perm_tail_kernel(const unsigned *__restrict__ raw, long long n_avail, int n, int low,
                                                       int *__restrict__ J, long long *__restrict__ out) {
    __shared__ unsigned W[4096];
    const int lane = threadIdx.x;
    const bool rounds_had_steps = n - 1 >= low;
    if (rounds_had_steps && out[3] != 0) return;           // (the rounds never reached `low`: the words ran out, or not settled yet)
    long long p = rounds_had_steps ? out[2] : 0;
    int i = rounds_had_steps ? low - 1 : n - 1;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    bool failed = false;
    while (i >= 1) {
        // the next 4096 words into LDS (all loads in flight together), then runs of 64 out of it
        const long long base = p;
        const int len = (int)(base + 4096 < n_avail ? 4096 : n_avail - base);
        if (len < 64) { failed = true; break; }
        __syncthreads();
        for (int k = lane; k < len; k += 64) W[k] = perm_temper(raw[base + k]);
        __syncthreads();
        int q = 0;
        while (i >= 1 && q + 64 <= len) {
            const unsigned mask = 0xffffffffu >> __builtin_clz((unsigned)i);
            const int lowi = (int)(mask >> 1) + 1;
            const int a = (int)(W[q + lane] & mask);
            unsigned long long acc = __ballot(a <= i);
            for (;;) {
                const int c = __builtin_popcountll(acc & below);
                const unsigned long long acc2 = __ballot(a <= i - c);
                if (acc2 == acc) break;
                acc = acc2;
            }
            const int c = __builtin_popcountll(acc & below);
            const int s = i - c;
            const bool mine = (acc >> lane) & 1ull;
            const unsigned long long last = __ballot(mine && s == lowi);
            const int cut = last ? (int)__builtin_ctzll(last) + 1 : 64;
            if (mine && lane < cut) J[s] = a;
            const unsigned long long used = cut == 64 ? acc : (acc & (~0ull >> (64 - cut)));
            i -= (int)__builtin_popcountll(used);
            q += cut;
        }
        p = base + q;
    }
    if (lane == 0) { out[0] = p; out[1] = failed ? 1 : 0; }
}

// The counts round 1 starts from: the EXPECTED progress of the rejection sampling (a word is accepted with probability
// (i + 1) / 2^k at step i of a mask of k bits), segment by segment -- on the host, a microsecond's worth of arithmetic.
void perm_guess_host(int T, long long n_avail, int n, int low, int *cnt) {
    double i = (double)(n - 1);
    for (int t = 0; t < T; ++t) {
        const long long seg0 = (long long)t * kPermSeg;
        double words = (double)(seg0 + kPermSeg < n_avail ? kPermSeg : n_avail - seg0);
        const double i_in = i;
        while (words > 0.0 && i >= (double)low) {
            const unsigned ii = (unsigned)i;
            const double m = (double)(0xffffffffu >> __builtin_clz(ii)) + 1.0, lowi = 0.5 * m;
            const double need = m * log((i + 1.0) / lowi);           // words this mask's range still takes (sum of m / (j + 1))
            if (need <= words) { words -= need; i = lowi - 1.0; }
            else { i = (i + 1.0) * exp(-words / m) - 1.0; words = 0.0; }
        }
        if (i < (double)low - 1.0) i = (double)low - 1.0;
        cnt[t] = i_in >= (double)low ? (int)(i_in - i + 0.5) : 0;
    }
}

__global__ void perm_init_kernel(int n, int *__restrict__ pred, int *__restrict__ ptr) {
    const int v = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (v >= n) return;
    pred[v] = -1;
    ptr[v] = v;
}

// ks / idx: the steps 1 .. n-1 sorted by target (stable: ascending step inside a target), m = n - 1 of them
__global__ void perm_links_kernel(int m, const unsigned *__restrict__ ks, const unsigned *__restrict__ idx,
                                  int *__restrict__ pred, int *__restrict__ ptr) {
    const int q = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (q >= m) return;
    const unsigned v = ks[q], i = idx[q];
    if (v > (unsigned)m) return;                       // (not a target: only in a run whose draws have not settled)
    const bool next_same = q + 1 < m && ks[q + 1] == v;
    if (next_same) pred[i] = (int)idx[q + 1];
    if (q == 0 || ks[q - 1] != v) {
        const int pv = (i != v) ? (int)i : (next_same ? (int)idx[q + 1] : -1);
        if (pv >= 0) ptr[v] = pv;
    }
}

// in place: every pointer only ever moves towards its root, so a half-updated neighbour is still an ancestor
__global__ void perm_jump_kernel(int n, int *__restrict__ ptr, int *__restrict__ changed) {
    const int v = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (v >= n) return;
    const int p = ptr[v];
    const int pp = ptr[p];
    if (pp != p) {
        ptr[v] = pp;
        if (changed) *changed = 1;
    }
}

// V(v): the root of v's chain of "who was swapped in here before" links (ptr[v] = predV(v), or v itself at a root).  The
// chains are a handful of links long (each link goes to a uniformly later step; the longest of 2e6 about forty), so every
// position simply walks its own: nothing is written that another thread reads.
__device__ __forceinline__ int perm_root(const int *__restrict__ ptr, int v) {
    int p = ptr[v];
    while (true) {
        const int pp = ptr[p];
        if (pp == p) return p;
        p = pp;
    }
}

__global__ void perm_final_kernel(int n, const int *__restrict__ J, const int *__restrict__ pred, const int *__restrict__ ptr,
                                  long long *__restrict__ order) {
    const int i = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (i >= n) return;
    int val;
    if (i == 0) {
        val = perm_root(ptr, 0);
    } else {
        const int j = J[i];
        if (j == i) val = perm_root(ptr, i);
        else {
            const int pr = pred[i];
            val = pr >= 0 ? perm_root(ptr, pr) : j;
        }
    }
    order[i] = (long long)val;
}

// the generator state behind `used` words: block nb = (pos + used - 1) / 624 of the stream and the position in it
__global__ void perm_state_kernel(const unsigned *__restrict__ key_in, const unsigned *__restrict__ raw, int pos,
                                  const long long *__restrict__ out, unsigned *__restrict__ key_out, int *__restrict__ pos_out) {
    const long long used = out[0];
    const long long g = (long long)pos + used;
    const long long nb = used > 0 ? (g - 1) / 624 : 0;
    for (int k = threadIdx.x; k < 624; k += blockDim.x)
        key_out[k] = nb == 0 ? key_in[k] : raw[624 * nb - pos + k];
    if (threadIdx.x == 0) pos_out[0] = (int)(g - 624 * nb);
}

// ---- a generation in flight: rounds, tail and state in the rounds' launches ------------------------------------------------
// The draws of a generation that follows another one on the device (api_perm.hip "permutations in flight").  As
// perm_draw_kernel, and:
//   * a segment's start comes from block sums (64 segments each) + the counts of its own block: two loads per lane instead of
//     up to sixty;
//   * round 1 starts from the expected progress itself (pre0: its prefix, constant per N) and the counters a generation
//     needs cleared (block sums, round flags) come in two alternating blocks, each cleared by the generation in front: no
//     copy, no clearing launch between generations;
//   * the rounds serve every step down to kPermChainLow = 256 (round = launch either way: a serial tail of 16 384 steps was
//     84 us of ONE wavefront), and the wavefront that serves step 256 in the write pass walks the ~350 words that are left,
//     writes the generator state behind the generation and where the next one starts (*goff_out).
// zero: {block sums [nblk_pad], round flags [64]}; flags_out [64]: the round flags for the host's statistics;
// pos_out [4]: {position, went through, -, the swaps' overflow flag (cleared here)}.
static constexpr int kPermChainLow = 256;

__global__ __launch_bounds__(256) void perm_draw_chained_kernel(const unsigned *__restrict__ era_raw, long long n_avail, int n, int round,
                                                                int rounds, const int *__restrict__ pre0, int *cnt, int *__restrict__ seen,
                                                                int *zero, int *__restrict__ next_zero, int nblk_pad, int *__restrict__ J,
                                                                long long *__restrict__ out, int *__restrict__ flags_out,
                                                                const long long *__restrict__ goff_in, const unsigned *__restrict__ era_key,
                                                                int era_pos, unsigned *__restrict__ key_out, int *__restrict__ pos_out,
                                                                long long *__restrict__ goff_out) {
    __shared__ unsigned ws[4][kPermSeg];
    constexpr int low = kPermChainLow;
    int *bsum = zero, *flags = zero + nblk_pad;
    const long long off = *goff_in;
    if (round == 1 && blockIdx.x == 0) {
        for (int k = threadIdx.x; k < nblk_pad + 64; k += 256) next_zero[k] = 0;
        // (failed until the wavefront that finishes the generation says otherwise)
        if (threadIdx.x == 0) { pos_out[1] = 0; pos_out[3] = 0; *goff_out = -1; out[0] = 0; out[1] = 1; }
    }
    if (off < 0) return;                                   // (the generation in front of this one failed)
    const unsigned *__restrict__ raw = era_raw + off;
    const bool settled = round >= 2 && flags[round - 1] == 0;
    if (settled && round >= 3 && flags[round - 2] == 0) return;
    const bool write_pass = settled;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int t = (int)blockIdx.x * 4 + wv;
    const long long seg0 = (long long)t * kPermSeg;
    if (seg0 >= n_avail) return;
    const int len = (int)(seg0 + kPermSeg < n_avail ? kPermSeg : n_avail - seg0);
    long long before;
    if (round == 1) {
        before = pre0[t];
    } else {
        const int blk = t >> 6;
        int v = 0;
        for (int b = lane; b < blk; b += 64) v += __hip_atomic_load(bsum + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int k = (blk << 6) + lane;
        if (k < t) v += __hip_atomic_load(cnt + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        before = v;
    }
    const long long i0l = (long long)(n - 1) - before;
    const int i0 = i0l > 0 ? (int)i0l : 0;
    int *__restrict__ mine_seen = seen + 3 * (long long)t;
    const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
    unsigned *__restrict__ W = ws[wv];
    int accepted = 0, end = -1;
    if (round >= 2 && !write_pass && mine_seen[0] == i0) {
        accepted = mine_seen[1];
        end = mine_seen[2];
    } else if (i0 >= low) {
        for (int k = lane; k < len; k += 64) W[k] = perm_temper(raw[seg0 + k]);
        int i = i0, p = 0;
        while (i >= low && p < len) {
            const unsigned mask = 0xffffffffu >> __builtin_clz((unsigned)i);
            const int lowi = (int)(mask >> 1) + 1;
            const int a = (p + lane < len) ? (int)(W[p + lane] & mask) : 0x7fffffff;
            unsigned long long acc = __ballot(a <= i);
            for (;;) {
                const int c = __builtin_popcountll(acc & below);
                const unsigned long long acc2 = __ballot(a <= i - c);
                if (acc2 == acc) break;
                acc = acc2;
            }
            const int c = __builtin_popcountll(acc & below);
            const int sstep = i - c;
            const bool mine = (acc >> lane) & 1ull;
            const unsigned long long last = __ballot(mine && sstep == lowi);     // (low is a power of two: step `low` ends a range)
            const int cut = last ? (int)__builtin_ctzll(last) + 1 : 64;
            if (write_pass && mine && lane < cut) J[sstep] = a;
            const unsigned long long used = cut == 64 ? acc : (acc & (~0ull >> (64 - cut)));
            const int k = (int)__builtin_popcountll(used);
            i -= k;
            accepted += k;
            p += cut;
        }
        if (i < low) end = p < len ? p : len;
    }
    if (lane == 0) {
        mine_seen[0] = i0; mine_seen[1] = accepted; mine_seen[2] = end;
        const int was = round == 1 ? 0 : __hip_atomic_load(cnt + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (round == 1 || was != accepted) {
            __hip_atomic_store(cnt + t, accepted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(bsum + (t >> 6), accepted - was, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (round 1 is measured against the expected count the segments behind it started from)
            const int expected = round == 1 ? pre0[t + 1] - pre0[t] : was;
            if (!write_pass && expected != accepted) flags[round] = 1;
        }
    }
    if (!(write_pass && accepted > 0 && before + accepted == (long long)(n - low) && end >= 0)) return;
    // ---- this wavefront served step `low`: the steps low - 1 .. 1 from the word behind it, then the state behind them
    long long p = seg0 + end;
    int i = low - 1;
    bool failed = false;
    while (i >= 1) {
        const long long base = p;
        const int tl = (int)(base + kPermSeg < n_avail ? kPermSeg : n_avail - base);
        if (tl < 64) { failed = true; break; }
        for (int k = lane; k < tl; k += 64) W[k] = perm_temper(raw[base + k]);
        int q = 0;
        while (i >= 1 && q + 64 <= tl) {
            const unsigned mask = 0xffffffffu >> __builtin_clz((unsigned)i);
            const int lowi = (int)(mask >> 1) + 1;
            const int a = (int)(W[q + lane] & mask);
            unsigned long long acc = __ballot(a <= i);
            for (;;) {
                const int c = __builtin_popcountll(acc & below);
                const unsigned long long acc2 = __ballot(a <= i - c);
                if (acc2 == acc) break;
                acc = acc2;
            }
            const int c = __builtin_popcountll(acc & below);
            const int sstep = i - c;
            const bool mine = (acc >> lane) & 1ull;
            const unsigned long long last = __ballot(mine && sstep == lowi);
            const int cut = last ? (int)__builtin_ctzll(last) + 1 : 64;
            if (mine && lane < cut) J[sstep] = a;
            const unsigned long long used = cut == 64 ? acc : (acc & (~0ull >> (64 - cut)));
            i -= (int)__builtin_popcountll(used);
            q += cut;
        }
        p = base + q;
    }
    const long long total = off + p;
    const long long g = (long long)era_pos + total;
    const long long nb = total > 0 ? (g - 1) / 624 : 0;
    if (!failed)
        for (int k = lane; k < 624; k += 64) key_out[k] = nb == 0 ? era_key[k] : era_raw[624 * nb - era_pos + k];
    flags_out[lane] = (lane >= 1 && lane <= rounds) ? flags[lane] : 0;
    if (lane == 0) {
        out[0] = p; out[1] = failed ? 1 : 0;
        pos_out[0] = (int)(g - 624 * nb);
        pos_out[1] = failed ? 0 : 1;
        *goff_out = failed ? -1 : total;
    }
}

// pre0 [T + 1]: prefix of the expected counts (host); returns the block sums' padded length
int perm_chain_guess(long long n_avail, int n, int *pre0_host) {
    const int T = perm_segments(n_avail);
    std::vector<int> g((size_t)T);
    perm_guess_host(T, n_avail, n, kPermChainLow, g.data());
    pre0_host[0] = 0;
    for (int t = 0; t < T; ++t) pre0_host[t + 1] = pre0_host[t] + g[(size_t)t];
    return 64 * ((T + 63) / 64 / 64 + 1);
}
// rounds <= 60 launches, the last of them long returned when the generation settles early
bool launch_permutation_draws_chained(const unsigned *era_raw, const unsigned *era_key, int era_pos, const long long *goff_in,
                                       long long *goff_out, long long n_avail, int n, int *J, int *cnt, const int *pre0, int *zero,
                                       int *next_zero, int nblk_pad, int *flags_out, long long *out, unsigned *key_out, int *pos_out,
                                       int rounds, hipStream_t st) {
    const int T = perm_segments(n_avail);
    int *seen = cnt + 2 * (long long)T;
    for (int r = 1; r <= rounds; ++r)
        hipLaunchKernelGGL(perm_draw_chained_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, st, era_raw, n_avail, n, r, rounds, pre0, cnt,
                           seen, zero, next_zero, nblk_pad, J, out, flags_out, goff_in, era_key, era_pos, key_out, pos_out, goff_out);
    return hipGetLastError() == hipSuccess;
}

// ---- the swaps' links without a radix sort --------------------------------------------------------------------------------
// pred / predV need, for every target, the steps that hit it in ascending order.  A full sort of the (target, step) pairs is
// 21 launches of rocPRIM (156 us at N = 1e6); the generations in flight do it in two: the pairs are dealt into buckets of
// targets whose EXPECTED load is equal -- the targets are not uniform (step i hits v <= i with probability 1 / (i + 1): v = 0
// collects ln n hits, v = n - 1 hardly any), so the bucket boundaries are the quantiles of that law (perm_bucket_bounds, host,
// once per N) and every bucket receives (n - 1) / NB +- a few dozen pairs, 2 048 at most on average in 3 072 slots -- and one
// workgroup per bucket sorts its pairs in LDS (bitonic, 64-bit keys target << 32 | step) and writes the links.  A bucket that
// overflows raises a flag and the generation is drawn again the old way (perm_final_kernel's input would be incomplete).
static constexpr int kPermBucketCap = 2048;
int perm_bucket_cap() { return kPermBucketCap; }
// NB buckets (<= 2048; 0: N too large for this route), bnd[NB + 1]: bucket b holds targets [bnd[b], bnd[b + 1])
int perm_bucket_bounds(int n, std::vector<int> &bnd) {
    // (1 536 pairs per bucket on average: 2 048 -- the size the buckets' sort is built for -- is thirteen standard deviations away)
    const int NB = (int)(((long long)n - 1 + 1535) / 1536);
    if (NB > 2048) return 0;
    // rho[v] = sum_{i = max(v, 1)}^{n - 1} 1 / (i + 1): expected number of steps whose target is v
    std::vector<double> cum((size_t)n + 1);
    double suffix = 0.0;
    std::vector<double> rho((size_t)n);
    for (int i = n - 1; i >= 1; --i) { suffix += 1.0 / (double)(i + 1); rho[(size_t)i] = suffix; }
    rho[0] = suffix;
    cum[0] = 0.0;
    for (int v = 0; v < n; ++v) cum[(size_t)v + 1] = cum[(size_t)v] + rho[(size_t)v];
    bnd.assign((size_t)NB + 1, 0);
    const double per = cum[(size_t)n] / (double)NB;
    int v = 0;
    for (int b = 1; b < NB; ++b) {
        while (v < n && cum[(size_t)v + 1] < per * b) ++v;
        bnd[(size_t)b] = v;
    }
    bnd[(size_t)NB] = n;
    return NB;
}

// the pairs (J[i], i), i = 1 .. n - 1, into their buckets' slots (cursor[b]: pairs in bucket b so far; zero on entry)
__global__ __launch_bounds__(256) void perm_bucket_scatter_kernel(int n, int NB, const int *__restrict__ bnd, const int *__restrict__ J,
                                                                  int *__restrict__ cursor, unsigned long long *__restrict__ slots,
                                                                  int *__restrict__ overflow) {
    extern __shared__ int lds[];                   // [NB + 1] boundaries, [NB] counts, [NB] reserved bases
    int *b_lo = lds, *cnt = lds + NB + 1, *base = cnt + NB;
    constexpr int PER = 16;
    for (int k = threadIdx.x; k <= NB; k += 256) b_lo[k] = bnd[k];
    for (int k = threadIdx.x; k < NB; k += 256) cnt[k] = 0;
    __syncthreads();
    const int i0 = 1 + (int)blockIdx.x * 256 * PER;
    int myb[PER], myr[PER], myv[PER];
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int i = i0 + t * 256 + (int)threadIdx.x;
        myv[t] = i < n ? J[i] : -1;
    }
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        myb[t] = -1;
        if (myv[t] >= 0) {
            int lo = 0, hi = NB;                   // (the bucket: the last boundary <= v)
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (b_lo[mid] <= myv[t]) lo = mid; else hi = mid;
            }
            myb[t] = lo;
            myr[t] = atomicAdd(&cnt[lo], 1);
        }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < NB; k += 256) base[k] = cnt[k] ? atomicAdd(&cursor[k], cnt[k]) : 0;
    __syncthreads();
#pragma unroll
    for (int t = 0; t < PER; ++t)
        if (myb[t] >= 0) {
            const int at = base[myb[t]] + myr[t];
            const int i = i0 + t * 256 + (int)threadIdx.x;
            if (at < kPermBucketCap) slots[(long long)myb[t] * kPermBucketCap + at] = ((unsigned long long)(unsigned)myv[t] << 32) | (unsigned)i;
            else *overflow = 1;
        }
}

// one workgroup per bucket: its pairs sorted (target, then step), the links pred / ptr of perm_links_kernel for them -- and
// the defaults perm_init_kernel would have set for the bucket's targets; the bucket's cursor cleared for the next generation
__global__ __launch_bounds__(256) void perm_bucket_links_kernel(int n, const int *__restrict__ bnd, int *__restrict__ cursor,
                                                                const unsigned long long *__restrict__ slots, int *__restrict__ pred,
                                                                int *__restrict__ ptr) {
    constexpr int P2 = kPermBucketCap;
    __shared__ unsigned long long key[P2];
    const int b = (int)blockIdx.x, tid = (int)threadIdx.x;
    int m = cursor[b];
    if (m > kPermBucketCap) m = kPermBucketCap;
#pragma unroll
    for (int t = 0; t < P2 / 256; ++t) {
        const int e = tid + t * 256;
        key[e] = e < m ? slots[(long long)b * kPermBucketCap + e] : ~0ull;
    }
    const int v_lo = bnd[b], v_hi = bnd[b + 1];
    for (int v = v_lo + tid; v < v_hi; v += 256) ptr[v] = v;
    __syncthreads();
    // bitonic network, every thread a PAIR per step (lower index: bit log2(j) of the pair number inserted as 0)
    for (int k = 2; k <= P2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
            for (int t = 0; t < P2 / 512; ++t) {
                const int pr = tid + t * 256;
                const int e = ((pr & ~(j - 1)) << 1) | (pr & (j - 1));
                const unsigned long long a = key[e], c = key[e | j];
                const bool up = (e & k) == 0;
                if ((a > c) == up) { key[e] = c; key[e | j] = a; }
            }
            __syncthreads();
        }
    for (int q = tid; q < m; q += 256) {
        const unsigned v = (unsigned)(key[q] >> 32), i = (unsigned)key[q];
        const bool next_same = q + 1 < m && (unsigned)(key[q + 1] >> 32) == v;
        pred[i] = next_same ? (int)(unsigned)key[q + 1] : -1;
        if (q == 0 || (unsigned)(key[q - 1] >> 32) != v) {
            const int pv = (i != v) ? (int)i : (next_same ? (int)(unsigned)key[q + 1] : -1);
            if (pv >= 0) ptr[v] = pv;
        }
    }
    if (tid == 0) cursor[b] = 0;
}

// the swaps of a generation in flight: buckets, links, assembly (three launches)
bool launch_permutation_swaps_bucketed(int n, int NB, const int *bnd, const int *J, int *cursor, unsigned long long *slots, int *overflow,
                                       int *pred, int *ptr, long long *order, hipStream_t st) {
    const unsigned gs = (unsigned)((n - 1 + 256 * 16 - 1) / (256 * 16));
    hipLaunchKernelGGL(perm_bucket_scatter_kernel, dim3(gs), dim3(256), (size_t)(3 * NB + 1) * sizeof(int), st, n, NB, bnd, J, cursor, slots,
                       overflow);
    hipLaunchKernelGGL(perm_bucket_links_kernel, dim3((unsigned)NB), dim3(256), 0, st, n, bnd, cursor, slots, pred, ptr);
    hipLaunchKernelGGL(perm_final_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, J, (const int *)pred, (const int *)ptr, order);
    return hipGetLastError() == hipSuccess;
}

static unsigned perm_key_bits(int n) {
    unsigned bits = 1;
    while ((1ll << bits) < (long long)n) ++bits;
    return bits;
}

size_t perm_sort_temp_bytes(int n) {
    size_t bytes = 0;
    unsigned *nul = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, nul, nul, nul, nul, (size_t)(n > 1 ? n - 1 : 1), 0, perm_key_bits(n),
                                    hipStream_t(nullptr));
    return bytes;
}

static void queue_draw_rounds(const unsigned *raw, long long n_avail, int n, int *J, int *cnt, int *flags, long long *out, int parity,
                              hipStream_t st) {
    const int T = perm_segments(n_avail);
    int *seen = cnt + 2 * (long long)T;              // [3 T]: perm_draw_kernel's memo (round 1 ignores what it holds)
    (void)parity;
    const int rounds = perm_rounds_now();
    hipLaunchKernelGGL(perm_reflag_kernel, dim3(1), dim3(64), 0, st, rounds, flags, out);
    const int low = perm_low(n);
    for (int r = 1; r <= rounds; ++r)
        hipLaunchKernelGGL(perm_draw_kernel, dim3((unsigned)((T + 3) / 4)), dim3(256), 0, st, raw, n_avail, n, low, r, rounds, cnt, seen, J, out,
                           flags);
    hipLaunchKernelGGL(perm_tail_kernel, dim3(1), dim3(64), 0, st, raw, n_avail, n, low, J, out);
}

// the swaps: sort by target, links, assembly
bool launch_permutation_swaps(int n, int *J, int *pred, int *ptr, unsigned *ks, unsigned *idx, unsigned *iota, void *temp,
                              size_t temp_bytes, int *changed, long long *order, hipStream_t st) {
    const unsigned g = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(perm_init_kernel, dim3(g), dim3(256), 0, st, n, pred, ptr);
    if (n > 1) {
        // values: the steps 1 .. n-1 (iota[k] = k, filled once by the caller)
        if (rocprim::radix_sort_pairs(temp, temp_bytes, (const unsigned *)(J + 1), ks, (const unsigned *)(iota + 1), idx, (size_t)(n - 1), 0,
                                      perm_key_bits(n), st) != hipSuccess)
            return false;
        hipLaunchKernelGGL(perm_links_kernel, dim3((unsigned)((n - 1 + 255) / 256)), dim3(256), 0, st, n - 1, ks, idx, pred, ptr);
        // (the chains of links are walked by perm_final_kernel itself: no rounds of pointer jumping)
    }
    (void)hipMemsetAsync(changed, 0, sizeof(int), st);
    hipLaunchKernelGGL(perm_final_kernel, dim3(g), dim3(256), 0, st, n, J, pred, ptr, order);
    return hipGetLastError() == hipSuccess;
}

// everything behind the draws: the swaps and the generator state
bool launch_permutation_tail(const unsigned *raw, int n, const unsigned *key_in, int pos, int *J, int *pred, int *ptr, unsigned *ks,
                             unsigned *idx, unsigned *iota, void *temp, size_t temp_bytes, long long *out, int *changed, long long *order,
                             unsigned *key_out, int *pos_out, hipStream_t st) {
    if (!launch_permutation_swaps(n, J, pred, ptr, ks, idx, iota, temp, temp_bytes, changed, order, st)) return false;
    hipLaunchKernelGGL(perm_state_kernel, dim3(1), dim3(256), 0, st, key_in, raw, pos, out, key_out, pos_out);
    return hipGetLastError() == hipSuccess;
}

// raw / n_avail: untempered words behind the caller's position.  Scratch (device): J, pred, ptr [n] ints; cnt [5 x
// perm_segments] ints; flags [perm_rounds + 2] ints (flags[perm_rounds + 1] == 0 afterwards: the draws have not settled --
// launch_permutation_draw_more, then launch_permutation_tail again); ks, idx, iota [n] unsigned; temp
// (perm_sort_temp_bytes); out [4] long long; changed [1] int (!= 0 afterwards: launch_permutation_more).  order: the
// permutation, int64 [n].  guess_pinned: perm_segments ints of pinned host memory.  key_in (device, 624 words) / pos: the state the words start from; key_out / pos_out (device):
// the state behind the words consumed.
bool launch_permutation(const unsigned *raw, long long n_avail, int n, const unsigned *key_in, int pos, int *J, int *pred, int *ptr,
                        int *cnt, int *guess_pinned, int *flags, unsigned *ks, unsigned *idx, unsigned *iota, void *temp, size_t temp_bytes,
                        long long *out, int *changed, long long *order, unsigned *key_out, int *pos_out, hipStream_t st) {
    const int T = perm_segments(n_avail);
    perm_guess_host(T, n_avail, n, perm_low(n), guess_pinned);
    if (hipMemcpyAsync(cnt, guess_pinned, sizeof(int) * (size_t)T, hipMemcpyHostToDevice, st) != hipSuccess) return false;   // (round 1 reads cnt[0 .. T))
    queue_draw_rounds(raw, n_avail, n, J, cnt, flags, out, 0, st);
    return launch_permutation_tail(raw, n, key_in, pos, J, pred, ptr, ks, idx, iota, temp, temp_bytes, out, changed, order, key_out,
                                   pos_out, st);
}

// perm_rounds() more rounds of draws from the counts the last round left (updated in place: they are simply carried on in
// cnt[0 .. T))
void launch_permutation_draw_more(const unsigned *raw, long long n_avail, int n, int *J, int *cnt, int *flags, long long *out,
                                  hipStream_t st) {
    queue_draw_rounds(raw, n_avail, n, J, cnt, flags, out, 0, st);
}

// more rounds of pointer jumping + the final assembly again (the rare case `changed` reported)
void launch_permutation_more(int n, const int *J, const int *pred, int *ptr, int *changed, long long *order, hipStream_t st) {
    const unsigned g = (unsigned)((n + 255) / 256);
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(perm_jump_kernel, dim3(g), dim3(256), 0, st, n, ptr, (int *)nullptr);
    (void)hipMemsetAsync(changed, 0, sizeof(int), st);
    hipLaunchKernelGGL(perm_jump_kernel, dim3(g), dim3(256), 0, st, n, ptr, changed);
    hipLaunchKernelGGL(perm_final_kernel, dim3(g), dim3(256), 0, st, n, J, pred, ptr, order);
}

__global__ void perm_iota_kernel(int n, unsigned *__restrict__ iota) {
    const int v = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (v < n) iota[v] = (unsigned)v;
}
void launch_perm_iota(int n, unsigned *iota, hipStream_t st) {
    hipLaunchKernelGGL(perm_iota_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, iota);
}

// Pruned windows: certified stays (certify_kernel), exact pruning for full covariance
// (score_mfma_prune_kernel) and for diagonal / fixed-variance components (score_diag_prune_kernel).
// Each decides as much as it rigorously can and leaves the rest to the next; the full evaluation of
// every pair lives in kernels_score.hip.
#include "score_common.h"
#include <cstdlib>

// ------------------------------------------------------------------------------------------
// MFMA kernel with EXACT pruning of negligible components (fresh windows only).
//
// For a slot s with Lambda_s >= lambda_max(S_N) (Gershgorin at every from-scratch refresh, raised by
// a |d|^2 at every rank-1 addition; slot_math.h) the quadratic form is bounded from below by the
// Euclidean distance:   q_s(x) = (mu-x)' S_N^-1 (mu-x) >= |mu - x|^2 / Lambda_s =: q_lb,   and
//     lp_ub = logseat + A - half_vd * L(q_lb * inv_cv),     L(t) <= log(1 + t)  (cheap minorant)
// is a rigorous upper bound of the component's log score for that visit; any lower bound of
// |mu - x| can stand in for the distance.  Every visit also has a lower bound M_lb of its maximum
// log score: the "new table" entry log(alpha) + log_prior[i] (crpmm.py:74) to start with, raised
// by (a lower bound of) every exact score computed so far -- its own component first.  A slot whose
// lp_ub < M_lb - kPruneMargin, and that is not the visit's home, has weight
// exp(lp - max) < e^-80 ~ 2e-35 in that draw -- twenty orders of magnitude below the rounding
// noise of the normaliser -- and is not scored: nothing is written for it, which the draw kernel
// treats as an exact zero weight.  The bounds hold against the frozen state only, which is why a
// pruned window ends at its first move (slot_math.h).
//
// Three levels, cheapest first, each only for what survived the one before:
//   coarse   once per wave (32 visits) and label: |mu_t - x| >= |mu_t - mu_h| - max_v |x_v - mu_h|
//            over the visits whose home is h (the visits are sorted by home, a wave has one or two),
//            from the centre-to-centre table pr_dcc;  no matrix work at all
//   level 0  per visit, distance on the leading 32 dimensions: 8 v_mfma_f64_16x16x4 per 16 visits
//            and 16 labels
//   level 1  per visit, all Dp dimensions (the remaining Dp/4 - 8 MFMAs)
// and what survives those is scored exactly (2 nJ (nJ+1) MFMAs per 16 visits and label).
// ------------------------------------------------------------------------------------------
static constexpr double kPruneMargin = 80.0;

__device__ __forceinline__ long long readlane64(long long v, int l) {
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xFFFFFFFFll), l);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), l);
    return ((long long)hi << 32) | (unsigned int)lo;
}

// minorant of log(1 + t), t >= 0:  1 + t = m 2^e with 0.5 <= m < 1, and log is concave, so
// log(m) >= (2m - 2) ln 2 (its chord over [0.5, 1]):  log(1 + t) >= (e + 2m - 2) ln 2.
// No division, no branch; at most 0.06 below the logarithm.
__device__ __forceinline__ double log1p_lower(double t) {
    const double y = 1.0 + t;
    const double m = __builtin_amdgcn_frexp_mant(y);
    const int e = __builtin_amdgcn_frexp_exp(y);
    return 0.6931471805599453 * ((double)(e - 2) + 2.0 * m);
}

// ------------------------------------------------------------------------------------------
// Certified stays.  In a converged chain almost every visit keeps its component with probability
// 1 - epsilon, epsilon far below the resolution of the uniform.  certify_kernel proves that per
// visit without touching the data row.  score_mfma_prune_kernel leaves two numbers per point
// (PCache): its exact quadratic form under its home component and its squared distance to that
// component's mean, tagged with the home slot and the version of that slot's state (any change of a
// slot bumps its version).  While the tag still matches, the home's score is known
// (slot_score_lower: within the dropped -0.5 log(1 - a1 q) >= 0 and log(1+t) <= t), the new
// table's score is log(alpha) + log_prior[i], and every other component is bounded from above by
// ftab[home label][radius bin] (kernels_state.hip: prune_ftable_kernel -- the coarse triangle
// bound of the pruning kernel, maximised over the other labels, tabulated over the distance to
// the home's mean).  If every alternative lies more than 38 + log(K + 1) nats below the home's
// lower bound, their total weight relative to the home is < e^-38 = 3e-17 < 2^-53: the reference's
// normaliser rounds to the home's score, p_home = exp(0) = 1 exactly, everything before it in the
// scan subtracts < 3e-17 from a uniform that is at least 2^-53 (an exact zero disables pruning for
// the sweep, api_inputs.hip), and `u - 1 < 0` returns the home.
// One thread per window row, in visiting order; the rows it certifies are left out of the bucket
// sort and of everything behind it.
// ------------------------------------------------------------------------------------------
// tier 2 of certify_kernel for one visit (its home slot h is live)
__device__ __forceinline__ bool certify_tier2(const Dev &d, const Ctrl *c, long long i, int h, double margin) {
    const PCache pc = d.pcache[i];
    // (the per-home table must belong to the current state: a full step has just rebuilt it
    // if need be, a lean step has not)
    if ((d.lean_step && !c->tables_valid) || pc.tag != (((long long)h << 32) | (unsigned int)d.mu_ver[h])) return false;
    // <= the exact home score (diag / fixed: the cache holds the one-point-removed log density)
    const double hlb = d.cov_type == COV_FULL ? slot_score_lower(d.sc[h], pc.qhome, true)
                                              : d.sc[h].logseat1 + pc.qhome;
    const double thr = hlb - margin;
    if (!(d.log_alpha + d.log_prior[i] < thr)) return false;                // the new table is not negligible
    const int a = d.label_of_slot[h];
    const double rad = sqrt(pc.rho2 * (1.0 + 1e-9)) * (1.0 + 1e-9);
    const double jf = rad * d.finv[a];
    return jf < 62.0 && d.ftab[(long long)a * 64 + (int)jf + 1] < thr;      // (radius rounded up)
}

// kCertVisits visits per thread (rows r0 + 256 k: every load instruction of a wave stays contiguous):
// their loads (index, then the tier-1 record) are issued side by side -- the kernel is a chain of
// dependent loads per visit, so what it costs is round trips, not bytes.
constexpr int kCertVisits = 4;
__global__ __launch_bounds__(256) void certify_kernel(Dev d) {
    const Ctrl *c = d.ctrl;
    if (!job_is_pruned(d, c->job.mode, c->job.prune)) return;
    // (the bucket sort's bins are free again: cleared for the next pruned window -- the sparse draw
    // kernel, which also does this, is not part of a lean step)
    if (blockIdx.x == 0)
        for (int b = threadIdx.x; b < d.nslots + 2; b += 256) d.bucket_bins[b] = 0;
    const long long base = c->job.win_base;
    const long long nrows = c->job.win_hi - base;
    const long long epoch = c->state_epoch;
    const double margin = 38.0 + log((double)c->job.K + 1.0);
    const long long r0 = (long long)blockIdx.x * 256 * kCertVisits + threadIdx.x;
    // A lean step only asks whether EVERY visit of the window is certified (apply_kernel compares the
    // count with the window; the flags are not read).  When the window is the whole sweep, the points
    // can be examined in storage order instead of a permuted visiting order: contiguous reads instead
    // of two gathered sectors per visit.  (All N points certified implies every visit certified, also
    // for an order with repeats.)
    const bool data_order = d.lean_step && base == 0 && nrows == d.N;
    long long iv[kCertVisits];
    PCacheExact pe[kCertVisits];
#pragma unroll
    for (int k = 0; k < kCertVisits; ++k) {
        const long long r = r0 + 256 * k;
        iv[k] = r < nrows ? ((d.order && !data_order) ? d.order[base + r] : base + r) : -1;
    }
    // tier 1: nothing at all has changed since the draw kernel last scored this visit -- the total
    // weight of its alternatives relative to the home is still the one it stored.  The record is only
    // written for a visit whose home is live (choice_sparse_kernel), and every change of any label,
    // count or component bumps the epoch: a matching epoch needs neither the label nor the count.
#pragma unroll
    for (int k = 0; k < kCertVisits; ++k) pe[k] = d.pcache2[iv[k] >= 0 ? iv[k] : 0];
    int n_ok = 0;
#pragma unroll
    for (int k = 0; k < kCertVisits; ++k) {
        bool ok = false;
        if (iv[k] >= 0) {
            if (pe[k].epoch == epoch) {
                ok = pe[k].log_alt <= -37.75;  // total alternative weight < e^-37.75 < 2^-53 (e^-36.74), 1 nat to spare
            } else {
                // tier 2: only the home component's state must be unchanged; the others are bounded
                // through the per-home table
                const int h = d.z[iv[k]];
                if (h >= 0 && d.n[h] >= 2) ok = certify_tier2(d, c, iv[k], h, margin);
            }
        }
        if (r0 + 256 * k < nrows) d.cert[r0 + 256 * k] = ok ? 1 : 0;
        n_ok += ok ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) n_ok += __shfl_xor(n_ok, o);
    if ((threadIdx.x & 63) == 0 && n_ok)
        atomicAdd(&d.pr_counts[768 + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 255)], (unsigned long long)n_ok);
}

void launch_certify(const Dev &d, long long max_rows, hipStream_t st) {
    if (max_rows <= 0) return;
    const long long per_block = 256ll * kCertVisits;
    hipLaunchKernelGGL(certify_kernel, dim3((unsigned)((max_rows + per_block - 1) / per_block)), dim3(256), 0, st, d);
}

// In a pruned window the visits are evaluated in the order of d.wrec (grouped by home component,
// kernels_state.hip: bucket_*_kernel), so that the visits of one wave mostly share a home and
// need the same one or two components in full.  The output is block-sparse: for every 16-visit
// evaluation block b a bitmask over labels (d.keep64) says which components were scored in full,
// and only those (block, label) lines -- 16 quadratic forms, 128 bytes -- are written to
// qb[(b * nslots + label) * 16 + v].  The draw kernel for pruned windows (choice_sparse_kernel)
// reads nothing else.  Labels are handed out in groups of 16 (group G = labels 16G .. 16G+15,
// one MFMA column each), group G to chunk G % chunks; everything a group needs comes from the
// label-ordered tables prune_tables_kernel keeps for the frozen state (coalesced 512-byte fragments).
//
// The 32 rows of a wave are staged through LDS (row-contiguous 512-byte global loads, then the
// A fragments x[row lr][4kk + lk] are read back; row stride Ds = 4 mod 32 doubles).
__host__ __device__ constexpr int prune_row_stride(int Dp) { return ((Dp + 27) / 32) * 32 + 4; }

template <int NJ, int RB>
__device__ __forceinline__ void prune_tile(const Dev &d, const JobView &job, double *__restrict__ q, unsigned bx) {
    extern __shared__ __attribute__((aligned(16))) double xs_all[];
    const int chunk = blockIdx.y;
    const int ngroups = (job.nlist + 15) >> 4;
    if (chunk >= job.chunks || chunk >= ngroups) return;
    constexpr int ROWS_W = 16 * RB;
    constexpr int NF = 2 * NJ * (NJ + 1);
    constexpr int NKK = NJ * 4;
    constexpr int NK0 = NKK < 8 ? NKK : 8;                        // fragments of the level-0 bound (32 dimensions)
    constexpr int Ds = prune_row_stride(NJ * 16);
    const long long nrows = prune_count(d);                       // the rows left to this kernel (not certified, not decided by home_kernel)
    const long long kb = (long long)bx * (4 * ROWS_W);
    if (kb >= nrows) return;
    const int D = d.D;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long kw = kb + w * ROWS_W;                         // first evaluation position of the wave
    const int lr = lane & 15, lk = lane >> 4;

    const long long nfrag64 = (long long)NF * 64;
    const long long blk0 = kw >> 4;                                // evaluation block of R = 0
    // ---- stage the wave's rows: lane r < 32 owns the data index of row r
    double *__restrict__ xs = xs_all + w * (ROWS_W * Ds);
    const long long kmine = kw + (lane & (ROWS_W - 1));
    WRec rmine;
    if (kmine < nrows) rmine = prune_list(d)[kmine];
    else { rmine.i = -1; rmine.home = -2; rmine.home_label = -1; rmine.mlb0 = INFINITY; }
    const long long imine = rmine.i;
    // gathers behind the bound, one value per row of the wave: the lower bound of the visit's best log
    // score (starts at the "new table" entry) and its home slot.  They live in LDS next to |x|^2
    // (wave-private side arrays behind the staging area; accumulator element (R, r) of lane (lk, .)
    // is row 16 R + lk + 4 r).
    const int side_stride = 176 + d.keep_stride;
    double *__restrict__ sideM = xs_all + 4 * (ROWS_W * Ds) + w * side_stride;   // Mlb[32]
    double *__restrict__ sideX2 = sideM + 32;                                 // |x|^2, all dimensions
    double *__restrict__ sideX2p = sideM + 64;                                // |x|^2, leading dimensions
    int *__restrict__ sideH = (int *)(sideM + 96);                            // home slot
    double *__restrict__ sideRho = sideM + 112;                               // |x - mu_home|^2
    long long *__restrict__ sideI = (long long *)(sideM + 144);                 // data index
    unsigned long long *__restrict__ sideC = (unsigned long long *)(sideM + 176);   // coarse label mask
    const int hmine = rmine.home;                                  // home slot of row (lane & 31); dead rows -2
    if (lane < ROWS_W) {
        sideM[lane] = rmine.mlb0;                                  // dead rows: +inf, never keep a slot alive
        sideH[lane] = hmine;
        sideI[lane] = imine;
    }
    // ---- the home components come first (the visits are grouped by home: mostly one or two per
    // wave), so that every bound below is taken against a tight Mlb.  Their q lines are stored for
    // both blocks; the group loop skips them.  Up to 4 distinct homes; the rest is found by the loop.
    int done0 = -1, done1 = -1, done2 = -1, done3 = -1;
    int lab0 = 0, lab1 = 0, lab2 = 0, lab3 = 0, n_home = 0;
    unsigned long long pending_homes;
    {
        unsigned long long pending = __ballot(lane < ROWS_W && hmine >= 0);
#pragma unroll 1
        for (int it = 0; it < 4 && pending; ++it) {
            const int first = __ffsll((long long)pending) - 1;
            const int s = __builtin_amdgcn_readfirstlane(__shfl(hmine, first));
            pending &= ~__ballot(lane < ROWS_W && hmine == s);
            const int lab = __builtin_amdgcn_readfirstlane(__shfl(rmine.home_label, first));
            if (it == 0) { done0 = s; lab0 = lab; } else if (it == 1) { done1 = s; lab1 = lab; }
            else if (it == 2) { done2 = s; lab2 = lab; } else { done3 = s; lab3 = lab; }
            ++n_home;
        }
        pending_homes = pending;                                   // homes beyond the first four
    }
    // the first ring of inverse-factor tiles of the first home travels together with the rows
    constexpr int PFK = pick_ring(NF, 20);
    double ringk[PFK];
    {
        const double *__restrict__ wf = d.Wfrag + (long long)(done0 >= 0 ? done0 : 0) * nfrag64 + lane;
#pragma unroll
        for (int i = 0; i < PFK; ++i) ringk[i] = wf[i * 64];
    }
    {
        // all row loads in flight at once (unconditional, clamped addresses), then the LDS writes
        constexpr int NP = (NJ * 16 + 63) / 64;
        double tmp[ROWS_W][NP];
#pragma unroll
        for (int row = 0; row < ROWS_W; ++row) {
            const long long i = readlane64(imine, row);
            const double *__restrict__ xrow = d.X + (i >= 0 ? i : 0) * D;
#pragma unroll
            for (int pss = 0; pss < NP; ++pss) {
                const int l = pss * 64 + lane;
                tmp[row][pss] = xrow[l < D ? l : 0];
            }
        }
#pragma unroll
        for (int row = 0; row < ROWS_W; ++row) {
            const long long i = readlane64(imine, row);
#pragma unroll
            for (int pss = 0; pss < NP; ++pss) {
                const int l = pss * 64 + lane;
                const double v = (i >= 0 && l < D) ? tmp[row][pss] : 0.0;
                if (NJ * 16 >= (pss + 1) * 64 || l < NJ * 16) xs[row * Ds + l] = v;
            }
        }
    }
    __syncthreads();
    if (kw >= nrows) return;
    // Euclidean distance of every row to the mean of its own (home) component: the radius of the
    // coarse triangle bound.  Lane = dimension; the 32 mean rows are in flight together; the 32 sums
    // over 64 lanes are formed by a transposing butterfly (32 shuffles), row r ends up in lanes 2r, 2r+1.
    {
        static_assert(ROWS_W == 32, "the butterfly below reduces 32 rows");
        constexpr int NP = (NJ * 16 + 63) / 64;
        double mu_t[ROWS_W][NP];
#pragma unroll
        for (int row = 0; row < ROWS_W; ++row) {
            const int h = __builtin_amdgcn_readlane(hmine, row);
            const double *__restrict__ mrow = d.mu + (long long)(h >= 0 ? h : 0) * D;
#pragma unroll
            for (int pss = 0; pss < NP; ++pss) {
                const int l = pss * 64 + lane;
                mu_t[row][pss] = mrow[l < D ? l : 0];
            }
        }
        double acc[ROWS_W];
#pragma unroll
        for (int row = 0; row < ROWS_W; ++row) {
            double a = 0.0;
#pragma unroll
            for (int pss = 0; pss < NP; ++pss) {
                const int l = pss * 64 + lane;
                const double df = l < D ? xs[row * Ds + (l < NJ * 16 ? l : 0)] - mu_t[row][pss] : 0.0;
                a = fma(df, df, a);
            }
            acc[row] = a;
        }
#pragma unroll
        for (int half = 16; half >= 1; half >>= 1) {
            const bool up = (lane & (2 * half)) != 0;                 // lane bit 5, 4, 3, 2, 1
#pragma unroll
            for (int r = 0; r < half; ++r) {
                const double lo = acc[r], hi = acc[r + half];
                acc[r] = (up ? hi : lo) + __shfl_xor(up ? lo : hi, 2 * half);
            }
        }
        acc[0] += __shfl_xor(acc[0], 1);
        {
            // publish: the wave's side array, and the per-point cache certify_kernel reads next sweep
            const long long irow = __shfl(imine, lane >> 1);
            const int hrow = __shfl(hmine, lane >> 1);
            if ((lane & 1) == 0) {
                sideRho[lane >> 1] = acc[0];                                // (rows without a home: unused)
                if (irow >= 0 && hrow >= 0) {
                    const long long tg = ((long long)hrow << 32) | (unsigned int)d.mu_ver[hrow];
                    d.pcache[irow].tag = tg;
                    d.pcache[irow].rho2 = acc[0];
                }
            }
        }
    }
    double xf[RB][NKK];
#pragma unroll
    for (int R = 0; R < RB; ++R)
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) xf[R][kk] = xs[(R * 16 + lr) * Ds + 4 * kk + lk];

    unsigned short *__restrict__ keep16 = (unsigned short *)d.keep64;
    unsigned n_kept = 0, n_bound = 0, n_mfma = 0;

    // Work list of the wave (reuses its staging area in LDS -- the rows live in registers now):
    // two ints per entry {slot, label | store0 << 30 | store1 << 31}.  The kernel alternates between
    // "score everything on the list exactly" and "bound groups of labels until the list is full", so
    // that the two register-hungry parts are never live together.
    int *__restrict__ wlist = (int *)xs;
    constexpr int LIST_CAP = ROWS_W * Ds - 16;                     // entries; 16 spare per group
    int n_list = n_home;
    if (n_home > 0) { wlist[0] = done0; wlist[1] = lab0 | (3 << 30); }
    if (n_home > 1) { wlist[2] = done1; wlist[3] = lab1 | (3 << 30); }
    if (n_home > 2) { wlist[4] = done2; wlist[5] = lab2 | (3 << 30); }
    if (n_home > 3) { wlist[6] = done3; wlist[7] = lab3 | (3 << 30); }
    bool ring_ready = n_home > 0;                                  // ringk holds the first tiles of entry 0

    int G = chunk;
    bool tables_ready = false, norms_ready = false;
#pragma unroll 1
    for (;;) {
        // ================= exact quadratic forms of the listed (slot, label) entries =================
#pragma unroll 1
        for (int e = 0; e < n_list; ++e) {
            const int s = __builtin_amdgcn_readfirstlane(wlist[2 * e]);
            const int lf = __builtin_amdgcn_readfirstlane(wlist[2 * e + 1]);
            const int label = lf & 0x3FFFFFFF;
            const double *__restrict__ wf = d.Wfrag + (long long)s * nfrag64 + lane;
            const double *__restrict__ cvp = d.cvec + (long long)s * d.Dp + lr;
            // software pipeline over the slot's NF tiles: a ring of PFK loads in flight, refilled with
            // the NEXT entry's first tiles as this entry's run out (NF is a multiple of PFK)
            if (!ring_ready) {
#pragma unroll
                for (int i = 0; i < PFK; ++i) ringk[i] = wf[i * 64];
            }
            n_mfma += RB * NF;
            const bool has_next = e + 1 < n_list;
            const int s_next = has_next ? __builtin_amdgcn_readfirstlane(wlist[2 * e + 2]) : s;
            const double *__restrict__ wf_next = d.Wfrag + (long long)s_next * nfrag64 + lane;
            ring_ready = has_next;
            double cjk[NJ];
#pragma unroll
            for (int J = 0; J < NJ; ++J) cjk[J] = cvp[16 * J];
            const SlotConst scs = d.sc[s];
            const int ns = d.n[s];
            double qp[RB][4];
#pragma unroll
            for (int R = 0; R < RB; ++R)
#pragma unroll
                for (int r = 0; r < 4; ++r) qp[R][r] = 0.0;
#pragma unroll
            for (int J = 0; J < NJ; ++J) {
                v4d acc[RB];
#pragma unroll
                for (int R = 0; R < RB; ++R) acc[R] = (v4d){cjk[J], cjk[J], cjk[J], cjk[J]};
#pragma unroll
                for (int kk = 0; kk < 4 * (J + 1); ++kk) {
                    const int f = 2 * J * (J + 1) + kk;          // constant after unrolling
                    const double bfr = ringk[f % PFK];
                    if (f + PFK < NF) ringk[f % PFK] = wf[(f + PFK) * 64];
                    else if (has_next) ringk[f % PFK] = wf_next[(f + PFK - NF) * 64];
#pragma unroll
                    for (int R = 0; R < RB; ++R)
                        acc[R] = __builtin_amdgcn_mfma_f64_16x16x4f64(xf[R][kk], bfr, acc[R], 0, 0, 0);
                }
#pragma unroll
                for (int R = 0; R < RB; ++R)
#pragma unroll
                    for (int r = 0; r < 4; ++r) qp[R][r] = fma(acc[R][r], acc[R][r], qp[R][r]);
            }
#pragma unroll
            for (int R = 0; R < RB; ++R) {
                double v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = row16_sum(qp[R][r]);                   // visit lk + 4r, in all 16 lanes of the row
                    // a lower bound of this score is a lower bound of the visit's maximum (the home
                    // component counts with its one-point-removed form, as the draw kernel scores
                    // it); dead rows have Mlb = +inf already
                    if (lr == 0) {
                        const int row = R * 16 + lk + 4 * r;
                        const bool own = sideH[row] == s;
                        if (d.safe_mode) {
                            // proof pass of a safe-stay window (kernels_safe.hip): bounds and margins must hold for every
                            // state inside the window's budget -- the best-score bound is raised by the home's ROBUST
                            // lower bound only (the bound constants of the tables are the robust ones, too)
                            if (own) {
                                const double *__restrict__ rt = d.rtab + (long long)label * 8;
                                const double chi = (v[r] + rt[3]) * d.rtab[(long long)(d.nslots - 1) * 8 + 1];
                                if (rt[6] > 0.5 && chi < 1.0) sideM[row] = fmax(sideM[row], rt[4] + rt[5] * log(1.0 - chi));
                            }
                        } else {
                        if (!own || ns >= 2) sideM[row] = fmax(sideM[row], slot_score_lower(scs, v[r], own));
                        if (own) d.pcache[sideI[row]].qhome = v[r];     // (tag written with the distance)
                        }
                    }
                }
                // one 128-byte line per (block, label): lane (lk, lr < 4) stores visit lk + 4 lr
                const double mine = lr == 0 ? v[0] : (lr == 1 ? v[1] : (lr == 2 ? v[2] : v[3]));
                if (((lf >> (30 + R)) & 1) && lr < 4 && kw + R * 16 < nrows)
                    q[((blk0 + R) * (long long)d.nslots + label) * 16 + lk + 4 * lr] = mine;
            }
        }
        n_list = 0;
        ring_ready = false;                 // (already so: the last entry has no successor)
        if (G >= ngroups) break;

        // ================= bounds, group by group, until the list is full =================
        if (!tables_ready) {
            tables_ready = true;
            // ---- coarse bound, once per wave and label (triangle inequality through the home means):
            // for the visits whose home is h,  |mu_t - x| >= |mu_t - mu_h| - max |x - mu_h|.  A label
            // that this prunes for every home present in the wave is skipped for all 32 visits.
            {
                const bool all_homes = pending_homes == 0ull;
                double rho[4], mmin[4];
                const int hh = lane < ROWS_W ? sideH[lane] : -3;
                const double r2 = lane < ROWS_W ? sideRho[lane] : 0.0;
                const double ml = lane < ROWS_W ? sideM[lane] : INFINITY;
                bool covered = hh == -2 || hh == -3;                   // dead rows / lanes
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int dj = j == 0 ? done0 : (j == 1 ? done1 : (j == 2 ? done2 : done3));
                    const bool sel = dj >= 0 && hh == dj;
                    covered = covered || sel;
                    double a = sel ? r2 : 0.0, b = sel ? ml : INFINITY;
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) {
                        a = fmax(a, __shfl_xor(a, o));
                        b = fmin(b, __shfl_xor(b, o));
                    }
                    rho[j] = sqrt(a) * (1.0 + 1e-9);
                    mmin[j] = b;
                }
                const bool coarse_ok = all_homes && __ballot(!covered) == 0ull;
                // four batches of 64 labels per pass: all loads of a pass in flight together
                for (int t00 = 0; t00 < job.nlist; t00 += 256) {
                    double cb[4], ch[4], ct[4], cd[4][4];
#pragma unroll
                    for (int bq = 0; bq < 4; ++bq) {
                        const int t = t00 + 64 * bq + lane;
                        const int tc = t < job.nlist ? t : 0;
                        const double *__restrict__ g = d.pr_const + (long long)(tc >> 4) * 128 + (tc & 15);
                        cb[bq] = g[0]; ch[bq] = g[16]; ct[bq] = g[32];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int lj = j == 0 ? lab0 : (j == 1 ? lab1 : (j == 2 ? lab2 : lab3));
                            cd[bq][j] = d.pr_dcc[(long long)(j < n_home ? lj : 0) * d.nslots + tc];
                        }
                    }
#pragma unroll
                    for (int bq = 0; bq < 4; ++bq) {
                        const int t = t00 + 64 * bq + lane;
                        if (t00 + 64 * bq < job.nlist) {               // (uniform)
                            bool need = t < job.nlist;
                            if (coarse_ok && need) {
                                need = false;
#pragma unroll
                                for (int j = 0; j < 4; ++j) {
                                    const int lj = j == 0 ? lab0 : (j == 1 ? lab1 : (j == 2 ? lab2 : lab3));
                                    if (j < n_home) {
                                        double dl = cd[bq][j] * (1.0 - 1e-9) - rho[j];
                                        dl = dl > 0.0 ? dl : 0.0;
                                        const double ub = cb[bq] - ch[bq] * log1p_lower(dl * dl * ct[bq]);
                                        need = need || (ub >= mmin[j] - kPruneMargin) || t == lj;
                                    }
                                }
                            }
                            const unsigned long long m = __ballot(need);
                            if (lane == 0) sideC[(t00 >> 6) + bq] = m;
                        }
                    }
                }
            }
        }
        // |x|^2 of the rows (needed by the per-visit bounds only): computed when the first group that
        // needs them comes up
        auto rows_norms = [&]() {
#pragma unroll
            for (int R = 0; R < RB; ++R) {
                // from the A fragments (sum over kk, then over the 4 lk lanes): all dimensions and
                // leading dimensions only
                double part0 = 0.0, part = 0.0;
#pragma unroll
                for (int kk = 0; kk < NK0; ++kk) part0 = fma(xf[R][kk], xf[R][kk], part0);
#pragma unroll
                for (int kk = NK0; kk < NKK; ++kk) part = fma(xf[R][kk], xf[R][kk], part);
                part += part0;
                part += __shfl_xor(part, 16);
                part += __shfl_xor(part, 32);
                part0 += __shfl_xor(part0, 16);
                part0 += __shfl_xor(part0, 32);
                if (lk == 0) { sideX2[R * 16 + lr] = part; sideX2p[R * 16 + lr] = part0; }
            }
        };
        // Software pipeline over the groups: the level-0 fragments + constants of the next group are
        // in flight while this one is being bounded.
        struct GroupConst { double base, hvd, tcoef, mu2, mu2p; };
        auto load_group = [&](int Gq, int &sg_o, double (&bm_o)[NK0], GroupConst &gc_o) {
            const bool in = Gq < ngroups;
            const int Gc = in ? Gq : chunk;
            sg_o = in ? d.pr_slot[Gc * 16 + lr] : -1;
            const double *__restrict__ mf = d.pr_mufrag + (long long)Gc * (NKK * 64) + lane;
#pragma unroll
            for (int kk = 0; kk < NK0; ++kk) bm_o[kk] = mf[kk * 64];
            const double *__restrict__ g = d.pr_const + (long long)Gc * 128 + lr;
            gc_o.base = g[0]; gc_o.hvd = g[16]; gc_o.tcoef = g[32]; gc_o.mu2 = g[48]; gc_o.mu2p = g[64];
        };
        // the 16 coarse bits of group Gq (wave-uniform), and the next group >= Gq of this chunk that
        // has any
        auto coarse16 = [&](int Gq) -> unsigned {
            const unsigned long long wd = sideC[Gq >> 2];
            return (unsigned)__builtin_amdgcn_readfirstlane((int)((wd >> (16 * (Gq & 3))) & 0xFFFFull));
        };
        // the labels of the homes scored up front that fall into group Gq: kept without any test
        auto home16 = [&](int Gq) -> unsigned {
            unsigned hb = 0;
            if (n_home > 0 && (lab0 >> 4) == Gq) hb |= 1u << (lab0 & 15);
            if (n_home > 1 && (lab1 >> 4) == Gq) hb |= 1u << (lab1 & 15);
            if (n_home > 2 && (lab2 >> 4) == Gq) hb |= 1u << (lab2 & 15);
            if (n_home > 3 && (lab3 >> 4) == Gq) hb |= 1u << (lab3 & 15);
            return hb;
        };
        // next group >= Gq of this chunk in which a label other than those homes survives
        auto next_needed = [&](int Gq) -> int {
            while (Gq < ngroups && (coarse16(Gq) & ~home16(Gq)) == 0u) Gq += job.chunks;
            return Gq;
        };
        int sg_next;
        double bm_next[NK0];
        GroupConst gc_next;
        int G_next = next_needed(G);
        load_group(G_next, sg_next, bm_next, gc_next);
#pragma unroll 1
        for (; G < ngroups && n_list <= LIST_CAP; G += job.chunks) {
            {
                const int left = job.nlist - 16 * G;
                n_bound += RB * (left < 16 ? left : 16);
            }
            if (G != G_next) {
                // every label of the group is out for the whole wave, except homes scored up front
                // (their q lines exist for both blocks)
                const unsigned hk = coarse16(G) & home16(G);
                if (lane == 0) {
#pragma unroll
                    for (int R = 0; R < RB; ++R)
                        if (kw + R * 16 < nrows)
                            keep16[(blk0 + R) * (4ll * d.keep_stride) + G] = (unsigned short)hk;
                }
                n_kept += RB * __popc(hk);
                continue;
            }
            if (!norms_ready) { norms_ready = true; rows_norms(); }
            const unsigned c16 = coarse16(G);
            const int sg = sg_next;
            const GroupConst gc = gc_next;
            double bm[NK0];
#pragma unroll
            for (int kk = 0; kk < NK0; ++kk) bm[kk] = bm_next[kk];
            G_next = next_needed(G + job.chunks);
            load_group(G_next, sg_next, bm_next, gc_next);
            // distances to the 16 means: X . Mu'  (B fragment: mu_sg[4kk + lk]); leading dimensions
            v4d accG[RB];
#pragma unroll
            for (int R = 0; R < RB; ++R) accG[R] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < NK0; ++kk)
#pragma unroll
                for (int R = 0; R < RB; ++R)
                    accG[R] = __builtin_amdgcn_mfma_f64_16x16x4f64(xf[R][kk], bm[kk], accG[R], 0, 0, 0);
            n_mfma += RB * NK0;
            bool need[RB];
            bool any0 = false;
#pragma unroll
            for (int R = 0; R < RB; ++R) {
                need[R] = false;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = R * 16 + lk + 4 * r;
                    const double nrm = sideX2p[row] + gc.mu2p;
                    double dist2 = fma(-2.0, accG[R][r], nrm) - 1e-9 * nrm;     // (rounding of the difference)
                    dist2 = dist2 > 0.0 ? dist2 : 0.0;
                    const double ub = gc.base - gc.hvd * log1p_lower(dist2 * gc.tcoef);
                    need[R] = need[R] || (ub >= sideM[row] - kPruneMargin) || (sideH[row] == sg);
                }
                need[R] = need[R] && sg >= 0;
                any0 = any0 || need[R];
            }
            if (NKK > NK0 && __ballot(any0) != 0ull) {
                // somebody survives the leading-dimension bound: the distance on all dimensions
                n_mfma += RB * (NKK - NK0);
                const double *__restrict__ mf = d.pr_mufrag + (long long)G * (NKK * 64) + lane;
                double bmr[NKK > NK0 ? NKK - NK0 : 1];
#pragma unroll
                for (int kk = NK0; kk < NKK; ++kk) bmr[kk - NK0] = mf[kk * 64];
#pragma unroll
                for (int kk = NK0; kk < NKK; ++kk)
#pragma unroll
                    for (int R = 0; R < RB; ++R)
                        accG[R] = __builtin_amdgcn_mfma_f64_16x16x4f64(xf[R][kk], bmr[kk - NK0], accG[R], 0, 0, 0);
#pragma unroll
                for (int R = 0; R < RB; ++R) {
                    bool nd = false;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = R * 16 + lk + 4 * r;
                        const double nrm = sideX2[row] + gc.mu2;
                        double dist2 = fma(-2.0, accG[R][r], nrm) - 1e-9 * nrm;
                        dist2 = dist2 > 0.0 ? dist2 : 0.0;
                        const double ub = gc.base - gc.hvd * log1p_lower(dist2 * gc.tcoef);
                        nd = nd || (ub >= sideM[row] - kPruneMargin) || (sideH[row] == sg);
                    }
                    need[R] = need[R] && nd;
                }
            }
            // fold the votes of the 4 lk lanes (and 4 r's) of every label column; publish the block's
            // 16 mask bits of this group (every (block, group) is written exactly once per window)
            unsigned keepmask[RB];
#pragma unroll
            for (int R = 0; R < RB; ++R) {
                const unsigned long long bl = __ballot(need[R]);
                keepmask[R] = ((unsigned)((bl | (bl >> 16) | (bl >> 32) | (bl >> 48)) & 0xFFFFull) | home16(G)) & c16;
                if (lane == 0 && kw + R * 16 < nrows)
                    keep16[(blk0 + R) * (4ll * d.keep_stride) + G] = (unsigned short)keepmask[R];
                n_kept += __popc(keepmask[R]);
            }
            // the labels somebody needs go on the list (the homes scored up front are done already)
            unsigned todo = keepmask[0] | keepmask[RB - 1];
            while (todo) {
                const int jbit = __ffs(todo) - 1;
                todo &= todo - 1;
                const int s = __builtin_amdgcn_readlane(sg, jbit);
                if (s == done0 || s == done1 || s == done2 || s == done3) continue;
                wlist[2 * n_list] = s;
                wlist[2 * n_list + 1] = (16 * G + jbit) | (((keepmask[0] >> jbit) & 1u) << 30)
                                        | (((keepmask[RB - 1] >> jbit) & 1u) << 31);
                ++n_list;
            }
        }
    }
    if (lane == 0) {
        // (counters spread over 256 addresses; apply_kernel folds them)
        atomicAdd(&d.pr_counts[bx & 255], (unsigned long long)n_kept);
        atomicAdd(&d.pr_counts[256 + (bx & 255)], (unsigned long long)n_bound);
        atomicAdd(&d.pr_counts[512 + (bx & 255)], (unsigned long long)n_mfma);
    }
}

// One 128-row tile per workgroup; behind home_kernel, whose residual list is usually short, the host launches
// a short grid of the WALK variant instead, whose workgroups walk the tiles of the list.
template <int NJ, int RB, int MINW, bool WALK>
__global__ __launch_bounds__(256, MINW) void score_mfma_prune_kernel(Dev d, const Job *__restrict__ jobp,
                                                                  double *__restrict__ q, long long qstride) {
    const JobView job = load_job(jobp);
    if (!job_is_pruned(d, job.mode, job.prune) || (d.safe_mode && d.ctrl->safe_epoch_valid)) return;
    // (a proof pass whose table bound left only a window's worth of visits open: the resolver walks them anyway -- this
    // kernel's fixed cost, ~0.2 ms at K = 200, would buy nothing; safe_choice_kernel makes the same call)
    if (d.safe_mode && d.ctrl->n_resid <= kSafeResidSkip) return;
    if constexpr (!WALK) {
        prune_tile<NJ, RB>(d, job, q, blockIdx.x);
    } else {
        const long long ntiles = (prune_count(d) + 4 * 16 * RB - 1) / (4 * 16 * RB);
        bool first = true;
        for (long long bx = blockIdx.x; bx < ntiles; bx += gridDim.x) {
            if (!first) __syncthreads();                            // (the staging area is reused)
            first = false;
            prune_tile<NJ, RB>(d, job, q, (unsigned)bx);
        }
    }
    (void)qstride;
}

template <int NJ, bool WALK>
static void launch_mfma_prune_v(const Dev &d, const Job *job, double *q, long long qstride, unsigned gx, hipStream_t st) {
    const int lds = (4 * 32 * prune_row_stride(NJ * 16) + 4 * (176 + d.keep_stride)) * (int)sizeof(double);
    auto kern = score_mfma_prune_kernel<NJ, 2, (NJ <= 4 ? 2 : 1), WALK>;
    static PerDeviceLds attr;
    attr.ensure((const void *)kern, lds);
    hipLaunchKernelGGL(kern, dim3(gx, kMaxChunks), dim3(256), lds, st, d, job, q, qstride);
}

template <int NJ>
static void launch_mfma_prune(const Dev &d, const Job *job, double *q, long long qstride, long long max_rows,
                              hipStream_t st) {
    const unsigned gx = (unsigned)((max_rows + kMfmaRows - 1) / kMfmaRows);
    // (home_kernel has decided most of the window's rows: a short grid walks what is left)
    if (d.use_home && gx > 1024) launch_mfma_prune_v<NJ, true>(d, job, q, qstride, 1024, st);
    else launch_mfma_prune_v<NJ, false>(d, job, q, qstride, gx, st);
}

static void launch_diag_prune(const Dev &d, const Job *job, double *q, long long max_rows, hipStream_t st);

// Fresh-window scoring with pruning
bool launch_score_pruned(const Dev &d, const Job *job, double *q, long long qstride, long long max_rows,
                         hipStream_t st) {
    if (max_rows <= 0) return true;
    if (d.cov_type != COV_FULL) { launch_diag_prune(d, job, q, max_rows, st); return true; }
    switch (d.Dp / 16) {
        case 1: launch_mfma_prune<1>(d, job, q, qstride, max_rows, st); return true;
        case 2: launch_mfma_prune<2>(d, job, q, qstride, max_rows, st); return true;
        case 3: launch_mfma_prune<3>(d, job, q, qstride, max_rows, st); return true;
        case 4: launch_mfma_prune<4>(d, job, q, qstride, max_rows, st); return true;
        case 5: launch_mfma_prune<5>(d, job, q, qstride, max_rows, st); return true;
        case 6: launch_mfma_prune<6>(d, job, q, qstride, max_rows, st); return true;
        case 7: launch_mfma_prune<7>(d, job, q, qstride, max_rows, st); return true;
        case 8: launch_mfma_prune<8>(d, job, q, qstride, max_rows, st); return true;
        default: return false;
    }
}

// ------------------------------------------------------------------------------------------
// Pruned windows for diagonal / fixed-variance components.  Same contract as
// score_mfma_prune_kernel (evaluation order d.wrec, block-sparse output: keep64 label masks + one
// line of 16 values per kept (block, label); the values are log densities here).  Bound:
//   diag   sum_d log(1 + a_d) >= log(1 + sum_d a_d) >= log(1 + w_min |x - mu|^2),   w_min = min_d dw_d
//   fixed  sum_d (x_d - mu_d)^2 pp_d >= pp_min |x - mu|^2
// with |x - mu_t| >= |mu_t - mu_home| - |x - mu_home| from the centre-to-centre table (no distance
// work per pair at all).  A workgroup owns 64 consecutive visits (lane = visit, rows transposed in
// LDS); its 4 waves split the label range by 64-bit mask words; what survives the bound is scored
// exactly by the lanes (D logarithms / squares per pair).
// ------------------------------------------------------------------------------------------
// as-is log density of visit `lane` under slot s
__device__ __forceinline__ double diag_pair_score(const Dev &d, const double *__restrict__ xs, const double *__restrict__ xg,
                                                  int lane, int s) {
    const int D = d.D;
    const bool xlds = D <= kDiagLdsMaxD;
#define XAT(l) (xlds ? xs[(l) * kDiagLd + lane] : xg[(l)])
    const double *__restrict__ mu = d.mu + (long long)s * D;
    const double *__restrict__ dw = d.dw + (long long)s * D;
    double acc = 0.0;
    if (d.cov_type == COV_FIXED) {
        for (int l = 0; l < D; ++l) {
            const double dl = XAT(l) - mu[l];
            acc += (dl * dl) * dw[l];
        }
    } else {
        for (int l = 0; l < D; ++l) {
            const double dl = XAT(l) - mu[l];
            acc += log(1.0 + dl * dl * dw[l]);
        }
    }
    return d.sc[s].A - d.sc[s].half_vd * acc;
#undef XAT
}

__global__ __launch_bounds__(256) void score_diag_prune_kernel(Dev d, const Job *__restrict__ jobp,
                                                               double *__restrict__ q) {
    extern __shared__ __attribute__((aligned(16))) double xs[];   // [D][64], then the per-visit arrays
    const JobView job = load_job(jobp);
    if (!job_is_pruned(d, job.mode, job.prune)) return;
    const long long nrows = d.ctrl->n_sorted;                     // rows of the bucket sort (all of the window here)
    const long long k0 = (long long)blockIdx.x * kValuRows;
    if (k0 >= nrows) return;
    const int D = d.D, K = job.nlist;
    const bool xlds = D <= kDiagLdsMaxD;                          // (beyond: no tile, the rows through the cache)
    double *__restrict__ sM = xs + (xlds ? D * kDiagLd : 0);      // best-score lower bound per visit
    double *__restrict__ sRho = sM + kValuRows;                   // |x - mu_home|
    long long *__restrict__ sI = (long long *)(sRho + kValuRows); // data index (-1: dead row)
    int *__restrict__ sH = (int *)(sI + kValuRows);               // home slot
    int *__restrict__ sLab = sH + kValuRows;                      // its label
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < kValuRows) {
        const long long k = k0 + tid;
        if (k < nrows) {
            const WRec r = d.wrec[k];
            sI[tid] = r.i; sH[tid] = r.home; sLab[tid] = r.home_label; sM[tid] = r.mlb0;
        } else {
            sI[tid] = -1; sH[tid] = -2; sLab[tid] = -1; sM[tid] = INFINITY;
        }
    }
    __syncthreads();
    // (8 row-contiguous loads in flight per thread, then the transposing LDS writes)
    for (int e0 = tid; xlds && e0 < kValuRows * D; e0 += 256 * 8) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = e0 + 256 * j;
            const bool in = e < kValuRows * D;
            const long long i = in ? sI[e / D] : -1;
            v[j] = d.X[(i >= 0 ? i : 0) * D + (in ? e % D : 0)];
            if (i < 0) v[j] = 0.0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = e0 + 256 * j;
            if (e < kValuRows * D) xs[(e % D) * kDiagLd + e / D] = v[j];
        }
    }
    __syncthreads();
    // the home component of every visit: distance to its mean (the radius of the triangle bound)
    // and its exact one-point-removed score (raises the visit's best-score bound, and is the value
    // the label loop stores for the home).  The D dimensions are split over the 4 waves; partial
    // sums meet in LDS.
    double *__restrict__ sPart = (double *)(sLab + kValuRows);    // [4 waves][3][64]
    double *__restrict__ sHomeLp = sPart + 12 * kValuRows;        // exact home score (no seating weight)
    const int hv = sH[lane];
    const bool live = sI[lane] >= 0;
    const double *__restrict__ xg = d.X + (live ? sI[lane] : 0) * D;
#define XAT(l) (xlds ? xs[(l) * kDiagLd + lane] : xg[(l)])
    const bool fixed = d.cov_type == COV_FIXED;
    const int nhv = hv >= 0 ? d.n[hv] : 0;
    {
        double r2 = 0.0, t1 = 0.0, t2 = 0.0;                      // radius^2; the two sums of the home form
        if (hv >= 0) {
            const double *__restrict__ mh = d.mu + (long long)hv * D;
            const double *__restrict__ mS = d.m + (long long)hv * D;
            const double *__restrict__ SS = d.S + (long long)hv * (fixed ? 2 * D : D);
            const double k1 = d.k0 + (double)(nhv - 1);
            const long long v1 = d.v0 + nhv - 1;
            const double scale1 = (k1 + 1.0) / (k1 * (double)v1), inv_v1 = 1.0 / (double)v1;
            for (int l = w; l < D; l += 4) {
                const double x = XAT(l);
                const double t = x - mh[l];
                r2 = fma(t, t, r2);
                if (nhv >= 2) {
                    if (fixed) {        // gaussian_components_fixedvar.py:164-176: numerator -= p x, precision_N -= p
                        const double p = d.prior_S[D + l];
                        const double mn = __dsub_rn(mS[l], __dmul_rn(p, x));
                        const double pN = __dsub_rn(SS[l], p);
                        const double pp = pN * p / (pN + p);
                        const double dl = x - mn / pN;
                        t1 += log(pp);
                        t2 += (dl * dl) * pp;
                    } else {            // gaussian_components_diag.py:178-193
                        const double m1 = __dsub_rn(mS[l], x);
                        const double S1 = __dsub_rn(SS[l], __dmul_rn(x, x));
                        const double mean = m1 / k1;
                        const double var = scale1 * (S1 - k1 * (mean * mean));
                        const double dl = x - mean;
                        t1 += log(var);
                        t2 += log(1.0 + inv_v1 * (dl * dl) * (1.0 / var));
                    }
                }
            }
        }
        sPart[(w * 3 + 0) * kValuRows + lane] = r2;
        sPart[(w * 3 + 1) * kValuRows + lane] = t1;
        sPart[(w * 3 + 2) * kValuRows + lane] = t2;
    }
    __syncthreads();
    if (w == 0) {
        double r2 = 0.0, t1 = 0.0, t2 = 0.0;
        for (int ww = 0; ww < 4; ++ww) {
            r2 += sPart[(ww * 3 + 0) * kValuRows + lane];
            t1 += sPart[(ww * 3 + 1) * kValuRows + lane];
            t2 += sPart[(ww * 3 + 2) * kValuRows + lane];
        }
        sRho[lane] = sqrt(r2) * (1.0 + 1e-9);
        if (hv >= 0 && nhv >= 2) {
            const long long v1 = d.v0 + nhv - 1;
            const double lp = fixed ? -0.5 * (double)D * log(2.0 * 3.14159265358979323846) + 0.5 * t1 - 0.5 * t2
                                    : (double)D * (d.tab_lgam[v1 + 1] - d.tab_lgam[v1] - 0.5 * d.tab_log[v1] - 0.5 * BGMM_LOG_PI)
                                          - 0.5 * t1 - 0.5 * (double)(v1 + 1) * t2;
            sHomeLp[lane] = lp;
            sM[lane] = fmax(sM[lane], d.sc[hv].logseat1 + lp);
            if (live) {                 // what certify_kernel reads next sweep (bgmm_device.h: PCache)
                PCache pc;
                pc.tag = ((long long)hv << 32) | (unsigned int)d.mu_ver[hv];
                pc.qhome = lp;
                pc.rho2 = r2;
                pc.pad = 0.0;
                d.pcache[sI[lane]] = pc;
            }
        }
    }
    __syncthreads();
    const double rho = sRho[lane], thr = sM[lane] - kPruneMargin;
    const int labv = sLab[lane];
    const long long blk0 = k0 >> 4;
    unsigned n_kept = 0, n_bound = 0;
    const int nw = (K + 63) >> 6;
    // Coarse pass, lane = LABEL: with at most four distinct homes among the 64 visits (they are
    // sorted by home: usually one), a label is tested once per home against the largest radius and
    // the weakest threshold of that home's visits; only the labels that survive are looked at per
    // visit below.
    int hs[4] = {-1, -1, -1, -1}, hl[4] = {0, 0, 0, 0}, nh = 0;
    double hr[4] = {0.0, 0.0, 0.0, 0.0}, ht[4] = {0.0, 0.0, 0.0, 0.0};
    bool coarse_ok;
    {
        unsigned long long pending = __ballot(live && hv >= 0);
        const bool unassigned = __ballot(live && hv < 0) != 0ull;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            if (pending) {
                const int first = __ffsll((long long)pending) - 1;
                const int sh = __builtin_amdgcn_readfirstlane(__shfl(hv, first));
                const bool sel = live && hv == sh;
                pending &= ~__ballot(sel);
                double a = sel ? rho : 0.0, bmin = sel ? thr : INFINITY;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    a = fmax(a, __shfl_xor(a, o));
                    bmin = fmin(bmin, __shfl_xor(bmin, o));
                }
                hs[it] = sh; hl[it] = __builtin_amdgcn_readfirstlane(__shfl(labv, first));
                hr[it] = a; ht[it] = bmin;
                nh = it + 1;
            }
        }
        coarse_ok = !unassigned && pending == 0ull;
    }
    for (int wi = w; wi < nw; wi += 4) {
        unsigned long long mword0 = 0, mword1 = 0, mword2 = 0, mword3 = 0;
        unsigned long long cmask;
        {
            const int t = wi * 64 + lane;
            bool cand = t < K;
            n_bound += 4 * __popcll(__ballot(cand));
            if (coarse_ok && cand) {
                const int st = d.perm[t];
                const SlotConst *__restrict__ scp = d.sc + st;
                const double base = scp->logseat + scp->A, hvd = scp->half_vd, tcoef = scp->inv_lam * scp->inv_cv;
                cand = false;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (j < nh) {
                        double dl = d.pr_dcc[(long long)hl[j] * d.nslots + t] * (1.0 - 1e-9) - hr[j];
                        dl = dl > 0.0 ? dl : 0.0;
                        const double tt = dl * dl * tcoef;
                        const double ub = base - hvd * (fixed ? tt : log1p_lower(tt));
                        cand = cand || ub >= ht[j] || st == hs[j];
                    }
                }
            }
            cmask = __ballot(cand);
        }
        while (cmask) {
            const int t = wi * 64 + __ffsll((long long)cmask) - 1;
            cmask &= cmask - 1;
            const int s = d.perm[t];
            const SlotConst *__restrict__ scp = d.sc + s;
            const double base = scp->logseat + scp->A, hvd = scp->half_vd, tcoef = scp->inv_lam * scp->inv_cv;
            bool need = false;
            if (live) {
                if (hv == s) need = true;
                else if (labv < 0) need = true;                 // unassigned visit: no centre to bound from
                else {
                    double dl = d.pr_dcc[(long long)labv * d.nslots + t] * (1.0 - 1e-9) - rho;
                    dl = dl > 0.0 ? dl : 0.0;
                    const double tt = dl * dl * tcoef;
                    const double ub = base - hvd * (fixed ? tt : log1p_lower(tt));
                    need = ub >= thr;
                }
            }
            const unsigned long long bl = __ballot(need);
            if (bl == 0ull) continue;
            // somebody needs this label: exact scores for the 16-visit blocks that do
            const unsigned long long bit = 1ull << (t & 63);
            if (bl & 0xFFFFull) mword0 |= bit;
            if (bl & 0xFFFF0000ull) mword1 |= bit;
            if (bl & 0xFFFF00000000ull) mword2 |= bit;
            if (bl & 0xFFFF000000000000ull) mword3 |= bit;
            n_kept += ((bl & 0xFFFFull) != 0) + ((bl & 0xFFFF0000ull) != 0) + ((bl & 0xFFFF00000000ull) != 0)
                      + ((bl & 0xFFFF000000000000ull) != 0);
            const bool my_block = ((bl >> (lane & 48)) & 0xFFFFull) != 0ull;
            if (my_block && live) {
                const bool own = hv == s && nhv >= 2;
                const double lp = own ? sHomeLp[lane] : diag_pair_score(d, xs, xg, lane, s);
                q[((blk0 + (lane >> 4)) * (long long)d.nslots + t) * 16 + (lane & 15)] = lp;
            }
        }
        if (lane == 0) {
            unsigned long long *__restrict__ kp = d.keep64 + blk0 * d.keep_stride + wi;
            if (k0 < nrows) kp[0] = mword0;
            if (k0 + 16 < nrows) kp[d.keep_stride] = mword1;
            if (k0 + 32 < nrows) kp[2 * d.keep_stride] = mword2;
            if (k0 + 48 < nrows) kp[3 * d.keep_stride] = mword3;
        }
    }
    if (lane == 0) {
        atomicAdd(&d.pr_counts[blockIdx.x & 255], (unsigned long long)n_kept);
        atomicAdd(&d.pr_counts[256 + (blockIdx.x & 255)], (unsigned long long)n_bound);
    }
}
#undef XAT

static void launch_diag_prune(const Dev &d, const Job *job, double *q, long long max_rows, hipStream_t st) {
    const unsigned gx = (unsigned)((max_rows + kValuRows - 1) / kValuRows);
    const int lds = ((d.D <= kDiagLdsMaxD ? d.D * kDiagLd : 0) + 17 * kValuRows) * (int)sizeof(double);   // tile + per-visit arrays + partial sums
    hipLaunchKernelGGL(score_diag_prune_kernel, dim3(gx), dim3(256), lds, st, d, job, q);
}


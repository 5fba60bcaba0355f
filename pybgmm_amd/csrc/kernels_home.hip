// Pruned windows, first pass: every visit against its OWN component, decided on the spot when nothing
// else can matter.
//
// In a chain whose movers are sparse almost every visit has two candidates: the component it sits in and
// a new table (igmm/crpmm.py:74).  Every other component is excluded by the triangle bound of the pruning
// kernel, tabulated per home over the distance to the home's mean (ftab, kernels_state.hip:
// prune_ftable_kernel): its weight in the draw is below e^-80 of the best (kernels_prune.hip).  For such a
// visit the whole reassignment step of the reference -- del_item's one-point-removed predictive
// (gaussian_components.py:171-186, 228-251), the seating weights (crpmm.py:68-75), the draw
// (utils/utils.py:7-20) -- needs one exact quadratic form.  home_kernel streams the window's rows once
// (they come grouped by home from the bucket sort), evaluates that form with v_mfma_f64_16x16x4_f64, the
// distance to the home's mean on the side, draws, and leaves the per-point caches certify_kernel reads next
// sweep.  Visits the table bound cannot decide (and unassigned / singleton ones) are appended to the
// residual list, which score_mfma_prune_kernel / choice_sparse_kernel work through as before.
//
// Shape of the launch.  The rows of X are a random gather (512 bytes each at D = 64), and on this chip the
// FP64 matrix pipe is no faster than the gather (2 560 pipe cycles per 16 rows at D = 64 against ~80 us for the
// 512 MB): both have to be kept busy at once.  Loads return in order per wavefront (one vmcnt), so a wavefront
// that streams factor tiles from L2 behind a row prefetch waits for the prefetch at the first tile.  Hence:
//   * persistent workgroups (as many as are resident), each with a contiguous run of 256-row blocks, the runs
//     dealt so that the workgroups of one XCD (one L2) hold neighbouring ones;
//   * the inverse factor of the block's home -- the MFMA B operand, 20 KB at D = 64, 72 KB at D = 128 -- sits in
//     LDS, loaded when the home changes (the rows are grouped by home: about once per workgroup), together
//     with the home's constants; the matrix loop reads it with ds_read (lgkmcnt), the only traffic on vmcnt
//     are the rows and their records;
//   * the rows go from memory STRAIGHT into A-operand registers: lane (lr, lk) of a 16-row tile takes 16 bytes
//     of row lr at column 8 j + 2 lk, j = 0 .. D/8 - 1 (64 contiguous bytes per row and instruction; measured
//     5.1 TB/s for this pattern against 6.5 for whole-row loads, tools/gather_bw.hip).  The two doubles are the
//     lane's entries of k-slices 2 j and 2 j + 1, i.e. the 16 columns of a block are dealt to the four k-lanes as
//     {0,2,4,6 | 1,3,5,7 | 8,10,12,14 | 9,11,13,15} instead of {0..3 | 4..7 | ...}; the factor fragments are
//     permuted to match when they are copied into LDS (a sum over the same products in another order).  No
//     staging tile, no transposition, a tile in flight costs its 2 D/16 registers and nothing else;
//   * every wavefront (64 rows of the block, 16 at a time) has the next tiles on their way while the current one
//     is in the matrix pipe (three of them up to D = 32); records a block ahead, issued before the block's row loads;
//     the visit's uniform is fetched only when its draw depends on it;
//   * the evaluation order comes padded (bucket_prefix_kernel: every home's run of rows fills whole 256-row blocks), so a
//     block has ONE home -- or is a block of the unassigned bin, whose rows go on the residual list as they always did.
//     (Until round 5 a block could straddle two homes and took a "general path" here -- factor tiles from L2 through a
//     register ring, home by home; with the padded order that path only ever saw unassigned rows, computed nothing for
//     them, and cost the hot loop 25 registers, 2 of them spilled.  Round 6: a block that is not one home's passes its
//     rows on, whatever they are.)
//   * SAFE (template): the proof pass of a safe-stay window (kernels_safe.hip) is its own instantiation -- its tail, its
//     tables and its constants are not live in the kernel that decides a chain at rest.
// Per wavefront lane = row for everything scalar (the record, the tail); quadratic forms and distances meet their
// rows' lanes through LDS.  The tail is short on purpose: FP64 VALU instructions queue behind the other
// wavefronts' MFMAs (the same pipe), so a visit whose new table weighs less than 2^-53 of its home is decided
// without the exponentials (their outcome is exactly 1).
#include "score_common.h"
#include "wave_ops.h"
#include "fast_math.h"

#define LDS_AS __attribute__((address_space(3)))
#ifdef BGMM_HOME_TEMPORAL
#define HOME_ROW_LOAD(p) (*(p))
#else
#define HOME_ROW_LOAD(p) __builtin_nontemporal_load(p)
#endif

#ifdef BGMM_HOME_PROF
#define HP(k) { const long long tk1_ = clock64(); pf[k] += tk1_ - tk0; tk0 = tk1_; }
#else
#define HP(k)
#endif

// A visit is decided here when every component but its home lies, by the table bound, more than
// 38 + log(K + 1) nats below the better of its two candidates -- certify_kernel's margin (kernels_prune.hip):
// together they weigh less than e^-38 = 0.28 * 2^-53 of the total.  (The pruning kernel behind this one keeps
// its 80 nats: it drops labels one by one.)
static constexpr double kHomeFar = 80.0;             // beyond this the excluded labels count as K e^-80 in log_alt

// LDS plan (doubles).  Per workgroup: the home's factor fragments (permuted), cvec, mu (permuted, zero padded),
// its row of ftab, 16 scalars; per wavefront: quadratic forms, distances, home slots of its 64 rows.
// neighbours scored exactly (kHomeNbr of them) up to D = 32: their fragments, cvec and 8 scalars each + the second
// bound table + the candidates' order; per wavefront their quadratic forms
__host__ __device__ constexpr int home_nbr(int Dp) { return Dp <= 32 ? kHomeNbr : 0; }
__host__ __device__ constexpr int home_wave_doubles(int Dp) { return 64 + 64 + 32 + 64 * home_nbr(Dp); }
__host__ __device__ constexpr int home_shared_doubles(int Dp) {
    return bgmm_nfrag(Dp) * 64 + 2 * Dp + 64 + 16 + 8 + (home_nbr(Dp) ? home_nbr(Dp) * (bgmm_nfrag(Dp) * 64 + Dp + 8) + 64 + 8 : 0);
}
__host__ __device__ constexpr int home_lds_bytes(int Dp) { return (home_shared_doubles(Dp) + 4 * home_wave_doubles(Dp)) * 8; }
// wavefronts per SIMD (a tile of 16 rows is 2 D/16 registers; one in the matrix pipe, one on its way): two up to
// D = 64, one above (at two the D = 128 kernel spills, and a scratch access waits for every row load in flight)
__host__ __device__ constexpr int home_waves_per_simd(int NJ) {
    const int by_lds = (160 * 1024) / home_lds_bytes(NJ * 16);
    const int want = NJ <= 4 ? 2 : 1;
    return by_lds < want ? (by_lds < 1 ? 1 : by_lds) : want;
}

// register slots for tiles (one in the matrix pipe, the others on their way; 2 D/16 registers each)
__host__ __device__ constexpr int home_slots(int NJ) { return NJ <= 2 ? 4 : 2; }

// column of X (inside its block of 16) that k-lane lk holds in k-slice kq (0..3) of the block
__host__ __device__ constexpr int home_col(int kq, int lk) { return 8 * (kq >> 1) + 2 * lk + (kq & 1); }

typedef double home_d2 __attribute__((ext_vector_type(2), aligned(8)));

// Four values per lane, each to be summed over the 16 lanes of its row: a transposing reduction -- after two
// exchange steps inside the quads a lane carries ONE of the four (the one numbered lane & 3), two rotations by whole
// quads finish it.  5 additions and 10 DPP moves instead of 16 and 32 (FP64 VALU instructions share the pipe with the
// other wavefronts' MFMAs).  Returns the sum of a[lane & 3] over the row, in every lane.
template <int CTRL>
__device__ __forceinline__ double home_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double home_row_sum4(const double (&a)[4], int lane) {
    const bool p = lane & 1, q = lane & 2;
    const double k0 = p ? a[1] : a[0], s0 = p ? a[0] : a[1];
    const double k1 = p ? a[3] : a[2], s1 = p ? a[2] : a[3];
    const double b0 = k0 + home_dpp<0xB1>(s0);          // quad_perm [1,0,3,2]: pair sums of a[p]
    const double b1 = k1 + home_dpp<0xB1>(s1);          //                                   a[2 + p]
    const double kk = q ? b1 : b0, ss = q ? b0 : b1;
    double c = kk + home_dpp<0x4E>(ss);                 // quad_perm [2,3,0,1]: quad sum of a[lane & 3]
    c += home_dpp<0x124>(c);                            // row_ror:4
    c += home_dpp<0x128>(c);                            // row_ror:8
    return c;
}

// minorant of log(1 + t), t >= 0 (kernels_prune.hip: log1p_lower): 1 + t = m 2^e, the chord of log over [0.5, 1]
__device__ __forceinline__ double home_log1p_lower(double t) {
    const double y = 1.0 + t;
    const double m = __builtin_amdgcn_frexp_mant(y);
    const int e = __builtin_amdgcn_frexp_exp(y);
    return 0.6931471805599453 * ((double)(e - 2) + 2.0 * m);
}

// WHOLE: D is a multiple of 16 (no padded columns: every 16-byte piece of a tile lies inside its row)
template <int NJ, bool WHOLE, bool SAFE>
__global__ __launch_bounds__(256, home_waves_per_simd(NJ)) void home_kernel(Dev d) {
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    Ctrl *c = d.ctrl;
    if (!job_is_pruned(d, c->job.mode, c->job.prune) || (SAFE && c->safe_epoch_valid)) return;
    // (a short step queued neither the table kernels nor -- short_step 1 -- the bucket sort: without them there is
    // nothing sound to do here, and apply_kernel will refuse the step)
    if (d.short_step && (!c->tables_valid || (d.short_step == 1 && !c->skip_sort))) return;
    constexpr int Dp = NJ * 16, NF = 2 * NJ * (NJ + 1), NKK = NJ * 4, NJ8 = NJ * 2;
    constexpr int LRING = pick_ring(NF, 4);                           // factor tiles in flight from LDS
    constexpr int NS = home_slots(NJ);
    constexpr int NB = (WHOLE && !SAFE) ? home_nbr(NJ * 16) : 0;     // neighbours of the home scored exactly (D = 16, 32)
    const long long nrows = c->n_sorted_pad;                          // (every home's run padded to whole blocks: bucket_prefix_kernel)
    // the records were written for this very window (bucket_scatter_kernel ran in front): they carry the visits' uniforms
    const bool u_in_rec = !c->skip_sort;
    const long long nblocks = (nrows + 255) >> 8;
    const int D = d.D, K = c->job.K;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the workgroup's run of blocks
    long long b0, b1;
    {
        const int nb = (int)gridDim.x, b = (int)blockIdx.x;
        const int per = nb >> 3, rem = nb & 7, xcd = b & 7;
        const int lb = xcd * per + (xcd < rem ? xcd : rem) + (b >> 3);
        const long long bpw = (nblocks + nb - 1) / nb;
        b0 = (long long)lb * bpw;
        b1 = b0 + bpw < nblocks ? b0 + bpw : nblocks;
    }
    if (b0 >= b1) return;
    const int lr = lane & 15, lk = lane >> 4;
    LDS_AS double *const L = (LDS_AS double *)lds_all;
    LDS_AS double *const Bf = L;                                      // [NF][64] the home's factor fragments, permuted
    LDS_AS double *const hcv = Bf + NF * 64;                          // cvec [Dp]
    LDS_AS double *const hmu = hcv + Dp;                              // hmu[4 kk + lk] = mu[column of (kk, lk)], zero beyond D
    LDS_AS double *const hft = hmu + Dp;                              // ftab[label of the home][64]
    LDS_AS double *const hsc = hft + 64;                              // SlotConst (12), finv, (n, version)
    LDS_AS double *const hsr = hsc + 16;                              // safe-stay windows: the home's robust constants (rtab row)
    // (NB > 0) the neighbours: fragments [NB][NF][64], cvec [NB][Dp], scalars [NB][8] = {logseat + A, half_vd, inv_cv, label,
    // slot}, the second bound table [64], [8] ints: how many neighbours, then the candidates (home = -1, neighbour m) in
    // label order
    LDS_AS double *const nBf = hsr + 8;
    LDS_AS double *const ncv = nBf + NB * NF * 64;
    LDS_AS double *const nsc = ncv + NB * Dp;
    LDS_AS double *const hft2 = nsc + NB * 8;
    LDS_AS int *const ncand = (LDS_AS int *)(hft2 + (NB ? 64 : 0));
    LDS_AS double *const wave0 = hsr + 8 + (NB ? NB * (NF * 64 + Dp + 8) + 64 + 8 : 0);
    LDS_AS double *const sideQ = wave0 + w * home_wave_doubles(Dp);   // exact home form of row rho [64]
    LDS_AS double *const sideRho = sideQ + 64;                        // |x - mu_home|^2 [64]
    LDS_AS int *const sideH = (LDS_AS int *)(sideRho + 64);           // home slot [64]
    LDS_AS double *const sideQn = sideRho + 64 + 32;                  // (NB > 0) [NB][64] the rows' forms under the neighbours
    const bool keep_caches = d.use_certify != 0;                      // (nobody reads the per-point caches otherwise)
    const long long win_base = c->job.win_base;
    const long long epoch = c->state_epoch;
    // (With certified stays on, a visit decided here should come back certified next sweep: that needs its
    //  alternatives' weight known to e^-37.75 of the home's (tier 1), which the table bound delivers only when it lies
    //  far below -- so the wide margin there, and the rows in between go to the pruning kernel, whose draw kernel
    //  sums the alternatives exactly.)
    const double margin = keep_caches ? kHomeFar : 38.0 + fm_log((double)K + 1.0);

    // lane rho's row of the current block and of the next one (record, window row, uniform: a block ahead),
    // the homes of the block's first and last row
    WRec rcur, rnext;
    int wrow_cur = 0, wrow_next = 0, hf_cur = -3, hl_cur = -3, hf_next = -3, hl_next = -3;
    // (Unconditional loads from clamped indices, validity kept on the side: a load inside a divergent branch is
    //  waited for where the branch joins -- and with it every row load in flight.)
#define HOME_LOAD_REC(B, R, WR, HF, HL, OK)                                                \
    {                                                                                      \
        const bool bok_ = (B) < b1;                                                        \
        const long long kf_ = bok_ ? (B) * 256 : 0, k_ = kf_ + w * 64 + lane;              \
        const long long kl_ = kf_ + 255 < nrows ? kf_ + 255 : nrows - 1;                   \
        const long long kc_ = k_ < nrows ? k_ : nrows - 1;                                 \
        HF = d.wrec[kf_].home; HL = d.wrec[kl_].home;                                      \
        R = d.wrec[kc_]; WR = d.wperm[kc_];                                                \
        OK = bok_ && k_ < nrows;                                                           \
    }
    bool ok_cur, ok_next = false;
    HOME_LOAD_REC(b0, rcur, wrow_cur, hf_cur, hl_cur, ok_cur)
    rnext = rcur; wrow_next = 0;
    // A tile on its way in: lane (lr, lk) fetches 16 bytes of row lr per instruction.  The lane's row offset comes
    // from the lane that owns the row's record (ds_bpermute); columns beyond D read column 0 and are zeroed.
    // NS register slots hold tiles: tile t of a block sits in slot t % NS (4 tiles per block, NS = 2 or 4: the slot
    // of a tile is a compile-time constant), NS - 1 tiles are on their way while one is in the matrix pipe.
    double xt[NS][NKK];
    int coff[WHOLE ? 1 : NKK];                                       // (padded columns: where the lane's entries are, -1 = none)
    if (!WHOLE) {
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const int col = 16 * (kk >> 2) + home_col(kk & 3, lk);
            coff[WHOLE ? 0 : kk] = col < D ? col : -1;
        }
    }
#define HOME_ISSUE(DST, XO, TN)                                                            \
    {                                                                                      \
        const int src_ = 16 * (TN) + lr;                                                   \
        const long long o_ = ((long long)__shfl((int)((XO) >> 32), src_) << 32) | (unsigned int)__shfl((int)(XO), src_); \
        if (WHOLE) {                                                                       \
            const home_d2 *__restrict__ xrow = (const home_d2 *)(d.X + o_ + 2 * lk);       \
            _Pragma("unroll") for (int j = 0; j < NJ8; ++j) {                              \
                const home_d2 v_ = HOME_ROW_LOAD(xrow + 4 * j);                            \
                DST[2 * j] = v_.x; DST[2 * j + 1] = v_.y;                                  \
            }                                                                              \
        } else {                                                                           \
            const double *__restrict__ xrow = d.X + o_;                                    \
            _Pragma("unroll") for (int kk = 0; kk < NKK; ++kk) {                           \
                const int co_ = coff[WHOLE ? 0 : kk];                                      \
                const double a_ = __builtin_nontemporal_load(xrow + (co_ >= 0 ? co_ : 0)); \
                DST[kk] = co_ >= 0 ? a_ : 0.0;                                             \
            }                                                                              \
        }                                                                                  \
    }
    // (element offset of the lane's row in X: one 64-bit multiply per record)
    // of the current block's rows, of the next block's (wanted NS - 1 tiles before that block starts: its record
    // index is fetched a block earlier than the rest of the record)
    long long xo_cur = (ok_cur && rcur.i >= 0 ? rcur.i : 0) * D, xo_next, i_after = 0;
    {
        const long long k_ = (b0 + 1) * 256 + w * 64 + lane;
        const long long i1 = d.wrec[k_ < nrows ? k_ : nrows - 1].i;
        xo_next = (b0 + 1 < b1 && k_ < nrows && i1 >= 0 ? i1 : 0) * D;
    }
#pragma unroll
    for (int t = 0; t < NS - 1; ++t) HOME_ISSUE(xt[t], xo_cur, t)
    unsigned n_mfma = 0, n_homes = 0;
    int cur_home = -1;
#ifdef BGMM_HOME_PROF
    long long pf[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tk0 = clock64();
#endif
#pragma unroll 1
    for (long long b = b0; b < b1; ++b) {
        const long long imine = ok_cur ? rcur.i : -1;
        const int hmine = ok_cur ? rcur.home : -2;
        sideH[lane] = hmine;
        // The next block's records and this block's uniforms set off here, BEFORE the block's row loads: loads return
        // in order, and they are wanted at the last tile (the next block's first rows) and in the tail -- by then
        // everything older has been consumed anyway, and nothing younger is held up.
        HOME_LOAD_REC(b + 1, rnext, wrow_next, hf_next, hl_next, ok_next)
        const long long k_after = (b + 2) * 256 + w * 64 + lane;
        const bool ok_after = b + 2 < b1 && k_after < nrows;
        i_after = d.wrec[ok_after ? k_after : 0].i;
        const int hf = __builtin_amdgcn_readfirstlane(hf_cur), hl = __builtin_amdgcn_readfirstlane(hl_cur);
        const bool one_home = hf == hl && hf >= 0;                    // (the same decision in all four wavefronts)
        HP(6)
        if (one_home && hf != cur_home) {
            __syncthreads();                                          // everybody is done with the previous home
            // (read in storage order, two doubles per thread and step; written where the permutation puts them:
            //  column c16 = 4 (ks % 4) + lks of its block goes to k-slice 2 (c16 / 8) + (c16 & 1), k-lane (c16 & 7) / 2
            //  -- six loads in flight per thread: one round trip to L2 per batch instead of one per piece)
            auto copy_factor = [&](int slot, LDS_AS double *dstB) {
                const double *__restrict__ src = d.Wfrag + (long long)slot * (NF * 64);
                constexpr int FB = 6;
                for (int e0 = tid; e0 < NF * 32; e0 += 256 * FB) {
                    home_d2 v[FB];
#pragma unroll
                    for (int k = 0; k < FB; ++k) {
                        const int e2 = e0 + 256 * k;
                        v[k] = ((const home_d2 *)src)[e2 < NF * 32 ? e2 : tid];
                    }
#pragma unroll
                    for (int k = 0; k < FB; ++k) {
                        const int e2 = e0 + 256 * k;
                        const int e = 2 * e2, f = e >> 6, ln = e & 63;       // lanes ln, ln + 1: same fragment, same k-lane
                        const int ks = f & 3, lks = ln >> 4;                  // (2 J (J + 1) is a multiple of 4)
                        const int c16 = 4 * ks + lks;
                        const int dst = (f - ks + 2 * (c16 >> 3) + (c16 & 1)) * 64 + (ln & 15) + 16 * ((c16 & 7) >> 1);
                        if (e2 < NF * 32) { dstB[dst] = v[k].x; dstB[dst + 1] = v[k].y; }
                    }
                }
            };
            copy_factor(hf, Bf);
            const int a = d.label_of_slot[hf];
            if (NB > 0) {
                // the home's neighbours (prune_ftable_kernel): factor, cvec, the constants of the as-is predictive
                const int *__restrict__ nl = d.nbr + (long long)a * 4;
                int nn = 0;
#pragma unroll
                for (int m = 0; m < NB; ++m) {
                    const int t = nl[m];
                    if (t < 0) continue;
                    const int s = d.perm[t];
                    copy_factor(s, nBf + nn * NF * 64);
                    for (int e = tid; e < Dp; e += 256) ncv[nn * Dp + e] = d.cvec[(long long)s * d.Dp + e];
                    if (tid == 100 + m) {
                        const SlotConst sc = d.sc[s];
                        nsc[nn * 8 + 0] = sc.logseat + sc.A; nsc[nn * 8 + 1] = sc.half_vd; nsc[nn * 8 + 2] = sc.inv_cv;
                        LDS_AS int *ni = (LDS_AS int *)(nsc + nn * 8 + 3);
                        ni[0] = t; ni[1] = s;
                    }
                    ++nn;
                }
                if (tid >= 128 && tid < 192) hft2[tid - 128] = d.ftab2[(long long)a * 64 + (tid - 128)];
                if (tid == 95) {
                    // the candidates in label order: the neighbours come sorted, the home goes in where its label belongs
                    ncand[0] = nn;
                    int pos = 1, mm = 0;
                    bool home_in = false;
                    for (int m = 0; m < NB; ++m) {
                        const int t = nl[m];
                        if (t < 0) continue;
                        if (!home_in && a < t) { ncand[pos++] = -1; home_in = true; }
                        ncand[pos++] = mm++;
                    }
                    if (!home_in) ncand[pos++] = -1;
                }
            }
            for (int e = tid; e < Dp; e += 256) {
                hcv[e] = d.cvec[(long long)hf * d.Dp + e];
                const int col = 16 * (e >> 4) + home_col((e >> 2) & 3, e & 3);     // e = 4 kk + lk
                hmu[e] = col < D ? d.mu[(long long)hf * D + col] : 0.0;
            }
            if (tid < 64) hft[tid] = (SAFE ? d.ftabR : d.ftab)[(long long)a * 64 + tid];
            if (SAFE && tid >= 128 && tid < 136) hsr[tid - 128] = d.rtab[(long long)a * 8 + (tid - 128)];
            if (tid == 64) {
                const SlotConst sc = d.sc[hf];
                hsc[0] = sc.A; hsc[1] = sc.half_vd; hsc[2] = sc.inv_cv; hsc[3] = sc.A1; hsc[4] = sc.half_vd1;
                hsc[5] = sc.coef1; hsc[6] = sc.a1; hsc[7] = sc.logdetC; hsc[8] = sc.logseat; hsc[9] = sc.logseat1;
                hsc[10] = sc.inv_lam; hsc[11] = sc.mu2; hsc[12] = d.finv[a];
                LDS_AS int *hi = (LDS_AS int *)(hsc + 13);
                hi[0] = d.n[hf]; hi[1] = d.mu_ver[hf];
            }
            cur_home = hf;
            __syncthreads();
        }
        HP(7)
        if (one_home) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int r0 = 16 * t;
                asm volatile("" ::: "memory");      // (keeps the home's LDS constants from being hoisted into registers)
                const double (&xf)[NKK] = xt[t % NS];
                HP(0)
                // tile t + NS - 1 sets off into the slot tile t - 1 has left (beyond the block: the next block's rows)
                if (t + NS - 1 < 4) { HOME_ISSUE(xt[(t + NS - 1) % NS], xo_cur, t + NS - 1) }
                else { HOME_ISSUE(xt[(t + NS - 1) % NS], xo_next, t + NS - 1 - 4) }
                HP(1)
                {
                // ---- every row of the tile under the block's home: factor and constants from LDS
                double dpart = 0.0;
                LDS_AS const double *const wf = Bf + lane;
                double ringk[LRING];
#pragma unroll
                for (int i = 0; i < LRING; ++i) ringk[i] = wf[i * 64];
                double qp[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int J = 0; J < NJ; ++J) {
                    // (the distance to the home's mean, one block of 16 columns per block row: four constants live at a
                    //  time instead of D/4, and the FP64 VALU work spread between the MFMAs)
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int kk = 4 * J; kk < 4 * J + 4; ++kk) { const double tq = xf[kk] - hmu[4 * kk + lk]; dpart = fma(tq, tq, dpart); }
                    const double cj = hcv[16 * J + lr];
                    v4d acc = (v4d){cj, cj, cj, cj};
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
                n_mfma += NF;
                n_homes += 1;
                {
                    // (row lk + 4 r of the tile, r = lane & 3, in the row group's lanes 0 .. 3)
                    const double v = home_row_sum4(qp, lane);
                    if (lr < 4) sideQ[r0 + lk + 4 * lr] = v;
                }
                dpart += __shfl_xor(dpart, 16);
                dpart += __shfl_xor(dpart, 32);
                if (lk == 0) sideRho[r0 + lr] = dpart;
                if (NB > 0) {
                    // ---- the same rows under the home's neighbours (their as-is forms: nobody is removed from them)
                    const int nn = __builtin_amdgcn_readfirstlane(ncand[0]);
                    // (a neighbour's fragments come into registers in one batch, the next neighbour's while this one's
                    //  products run: a ds_read per MFMA would put an LDS round trip between every two of them)
#pragma unroll
                    for (int m = 0; m < NB; ++m) {
                        if (m >= nn) break;
                        LDS_AS const double *const wn = nBf + m * NF * 64 + lane;
                        double bn[NF], cjn[NJ];
#pragma unroll
                        for (int f = 0; f < NF; ++f) bn[f] = wn[f * 64];
#pragma unroll
                        for (int J = 0; J < NJ; ++J) cjn[J] = ncv[m * Dp + 16 * J + lr];
                        double qn[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
                        for (int J = 0; J < NJ; ++J) {
                            v4d acc = (v4d){cjn[J], cjn[J], cjn[J], cjn[J]};
#pragma unroll
                            for (int kk = 0; kk < 4 * (J + 1); ++kk)
                                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xf[kk], bn[2 * J * (J + 1) + kk], acc, 0, 0, 0);
#pragma unroll
                            for (int r = 0; r < 4; ++r) qn[r] = fma(acc[r], acc[r], qn[r]);
                        }
                        n_mfma += NF;
                        const double v = home_row_sum4(qn, lane);
                        if (lr < 4) sideQn[m * 64 + r0 + lk + 4 * lr] = v;
                    }
                }
                }
                HP(3)
            }
        } else {
            // ---- not one home's block (the unassigned bin; any block, should the order ever come unpadded): its live rows
            // go on the residual list below, where every label they cannot exclude is scored exactly (kernels_prune.hip).
            // The slots in flight hold this block's first rows: refilled with the next block's.
#pragma unroll
            for (int t = 0; t < NS - 1; ++t) HOME_ISSUE(xt[t], xo_next, t)
            HP(3)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // ---- the scalar tail: lane = row
        const bool live = imine >= 0;
        bool easy = false;
        if (SAFE) {
            // ---- proof pass of a safe-stay window (kernels_safe.hip): does the visit stay under the frozen state AND
            // every state the window's budget allows?  Nothing is drawn here; the rows left unproven are walked in order
            // by the frozen-factor resolver.
            bool safe = false, resid = false;
#ifdef BGMM_SAFE_DEBUG
            int why = live ? 1 : 0;
#endif
            if (live && hmine >= 0 && one_home) {
                const double q_t = sideQ[lane], rho2_t = sideRho[lane];
                const double lb0 = hsr[4], hv1m = hsr[5], okf = hsr[6], ik0 = hsr[3], finv_a = hsc[12];
                const double *__restrict__ gg = d.rtab + (long long)(d.nslots - 1) * 8;
                const double chi = (q_t + ik0) * gg[1];                 // c_0(x, x) e^cap
                if (okf > 0.5 && q_t >= 0.0 && chi < 1.0) {
                    const double lb = lb0 + hv1m * log(1.0 - chi);      // the home's score, from below
                    const double rad = sqrt(rho2_t * (1.0 + 1e-9)) * (1.0 + 1e-9);
                    const double jf = rad * finv_a;
                    double bound = INFINITY;
                    if (jf < 62.0) bound = hft[(int)jf + 1];
                    // everything but the home, relative to it: the other labels (together below `bound`), the new table
                    // (exact); a component opened inside the window ends it (kernels_gram.hip)
                    const double R = (exp(bound - lb) + exp(rcur.mlb0 - lb)) * (1.0 + 1e-6);
                    const double u_cur = u_in_rec ? rcur.u : d.u[win_base + wrow_cur];
                    safe = R < 0.25 && u_cur >= R + 1e-12 && u_cur <= 1.0 - R - 1e-12;
                    // the table's triangle bound may be what fails: such a visit goes on the residual list, where the
                    // components it cannot exclude are scored exactly (score_mfma_prune_kernel, safe_choice_kernel)
                    resid = !safe && u_cur >= 1e-9 && u_cur <= 1.0 - 1e-9 && exp(rcur.mlb0 - lb) < 0.25;
#ifdef BGMM_SAFE_DEBUG
                    why = safe ? 7 : (!(jf < 62.0) ? 3 : (exp(bound - lb) >= 0.125 ? 4 : (exp(rcur.mlb0 - lb) >= 0.125 ? 5 : 6)));
                } else {
                    why = okf > 0.5 ? 2 : 1;
#endif
                }
            }
#ifdef BGMM_SAFE_DEBUG
            for (int k = 1; k <= 7; ++k) {
                const unsigned long long mk = __ballot(why == k);
                if (lane == 0 && mk) atomicAdd((unsigned long long *)&c->prof[k], (unsigned long long)__popcll(mk));
            }
#endif
            if (live) d.cert[wrow_cur] = safe ? 1 : 0;
            easy = !resid;
        } else if (live && hmine >= 0 && one_home) {
            const double q_t = sideQ[lane], rho2_t = sideRho[lane];
            const int a = rcur.home_label;
            LDS_AS const int *hi = (LDS_AS const int *)(hsc + 13);
            const int nh = hi[0], ver = hi[1];
            const double a1 = hsc[6], coef1 = hsc[5], half_vd1 = hsc[4], base1 = hsc[9] + hsc[3], finv_a = hsc[12];
            if (keep_caches) {
                // per-point cache for certify_kernel (bgmm_device.h: PCache)
                PCache pc;
                pc.tag = ((long long)hmine << 32) | (unsigned int)ver;
                pc.qhome = q_t; pc.rho2 = rho2_t; pc.pad = 0.0;
                d.pcache[imine] = pc;
            }
            // ---- a chain at rest (D >= 48: no neighbour pass): every row of the wavefront stays by a wide margin, and that can be
            // seen without the two logarithms, the division and the square root of the exact tail.  With t_ub >= coef1 q / den
            // (hardware reciprocal, rounded up): vh >= vh_lb = base1 - half_vd1 t_ub (log(1 + t) <= t; -log(den) / 2 >= 0), the
            // radius from above by the hardware square root rounded up (the table grows with the radius).  A row with
            // vnew < vh_lb - 37.5 and every other component below vh_lb - margin is exactly a row the exact tail lets stay
            // without its uniform (dv < -37 there): the same decision, nothing else is left behind when the caches are off.
            // Round 6, D = 16 / 32 too (SQ counters: ~1 600 VALU instructions per 64 rows there, the software logarithms and
            // exponentials of up to six candidates per row, the matrix pipe 26 % busy): where the first table does not exclude
            // everybody, the neighbours' EXACT forms are at hand (sideQn) and bound their scores from above without a
            // logarithm -- log(1 + t) >= t - t^2 / 2 for small t (their components are large: t = q / v ~ 0.01, the bound is
            // tight to 1e-6 nats), the frexp chord beyond -- and the second table excludes the rest.
            bool all_fast = false;
            if (!keep_caches && nh >= 2) {
                const double den = 1.0 - a1 * q_t;
                const double t_ub = coef1 * q_t * __builtin_amdgcn_rcp(den) * (1.0 + 1e-6);
                const double vh_lb = base1 - half_vd1 * t_ub;
                const double jf_ub = __builtin_amdgcn_sqrt(rho2_t) * (1.0 + 1e-6) * finv_a;
                // (the new table: below e^-37.5 of the home it does not exist for the draw; up to e^-20 = 2.1e-9 of it, it takes
                //  the visit only for a uniform beyond 1 - 2.1e-9 -- the reference's scan, utils.py:15-20, subtracts the home's
                //  probability >= 1 - 2.1e-9 from u minus at most 3e-17: negative for every u <= 1 - 1e-8.  At D = 16 the prior
                //  predictive sits 24 - 45 nats below a point's own component: 70 % pass the first test, all the second.)
                bool fast = den > 0.5 && q_t >= 0.0 && jf_ub < 62.0 &&
                            (rcur.mlb0 < vh_lb - 37.5 || (u_in_rec && rcur.mlb0 < vh_lb - 20.0 && rcur.u <= 1.0 - 1e-8));
                const int jt = jf_ub < 62.0 ? (int)jf_ub + 1 : 63;
                bool others = hft[jt] < vh_lb - margin;
                if (NB > 0 && !others) {
                    others = hft2[jt] < vh_lb - margin;
                    const int nn = ncand[0];
#pragma unroll
                    for (int m = 0; m < NB; ++m) {
                        if (m >= nn) continue;
                        double t = sideQn[m * 64 + lane] * nsc[m * 8 + 2];
                        t = t > 0.0 ? t : 0.0;
                        const double lo = t < 0.25 ? t * (1.0 - 0.5 * t) : home_log1p_lower(t);
                        others = others && (nsc[m * 8] - nsc[m * 8 + 1] * lo < vh_lb - margin);
                    }
                }
                fast = fast && others;
                all_fast = __ballot(!fast) == 0ull;
#ifdef BGMM_HOME_PROF
                {   // (why a wavefront takes the exact tail: rows failing the new-table test / the tables + neighbours / the rest)
                    const bool f_new = rcur.mlb0 < vh_lb - 37.5 || (u_in_rec && rcur.mlb0 < vh_lb - 20.0 && rcur.u <= 1.0 - 1e-8);
                    const unsigned long long m0 = __ballot(true), m1 = __ballot(!f_new), m2 = __ballot(!others), m3 = __ballot(!fast);
                    if (lane == __ffsll((long long)m0) - 1) {
                        atomicAdd((unsigned long long *)&c->prof[8], 1ull);
                        atomicAdd((unsigned long long *)&c->prof[9], all_fast ? 1ull : 0ull);
                        atomicAdd((unsigned long long *)&c->prof[10], (unsigned long long)__popcll(m1));
                        atomicAdd((unsigned long long *)&c->prof[11], (unsigned long long)__popcll(m2));
                        atomicAdd((unsigned long long *)&c->prof[12], (unsigned long long)__popcll(m3));
                        atomicAdd((unsigned long long *)&c->prof[13], (unsigned long long)__popcll(m0));
                    }
                }
#endif
            }
            if (all_fast) {
                easy = true;
            } else if (nh >= 2) {
                // the visited point removed from its own component (slot_math.h: home form)
                const double den = 1.0 - a1 * q_t;
                const double vh = base1 - 0.5 * fm_log(den) - half_vd1 * fm_log(1.0 + fm_div(coef1 * q_t, den));
                const double vnew = rcur.mlb0;
                const double mx = fmax(vh, vnew);
                const double rad = sqrt(rho2_t * (1.0 + 1e-9)) * (1.0 + 1e-9);
                const double jf = rad * finv_a;
                double bound = INFINITY;
                if (jf < 62.0) bound = hft[(int)jf + 1];
                if (NB > 0) {
                    // ---- D = 16 / 32: ONE draw for every row (a wavefront executes both sides of a branch its lanes split
                    // over, and here nearly every wavefront has rows of both kinds): the candidates are the home, the new
                    // table and -- for a row the first table cannot settle -- the home's neighbours, scored exactly; everyone
                    // else is excluded by the first table (no neighbours needed) or by the second.  The reference's
                    // arithmetic over the candidates that are left (crpmm.py:75, utils.py:15-20): weights exp(v - max) in
                    // label order, p = weight / total, u -= p.
                    const bool first_ok = den > 0.0 && bound < mx - margin;
                    const int nn = ncand[0];
                    double vn[NB > 0 ? NB : 1];
                    double mx2 = mx;
#pragma unroll
                    for (int m = 0; m < NB; ++m) {
                        vn[m] = -INFINITY;
                        if (m < nn && !first_ok) {
                            const double qm = sideQn[m * 64 + lane];
                            vn[m] = nsc[m * 8] - nsc[m * 8 + 1] * fm_log(1.0 + qm * nsc[m * 8 + 2]);
                            mx2 = fmax(mx2, vn[m]);
                        }
                    }
                    const double b2 = jf < 62.0 ? hft2[(int)jf + 1] : INFINITY;
                    if (den > 0.0 && (first_ok || b2 < mx2 - margin)) {
                        easy = true;
                        const long long p = win_base + wrow_cur;
                        double tot = 0.0, toth = 0.0, ec[NB + 1];
#pragma unroll
                        for (int ci = 0; ci < NB + 1; ++ci) {
                            ec[ci] = 0.0;
                            if (ci > nn) continue;
                            const int who = ncand[1 + ci];
                            double v = vh;
#pragma unroll
                            for (int m = 0; m < NB; ++m) v = who == m ? vn[m] : v;
                            const double e = v == -INFINITY ? 0.0 : fm_exp(v - mx2);      // (a neighbour left out weighs nothing)
                            ec[ci] = e;
                            tot += e;
                            if (who >= 0) toth += e;
                        }
                        const double en = fm_exp(vnew - mx2);
                        tot += en; toth += en;
                        const double inv_tot = fm_div(1.0, tot);
                        // (the visit's uniform is fetched only when the draw depends on it -- a scattered 8-byte read per
                        //  visit otherwise: when everything but the home weighs less than 1e-17 of the total, the labels in
                        //  front of the home subtract less than that from a uniform that is at least 2^-53 and the home's own
                        //  probability is 1 to rounding: the visit stays)
                        const bool need_u = !(toth * inv_tot < 1e-17);
                        double uu = u_in_rec ? rcur.u : (need_u ? d.u[p] : 0.5);
                        int pick = K;
#pragma unroll
                        for (int ci = 0; ci < NB + 1; ++ci) {
                            if (ci > nn || pick != K) continue;
                            const int who = ncand[1 + ci];
                            int lab = a;
#pragma unroll
                            for (int m = 0; m < NB; ++m)
                                if (who == m) lab = ((LDS_AS const int *)(nsc + m * 8 + 3))[0];
                            uu -= ec[ci] * inv_tot;
                            if (uu < 0.0) pick = lab;
                        }
                        if (pick != a) d.choice[wrow_cur] = pick;
                        if (keep_caches) {
                            PCacheExact pe;
                            pe.epoch = epoch;
                            const double bb = first_ok ? bound : b2;
                            const double rest = bb < mx2 - kHomeFar ? 1.8048513878454153e-35 : fm_exp(bb - mx2);
                            pe.log_alt = mx2 - vh + fm_log(toth + (double)K * rest);
                            d.pcache2[imine] = pe;
                        }
                        if (pick != a) atomicMin(&c->first_mover, (unsigned long long)p);
                    }
                } else
                if (den > 0.0 && bound < mx - margin) {
                    // two candidates: prob = exp(lp - logsumexp), u -= prob in label order (crpmm.py:75, utils.py:15-20).
                    // With eh = exp(vh - mx), en = exp(vnew - mx) one of the two is exp(0) = 1; and when the new table
                    // lies more than 37 nats below the home, 1 + en rounds to 1, the log-sum-exp to vh, the home's
                    // probability to exp(0) = 1 and u - 1 is negative for every uniform: the visit stays.
                    easy = true;
                    const double dv = vnew - vh;                       // (mx = vh iff dv <= 0)
                    int pick = a;
                    double en = 0.0;
                    const long long p = win_base + wrow_cur;
                    if (!(dv < -37.0)) {
                        // (the visit's uniform is fetched only here: a scattered 8-byte read per visit otherwise, for a
                        //  draw whose outcome does not depend on it)
                        const double u_cur = u_in_rec ? rcur.u : d.u[p];
                        const double eo = fm_exp(-fabs(dv));
                        const double eh = dv <= 0.0 ? 1.0 : eo;
                        en = dv <= 0.0 ? eo : 1.0;
                        const double lse = fm_log(eh + en) + mx;
                        pick = u_cur - fm_exp(vh - lse) < 0.0 ? a : K;
                    }
                    // (apply_kernel reads the drawn label of the window's first mover only: a stay leaves nothing behind)
                    if (pick != a) d.choice[wrow_cur] = pick;
                    if (keep_caches) {
                        if (dv < -37.0) en = fm_exp(dv);
                        PCacheExact pe;
                        pe.epoch = epoch;
                        // log of the alternatives' total weight relative to the home's: the new table exactly, every other
                        // label below its table bound (e^-80 of the best when the bound is that far down)
                        const double rest = bound < mx - kHomeFar ? 1.8048513878454153e-35 : fm_exp(bound - mx);
                        pe.log_alt = mx - vh + fm_log(en + (double)K * rest);
                        d.pcache2[imine] = pe;
                    }
                    if (pick != a) atomicMin(&c->first_mover, (unsigned long long)p);
                }
            }
        }
        // ---- what the table bound could not decide goes on the residual list, a contiguous run per wavefront
        const bool hard = live && !easy;
        const unsigned long long mhard = __ballot(hard);
        if (mhard) {
            const int fl = __ffsll((long long)mhard) - 1;
            int base = 0;
            if (lane == fl) base = atomicAdd(&c->n_resid, __popcll(mhard));
            base = __builtin_amdgcn_readlane(base, fl);
            if (hard) {
                const int pos = base + __popcll(mhard & ((1ull << lane) - 1ull));
                d.wrecR[pos] = rcur;
                d.wpermR[pos] = wrow_cur;
            }
        }
        HP(4)
        // the next block's row becomes the current one; its uniform and the record after it set off
        rcur = rnext; wrow_cur = wrow_next; hf_cur = hf_next; hl_cur = hl_next; ok_cur = ok_next;
        xo_cur = xo_next;
        xo_next = (ok_after && i_after >= 0 ? i_after : 0) * D;
        HP(5)
    }
#ifdef BGMM_HOME_PROF
    if (lane == 0) {
        for (int k = 0; k < 8; ++k) atomicAdd((unsigned long long *)&c->prof[k], (unsigned long long)pf[k]);
        atomicAdd((unsigned long long *)&c->prof[15], 1ull);
    }
#endif
#undef HOME_LOAD_REC
#undef HOME_ISSUE
    if (lane == 0) {
        atomicAdd(&d.pr_counts[blockIdx.x & 255], (unsigned long long)n_homes);
        atomicAdd(&d.pr_counts[512 + (blockIdx.x & 255)], (unsigned long long)n_mfma);
    }
}

// as many workgroups as are resident, unless the window is shorter than that
static int home_cu_count() {
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n;
    }
    return cus[dev];
}

template <int NJ, bool WHOLE, bool SAFE>
static void launch_home_ts(const Dev &d, long long max_rows, hipStream_t st) {
    const long long want = (max_rows + 255) / 256 + d.nslots + 1;     // (+ the pads of the evaluation order: at most one block per bin)
    const long long cap = (long long)home_waves_per_simd(NJ) * home_cu_count();      // (a workgroup = one wavefront per SIMD)
    const unsigned gx = (unsigned)(want < cap ? want : cap);
    constexpr int lds = home_lds_bytes(NJ * 16);
    static PerDeviceLds attr;
    attr.ensure((const void *)home_kernel<NJ, WHOLE, SAFE>, lds);
    hipLaunchKernelGGL((home_kernel<NJ, WHOLE, SAFE>), dim3(gx), dim3(256), lds, st, d);
}
template <int NJ, bool WHOLE>
static void launch_home_t(const Dev &d, long long max_rows, hipStream_t st) {
    if (d.safe_mode) launch_home_ts<NJ, WHOLE, true>(d, max_rows, st);
    else launch_home_ts<NJ, WHOLE, false>(d, max_rows, st);
}

void launch_home(const Dev &d, long long max_rows, hipStream_t st) {
    if (max_rows <= 0) return;
    switch (d.Dp / 16) {
        case 1: if (d.D == d.Dp) launch_home_t<1, true>(d, max_rows, st); else launch_home_t<1, false>(d, max_rows, st); return;
        case 2: if (d.D == d.Dp) launch_home_t<2, true>(d, max_rows, st); else launch_home_t<2, false>(d, max_rows, st); return;
        case 3: if (d.D == d.Dp) launch_home_t<3, true>(d, max_rows, st); else launch_home_t<3, false>(d, max_rows, st); return;
        case 4: if (d.D == d.Dp) launch_home_t<4, true>(d, max_rows, st); else launch_home_t<4, false>(d, max_rows, st); return;
        case 5: if (d.D == d.Dp) launch_home_t<5, true>(d, max_rows, st); else launch_home_t<5, false>(d, max_rows, st); return;
        case 6: if (d.D == d.Dp) launch_home_t<6, true>(d, max_rows, st); else launch_home_t<6, false>(d, max_rows, st); return;
        case 7: if (d.D == d.Dp) launch_home_t<7, true>(d, max_rows, st); else launch_home_t<7, false>(d, max_rows, st); return;
        case 8: if (d.D == d.Dp) launch_home_t<8, true>(d, max_rows, st); else launch_home_t<8, false>(d, max_rows, st); return;
        default: return;
    }
}

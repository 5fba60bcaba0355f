// Frozen-factor windows: the mover-dense path (burn-in, overlapping clusters).
//
// In this regime nearly every visit changes the state, so the reference's loop (igmm/crpmm.py:57-88,
// igmm/pcrpmm.py:93-131) is one long dependent chain and what a move costs is the latency of that
// chain.  The factors of ALL components stay FROZEN for a window of kGramRows consecutive visits; every
// change a move makes to a component is carried as a rank-1 TERM of the augmented scatter matrix
//     A = [[S, m], [m', k_N]],   A += sigma * [x; 1][x; 1]'   (sigma = +1: x joins, -1: x leaves)
// -- no mean bookkeeping: with c(y, z) = [y; 1]' A^-1 [z; 1] the quadratic form of the Student-t
// predictive (gaussian/gaussian_components.py:240-244) is q(y) = c(y, y) - 1/k_N and det A = k_N det S_N.
// Sherman-Morrison on the bilinear form:
//     w_t(y)   = c_{t-1}(y, x_t) = c_0(y, x_t) - sum_{i<t} w_i(y) w_i(x_t) / D_i,   D_t = sigma_t + w_t(x_t)
//     c_t(y,y) = c_0(y, y) - sum_{i<=t} w_i(y)^2 / D_i
//     logdet S_N,t = logdet S_N,t-1 + log(k_{t-1} / k_t) + log(sigma_t D_t)
// Everything on the right is a SCALAR once the frozen cross forms c_0(y_r, y_r') = a(y_r).a(y_r') + 1/k_N,
// a(y) = Winv (y - mu), of the window's rows are known.  So:
//
//   gram_kernel      one workgroup per component (and one for the bare prior, from which new components
//                    grow): a(y_r) for the window's 64 rows with v_mfma_f64_16x16x4_f64 (the batched
//                    contraction of log_post_pred), then their 64 x 64 Gram matrix with the same
//                    instruction, the frozen log scores of the rows on the side;
//   gram_resolve     ONE workgroup walks the window's visits in order.  A draw (utils/utils.py:7-20) reads
//                    one weight per label; a move appends two terms, and the only D-independent work on
//                    the chain is O(R) per term -- no factor is loaded, updated or written here (the
//                    in-launch resolver of kernels_resolve.hip spends 2/3 of a move on exactly that);
//   gram_finish      (kernels_state.hip) one workgroup per touched component: replays the logged moves on
//                    (m, S) with the reference's roundings and rebuilds the factor from scratch.
//
// The reference's semantics restated: del_item / add_item / del_component of
// gaussian/gaussian_components.py:154-205, the seating weights of igmm/crpmm.py:68-75, a stay leaves
// the state untouched (igmm/crpmm.py:82-85).
#include "score_common.h"
#include "slot_math.h"
#include "wave_ops.h"
#include "fast_math.h"

#define LDS_AS __attribute__((address_space(3)))
typedef LDS_AS double *lds_f64;
typedef LDS_AS long long *lds_i64;
typedef LDS_AS int *lds_i32;

static constexpr int GR = kGramRows;

__device__ __forceinline__ long long safecol_pack(const SafeCol &e) {
    return (long long)(((unsigned long long)__float_as_uint(e.w)) | ((unsigned long long)(unsigned short)e.lo << 32) |
                       ((unsigned long long)(unsigned short)e.hi << 48));
}
__device__ __forceinline__ SafeCol safecol_unpack(long long v) {
    SafeCol e;
    e.w = __uint_as_float((unsigned)((unsigned long long)v & 0xffffffffull));
    e.lo = (short)(((unsigned long long)v >> 32) & 0xffffull);
    e.hi = (short)(((unsigned long long)v >> 48) & 0xffffull);
    e.wm = 0.0f; e.pad = 0;                                     // (the removals' account travels beside the packed word)
    return e;
}

// The window's rows: the next GR visits, or -- safe-stay windows (kernels_safe.hip) -- the listed ones: the visits of
// [job.pos, gl_end) that the proof pass could not prove to stay; every visit in between stays whatever the listed ones do.
__device__ __forceinline__ int gram_nrows(const Dev &d, const Ctrl *c) {
    if (d.safe_mode) return c->gl_n;
    const long long left = c->n_visits - c->job.pos;
    return left < GR ? (int)left : GR;
}
// (pipelined windows: the launch's window starts where the host predicted, whatever window the control block has open)
__device__ __forceinline__ long long gram_pos0(const Dev &d, const Ctrl *c) { return d.pipe ? d.pipe_pos : c->job.pos; }
__device__ __forceinline__ int gram_nrows_at(const Dev &d, const Ctrl *c, long long pos0) {
    if (d.safe_mode) return c->gl_n;
    const long long left = c->n_visits - pos0;
    return left < GR ? (left > 0 ? (int)left : 0) : GR;
}
__device__ __forceinline__ long long gram_pos(const Dev &d, long long pos0, int row) {
    return d.safe_mode ? d.glist[d.ctrl->gl_off + row] : pos0 + row;
}

// ------------------------------------------------------------------------------------------
// gram_kernel<NJ>: grid = columns, 256 threads (wave w: rows 16 w .. 16 w + 15).
// Fragment conventions as in kernels_score.hip.  LDS: Ys[64][Dp + 2] (row stride = 2 mod 32 doubles:
// the fragment reads Ys[16 t + (lane & 15)][4 kk + (lane >> 4)] are conflict free).
// ------------------------------------------------------------------------------------------
template <int NJ>
__device__ __forceinline__ void gram_body(const Dev &d, bool with_previous = false) {
    extern __shared__ __attribute__((aligned(16))) double Ys[];
    const Ctrl *c = d.ctrl;
    if (c->job.mode == MODE_DONE || c->error != 0 || (d.pipe && c->pipe_break)) return;
    const int K = c->job.K;
    if (K + kGramColSlack > d.gcols) return;               // (the resolver reports the stall)
    const int col = blockIdx.x;
    if (col > K) return;
    const int s = col < K ? d.perm[col] : d.K_max;
    const long long pos0 = gram_pos0(d, c);
    const int nrows = gram_nrows_at(d, c, pos0);
    if (nrows <= 0) return;
    constexpr int Dp = 16 * NJ, LD = Dp + 2, NF = 2 * NJ * (NJ + 1), PF = pick_pf(NF);
    const int D = d.D;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lr = lane & 15, lk = lane >> 4;

    // A fragments of this wave's 16 rows
    double xf[NJ * 4];
    {
        const int row = w * 16 + lr;
        const bool live = row < nrows;
        const long long p = live ? gram_pos(d, pos0, row) : 0;
        const long long i = live ? (d.order ? d.order[p] : p) : 0;
        const double *__restrict__ xrow = d.X + i * D;
#pragma unroll
        for (int kk = 0; kk < NJ * 4; ++kk) {
            const int l = 4 * kk + lk;
            xf[kk] = (live && l < D) ? xrow[l] : 0.0;
        }
    }
    const double *__restrict__ wf = d.Wfrag + (long long)s * NF * 64 + lane;
    double ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = wf[i * 64];
    double cj[NJ];
#pragma unroll
    for (int J = 0; J < NJ; ++J) cj[J] = d.cvec[(long long)s * d.Dp + 16 * J + lr];

    double qp[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int J = 0; J < NJ; ++J) {
        v4d acc = (v4d){cj[J], cj[J], cj[J], cj[J]};
#pragma unroll
        for (int kk = 0; kk < 4 * (J + 1); ++kk) {
            const int f = 2 * J * (J + 1) + kk;
            const double b = ring[f % PF];
            if (f + PF < NF) ring[f % PF] = wf[(f + PF) * 64];
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xf[kk], b, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            qp[r] = fma(acc[r], acc[r], qp[r]);
            Ys[(w * 16 + lk + 4 * r) * LD + 16 * J + lr] = acc[r];     // y_row[16 J + lr]
        }
    }
    // q0 and the frozen log score of (row, this column)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const double v = row16_sum(qp[r]);
        const int row = w * 16 + lk + 4 * r;
        if (lr == r && row < nrows) {
            d.gq0[(long long)col * GR + row] = v;
            if (col < K) {
                const long long p = gram_pos(d, pos0, row);
                const long long i = d.order ? d.order[p] : p;
                const bool own = d.z[i] == s && d.n[s] >= 2;
                d.glp0[(long long)row * d.gcols + col] = slot_score_exact(d.sc[s], v, own);
            }
        }
    }
    // what an update of this column needs besides its loads, for every count two terms can take it to
    if (threadIdx.x < 5) {
        const int n0 = col < K ? d.n[s] : 0, nn = n0 + (int)threadIdx.x - 2;
        const double kN0 = d.k0 + (double)n0, logdet0 = d.sc[s].logdetC;
        double *g = d.gcc + ((long long)col * 5 + threadIdx.x) * 8;
        g[0] = 1.0 / kN0; g[6] = logdet0; g[5] = -1.0;
        if (nn >= 1) {
            const SlotTab tab = load_slot_tab(d, nn);
            const double kN = d.k0 + (double)nn;
            const long long v = d.v0 + nn - d.D + 1;
            g[1] = 1.0 / kN;
            g[2] = kN / (kN + 1.0);
            g[3] = 0.5 * (double)(v + d.D);
            g[4] = tab.seat + (tab.g - 0.5 * ((double)d.D * tab.lc + logdet0)) - 0.5 * log(kN0 / kN);
            g[5] = (double)nn;
            g[7] = tab.seat1 + tab.g1 - 0.5 * (double)d.D * tab.lc1;      // (home form: the visited point removed)
        }
    }
    __syncthreads();
    // Gram tiles (ti, tj), tj >= ti: G[16 ti + i][16 tj + j] = y_{16 ti + i} . y_{16 tj + j}
    double *__restrict__ Cc = d.gC + (long long)col * GR * GR;
    for (int t = w; t < 10; t += 4) {
        int ti = 0, rem = t;
        while (rem >= 4 - ti) { rem -= 4 - ti; ++ti; }
        const int tj = ti + rem;
        if (ti * 16 >= nrows) continue;
        const double *__restrict__ ya = Ys + (ti * 16 + lr) * LD + lk;
        const double *__restrict__ yb = Ys + (tj * 16 + lr) * LD + lk;
        v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
        for (int kk = 0; kk < Dp / 4; ++kk)
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[4 * kk], yb[4 * kk], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) Cc[(ti * 16 + lk + 4 * r) * GR + tj * 16 + lr] = acc[r];
    }
    // ---- pipelined windows: the cross forms between this window's rows and the rows of the window BEFORE it (whose moves
    // are not in the factor yet: gram_carry_kernel applies them as terms) -- a(x) of the old rows into the second half of
    // the tile, then the full 64 x 64 block new x old
    if (d.pipe && with_previous && pos0 >= GR) {
        double *__restrict__ Ys2 = Ys + GR * LD;
        {
            const int row = w * 16 + lr;
            const long long p = pos0 - GR + row;
            const long long i = d.order ? d.order[p] : p;
            const double *__restrict__ xrow = d.X + i * D;
            double xo[NJ * 4];
#pragma unroll
            for (int kk = 0; kk < NJ * 4; ++kk) {
                const int l = 4 * kk + lk;
                xo[kk] = l < D ? xrow[l] : 0.0;
            }
            double ring2[PF];
#pragma unroll
            for (int i2 = 0; i2 < PF; ++i2) ring2[i2] = wf[i2 * 64];
#pragma unroll
            for (int J = 0; J < NJ; ++J) {
                v4d acc = (v4d){cj[J], cj[J], cj[J], cj[J]};
#pragma unroll
                for (int kk = 0; kk < 4 * (J + 1); ++kk) {
                    const int f = 2 * J * (J + 1) + kk;
                    const double b = ring2[f % PF];
                    if (f + PF < NF) ring2[f % PF] = wf[(f + PF) * 64];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xo[kk], b, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) Ys2[(w * 16 + lk + 4 * r) * LD + 16 * J + lr] = acc[r];
            }
        }
        __syncthreads();
        double *__restrict__ Xc = d.gX + (long long)col * GR * GR;
        for (int t = w; t < 16; t += 4) {
            const int ti = t >> 2, tj = t & 3;
            if (ti * 16 >= nrows) continue;
            const double *__restrict__ ya = Ys + (ti * 16 + lr) * LD + lk;
            const double *__restrict__ yb = Ys2 + (tj * 16 + lr) * LD + lk;
            v4d acc = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll 8
            for (int kk = 0; kk < Dp / 4; ++kk)
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ya[4 * kk], yb[4 * kk], acc, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) Xc[(ti * 16 + lk + 4 * r) * GR + tj * 16 + lr] = acc[r];
        }
    }
}
template <int NJ>
__global__ __launch_bounds__(256) void gram_kernel(Dev d) { gram_body<NJ>(d); }
template <int NJ>
__global__ __launch_bounds__(256) void gram_cross_kernel(Dev d, int with_previous) { gram_body<NJ>(d, with_previous != 0); }
// (several chains in one launch: workgroup (x, c) works for chain group[c] -- bgmm_group_sweep_staged)
template <int NJ>
__global__ __launch_bounds__(256) void gram_group_kernel(const Dev *__restrict__ group) {
    const Dev d = group[blockIdx.y];               // (a private copy: nothing the body writes can alias it)
    gram_body<NJ>(d);
}
// Window k of a PIPELINED batch of several chains (api_group.hip: gram_group_pipe_launch): v0 / v1 hold every chain's view with
// the window buffers of set 0 / 1 (xp_in = the other set's xp_out, pipe_pos = the visit the chain's batch starts at); window
// k works in set k & 1, 64 k visits further on -- what gram_pipe_batch (api_sweep.hip) patches into the view it passes by value.
__device__ __forceinline__ Dev gram_pgroup_view(const Dev *__restrict__ v0, const Dev *__restrict__ v1, int k, int chain) {
    Dev d = ((k & 1) ? v1 : v0)[chain];
    d.pipe = k == 0 ? 2 : 1;
    d.pipe_pos += (long long)kGramRows * k;
    return d;
}
template <int NJ>
__global__ __launch_bounds__(256) void gram_cross_pgroup_kernel(const Dev *__restrict__ v0, const Dev *__restrict__ v1, int k, int with_previous) {
    const Dev d = gram_pgroup_view(v0, v1, k, (int)blockIdx.y);
    gram_body<NJ>(d, with_previous != 0);
}


// ------------------------------------------------------------------------------------------
// gram_weights_kernel: one workgroup per window row.  Reference point M_r = max of the row's frozen log
// scores (a singleton home is no candidate), frozen weights e0[r][c] = exp(lp0 - M_r), the new table's
// weight (igmm/crpmm.py:74).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void gram_weights_body(const Dev &d) {
    __shared__ double red[4];
    const Ctrl *c = d.ctrl;
    if (c->job.mode == MODE_DONE || c->error != 0 || (d.pipe && c->pipe_break)) return;
    const int K = c->job.K;
    if (K + kGramColSlack > d.gcols) return;
    const long long pos0 = gram_pos0(d, c);
    const int nrows = gram_nrows_at(d, c, pos0);
    const int r = blockIdx.x;
    if (r >= nrows) return;
    const long long p = gram_pos(d, pos0, r);
    const long long i = d.order ? d.order[p] : p;
    const int h = d.z[i];
    const int excl = (h >= 0 && d.n[h] == 1) ? d.label_of_slot[h] : -1;
    const double lp_new = d.log_alpha + d.log_prior[i];
    const long long gld = d.gcols;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double lpv[2];
    double mx = lp_new;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int j = threadIdx.x + 256 * t;
        lpv[t] = j < K ? d.glp0[r * gld + j] : -INFINITY;
        if (j != excl) mx = fmax(mx, lpv[t]);
    }
    mx = wv_max(mx);
    if (lane == 0) red[w] = mx;
    __syncthreads();
    mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int j = threadIdx.x + 256 * t;
        if (j < K) d.ge0[r * gld + j] = exp(lpv[t] - mx);
    }
    if (threadIdx.x == 0) { d.gM[r] = mx; d.gM[GR + r] = exp(lp_new - mx); }
}
__global__ __launch_bounds__(256) void gram_weights_kernel(Dev d) { gram_weights_body(d); }
// (several chains in one launch: workgroup (x, c) works for chain group[c] -- bgmm_group_sweep_staged)
__global__ __launch_bounds__(256) void gram_weights_group_kernel(const Dev *__restrict__ group) {
    const Dev d = group[blockIdx.y];               // (a private copy: nothing the body writes can alias it)
    gram_weights_body(d);
}
__global__ __launch_bounds__(256) void gram_weights_pgroup_kernel(const Dev *__restrict__ v0, const Dev *__restrict__ v1, int k) {
    const Dev d = gram_pgroup_view(v0, v1, k, (int)blockIdx.y);
    gram_weights_body(d);
}


// ------------------------------------------------------------------------------------------
// gram_resolve_kernel: one workgroup, 512 threads.
//   wave 0      draws the visits in order.  What a draw needs is one weight per label: the frozen
//               exp(lp0 - M_r) (HBM, fetched a visit ahead) or, once a term has touched the column, its
//               LDS line.  The wave keeps its labels' columns and lines in registers, the rows' inputs
//               in lane r's registers; at a move lanes 0 / 1 do the bookkeeping of the reference's
//               del_item / add_item for the two touched columns side by side
//   waves 1, 2  one touched column each: the new term's w, D, the column's logdet and count
//               constants, its weights for all later rows (lane = row); wave 1 fetches the next
//               visit's home side ahead
//   waves 3..7  read the next visits' rows of the cross-form matrices (any column may be drawn) so
//               that the one dependent fetch of a move is served by this XCD's L2
// LDS plan: etT[T][64] weights of touched columns, wv[T][64] the terms' w vectors; per column / label /
// term bookkeeping.
// ------------------------------------------------------------------------------------------
// What wave 0 publishes about a move (one LDS round trip for the update waves).  mode 0: the usual move,
// the update waves read and write their column's bookkeeping themselves; mode 1: a component was
// deleted / opened, wave 0 has done the bookkeeping and filled upd[].
struct GramUpd {
    int col, sigma, term, base, slot, n_new, prev, eidx;
    double rcf;                    // prod |D_t|^-1/2 over the column's terms so far
};

struct GramShared {
    int active, nrows, K0, ncols, cprior, err, event, cur;
    int pub_mode, pub_hcol, pub_pcol, pub_has0, pub_term, upd_n, K, nmoves;
    int cut, pad_;                 // safe-stay windows: a column ran out of budget (1) / drifted too far or was opened (2):
                                   // the window ends behind this move
    int nterms_end, mapver_end;    // at the end of the walk: terms made / changes of the label map (pipelined windows)
    double cap;                    // the budget the robust tables of this window were built for
    double wsum2[2];               // sum of |log |D_t|| over the window's terms, and their number (per update wave)
    int wterms2[2];
    GramUpd upd[2];
    long long pos0, lik, last_mover;
    double ema_run;
    long long prof[16];
};

// LDS plan, all offsets compile-time (KC columns, T terms): byte offsets from the start of dynamic LDS
template <int KC, int T>
struct GramPlan {
    static constexpr unsigned oS = 0;
    static constexpr unsigned oEt = 512;                          // etT[T][64]: weights of a column's current term, per row
    static constexpr unsigned oWv = oEt + T * GR * 8;             // wv[T][64]: the terms' w vectors
    static constexpr unsigned oRowM = oWv + T * GR * 8;           // rowM[64]
    static constexpr unsigned oInvD = oRowM + GR * 8;             // termInvD[T]
    static constexpr unsigned oRcf = oInvD + T * 8;               // colRCF[KC]
    static constexpr unsigned oColW = oRcf + KC * 8;              // colE[KC] (SafeCol): budget used since the proof pass, counts allowed
    static constexpr unsigned oMvI = oColW + KC * 8;              // move log: data index [64]
    static constexpr unsigned oRowHome = oMvI + GR * 8;           // [64]
    static constexpr unsigned oRowHcol = oRowHome + GR * 4;       // [64]
    static constexpr unsigned oMv = oRowHcol + GR * 4;            // move log: sub slot, add slot, init flag [3][64]
    static constexpr unsigned oTermPrev = oMv + 3 * GR * 4;       // [T]
    static constexpr unsigned oColSlot = oTermPrev + T * 4;       // per column [KC] ints from here on
    static constexpr unsigned oColN = oColSlot + KC * 4;
    static constexpr unsigned oColN0 = oColN + KC * 4;
    static constexpr unsigned oColBase = oColN0 + KC * 4;
    static constexpr unsigned oColLab = oColBase + KC * 4;
    static constexpr unsigned oColLast = oColLab + KC * 4;
    static constexpr unsigned oLabCol = oColLast + KC * 4;
    static constexpr unsigned oPermL = oLabCol + KC * 4;
    static constexpr unsigned oColWm = oPermL + KC * 4;           // colWm[KC] floats: the removals' account (SafeCol::wm)
    static constexpr unsigned oTermRow = oColWm + KC * 4;         // per term [T] ints: the row that made it, its sign, its column
    static constexpr unsigned oTermSig = oTermRow + T * 4;        // (what the next window's carry needs: pipelined windows)
    static constexpr unsigned oTermCol = oTermSig + T * 4;
    static constexpr unsigned bytes = oTermCol + T * 4;
};

// the two plans the host chooses between (columns, terms)
static constexpr int kPlanA_KC = 448, kPlanA_T = 128;      // up to ~250 labels
static constexpr int kPlanB_KC = 1024, kPlanB_T = 64;      // up to ~440 labels
static_assert(GramPlan<kPlanA_KC, kPlanA_T>::bytes <= 160 * 1024, "plan A exceeds the LDS of a CU");
static_assert(GramPlan<kPlanB_KC, kPlanB_T>::bytes <= 160 * 1024, "plan B exceeds the LDS of a CU");

bool gram_plan_for(int K, int *gcols, int *terms, int *lds) {
    if (K + kPlanA_T / 2 + 2 + 128 <= kPlanA_KC) { *gcols = kPlanA_KC; *terms = kPlanA_T; *lds = (int)GramPlan<kPlanA_KC, kPlanA_T>::bytes; return true; }
    if (K + kPlanB_T / 2 + 2 + 32 <= 512) { *gcols = kPlanB_KC; *terms = kPlanB_T; *lds = (int)GramPlan<kPlanB_KC, kPlanB_T>::bytes; return true; }
    return false;
}

__device__ __forceinline__ void gram_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

enum { GEV_DONE = 0, GEV_MOVE = 1, GEV_CUT = 2 };

// base[idx] with a 32-bit element index (the buffers of a window hold far fewer than 2^29 doubles):
// one scalar base + 32-bit vector offset instead of 64-bit address arithmetic on the chain
__device__ __forceinline__ double gram_ld(const double *base, unsigned idx) {
    return *(const double *)((const char *)base + (size_t)(idx << 3));
}

#define GRT 256

#ifdef BGMM_PROFILE
#define GPROF(i) do { if (lane == 0) { tk2 = clock64(); S.prof[i] += tk2 - tk; tk = tk2; } } while (0)
#else
#define GPROF(i) do { } while (0)
#endif

// LPL: labels per lane of the draw wave (label j = lane * LPL + t): 64 LPL - 1 bounds the labels a window can
// reach.  KC / T: the LDS plan (columns, terms; a term's index is also its line of weights).
template <int LPL, int KC, int T>
__device__ __forceinline__ void gram_resolve_body(const Dev &d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    static_assert(sizeof(GramShared) <= 512, "GramShared has 512 bytes of LDS");
    using P = GramPlan<KC, T>;
    LDS_AS unsigned char *const lb = (LDS_AS unsigned char *)lds_raw;
    LDS_AS GramShared &S = *(LDS_AS GramShared *)lb;
    const lds_f64 etT = (lds_f64)(lb + P::oEt), wvv = (lds_f64)(lb + P::oWv), rowM = (lds_f64)(lb + P::oRowM),
                  termInvD = (lds_f64)(lb + P::oInvD), colRCF = (lds_f64)(lb + P::oRcf);
    const lds_i64 colE = (lds_i64)(lb + P::oColW);             // SafeCol, packed (safecol_pack / safecol_unpack)
    const lds_i64 mvI = (lds_i64)(lb + P::oMvI);
    const lds_i32 rowhome = (lds_i32)(lb + P::oRowHome), rowhcol = (lds_i32)(lb + P::oRowHcol),
                  mvSub = (lds_i32)(lb + P::oMv), mvAdd = mvSub + GR, mvInit = mvAdd + GR,
                  termPrev = (lds_i32)(lb + P::oTermPrev), colSlot = (lds_i32)(lb + P::oColSlot),
                  colN = (lds_i32)(lb + P::oColN), colN0 = (lds_i32)(lb + P::oColN0), colBase = (lds_i32)(lb + P::oColBase),
                  colLab = (lds_i32)(lb + P::oColLab), colLast = (lds_i32)(lb + P::oColLast),
                  labCol = (lds_i32)(lb + P::oLabCol), permL = (lds_i32)(lb + P::oPermL);
    LDS_AS float *const colWm = (LDS_AS float *)(lb + P::oColWm);
    const lds_i32 termRow = (lds_i32)(lb + P::oTermRow), termSig = (lds_i32)(lb + P::oTermSig), termCol = (lds_i32)(lb + P::oTermCol);
    Ctrl *c = d.ctrl;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr unsigned gld = KC;
#ifdef BGMM_PROFILE
    long long tk = clock64(), tk2;
    const long long prof_c0 = clock64(), prof_w0 = wall_clock64();
#endif

    if (tid == 0) {
        S.active = 0;
        d.gfin[0] = 0;                          // (an idle step must not replay the last window's lists)
        d.gfin[1] = 0;
        d.gfin[2] = 0;
        if (d.pipe) ((GramXp *)d.xp_out)->hdr[2] = 0;
        const Job &j = c->job;
        if (d.pipe && j.mode != MODE_DONE && c->error == 0 && (c->pipe_break || j.pos != d.pipe_pos)) {
            c->pipe_break = 1;                  // (the window is not where the carried cross forms are: the batch stands still)
        } else if (j.mode != MODE_DONE && c->error == 0) {
            if (d.gcols != KC || j.K + kGramColSlack > KC || j.K + T / 2 + 2 > 64 * LPL) {
                c->gram_stall = 1;              // more columns than the plan / labels than the draw wave holds
                if (d.pipe) c->pipe_break = 1;
            } else {
                S.active = 1;
                S.pos0 = j.pos;
                S.nrows = gram_nrows(d, c);
                S.cut = 0;
                S.cap = c->safe_cap_built;
                S.wsum2[0] = 0.0; S.wsum2[1] = 0.0; S.wterms2[0] = 0; S.wterms2[1] = 0;
                S.K0 = j.K; S.K = j.K;
                S.cprior = j.K;
                S.ncols = j.K + 1;
                S.err = 0; S.cur = 0; S.event = GEV_DONE; S.nmoves = 0; S.upd_n = 0; S.pub_mode = 0;
                S.lik = 0;
                S.ema_run = c->ema_run; S.last_mover = c->last_mover;
                for (int k = 0; k < 16; ++k) S.prof[k] = 0;
            }
        }
    }
    __syncthreads();
    if (!S.active) return;
    const int nrows = S.nrows, K0 = S.K0, cprior = S.cprior;
    const long long pos0 = S.pos0;
    // rows: lane r of wave 0 keeps row r's inputs in registers; the update waves read home / home column / M_r from LDS
    int rw_home = -1, rw_hcol = -1;
    long long rw_i = 0, rw_p = 0;
    double rw_u = 0.0, rw_En = 0.0;
    if (wave == 0 && lane < nrows) {
        const long long p = gram_pos(d, pos0, lane);
        rw_p = p;
        rw_i = d.order ? d.order[p] : p;
        rw_home = d.z[rw_i];
        rw_hcol = rw_home >= 0 ? d.label_of_slot[rw_home] : -1;
        rw_u = d.u[p];
        rw_En = d.gM[GR + lane];
        rowhome[lane] = rw_home;
        rowhcol[lane] = rw_hcol;
        rowM[lane] = d.gM[lane];
    }
    for (int j = tid; j < KC; j += GRT) {
        const int s = j < d.nslots ? d.perm[j] : -1;
        permL[j] = s;
        labCol[j] = j;
        colLab[j] = j < K0 ? j : -1;
        colLast[j] = -1;
        colBase[j] = j <= K0 ? j : K0;
        colRCF[j] = 1.0;
        {
            SafeCol e; e.w = 0.0f; e.lo = 0; e.hi = 0;                      // (a column opened inside the window: no allowance)
            e.wm = 0.0f; e.pad = 0;
            if (d.safe_mode && j < K0 && s >= 0) e = d.ep_state[s];
            colE[j] = safecol_pack(e);
            colWm[j] = e.wm;
        }
        if (j < K0) {
            // (a carried window: the counts behind the window before -- Dev::n is being brought up to date beside this kernel)
            const int n = d.pipe == 1 ? ((const GramXp *)d.xp_in)->colN[j] : d.n[s];
            colSlot[j] = s;
            colN[j] = n;
            colN0[j] = n;
        } else {
            colSlot[j] = j == K0 ? d.K_max : -1;
            colN[j] = 0;
            colN0[j] = 0;
        }
    }
    __syncthreads();
    GPROF(0);

    int cur = 0;
    // ---- wave 0's registers ------------------------------------------------------------------
    int K = K0, nterms = 0, nmoves = 0, mapver = 0, ncols = K0 + 1;
    long long lik = 0, last_mover = S.last_mover;
    double ema_run = S.ema_run;
    int cls[LPL], tix[LPL], my_ver = -1;       // column / current term (= LDS line) of the lane's labels lane * LPL + t
    double pf[LPL];                            // frozen weights of visit pf_row, fetched a visit ahead under the label map pf_ver
    int pf_row = -1, pf_ver = -2;
    // the pre-drawn visit pr_row: cumulative weight through each of the lane's labels, with the columns
    // pr_x0 / pr_x1 (being updated by waves 1 / 2 meanwhile) left out; pm0 / pm1: 1.0 where the label
    // lies at or behind pr_x0 / pr_x1
    double pc[LPL], pm0[LPL], pm1[LPL], pr_tot = 0.0;
    int pr_row = -1, pr_ver = -2, pr_x0 = -1, pr_x1 = -1, pr_tix0 = 0, pr_tix1 = 0;
#pragma unroll
    for (int t = 0; t < LPL; ++t) { cls[t] = 0; tix[t] = -1; pf[t] = 0.0; pc[t] = 0.0; pm0[t] = 0.0; pm1[t] = 0.0; }
    // what a move leaves for the time the update waves work
    int hk = 0, hk_h = -1, hk_hcol = -1, hk_pcol = -1, hk_t0 = 0, hk_t1 = 0, hk_has0 = 0;
    double tch0 = 0.0, tch1 = 0.0, tch2 = 0.0;
    long long hk_i = 0;
    // ---- wave 1's registers: the home side of the next visit, fetched ahead ----------------------
    int hp_base = -1, hp_r = -1;
    double hp_crow = 0.0, hp_cd0 = 0.0, hp_cc = 0.0;

    // Everything of visit r that does not depend on the columns x0 / x1: the lane's cumulative weights.
    // Needs the label cache (cls, tix) and the frozen weights pf of visit r; fetches those of visit r + 1.
    auto pre_draw = [&](int r, int x0, int x1, int tx0, int tx1) {
        double ev_l[LPL];
#pragma unroll
        for (int t = 0; t < LPL; ++t) ev_l[t] = etT[(tix[t] < 0 ? 0 : tix[t]) * GR + r];
        const double En = wv_readlane(rw_En, r);
        const int lab0 = x0 >= 0 ? colLab[x0] : 0x7fffffff;
        const int lab1 = x1 >= 0 ? colLab[x1] : 0x7fffffff;
        double pfn[LPL];
        {
            const unsigned rn = r + 1 < nrows ? r + 1 : r;
#pragma unroll
            for (int t = 0; t < LPL; ++t) pfn[t] = gram_ld(d.ge0, rn * gld + (unsigned)cls[t]);
        }
        double run = 0.0;
#pragma unroll
        for (int t = 0; t < LPL; ++t) {
            const int j = lane * LPL + t;
            double v = j < K ? (tix[t] >= 0 ? ev_l[t] : pf[t]) : (j == K ? En : 0.0);
            v = (j < K && (cls[t] == x0 || cls[t] == x1)) ? 0.0 : v;
            run += v;
            pc[t] = run;
        }
        const double incl = wv_scan(run, lane);
        pr_tot = wv_readlane(incl, 63);
        const double off = incl - run;
#pragma unroll
        for (int t = 0; t < LPL; ++t) {
            const int j = lane * LPL + t;
            pc[t] += off; pf[t] = pfn[t];
            pm0[t] = j >= lab0 ? 1.0 : 0.0;
            pm1[t] = j >= lab1 ? 1.0 : 0.0;
        }
        pf_row = r + 1; pf_ver = mapver;
        pr_row = r; pr_ver = mapver; pr_x0 = x0; pr_x1 = x1; pr_tix0 = tx0; pr_tix1 = tx1;
    };

    for (;;) {
        hk = 0;
        if (wave == 0) {
            // ---- draws, until a visit does not stay ------------------------------------------
            int ev = GEV_DONE;
            for (; cur < nrows; ++cur) {
                const int r = cur;
                const int h = __builtin_amdgcn_readlane(rw_home, r), hcol = __builtin_amdgcn_readlane(rw_hcol, r);
                const double u = wv_readlane(rw_u, r);
                if (my_ver != mapver) {             // (rare) the label map changed (new / deleted component)
#pragma unroll
                    for (int t = 0; t < LPL; ++t) {
                        const int j = lane * LPL + t;
                        cls[t] = j < K ? labCol[j] : 0;
                        tix[t] = j < K ? colLast[cls[t]] : -1;
                    }
                    my_ver = mapver;
                }
                // (the count is read here, behind the barrier: the update waves have written it)
                const int nh = h >= 0 ? colN[hcol] : 0;
                const double a0 = pr_x0 >= 0 ? etT[pr_tix0 * GR + r] : 0.0;     // the two columns rewritten meanwhile
                const double a1 = pr_x1 >= 0 ? etT[pr_tix1 * GR + r] : 0.0;
                const bool home_live = nh >= 2, singleton = nh == 1;
                const int Lr = singleton ? K - 1 : K;                // labels after the removal
                int pick = Lr, pcol = -1;                             // fallback: the last entry (utils.py:20)
                bool bad_tot = false;
                if (!singleton) {
                    double b0 = a0, b1 = a1;
                    if (!(pr_row == r && pr_ver == mapver)) {
                        if (!(pf_row == r && pf_ver == mapver)) {        // (first visit, or the map changed)
#pragma unroll
                            for (int t = 0; t < LPL; ++t) pf[t] = gram_ld(d.ge0, (unsigned)r * gld + (unsigned)cls[t]);
                        }
                        pre_draw(r, -1, -1, 0, 0);
                        b0 = 0.0; b1 = 0.0;
                    }
                    const double tot = pr_tot + (b0 + b1);
                    bad_tot = !(tot > 1e-200 && tot < 1e200);            // M_r went stale: a fresh window
                    const double ut = u * tot;
                    int hit = 0x7fffffff, hitcol = -1;
#pragma unroll
                    for (int t = LPL - 1; t >= 0; --t) {
                        // (descending: the first label whose cumulative weight exceeds u wins)
                        const int j = lane * LPL + t;
                        const double ct = fma(pm1[t], b1, fma(pm0[t], b0, pc[t]));
                        const bool ok = j <= K && (ut - ct) < 0.0;
                        hit = ok ? j : hit;
                        hitcol = ok ? (j < K ? cls[t] : -1) : hitcol;
                    }
                    const unsigned long long mh = __ballot(hit != 0x7fffffff);
                    if (mh) {
                        const int fl = __ffsll((long long)mh) - 1;
                        pick = __builtin_amdgcn_readlane(hit, fl);
                        pcol = __builtin_amdgcn_readlane(hitcol, fl);
                    }
                } else {
                    // (rare) the home is a singleton: its label's place is taken by the last label (swap with last)
                    const int lab_h = colLab[hcol];
                    const double En = wv_readlane(rw_En, r);
                    double e[LPL];
                    int ecol[LPL];
#pragma unroll 1
                    for (int t = 0; t < LPL; ++t) {
                        const int j = lane * LPL + t;
                        double v = 0.0;
                        int cl = 0;
                        if (j < Lr) {
                            cl = labCol[j == lab_h ? K - 1 : j];
                            const int tx = colLast[cl];
                            v = tx >= 0 ? etT[tx * GR + r] : gram_ld(d.ge0, (unsigned)r * gld + (unsigned)cl);
                        } else if (j == Lr) {
                            v = En;
                        }
                        e[t] = v; ecol[t] = cl;
                    }
                    double lsum = 0.0;
#pragma unroll
                    for (int t = 0; t < LPL; ++t) lsum += e[t];
                    const double incl = wv_scan(lsum, lane);
                    const double tot = wv_readlane(incl, 63);
                    bad_tot = !(tot > 1e-200 && tot < 1e200);
                    const double ut = u * tot;
                    double ct = incl - lsum;
                    int hit = 0x7fffffff, hitcol = -1;
#pragma unroll
                    for (int t = 0; t < LPL; ++t) {
                        ct += e[t];
                        const int j = lane * LPL + t;
                        const bool ok = hit == 0x7fffffff && j <= Lr && (ut - ct) < 0.0;
                        hit = ok ? j : hit;
                        hitcol = ok ? (j < Lr ? ecol[t] : -1) : hitcol;
                    }
                    const unsigned long long mh = __ballot(hit != 0x7fffffff);
                    if (mh) {
                        const int fl = __ffsll((long long)mh) - 1;
                        pick = __builtin_amdgcn_readlane(hit, fl);
                        pcol = __builtin_amdgcn_readlane(hitcol, fl);
                    }
                }
                pr_row = -1;
                GPROF(1);
                if (bad_tot) { ev = GEV_CUT; break; }
                const bool stay = home_live && pick < Lr && pcol == hcol;
                if (stay) { lik += K; continue; }
                if (nterms + 2 > T) { ev = GEV_CUT; break; }
                ev = GEV_MOVE;
                if (!singleton && pick < K) {
                    // ---- the usual move: published in two stores; the update waves look after their columns ----
                    const int has0 = h >= 0 ? 1 : 0;
                    if (lane == 0) {
                        S.pub_mode = 0; S.pub_hcol = hcol; S.pub_pcol = pcol; S.pub_has0 = has0; S.pub_term = nterms;
                        S.upd_n = 1 + has0;
                    }
                    hk = 1; hk_h = h; hk_hcol = hcol; hk_pcol = pcol; hk_has0 = has0;
                    hk_t0 = nterms; hk_t1 = nterms + has0;
                    nterms += 1 + has0;
                    {
                        // The joined column's row of cross forms, its diagonal and its constants are what the update wave
                        // fetches on the chain -- ~1 000 cycles behind this point (barrier, bookkeeping reads) and from far
                        // away (written on other XCDs).  The compute unit's L1 serves all its wavefronts: the same lines are
                        // asked for HERE, the moment the label is known; nobody waits for them on this wavefront (the values
                        // are dropped behind the barrier).
                        const unsigned pb = (unsigned)(pcol < K0 ? pcol : K0);
                        const unsigned l0 = (unsigned)((lane >= r && lane < nrows) ? lane : r);
                        tch0 = gram_ld(d.gC, (pb * GR + (unsigned)r) * GR + l0);
                        tch1 = gram_ld(d.gq0, pb * GR + l0);
                        tch2 = gram_ld(d.gcc, pb * 40 + (unsigned)(lane < 40 ? lane : 0));
                    }
                } else {
                    // ---- (rare) a component is deleted and / or opened: lane 0, step by step ----
                    const long long i_mv = wv_readlane_i64(rw_i, r);
                    int Kn = K, Krem = K, nu = 0, sub_slot = -1, add_slot = -1, add_init = 0, nt = nterms, nc = ncols, err = 0;
                    if (lane == 0) {
                        if (h >= 0) {
                            const int n1 = nh - 1;
                            colN[hcol] = n1;
                            if (n1 > 0) {
                                sub_slot = h;
                                S.upd[nu].col = hcol; S.upd[nu].sigma = -1; S.upd[nu].slot = h; S.upd[nu].n_new = n1; ++nu;
                            } else {                    // swap-with-last delete of its label (gaussian_components.py:188-205)
                                const int lab = colLab[hcol], last = Kn - 1;
                                const int c_last = labCol[last], s_last = permL[last];
                                labCol[lab] = c_last; permL[lab] = s_last; colLab[c_last] = lab;
                                labCol[last] = hcol; permL[last] = h; colLab[hcol] = -1;
                                colSlot[hcol] = -1;       // retired: the slot may come back in a new column
                                d.perm[lab] = s_last; d.label_of_slot[s_last] = lab;
                                d.perm[last] = h; d.label_of_slot[h] = last;
                                d.n[h] = 0;
                                Kn = last;
                            }
                        }
                        Krem = Kn;                       // labels the draw chose among
                        int dcol = -1;
                        if (pick >= Kn) {                // a new component
                            if (Kn >= d.K_max || nc >= KC) {
                                err = -3;
                            } else {
                                const int t = permL[Kn];
                                dcol = nc++;
                                colSlot[dcol] = t; colN[dcol] = 0; colBase[dcol] = cprior;
                                colN0[dcol] = 0; colRCF[dcol] = 1.0; colE[dcol] = 0ll; colWm[dcol] = 0.0f;
                                colLast[dcol] = -1;
                                colLab[dcol] = Kn; labCol[Kn] = dcol;
                                d.label_of_slot[t] = Kn;
                                d.nupd[t] = 0;
                                add_init = 1;
                                Kn += 1;
                            }
                        } else {
                            dcol = pcol;
                        }
                        if (dcol >= 0) {
                            const int nn = colN[dcol] + 1;
                            colN[dcol] = nn;
                            add_slot = colSlot[dcol];
                            S.upd[nu].col = dcol; S.upd[nu].sigma = 1; S.upd[nu].slot = add_slot; S.upd[nu].n_new = nn; ++nu;
                        }
                        for (int k = 0; k < nu; ++k) {
                            const int cl = S.upd[k].col;
                            const int t = nt++;
                            const int prev = colLast[cl];
                            S.upd[k].term = t; S.upd[k].prev = prev;
                            S.upd[k].base = colBase[cl];
                            const int ee = S.upd[k].n_new - colN0[cl] + 2;
                            S.upd[k].eidx = (ee >= 0 && ee <= 4 && S.upd[k].n_new >= 1) ? ee : -1;
                            S.upd[k].rcf = colRCF[cl];
                            termPrev[t] = prev;
                            colLast[cl] = t;
                        }
                        S.upd_n = nu;
                        S.pub_mode = 1;
                        if (err) S.err = err;
                        mvI[nmoves] = i_mv; mvSub[nmoves] = sub_slot; mvAdd[nmoves] = add_slot; mvInit[nmoves] = add_init;
                    }
                    K = __builtin_amdgcn_readlane(Kn, 0);
                    nterms = __builtin_amdgcn_readlane(nt, 0);
                    ncols = __builtin_amdgcn_readlane(nc, 0);
                    nmoves += 1;
                    lik += __builtin_amdgcn_readlane(Krem, 0);
                    mapver += 1;                    // (labels and lines are re-read from LDS at the next draw)
                    const long long p = wv_readlane_i64(rw_p, r);
                    ema_run = ema_after_mover(ema_run, (double)(p - last_mover));
                    last_mover = p;
                }
                break;
            }
            if (lane == 0) { S.event = ev; S.cur = cur; }
            GPROF(2);
        }
        gram_lds_barrier();
        const int ev = S.event;
        if (ev != GEV_MOVE) break;
        if (S.err < 0) break;
        cur = S.cur;
#ifdef BGMM_PROFILE
        tk = clock64();
#endif
        if (wave == 0) {
            if (hk) {
                asm volatile("" :: "v"(tch0), "v"(tch1), "v"(tch2));      // (the touched lines: dropped)
                // ---- what the move leaves to do, while the update waves work ----------------------
#pragma unroll
                for (int t = 0; t < LPL; ++t) {
                    tix[t] = (hk_has0 && cls[t] == hk_hcol) ? hk_t0 : tix[t];
                    tix[t] = cls[t] == hk_pcol ? hk_t1 : tix[t];
                }
                const long long p = wv_readlane_i64(rw_p, cur);
                ema_run = ema_after_mover(ema_run, (double)(p - last_mover));
                last_mover = p;
                lik += K;
                hk_i = wv_readlane_i64(rw_i, cur);
                if (lane == 0) {
                    mvI[nmoves] = hk_i; mvSub[nmoves] = hk_has0 ? hk_h : -1; mvAdd[nmoves] = colSlot[hk_pcol]; mvInit[nmoves] = 0;
                }
                nmoves += 1;
                // the next visit, without the two columns in the making (a singleton home is found out behind the barrier)
                const int rn = cur + 1;
                if (rn < nrows && pf_row == rn && pf_ver == mapver)
                    pre_draw(rn, hk_has0 ? hk_hcol : -1, hk_pcol, hk_t0, hk_t1);
            }
            GPROF(5);
        } else if ((wave == 1 || wave == 2) && (S.pub_mode == 0 ? (wave == 2 || S.pub_has0) : wave - 1 < S.upd_n)) {
            // ---- one touched column: the new term and the column's weights for the later rows ----
            // (mode 0: wave 1 = the column x leaves, wave 2 = the column it joins)
            const int mode = S.pub_mode;
            const int r = cur;
            int cl, sg, t, base, slot, n_new, prev0, eidx;
            double rcf_old;
            if (mode == 0) {
                const int k = wave - 1;
                cl = k == 0 ? S.pub_hcol : S.pub_pcol;
                sg = k == 0 ? -1 : 1;
                t = S.pub_term + (k == 1 ? S.pub_has0 : 0);
                prev0 = colLast[cl];
                base = __builtin_amdgcn_readfirstlane(colBase[cl]);
                const int n_old = colN[cl], n0 = colN0[cl];
                slot = colSlot[cl];
                rcf_old = colRCF[cl];
                n_new = __builtin_amdgcn_readfirstlane(n_old + sg);
                const int ee = n_new - n0 + 2;
                eidx = __builtin_amdgcn_readfirstlane((ee >= 0 && ee <= 4 && n_new >= 1) ? ee : -1);
            } else {
                const int k = wave - 1;
                cl = S.upd[k].col; sg = S.upd[k].sigma; t = S.upd[k].term; slot = S.upd[k].slot; prev0 = S.upd[k].prev;
                base = __builtin_amdgcn_readfirstlane(S.upd[k].base);
                n_new = __builtin_amdgcn_readfirstlane(S.upd[k].n_new);
                eidx = __builtin_amdgcn_readfirstlane(S.upd[k].eidx);
                rcf_old = S.upd[k].rcf;
            }
            const bool act = lane >= r && lane < nrows;
            const int lrow = act ? lane : r;
#ifdef BGMM_PROFILE
            long long uk = tk, uk2;
#define UP(i, V) do { asm volatile("" :: "v"(V)); if (wave == 2 && lane == 0) { uk2 = clock64(); S.prof[i] += uk2 - uk; uk = uk2; } } while (0)
            UP(9, lrow);
#else
#define UP(i, V) do { } while (0)
#endif
            // the row of the cross forms, the diagonal, the column's constants for its new count: one round trip
            // (the column's constants for ALL five counts it may have, lane k = the k-th of its 40 doubles: one vector load
            //  beside the row of cross forms -- the count picks its eight by v_readlane afterwards.  Scalar loads from an
            //  address that depends on the count were a second round trip to memory behind the first.)
            double crow, cd0, ccv;
            if (wave == 1 && hp_base == base && hp_r == r) {
                crow = hp_crow; cd0 = hp_cd0; ccv = hp_cc;
            } else {
                crow = gram_ld(d.gC, ((unsigned)base * GR + (unsigned)r) * GR + (unsigned)lrow);
                cd0 = gram_ld(d.gq0, (unsigned)base * GR + (unsigned)lrow);
                ccv = gram_ld(d.gcc, (unsigned)base * 40 + (unsigned)(lane < 40 ? lane : 0));
            }
            const int ce = 8 * (eidx >= 0 ? eidx : 2);
            const double ik0 = wv_readlane(ccv, ce), c_ikn = wv_readlane(ccv, ce + 1), c_icv = wv_readlane(ccv, ce + 2),
                         c_hv = wv_readlane(ccv, ce + 3), c_cb = wv_readlane(ccv, ce + 4), logdet0 = wv_readlane(ccv, ce + 6),
                         c_c7 = wv_readlane(ccv, ce + 7);
            UP(10, ik0);
            const double rM = rowM[lrow];
            const bool own = rowhome[lrow] == slot && n_new >= 2;
            const double kN_new = d.k0 + (double)n_new;
            double ikn = c_ikn, inv_cv = c_icv, hv = c_hv, cbase = c_cb, c7 = c_c7;
            if (eidx < 0) {
                // (rare) a count further from the frozen one than the prepared constants reach
                const SlotTab tab = load_slot_tab(d, n_new);
                const long long v = d.v0 + n_new - d.D + 1;
                ikn = fm_div(1.0, kN_new);
                inv_cv = fm_div(kN_new, kN_new + 1.0);
                hv = 0.5 * (double)(v + d.D);
                cbase = tab.seat + (tab.g - 0.5 * ((double)d.D * tab.lc + logdet0)) - 0.5 * log(fm_div(1.0, ik0 * kN_new));
                c7 = tab.seat1 + tab.g1 - 0.5 * (double)d.D * tab.lc1;
            }
            double acc = crow + ik0, cdv = cd0 + ik0;
            for (int tt = prev0; tt >= 0; tt = termPrev[tt]) {
                const double wr = wvv[tt * GR + r], wl = wvv[tt * GR + lrow], id = termInvD[tt];
                acc = fma(-(wl * wr), id, acc);
                cdv = fma(-(wl * wl), id, cdv);
            }
            // c_t(y, y) = cdv - acc^2 / D_t,  D_t = sigma + c_{t-1}(x, x);  det S_N now = det S_N (frozen) k_N0 / k_N prod |D_t|
            UP(11, acc);
            const double Dt = (double)sg + wv_readlane(acc, r);
            const double invD = fm_div(1.0, Dt);
            UP(12, invD);
            const double qv = fma(-(acc * acc), invD, cdv) - ikn;        // quadratic form of the predictive
            const double f = qv * inv_cv;
            const double rcf_new = rcf_old * fm_rsqrt(fabs(Dt));
            const bool fsmall = fabs(f) < 0.28;
            const double l1 = __builtin_amdgcn_ballot_w64(!fsmall) == 0 ? fm_log1p_small(f) : fm_log(1.0 + f);
            UP(13, l1);
            double ee = fm_exp((cbase - rM) - hv * l1) * rcf_new;
            UP(14, ee);
            const unsigned long long mown = __ballot(own && act && lane > r);
#ifdef BGMM_PROFILE
            if (wave == 2 && lane == 0 && mown) S.prof[7] += 1;
#endif
            if (mown) {
                // rows whose home this column is: the visited point removed from it (slot_math.h, home form; the constants
                // that depend on the count alone came with the column's other constants: no table look-up on the chain)
                const double a1 = fm_div(kN_new, kN_new - 1.0);
                const double den = 1.0 - a1 * qv;
                const double hv1 = hv - 0.5;
                const double logdet_now = logdet0 - fm_log(ik0 * kN_new * (rcf_new * rcf_new));
                const double cb1 = (c7 - 0.5 * logdet_now) - rM;
                const double e1 = fm_exp(cb1 - 0.5 * fm_log(den) - hv1 * fm_log(1.0 + fm_div(a1 * qv, den)));
                ee = own ? e1 : ee;
            }
            UP(15, ee);
            if (act && lane > r) etT[t * GR + lane] = ee;
            wvv[t * GR + lane] = act ? acc : 0.0;
            if (lane == 0) {
                termInvD[t] = invD;
                termRow[t] = r; termSig[t] = sg; termCol[t] = cl;
                colRCF[cl] = rcf_new;
                if (mode == 0) { termPrev[t] = prev0; colLast[cl] = t; colN[cl] = n_new; }
                if (!((double)sg * Dt > 0.0) || !(rcf_new > 0.0)) S.err = -4;
                else if (d.safe_mode) {
                    // The visits between the listed rows were proven to stay under the frozen state and ANY terms within
                    // the budget (kernels_safe.hip): |log c_t(y, y) - log c_0(y, y)| <= sum |log |D_i||  for every y.
                    // A column that leaves its budget (or drifts kSafeDn members from its frozen count) ends the window
                    // right behind this move -- and so does a component OPENED by it: the proofs are about the frozen labels.
                    SafeCol e = safecol_unpack(colE[cl]);
                    float wm = colWm[cl];
                    const bool small_col = e.lo == 32767;                        // (a small label: what leaves it is free)
                    const double wterm = fabs(log(fabs(Dt)));
                    // (two accounts, bgmm_device.h: joins and leaves are charged separately, rounded up)
                    if (sg > 0) e.w = (float)((double)e.w + wterm) * 1.000001f;
                    else if (!small_col) wm = (float)((double)wm + wterm) * 1.000001f;
                    colE[cl] = safecol_pack(e);
                    colWm[cl] = wm;
                    S.wsum2[wave - 1] += wterm;                                  // (waves 1 and 2 each keep their own)
                    S.wterms2[wave - 1] += 1;
                    const int dn = n_new - colN0[cl];
                    if ((double)(e.w > wm ? e.w : wm) > S.cap) S.cut = 1;        // (out of budget: the budget follows, below)
                    else if (cl >= K0 || dn > (int)e.hi || -dn > (int)e.lo) S.cut = 2;
                }
            }
            if (wave == 1 && r + 1 < nrows) {
                // the home side of the next visit (its column is known; wasted if that visit stays)
                const int hc = rowhcol[r + 1];
                if (hc >= 0) {
                    const int hb = __builtin_amdgcn_readfirstlane(colBase[hc]);
                    hp_base = hb; hp_r = r + 1;
                    const int l2 = (lane >= r + 1 && lane < nrows) ? lane : r + 1;
                    hp_crow = gram_ld(d.gC, ((unsigned)hb * GR + (unsigned)(r + 1)) * GR + (unsigned)l2);
                    hp_cd0 = gram_ld(d.gq0, (unsigned)hb * GR + (unsigned)l2);
                    hp_cc = gram_ld(d.gcc, (unsigned)hb * 40 + (unsigned)(lane < 40 ? lane : 0));
                }
            }
#ifdef BGMM_PROFILE
            if (lane == 0) { tk2 = clock64(); S.prof[3 + (wave - 1)] += tk2 - tk; }
#endif
        }
        gram_lds_barrier();
        if (S.err < 0 || S.cut) break;
        cur += 1;
#ifdef BGMM_PROFILE
        tk = clock64();
#endif
    }
    if (wave == 0 && lane == 0) {
        S.K = K; S.nmoves = nmoves; S.lik = lik; S.ema_run = ema_run; S.last_mover = last_mover; S.ncols = ncols;
        S.nterms_end = nterms; S.mapver_end = mapver;      // (mapver > 0: a component was opened or deleted inside the window)
    }
    __syncthreads();

    // ---- close the window ------------------------------------------------------------------
    const int consumed = S.event == GEV_MOVE ? S.cur + 1 : S.cur;     // (an error stops behind the visit that raised it)
    // where the next window starts.  Listed rows: behind the move that ended the window, AT the row that could not be
    // drawn (stale reference point / no room for its terms), or -- every listed row walked -- at the end of the stretch
    // the proof pass vouched for.
    long long next_pos = pos0 + consumed;
    if (d.safe_mode) {
        if (S.event == GEV_MOVE) next_pos = d.glist[c->gl_off + S.cur] + 1;
        else if (S.event == GEV_CUT) next_pos = d.glist[c->gl_off + S.cur];
        else next_pos = c->gl_end;
    }
    for (int k = tid; k < S.nmoves; k += GRT) {
        GramMove mv;
        mv.i = mvI[k]; mv.sub_slot = mvSub[k]; mv.add_slot = mvAdd[k]; mv.add_init = mvInit[k]; mv.pad = 0;
        d.gmoves[k] = mv;
        d.z[mv.i] = mv.add_slot;
    }
    // counts of every column that changed; the live ones go on the finish kernel's list
    for (int cl = tid; cl < S.ncols; cl += GRT) {
        if (cl == cprior || colLast[cl] < 0 || colSlot[cl] < 0) continue;
        const int s = colSlot[cl], n = colN[cl];
        // (pipelined windows: gram_finish_kernel writes the count -- it runs beside the NEXT resolver, whose close would
        //  otherwise overwrite the count this window's finish still has to read.  EVERY count, a deleted component's zero too:
        //  until round 6 the resolver wrote that one itself, and the finish kernel of the window BEFORE -- eight chains' at
        //  D = 128 is three rounds of workgroups, 437 us against this kernel's 208 -- could get to the slot after this close
        //  and write its old count back: a dead slot with a member, one failed C5 chain in ten at the end of a first sweep,
        //  where the random start's components die)
        if (!d.pipe) d.n[s] = n;
        if (n > 0 || d.pipe) {
            const int at = atomicAdd(&d.gfin[0], 1);
            d.gtouched[at] = s;
            d.gfin[16 + at] = n;
            if (d.pipe) ((GramXp *)d.xp_out)->tcol[at] = cl;
            // (gram_finish_kernel: rebuilds come in bunches -- not for pipelined windows, whose finish kernels run beside
            //  the next resolver: a flag read from counters they are updating would make the route a matter of timing)
            if (!d.pipe && d.nupd[s] + 16 > kGramRefreshEvery) d.gfin[2] = 1;
        }
        if (d.safe_mode) {
            // what the label has used of its budget, and what it may still lose / gain, for the stretch's next window
            SafeCol e = safecol_unpack(colE[cl]);
            const int dn = n - colN0[cl];
            if (e.lo != 32767) e.lo = (short)((int)e.lo + dn);
            if (e.hi != 32767) e.hi = (short)((int)e.hi - dn);
            e.wm = colWm[cl]; e.pad = 0;
            d.ep_state[s] = e;
        }
    }
    if (d.pipe) {
        // ---- what the next window's carry needs: the terms (column, row, sign, 1 / D, chain links, w vectors over this
        // window's rows) and every column's count before / after, slot, last term and prod |D|^-1/2
        GramXp *xp = (GramXp *)d.xp_out;
        const int nt = S.nterms_end;
        for (int t = tid; t < nt; t += GRT) {
            xp->termCol[t] = termCol[t]; xp->termRow[t] = termRow[t]; xp->termPrev[t] = termPrev[t]; xp->termSigma[t] = termSig[t];
            xp->termInvD[t] = termInvD[t];
        }
        for (int e = tid; e < nt * GR; e += GRT) xp->wv[e] = wvv[e];
        for (int cl = tid; cl < S.ncols; cl += GRT) {
            xp->colLast[cl] = colLast[cl]; xp->colN[cl] = colN[cl]; xp->colN0[cl] = colN0[cl]; xp->colSlot[cl] = colSlot[cl];
            xp->colRCF[cl] = colRCF[cl];
        }
        __syncthreads();                               // (the list of touched columns is complete)
        if (tid == 0) {
            // carried on from here only if the window was walked to its end as it was laid out: all its rows, the labels
            // unchanged, nothing wrong
            const bool clean = S.err >= 0 && S.cut == 0 && S.mapver_end == 0 && consumed == nrows && S.event == GEV_DONE;
            xp->hdr[0] = atomicAdd(&d.gfin[0], 0);
            xp->hdr[1] = nt;
            xp->hdr[3] = S.ncols;
            __threadfence();
            xp->hdr[2] = clean ? 1 : 0;
            if (!clean) c->pipe_break = 1;
        }
    }
    if (tid == 0) {
        const Job &j = c->job;
        c->n_steps += 1;
        c->n_score_launches += 1;
        c->gram_windows += 1;
        c->win_seq += 1;                               // (gram_finish stamps the slots this window touched with it: Dev::touch_seq)
        c->gram_rows_total += consumed;
        c->n_scored += (long long)nrows * (j.K + 1);
        c->n_pairs_exact += (unsigned long long)nrows * (unsigned long long)(j.K + 1);
        c->lik_evals += S.lik;
        c->n_moves += S.nmoves;
        d.gfin[1] = S.nmoves;
        if (d.safe_mode) {
            const long long adv = next_pos - pos0;
            c->safe_windows += 1;
            c->safe_rows += consumed;
            c->safe_cuts += S.cut ? 1 : 0;
            c->lik_evals += (adv - consumed) * (long long)j.K;         // (pairs decided without being scored)
            if (!(d.safe_cap > 0.0)) {
                // the budget follows the chain (bgmm_device.h: the controller's fields)
                if (S.wterms2[0] + S.wterms2[1] > 0) {
                    const double wm = (S.wsum2[0] + S.wsum2[1]) / (double)(S.wterms2[0] + S.wterms2[1]);
                    c->safe_wbar = c->safe_wbar > 0.0 ? 0.9 * c->safe_wbar + 0.1 * wm : wm;
                }
                c->safe_adv_sum += (double)adv;
                c->safe_phase_cnt += 1;
                if (c->safe_try == 0 && c->safe_phase_cnt >= 12) {
                    c->safe_base_rate = c->safe_adv_sum / (double)c->safe_phase_cnt;
                    int dir = c->safe_next_dir >= 0 ? 1 : -1;
                    double trial = dir > 0 ? c->safe_mult * 2.0 : c->safe_mult * 0.5;
                    if (trial > 64.0 || trial < 1.0) { dir = -dir; trial = dir > 0 ? c->safe_mult * 2.0 : c->safe_mult * 0.5; }
                    c->safe_mult_base = c->safe_mult;
                    c->safe_mult = trial;
                    c->safe_try = dir;
                    c->safe_adv_sum = 0.0; c->safe_phase_cnt = 0;
                } else if (c->safe_try != 0 && c->safe_phase_cnt >= 6) {
                    const double rate = c->safe_adv_sum / (double)c->safe_phase_cnt;
                    if (rate > 1.1 * c->safe_base_rate) c->safe_next_dir = c->safe_try;          // keep it, and try further the same way
                    else { c->safe_mult = c->safe_mult_base; c->safe_next_dir = -c->safe_try; }
                    c->safe_try = 0;
                    c->safe_adv_sum = 0.0; c->safe_phase_cnt = 0;
                }
                if (c->safe_wbar > 0.0) {
                    // (what a budget costs the proofs: the home's bound gives way by about D / 2 (e^cap - 1) nats -- kept below 30)
                    const double cap_max = log1p(60.0 / (double)d.D);
                    double cap = c->safe_mult * c->safe_wbar;
                    cap = cap < 1.0 / 1024.0 ? 1.0 / 1024.0 : (cap > cap_max ? cap_max : cap);
                    // (quantised to a twelfth of an octave: the robust tables are rebuilt only when the budget really moves)
                    c->safe_cap = exp2(rint(log2(cap) * 12.0) / 12.0);
                }
            }
            // the stretch's proofs: they stand while every label stays inside its budget and allowance and no label is opened;
            // the next stretch is sized by what this one got through (twice that: half of a proof pass wasted at worst)
            c->gl_off += consumed;
            if (S.cut != 0 || next_pos >= c->gl_stretch_end || S.err < 0) {
                c->safe_epoch_valid = 0;
                long long L = next_pos >= c->gl_stretch_end && S.cut == 0 ? 2ll * (c->gl_stretch_end - c->safe_epoch_pos0)
                                                                          : 2ll * (next_pos - c->safe_epoch_pos0) + 64;
                // (one window per stretch with the look-ahead -- kernels_safe.hip: safe_compact_kernel: a cut says nothing about
                //  how far the next window's worth of unproven visits reaches; the stretch keeps its length)
                if (d.safe_dense && d.ahead_C > 0 && S.cut != 0 && L < c->safe_L) L = c->safe_L;
                if (L < 1024) L = 1024;
                if (L > (1ll << 22)) L = 1ll << 22;
                c->safe_L = (int)L;
            } else {
                safe_next_window(d, c);
            }
        }
        c->ema_run = S.ema_run;
        c->last_mover = S.last_mover;
        if (S.nmoves > 0) { c->tables_valid = 0; c->wsort_valid = 0; c->state_epoch += 1; }
        c->first_mover = kNoMover;
        c->n_refresh = 0;
        c->skip_apply = 0;
        c->job.K = S.K;
#ifdef BGMM_PROFILE
        for (int k = 0; k < 8; ++k) c->prof[k] += S.prof[k];
        c->prof[8] += 1;
        for (int k = 9; k < 16; ++k) c->prof[k] += S.prof[k];
        c->prof[7] += S.prof[7];
        c->prof[6] += clock64() - prof_c0;           // (the whole launch in s_memtime ticks)
#endif
        if (S.err < 0) {
            atomicCAS(&c->error, 0, S.err);
            c->job.mode = MODE_DONE;
        } else {
            c->win_size = (int)window_for_rate(c);
            start_window(d, c, next_pos);
            if (d.safe_mode && !c->safe_epoch_valid) safe_open_window(d, c);
        }
    }
}
template <int LPL, int KC, int T>
__global__ __launch_bounds__(GRT) void gram_resolve_kernel(Dev d) { gram_resolve_body<LPL, KC, T>(d); }
// (several chains in one launch: workgroup (x, c) works for chain group[c] -- bgmm_group_sweep_staged)
template <int LPL, int KC, int T>
__global__ __launch_bounds__(GRT) void gram_resolve_group_kernel(const Dev *__restrict__ group) {
    const Dev d = group[blockIdx.y];               // (a private copy: nothing the body writes can alias it)
    gram_resolve_body<LPL, KC, T>(d);
}
template <int LPL, int KC, int T>
__global__ __launch_bounds__(GRT) void gram_resolve_pgroup_kernel(const Dev *__restrict__ v0, const Dev *__restrict__ v1, int k) {
    const Dev d = gram_pgroup_view(v0, v1, k, (int)blockIdx.y);
    gram_resolve_body<LPL, KC, T>(d);
}



// ------------------------------------------------------------------------------------------
// Pipelined windows.  A window costs the resolver's walk (one workgroup, ~180 us for 64 moves at D = 64) plus, in series
// with it, gram_finish (the touched factors rebuilt: ~48 us), the next window's cross forms (~14 us), its weights and four
// launch gaps: a quarter of the chain is work that does not have to be on it.  The reference's state after window w - 1
// is the state before it PLUS that window's logged terms (igmm/crpmm.py:82-88: a move is a rank-1 change of two
// components) -- and terms are what the resolver is built on.  So the cross forms of window w are made against the factors
// as they stood BEFORE window w - 1 (gram_cross_kernel: on a second stream, beside the resolver of window w - 1, together
// with the cross forms between the two windows' rows), and gram_carry_kernel -- one workgroup per component window w - 1
// touched, in parallel, ~10 us on the chain -- applies window w - 1's terms to them by the recursion of the header,
//     w_i(y) = c_0(y, x_i) - sum_{j < i} w_j(y) w_j(x_i) / D_j,      c'(y, y') = c_0(y, y') - sum_i w_i(y) w_i(y') / D_i,
// counts and log determinants included: what the resolver of window w then sees is a frozen state equal to the true one,
// and nothing else about it changes.  gram_finish of window w - 1 and the cross forms of window w + 1 follow on the second
// stream while window w is walked.  A window that ends early, opens or deletes a component or fails breaks the chain
// (Ctrl::pipe_break): the rest of the batch stands still and the host goes on with plain windows.
// ------------------------------------------------------------------------------------------
// Terms of ONE column a carry workgroup holds in LDS (16 KB).  Room for all kGramMaxTerms = 128 of a window (64 KB) let two
// workgroups onto a compute unit -- eight chains' carry launches (up to 1 024 workgroups) went round twice and took 73 us on
// the chain of every window against 14 for one chain; a column takes one or two terms per window, thirty-two never in the runs
// measured -- and if it ever does the chain breaks like for any other window the pipeline cannot take.
static constexpr int kGramCarryTerms = 32;
__device__ __forceinline__ void gram_carry_body(const Dev &d) {
    extern __shared__ __attribute__((aligned(16))) double wn_raw[];      // [terms of the column][64] w over the NEW rows
    __shared__ int chain[kGramMaxTerms];
    __shared__ double chain_id[kGramMaxTerms];
    Ctrl *c = d.ctrl;
    if (c->job.mode == MODE_DONE || c->error != 0 || c->pipe_break) return;
    const GramXp *__restrict__ xp = (const GramXp *)d.xp_in;
    const int b = blockIdx.x;
    // (one round trip: the verdict on the window before, how many columns it touched, this workgroup's column, the terms)
    const int clean = xp->hdr[2], ntouched = xp->hdr[0], nt = xp->hdr[1];
    const int cl_ = xp->tcol[b < kGramMaxTerms ? b : 0];
    const int tid = threadIdx.x, lane = tid & 63;
    int tc0 = -1, tc1 = -1;
    if (tid < 64) { tc0 = xp->termCol[lane]; tc1 = xp->termCol[64 + lane]; }
    if (!clean || b >= ntouched) return;              // (not clean: that window's resolver has broken the chain)
    const int cl = cl_;
    const long long pos0 = d.pipe_pos;
    const int nrows = gram_nrows_at(d, c, pos0);
    if (nrows <= 0) return;
    LDS_AS double *const wn = (LDS_AS double *)wn_raw;
    const double *__restrict__ Xc = d.gX + (long long)cl * GR * GR;
    double *__restrict__ Cc = d.gC + (long long)cl * GR * GR;
    // the column's block of cross forms sets off now (16 entries per thread), its terms are sorted out meanwhile
    double cv[GR * GR / 256];
#pragma unroll
    for (int k = 0; k < GR * GR / 256; ++k) cv[k] = Cc[tid + 256 * k];
    const int s = xp->colSlot[cl];
    const int n0 = xp->colN0[cl], n1 = xp->colN[cl];
    const double rcf = xp->colRCF[cl];
    const double logdet0 = d.gcc[((long long)cl * 5 + 2) * 8 + 6];
    // the column's terms in the order they were made: a column's terms carry increasing indices
    int m = 0;
    if (tid < 64) {
        const unsigned long long m0 = __ballot(lane < nt && tc0 == cl), m1 = __ballot(64 + lane < nt && tc1 == cl);
        const int c0n = __popcll(m0);
        m = c0n + __popcll(m1);
        if (lane < nt && tc0 == cl) chain[__popcll(m0 & ((1ull << lane) - 1ull))] = lane;
        if (64 + lane < nt && tc1 == cl) chain[c0n + __popcll(m1 & ((1ull << lane) - 1ull))] = 64 + lane;
    }
    const double kN0 = d.k0 + (double)n0, kN1 = d.k0 + (double)n1, ik0 = 1.0 / kN0, ik1 = 1.0 / kN1;
    // ---- the terms' w over the new rows (wave 0, lane = new row: the recursion runs along the column's short chain)
    double q_new = 0.0;
    if (tid < 64 && m > kGramCarryTerms) {
        // (more terms on ONE column than the workgroup's LDS holds -- 64 moves of a window would have to pile up on it: the
        //  chain breaks here, in front of the window's resolver, and the host goes on with plain windows)
        if (lane == 0) { __hip_atomic_store(&c->pipe_break, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); chain[kGramMaxTerms - 1] = m; }
    } else if (tid < 64) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        double xr[4], idv[4];
        int rov[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {                 // (the first terms' loads side by side: most columns have one or two)
            const int t = a < m ? chain[a] : 0;
            rov[a] = xp->termRow[t];
            idv[a] = xp->termInvD[t];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) xr[a] = Xc[lane * GR + rov[a]];
        for (int a = 0; a < m; ++a) {
            const int t = chain[a];
            const int ro = a < 4 ? rov[a] : xp->termRow[t];
            double w = (a < 4 ? xr[a] : Xc[lane * GR + ro]) + ik0;
            for (int q = 0; q < a; ++q) w = fma(-(wn[q * GR + lane] * xp->wv[chain[q] * GR + ro]), chain_id[q], w);
            wn[a * GR + lane] = w;
            if (lane == 0) chain_id[a] = a < 4 ? idv[a] : xp->termInvD[t];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        double v = d.gq0[(long long)cl * GR + lane] + (ik0 - ik1);
        for (int a = 0; a < m; ++a) v = fma(-(wn[a * GR + lane] * wn[a * GR + lane]), chain_id[a], v);
        d.gq0[(long long)cl * GR + lane] = v;
        q_new = v;                                     // c'(y, y) - 1 / k_N1: the quadratic form of the predictive at the new count
        if (lane == 0) chain[kGramMaxTerms - 1] = m;
    }
    __syncthreads();
    m = chain[kGramMaxTerms - 1];
    if (m > kGramCarryTerms) return;
    // ---- the cross forms of the new window under the true state: every entry of the column's 64 x 64 block (the stored
    // form leaves 1 / k_N out: the resolver adds the NEW count's, so the old one's is swapped for it here)
    const double shift = ik0 - ik1;
#pragma unroll
    for (int k = 0; k < GR * GR / 256; ++k) {
        const int e = tid + 256 * k, r = e >> 6, r2 = e & 63;
        double v = cv[k] + shift;
        for (int a = 0; a < m; ++a) v = fma(-(wn[a * GR + r] * wn[a * GR + r2]), chain_id[a], v);
        Cc[e] = v;
    }
    // ---- the column's constants for the counts around its new one (gram_body's block, with the carried log determinant:
    // logdet S_N now = logdet S_N (frozen) + log(k_N0 / k_N1) + sum log |D_i|, and prod |D_i|^-1/2 is what the resolver kept)
    const double logdet1 = logdet0 + fm_log(kN0 * ik1) - 2.0 * fm_log(rcf);
    if (tid >= 64 && tid < 69) {
        const int e5 = tid - 64, nn = n1 + e5 - 2;
        double *g = d.gcc + ((long long)cl * 5 + e5) * 8;
        g[0] = ik1; g[6] = logdet1; g[5] = -1.0;
        if (nn >= 1) {
            const SlotTab tab = load_slot_tab(d, nn);
            const double kN = d.k0 + (double)nn;
            const long long v = d.v0 + nn - d.D + 1;
            g[1] = 1.0 / kN;
            g[2] = kN / (kN + 1.0);
            g[3] = 0.5 * (double)(v + d.D);
            g[4] = tab.seat + (tab.g - 0.5 * ((double)d.D * tab.lc + logdet1)) - 0.5 * fm_log(kN1 / kN);
            g[5] = (double)nn;
            g[7] = tab.seat1 + tab.g1 - 0.5 * (double)d.D * tab.lc1;
        }
    }
    // ---- the frozen weights of the new rows under this column: exp(lp - M_r), as an update wave of the resolver would
    // leave them for a column without a term
    if (tid < 64 && lane < nrows && n1 >= 1) {
        const SlotTab tab = load_slot_tab(d, n1);
        const long long v = d.v0 + n1 - d.D + 1;
        const double inv_cv = kN1 / (kN1 + 1.0), hv = 0.5 * (double)(v + d.D);
        const double cbase = tab.seat + (tab.g - 0.5 * ((double)d.D * tab.lc + logdet1));
        const double rM = d.gM[lane];
        const long long p = pos0 + lane;
        const long long i = d.order ? d.order[p] : p;
        const bool own = d.z[i] == s && n1 >= 2;
        double ee = fm_exp((cbase - rM) - hv * fm_log(1.0 + q_new * inv_cv));
        if (own) {
            const double a1 = kN1 / (kN1 - 1.0), den = 1.0 - a1 * q_new, hv1 = hv - 0.5;
            const double c7 = tab.seat1 + tab.g1 - 0.5 * (double)d.D * tab.lc1;
            ee = fm_exp(((c7 - 0.5 * logdet1) - rM) - 0.5 * fm_log(den) - hv1 * fm_log(1.0 + a1 * q_new / den));
        }
        d.ge0[(long long)lane * d.gcols + cl] = ee;
    }
}

__global__ __launch_bounds__(256) void gram_carry_kernel(Dev d) { gram_carry_body(d); }
__global__ __launch_bounds__(256) void gram_carry_pgroup_kernel(const Dev *__restrict__ v0, const Dev *__restrict__ v1, int k) {
    const Dev d = gram_pgroup_view(v0, v1, k, (int)blockIdx.y);
    gram_carry_body(d);
}

void launch_gram_carry(const Dev &d, hipStream_t st) {
    const int lds = kGramCarryTerms * GR * (int)sizeof(double);
    static PerDeviceLds attr;
    attr.ensure((const void *)gram_carry_kernel, lds);
    hipLaunchKernelGGL(gram_carry_kernel, dim3(kGramMaxTerms), dim3(256), lds, st, d);
}

template <int NJ>
static void launch_gram_t(const Dev &d, hipStream_t st) {
    const int lds = GR * (16 * NJ + 2) * (int)sizeof(double);
    hipLaunchKernelGGL((gram_kernel<NJ>), dim3(d.gcols), dim3(256), lds, st, d);
}

template <int NJ>
static void launch_gram_group_t(const Dev &lead, const Dev *group, int G, hipStream_t st) {
    const int lds = GR * (16 * NJ + 2) * (int)sizeof(double);
    hipLaunchKernelGGL((gram_group_kernel<NJ>), dim3(lead.gcols, G), dim3(256), lds, st, group);
}

template <int NJ>
static void configure_gram_t() {
    const int lds = GR * (16 * NJ + 2) * (int)sizeof(double);
    if (lds > 64 * 1024) {
        (void)hipFuncSetAttribute((const void *)gram_kernel<NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute((const void *)gram_group_kernel<NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    }
}

template <int LPL, int KC, int T>
static void configure_resolve_t() {
    (void)hipFuncSetAttribute((const void *)gram_resolve_kernel<LPL, KC, T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)GramPlan<KC, T>::bytes);
    (void)hipFuncSetAttribute((const void *)gram_resolve_group_kernel<LPL, KC, T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)GramPlan<KC, T>::bytes);
    (void)hipFuncSetAttribute((const void *)gram_resolve_pgroup_kernel<LPL, KC, T>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)GramPlan<KC, T>::bytes);
}

// Kernel attributes are per device: set for the device of the calling context (current device).
void gram_configure(const Dev &d, int resolve_lds) {
    configure_gram_t<5>(); configure_gram_t<6>(); configure_gram_t<7>(); configure_gram_t<8>();
    configure_resolve_t<2, kPlanA_KC, kPlanA_T>(); configure_resolve_t<4, kPlanA_KC, kPlanA_T>();
    configure_resolve_t<6, kPlanA_KC, kPlanA_T>(); configure_resolve_t<8, kPlanB_KC, kPlanB_T>();
    (void)d; (void)resolve_lds;
}

// One frozen-factor step: cross forms, weights, the sequential walk, statistics + factors of the touched
// slots.  ev0 / ev1 (optional) bracket the likelihood kernel.
bool launch_gram_step(const Dev &d, int resolve_lds, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    return launch_gram_core(d, resolve_lds, st, ev0, ev1);
}

bool launch_gram_core(const Dev &d, int resolve_lds, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
    if (ev0) (void)hipEventRecord(ev0, st);
    switch (d.Dp / 16) {
        case 1: launch_gram_t<1>(d, st); break;
        case 2: launch_gram_t<2>(d, st); break;
        case 3: launch_gram_t<3>(d, st); break;
        case 4: launch_gram_t<4>(d, st); break;
        case 5: launch_gram_t<5>(d, st); break;
        case 6: launch_gram_t<6>(d, st); break;
        case 7: launch_gram_t<7>(d, st); break;
        case 8: launch_gram_t<8>(d, st); break;
        default: return false;
    }
    if (ev1) (void)hipEventRecord(ev1, st);
    hipLaunchKernelGGL(gram_weights_kernel, dim3(GR), dim3(256), 0, st, d);
    // labels per lane of the draw wave: the labels a window can reach (room for the components a
    // batch of windows may open before the host looks again: the kernel stalls the step otherwise)
    const int reach = d.gram_K + d.gram_terms / 2 + 2 + 32;
    if (d.gcols == kPlanA_KC) {
        if (reach <= 128) hipLaunchKernelGGL((gram_resolve_kernel<2, kPlanA_KC, kPlanA_T>), dim3(1), dim3(GRT), resolve_lds, st, d);
        else if (reach <= 256) hipLaunchKernelGGL((gram_resolve_kernel<4, kPlanA_KC, kPlanA_T>), dim3(1), dim3(GRT), resolve_lds, st, d);
        else hipLaunchKernelGGL((gram_resolve_kernel<6, kPlanA_KC, kPlanA_T>), dim3(1), dim3(GRT), resolve_lds, st, d);
    } else {
        hipLaunchKernelGGL((gram_resolve_kernel<8, kPlanB_KC, kPlanB_T>), dim3(1), dim3(GRT), resolve_lds, st, d);
    }
    launch_gram_finish(d, st);
    return true;
}

template <int NJ>
static void launch_gram_cross_t(const Dev &d, bool with_previous, hipStream_t st) {
    const int lds = 2 * GR * (16 * NJ + 2) * (int)sizeof(double);
    static PerDeviceLds attr;
    attr.ensure((const void *)gram_cross_kernel<NJ>, lds);
    hipLaunchKernelGGL((gram_cross_kernel<NJ>), dim3(d.gcols), dim3(256), lds, st, d, with_previous ? 1 : 0);
}

bool launch_gram_cross(const Dev &d, bool with_previous, hipStream_t st) {
    switch (d.Dp / 16) {
        case 1: launch_gram_cross_t<1>(d, with_previous, st); break;
        case 2: launch_gram_cross_t<2>(d, with_previous, st); break;
        case 3: launch_gram_cross_t<3>(d, with_previous, st); break;
        case 4: launch_gram_cross_t<4>(d, with_previous, st); break;
        case 5: launch_gram_cross_t<5>(d, with_previous, st); break;
        case 6: launch_gram_cross_t<6>(d, with_previous, st); break;
        case 7: launch_gram_cross_t<7>(d, with_previous, st); break;
        case 8: launch_gram_cross_t<8>(d, with_previous, st); break;
        default: return false;
    }
    hipLaunchKernelGGL(gram_weights_kernel, dim3(GR), dim3(256), 0, st, d);
    return true;
}

bool launch_gram_resolve_only(const Dev &d, int resolve_lds, hipStream_t st) {
    const int reach = d.gram_K + d.gram_terms / 2 + 2 + 32;
    if (d.gcols == kPlanA_KC) {
        if (reach <= 128) hipLaunchKernelGGL((gram_resolve_kernel<2, kPlanA_KC, kPlanA_T>), dim3(1), dim3(GRT), resolve_lds, st, d);
        else if (reach <= 256) hipLaunchKernelGGL((gram_resolve_kernel<4, kPlanA_KC, kPlanA_T>), dim3(1), dim3(GRT), resolve_lds, st, d);
        else hipLaunchKernelGGL((gram_resolve_kernel<6, kPlanA_KC, kPlanA_T>), dim3(1), dim3(GRT), resolve_lds, st, d);
    } else {
        hipLaunchKernelGGL((gram_resolve_kernel<8, kPlanB_KC, kPlanB_T>), dim3(1), dim3(GRT), resolve_lds, st, d);
    }
    return true;
}

// The same step for G chains of one shape (same D, same column plan) in SHARED launches: workgroup (x, c) of every kernel
// works for chain group[c] (a device array of their views).  `reach`: the labels the widest of them can reach in a window.
bool launch_gram_group_step(const Dev &lead, const Dev *group, int G, int reach, int resolve_lds, hipStream_t st) {
    switch (lead.Dp / 16) {
        case 1: launch_gram_group_t<1>(lead, group, G, st); break;
        case 2: launch_gram_group_t<2>(lead, group, G, st); break;
        case 3: launch_gram_group_t<3>(lead, group, G, st); break;
        case 4: launch_gram_group_t<4>(lead, group, G, st); break;
        case 5: launch_gram_group_t<5>(lead, group, G, st); break;
        case 6: launch_gram_group_t<6>(lead, group, G, st); break;
        case 7: launch_gram_group_t<7>(lead, group, G, st); break;
        case 8: launch_gram_group_t<8>(lead, group, G, st); break;
        default: return false;
    }
    hipLaunchKernelGGL(gram_weights_group_kernel, dim3(GR, G), dim3(256), 0, st, group);
    if (lead.gcols == kPlanA_KC) {
        if (reach <= 128) hipLaunchKernelGGL((gram_resolve_group_kernel<2, kPlanA_KC, kPlanA_T>), dim3(1, G), dim3(GRT), resolve_lds, st, group);
        else if (reach <= 256) hipLaunchKernelGGL((gram_resolve_group_kernel<4, kPlanA_KC, kPlanA_T>), dim3(1, G), dim3(GRT), resolve_lds, st, group);
        else hipLaunchKernelGGL((gram_resolve_group_kernel<6, kPlanA_KC, kPlanA_T>), dim3(1, G), dim3(GRT), resolve_lds, st, group);
    } else {
        hipLaunchKernelGGL((gram_resolve_group_kernel<8, kPlanB_KC, kPlanB_T>), dim3(1, G), dim3(GRT), resolve_lds, st, group);
    }
    launch_gram_finish_group(lead, group, G, st);
    return true;
}

// ---- the pipelined windows of several chains in shared launches (api_group.hip: gram_group_pipe_launch) -----------------------
// (cols: columns 0 .. cols - 1 get a workgroup.  A pipelined batch stands still from the window on that opens or deletes a
//  component, so a chain has the K labels -- and the new table's column K -- it started the batch with: the plan's 448
//  columns, most of whose workgroups returned at once but waited for 67 KB of LDS first, were seven rounds of eight chains'
//  launch)
template <int NJ>
static void launch_gram_cross_pgroup_t(const Dev &lead, const Dev *v0, const Dev *v1, int G, int k, bool with_previous, int cols, hipStream_t st) {
    const int lds = 2 * GR * (16 * NJ + 2) * (int)sizeof(double);
    static PerDeviceLds attr;
    attr.ensure((const void *)gram_cross_pgroup_kernel<NJ>, lds);
    hipLaunchKernelGGL((gram_cross_pgroup_kernel<NJ>), dim3(cols, G), dim3(256), lds, st, v0, v1, k, with_previous ? 1 : 0);
}
bool launch_gram_cross_pgroup(const Dev &lead, const Dev *v0, const Dev *v1, int G, int k, bool with_previous, int max_K, hipStream_t st) {
    const int cols = max_K + 2 < lead.gcols ? max_K + 2 : lead.gcols;
    switch (lead.Dp / 16) {
        case 1: launch_gram_cross_pgroup_t<1>(lead, v0, v1, G, k, with_previous, cols, st); break;
        case 2: launch_gram_cross_pgroup_t<2>(lead, v0, v1, G, k, with_previous, cols, st); break;
        case 3: launch_gram_cross_pgroup_t<3>(lead, v0, v1, G, k, with_previous, cols, st); break;
        case 4: launch_gram_cross_pgroup_t<4>(lead, v0, v1, G, k, with_previous, cols, st); break;
        case 5: launch_gram_cross_pgroup_t<5>(lead, v0, v1, G, k, with_previous, cols, st); break;
        case 6: launch_gram_cross_pgroup_t<6>(lead, v0, v1, G, k, with_previous, cols, st); break;
        case 7: launch_gram_cross_pgroup_t<7>(lead, v0, v1, G, k, with_previous, cols, st); break;
        case 8: launch_gram_cross_pgroup_t<8>(lead, v0, v1, G, k, with_previous, cols, st); break;
        default: return false;
    }
    hipLaunchKernelGGL(gram_weights_pgroup_kernel, dim3(GR, G), dim3(256), 0, st, v0, v1, k);
    return true;
}
void launch_gram_carry_pgroup(const Dev *v0, const Dev *v1, int G, int k, hipStream_t st) {
    const int lds = kGramCarryTerms * GR * (int)sizeof(double);
    static PerDeviceLds attr;
    attr.ensure((const void *)gram_carry_pgroup_kernel, lds);
    hipLaunchKernelGGL(gram_carry_pgroup_kernel, dim3(kGramMaxTerms, G), dim3(256), lds, st, v0, v1, k);
}
void launch_gram_resolve_pgroup(const Dev &lead, const Dev *v0, const Dev *v1, int G, int k, int reach, int resolve_lds, hipStream_t st) {
    if (lead.gcols == kPlanA_KC) {
        if (reach <= 128) hipLaunchKernelGGL((gram_resolve_pgroup_kernel<2, kPlanA_KC, kPlanA_T>), dim3(1, G), dim3(GRT), resolve_lds, st, v0, v1, k);
        else if (reach <= 256) hipLaunchKernelGGL((gram_resolve_pgroup_kernel<4, kPlanA_KC, kPlanA_T>), dim3(1, G), dim3(GRT), resolve_lds, st, v0, v1, k);
        else hipLaunchKernelGGL((gram_resolve_pgroup_kernel<6, kPlanA_KC, kPlanA_T>), dim3(1, G), dim3(GRT), resolve_lds, st, v0, v1, k);
    } else {
        hipLaunchKernelGGL((gram_resolve_pgroup_kernel<8, kPlanB_KC, kPlanB_T>), dim3(1, G), dim3(GRT), resolve_lds, st, v0, v1, k);
    }
}

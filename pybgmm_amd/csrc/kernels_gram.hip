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

// ------------------------------------------------------------------------------------------
// gram_kernel<NJ>: grid = columns, 256 threads (wave w: rows 16 w .. 16 w + 15).
// Fragment conventions as in kernels_score.hip.  LDS: Ys[64][Dp + 2] (row stride = 2 mod 32 doubles:
// the fragment reads Ys[16 t + (lane & 15)][4 kk + (lane >> 4)] are conflict free).
// ------------------------------------------------------------------------------------------
template <int NJ>
__global__ __launch_bounds__(256) void gram_kernel(Dev d) {
    extern __shared__ __attribute__((aligned(16))) double Ys[];
    const Ctrl *c = d.ctrl;
    if (c->job.mode == MODE_DONE || c->error != 0) return;
    const int K = c->job.K;
    if (K + kGramColSlack > d.gcols) return;               // (the resolver reports the stall)
    const int col = blockIdx.x;
    if (col > K) return;
    const int s = col < K ? d.perm[col] : d.K_max;
    const long long pos0 = c->job.pos;
    const long long left = c->n_visits - pos0;
    const int nrows = left < GR ? (int)left : GR;
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
        const long long p = pos0 + row;
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
                const long long p = pos0 + row;
                const long long i = d.order ? d.order[p] : p;
                const bool own = d.z[i] == s && d.n[s] >= 2;
                d.glp0[(long long)row * d.gcols + col] = slot_score_exact(d.sc[s], v, own);
            }
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
}

// ------------------------------------------------------------------------------------------
// gram_weights_kernel: one workgroup per window row.  Reference point M_r = max of the row's frozen log
// scores (a singleton home is no candidate), frozen weights e0[r][c] = exp(lp0 - M_r), the new table's
// weight (igmm/crpmm.py:74).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gram_weights_kernel(Dev d) {
    __shared__ double red[4];
    const Ctrl *c = d.ctrl;
    if (c->job.mode == MODE_DONE || c->error != 0) return;
    const int K = c->job.K;
    if (K + kGramColSlack > d.gcols) return;
    const long long pos0 = c->job.pos;
    const long long left = c->n_visits - pos0;
    const int nrows = left < GR ? (int)left : GR;
    const int r = blockIdx.x;
    if (r >= nrows) return;
    const long long p = pos0 + r;
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
// One touched column of a move, everything waves 1 / 2 need (one LDS round trip)
struct GramUpd {
    int col, sigma, term, tix, base, slot, n_new, prev;
    double ik, logdet0, logf, pad;
};

struct GramShared {
    int active, nrows, K0, ncols, cprior, err, event, cur;
    int upd_n, K, nmoves, pad0;
    GramUpd upd[2];
    long long pos0, lik, last_mover;
    double ema_run;
    long long prof[8];
};

struct GramLds {
    LDS_AS GramShared *S;
    lds_f64 etT, wv, rowM, termInvD, colLogdet, colIk, colLogF;
    lds_i64 mvI;                 // the window's move log (GramMove fields)
    lds_i32 rowhome, rowhcol, termPrev, colSlot, colN, colTix, colBase, colLab, colLast, labCol, permL;
    lds_i32 mvSub, mvAdd, mvInit;
};

__host__ __device__ inline size_t gram_carve(int Kc, int T, unsigned *o /* 21 offsets or null */) {
    size_t off = 512;                                   // GramShared
    unsigned dummy[21];
    if (!o) o = dummy;
    int k = 0;
    auto take = [&](size_t bytes) { o[k++] = (unsigned)off; off += (bytes + 15) & ~(size_t)15; };
    take(sizeof(double) * (size_t)T * GR);      // 0 etT
    take(sizeof(double) * (size_t)T * GR);      // 1 wv
    take(sizeof(double) * GR);                  // 2 rowM
    take(sizeof(double) * T);                   // 3 termInvD
    take(sizeof(double) * Kc);                  // 4 colLogdet (of the frozen state)
    take(sizeof(double) * Kc);                  // 5 colIk
    take(sizeof(long long) * GR);               // 6 move log: data index
    take(sizeof(int) * GR);                     // 7 rowhome
    take(sizeof(int) * GR);                     // 8 rowhcol
    take(sizeof(int) * T);                      // 9 termPrev
    take(sizeof(int) * Kc);                     // 10 colSlot
    take(sizeof(int) * Kc);                     // 11 colN
    take(sizeof(int) * Kc);                     // 12 colTix
    take(sizeof(int) * Kc);                     // 13 colBase
    take(sizeof(int) * Kc);                     // 14 colLab
    take(sizeof(int) * Kc);                     // 15 colLast
    take(sizeof(int) * Kc);                     // 16 labCol
    take(sizeof(int) * Kc);                     // 17 permL
    take(sizeof(int) * 3 * GR);                 // 18 move log: sub slot, add slot, init flag
    take(sizeof(double) * Kc);                  // 19 colLogF: log(det S_N now / det S_N frozen) of the column
    return off;
}

int gram_resolve_lds_bytes(int gcols, int terms) { return (int)gram_carve(gcols, terms, nullptr); }

__device__ __forceinline__ void gram_lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

enum { GEV_DONE = 0, GEV_MOVE = 1, GEV_CUT = 2 };

#define GRT 256

#ifdef BGMM_PROFILE
#define GPROF(i) do { if (lane == 0) { tk2 = clock64(); S.prof[i] += tk2 - tk; tk = tk2; } } while (0)
#else
#define GPROF(i) do { } while (0)
#endif

__device__ __forceinline__ long long wv_readlane_i64(long long v, int t) {
    const int lo = __builtin_amdgcn_readlane((int)(v & 0xffffffffll), t);
    const int hi = __builtin_amdgcn_readlane((int)(v >> 32), t);
    return ((long long)hi << 32) | (unsigned int)lo;
}

// LPL: labels per lane of the draw wave (label j = lane * LPL + t); 64 LPL - 1 bounds the labels a
// window can reach.
template <int LPL>
__global__ __launch_bounds__(GRT) void gram_resolve_kernel(Dev d, int T) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    static_assert(sizeof(GramShared) <= 512, "GramShared has 512 bytes of LDS");
    GramLds L;
    const int Kc = d.gcols;
    {
        unsigned o[21];
        gram_carve(Kc, T, o);
        LDS_AS unsigned char *b = (LDS_AS unsigned char *)lds_raw;
        L.S = (LDS_AS GramShared *)b;
        L.etT = (lds_f64)(b + o[0]); L.wv = (lds_f64)(b + o[1]); L.rowM = (lds_f64)(b + o[2]);
        L.termInvD = (lds_f64)(b + o[3]); L.colLogdet = (lds_f64)(b + o[4]); L.colIk = (lds_f64)(b + o[5]);
        L.mvI = (lds_i64)(b + o[6]);
        L.rowhome = (lds_i32)(b + o[7]); L.rowhcol = (lds_i32)(b + o[8]); L.termPrev = (lds_i32)(b + o[9]);
        L.colSlot = (lds_i32)(b + o[10]); L.colN = (lds_i32)(b + o[11]); L.colTix = (lds_i32)(b + o[12]);
        L.colBase = (lds_i32)(b + o[13]); L.colLab = (lds_i32)(b + o[14]); L.colLast = (lds_i32)(b + o[15]);
        L.labCol = (lds_i32)(b + o[16]); L.permL = (lds_i32)(b + o[17]);
        L.mvSub = (lds_i32)(b + o[18]); L.mvAdd = L.mvSub + GR; L.mvInit = L.mvAdd + GR;
        L.colLogF = (lds_f64)(b + o[19]);
    }
    LDS_AS GramShared &S = *L.S;
    Ctrl *c = d.ctrl;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long gld = d.gcols;
#ifdef BGMM_PROFILE
    long long tk = clock64(), tk2;
#endif

    if (tid == 0) {
        S.active = 0;
        c->gram_ntouched = 0;                   // (an idle step must not replay the last window's lists)
        c->gram_nmoves = 0;
        const Job &j = c->job;
        if (j.mode != MODE_DONE && c->error == 0) {
            if (j.K + kGramColSlack > d.gcols || j.K + T / 2 + 2 > 64 * LPL) {
                c->gram_stall = 1;              // more columns than allocated / labels than the draw wave holds
            } else {
                S.active = 1;
                S.pos0 = j.pos;
                const long long left = c->n_visits - j.pos;
                S.nrows = left < GR ? (int)left : GR;
                S.K0 = j.K; S.K = j.K;
                S.cprior = j.K;
                S.ncols = j.K + 1;
                S.err = 0; S.cur = 0; S.event = GEV_DONE; S.nmoves = 0; S.upd_n = 0;
                S.lik = 0;
                S.ema_run = c->ema_run; S.last_mover = c->last_mover;
                for (int k = 0; k < 8; ++k) S.prof[k] = 0;
            }
        }
    }
    __syncthreads();
    if (!S.active) return;
    const int nrows = S.nrows, K0 = S.K0, cprior = S.cprior;
    const long long pos0 = S.pos0;
    // rows: lane r of wave 0 keeps row r's inputs in registers; the update waves read home / home column / M_r from LDS
    int rw_home = -1, rw_hcol = -1;
    long long rw_i = 0;
    double rw_u = 0.0, rw_En = 0.0;
    if (wave == 0 && lane < nrows) {
        const long long p = pos0 + lane;
        rw_i = d.order ? d.order[p] : p;
        rw_home = d.z[rw_i];
        rw_hcol = rw_home >= 0 ? d.label_of_slot[rw_home] : -1;
        rw_u = d.u[p];
        rw_En = d.gM[GR + lane];
        L.rowhome[lane] = rw_home;
        L.rowhcol[lane] = rw_hcol;
        L.rowM[lane] = d.gM[lane];
    }
    for (int j = tid; j < Kc; j += GRT) {
        const int s = j < d.nslots ? d.perm[j] : -1;
        L.permL[j] = s;
        L.labCol[j] = j;
        L.colLab[j] = j < K0 ? j : -1;
        L.colTix[j] = -1;
        L.colLast[j] = -1;
        L.colBase[j] = j <= K0 ? j : K0;
        L.colLogF[j] = 0.0;
        if (j < K0) {
            const int n = d.n[s];
            L.colSlot[j] = s;
            L.colN[j] = n;
            L.colLogdet[j] = d.sc[s].logdetC;
            L.colIk[j] = 1.0 / (d.k0 + (double)n);
        } else {
            L.colSlot[j] = j == K0 ? d.K_max : -1;
            L.colN[j] = 0;
            L.colLogdet[j] = d.sc[d.K_max].logdetC;
            L.colIk[j] = 1.0 / d.k0;
        }
    }
    __syncthreads();
    GPROF(0);

    int cur = 0;
    // ---- wave 0's registers ------------------------------------------------------------------
    int K = K0, ntix = 0, nterms = 0, nmoves = 0, mapver = 0, ncols = K0 + 1;
    long long lik = 0, last_mover = S.last_mover;
    double ema_run = S.ema_run;
    int cls[LPL], tix[LPL], my_ver = -1;       // column / LDS line of the lane's labels lane * LPL + t
    double pf[LPL];                            // frozen weights of visit pf_row, fetched a visit ahead under the label map pf_ver
    int pf_row = -1, pf_ver = -2;
    // the pre-drawn visit pr_row: cumulative weight through each of the lane's labels, with the columns
    // pr_x0 / pr_x1 (being updated by waves 1 / 2 meanwhile) left out
    double pc[LPL], pr_tot = 0.0;
    int pr_row = -1, pr_ver = -2, pr_x0 = -1, pr_x1 = -1, pr_lab0 = 0x7fffffff, pr_lab1 = 0x7fffffff, pr_tix0 = 0, pr_tix1 = 0;
#pragma unroll
    for (int t = 0; t < LPL; ++t) { cls[t] = 0; tix[t] = -1; pf[t] = 0.0; pc[t] = 0.0; }
    // housekeeping a move leaves for the time the update waves work (lanes 0 / 1: one column each)
    int hk = 0, hk_col = 0, hk_tix = 0, hk_term = 0, hk_prev = 0, hk_n = 0, hk_act = 0, hk_h = -1, hk_hcol = -1, hk_pcol = -1,
        hk_dslot = -1, hk_t0x = 0, hk_t1x = 0, hk_has0 = 0;
    long long hk_i = 0;
    // ---- wave 1's registers: the home side of the next visit, fetched ahead ----------------------
    int hp_base = -1, hp_r = -1, hp_n = -1;
    double hp_crow = 0.0, hp_cd0 = 0.0;
    SlotTab hp_tab = {};
#ifdef BGMM_PROFILE
    long long pq[5] = {0, 0, 0, 0, 0};
#endif

    // Everything of visit r that does not depend on the columns x0 / x1: the lane's cumulative weights.
    // Needs the label cache (cls, tix) and the frozen weights pf of visit r; fetches those of visit r + 1.
    auto pre_draw = [&](int r, int x0, int x1, int tx0, int tx1) {
        double ev_l[LPL];
#pragma unroll
        for (int t = 0; t < LPL; ++t) ev_l[t] = L.etT[(tix[t] < 0 ? 0 : tix[t]) * GR + r];
        const double En = wv_readlane(rw_En, r);
        pr_lab0 = x0 >= 0 ? L.colLab[x0] : 0x7fffffff;
        pr_lab1 = x1 >= 0 ? L.colLab[x1] : 0x7fffffff;
        double pfn[LPL];
        {
            const long long rn = r + 1 < nrows ? r + 1 : r;
#pragma unroll
            for (int t = 0; t < LPL; ++t) pfn[t] = d.ge0[rn * gld + cls[t]];
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
        for (int t = 0; t < LPL; ++t) { pc[t] += off; pf[t] = pfn[t]; }
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
                        cls[t] = j < K ? L.labCol[j] : 0;
                        tix[t] = j < K ? L.colTix[cls[t]] : -1;
                    }
                    my_ver = mapver;
                }
                const int nh = h >= 0 ? L.colN[hcol] : 0;
                const bool home_live = nh >= 2, singleton = nh == 1;
                const int Lr = singleton ? K - 1 : K;                // labels after the removal
                int pick = Lr, pcol = -1;                             // fallback: the last entry (utils.py:20)
                bool bad_tot = false;
                if (!singleton) {
                    if (!(pr_row == r && pr_ver == mapver)) {
                        if (!(pf_row == r && pf_ver == mapver)) {        // (first visit, or the map changed)
#pragma unroll
                            for (int t = 0; t < LPL; ++t) pf[t] = d.ge0[r * gld + cls[t]];
                        }
                        pre_draw(r, -1, -1, 0, 0);
                    }
                    // the two columns the update waves have just rewritten
                    const double a0 = pr_x0 >= 0 ? L.etT[pr_tix0 * GR + r] : 0.0;
                    const double a1 = pr_x1 >= 0 ? L.etT[pr_tix1 * GR + r] : 0.0;
                    const double tot = pr_tot + (a0 + a1);
                    bad_tot = !(tot > 1e-200 && tot < 1e200);            // M_r went stale: a fresh window
                    const double ut = u * tot;
                    int hit = 0x7fffffff, hitcol = -1;
#pragma unroll
                    for (int t = LPL - 1; t >= 0; --t) {
                        // (descending: the first label whose cumulative weight exceeds u wins)
                        const int j = lane * LPL + t;
                        const double ct = pc[t] + ((j >= pr_lab0 ? a0 : 0.0) + (j >= pr_lab1 ? a1 : 0.0));
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
                    const int lab_h = L.colLab[hcol];
                    const double En = wv_readlane(rw_En, r);
                    double e[LPL];
                    int ecol[LPL];
#pragma unroll 1
                    for (int t = 0; t < LPL; ++t) {
                        const int j = lane * LPL + t;
                        double v = 0.0;
                        int cl = 0;
                        if (j < Lr) {
                            cl = L.labCol[j == lab_h ? K - 1 : j];
                            const int tx = L.colTix[cl];
                            v = tx >= 0 ? L.etT[tx * GR + r] : d.ge0[r * gld + cl];
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
                    pr_row = -1;
                }
                GPROF(1);
                if (bad_tot) { ev = GEV_CUT; break; }
                const bool stay = home_live && pick < Lr && pcol == hcol;
                if (stay) { lik += K; continue; }
                if (nterms + 2 > T) { ev = GEV_CUT; break; }
                ev = GEV_MOVE;
                const long long i_mv = wv_readlane_i64(rw_i, r);
                const long long p = pos0 + r;
                ema_run = ema_after_mover(ema_run, (double)(p - last_mover));
                last_mover = p;
                if (!singleton && pick < K) {
                    // ---- the usual move: lane 0 = the column x leaves, lane 1 = the column it joins.  Before
                    // the barrier only what the update waves need; the rest while they work. ----
                    const bool has0 = h >= 0;
                    const int mycol = (lane == 0 && has0) ? hcol : pcol;
                    const int ctix = L.colTix[mycol], clast = L.colLast[mycol], cbase = L.colBase[mycol];
                    const int cn = L.colN[mycol], cslot = L.colSlot[mycol];
                    const double cik = L.colIk[mycol], cld = L.colLogdet[mycol], cf = L.colLogF[mycol];
                    const int sg = lane == 0 ? -1 : 1;
                    const int need = ctix < 0 ? 1 : 0;
                    const int need0 = has0 ? __builtin_amdgcn_readlane(need, 0) : 0;
                    const int need1 = __builtin_amdgcn_readlane(need, 1);
                    const int my_tix = ctix >= 0 ? ctix : (lane == 0 ? ntix : ntix + need0);
                    const int my_term = lane == 0 ? nterms : nterms + (has0 ? 1 : 0);
                    const int my_k = lane == 0 ? 0 : (has0 ? 1 : 0);
                    const bool actl = (lane == 0 && has0) || lane == 1;
                    if (actl) {
                        S.upd[my_k].col = mycol; S.upd[my_k].sigma = sg; S.upd[my_k].term = my_term; S.upd[my_k].tix = my_tix;
                        S.upd[my_k].base = cbase; S.upd[my_k].slot = cslot; S.upd[my_k].n_new = cn + sg; S.upd[my_k].prev = clast;
                        S.upd[my_k].ik = cik; S.upd[my_k].logdet0 = cld; S.upd[my_k].logf = cf;
                    }
                    if (lane == 0) S.upd_n = has0 ? 2 : 1;
                    hk = 1; hk_col = mycol; hk_tix = my_tix; hk_term = my_term; hk_prev = clast; hk_n = cn + sg; hk_act = actl ? 1 : 0;
                    hk_h = h; hk_hcol = hcol; hk_pcol = pcol; hk_i = i_mv; hk_has0 = has0 ? 1 : 0;
                    hk_dslot = __builtin_amdgcn_readlane(cslot, 1);
                    hk_t0x = __builtin_amdgcn_readlane(my_tix, 0); hk_t1x = __builtin_amdgcn_readlane(my_tix, 1);
                    ntix += need0 + need1;
                    nterms += has0 ? 2 : 1;
                    lik += K;
                } else {
                    // ---- (rare) a component is deleted and / or opened: lane 0, step by step ----
                    int Kn = K, Krem = K, nu = 0, sub_slot = -1, add_slot = -1, add_init = 0, nt = nterms, nx = ntix, nc = ncols, err = 0;
                    if (lane == 0) {
                        if (h >= 0) {
                            const int n1 = nh - 1;
                            L.colN[hcol] = n1;
                            if (n1 > 0) {
                                sub_slot = h;
                                S.upd[nu].col = hcol; S.upd[nu].sigma = -1; S.upd[nu].slot = h; S.upd[nu].n_new = n1; ++nu;
                            } else {                    // swap-with-last delete of its label (gaussian_components.py:188-205)
                                const int lab = L.colLab[hcol], last = Kn - 1;
                                const int c_last = L.labCol[last], s_last = L.permL[last];
                                L.labCol[lab] = c_last; L.permL[lab] = s_last; L.colLab[c_last] = lab;
                                L.labCol[last] = hcol; L.permL[last] = h; L.colLab[hcol] = -1;
                                L.colSlot[hcol] = -1;       // retired: the slot may come back in a new column
                                d.perm[lab] = s_last; d.label_of_slot[s_last] = lab;
                                d.perm[last] = h; d.label_of_slot[h] = last;
                                d.n[h] = 0;
                                Kn = last;
                            }
                        }
                        Krem = Kn;                       // labels the draw chose among
                        int dcol = -1;
                        if (pick >= Kn) {                // a new component
                            if (Kn >= d.K_max || nc >= Kc) {
                                err = -3;
                            } else {
                                const int t = L.permL[Kn];
                                dcol = nc++;
                                L.colSlot[dcol] = t; L.colN[dcol] = 0; L.colBase[dcol] = cprior;
                                L.colLogdet[dcol] = L.colLogdet[cprior]; L.colIk[dcol] = L.colIk[cprior]; L.colLogF[dcol] = 0.0;
                                L.colTix[dcol] = -1; L.colLast[dcol] = -1;
                                L.colLab[dcol] = Kn; L.labCol[Kn] = dcol;
                                d.label_of_slot[t] = Kn;
                                d.nupd[t] = 0;
                                add_init = 1;
                                Kn += 1;
                            }
                        } else {
                            dcol = pcol;
                        }
                        if (dcol >= 0) {
                            const int nn = L.colN[dcol] + 1;
                            L.colN[dcol] = nn;
                            add_slot = L.colSlot[dcol];
                            S.upd[nu].col = dcol; S.upd[nu].sigma = 1; S.upd[nu].slot = add_slot; S.upd[nu].n_new = nn; ++nu;
                        }
                        for (int k = 0; k < nu; ++k) {
                            const int cl = S.upd[k].col;
                            int tx = L.colTix[cl];
                            if (tx < 0) { tx = nx++; L.colTix[cl] = tx; }
                            const int t = nt++;
                            const int prev = L.colLast[cl];
                            S.upd[k].tix = tx; S.upd[k].term = t; S.upd[k].prev = prev;
                            S.upd[k].base = L.colBase[cl]; S.upd[k].ik = L.colIk[cl]; S.upd[k].logdet0 = L.colLogdet[cl];
                            S.upd[k].logf = L.colLogF[cl];
                            L.termPrev[t] = prev;
                            L.colLast[cl] = t;
                        }
                        S.upd_n = nu;
                        if (err) S.err = err;
                        L.mvI[nmoves] = i_mv; L.mvSub[nmoves] = sub_slot; L.mvAdd[nmoves] = add_slot; L.mvInit[nmoves] = add_init;
                    }
                    K = __builtin_amdgcn_readlane(Kn, 0);
                    nterms = __builtin_amdgcn_readlane(nt, 0);
                    ntix = __builtin_amdgcn_readlane(nx, 0);
                    ncols = __builtin_amdgcn_readlane(nc, 0);
                    nmoves += 1;
                    lik += __builtin_amdgcn_readlane(Krem, 0);
                    mapver += 1;                    // (labels and lines are re-read from LDS at the next draw)
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
                // ---- what the move leaves to do, while the update waves work ----------------------
                if (hk_act) {
                    L.colN[hk_col] = hk_n;
                    L.colTix[hk_col] = hk_tix;
                    L.colLast[hk_col] = hk_term;
                    L.termPrev[hk_term] = hk_prev;
                }
#pragma unroll
                for (int t = 0; t < LPL; ++t) {
                    tix[t] = (hk_has0 && cls[t] == hk_hcol) ? hk_t0x : tix[t];
                    tix[t] = cls[t] == hk_pcol ? hk_t1x : tix[t];
                }
                if (lane == 0) {
                    L.mvI[nmoves] = hk_i; L.mvSub[nmoves] = hk_has0 ? hk_h : -1; L.mvAdd[nmoves] = hk_dslot; L.mvInit[nmoves] = 0;
                }
                nmoves += 1;
                // the next visit, without the two columns in the making
                const int rn = cur + 1;
                if (rn < nrows && pf_row == rn && pf_ver == mapver) {
                    const int hn = __builtin_amdgcn_readlane(rw_home, rn), hcn = __builtin_amdgcn_readlane(rw_hcol, rn);
                    const int nhn = hn >= 0 ? L.colN[hcn] : 0;
                    if (nhn != 1) pre_draw(rn, hk_has0 ? hk_hcol : -1, hk_pcol, hk_t0x, hk_t1x);
                }
            }
            GPROF(5);
        } else if ((wave == 1 || wave == 2) && wave - 1 < S.upd_n) {
            // ---- one touched column: the new term and the column's weights for the later rows ----
            const int k = wave - 1;
            const int cl = S.upd[k].col, sg = S.upd[k].sigma, t = S.upd[k].term, tix_c = S.upd[k].tix;
            const int base = S.upd[k].base, slot = S.upd[k].slot, prev0 = S.upd[k].prev;
            const int n_new = __builtin_amdgcn_readfirstlane(S.upd[k].n_new);
            const double ik = S.upd[k].ik, logdet0 = S.upd[k].logdet0, logf_old = S.upd[k].logf;
            const int r = cur;
            const bool act = lane >= r && lane < nrows;
            const int lrow = act ? lane : r;
            double crow, cd0;
            if (wave == 1 && hp_base == base && hp_r == r) {
                crow = hp_crow; cd0 = hp_cd0;
            } else {
                crow = d.gC[((long long)base * GR + r) * GR + lrow];
                cd0 = d.gq0[(long long)base * GR + lrow];
            }
            const SlotTab tab = (wave == 1 && hp_n == n_new) ? hp_tab : load_slot_tab(d, n_new);
#ifdef BGMM_PROFILE
            if (k == 1 && lane == 0) { tk2 = clock64(); pq[0] += tk2 - tk; }
#endif
            const double rowM = L.rowM[lrow];
            const bool own = L.rowhome[lrow] == slot && n_new >= 2;
            // what depends on the count alone (Student-t constants of gaussian_components.py:228-251)
            const double Dd = (double)d.D;
            const double kN_new = d.k0 + (double)n_new;
            const long long v = d.v0 + n_new - d.D + 1;
            const double ikn = fm_div(1.0, kN_new);
            const double inv_cv = fm_div(kN_new, kN_new + 1.0);
            const double hv = 0.5 * (double)(v + d.D);
            const double log_ratio = fm_log(1.0 - (double)sg * ikn);          // log(k_N before / k_N now)
            const double cb = tab.seat + (tab.g - 0.5 * (Dd * tab.lc + logdet0)) - 0.5 * (logf_old + log_ratio) - rowM;
            double acc = crow + ik, cdv = cd0 + ik;
#ifdef BGMM_PROFILE
            if (k == 1) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (lane == 0) { tk2 = clock64(); pq[1] += tk2 - tk; } }
#endif
            for (int tt = prev0; tt >= 0; tt = L.termPrev[tt]) {
                const double wr = L.wv[tt * GR + r], wl = L.wv[tt * GR + lrow], id = L.termInvD[tt];
                acc = fma(-(wl * wr), id, acc);
                cdv = fma(-(wl * wl), id, cdv);
            }
            // c_t(y, y) = cdv - acc^2 / Dt;  1 + q inv_cv = (Dt (1 + (cdv - 1/k_N) inv_cv) - acc^2 inv_cv) / Dt
            const double wrr = wv_readlane(acc, r);
            const double Dt = (double)sg + wrr;
            const double aD = fabs(Dt);                                 // sigma Dt = 1 + sigma c_{t-1}(x, x) > 0
            const double base1 = fma(cdv - ikn, inv_cv, 1.0);
            const double numer = fabs(fma(Dt, base1, -(acc * acc) * inv_cv));
#ifdef BGMM_PROFILE
            if (k == 1 && lane == 0) { tk2 = clock64(); pq[2] += tk2 - tk; }
#endif
            const double lD = fm_log(aD), lN = fm_log(numer);
#ifdef BGMM_PROFILE
            if (k == 1 && lane == 0 && lN != 12345.0) { tk2 = clock64(); pq[3] += tk2 - tk; }
#endif
            double ee = fm_exp(cb + ((hv - 0.5) * lD - hv * lN));
#ifdef BGMM_PROFILE
            if (k == 1 && lane == 0 && ee != 12345.0) { tk2 = clock64(); pq[4] += tk2 - tk; }
#endif
            const double invD = fm_div(1.0, Dt);
            const unsigned long long mown = __ballot(own && act && lane > r);
            if (mown) {
                // (rare) rows whose home this column is: the visited point removed from it (slot_math.h, home form)
                const double ct = fma(-(acc * acc), invD, cdv);
                const double qv = ct - ikn;
                const double a1 = fm_div(kN_new, kN_new - 1.0);
                const double den = 1.0 - a1 * qv;
                const double hv1 = 0.5 * (double)(v - 1 + d.D);
                const double cb1 = tab.seat1 + (tab.g1 - 0.5 * (Dd * tab.lc1 + logdet0)) - 0.5 * (logf_old + log_ratio + lD) - rowM;
                const double e1 = fm_exp(cb1 - 0.5 * fm_log(den) - hv1 * fm_log(1.0 + fm_div(a1 * qv, den)));
                ee = own ? e1 : ee;
            }
            if (act && lane > r) L.etT[tix_c * GR + lane] = ee;
            L.wv[t * GR + lane] = act ? acc : 0.0;
            if (lane == 0) {
                L.termInvD[t] = invD;
                L.colLogF[cl] = logf_old + log_ratio + lD;
                if (!((double)sg * Dt > 0.0) || !(lD == lD)) S.err = -4;
            }
            if (wave == 1 && r + 1 < nrows) {
                // the home side of the next visit (its column is known; wasted if that visit stays; the
                // count is checked against the published one when it is used)
                const int hc = L.rowhcol[r + 1];
                if (hc >= 0) {
                    const int hb = L.colBase[hc];
                    hp_n = __builtin_amdgcn_readfirstlane(L.colN[hc]) - 1;
                    hp_base = hb; hp_r = r + 1;
                    const int l2 = (lane >= r + 1 && lane < nrows) ? lane : r + 1;
                    hp_crow = d.gC[((long long)hb * GR + r + 1) * GR + l2];
                    hp_cd0 = d.gq0[(long long)hb * GR + l2];
                    if (hp_n >= 1) hp_tab = load_slot_tab(d, hp_n); else hp_n = -1;
                }
            }
#ifdef BGMM_PROFILE
            if (lane == 0) { tk2 = clock64(); S.prof[3 + k] += tk2 - tk; }
#endif
        }
        gram_lds_barrier();
        if (S.err < 0) break;
        cur += 1;
#ifdef BGMM_PROFILE
        tk = clock64();
#endif
    }
#ifdef BGMM_PROFILE
    if (wave == 2 && lane == 0) for (int k = 0; k < 5; ++k) c->prof[9 + k] += pq[k];
#endif
    if (wave == 0 && lane == 0) {
        S.K = K; S.nmoves = nmoves; S.lik = lik; S.ema_run = ema_run; S.last_mover = last_mover; S.ncols = ncols;
    }
    __syncthreads();

    // ---- close the window ------------------------------------------------------------------
    const int consumed = S.event == GEV_MOVE ? S.cur + 1 : S.cur;     // (an error stops behind the visit that raised it)
    for (int k = tid; k < S.nmoves; k += GRT) {
        GramMove mv;
        mv.i = L.mvI[k]; mv.sub_slot = L.mvSub[k]; mv.add_slot = L.mvAdd[k]; mv.add_init = L.mvInit[k]; mv.pad = 0;
        d.gmoves[k] = mv;
        d.z[mv.i] = mv.add_slot;
    }
    // counts of every column that changed; the live ones go on the finish kernel's list
    for (int cl = tid; cl < S.ncols; cl += GRT) {
        if (cl == cprior || L.colTix[cl] < 0 || L.colSlot[cl] < 0) continue;
        const int s = L.colSlot[cl], n = L.colN[cl];
        d.n[s] = n;
        if (n > 0) d.gtouched[atomicAdd(&c->gram_ntouched, 1)] = s;
    }
    if (tid == 0) {
        const Job &j = c->job;
        c->n_steps += 1;
        c->n_score_launches += 1;
        c->gram_windows += 1;
        c->gram_rows_total += consumed;
        c->n_scored += (long long)nrows * (j.K + 1);
        c->n_pairs_exact += (unsigned long long)nrows * (unsigned long long)(j.K + 1);
        c->lik_evals += S.lik;
        c->n_moves += S.nmoves;
        c->gram_nmoves = S.nmoves;
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
#endif
        if (S.err < 0) {
            atomicCAS(&c->error, 0, S.err);
            c->job.mode = MODE_DONE;
        } else {
            c->win_size = (int)window_for_rate(c);
            start_window(d, c, pos0 + consumed);
        }
    }
}

template <int NJ>
static void launch_gram_t(const Dev &d, hipStream_t st) {
    const int lds = GR * (16 * NJ + 2) * (int)sizeof(double);
    hipLaunchKernelGGL((gram_kernel<NJ>), dim3(d.gcols), dim3(256), lds, st, d);
}

template <int NJ>
static void configure_gram_t() {
    const int lds = GR * (16 * NJ + 2) * (int)sizeof(double);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute((const void *)gram_kernel<NJ>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

// Kernel attributes are per device: set for the device of the calling context (current device).
void gram_configure(const Dev &d, int resolve_lds) {
    configure_gram_t<5>(); configure_gram_t<6>(); configure_gram_t<7>(); configure_gram_t<8>();
    (void)hipFuncSetAttribute((const void *)gram_resolve_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, resolve_lds);
    (void)hipFuncSetAttribute((const void *)gram_resolve_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, resolve_lds);
    (void)hipFuncSetAttribute((const void *)gram_resolve_kernel<6>, hipFuncAttributeMaxDynamicSharedMemorySize, resolve_lds);
    (void)hipFuncSetAttribute((const void *)gram_resolve_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, resolve_lds);
    (void)d;
}

// One frozen-factor step: cross forms, weights, the sequential walk, statistics + factors of the touched
// slots.  ev0 / ev1 (optional) bracket the likelihood kernel.
bool launch_gram_step(const Dev &d, int resolve_lds, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1) {
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
    if (reach <= 128) hipLaunchKernelGGL(gram_resolve_kernel<2>, dim3(1), dim3(GRT), resolve_lds, st, d, d.gram_terms);
    else if (reach <= 256) hipLaunchKernelGGL(gram_resolve_kernel<4>, dim3(1), dim3(GRT), resolve_lds, st, d, d.gram_terms);
    else if (reach <= 384) hipLaunchKernelGGL(gram_resolve_kernel<6>, dim3(1), dim3(GRT), resolve_lds, st, d, d.gram_terms);
    else hipLaunchKernelGGL(gram_resolve_kernel<8>, dim3(1), dim3(GRT), resolve_lds, st, d, d.gram_terms);
    launch_gram_finish(d, st);
    return true;
}

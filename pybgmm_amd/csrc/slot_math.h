// Per-slot derived-state math shared by the state kernels and the in-launch mover resolver.
// Everything here is written for a thread team of NT threads with team-local thread id `tid`
// (a whole block, or one half of the resolver block); functions that contain barriers must be
// called by ALL threads of the block (teams that have no work pass active = false).
//
// Reference behaviour restated: gaussian_components.py:319-331 (covariance refresh) and
// :228-251 (the constants of the Student-t predictive it feeds).
#pragma once
#include "bgmm_device.h"

__device__ __forceinline__ double student_const(const Dev &d, long long v) {
    const double hd = 0.5 * (double)d.D;
    return d.tab_lgam[v + d.D] - d.tab_lgam[v] - hd * d.tab_log[v] - hd * BGMM_LOG_PI;
}

__device__ __forceinline__ double seat_weight(const Dev &d, int n) {
    if (n <= 0) return 0.0;
    return d.use_power ? log(pow((double)n, d.power)) : log((double)n);
}

// scale of the predictive covariance: Sigma = cs * S_N with cs = (k_N+1)/(k_N (v_N-D+1))
__device__ __forceinline__ double cov_scale(const Dev &d, int n) {
    const double k_N = d.k0 + (double)n;
    return (k_N + 1.0) / (k_N * (double)(d.v0 + n - d.D + 1));
}

// constants of the predictive for a slot with count n and logdet(S_N) = logdetC.  Everything
// that depends on n alone comes from the device-built tables (no transcendental on this path:
// it sits on the critical chain of every move).
struct SlotTab { double g, lc, seat, g1, lc1, seat1; };

__device__ __forceinline__ SlotTab load_slot_tab(const Dev &d, int n) {
    SlotTab t;
    const long long v = d.v0 + n - d.D + 1;
    t.g = d.tabG[v]; t.lc = d.tabLogC[n]; t.seat = d.tabSeat[n];
    t.g1 = n >= 2 ? d.tabG[v - 1] : 0.0;
    t.lc1 = n >= 1 ? d.tabLogC[n - 1] : 0.0;
    t.seat1 = n >= 1 ? d.tabSeat[n - 1] : 0.0;
    return t;
}

__device__ inline SlotConst make_consts_from(const Dev &d, int n, const SlotTab &t, double logdetC, double lam,
                                             double mu2) {
    SlotConst c;
    const int D = d.D;
    const double Dd = (double)D;
    const double k_N = d.k0 + (double)n;
    const long long v = d.v0 + n - D + 1;
    const double g = t.g, lc = t.lc, seat = t.seat;
    const double g1 = t.g1, lc1 = t.lc1;
    const double seat1 = t.seat1;
    const double cs = (k_N + 1.0) / (k_N * (double)v);
    c.logdetC = logdetC;
    c.A = g - 0.5 * (Dd * lc + logdetC);
    c.half_vd = 0.5 * (double)(v + D);
    c.inv_cv = 1.0 / (cs * (double)v);
    c.logseat = seat;
    c.logseat1 = seat1;
    c.A1 = 0.0; c.half_vd1 = 0.0; c.coef1 = 0.0; c.a1 = 0.0;
    if (n >= 2) {
        const double k1 = k_N - 1.0;
        const long long v1 = v - 1;
        const double c1 = k_N / (k1 * (double)v1);
        const double a = k_N / k1;
        c.a1 = a;
        c.A1 = g1 - 0.5 * (Dd * lc1 + logdetC);
        c.half_vd1 = 0.5 * (double)(v1 + D);
        c.coef1 = a * a / (c1 * (double)v1);
    }
    c.inv_lam = (lam > 0.0 && lam < 1e300) ? 1.0 / lam : 0.0;
    c.mu2 = mu2;
    return c;
}

__device__ inline SlotConst make_consts(const Dev &d, int n, double logdetC, double lam, double mu2) {
    return make_consts_from(d, n, load_slot_tab(d, n), logdetC, lam, mu2);
}

// log score of one (visit, slot) pair from its quadratic form (seating weight included)
__device__ __forceinline__ double slot_log_score(const SlotConst &sc, double qv, bool home_minus_one) {
    if (home_minus_one) {
        const double den = 1.0 - sc.a1 * qv;
        return sc.logseat1 + sc.A1 - 0.5 * log(den) - sc.half_vd1 * log(1.0 + sc.coef1 * qv / den);
    }
    return sc.logseat + sc.A - sc.half_vd * log(1.0 + qv * sc.inv_cv);
}

// Write a slot's derived state from Winv (LDS, lower triangle, leading dimension ld) and mu
// (LDS).  cv_lds (optional) receives cvec as well.  No barrier inside; the caller's data must
// be complete (barrier before), and cv_lds is valid after the caller's next barrier.
template <int NT>
__device__ inline void write_slot(const Dev &d, int s, const double *W, int ld, const double *mu,
                                  double logdetC, double lam, int tid, double *cv_lds, bool active) {
    if (!active) return;
    const int D = d.D, Dp = d.Dp;
    for (int j = tid; j < Dp; j += NT) {
        double acc = 0.0;
        if (j < D)
            for (int l = 0; l <= j; ++l) acc = fma(W[j * ld + l], mu[l], acc);
        d.cvec[(long long)s * Dp + j] = acc;
        if (cv_lds && j < D) cv_lds[j] = acc;
    }
    for (int e = tid; e < D * D; e += NT) {
        const int a = e / D, b = e - a * D;
        d.Wrm[(long long)s * D * D + e] = (b <= a) ? W[a * ld + b] : 0.0;
    }
    double *wf = d.Wfrag + (long long)s * d.nfrag * 64;
    for (int e = tid; e < d.nfrag * 64; e += NT) {
        const int f = e >> 6, lane = e & 63;
        int J = 0;
        while (2 * (J + 1) * (J + 2) <= f) ++J;
        const int kk = f - 2 * J * (J + 1);
        const int j = 16 * J + (lane & 15), l = 4 * kk + (lane >> 4);
        wf[e] = (j < D && l <= j) ? -W[j * ld + l] : 0.0;
    }
    for (int l = tid; l < D; l += NT) d.mu[(long long)s * D + l] = mu[l];
    if (tid == 0) {
        double mu2 = 0.0;
        for (int l = 0; l < D; ++l) mu2 = fma(mu[l], mu[l], mu2);
        d.sc[s] = make_consts(d, d.n[s], logdetC, lam, mu2);
        d.mu_ver[s] += 1;                     // what is cached per point against this slot's state is stale now
    }
}

// Gershgorin bound of lambda_max of the symmetric matrix whose lower triangle is in A (LDS):
// max_i sum_j |A_ij|.  One barrier; result in *out (LDS) for every thread after it.
template <int NT>
__device__ inline void gershgorin_bound(const double *A, int ld, int D, double *rowbuf, double *out,
                                        int tid, bool active) {
    // 4 threads per row (a quad of lanes), partial sums folded with two shuffles
    if (active)
        for (int i0 = 0; i0 < D; i0 += NT / 4) {
            const int i = i0 + (tid >> 2), part = tid & 3;
            double sacc = 0.0;
            if (i < D)
                for (int j = part; j < D; j += 4) sacc += fabs(j <= i ? A[i * ld + j] : A[j * ld + i]);
            sacc += __shfl_xor(sacc, 1);
            sacc += __shfl_xor(sacc, 2);
            if (i < D && part == 0) rowbuf[i] = sacc;
        }
    __syncthreads();
    if (active && tid == 0) {
        double mx = 0.0;
        for (int i = 0; i < D; ++i) mx = fmax(mx, rowbuf[i]);
        *out = mx;
    }
}

// Lambda after a rank-1 change S_N' = S_N + a dd':  lambda_max grows by at most a |d|^2 (a > 0)
// and cannot grow when a < 0.
__device__ __forceinline__ double lam_after_rank1(double inv_lam_old, double a, double d2) {
    if (!(inv_lam_old > 0.0)) return 0.0;                 // unknown stays unknown
    const double lam = 1.0 / inv_lam_old;
    return a > 0.0 ? lam + a * d2 : lam;
}

// In-place Cholesky (right looking) followed by the in-place inverse of the factor.
// A: LDS, lower triangle of S_N on entry, Winv on exit.  rowbuf: D doubles of LDS scratch.
// Returns (to every thread that reads *logdet_out after the final barrier) logdet S_N.
// Contains barriers: all NT threads of the team's BLOCK must call it (block-wide barriers), with
// active = false for teams that have nothing to do.
template <int NT>
__device__ inline void chol_inverse(double *A, int ld, int D, double *rowbuf, double *logdet_out,
                                    int *bad_out, int tid, bool active) {
    const int tx = tid & 15, ty = tid >> 4;
    bool bad = false;
    for (int j = 0; j < D; ++j) {
        __syncthreads();
        if (active) {
            const double djj = A[j * ld + j];
            if (!(djj > 0.0)) bad = true;
            const double piv = sqrt(djj);
            for (int i = j + 1 + tid; i < D; i += NT) A[i * ld + j] = A[i * ld + j] / piv;
        }
        __syncthreads();
        if (active) {
            if (tid == 0) A[j * ld + j] = sqrt(A[j * ld + j]);
            for (int i = j + 1 + ty; i < D; i += NT / 16) {
                const double lij = A[i * ld + j];
                for (int l = j + 1 + tx; l <= i; l += 16) A[i * ld + l] = fma(-lij, A[l * ld + j], A[i * ld + l]);
            }
        }
    }
    __syncthreads();
    if (active && tid == 0) {
        double ldt = 0.0;
        for (int j = 0; j < D; ++j) ldt += log(A[j * ld + j]);
        ldt *= 2.0;
        *logdet_out = ldt;
        *bad_out = (bad || !(ldt == ldt)) ? 1 : 0;
    }
    // inverse, row by row: Winv[i][c] = -(sum_{t=c}^{i-1} L[i][t] Winv[t][c]) / L[i][i]
    int P = 1;
    while (P < D) P <<= 1;
    int tpc = NT / P;
    if (tpc < 1) tpc = 1;
    if (tpc > 64) tpc = 64;
    const int col = tid / tpc, part = tid % tpc;
    for (int i = 0; i < D; ++i) {
        if (active)
            for (int t = tid; t <= i; t += NT) rowbuf[t] = A[i * ld + t];
        __syncthreads();
        if (active) {
            const double inv_d = 1.0 / rowbuf[i];
            for (int c0 = 0; c0 <= i; c0 += NT / tpc) {
                const int c = c0 + col;
                double acc = 0.0;
                if (c < i)
                    for (int t = c + part; t < i; t += tpc) acc = fma(rowbuf[t], A[t * ld + c], acc);
                for (int o = tpc >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
                if (part == 0) {
                    if (c < i) A[i * ld + c] = -acc * inv_d;
                    else if (c == i) A[i * ld + i] = inv_d;
                }
            }
        }
        __syncthreads();
    }
}

// Rank-1 change of the inverse factor held in LDS:
//   S_N' = S_N + a dd',  p = Winv d,  P_i = sum_{k<=i} p_k^2,
//   Winv' = T Winv,  T_ii = l_i = sqrt((1 + a P_{i-1})/(1 + a P_i)),  T_ik = t_i p_k (k < i),
//   t_i = -a p_i / ((1 + a P_i) l_i),   logdet' = logdet + log(1 + a P_{D-1}).
// W, dv (= x - mu_old), pv, lv, tv: LDS.  *s_out receives P_{D-1}; *bad_out is set on loss of
// positive definiteness.  Barriers inside (block wide).
template <int NT>
__device__ inline void rank1_inverse_factor(double *W, int ld, int D, double a, const double *dv,
                                            double *pv, double *lv, double *tv, double *s_out,
                                            int *bad_out, int tid, bool active) {
    // p = W d : 8 threads per row
    if (active) {
        for (int r0 = 0; r0 < D; r0 += NT / 8) {
            const int r = r0 + (tid >> 3), part = tid & 7;
            double acc = 0.0;
            if (r < D)
                for (int l = part; l <= r; l += 8) acc = fma(W[r * ld + l], dv[l], acc);
            acc += __shfl_xor(acc, 1);
            acc += __shfl_xor(acc, 2);
            acc += __shfl_xor(acc, 4);
            if (r < D && part == 0) pv[r] = acc;
        }
    }
    __syncthreads();
    if (active && D <= 64) {
        // prefix sums of p^2 by the first wavefront of the team
        if (tid < 64) {
            const double v = tid < D ? pv[tid] * pv[tid] : 0.0;
            double P = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const double t = __shfl_up(P, o);
                if (tid >= o) P += t;
            }
            double Pm1 = __shfl_up(P, 1);
            if (tid == 0) Pm1 = 0.0;
            const double num = 1.0 + a * Pm1, den = 1.0 + a * P;
            const bool badl = tid < D && (!(den > 0.0) || !(num > 0.0));
            if (tid < D) {
                const double l = sqrt(num / den);
                lv[tid] = l;
                tv[tid] = -a * pv[tid] / (den * l);
            }
            const unsigned long long anybad = __ballot(badl);
            if (tid == D - 1) { *s_out = P; *bad_out = anybad ? 1 : 0; }
        }
    } else if (active && tid == 0) {
        double P = 0.0;
        bool bad = false;
        for (int r = 0; r < D; ++r) {
            const double num = 1.0 + a * P;
            P = fma(pv[r], pv[r], P);
            const double den = 1.0 + a * P;
            if (!(den > 0.0) || !(num > 0.0)) bad = true;
            const double l = sqrt(num / den);
            lv[r] = l;
            tv[r] = -a * pv[r] / (den * l);
        }
        *s_out = P;
        *bad_out = bad ? 1 : 0;
    }
    __syncthreads();
    if (active) {
        for (int j = tid; j < D; j += NT) {        // column j: running sum_{k<i} p_k W[k][j]
            double r = 0.0;
            for (int row = j; row < D; ++row) {
                const double w = W[row * ld + j];
                W[row * ld + j] = fma(tv[row], r, lv[row] * w);
                r = fma(pv[row], w, r);
            }
        }
    }
    __syncthreads();
}

// The same step for a whole block of 256 threads with every dependent loop cut short (gram_finish_kernel: the kernel lasts
// as long as its slowest workgroup, and the plain routine's column sweep is 64 .. 128 dependent LDS round trips on a
// quarter of the block):
//   p = W d          four threads per row, partial sums folded with two shuffles;
//   P, l, t          prefix sums of p^2 through LDS (log2 D doubling steps) -- any D <= 256;
//   W' = T W         column j by FOUR threads, a quarter of the rows each: first the quarter's sum of p_k W[k][j], then
//                    the sweep over its own rows from the sums of the quarters above, every load of a quarter set off together.
// scan: 256 doubles of LDS scratch.  Barriers inside (block wide); all 256 threads call it.
__device__ inline void rank1_inverse_factor_wide(double *W, int ld, int D, double a, const double *dv,
                                                 double *pv, double *lv, double *tv, double *scan, double *s_out,
                                                 int *bad_out, int tid) {
    constexpr int NT = 256;
    for (int r0 = 0; r0 < D; r0 += NT / 4) {
        const int r = r0 + (tid >> 2), part = tid & 3;
        double acc = 0.0;
        if (r < D) {
            const double *__restrict__ wr = W + r * ld;
#pragma unroll 4
            for (int l = part; l <= r; l += 4) acc = fma(wr[l], dv[l], acc);
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (r < D && part == 0) pv[r] = acc;
    }
    __syncthreads();
    {
        const double mine = tid < D ? pv[tid] * pv[tid] : 0.0;
        scan[tid] = mine;
        for (int o = 1; o < D; o <<= 1) {
            __syncthreads();
            const double t = tid >= o ? scan[tid - o] : 0.0;
            __syncthreads();
            scan[tid] += t;
        }
        __syncthreads();
        if (tid < D) {
            const double P = scan[tid], Pm1 = tid > 0 ? scan[tid - 1] : 0.0;
            const double num = 1.0 + a * Pm1, den = 1.0 + a * P;
            const bool badl = !(den > 0.0) || !(num > 0.0);
            const double l = sqrt(num / den);
            lv[tid] = l;
            tv[tid] = -a * pv[tid] / (den * l);
            if (badl) *bad_out = 1;
            if (tid == D - 1) *s_out = P;
        }
    }
    __syncthreads();
    // column sweep: thread (j, qr) owns rows [qr * seg, qr * seg + seg) of column j
    const int seg = (D + 3) >> 2;
    double *part_sum = scan;                         // [4][64] per pass of 64 columns
    for (int j0 = 0; j0 < D; j0 += 64) {
        const int j = j0 + (tid & 63), qr = tid >> 6;
        const int lo = qr * seg, hi = lo + seg < D ? lo + seg : D;
        double sacc = 0.0;
        if (j < D) {
#pragma unroll 8
            for (int row = lo > j ? lo : j; row < hi; ++row) sacc = fma(pv[row], W[row * ld + j], sacc);
        }
        part_sum[qr * 64 + (tid & 63)] = sacc;
        __syncthreads();
        if (j < D) {
            double r = 0.0;
            for (int q2 = 0; q2 < qr; ++q2) r += part_sum[q2 * 64 + (tid & 63)];
#pragma unroll 8
            for (int row = lo > j ? lo : j; row < hi; ++row) {
                const double w = W[row * ld + j];
                W[row * ld + j] = fma(tv[row], r, lv[row] * w);
                r = fma(pv[row], w, r);
            }
        }
        __syncthreads();
    }
}

// write_slot for a block of 256 threads with the dependent loops cut short (cvec = Winv mu by four threads per row, |mu|^2
// by a wavefront).  Same outputs.  Barrier-free; red: 8 doubles of LDS that the caller does not touch meanwhile.
__device__ inline void write_slot_wide(const Dev &d, int s, const double *W, int ld, const double *mu,
                                       double logdetC, double lam, int tid) {
    constexpr int NT = 256;
    const int D = d.D, Dp = d.Dp;
    for (int j0 = 0; j0 < Dp; j0 += NT / 4) {
        const int j = j0 + (tid >> 2), part = tid & 3;
        double acc = 0.0;
        if (j < D) {
            const double *__restrict__ wr = W + j * ld;
#pragma unroll 4
            for (int l = part; l <= j; l += 4) acc = fma(wr[l], mu[l], acc);
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (j < Dp && part == 0) d.cvec[(long long)s * Dp + j] = acc;
    }
    for (int e = tid; e < D * D; e += NT) {
        const int a = e / D, b = e - a * D;
        d.Wrm[(long long)s * D * D + e] = (b <= a) ? W[a * ld + b] : 0.0;
    }
    double *wf = d.Wfrag + (long long)s * d.nfrag * 64;
    for (int e = tid; e < d.nfrag * 64; e += NT) {
        const int f = e >> 6, lane = e & 63;
        int J = 0;
        while (2 * (J + 1) * (J + 2) <= f) ++J;
        const int kk = f - 2 * J * (J + 1);
        const int j = 16 * J + (lane & 15), l = 4 * kk + (lane >> 4);
        wf[e] = (j < D && l <= j) ? -W[j * ld + l] : 0.0;
    }
    for (int l = tid; l < D; l += NT) d.mu[(long long)s * D + l] = mu[l];
    if (tid < 64) {
        double m2 = 0.0;
        for (int l = tid; l < D; l += 64) m2 = fma(mu[l], mu[l], m2);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m2 += __shfl_xor(m2, o);
        if (tid == 0) {
            d.sc[s] = make_consts(d, d.n[s], logdetC, lam, m2);
            d.mu_ver[s] += 1;                 // what is cached per point against this slot's state is stale now
        }
    }
}

// ------------------------------------------------------------------------------------------
// Window bookkeeping (one thread)
// ------------------------------------------------------------------------------------------
__device__ inline void set_chunks(const Dev &d, Job &j) {
    const long long rows = j.win_hi - j.pos;
    const long long rb = (rows + d.rows_per_block - 1) / d.rows_per_block;
    const int nlist = (j.mode == MODE_FRESH) ? j.K : j.n_dirty;
    long long ch = rb > 0 ? (d.target_blocks + rb - 1) / rb : 1;
    if (ch > kMaxChunks) ch = kMaxChunks;
    if (ch > nlist) ch = nlist;
    if (job_is_pruned(d, j.mode, j.prune)) {
        // the pruning kernel hands out GROUPS of 16 labels; a wave's fixed cost (its 32 rows of X,
        // the gathers behind Mlb) is paid once per chunk, so as few chunks as fill the chip
        const long long ngroups = (nlist + 15) / 16;
        if (ch > ngroups) ch = ngroups;
    }
    if (ch < 1) ch = 1;
    j.chunks = (int)ch;
}

__device__ inline void start_window(const Dev &d, Ctrl *c, long long pos) {
    Job &j = c->job;
    j.pos = pos;
    j.win_base = pos;
    // Pruned scores are valid against the FROZEN state only (a move can lower a visit's best
    // score and promote a pruned component), so they are used while moves are sparse; the
    // mover-dense path (resolver) always works on complete scores.
    const bool prune = d.prune_enabled == 2 || (d.prune_enabled == 1 && c->ema_run >= kPruneMinRun);
    // (win_size is about half the mean distance between movers: right for a pruned window, which
    // ends at its first mover; a dense window is only re-scored for two components behind a mover,
    // so it may as well reach a few movers ahead and save the steps of the clean windows in between)
    long long w = prune ? (long long)c->win_size : 4ll * c->win_size;
    if (w > d.batch_rows) w = d.batch_rows;
    long long hi = pos + w;
    if (hi > c->n_visits) hi = c->n_visits;
    j.win_hi = hi;
    j.n_dirty = 0;
    j.prune = 0;
    // (with certify_kernel in front the sorted subset follows its verdicts: re-sorted every time)
    c->skip_sort = (!d.use_certify && c->wsort_valid && c->wsort_base == pos && c->wsort_hi == hi &&
                    c->wsort_padded == (d.use_home ? 1 : 0)) ? 1 : 0;
    if (pos >= c->n_visits) {
        j.mode = MODE_DONE;
    } else {
        j.mode = MODE_FRESH;
        c->n_windows += 1;
        j.prune = prune ? 1 : 0;
    }
    set_chunks(d, j);
}

// Safe-stay windows (kernels_safe.hip): the open window becomes the stretch the next proof pass examines.
__device__ inline void safe_open_window(const Dev &d, Ctrl *c) {
    Job &j = c->job;
    if (j.mode == MODE_DONE) return;
    if (c->safe_L < 256) c->safe_L = 256;
    long long w = c->safe_L;
    if (w > d.batch_rows) w = d.batch_rows;
    long long hi = j.pos + w;
    if (hi > c->n_visits) hi = c->n_visits;
    if (d.safe_dense && d.ahead_C > 0) {          // (the look-ahead's ring holds whole chunks: a stretch lies inside one)
        const long long chunk_end = (j.pos / d.ahead_C + 1) * (long long)d.ahead_C;
        if (hi > chunk_end) hi = chunk_end;
    }
    j.win_base = j.pos;
    j.win_hi = hi;
    j.mode = MODE_FRESH;
    j.n_dirty = 0;
    j.prune = 1;
    c->skip_sort = 0;
    c->first_mover = kNoMover;
    set_chunks(d, j);
}

// The next window of a stretch whose proofs stand: the next kGramRows listed visits.
__device__ inline void safe_next_window(const Dev &d, Ctrl *c) {
    const int off = c->gl_off, total = c->gl_total;
    const int n = total - off < kGramRows ? total - off : kGramRows;
    c->gl_n = n > 0 ? n : 0;
    c->gl_end = off + kGramRows < total ? d.glist[off + kGramRows] : c->gl_stretch_end;
}

// Running mean distance between movers.  A run far below the mean (the chain has just been
// disturbed: the mean still remembers the quiet stretch before) pulls it down fast -- every mover
// that arrives while the windows are still sized for the old mean throws a whole window away.
__device__ __forceinline__ double ema_after_mover(double ema, double run) {
    return run < 0.25 * ema ? 0.5 * ema + 0.5 * run : 0.875 * ema + 0.125 * run;
}

// window size from the running mean distance between movers: about half of it, a power of two
__device__ inline long long window_for_rate(const Ctrl *c) {
    long long w = 64;
    while (w < c->win_cap && (double)w < 0.5 * c->ema_run) w <<= 1;
    if (w > c->win_cap) w = c->win_cap;
    return w;
}

// ------------------------------------------------------------------------------------------
// Diagonal covariance (reference gaussian_components_diag.py:325-338): everything derived from
// (n, m[D], S[D]) of one slot.  red: 2*NT doubles of LDS.  Barriers inside.
// ------------------------------------------------------------------------------------------
template <int NT>
__device__ inline void refresh_diag_slot(const Dev &d, int s, double *red, int tid) {
    const int D = d.D;
    const int n = d.n[s];
    const double k_N = d.k0 + (double)n;
    const long long v_N = d.v0 + n;
    const double scale = (k_N + 1.0) / (k_N * (double)v_N);
    const double inv_v = 1.0 / (double)v_N;
    double lpv = 0.0, lsn = 0.0, wmin = INFINITY;
    bool bad = false;
    for (int l = tid; l < D; l += NT) {
        const double mean = d.m[(long long)s * D + l] / k_N;
        const double sn = d.S[(long long)s * D + l] - k_N * (mean * mean);
        const double var = scale * sn;
        if (!(sn > 0.0)) bad = true;
        d.mu[(long long)s * D + l] = mean;
        const double wl = (1.0 / var) * inv_v;
        d.dw[(long long)s * D + l] = wl;
        wmin = fmin(wmin, wl);
        lpv += log(var);
        lsn += log(sn);
    }
    red[tid] = lpv;
    red[NT + tid] = lsn;
    if (bad) atomicCAS(&d.ctrl->error, 0, -4);
    __syncthreads();
    double a = 0.0, b = 0.0;
    if (tid == 0)
        for (int t = 0; t < NT; ++t) { a += red[t]; b += red[NT + t]; }
    __syncthreads();
    red[tid] = wmin;
    __syncthreads();
    if (tid == 0) {
        for (int t = 0; t < NT; ++t) wmin = fmin(wmin, red[t]);
        SlotConst c;
        c.A = (double)D * (d.tab_lgam[v_N + 1] - d.tab_lgam[v_N] - 0.5 * d.tab_log[v_N] - 0.5 * BGMM_LOG_PI) - 0.5 * a;
        c.half_vd = 0.5 * (double)(v_N + 1);
        c.A1 = a;                 // log prod of the predictive variances (reference log_prod_vars)
        c.half_vd1 = 0.0; c.coef1 = 0.0; c.a1 = 0.0;
        c.logdetC = b;            // sum_d log S_N,d (log_marg_k)
        c.logseat = d.tabSeat[n];
        c.logseat1 = n >= 1 ? d.tabSeat[n - 1] : 0.0;
        // pruned windows: sum_d log(1 + a_d) >= log(1 + sum_d a_d) >= log(1 + min_d(w_d) |x - mu|^2)
        c.inv_cv = 1.0;
        c.inv_lam = (wmin > 0.0 && wmin < INFINITY) ? wmin * (1.0 - 1e-12) : 0.0;
        c.mu2 = 0.0;
        d.sc[s] = c;
        d.mu_ver[s] += 1;                     // (per-point caches against this slot's state are stale now)
        d.nupd[s] = 0;
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------
// Fixed-variance components (reference gaussian_components_fixedvar.py:282-291): predictive
// precision pN p / (pN + p) per dimension; the bare prior (pseudo slot K_max) is scored with
// precision_0 itself, as the reference's log_prior does (:205-212).  red: NT doubles of LDS.
// ------------------------------------------------------------------------------------------
template <int NT>
__device__ inline void refresh_fixed_slot(const Dev &d, int s, double *red, int tid) {
    const int D = d.D;
    const int n = d.n[s];
    double lpp = 0.0, wmin = INFINITY;
    for (int l = tid; l < D; l += NT) {
        const double pN = d.S[(long long)s * 2 * D + l];
        const double p = d.prior_S[D + l];
        const double pp = s == d.K_max ? pN : pN * p / (pN + p);
        d.mu[(long long)s * D + l] = d.m[(long long)s * D + l] / pN;
        d.dw[(long long)s * D + l] = pp;
        wmin = fmin(wmin, pp);
        lpp += log(pp);
    }
    red[tid] = lpp;
    __syncthreads();
    double a = 0.0;
    if (tid == 0)
        for (int t = 0; t < NT; ++t) a += red[t];
    __syncthreads();
    red[tid] = wmin;
    __syncthreads();
    if (tid == 0) {
        for (int t = 0; t < NT; ++t) wmin = fmin(wmin, red[t]);
        SlotConst c;
        c.A = -0.5 * (double)D * log(2.0 * 3.14159265358979323846) + 0.5 * a;
        c.half_vd = 0.5;
        c.inv_cv = 1.0;
        c.A1 = a;                 // log prod of the predictive precisions
        c.half_vd1 = 0.0; c.coef1 = 0.0; c.a1 = 0.0;
        c.logdetC = 0.0;
        c.logseat = d.tabSeat[n];
        c.logseat1 = n >= 1 ? d.tabSeat[n - 1] : 0.0;
        // pruned windows: sum_d (x_d - mu_d)^2 pp_d >= min_d(pp_d) |x - mu|^2
        c.inv_lam = (wmin > 0.0 && wmin < INFINITY) ? wmin * (1.0 - 1e-12) : 0.0;
        c.mu2 = 0.0;
        d.sc[s] = c;
        d.mu_ver[s] += 1;                     // (per-point caches against this slot's state are stale now)
        d.nupd[s] = 0;
    }
    __syncthreads();
}

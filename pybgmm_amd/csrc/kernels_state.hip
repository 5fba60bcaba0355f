// Component-state kernels: sufficient-statistics build, covariance refresh (Cholesky +
// triangular inverse + Student-t constants), the move applier that advances the speculative
// window, and the small read-out kernels (labels, log marginal, stats export).
//
// Reference behaviour restated here (file:line in the reference checkout):
//   init_stats_kernel ........ gaussian_components.py:96-111, 154-169 (k ascending, i ascending)
//   refresh_kernel ........... gaussian_components.py:319-331 (what it feeds: :228-251)
//   apply_kernel / item ops .. gaussian_components.py:154-205, igmm/crpmm.py:82-88
//   log_marg_kernel .......... igmm/igmm.py:199-215, gaussian_components.py:253-289
// Compiled with -ffp-contract=off: `m += x`, `S += x*x'` must round the product and the sum
// separately so that the statistics stay bit-identical to numpy's.
#include "bgmm_device.h"
#include "slot_math.h"

#define TPB 256

// ------------------------------------------------------------------------------------------
// Initial statistics: one block per initial label, members visited in ascending i.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void init_stats_kernel(Dev d, const int *__restrict__ members,
                                                         const long long *__restrict__ offsets) {
    const int k = blockIdx.x;                 // initial label == slot
    const int D = d.D;
    const long long lo = offsets[k], hi = offsets[k + 1];
    if (d.cov_type == COV_FIXED) {            // gaussian_components_fixedvar.py:146-162
        for (int a = threadIdx.x; a < D; a += TPB) {
            double mn = d.prior_m[a], pN = d.prior_S[a], sq = 0.0;
            const double p = d.prior_S[D + a];
            for (long long t = lo; t < hi; ++t) {
                const double x = d.X[(long long)members[t] * D + a];
                mn = __dadd_rn(mn, __dmul_rn(p, x));
                pN = __dadd_rn(pN, p);
                sq = __dadd_rn(sq, __dmul_rn(x, x));
            }
            d.m[(long long)k * D + a] = mn;
            d.S[(long long)k * 2 * D + a] = pN;
            d.S[(long long)k * 2 * D + D + a] = sq;
        }
        if (threadIdx.x == 0) d.n[k] = (int)(hi - lo);
        return;
    }
    if (d.cov_type == COV_DIAG) {             // gaussian_components_diag.py:162-176: S += square(x)
        for (int a = threadIdx.x; a < D; a += TPB) {
            double accS = d.prior_S[a], accm = d.prior_m[a];
            for (long long t = lo; t < hi; ++t) {
                const double x = d.X[(long long)members[t] * D + a];
                accS = __dadd_rn(accS, __dmul_rn(x, x));
                accm = __dadd_rn(accm, x);
            }
            d.S[(long long)k * D + a] = accS;
            d.m[(long long)k * D + a] = accm;
        }
        if (threadIdx.x == 0) d.n[k] = (int)(hi - lo);
        return;
    }
    for (int e = threadIdx.x; e < D * D; e += TPB) {
        const int a = e / D, b = e % D;
        double acc = d.prior_S[e];
        for (long long t = lo; t < hi; ++t) {
            const double *x = d.X + (long long)members[t] * D;
            acc = __dadd_rn(acc, __dmul_rn(x[a], x[b]));
        }
        d.S[(long long)k * D * D + e] = acc;
    }
    for (int a = threadIdx.x; a < D; a += TPB) {
        double acc = d.prior_m[a];
        for (long long t = lo; t < hi; ++t) acc = __dadd_rn(acc, d.X[(long long)members[t] * D + a]);
        d.m[(long long)k * D + a] = acc;
    }
    if (threadIdx.x == 0) d.n[k] = (int)(hi - lo);
}

__global__ void init_labels_kernel(Dev d, const long long *__restrict__ z_in, int K_init) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < d.N) d.z[i] = (int)z_in[i];
    if (i < d.K_max) { d.perm[i] = (int)i; d.label_of_slot[i] = (int)i; }
    if (i == 0) {
        Ctrl *c = d.ctrl;
        c->job.K = K_init;
        c->job.mode = MODE_DONE;
        c->error = 0;
        c->first_mover = kNoMover;
        c->n_refresh = 0;
    }
    // slots that are not part of the initial labelling start empty
    if (i >= K_init && i < d.K_max) d.n[i] = 0;
}

void launch_init_stats(const Dev &d, const int *members, const long long *offsets, int K_init,
                       hipStream_t st) {
    // (labels first: it zeroes the counts of unused slots)
    (void)members; (void)offsets;
    if (K_init > 0) hipLaunchKernelGGL(init_stats_kernel, dim3(K_init), dim3(TPB), 0, st, d, members, offsets);
}

// ------------------------------------------------------------------------------------------
// Refresh: everything derived from (n, m, S) of one slot.
//   C = S - k_N mu mu^T  (the reference's S_N; its covariance is c*C with
//   c = (k_N+1)/(k_N (v_N-D+1))),  C = L L^T,  Winv = L^-1,  logdetC = 2 sum log L_jj.
// Two routes (slot_math.h): from scratch, O(D^3); or a rank-1 change of Winv, O(D^2).
// LDS: W[D][D+1] + 6 D + 4 doubles.
// ------------------------------------------------------------------------------------------
int refresh_lds_bytes(int D) {
    const int full = (D * (D + 1) + 6 * D + 4) * (int)sizeof(double), diag = 2 * 256 * (int)sizeof(double);
    return full > diag ? full : diag;
}

__device__ void refresh_slot(const Dev &d, int s, double *sm) {
    const int D = d.D, ld = D + 1, tid = threadIdx.x;
    double *A = sm;
    double *mu = sm + D * ld, *row = mu + D;
    double *scal = row + 5 * D;                    // [0] logdet, [1] bad flag (as int)
    const double k_N = d.k0 + (double)d.n[s];
    const double *m = d.m + (long long)s * D;
    const double *S = d.S + (long long)s * D * D;
    for (int a = tid; a < D; a += TPB) mu[a] = m[a] / k_N;
    __syncthreads();
    for (int e = tid; e < D * D; e += TPB) {
        const int a = e / D, b = e % D;
        if (b <= a) A[a * ld + b] = S[e] - k_N * (mu[a] * mu[b]);
    }
    __syncthreads();
    gershgorin_bound<TPB>(A, ld, D, row, &scal[2], tid, true);
    chol_inverse<TPB>(A, ld, D, row, &scal[0], (int *)&scal[1], tid, true);
    if (tid == 0 && *(int *)&scal[1]) atomicCAS(&d.ctrl->error, 0, -4);
    write_slot<TPB>(d, s, A, ld, mu, scal[0], scal[2], tid, nullptr, true);
    if (tid == 0) d.nupd[s] = 0;
}

// Rank-1 route after point i joined (kind ADD / NEW) or left (SUB) slot dst; `src` holds the
// state before the change (dst itself, or the prior pseudo slot for a new component).
__device__ void rank1_slot(const Dev &d, int src, int dst, long long i, int kind, double *sm) {
    const int D = d.D, ld = D + 1, tid = threadIdx.x;
    double *W = sm;
    double *mu = sm + D * ld, *dv = mu + D, *pv = dv + D, *lv = pv + D, *tv = lv + D;
    double *scal = tv + 2 * D;
    const double logdet_src = d.sc[src].logdetC;      // read before anything of dst is rewritten
    const double inv_lam_src = d.sc[src].inv_lam;
    const int n_new = d.n[dst];
    const double k_before = d.k0 + (double)(kind == REFRESH_SUB ? n_new + 1 : n_new - 1);
    const double a = kind == REFRESH_SUB ? -k_before / (k_before - 1.0) : k_before / (k_before + 1.0);
    const double *x = d.X + i * D;
    const double *Wsrc = d.Wrm + (long long)src * D * D;
    for (int e = tid; e < D * D; e += TPB) W[(e / D) * ld + (e % D)] = Wsrc[e];
    for (int l = tid; l < D; l += TPB) {
        dv[l] = x[l] - d.mu[(long long)src * D + l];
        mu[l] = d.m[(long long)dst * D + l] / (d.k0 + (double)n_new);
    }
    __syncthreads();
    rank1_inverse_factor<TPB>(W, ld, D, a, dv, pv, lv, tv, &scal[0], (int *)&scal[1], tid, true);
    if (tid == 0 && *(int *)&scal[1]) atomicCAS(&d.ctrl->error, 0, -4);
    double d2 = 0.0;
    if (tid == 0) for (int l = 0; l < D; ++l) d2 = fma(dv[l], dv[l], d2);
    write_slot<TPB>(d, dst, W, ld, mu, logdet_src + log(1.0 + a * scal[0]), lam_after_rank1(inv_lam_src, a, d2),
                    tid, nullptr, true);
    if (tid == 0) d.nupd[dst] += 1;
}

__global__ __launch_bounds__(TPB) void refresh_list_kernel(Dev d, const int *__restrict__ slots, int n) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if ((int)blockIdx.x >= n) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) { d.ctrl->tables_valid = 0; d.ctrl->wsort_valid = 0; d.ctrl->state_epoch += 1; }
    const int s = slots ? slots[blockIdx.x] : (int)blockIdx.x;
    if (d.cov_type == COV_DIAG) refresh_diag_slot<TPB>(d, s, sm, threadIdx.x);
    else if (d.cov_type == COV_FIXED) refresh_fixed_slot<TPB>(d, s, sm, threadIdx.x);
    else refresh_slot(d, s, sm);
}

// From-scratch rebuild of every live slot that has taken rank-1 steps since its last rebuild.  Run
// once at the start of a sweep that follows a sweep with moves: the eigenvalue bound behind the
// pruning only ever grows under rank-1 steps (slot_math.h: lam_after_rank1), so after a burn-in it
// would stay loose until a slot's 64th update.
__global__ __launch_bounds__(TPB) void refresh_stale_kernel(Dev d) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    if ((int)blockIdx.x >= d.ctrl->job.K) return;
    const int s = d.perm[blockIdx.x];
    if (d.nupd[s] == 0) return;
    if (threadIdx.x == 0) { d.ctrl->tables_valid = 0; atomicAdd((unsigned long long *)&d.ctrl->state_epoch, 1ull); }   // (means and bounds change; the homes do not)
    refresh_slot(d, s, sm);
}

__global__ __launch_bounds__(TPB) void refresh_ctrl_kernel(Dev d) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    const Ctrl *c = d.ctrl;
    if ((int)blockIdx.x >= c->n_refresh) return;
    const int s = c->refresh[blockIdx.x], kind = c->refresh_kind[blockIdx.x];
    if (d.cov_type == COV_DIAG) refresh_diag_slot<TPB>(d, s, sm, threadIdx.x);
    else if (d.cov_type == COV_FIXED) refresh_fixed_slot<TPB>(d, s, sm, threadIdx.x);
    else if (kind == REFRESH_SCRATCH) refresh_slot(d, s, sm);
    else rank1_slot(d, kind == REFRESH_NEW ? d.K_max : s, s, c->refresh_i, kind, sm);
}

static void ensure_lds(const void *fn, int bytes) {
    if (bytes > 64 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
}

void launch_refresh_list(const Dev &d, const int *slots, int n, hipStream_t st) {
    if (n <= 0) return;
    const int lds = refresh_lds_bytes(d.D);
    ensure_lds((const void *)refresh_list_kernel, lds);
    hipLaunchKernelGGL(refresh_list_kernel, dim3(n), dim3(TPB), lds, st, d, slots, n);
}

void launch_refresh_stale(const Dev &d, int K, hipStream_t st) {
    if (K <= 0) return;
    const int lds = refresh_lds_bytes(d.D);
    ensure_lds((const void *)refresh_stale_kernel, lds);
    hipLaunchKernelGGL(refresh_stale_kernel, dim3(K), dim3(TPB), lds, st, d);
}

void launch_refresh_ctrl(const Dev &d, hipStream_t st) {
    const int lds = refresh_lds_bytes(d.D);
    ensure_lds((const void *)refresh_ctrl_kernel, lds);
    hipLaunchKernelGGL(refresh_ctrl_kernel, dim3(2), dim3(TPB), lds, st, d);
}

// ------------------------------------------------------------------------------------------
// Window bookkeeping (thread 0 of the applier / sweep_begin only)
// ------------------------------------------------------------------------------------------
// tabG[v], v = 1 .. tab_len-D-1;  tabLogC[n], n = 0 .. N+1 (log c1 of a slot with n is tabLogC[n-1])
__global__ void build_tables_kernel(Dev d, double *tabG, double *tabLogC) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= 1 && t + d.D < d.tab_len) tabG[t] = student_const(d, t);
    if (t <= d.N + 1) tabLogC[t] = log(cov_scale(d, (int)t));
}
__global__ void build_seat_table_kernel(Dev d, double *tabSeat) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t <= d.N + 1) tabSeat[t] = seat_weight(d, (int)t);
}
void launch_build_tables(const Dev &d, double *tabG, double *tabLogC, hipStream_t st) {
    hipLaunchKernelGGL(build_tables_kernel, dim3((unsigned)((d.tab_len + 255) / 256)), dim3(256), 0, st, d, tabG, tabLogC);
}
void launch_build_seat_table(const Dev &d, double *tabSeat, hipStream_t st) {
    hipLaunchKernelGGL(build_seat_table_kernel, dim3((unsigned)((d.N + 2 + 255) / 256)), dim3(256), 0, st, d, tabSeat);
}

__global__ __launch_bounds__(TPB) void sweep_begin_kernel(Dev d) {
    Ctrl *c = d.ctrl;
    const int K = c->job.K;
    // seating weights depend on the sweep's exponent (tabSeat was rebuilt by the host if it changed)
    for (int j = threadIdx.x; j < K; j += TPB) {
        const int s = d.perm[j];
        d.sc[s].logseat = d.tabSeat[d.n[s]];
        d.sc[s].logseat1 = d.n[s] >= 1 ? d.tabSeat[d.n[s] - 1] : 0.0;
    }
    if (threadIdx.x == 0) {
        // (a whole sweep without a move: the chain is at rest, back to the optimistic window plan it
        // started with -- the mover-free stretch inside one sweep cannot say more than N)
        if (c->n_visits > 0 && c->n_moves == 0 && c->ema_run < 4.0 * (double)c->win_cap)
            c->ema_run = 4.0 * (double)c->win_cap;
        c->n_visits = d.N;
        c->first_mover = kNoMover;
        c->n_refresh = 0;
        c->skip_apply = 0;
        c->lik_evals = 0; c->n_moves = 0; c->n_windows = 0; c->n_steps = 0;
        c->n_score_launches = 0; c->n_scored = 0;
        c->n_kept_blocks = 0; c->n_bound_blocks = 0; c->n_prune_mfma = 0; c->n_certified = 0;
        if (d.seat_dirty) { c->tables_valid = 0; c->state_epoch += 1; }   // (the tables carry log seating weights)
        if (d.order) c->wsort_valid = 0; // (a fresh permutation every sweep)
        c->last_mover = -1;
        if (c->win_size < 64) c->win_size = 64;
        if (c->win_size > c->win_cap) c->win_size = c->win_cap;
        if (c->error == 0) start_window(d, c, 0);
        else c->job.mode = MODE_DONE;
    }
}

// Evaluation order of a pruned window: its visits grouped by home component -- a counting sort
// in three small launches (count per block -> exclusive prefix -> scatter).  wperm[k] = window
// row of the k-th visit in evaluation order.  The order INSIDE a bucket depends on atomic
// arrival order; that only changes which visits share a tile, never a visit's result.
// d.bucket_bins: nslots + 2 global counters (bin b = home slot b - 1; bin 0 = unassigned).
#define BUCKET_ROWS 1024
__global__ __launch_bounds__(256) void bucket_count_kernel(Dev d) {
    extern __shared__ int bins[];
    const Ctrl *c = d.ctrl;
    if (!job_is_pruned(d, c->job.mode, c->job.prune) || c->skip_sort) return;
    const long long base = c->job.win_base;
    const int nrows = (int)(c->job.win_hi - base);
    const int r0 = blockIdx.x * BUCKET_ROWS;
    if (r0 >= nrows) return;
    const int nb = d.nslots + 1;
    for (int b = threadIdx.x; b < nb; b += 256) bins[b] = 0;
    __syncthreads();
    for (int r = r0 + threadIdx.x; r < r0 + BUCKET_ROWS && r < nrows; r += 256) {
        if (d.use_certify && d.cert[r]) continue;          // (proved to stay: not part of the sort)
        const long long p = base + r;
        const long long i = d.order ? d.order[p] : p;
        atomicAdd(&bins[d.z[i] + 1], 1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += 256)
        if (bins[b]) atomicAdd(&d.bucket_bins[b], bins[b]);
}

// The label-ordered tables of the pruned-window kernels for the frozen state of this window
// (bgmm_device.h); rebuilt only after the state changed (Ctrl::tables_valid).
// blocks 0 .. n_tab_blocks-1: fragments / constants / slots, one wave per group of 16 labels;
// the rest: one label a each, its centre-to-centre distances.
__global__ __launch_bounds__(1024) void prune_tables_kernel(Dev d) {
    const Ctrl *c = d.ctrl;
    if (!job_is_pruned(d, c->job.mode, c->job.prune) || c->tables_valid) return;
    const int n_tab_blocks = ((d.nslots + 15) / 16 + 15) / 16;
    if ((int)blockIdx.x >= n_tab_blocks) {
        __shared__ double mua[BGMM_MAX_D];
        const int K = c->job.K, a = (int)blockIdx.x - n_tab_blocks, D = d.D;
        if (a >= K) return;
        const double *__restrict__ pa = d.mu + (long long)d.perm[a] * D;
        for (int l = threadIdx.x; l < D; l += 1024) mua[l] = pa[l];
        __syncthreads();
        for (int b = threadIdx.x; b < K; b += 1024) {
            const double *__restrict__ pb = d.mu + (long long)d.perm[b] * D;
            double acc = 0.0;
            for (int l = 0; l < D; ++l) { const double t = pb[l] - mua[l]; acc = fma(t, t, acc); }
            d.pr_dcc[(long long)a * d.nslots + b] = sqrt(acc);
        }
        return;
    }
    const int K = c->job.K, G = (int)blockIdx.x * 16 + (int)(threadIdx.x >> 6);
    if (16 * G >= K) return;
    const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
    const int t = 16 * G + lr;
    const int s = t < K ? d.perm[t] : -1;
    const int nkk = d.Dp / 4, D = d.D;
    double m2p = 0.0;                            // |mu|^2 over the leading 32 dimensions (level-0 bound)
    for (int kk = 0; kk < nkk; ++kk) {
        const int l = 4 * kk + lk;
        const double v = (s >= 0 && l < D) ? d.mu[(long long)s * D + l] : 0.0;
        d.pr_mufrag[((long long)G * nkk + kk) * 64 + lane] = v;
        if (kk < 8) m2p = fma(v, v, m2p);
    }
    m2p += __shfl_xor(m2p, 16);
    m2p += __shfl_xor(m2p, 32);
    if (lk == 0) {
        const SlotConst *sc = d.sc + (s >= 0 ? s : 0);
        double *g = d.pr_const + (long long)G * 128 + lr;
        g[0] = sc->logseat + sc->A;
        g[16] = sc->half_vd;
        g[32] = sc->inv_lam * sc->inv_cv;
        g[48] = sc->mu2;
        g[64] = m2p;
        d.pr_slot[G * 16 + lr] = s;
    }
}

// For every home label a (one block, 64 threads = 64 radii): ftab[a][j] = max over the other labels t
// of the upper bound of t's log score for a visit whose home is a and whose distance to a's mean is at
// most r_j = j / finv[a]  (triangle inequality through the centre distances, as in the coarse level of
// score_mfma_prune_kernel); the grid reaches half the distance to a's nearest neighbour.  Increasing
// in j.  Runs after prune_tables_kernel (same validity flag, set by apply_kernel afterwards).
__global__ __launch_bounds__(64) void prune_ftable_kernel(Dev d) {
    const Ctrl *c = d.ctrl;
    if (!job_is_pruned(d, c->job.mode, c->job.prune) || c->tables_valid) return;
    const int K = c->job.K, a = blockIdx.x, j = threadIdx.x;
    if (a >= K) return;
    const double *__restrict__ dc = d.pr_dcc + (long long)a * d.nslots;
    double dmin = INFINITY;
    for (int t = 0; t < K; ++t)
        if (t != a) dmin = fmin(dmin, dc[t]);
    const bool fixed = d.cov_type == COV_FIXED;
    const double step = K > 1 ? 0.5 * dmin / 63.0 : 1.0;
    const double rj = (double)j * step * (1.0 + 1e-9);
    double f = -INFINITY;
    for (int t = 0; t < K; ++t) {
        if (t == a) continue;
        const double *__restrict__ g = d.pr_const + (long long)(t >> 4) * 128 + (t & 15);
        double dl = dc[t] * (1.0 - 1e-9) - rj;
        dl = dl > 0.0 ? dl : 0.0;
        const double tt = dl * dl * g[32];
        // (log1p_lower of kernels_prune.hip, restated: frexp + chord)
        const double y = 1.0 + tt;
        const double m = __builtin_amdgcn_frexp_mant(y);
        const int e = __builtin_amdgcn_frexp_exp(y);
        const double L = 0.6931471805599453 * ((double)(e - 2) + 2.0 * m);
        f = fmax(f, g[0] - g[16] * (fixed ? tt : L));
    }
    d.ftab[(long long)a * 64 + j] = f;
    if (j == 0) d.finv[a] = (step > 0.0 && step < INFINITY) ? 1.0 / step : 0.0;
}

// exclusive prefix over the bins of the bucket sort (one block)
__global__ __launch_bounds__(1024) void bucket_prefix_kernel(Dev d) {
    __shared__ int wsum_[16];
    const Ctrl *c = d.ctrl;
    if (!job_is_pruned(d, c->job.mode, c->job.prune) || c->skip_sort) return;
    const int nb = d.nslots + 1;
    // exclusive prefix over nb <= ~1k bins: every thread owns a contiguous run
    const int per = (nb + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
    int local = 0;
    for (int b = lo; b < hi; ++b) local += d.bucket_bins[b];
    int incl = local;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) wsum_[w] = incl;
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < w; ++k) woff += wsum_[k];
    int run = woff + incl - local;
    for (int b = lo; b < hi; ++b) { const int v = d.bucket_bins[b]; d.bucket_bins[b] = run; run += v; }
    if (threadIdx.x == 1023) d.ctrl->n_sorted = run;          // rows that take part (all but the certified ones)
}

__global__ __launch_bounds__(256) void bucket_scatter_kernel(Dev d) {
    extern __shared__ int lds[];                  // [nb] local counts, then [nb] reserved bases
    const Ctrl *c = d.ctrl;
    if (!job_is_pruned(d, c->job.mode, c->job.prune) || c->skip_sort) return;
    const long long base = c->job.win_base;
    const int nrows = (int)(c->job.win_hi - base);
    const int r0 = blockIdx.x * BUCKET_ROWS;
    if (r0 >= nrows) return;
    const int nb = d.nslots + 1;
    int *cnt = lds, *res = lds + nb;
    for (int b = threadIdx.x; b < nb; b += 256) cnt[b] = 0;
    __syncthreads();
    int myb[BUCKET_ROWS / 256], myk[BUCKET_ROWS / 256];
#pragma unroll
    for (int t = 0; t < BUCKET_ROWS / 256; ++t) {
        const int r = r0 + threadIdx.x + t * 256;
        myb[t] = -1;
        if (r < nrows && !(d.use_certify && d.cert[r])) {
            const long long p = base + r;
            const long long i = d.order ? d.order[p] : p;
            myb[t] = d.z[i] + 1;
            myk[t] = atomicAdd(&cnt[myb[t]], 1);          // rank inside (block, bucket)
        }
    }
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += 256)
        res[b] = cnt[b] ? atomicAdd(&d.bucket_bins[b], cnt[b]) : 0;     // reserve this block's range
    __syncthreads();
#pragma unroll
    for (int t = 0; t < BUCKET_ROWS / 256; ++t)
        if (myb[t] >= 0) {
            const int r = r0 + threadIdx.x + t * 256;
            const int k = res[myb[t]] + myk[t];
            const long long p = base + r;
            d.wperm[k] = r;
            WRec rec;
            rec.i = d.order ? d.order[p] : p;
            rec.home = myb[t] - 1;
            rec.home_label = rec.home >= 0 ? d.label_of_slot[rec.home] : -1;
            rec.mlb0 = d.log_alpha + d.log_prior[rec.i];
            rec.pad = 0.0;
            d.wrec[k] = rec;
        }
}

void launch_prune_tables(const Dev &d, hipStream_t st) {
    const unsigned ngrp = (unsigned)((d.nslots + 15) / 16);
    hipLaunchKernelGGL(prune_tables_kernel, dim3((ngrp + 15) / 16 + d.nslots), dim3(1024), 0, st, d);
    // (always together with the tables, also in sweeps that do not certify: one validity flag)
    hipLaunchKernelGGL(prune_ftable_kernel, dim3(d.nslots), dim3(64), 0, st, d);
}

void launch_bucket_rows(const Dev &d, long long max_rows, hipStream_t st) {
    // (d.bucket_bins is zero on entry: cleared at create and by certify_kernel / choice_sparse_kernel)
    const int nb = d.nslots + 1;
    const unsigned g = (unsigned)((max_rows + BUCKET_ROWS - 1) / BUCKET_ROWS);
    hipLaunchKernelGGL(bucket_count_kernel, dim3(g), dim3(256), nb * (int)sizeof(int), st, d);
    hipLaunchKernelGGL(bucket_prefix_kernel, dim3(1), dim3(1024), 0, st, d);
    hipLaunchKernelGGL(bucket_scatter_kernel, dim3(g), dim3(256), 2 * nb * (int)sizeof(int), st, d);
}

void launch_sweep_begin(const Dev &d, hipStream_t st) {
    hipLaunchKernelGGL(sweep_begin_kernel, dim3(1), dim3(TPB), 0, st, d);
}

// ------------------------------------------------------------------------------------------
// Moves.  `unseat` / `seat` update the label<->slot maps and counts (thread 0) and report
// which slots need their (m, S) touched; the block then applies the rank-1 changes.
// ------------------------------------------------------------------------------------------
struct MovePlan {
    long long i;
    int sub_slot;     // slot to subtract x from (-1: none)
    int add_slot;     // slot to add x to (-1: none)
    int add_init;     // 1: add_slot is a freshly opened component (start from the prior)
};

// remove point i from its slot (del_item semantics).  Returns slot to subtract from, or -1.
__device__ int plan_unseat(const Dev &d, Ctrl *c, long long i) {
    const int h = d.z[i];
    if (h < 0) return -1;
    d.z[i] = -1;
    const int nh = d.n[h] - 1;
    d.n[h] = nh;
    if (nh > 0) return h;
    // swap-with-last delete of label lab (gaussian_components.py:188-205)
    const int lab = d.label_of_slot[h];
    const int last = c->job.K - 1;
    const int s_last = d.perm[last];
    d.perm[lab] = s_last;
    d.label_of_slot[s_last] = lab;
    d.perm[last] = h;
    d.label_of_slot[h] = last;
    c->job.K = last;
    return -1;
}

// seat point i at label `lab` (add_item semantics); returns 0 or an error code
__device__ int plan_seat(const Dev &d, Ctrl *c, long long i, int lab, MovePlan &mp) {
    int K = c->job.K;
    int t;
    mp.add_init = 0;
    if (lab >= K) {
        if (K >= d.K_max) return -3;
        t = d.perm[K];
        d.label_of_slot[t] = K;
        d.n[t] = 0;
        d.nupd[t] = 0;
        c->job.K = K + 1;
        mp.add_init = 1;
    } else {
        t = d.perm[lab];
    }
    d.n[t] += 1;
    d.z[i] = t;
    mp.add_slot = t;
    return 0;
}

__device__ void apply_rank1(const Dev &d, const MovePlan &mp) {
    const int D = d.D;
    const double *x = d.X + mp.i * D;
    if (d.cov_type == COV_FIXED) {
        for (int a = threadIdx.x; a < D; a += TPB) {
            const double p = d.prior_S[D + a];
            const double px = __dmul_rn(p, x[a]), xx = __dmul_rn(x[a], x[a]);
            if (mp.sub_slot >= 0) {
                const long long o = (long long)mp.sub_slot * D + a, o2 = (long long)mp.sub_slot * 2 * D + a;
                d.m[o] = __dsub_rn(d.m[o], px);
                d.S[o2] = __dsub_rn(d.S[o2], p);
                d.S[o2 + D] = __dsub_rn(d.S[o2 + D], xx);
            }
            if (mp.add_slot >= 0) {
                const long long o = (long long)mp.add_slot * D + a, o2 = (long long)mp.add_slot * 2 * D + a;
                d.m[o] = __dadd_rn(mp.add_init ? d.prior_m[a] : d.m[o], px);
                d.S[o2] = __dadd_rn(mp.add_init ? d.prior_S[a] : d.S[o2], p);
                d.S[o2 + D] = __dadd_rn(mp.add_init ? 0.0 : d.S[o2 + D], xx);
            }
        }
        return;
    }
    if (d.cov_type == COV_DIAG) {
        for (int a = threadIdx.x; a < D; a += TPB) {
            const double xx = __dmul_rn(x[a], x[a]);
            if (mp.sub_slot >= 0) {
                const long long o = (long long)mp.sub_slot * D + a;
                d.m[o] = __dsub_rn(d.m[o], x[a]);
                d.S[o] = __dsub_rn(d.S[o], xx);
            }
            if (mp.add_slot >= 0) {
                const long long o = (long long)mp.add_slot * D + a;
                d.m[o] = __dadd_rn(mp.add_init ? d.prior_m[a] : d.m[o], x[a]);
                d.S[o] = __dadd_rn(mp.add_init ? d.prior_S[a] : d.S[o], xx);
            }
        }
        return;
    }
    if (mp.sub_slot >= 0) {
        double *m = d.m + (long long)mp.sub_slot * D;
        double *S = d.S + (long long)mp.sub_slot * D * D;
        for (int a = threadIdx.x; a < D; a += TPB) m[a] = __dsub_rn(m[a], x[a]);
        for (int e = threadIdx.x; e < D * D; e += TPB)
            S[e] = __dsub_rn(S[e], __dmul_rn(x[e / D], x[e % D]));
    }
    if (mp.add_slot >= 0) {
        double *m = d.m + (long long)mp.add_slot * D;
        double *S = d.S + (long long)mp.add_slot * D * D;
        for (int a = threadIdx.x; a < D; a += TPB)
            m[a] = __dadd_rn(mp.add_init ? d.prior_m[a] : m[a], x[a]);
        for (int e = threadIdx.x; e < D * D; e += TPB)
            S[e] = __dadd_rn(mp.add_init ? d.prior_S[e] : S[e], __dmul_rn(x[e / D], x[e % D]));
    }
}

// rank1 = false: always rebuild from scratch (API item ops); else rank-1 steps with a
// from-scratch rebuild every kRefreshEvery steps per slot
__device__ void set_refresh(const Dev &d, Ctrl *c, const MovePlan &mp, bool rank1) {
    int nr = 0;
    c->refresh_i = mp.i;
    if (d.cov_type != COV_FULL) rank1 = false;      // the diag / fixed refresh is O(D) anyway
    if (mp.sub_slot >= 0) {
        // (A point far from the rest of its component inflates the eigenvalue bound behind the
        // pruning, and a rank-1 removal cannot shrink it again -- slot_math.h: lam_after_rank1.  The
        // slots that took rank-1 steps are rebuilt before the next sweep, refresh_stale_kernel; a
        // rebuild here, on the movers' chain, cost a disturbed chain 125 us per departing outlier.)
        const bool scratch = !(rank1 && d.nupd[mp.sub_slot] < kRefreshEvery);
        c->refresh_kind[nr] = scratch ? REFRESH_SCRATCH : REFRESH_SUB;
        c->refresh[nr++] = mp.sub_slot;
    }
    if (mp.add_slot >= 0 && mp.add_slot != mp.sub_slot) {
        int kind = REFRESH_SCRATCH;
        if (rank1) {
            if (mp.add_init) kind = REFRESH_NEW;
            else if (d.nupd[mp.add_slot] < kRefreshEvery) kind = REFRESH_ADD;
        }
        c->refresh_kind[nr] = kind;
        c->refresh[nr++] = mp.add_slot;
    }
    c->n_refresh = nr;
}

// The control block as thread 0 has just left it (and fenced), copied to the host-mapped mirror by
// the first lanes of the block: one 8-byte word each, read past the L1 -- contiguous stores that
// leave the GPU as a few bus writes instead of one per word.  Called by ALL threads of the block.
__device__ __forceinline__ void publish_ctrl_block(const Dev &d) {
    static_assert(sizeof(Ctrl) % sizeof(long long) == 0 && sizeof(Ctrl) / sizeof(long long) <= TPB,
                  "Ctrl is copied in 8-byte words, one per thread");
    __syncthreads();
    if (threadIdx.x < (int)(sizeof(Ctrl) / sizeof(long long))) {
        const long long v = __hip_atomic_load((const long long *)d.ctrl + threadIdx.x, __ATOMIC_RELAXED,
                                              __HIP_MEMORY_SCOPE_AGENT);
        ((long long *)d.ctrl_pub)[threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(TPB) void apply_kernel(Dev d) {
    __shared__ MovePlan mp;
    __shared__ int do_move;
    Ctrl *c = d.ctrl;
    if (d.lean_step && c->job.mode != MODE_DONE) {
        // only certify_kernel ran: the step stands iff it certified every visit of the window
        __shared__ unsigned long long csum[TPB];
        __shared__ int left;
        csum[threadIdx.x] = d.pr_counts[768 + threadIdx.x];
        __syncthreads();
        for (int o = TPB / 2; o > 0; o >>= 1) {
            if (threadIdx.x < o) csum[threadIdx.x] += csum[threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            left = (long long)csum[0] < c->job.win_hi - c->job.pos ? 1 : 0;
            if (left) { c->retry_full = 1; c->n_refresh = 0; __threadfence(); }
        }
        __syncthreads();
        if (left) {
            for (int t = 0; t < 4; ++t) d.pr_counts[t * 256 + threadIdx.x] = 0;     // (its counts are discarded)
            if (d.publish) publish_ctrl_block(d);
            return;
        }
    }
    if (job_is_pruned(d, c->job.mode, c->job.prune) && !c->skip_apply) {
        // fold (and clear) the pruning kernel's spread counters of this window
        __shared__ unsigned long long cnt_red[4 * TPB];
        for (int t = 0; t < 4; ++t) {
            cnt_red[t * TPB + threadIdx.x] = d.pr_counts[t * 256 + threadIdx.x];
            d.pr_counts[t * 256 + threadIdx.x] = 0;
        }
        __syncthreads();
        for (int o = TPB / 2; o > 0; o >>= 1) {
            if (threadIdx.x < o)
                for (int t = 0; t < 4; ++t) cnt_red[t * TPB + threadIdx.x] += cnt_red[t * TPB + threadIdx.x + o];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            c->n_kept_blocks += cnt_red[0];
            c->n_bound_blocks += cnt_red[TPB];
            c->n_prune_mfma += cnt_red[2 * TPB];
            c->n_certified += cnt_red[3 * TPB];
        }
    }
    if (threadIdx.x == 0) {
        do_move = 0;
        Job &j = c->job;
        c->n_refresh = 0;               // (a consumed or idle step must not re-run a refresh)
        if (c->skip_apply) {
            c->skip_apply = 0;          // the resolver already consumed this step
        } else if (j.mode != MODE_DONE) {
            c->n_steps += 1;
            c->n_score_launches += 1;
            c->n_scored += (j.win_hi - j.pos) * (long long)(j.mode == MODE_FRESH ? j.K : j.n_dirty);
            const unsigned long long fm = c->first_mover;
            c->first_mover = kNoMover;
            const bool was_pruned = job_is_pruned(d, j.mode, j.prune);
            if (was_pruned) {
                // this step's bucket / table kernels have run (or were skipped as still valid);
                // a lean step queues neither
                if (!d.lean_step) c->tables_valid = 1;
                // (the sort of a certifying sweep holds only the rows certify_kernel left: not reusable)
                c->wsort_valid = d.use_certify ? 0 : 1; c->wsort_base = j.win_base; c->wsort_hi = j.win_hi;
            }
            if (fm == kNoMover) {
                // every visit of the window keeps its component: the state is untouched
                c->lik_evals += (j.win_hi - j.pos) * (long long)j.K;
                // (the mover-free stretch seen so far is a lower bound of the next distance between
                // movers: a chain that has come to rest finds its way back to the long windows)
                const double since = (double)(j.win_hi - c->last_mover);
                if (since > c->ema_run) c->ema_run = since;
                if (j.pos == j.win_base) {           // a clean window: be more optimistic
                    long long w = 2ll * c->win_size;
                    c->win_size = (int)(w > c->win_cap ? c->win_cap : w);
                }
                start_window(d, c, j.win_hi);
            } else {
                const long long p = (long long)fm;
                c->lik_evals += (p - j.pos) * (long long)j.K;
                mp.i = d.order ? d.order[p] : p;
                const int lab = d.choice[p - j.win_base];
                mp.sub_slot = plan_unseat(d, c, mp.i);
                mp.add_slot = -1;
                c->lik_evals += j.K;                 // K after the removal
                const int rc = plan_seat(d, c, mp.i, lab, mp);
                if (rc != 0) {
                    atomicCAS(&c->error, 0, rc);
                    j.mode = MODE_DONE;
                } else {
                    do_move = 1;
                    c->tables_valid = 0;
                    c->wsort_valid = 0;
                    c->state_epoch += 1;
                    set_refresh(d, c, mp, true);
                    c->n_moves += 1;
                    // adaptive window: about half the running mean distance between movers
                    const double run = (double)(p - c->last_mover);
                    c->last_mover = p;
                    c->ema_run = ema_after_mover(c->ema_run, run);
                    const long long w = window_for_rate(c);
                    c->win_size = (int)w;
                    // movers are dense relative to what is left of this window: give up its tail
                    // (it is re-scored later as part of a fresh, smaller window) instead of
                    // re-evaluating all of it after every move
                    if (j.win_hi - (p + 1) > 2 * w) j.win_hi = p + 1 + w;
                    if (p + 1 >= j.win_hi || was_pruned) {   // (a pruned window ends at its first move)
                        start_window(d, c, p + 1);
                    } else {
                        j.pos = p + 1;
                        j.mode = MODE_PARTIAL;
                        j.n_dirty = c->n_refresh;
                        j.dirty[0] = c->refresh[0];
                        j.dirty[1] = c->refresh[1];
                        set_chunks(d, j);
                    }
                }
            }
        }
    }
    if (d.publish) {
        if (threadIdx.x == 0) __threadfence();
        publish_ctrl_block(d);
    }
    __syncthreads();
    if (do_move) apply_rank1(d, mp);
}

void launch_apply(const Dev &d, hipStream_t st) {
    hipLaunchKernelGGL(apply_kernel, dim3(1), dim3(TPB), 0, st, d);
}

// ------------------------------------------------------------------------------------------
// Sequential sweep for tiny dimensions (D <= 4, full covariance): ONE workgroup walks the N visits
// in order, as the reference's loop does (igmm/crpmm.py:57-88, igmm/pcrpmm.py:93-131).
// With D(D+1)/2 + D multiply-adds per (visit, component) there is nothing to tile; what a visit
// costs is the latency of one dependent chain (~400 instructions: a logarithm, an exponential, two
// divisions, three wave reductions), so everything it touches is kept where latency is short:
//   * the state of every ACTIVE LABEL lives in LDS for the whole sweep (statistics m, S; inverse
//     factor, Winv mu, predictive constants; slot id, count), struct-of-arrays with lane = label;
//     global memory gets z[i] at every move and the labels' state once, at the end;
//   * the per-visit inputs (index, row of X, home slot, log prior, uniform) sit in an LDS ring that
//     wave 0 refills 64 visits at a time, a batch ahead of their use.  z[i] may be fetched ahead
//     because only the visit of i itself writes it: the host takes this path only when the visiting
//     order is a permutation;
//   * kSeqWaves wavefronts evaluate kSeqWaves consecutive visits side by side against the same
//     state; the visits in front of the first one that does not stay are exact as they are (a stay
//     changes nothing), that one is applied by its own wavefront -- statistics with the roundings of
//     apply_rank1, the two touched labels rebuilt from scratch (Cholesky of S_N, its inverse, the
//     constants), one lane each -- and the round restarts behind it.  The windowed path's
//     speculation at the scale of a workgroup: no launches, no global round trips.
// When the labels outgrow the LDS plan (K + 1 > cap) the kernel opens a window at the next visit
// and returns; the host carries on with the windowed kernels (the global state is complete).
// ------------------------------------------------------------------------------------------
template <int DD>
struct SeqLayout {   // SoA fields, in units of `cap` doubles
    static constexpr int T = DD * (DD + 1) / 2;
    static constexpr int OM = 0, OS = DD, OW = DD + T, OCV = DD + 2 * T, OC = 2 * DD + 2 * T, NF = OC + 12;
    // constants: 0 logseat, 1 A, 2 half_vd, 3 inv_cv, 4 logseat1, 5 A1, 6 half_vd1, 7 coef1, 8 a1,
    //            9 logdetC, 10 inv_lam, 11 mu2 (carried for the write-back)
};


__device__ __forceinline__ double readlane_f64(double v, int t) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), t);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), t);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ constexpr int seq_pk(int r, int l) { return r * (r + 1) / 2 + l; }

// Wave-wide reductions without LDS round trips (a visit is one long dependent chain: what counts is
// latency): all-reduce inside each row of 16 lanes with DPP, the four row results through v_readlane.
template <int CTRL>
__device__ __forceinline__ double seq_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double seq_wave_max(double v) {
    v = fmax(v, seq_dpp<0xB1>(v));      // quad_perm [1,0,3,2]
    v = fmax(v, seq_dpp<0x4E>(v));      // quad_perm [2,3,0,1]
    v = fmax(v, seq_dpp<0x141>(v));     // row_half_mirror
    v = fmax(v, seq_dpp<0x140>(v));     // row_mirror
    return fmax(fmax(readlane_f64(v, 0), readlane_f64(v, 16)), fmax(readlane_f64(v, 32), readlane_f64(v, 48)));
}
__device__ __forceinline__ double seq_wave_sum(double v) {
    v += seq_dpp<0xB1>(v);
    v += seq_dpp<0x4E>(v);
    v += seq_dpp<0x141>(v);
    v += seq_dpp<0x140>(v);
    return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}
// inclusive prefix sum over the 64 lanes (row_shr 1, 2, 4, 8 inside a row; row totals via v_readlane)
__device__ __forceinline__ double seq_wave_scan(double v, int lane) {
    v += seq_dpp<0x111>(v);
    v += seq_dpp<0x112>(v);
    v += seq_dpp<0x114>(v);
    v += seq_dpp<0x118>(v);
    const double t0 = readlane_f64(v, 15), t1 = readlane_f64(v, 31), t2 = readlane_f64(v, 47);
    const double t01 = t0 + t1;
    return v + (lane < 16 ? 0.0 : (lane < 32 ? t0 : (lane < 48 ? t01 : t01 + t2)));
}

// Derived state of one label from its statistics (registers: st[0..DD) = m, st[DD..DD+T) = packed
// lower triangle of S) and count n; written to the label's LDS fields.  Called by one lane per
// touched label, both lanes in lockstep.  tab: the table entries of count n.
template <int DD>
__device__ __forceinline__ void seq_rebuild_label(const Dev &d, double *F, int cap, int *Lver, int *Lnupd, int lab,
                                                  int n, const double *st, const SlotTab &tab) {
    using Ly = SeqLayout<DD>;
    const double k_N = d.k0 + (double)n;
    double mu[DD], A[DD][DD], W[DD][DD];
#pragma unroll
    for (int a = 0; a < DD; ++a) mu[a] = st[a] / k_N;
#pragma unroll
    for (int a = 0; a < DD; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) A[a][b] = st[DD + seq_pk(a, b)] - k_N * (mu[a] * mu[b]);
    double lam = 0.0;                                  // Gershgorin bound of lambda_max(S_N)
#pragma unroll
    for (int a = 0; a < DD; ++a) {
        double sacc = 0.0;
#pragma unroll
        for (int b = 0; b < DD; ++b) sacc += fabs(b <= a ? A[a][b] : A[b][a]);
        lam = fmax(lam, sacc);
    }
    bool bad = false;
    double piv_prod = 1.0;
#pragma unroll
    for (int j = 0; j < DD; ++j) {                     // right-looking Cholesky (slot_math.h: chol_inverse)
        const double djj = A[j][j];
        if (!(djj > 0.0)) bad = true;
        piv_prod *= djj;
        const double piv = sqrt(djj);
#pragma unroll
        for (int i = j + 1; i < DD; ++i) A[i][j] = A[i][j] / piv;
        A[j][j] = piv;
#pragma unroll
        for (int i = j + 1; i < DD; ++i)
#pragma unroll
            for (int l = j + 1; l <= i; ++l) A[i][l] = fma(-A[i][j], A[l][j], A[i][l]);
    }
    // logdet S_N = log of the product of the pivots: one logarithm on the chain instead of DD (the
    // sum of logs when the product leaves the comfortable range)
    double ldt;
    if (piv_prod > 1e-200 && piv_prod < 1e200) {
        ldt = log(piv_prod);
    } else {
        ldt = 0.0;
#pragma unroll
        for (int j = 0; j < DD; ++j) ldt += log(A[j][j]);
        ldt *= 2.0;
    }
    if (bad || !(ldt == ldt)) atomicCAS(&d.ctrl->error, 0, -4);
#pragma unroll
    for (int i = 0; i < DD; ++i) {                     // inverse of the factor, row by row
        const double inv_d = 1.0 / A[i][i];
#pragma unroll
        for (int cc = 0; cc <= i; ++cc) {
            double acc = 0.0;
#pragma unroll
            for (int t = cc; t < i; ++t) acc = fma(A[i][t], W[t][cc], acc);
            W[i][cc] = cc < i ? -acc * inv_d : inv_d;
        }
    }
    double mu2 = 0.0;
#pragma unroll
    for (int l = 0; l < DD; ++l) mu2 = fma(mu[l], mu[l], mu2);
    const SlotConst sc = make_consts_from(d, n, tab, ldt, lam, mu2);
#pragma unroll
    for (int j = 0; j < DD; ++j) {
        double acc = 0.0;
#pragma unroll
        for (int l = 0; l <= j; ++l) acc = fma(W[j][l], mu[l], acc);
        F[(Ly::OCV + j) * cap + lab] = acc;
#pragma unroll
        for (int l = 0; l <= j; ++l) F[(Ly::OW + seq_pk(j, l)) * cap + lab] = W[j][l];
    }
    double *C = F + Ly::OC * cap + lab;
    C[0] = sc.logseat; C[cap] = sc.A; C[2 * cap] = sc.half_vd; C[3 * cap] = sc.inv_cv; C[4 * cap] = sc.logseat1;
    C[5 * cap] = sc.A1; C[6 * cap] = sc.half_vd1; C[7 * cap] = sc.coef1; C[8 * cap] = sc.a1;
    C[9 * cap] = sc.logdetC; C[10 * cap] = sc.inv_lam; C[11 * cap] = sc.mu2;
    Lver[lab] += 1;
    Lnupd[lab] = 0;
}

#ifdef BGMM_SEQ_PROF
#define PF(k) { tk1 = clock64(); pf[k] += tk1 - tk0; tk0 = tk1; }
#else
#define PF(k)
#endif

constexpr int kSeqWaves = 8;      // visits evaluated side by side (one wavefront each)
constexpr int kSeqRing = 256;     // visits whose inputs sit in LDS (a power of two)

// LDS of the plan for `cap` labels
int sweep_seq_lds_bytes(int D, int cap) {
    const int T = D * (D + 1) / 2, NF = 2 * D + 2 * T + 12;
    return cap * ((NF + kSeqWaves) * (int)sizeof(double) + 4 * (int)sizeof(int)) +
           kSeqRing * ((3 + D) * (int)sizeof(double) + (int)sizeof(int));
}

template <int DD>
__global__ __launch_bounds__(64 * kSeqWaves) void sweep_seq_kernel(Dev d, int cap) {
    using Ly = SeqLayout<DD>;
    constexpr int T = Ly::T, NS = DD + T;               // NS: statistics per label (m, packed S)
    constexpr int NW = kSeqWaves, RING = kSeqRing, NT = 64 * NW;
    extern __shared__ double F[];
    double *eb_all = F + Ly::NF * cap;                 // weights of one visit per wave (K + 2 > 64 only)
    double *ring_u = eb_all + NW * cap, *ring_lp = ring_u + RING, *ring_x = ring_lp + RING;   // ring_x[DD][RING]
    long long *ring_i = (long long *)(ring_x + DD * RING);
    int *Lslot = (int *)(ring_i + RING);               // slot of label j; entries K .. K_hi: free slots (perm[j])
    int *Ln = Lslot + cap, *Lver = Ln + cap, *Lnupd = Lver + cap, *ring_z = Lnupd + cap;
    __shared__ int sh_res[2][NW];
    __shared__ int sh_K, sh_Khi, sh_stop, sh_moved;
    __shared__ long long sh_stop_at;
    Ctrl *c = d.ctrl;
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    double *eb = eb_all + w * cap;
    if (c->error != 0 || c->job.mode == MODE_DONE) return;        // (sweep_begin has opened the sweep)
    int K = c->job.K;
    if (K + 1 > cap) return;                           // the open window goes to the windowed kernels
    int K_hi = K < d.K_max ? K : K - 1;                // Lslot / Lver are valid for indices <= K_hi
    const long long N = d.N;
    for (int j = tid; j <= K_hi; j += NT) {
        const int s = d.perm[j];
        Lslot[j] = s;
        Lver[j] = d.mu_ver[s];
        if (j >= K) continue;
        Ln[j] = d.n[s];
        Lnupd[j] = d.nupd[s];
        const SlotConst sc = d.sc[s];
        double *C = F + Ly::OC * cap + j;
        C[0] = sc.logseat; C[cap] = sc.A; C[2 * cap] = sc.half_vd; C[3 * cap] = sc.inv_cv; C[4 * cap] = sc.logseat1;
        C[5 * cap] = sc.A1; C[6 * cap] = sc.half_vd1; C[7 * cap] = sc.coef1; C[8 * cap] = sc.a1;
        C[9 * cap] = sc.logdetC; C[10 * cap] = sc.inv_lam; C[11 * cap] = sc.mu2;
#pragma unroll
        for (int a = 0; a < DD; ++a) {
            F[(Ly::OM + a) * cap + j] = d.m[(long long)s * DD + a];
            F[(Ly::OCV + a) * cap + j] = d.cvec[(long long)s * d.Dp + a];
#pragma unroll
            for (int b = 0; b <= a; ++b) {
                F[(Ly::OS + seq_pk(a, b)) * cap + j] = d.S[(long long)s * DD * DD + a * DD + b];
                F[(Ly::OW + seq_pk(a, b)) * cap + j] = d.Wrm[(long long)s * DD * DD + a * DD + b];
            }
        }
    }
    double pri[NS];                                    // the prior a new component starts from
#pragma unroll
    for (int a = 0; a < DD; ++a) {
        pri[a] = d.prior_m[a];
#pragma unroll
        for (int b = 0; b <= a; ++b) pri[DD + seq_pk(a, b)] = d.prior_S[a * DD + b];
    }
    // Per-visit inputs: a ring of RING visits in LDS, filled 64 visits at a time by wave 0, whose
    // registers hold the next batch while its loads are in flight.
    long long n_i = 0; double n_x[DD], n_lp = 0.0, n_u = 0.0; int n_z = -1;
#pragma unroll
    for (int a = 0; a < DD; ++a) n_x[a] = 0.0;
#define SEQ_FETCH(PB)                                                              \
    {                                                                              \
        const long long p_ = (PB) + lane;                                          \
        if (p_ < N) {                                                              \
            n_i = d.order ? d.order[p_] : p_;                                      \
            n_u = d.u[p_];                                                         \
            n_z = d.z[n_i];                                                        \
            n_lp = d.log_prior[n_i];                                               \
            _Pragma("unroll") for (int a = 0; a < DD; ++a) n_x[a] = d.X[n_i * DD + a]; \
        }                                                                          \
    }
#define SEQ_COMMIT(PB)                                                             \
    {                                                                              \
        const int sl_ = (int)(((PB) + lane) & (RING - 1));                         \
        ring_i[sl_] = n_i; ring_u[sl_] = n_u; ring_lp[sl_] = n_lp; ring_z[sl_] = n_z; \
        _Pragma("unroll") for (int a = 0; a < DD; ++a) ring_x[a * RING + sl_] = n_x[a]; \
    }
    long long filled = 0;                              // (wave 0) visits [0, filled) have been in the ring
    if (w < 2) {
        SEQ_FETCH(64 * w)
        SEQ_COMMIT(64 * w)
    }
    if (w == 0) {
        filled = 128;
        SEQ_FETCH(filled)
    }
    if (tid == 0) { sh_K = K; sh_Khi = K_hi; sh_stop = 0; sh_stop_at = -1; sh_moved = 0; }
    __syncthreads();
    long long lik = 0, moves = 0;
    long long p = 0;
    int stop = 0, round = 0;
#ifdef BGMM_SEQ_PROF
    long long pf[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tk0 = clock64(), tk1;
#endif
    while (p < N) {
        if (w == 0 && filled < N && filled < p + 128) {
            SEQ_COMMIT(filled)
            filled += 64;
            SEQ_FETCH(filled)
        }
        const long long pv = p + w;
        int res = -1;                                  // -1: stays (or no visit); >= 0: moves to this label; -2: error
        int h = -1, lab_h = -1, nh = 0, L = K, pick = K;
        bool home_live = false, singleton = false;
        double x[DD];
        if (pv < N) {
            const int sl = (int)(pv & (RING - 1));
            h = __builtin_amdgcn_readfirstlane(ring_z[sl]);
            const double u = ring_u[sl], lp = ring_lp[sl];
#pragma unroll
            for (int a = 0; a < DD; ++a) x[a] = ring_x[a * RING + sl];
            PF(0)
            // the quadratic form of the lane's own label (does not wait for the home's label)
            double q_own = 0.0;
            {
                const int jo = lane < K ? lane : 0;
#pragma unroll
                for (int r = 0; r < DD; ++r) {
                    double acc = F[(Ly::OCV + r) * cap + jo];
#pragma unroll
                    for (int l = 0; l <= r; ++l) acc = fma(-F[(Ly::OW + seq_pk(r, l)) * cap + jo], x[l], acc);
                    q_own = fma(acc, acc, q_own);
                }
            }
            if (h >= 0) {
                for (int j0 = 0; j0 < K; j0 += 64) {
                    const unsigned long long mm = __ballot(j0 + lane < K && Lslot[j0 + lane] == h);
                    if (mm) { lab_h = j0 + __ffsll((long long)mm) - 1; break; }
                }
            }
            if (h >= 0 && lab_h < 0) {
                res = -2;                              // the label maps are broken
            } else {
                nh = h >= 0 ? __builtin_amdgcn_readfirstlane(Ln[lab_h]) : 0;
                home_live = h >= 0 && nh >= 2;         // removal keeps the component
                singleton = h >= 0 && nh == 1;         // removal deletes it (swap with last)
                L = singleton ? K - 1 : K;             // labels after the removal
                PF(1)
                pick = L;
                if (K + 2 <= 64) {
                    // Everything of the visit in registers, one label per lane: lane L is the new table and,
                    // for a live home, lane L + 1 evaluates the second logarithm of the home form (the
                    // frozen-factor downdate, choice_kernel), so that every lane runs ONE log.
                    const bool is_lab = lane < L, is_aux = home_live && lane == L + 1;
                    const int jj = is_aux ? lab_h : ((singleton && lane == lab_h) ? K - 1 : (is_lab ? lane : 0));
                    const double q_home = readlane_f64(q_own, lab_h >= 0 ? lab_h : 0);
                    const double q_last = readlane_f64(q_own, K >= 1 ? K - 1 : 0);
                    const double qv = is_aux ? q_home : ((singleton && lane == lab_h) ? q_last : q_own);
                    const double *C = F + Ly::OC * cap + jj;
                    const bool homeform = home_live && jj == lab_h && (is_lab || is_aux);
                    const double den = homeform ? 1.0 - C[8 * cap] * qv : 1.0;
                    const double num = homeform ? C[7 * cap] * qv : qv * C[3 * cap];
                    double arg = 1.0 + num / den;
                    double hv = homeform ? C[6 * cap] : C[2 * cap];
                    double base = homeform ? C[4 * cap] + C[5 * cap] : C[0] + C[cap];
                    if (is_aux) arg = den;
                    if (!is_lab) { hv = 0.0; base = lane == L ? d.log_alpha + lp : -INFINITY; if (!is_aux) arg = 1.0; }
                    const double lg = log(arg);
                    const double lg_aux = readlane_f64(lg, L + 1);
                    if (homeform && is_lab) base = base - 0.5 * lg_aux;
                    const double v = base - hv * lg;
                    const double mx = seq_wave_max(v);
                    const double e = exp(v - mx);
                    const double tot = seq_wave_sum(e);
                    const double cum = seq_wave_scan(e / tot, lane);
                    const unsigned long long mhit = __ballot(lane <= L && (u - cum) < 0.0);
                    if (mhit) pick = __ffsll((long long)mhit) - 1;
                } else {
                    // pass 1: log scores (as choice_kernel)
                    double mx = -INFINITY;
                    for (int j0 = 0; j0 <= L; j0 += 64) {
                        const int j = j0 + lane;
                        double v = -INFINITY;
                        if (j == L) {
                            v = d.log_alpha + lp;
                        } else if (j < L) {
                            const int jj = (singleton && j == lab_h) ? K - 1 : j;
                            double qv = 0.0;
#pragma unroll
                            for (int r = 0; r < DD; ++r) {
                                double acc = F[(Ly::OCV + r) * cap + jj];
#pragma unroll
                                for (int l = 0; l <= r; ++l) acc = fma(-F[(Ly::OW + seq_pk(r, l)) * cap + jj], x[l], acc);
                                qv = fma(acc, acc, qv);
                            }
                            const double *C = F + Ly::OC * cap + jj;
                            if (home_live && jj == lab_h) {
                                const double den = 1.0 - C[8 * cap] * qv;
                                v = C[4 * cap] + C[5 * cap] - 0.5 * log(den) - C[6 * cap] * log(1.0 + C[7 * cap] * qv / den);
                            } else {
                                v = C[0] + C[cap] - C[2 * cap] * log(1.0 + qv * C[3 * cap]);
                            }
                        }
                        if (j <= L) eb[j] = v;
                        mx = fmax(mx, v);
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o));
                    // pass 2: exp and total
                    double tot = 0.0;
                    for (int j = lane; j <= L; j += 64) {
                        const double e = exp(eb[j] - mx);
                        eb[j] = e;
                        tot += e;
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) tot += __shfl_xor(tot, o);
                    // pass 3: sequential-subtract scan in label order, 64 labels at a time (utils.py:15-20)
                    double carry = 0.0;
                    for (int j0 = 0; j0 <= L; j0 += 64) {
                        const int j = j0 + lane;
                        double cum = j <= L ? eb[j] / tot : 0.0;
#pragma unroll
                        for (int o = 1; o < 64; o <<= 1) {
                            const double tt = __shfl_up(cum, o);
                            if (lane >= o) cum += tt;
                        }
                        cum = carry + cum;
                        const unsigned long long mhit = __ballot(j <= L && (u - cum) < 0.0);
                        if (mhit) { pick = j0 + __ffsll((long long)mhit) - 1; break; }
                        carry = __shfl(cum, 63);
                    }
                }
                res = (home_live && pick == lab_h) ? -1 : pick;
                PF(2)
            }
        }
        // ---- which of the NW visits is the first that does not stay?  Everything behind it was
        // evaluated against a state that is about to change and is evaluated again.
        if (lane == 0) sh_res[round & 1][w] = res;
        __syncthreads();
        const int r_l = lane < NW ? sh_res[round & 1][lane] : -1;
        const unsigned long long mmov = __ballot(r_l != -1);
        const int f = mmov ? __ffsll((long long)mmov) - 1 : NW;
        round += 1;
        if (pv < N && w <= f) lik += L;
        if (f == NW) { p += NW; continue; }
        if (w == f) {
            // ---- this wave's visit moves (del_item / add_item, gaussian_components.py:168-205).  All of
            // it in LDS and registers; global memory sees z[i] now and the labels' state at the end.
            const bool add_init = pick >= L;                   // a new component
            if (res == -2) {
                if (lane == 0) { atomicCAS(&c->error, 0, -5); sh_stop = 1; }
            } else if (add_init && L >= d.K_max) {
                if (lane == 0) { atomicCAS(&c->error, 0, -3); sh_stop = 1; }
            } else {
                const int pre = add_init ? 0 : ((singleton && pick == lab_h) ? K - 1 : pick);   // destination, old numbering
                // the rebuild lanes (0: destination, 1: home) ask for their count's table entries first
                const int n_dst = add_init ? 1 : __builtin_amdgcn_readfirstlane(Ln[pre]) + 1;
                const int n_mine = lane == 0 ? n_dst : ((lane == 1 && home_live) ? nh - 1 : 0);
                SlotTab tab = {};
                if (n_mine > 0) tab = load_slot_tab(d, n_mine);
                // statistics with the roundings of apply_rank1
                double hs[NS], ds[NS];
                const int hl = home_live ? lab_h : 0;
#pragma unroll
                for (int e = 0; e < NS; ++e) { hs[e] = F[(Ly::OM + e) * cap + hl]; ds[e] = F[(Ly::OM + e) * cap + pre]; }
                const int tslot = __builtin_amdgcn_readfirstlane(add_init ? (singleton ? h : Lslot[K]) : Lslot[pre]);
                const long long i = ring_i[(int)(pv & (RING - 1))];
#pragma unroll
                for (int e = 0; e < NS; ++e) ds[e] = add_init ? pri[e] : ds[e];
#pragma unroll
                for (int a = 0; a < DD; ++a) {
                    hs[a] = __dsub_rn(hs[a], x[a]);
                    ds[a] = __dadd_rn(ds[a], x[a]);
#pragma unroll
                    for (int b = 0; b <= a; ++b) {
                        const double xx = __dmul_rn(x[a], x[b]);
                        hs[DD + seq_pk(a, b)] = __dsub_rn(hs[DD + seq_pk(a, b)], xx);
                        ds[DD + seq_pk(a, b)] = __dadd_rn(ds[DD + seq_pk(a, b)], xx);
                    }
                }
                PF(3)
                int sub_lab = -1;
                if (home_live) {
                    sub_lab = lab_h;
                    if (lane == 0) {
                        Ln[lab_h] = nh - 1;
#pragma unroll
                        for (int e = 0; e < NS; ++e) F[(Ly::OM + e) * cap + lab_h] = hs[e];
                    }
                } else if (singleton) {
                    // swap-with-last delete; the freed slot (and its version) stays at index `last`
                    const int last = K - 1;
                    if (lab_h != last) {
                        const int s_last = Lslot[last], v_last = Lver[last], v_h = Lver[lab_h];
                        const int n_last = Ln[last], u_last = Lnupd[last];
                        for (int fi = lane; fi < Ly::NF; fi += 64) F[fi * cap + lab_h] = F[fi * cap + last];
                        if (lane == 0) {
                            Lslot[lab_h] = s_last; Ln[lab_h] = n_last; Lver[lab_h] = v_last; Lnupd[lab_h] = u_last;
                            Lslot[last] = h; Lver[last] = v_h;
                        }
                    }
                    K = last;
                }
                int add_lab = pick;
                if (add_init) {
                    add_lab = K;
                    K += 1;
                    if (K > K_hi && K < d.K_max && K < cap) {  // the next free slot comes from global memory,
                        if (lane == 0) {                       // untouched there beyond K_hi
                            const int s2 = d.perm[K];
                            Lslot[K] = s2;
                            Lver[K] = d.mu_ver[s2];
                        }
                        K_hi = K;
                    }
                }
                if (lane == 0) {
                    Ln[add_lab] = n_dst;
#pragma unroll
                    for (int e = 0; e < NS; ++e) F[(Ly::OM + e) * cap + add_lab] = ds[e];
                    d.z[i] = tslot;
                }
                PF(4)
                // derived state of the touched labels, one lane each
                const int rl = lane == 0 ? add_lab : (lane == 1 ? sub_lab : -1);
                if (rl >= 0) {
                    double st[NS];
#pragma unroll
                    for (int e = 0; e < NS; ++e) st[e] = lane == 0 ? ds[e] : hs[e];
                    seq_rebuild_label<DD>(d, F, cap, Lver, Lnupd, rl, n_mine, st, tab);
                }
                PF(5)
                moves += 1;
                if (lane == 0) {
                    sh_K = K; sh_Khi = K_hi; sh_moved = 1;
                    if (K + 1 > cap) { sh_stop = 2; sh_stop_at = pv + 1; }   // the labels outgrew the LDS plan
                }
            }
        }
        __syncthreads();
        K = __builtin_amdgcn_readfirstlane(sh_K);
        K_hi = __builtin_amdgcn_readfirstlane(sh_Khi);
        stop = __builtin_amdgcn_readfirstlane(sh_stop);
        p += f + 1;
        if (stop) break;
    }
#undef SEQ_FETCH
#undef SEQ_COMMIT
    __syncthreads();
    // ---- write the labels' state back (slot order of the windowed kernels: bgmm_device.h)
    for (int j = tid; j <= K_hi; j += NT) {
        const int s = Lslot[j];
        d.perm[j] = s;
        d.label_of_slot[s] = j;
        d.mu_ver[s] = Lver[j];
        if (j >= K) { d.n[s] = 0; continue; }
        const int n = Ln[j];
        d.n[s] = n;
        d.nupd[s] = Lnupd[j];
        const double k_N = d.k0 + (double)n;
        const double *C = F + Ly::OC * cap + j;
        SlotConst sc;
        sc.logseat = C[0]; sc.A = C[cap]; sc.half_vd = C[2 * cap]; sc.inv_cv = C[3 * cap]; sc.logseat1 = C[4 * cap];
        sc.A1 = C[5 * cap]; sc.half_vd1 = C[6 * cap]; sc.coef1 = C[7 * cap]; sc.a1 = C[8 * cap];
        sc.logdetC = C[9 * cap]; sc.inv_lam = C[10 * cap]; sc.mu2 = C[11 * cap];
        d.sc[s] = sc;
#pragma unroll
        for (int a = 0; a < DD; ++a) {
            const double mv = F[(Ly::OM + a) * cap + j];
            d.m[(long long)s * DD + a] = mv;
            d.mu[(long long)s * DD + a] = mv / k_N;
            d.cvec[(long long)s * d.Dp + a] = F[(Ly::OCV + a) * cap + j];
#pragma unroll
            for (int b = 0; b < DD; ++b) {
                const int lo = a >= b ? seq_pk(a, b) : seq_pk(b, a);
                d.S[(long long)s * DD * DD + a * DD + b] = F[(Ly::OS + lo) * cap + j];
                d.Wrm[(long long)s * DD * DD + a * DD + b] = b <= a ? F[(Ly::OW + lo) * cap + j] : 0.0;
            }
        }
    }
    // the MFMA fragments of the factors (only a forced MFMA kernel reads them at this D)
    for (int j = w; j < K; j += NW) {
        double *wf = d.Wfrag + (long long)Lslot[j] * d.nfrag * 64;
        const int jr = lane & 15, lc = lane >> 4;
        double v0 = 0.0;
#pragma unroll
        for (int a = 0; a < DD; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b)
                if (jr == a && lc == b) v0 = -F[(Ly::OW + seq_pk(a, b)) * cap + j];
        wf[lane] = v0;
        for (int fi = 1; fi < d.nfrag; ++fi) wf[fi * 64 + lane] = 0.0;
    }
    if (lane == 0) {
        atomicAdd((unsigned long long *)&c->lik_evals, (unsigned long long)lik);
        atomicAdd((unsigned long long *)&c->n_scored, (unsigned long long)lik);
        atomicAdd((unsigned long long *)&c->n_moves, (unsigned long long)moves);
#ifdef BGMM_SEQ_PROF
        for (int k = 0; k < 8; ++k) atomicAdd((unsigned long long *)&c->prof[k], (unsigned long long)pf[k]);
#endif
    }
    __threadfence();
    __syncthreads();
    if (tid == 0) {
        const long long stop_at = sh_stop_at;
        c->job.K = K;
        c->n_steps += 1;
        c->n_score_launches += 1;
        if (sh_moved) { c->tables_valid = 0; c->wsort_valid = 0; c->state_epoch += 1; }
        if (stop == 2 && stop_at >= 0 && stop_at < N) {
            start_window(d, c, stop_at);               // the windowed kernels take it from here
        } else {
            Job &j = c->job;
            j.pos = N; j.win_base = N; j.win_hi = N; j.mode = MODE_DONE; j.n_dirty = 0; j.prune = 0;
        }
    }
}
#undef PF

template <int DD>
static hipError_t launch_seq_t(const Dev &d, int cap, int lds, hipStream_t st) {
    hipError_t e = hipFuncSetAttribute((const void *)sweep_seq_kernel<DD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sweep_seq_kernel<DD>, dim3(1), dim3(64 * kSeqWaves), lds, st, d, cap);
    return hipSuccess;
}

// cap: labels (plus the next free slot) the LDS plan holds.  Returns false when D is out of range.
bool launch_sweep_seq(const Dev &d, int cap, hipStream_t st) {
    const int lds = sweep_seq_lds_bytes(d.D, cap);
    switch (d.D) {
        case 1: return launch_seq_t<1>(d, cap, lds, st) == hipSuccess;
        case 2: return launch_seq_t<2>(d, cap, lds, st) == hipSuccess;
        case 3: return launch_seq_t<3>(d, cap, lds, st) == hipSuccess;
        case 4: return launch_seq_t<4>(d, cap, lds, st) == hipSuccess;
        default: return false;
    }
}

// op 0: del_item(i); op 1: add_item(i, label)
__global__ __launch_bounds__(TPB) void item_kernel(Dev d, int op, long long i, int label) {
    __shared__ MovePlan mp;
    __shared__ int ok;
    Ctrl *c = d.ctrl;
    if (threadIdx.x == 0) {
        mp.i = i; mp.sub_slot = -1; mp.add_slot = -1; mp.add_init = 0;
        ok = 1;
        c->tables_valid = 0;
        c->wsort_valid = 0;
        c->state_epoch += 1;
        if (op == 0) {
            mp.sub_slot = plan_unseat(d, c, i);
        } else {
            const int rc = (label < 0 || label > c->job.K) ? -1 : plan_seat(d, c, i, label, mp);
            if (rc != 0) { atomicCAS(&c->error, 0, rc); ok = 0; }
        }
        set_refresh(d, c, mp, false);
        if (!ok) c->n_refresh = 0;
    }
    __syncthreads();
    if (ok) apply_rank1(d, mp);
}

void launch_item_op(const Dev &d, int op, long long i, int label, hipStream_t st) {
    hipLaunchKernelGGL(item_kernel, dim3(1), dim3(TPB), 0, st, d, op, i, label);
}

// ------------------------------------------------------------------------------------------
// Read-outs
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void log_marg_kernel(Dev d, double *out_total, double *out_per_label) {
    const Ctrl *c = d.ctrl;
    const int K = c->job.K, D = d.D;
    const double hd = 0.5 * (double)D;
    const double logdetS0 = d.sc[d.K_max].logdetC;     // pseudo slot: C == S_0
    for (int j = threadIdx.x; j < K; j += TPB) {
        const int s = d.perm[j];
        const int n = d.n[s];
        if (d.cov_type == COV_FIXED) {      // gaussian_components_fixedvar.py:248-270 from the slot's sums
            const double Nk = (double)n;
            double acc = 0.0;
            for (int a = 0; a < D; ++a) {
                const double p = d.prior_S[D + a], p0 = d.prior_S[a], mu0 = d.fv_mu0[a];
                const double sx = (d.m[(long long)s * D + a] - d.prior_m[a]) / p;
                const double sxx = d.S[(long long)s * 2 * D + D + a];
                const double den = Nk / p0 + 1.0 / p;
                acc += (Nk - 1.0) / 2.0 * log(p) - 0.5 * Nk * log(2.0 * 3.14159265358979323846)
                       - 0.5 * log(den) - 0.5 * p * sxx - 0.5 * p0 * (mu0 * mu0)
                       + 0.5 * ((sx * sx) * p / p0 + (mu0 * mu0) * p0 / p + 2.0 * sx * mu0) / den;
            }
            out_per_label[j] = acc;
            continue;
        }
        const double k_N = d.k0 + (double)n;
        const long long v_N = d.v0 + n;
        double gs = 0.0;
        if (d.cov_type == COV_DIAG) gs = (double)D * (d.tab_lgam[v_N] - d.tab_lgam[d.v0]);
        else
            for (int t = 1; t <= D; ++t) gs += d.tab_lgam[v_N + 1 - t] - d.tab_lgam[d.v0 + 1 - t];
        out_per_label[j] = -(double)n * hd * BGMM_LOG_PI + hd * log(d.k0) - hd * log(k_N)
                           + 0.5 * (double)d.v0 * logdetS0 - 0.5 * (double)v_N * d.sc[s].logdetC + gs;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double sum_n = 0.0, sum_lf = 0.0, px = 0.0;
        for (int j = 0; j < K; ++j) {
            const int n = d.n[d.perm[j]];
            sum_n += (double)n;
            if (n > 0) sum_lf += lgamma((double)n);
            px += out_per_label[j];
        }
        const double pz = (double)(K - 1) * d.log_alpha + lgamma(d.alpha) - lgamma(sum_n + d.alpha) + sum_lf;
        *out_total = pz + px;
    }
}

void launch_log_marg(const Dev &d, double *out_total, double *out_per_label, hipStream_t st) {
    hipLaunchKernelGGL(log_marg_kernel, dim3(1), dim3(TPB), 0, st, d, out_total, out_per_label);
}

__global__ void labels_kernel(Dev d, long long *z_out, long long *counts_out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (z_out && i < d.N) {
        const int s = d.z[i];
        z_out[i] = s < 0 ? -1 : d.label_of_slot[s];
    }
    if (counts_out && i < d.ctrl->job.K) counts_out[i] = d.n[d.perm[i]];
}

void launch_labels(const Dev &d, long long *z_out, long long *counts_out, hipStream_t st) {
    const long long n = d.N > d.K_max ? d.N : d.K_max;
    hipLaunchKernelGGL(labels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d, z_out, counts_out);
}

__global__ void prior_lp_kernel(Dev d, const double *__restrict__ qcol) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.N) return;
    if (d.cov_type != COV_FULL) { d.log_prior[i] = qcol[i]; return; }   // these kernels emit log densities
    const SlotConst c = d.sc[d.K_max];
    d.log_prior[i] = c.A - c.half_vd * log(1.0 + qcol[i] * c.inv_cv);
}

void launch_prior_lp(const Dev &d, const double *qcol, hipStream_t st) {
    hipLaunchKernelGGL(prior_lp_kernel, dim3((unsigned)((d.N + 255) / 256)), dim3(256), 0, st, d, qcol);
}

__global__ void post_pred_kernel(Dev d, const double *__restrict__ qrow, double *out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= d.ctrl->job.K) return;
    const int s = d.perm[j];
    if (d.cov_type != COV_FULL) { out[j] = qrow[s]; return; }
    const SlotConst c = d.sc[s];
    out[j] = c.A - c.half_vd * log(1.0 + qrow[s] * c.inv_cv);
}

void launch_post_pred(const Dev &d, const double *qrow, double *out, hipStream_t st) {
    hipLaunchKernelGGL(post_pred_kernel, dim3((unsigned)((d.K_max + 255) / 256)), dim3(256), 0, st, d, qrow, out);
}

// stats export in label order: m, S, logdet(covar), inv(covar) as the reference stores them
__global__ __launch_bounds__(TPB) void export_stats_kernel(Dev d, double *m_out, double *S_out,
                                                           double *logdet_out, double *inv_out) {
    const int j = blockIdx.x;
    if (j >= d.ctrl->job.K) return;
    const int s = d.perm[j], D = d.D;
    const int n = d.n[s];
    const double k_N = d.k0 + (double)n;
    if (d.cov_type == COV_FIXED) {       // mu_N_numerators, precision_Ns, log_prod_precision_preds, precision_preds
        for (int a = threadIdx.x; a < D; a += TPB) {
            if (m_out) m_out[(long long)j * D + a] = d.m[(long long)s * D + a];
            if (S_out) S_out[(long long)j * D + a] = d.S[(long long)s * 2 * D + a];
            if (inv_out) inv_out[(long long)j * D + a] = d.dw[(long long)s * D + a];
        }
        if (logdet_out && threadIdx.x == 0) logdet_out[j] = d.sc[s].A1;
        return;
    }
    if (d.cov_type == COV_DIAG) {        // m, S (K x D), log_prod_vars, inv_vars (K x D)
        for (int a = threadIdx.x; a < D; a += TPB) {
            if (m_out) m_out[(long long)j * D + a] = d.m[(long long)s * D + a];
            if (S_out) S_out[(long long)j * D + a] = d.S[(long long)s * D + a];
            if (inv_out) inv_out[(long long)j * D + a] = d.dw[(long long)s * D + a] * (double)(d.v0 + n);
        }
        if (logdet_out && threadIdx.x == 0) logdet_out[j] = d.sc[s].A1;
        return;
    }
    const double cs = (k_N + 1.0) / (k_N * (double)(d.v0 + n - D + 1));
    if (m_out) for (int a = threadIdx.x; a < D; a += TPB) m_out[(long long)j * D + a] = d.m[(long long)s * D + a];
    if (S_out) for (int e = threadIdx.x; e < D * D; e += TPB) S_out[(long long)j * D * D + e] = d.S[(long long)s * D * D + e];
    if (logdet_out && threadIdx.x == 0) logdet_out[j] = (double)D * log(cs) + d.sc[s].logdetC;
    if (inv_out) {
        const double *W = d.Wrm + (long long)s * D * D;
        for (int e = threadIdx.x; e < D * D; e += TPB) {
            const int a = e / D, b = e % D;
            double acc = 0.0;
            for (int t = (a > b ? a : b); t < D; ++t) acc = fma(W[t * D + a], W[t * D + b], acc);
            inv_out[(long long)j * D * D + e] = acc / cs;
        }
    }
}

void launch_export_stats(const Dev &d, int K, double *m_out, double *S_out, double *logdet_out,
                         double *inv_out, hipStream_t st) {
    if (K > 0) hipLaunchKernelGGL(export_stats_kernel, dim3(K), dim3(TPB), 0, st, d, m_out, S_out, logdet_out, inv_out);
}

void launch_init_labels(const Dev &d, const long long *z_in, int K_init, hipStream_t st) {
    const long long n = d.N > d.K_max ? d.N : d.K_max;
    hipLaunchKernelGGL(init_labels_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d, z_in, K_init);
}


// ---- record-dict metrics --------------------------------------------
__global__ void contingency_kernel(Dev d, const long long *__restrict__ true_idx, int K_true,
                                   unsigned long long *__restrict__ table) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= d.N) return;
    const int s = d.z[i];
    const long long t = true_idx[i];
    if (s < 0 || t < 0 || t >= K_true) return;
    atomicAdd(&table[t * d.ctrl->job.K + d.label_of_slot[s]], 1ull);
}

void launch_contingency(const Dev &d, const long long *true_idx, int K_true, unsigned long long *table,
                        hipStream_t st) {
    hipLaunchKernelGGL(contingency_kernel, dim3((unsigned)((d.N + 255) / 256)), dim3(256), 0, st, d, true_idx,
                       K_true, table);
}

// sum_i |x_i - mean|^2 = sum_d (sum x_d^2 - (sum x_d)^2 / n), with the prior's share removed from
// the stored statistics (m = k_0 m_0 + sum x,  S_dd = S_0,dd + k_0 m_0,d^2 + sum x_d^2)
__global__ __launch_bounds__(256) void dispersion_kernel(Dev d, double *__restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= d.ctrl->job.K) return;
    const int s = d.perm[j], D = d.D;
    const double n = (double)d.n[s];
    const long long sd = d.cov_type == COV_DIAG ? 1 : (long long)D + 1;      // stride of the diagonal
    const long long blk = d.cov_type == COV_DIAG ? (long long)D : (long long)D * D;
    double acc = 0.0;
    if (d.cov_type == COV_FIXED) {
        for (int a = 0; a < D; ++a) {
            const double sx = (d.m[(long long)s * D + a] - d.prior_m[a]) / d.prior_S[D + a];
            acc += d.S[(long long)s * 2 * D + D + a] - sx * sx / n;
        }
        out[j] = acc > 0.0 ? acc : 0.0;
        return;
    }
    for (int a = 0; a < D; ++a) {
        const double sx = d.m[(long long)s * D + a] - d.prior_m[a];
        const double sxx = d.S[(long long)s * blk + a * sd] - d.prior_S[a * sd];
        acc += sxx - sx * sx / n;
    }
    out[j] = acc > 0.0 ? acc : 0.0;
}

void launch_dispersion(const Dev &d, double *out, hipStream_t st) {
    hipLaunchKernelGGL(dispersion_kernel, dim3((unsigned)((d.K_max + 255) / 256)), dim3(256), 0, st, d, out);
}

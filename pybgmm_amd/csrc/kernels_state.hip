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
#include "refresh_blocked.h"

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
int refresh_lds_bytes(int D, int cov_type) {
    const int Dp = (D + 15) / 16 * 16;
    const int diag = 2 * 256 * (int)sizeof(double);
    if (cov_type != COV_FULL) return diag;              // (D-vector state: two reduction arrays, whatever D)
    if (D > BGMM_FAST_MAX_D) return diag;               // (the factor does not fit: the kernels work in Dev::big_ws)
    const int rank1 = (D * (D + 1) + 6 * D + 4) * (int)sizeof(double);
    const int blocked = refresh_blocked_lds_doubles(Dp) * (int)sizeof(double);
    int v = rank1 > diag ? rank1 : diag;
    return v > blocked ? v : blocked;
}

// doubles of scratch a rebuild / rank-1 workgroup needs (LDS up to D = 128, Dev::big_ws beyond)
long long refresh_ws_doubles(int D) {
    const int Dp = (D + 15) / 16 * 16;
    const long long rank1 = (long long)D * (D + 1) + 6 * D + 4, blocked = refresh_blocked_lds_doubles(Dp);
    return ((rank1 > blocked ? rank1 : blocked) + 63) / 64 * 64;
}
// the workgroup's scratch: its LDS, or -- BIG: the factor too large for it -- its stripe of the global workspace.  (A template
// parameter, not a run-time choice: with one the compiler could no longer tell that the small route's pointers are LDS.)
template <bool BIG>
__device__ __forceinline__ double *refresh_scratch(const Dev &d, double *lds) {
    if (BIG) return d.big_ws + (long long)blockIdx.x * d.big_ws_stride;
    return lds;
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

template <bool BIG>
__global__ __launch_bounds__(TPB) void refresh_list_kernel(Dev d, const int *__restrict__ slots, int n) {
    extern __shared__ __attribute__((aligned(16))) double sm_lds[];
    double *const sm = refresh_scratch<BIG>(d, sm_lds);
    if ((int)blockIdx.x >= n) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) { d.ctrl->tables_valid = 0; d.ctrl->wsort_valid = 0; d.ctrl->state_epoch += 1; }
    const int s = slots ? slots[blockIdx.x] : (int)blockIdx.x;
    if (d.cov_type == COV_DIAG) refresh_diag_slot<TPB>(d, s, sm, threadIdx.x);
    else if (d.cov_type == COV_FIXED) refresh_fixed_slot<TPB>(d, s, sm, threadIdx.x);
    else if (BIG) refresh_slot(d, s, sm);              // (the blocked rebuild holds its columns in registers: 8 blocks at most)
    else refresh_slot_blocked(d, s, sm);
}

// From-scratch rebuild of every live slot that has taken rank-1 steps since its last rebuild.  Run
// once at the start of a sweep that follows a sweep with moves: the eigenvalue bound behind the
// pruning only ever grows under rank-1 steps (slot_math.h: lam_after_rank1), so after a burn-in it
// would stay loose until a slot's 64th update.
template <bool BIG>
__global__ __launch_bounds__(TPB) void refresh_stale_kernel(Dev d) {
    extern __shared__ __attribute__((aligned(16))) double sm_lds[];
    double *const sm = refresh_scratch<BIG>(d, sm_lds);
    if ((int)blockIdx.x >= d.ctrl->job.K) return;
    const int s = d.perm[blockIdx.x];
    if (d.nupd[s] == 0) return;
    if (threadIdx.x == 0) { d.ctrl->tables_valid = 0; atomicAdd((unsigned long long *)&d.ctrl->state_epoch, 1ull); }   // (means and bounds change; the homes do not)
    if (BIG) refresh_slot(d, s, sm);
    else refresh_slot_blocked(d, s, sm);
}

template <bool BIG>
__global__ __launch_bounds__(TPB) void refresh_ctrl_kernel(Dev d) {
    extern __shared__ __attribute__((aligned(16))) double sm_lds[];
    double *const sm = refresh_scratch<BIG>(d, sm_lds);
    const Ctrl *c = d.ctrl;
    if ((int)blockIdx.x >= c->n_refresh) return;
    const int s = c->refresh[blockIdx.x], kind = c->refresh_kind[blockIdx.x];
    if (d.cov_type == COV_DIAG) refresh_diag_slot<TPB>(d, s, sm, threadIdx.x);
    else if (d.cov_type == COV_FIXED) refresh_fixed_slot<TPB>(d, s, sm, threadIdx.x);
    else if (kind == REFRESH_SCRATCH) { if (BIG) refresh_slot(d, s, sm); else refresh_slot_blocked(d, s, sm); }
    else rank1_slot(d, kind == REFRESH_NEW ? d.K_max : s, s, c->refresh_i, kind, sm);
}

static void ensure_lds(const void *fn, int bytes, PerDeviceLds &attr) {
    attr.ensure(fn, bytes);
}

// Frozen-factor windows (kernels_gram.hip), last kernel of a step: block b brings slot gtouched[b] up to
// date -- the window's logged moves replayed on (m, S) in visiting order with the roundings of
// apply_rank1 (bit-identical statistics), then the derived state.
// Two routes for the derived state.  A slot that took a few terms (the usual case: a window of 64 moves spreads 128
// terms over ~100 slots) follows them by RANK-1 steps of its inverse factor, one per term, O(D^2) each (slot_math.h:
// rank1_inverse_factor -- the route of the per-mover kernels), the factor in LDS throughout; the from-scratch
// factorisation (refresh_blocked.h, O(D^3): 52 000 of the kernel's 90 000 cycles at D = 64) is kept for slots that took
// many terms, were opened by the window, or are due (kGramRefreshEvery rank-1 steps since their last rebuild; bgmm_device.h).
// The slot's statistics stay in REGISTERS across its terms (D^2 / 256 doubles per thread): one read and one write
// of S per window instead of one per term.
// SREGS: registers a thread holds of the slot's statistics, D 2^ceil(log2 D) / 256 (16 up to D = 64: two workgroups per compute unit)
template <int SREGS>
__device__ __forceinline__ void gram_finish_body(const Dev &d) {
    extern __shared__ __attribute__((aligned(16))) double sm[];
    __shared__ long long mv_i[kGramMaxTerms];
    __shared__ int mv_op[kGramMaxTerms];          // 1: x leaves this slot, 2: joins it, 3: joins and opens it
    __shared__ int n_ops, any_open;
    __shared__ double xs[BGMM_MAX_D], ms[BGMM_MAX_D], wide_scan[256];
    if ((int)blockIdx.x >= d.gfin[0]) return;
    const int s = d.gtouched[blockIdx.x];
    // (the look-ahead of the dense proof pass re-scores a label iff a window closed after its chunk's request changed it)
    if (threadIdx.x == 0 && d.touch_seq) d.touch_seq[s] = d.ctrl->win_seq;
    if (d.pipe) {              // (the slot's count behind this window: the resolver left it on the list, see there)
        const int n_now = d.gfin[16 + blockIdx.x];
        if (threadIdx.x == 0) d.n[s] = n_now;
        if (n_now <= 0) return;                    // (the window deleted this component: its count is all there is to write)
        __syncthreads();
    }
    const int D = d.D, nm = d.gfin[1], tid = threadIdx.x;
    // this slot's moves, in visiting order (one load per thread, compacted by a ballot scan)
    if (tid == 0) { n_ops = 0; any_open = 0; }
    __syncthreads();
    for (int k0 = 0; k0 < nm; k0 += TPB) {
        const int k = k0 + tid;
        int op = 0;
        long long i = 0;
        if (k < nm) {
            const GramMove mv = d.gmoves[k];
            op = mv.sub_slot == s ? 1 : (mv.add_slot == s ? (mv.add_init ? 3 : 2) : 0);
            i = mv.i;
        }
        // (nm <= 64 moves per window: the first wavefront holds them all, in order)
        const unsigned long long m = __ballot(op != 0);
        if (tid < 64 && op != 0) {
            const int pos = n_ops + __popcll(m & ((1ull << (tid & 63)) - 1ull));
            mv_i[pos] = i; mv_op[pos] = op;
            if (op == 3) any_open = 1;
        }
        __syncthreads();
        if (tid == 0) n_ops += __popcll(m);
        __syncthreads();
    }
    const int nops = n_ops;
    double *m = d.m + (long long)s * D;
    double *S = d.S + (long long)s * D * D;
    // rank-1 steps cost ~3.5 us each at D = 64 against ~37 us for the factorisation
    const int few = D >= 24 ? D / 12 : 1;
    const int steps = d.nupd[s];
    const bool rank1 = nops >= 1 && nops <= few && !any_open && steps + nops <= kGramRefreshEvery &&
                       !(d.gfin[2] && steps > kGramRefreshEvery / 2);
#ifdef BGMM_PROFILE
    long long fk0 = clock64(), fk1;
#define FPROF(i) do { if (tid == 0 && blockIdx.x == 0) { fk1 = clock64(); d.ctrl->prof[i] += fk1 - fk0; fk0 = fk1; } } while (0)
#else
#define FPROF(i) do { } while (0)
#endif
#if defined(BGMM_FINISH_DIAG) && !defined(BGMM_PROFILE)
    // (diagnostics, off by default: why slots are rebuilt from scratch.  They share Ctrl::prof with the phase clocks of a
    //  -DBGMM_PROFILE build and cost every workgroup of this kernel contended global atomics, so a production build has
    //  neither: bgmm_get_phase_clocks returns zeros, as include/bgmm.h says)
    if (tid == 0) {
        Ctrl *cw = d.ctrl;
        atomicAdd((unsigned long long *)&cw->prof[13], 1ull);
        if (!rank1) atomicAdd((unsigned long long *)&cw->prof[14], 1ull);
        if (nops > few) atomicAdd((unsigned long long *)&cw->prof[15], 1ull);
        if (blockIdx.x == 0) atomicAdd((unsigned long long *)&cw->prof[8], 1ull);
    }
#endif
    if (!rank1) {
        for (int k = 0; k < nops; ++k) {
            const int op = mv_op[k];
            const double *__restrict__ x = d.X + mv_i[k] * D;
            for (int a = tid; a < D; a += TPB) {
                if (op == 1) m[a] = __dsub_rn(m[a], x[a]);
                else m[a] = __dadd_rn(op == 3 ? d.prior_m[a] : m[a], x[a]);
            }
            for (int e = tid; e < D * D; e += TPB) {
                const double xx = __dmul_rn(x[e / D], x[e % D]);
                if (op == 1) S[e] = __dsub_rn(S[e], xx);
                else S[e] = __dadd_rn(op == 3 ? d.prior_S[e] : S[e], xx);
            }
        }
        __syncthreads();
        refresh_slot_blocked(d, s, sm);
        return;
    }
    // ---- the rank-1 route ------------------------------------------------------------------------------------
    const int ld = D + 1;
    double *W = sm;
    double *mu = sm + D * ld, *dv = mu + D, *pv = dv + D, *lv = pv + D, *tv = lv + D;
    double *scal = tv + 2 * D;
    // thread -> (row, column) of S and W without a division: column sb = tid mod 2^lg (2^lg >= D), rows sa0 + q * srows
    constexpr int kSregs = SREGS;
    int lg = 4;
    while ((1 << lg) < D) ++lg;
    const int sb = tid & ((1 << lg) - 1), sa0 = tid >> lg, srows = TPB >> lg;
    const bool scol = sb < D;
    double Sreg[kSregs];
#pragma unroll
    for (int q = 0; q < kSregs; ++q) { const int a = sa0 + q * srows; Sreg[q] = (scol && a < D) ? S[a * D + sb] : 0.0; }
    {
        const double *__restrict__ Wsrc = d.Wrm + (long long)s * D * D;
        for (int a = sa0; a < D; a += srows)
            if (scol) W[a * ld + sb] = Wsrc[a * D + sb];
    }
    if (tid < D) ms[tid] = m[tid];
    __syncthreads();
    FPROF(10);
    int n_cur = d.n[s];                               // the count BEHIND the window (the resolver has written it) ...
    for (int k = 0; k < nops; ++k) n_cur += mv_op[k] == 1 ? 1 : -1;     // ... and in front of it
    double logdet = d.sc[s].logdetC;
    bool bad = false;
    // (the rows of the first terms set off together: one round trip to memory for all of them)
    double xpre[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) xpre[k] = (k < nops && tid < D) ? d.X[mv_i[k] * D + tid] : 0.0;
    for (int k = 0; k < nops; ++k) {
        const int op = mv_op[k];
        double xv = 0.0;
        if (tid < D) {
            xv = k < 4 ? (k == 0 ? xpre[0] : (k == 1 ? xpre[1] : (k == 2 ? xpre[2] : xpre[3]))) : d.X[mv_i[k] * D + tid];
            xs[tid] = xv;
        }
        const double k_before = d.k0 + (double)n_cur;
        if (tid < D) {
            dv[tid] = xv - ms[tid] / k_before;                       // x - mean of the component before the change
            ms[tid] = op == 1 ? __dsub_rn(ms[tid], xv) : __dadd_rn(ms[tid], xv);
        }
        const double a = op == 1 ? -k_before / (k_before - 1.0) : k_before / (k_before + 1.0);
        n_cur += op == 1 ? -1 : 1;
        __syncthreads();
        {
            const double xb = scol ? xs[sb] : 0.0;
#pragma unroll
            for (int q = 0; q < kSregs; ++q) {
                const int a = sa0 + q * srows;
                if (a < D) {
                    const double xx = __dmul_rn(xs[a], xb);
                    Sreg[q] = op == 1 ? __dsub_rn(Sreg[q], xx) : __dadd_rn(Sreg[q], xx);
                }
            }
        }
        if (tid == 0) *(int *)&scal[1] = 0;
        rank1_inverse_factor_wide(W, ld, D, a, dv, pv, lv, tv, wide_scan, &scal[0], (int *)&scal[1], tid);
        // (every thread reads the step's outcome behind the routine's closing barrier)
        bad = bad || *(int *)&scal[1] != 0;
        logdet += log(1.0 + a * scal[0]);
        __syncthreads();                                             // (xs, dv, scal are rewritten by the next term)
    }
    FPROF(11);
#pragma unroll
    for (int q = 0; q < kSregs; ++q) { const int a = sa0 + q * srows; if (scol && a < D) S[a * D + sb] = Sreg[q]; }
    if (tid < D) {
        m[tid] = ms[tid];
        mu[tid] = ms[tid] / (d.k0 + (double)d.n[s]);
    }
    if (tid == 0 && bad) atomicCAS(&d.ctrl->error, 0, -4);
    // The eigenvalue bound behind the pruning: the Gershgorin row sums of S_N = S - k_N mu mu', as the from-scratch route
    // takes them -- from the statistics in registers.  (Carried through the rank-1 steps it could only grow: a component
    // that has just lost a far outlier would keep the outlier's bound, and the pruned windows their loose exclusions.)
    double *rowsum = pv;
    if (tid < D) rowsum[tid] = 0.0;
    __syncthreads();
    {
        const double kN = d.k0 + (double)d.n[s];
        const double mub = scol ? mu[sb] : 0.0;
        const int width = lg < 6 ? (1 << lg) : 64;                     // lanes of a wavefront that share a row
#pragma unroll
        for (int q = 0; q < kSregs; ++q) {
            const int a = sa0 + q * srows;
            if (a < D) {                                               // (uniform per wavefront: a row spans whole wavefronts or parts of one)
                double v = scol ? fabs(Sreg[q] - kN * (mu[a] * mub)) : 0.0;
                for (int o = 1; o < width; o <<= 1) v += __shfl_xor(v, o);
                if ((tid & (width - 1)) == 0) atomicAdd(&rowsum[a], v);
            }
        }
    }
    __syncthreads();
    double lam_g = 0.0;
    if (tid < 64) {
        for (int l = tid; l < D; l += 64) lam_g = fmax(lam_g, rowsum[l]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) lam_g = fmax(lam_g, __shfl_xor(lam_g, o));
    }
    __syncthreads();                                                   // (rowsum = pv is free again; thread 0 holds the bound)
    write_slot_wide(d, s, W, ld, mu, logdet, lam_g, tid);
    if (tid == 0) d.nupd[s] += nops;
    __syncthreads();
    FPROF(12);
}
template <int SREGS>
__global__ __launch_bounds__(TPB) void gram_finish_kernel(Dev d) { gram_finish_body<SREGS>(d); }
// (several chains in one launch: workgroup (x, c) works for chain group[c] -- bgmm_group_sweep_staged)
// (three workgroups per compute unit: G x ~90 slots per window want the room; what that costs -- spills on the rare
// from-scratch route -- the single-chain kernel above does not pay)
// (D > 64: the factor's LDS -- 138 KB at D = 128 -- admits ONE workgroup per compute unit whatever the registers say; asking
// for three there bought 368 spilled registers and 804 bytes of scratch per lane and nothing else -- round 6)
template <int SREGS>
__global__ __launch_bounds__(TPB, SREGS <= 16 ? 3 : 1) void gram_finish_group_kernel(const Dev *__restrict__ group) {
    const Dev d = group[blockIdx.y];               // (a private copy: nothing the body writes can alias it)
    gram_finish_body<SREGS>(d);
}


// (window k of a pipelined batch of several chains: kernels_gram.hip, gram_pgroup_view)
template <int SREGS>
__global__ __launch_bounds__(TPB, SREGS <= 16 ? 3 : 1) void gram_finish_pgroup_kernel(const Dev *__restrict__ v0, const Dev *__restrict__ v1, int k) {
    Dev d = ((k & 1) ? v1 : v0)[blockIdx.y];
    d.pipe = k == 0 ? 2 : 1;
    d.pipe_pos += (long long)kGramRows * k;
    gram_finish_body<SREGS>(d);
}
void launch_gram_finish_pgroup(const Dev &lead, const Dev *v0, const Dev *v1, int G, int k, hipStream_t st) {
    const int lds = refresh_lds_bytes(lead.D, lead.cov_type);
    static PerDeviceLds attr16, attr64;
    if (lead.D <= 64) {
        ensure_lds((const void *)gram_finish_pgroup_kernel<16>, lds, attr16);
        hipLaunchKernelGGL(gram_finish_pgroup_kernel<16>, dim3(kGramMaxTerms, G), dim3(TPB), lds, st, v0, v1, k);
    } else {
        ensure_lds((const void *)gram_finish_pgroup_kernel<64>, lds, attr64);
        hipLaunchKernelGGL(gram_finish_pgroup_kernel<64>, dim3(kGramMaxTerms, G), dim3(TPB), lds, st, v0, v1, k);
    }
}

void launch_gram_finish_group(const Dev &lead, const Dev *group, int G, hipStream_t st) {
    const int lds = refresh_lds_bytes(lead.D, lead.cov_type);
    static PerDeviceLds attr16, attr64;
    if (lead.D <= 64) {
        ensure_lds((const void *)gram_finish_group_kernel<16>, lds, attr16);
        hipLaunchKernelGGL(gram_finish_group_kernel<16>, dim3(kGramMaxTerms, G), dim3(TPB), lds, st, group);
    } else {
        ensure_lds((const void *)gram_finish_group_kernel<64>, lds, attr64);
        hipLaunchKernelGGL(gram_finish_group_kernel<64>, dim3(kGramMaxTerms, G), dim3(TPB), lds, st, group);
    }
}

void launch_gram_finish(const Dev &d, hipStream_t st) {
    const int lds = refresh_lds_bytes(d.D, d.cov_type);
    static PerDeviceLds attr16, attr64;
    if (d.D <= 64) {
        ensure_lds((const void *)gram_finish_kernel<16>, lds, attr16);
        hipLaunchKernelGGL(gram_finish_kernel<16>, dim3(kGramMaxTerms), dim3(TPB), lds, st, d);
    } else {
        ensure_lds((const void *)gram_finish_kernel<64>, lds, attr64);
        hipLaunchKernelGGL(gram_finish_kernel<64>, dim3(kGramMaxTerms), dim3(TPB), lds, st, d);
    }
}


void launch_refresh_list(const Dev &d, const int *slots, int n, hipStream_t st) {
    if (n <= 0) return;
    const int lds = refresh_lds_bytes(d.D, d.cov_type);
    static PerDeviceLds attr;
    ensure_lds((const void *)refresh_list_kernel<false>, lds, attr);
    if (d.big_ws) hipLaunchKernelGGL(refresh_list_kernel<true>, dim3(n), dim3(TPB), lds, st, d, slots, n);
    else hipLaunchKernelGGL(refresh_list_kernel<false>, dim3(n), dim3(TPB), lds, st, d, slots, n);
}

void launch_refresh_stale(const Dev &d, int K, hipStream_t st) {
    if (K <= 0) return;
    const int lds = refresh_lds_bytes(d.D, d.cov_type);
    static PerDeviceLds attr;
    ensure_lds((const void *)refresh_stale_kernel<false>, lds, attr);
    if (d.big_ws) hipLaunchKernelGGL(refresh_stale_kernel<true>, dim3(K), dim3(TPB), lds, st, d);
    else hipLaunchKernelGGL(refresh_stale_kernel<false>, dim3(K), dim3(TPB), lds, st, d);
}

void launch_refresh_ctrl(const Dev &d, hipStream_t st) {
    const int lds = refresh_lds_bytes(d.D, d.cov_type);
    static PerDeviceLds attr;
    ensure_lds((const void *)refresh_ctrl_kernel<false>, lds, attr);
    if (d.big_ws) hipLaunchKernelGGL(refresh_ctrl_kernel<true>, dim3(2), dim3(TPB), lds, st, d);
    else hipLaunchKernelGGL(refresh_ctrl_kernel<false>, dim3(2), dim3(TPB), lds, st, d);
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

__device__ __forceinline__ void sweep_begin_body(const Dev &d) {
    Ctrl *c = d.ctrl;
    const int K = c->job.K;
    // seating weights depend on the sweep's exponent (tabSeat was rebuilt by the host if it changed).  Every change of a
    // count goes through make_consts, which takes them from the table: while the exponent stands and the last sweep
    // moved nothing they are what they were (three dependent round trips to memory in front of every sweep of a chain
    // at rest otherwise).
    if (d.seat_dirty || c->n_visits == 0 || c->n_moves != 0)
        for (int j = threadIdx.x; j < K; j += TPB) {
            const int s = d.perm[j];
            d.sc[s].logseat = d.tabSeat[d.n[s]];
            d.sc[s].logseat1 = d.n[s] >= 1 ? d.tabSeat[d.n[s] - 1] : 0.0;
        }
    __syncthreads();                   // (thread 0 resets the counters the condition reads)
    if (threadIdx.x == 0) {
        // (a whole sweep without a move: the chain is at rest, back to the optimistic window plan it
        // started with -- the mover-free stretch inside one sweep cannot say more than N)
        if (c->n_visits > 0 && c->n_moves == 0 && c->ema_run < 4.0 * (double)c->win_cap)
            c->ema_run = 4.0 * (double)c->win_cap;
        c->n_visits = d.sweep_visits > 0 && d.sweep_visits < d.N ? d.sweep_visits : d.N;
        c->first_mover = kNoMover;
        c->n_refresh = 0;
        c->skip_apply = 0;
        c->lik_evals = 0; c->n_moves = 0; c->n_windows = 0; c->n_steps = 0;
        c->n_score_launches = 0; c->n_scored = 0;
        c->n_kept_blocks = 0; c->n_bound_blocks = 0; c->n_prune_mfma = 0; c->n_certified = 0;
        c->n_resid = 0; c->home_in = 0; c->home_out = 0;
        c->n_pairs_exact = 0; c->gram_rows_total = 0; c->gram_windows = 0; c->gram_ntouched = 0; c->gram_nmoves = 0;
        c->safe_windows = 0; c->safe_scanned = 0; c->safe_rows = 0; c->safe_cuts = 0; c->safe_resid_sum = 0; c->safe_sorted_sum = 0;
        c->safe_epoch_valid = 0;        // (new uniforms, maybe a new visiting order: the proofs were about the old ones)
        if (d.seat_dirty) { c->tables_valid = 0; c->state_epoch += 1; }   // (the tables carry log seating weights)
        if (d.order) c->wsort_valid = 0; // (a fresh permutation every sweep)
        c->last_mover = -1;
        if (c->win_size < 64) c->win_size = 64;
        if (c->win_size > c->win_cap) c->win_size = c->win_cap;
        if (c->error == 0) start_window(d, c, 0);
        else c->job.mode = MODE_DONE;
    }
}

__global__ __launch_bounds__(TPB) void sweep_begin_kernel(Dev d) { sweep_begin_body(d); }
// (several chains in one launch: workgroup b opens the sweep of chain group[b] -- bgmm_group_sweep_staged)
__global__ __launch_bounds__(TPB) void sweep_begin_group_kernel(const Dev *__restrict__ group) {
    const Dev d = group[blockIdx.x];
    sweep_begin_body(d);
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
    if (!job_is_pruned(d, c->job.mode, c->job.prune) || c->skip_sort || (d.safe_mode && c->safe_epoch_valid)) return;
    const long long base = c->job.win_base;
    const int nrows = (int)(c->job.win_hi - base);
    const int r0 = blockIdx.x * BUCKET_ROWS;
    if (r0 >= nrows) return;
    const int nb = d.nslots + 1;
    // A window that is the WHOLE sweep of a visiting order that comes by every point once sorts every point: the bins are
    // the components' counts, no pass over the rows needed (a chain at rest under a fresh permutation per sweep: a third of
    // the sort's time).
    if (d.order_perm && !d.use_certify && !d.safe_mode && base == 0 && c->job.win_hi == d.N && c->n_visits == d.N) {
        if (blockIdx.x != 0) return;
        __shared__ int asg[256];
        int mine = 0;
        for (int s = threadIdx.x; s < d.nslots; s += 256) { const int v = d.n[s]; d.bucket_bins[s + 1] = v; mine += v; }
        asg[threadIdx.x] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            long long tot = 0;
            for (int t = 0; t < 256; ++t) tot += asg[t];
            d.bucket_bins[0] = (int)(d.N - tot);          // the unassigned points
        }
        return;
    }
    for (int b = threadIdx.x; b < nb; b += 256) bins[b] = 0;
    __syncthreads();
    // (the four rows of a thread side by side: their index loads, then their label loads, are in flight together --
    // two dependent round trips per thread instead of eight)
    long long ii[BUCKET_ROWS / 256];
    int zz[BUCKET_ROWS / 256];
#pragma unroll
    for (int t = 0; t < BUCKET_ROWS / 256; ++t) {
        const int r = r0 + threadIdx.x + t * 256;
        const bool ok = r < nrows && !(d.use_certify && d.cert[r < nrows ? r : 0]);     // (proved to stay: not part of the sort)
        const long long p = base + (r < nrows ? r : 0);
        ii[t] = ok ? (d.order ? d.order[p] : p) : -1;
    }
#pragma unroll
    for (int t = 0; t < BUCKET_ROWS / 256; ++t) zz[t] = ii[t] >= 0 ? d.z[ii[t]] : -2;
#pragma unroll
    for (int t = 0; t < BUCKET_ROWS / 256; ++t)
        if (zz[t] >= -1) atomicAdd(&bins[zz[t] + 1], 1);
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
    if (!job_is_pruned(d, c->job.mode, c->job.prune) || c->tables_valid || (d.safe_mode && c->safe_epoch_valid)) return;
    const int n_tab_blocks = ((d.nslots + 15) / 16 + 15) / 16;
    if ((int)blockIdx.x >= n_tab_blocks) {
        __shared__ double mua[BGMM_MAX_D_DIAG];
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
        // the label's own scale: root mean square distance of its members to its mean, sqrt(tr S_N / n) -- what the
        // radius grid of its bound table is sized by (prune_ftable_kernel)
        if (threadIdx.x < 64) {
            const int sa = d.perm[a], na = d.n[sa];
            double tr = 0.0;
            if (d.cov_type != COV_FIXED && na >= 1) {
                const double kN = d.k0 + (double)na;
                const long long DD = d.cov_type == COV_FULL ? (long long)D * D : (long long)D, st = d.cov_type == COV_FULL ? D + 1 : 1;
                const double *__restrict__ Sa = d.S + (long long)sa * DD;
                for (int l = threadIdx.x; l < D; l += 64) tr += Sa[(long long)l * st] - kN * mua[l] * mua[l];
            }
            for (int o = 32; o > 0; o >>= 1) tr += __shfl_xor(tr, o);
            if (threadIdx.x == 0) d.pr_rms[a] = (tr > 0.0 && na >= 1) ? sqrt(tr / (double)na) : 0.0;
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
__global__ __launch_bounds__(256) void prune_ftable_kernel(Dev d) {
    __shared__ double red[4][64];
    __shared__ int nbl[4];
    const Ctrl *c = d.ctrl;
    if (!job_is_pruned(d, c->job.mode, c->job.prune) || c->tables_valid || (d.safe_mode && c->safe_epoch_valid)) return;
    // thread = (radius j, quarter of the labels): the other labels are dealt to the four wavefronts, a maximum each
    const int K = c->job.K, a = blockIdx.x, j = threadIdx.x & 63, part = threadIdx.x >> 6;
    if (a >= K) return;
    const double *__restrict__ dc = d.pr_dcc + (long long)a * d.nslots;
    double dmin = INFINITY;
    for (int t = j; t < K; t += 64)
        if (t != a) dmin = fmin(dmin, dc[t]);
    for (int o = 32; o > 0; o >>= 1) dmin = fmin(dmin, __shfl_xor(dmin, o));
    const bool fixed = d.cov_type == COV_FIXED;
    // Radii: up to twice the label's own root mean square radius -- where its members are -- when that is known,
    // whatever lies nearer than that (a small component inside a large one has its centre well inside the other's
    // members: the bound then takes it at distance 0, which is right); else half the distance to the nearest centre.
    const double rms = d.pr_rms[a];
    const double step = K > 1 ? (rms > 0.0 ? 2.0 * rms : 0.5 * dmin) / 63.0 : 1.0;
    const double rj = (double)j * step * (1.0 + 1e-9);
    auto ub = [&](int t, double r) {
        const double *__restrict__ g = d.pr_const + (long long)(t >> 4) * 128 + (t & 15);
        double dl = dc[t] * (1.0 - 1e-9) - r;
        dl = dl > 0.0 ? dl : 0.0;
        const double tt = dl * dl * g[32];
        // (log1p_lower of kernels_prune.hip, restated: frexp + chord)
        const double y = 1.0 + tt;
        const double m = __builtin_amdgcn_frexp_mant(y);
        const int e = __builtin_amdgcn_frexp_exp(y);
        const double L = 0.6931471805599453 * ((double)(e - 2) + 2.0 * m);
        return g[0] - g[16] * (fixed ? tt : L);
    };
    double f = -INFINITY;
    for (int t = part; t < K; t += 4)
        if (t != a) f = fmax(f, ub(t, rj));
    red[part][j] = f;
    // the home's neighbours (home_kernel scores them exactly, full covariance): the kHomeNbr labels whose bound at the
    // largest tabulated radius is highest -- wave 0, lane = label mod 64, kHomeNbr rounds of a wave-wide arg max
    if (part == 0) {
        const double r63 = 63.0 * step * (1.0 + 1e-9);
        int picked[kHomeNbr];
        for (int m = 0; m < kHomeNbr; ++m) {
            double best = -INFINITY;
            int bt = -1;
            for (int t = j; t < K; t += 64) {
                if (t == a) continue;
                bool taken = false;
                for (int q = 0; q < m; ++q) taken = taken || picked[q] == t;
                if (taken) continue;
                const double v = ub(t, r63);
                if (v > best || bt < 0) { best = v; bt = t; }
            }
            for (int o = 32; o > 0; o >>= 1) {
                const double ov = __shfl_xor(best, o);
                const int ot = __shfl_xor(bt, o);
                if (ot >= 0 && (bt < 0 || ov > best || (ov == best && ot < bt))) { best = ov; bt = ot; }
            }
            picked[m] = bt;                                   // (-1: fewer than m + 1 other labels)
        }
        // ascending by label (the draw walks its candidates in label order), -1 last
        for (int x = 0; x < kHomeNbr; ++x)
            for (int y = x + 1; y < kHomeNbr; ++y) {
                const int px = picked[x], py = picked[y];
                if (py >= 0 && (px < 0 || py < px)) { picked[x] = py; picked[y] = px; }
            }
        static_assert(kHomeNbr <= 4, "nbr holds four labels per home");
        if (j < 4) { const int v = j < kHomeNbr ? picked[j < kHomeNbr ? j : 0] : -1; nbl[j] = v; d.nbr[(long long)a * 4 + j] = v; }
    }
    __syncthreads();
    // the same bound over every label but the home and its neighbours
    double f2 = -INFINITY;
    for (int t = part; t < K; t += 4)
        if (t != a && t != nbl[0] && t != nbl[1] && t != nbl[2] && t != nbl[3]) f2 = fmax(f2, ub(t, rj));
    f = fmax(fmax(red[0][j], red[1][j]), fmax(red[2][j], red[3][j]));
    __syncthreads();
    red[part][j] = f2;
    __syncthreads();
    if (part != 0) return;
    f2 = fmax(fmax(red[0][j], red[1][j]), fmax(red[2][j], red[3][j]));
    d.ftab[(long long)a * 64 + j] = f;
    d.ftab2[(long long)a * 64 + j] = f2;
    if (j == 0) d.finv[a] = (step > 0.0 && step < INFINITY) ? 1.0 / step : 0.0;
}

// exclusive prefix over the bins of the bucket sort (one block)
__global__ __launch_bounds__(1024) void bucket_prefix_kernel(Dev d) {
    __shared__ int wsum_[16], wlive_[16];
    const Ctrl *c = d.ctrl;
    if (!job_is_pruned(d, c->job.mode, c->job.prune) || c->skip_sort || (d.safe_mode && c->safe_epoch_valid)) return;
    const int nb = d.nslots + 1;
    // home_kernel walks the order in blocks of kHomeBlock rows with the block's home factor in LDS: a block that
    // straddled two homes took a path three times as long, and the kernel lasts as long as its slowest workgroup
    // (measured at C4: 131 us with one component, 155 with 200).  So with the home pass in front every bin's run is
    // padded to a multiple of kHomeBlock (bucket_scatter_kernel fills the pad with dead records); the other consumers
    // of the order (the pruning kernels without a home pass in front) get it compact.
    const bool pad = d.use_home != 0;
    // exclusive prefix over nb <= ~1k bins: every thread owns a contiguous run
    const int per = (nb + 1023) / 1024;
    const int lo = threadIdx.x * per, hi = lo + per < nb ? lo + per : nb;
    auto padded = [&](int v) { return pad ? (v + kHomeBlock - 1) / kHomeBlock * kHomeBlock : v; };
    int local = 0, live = 0;
    for (int b = lo; b < hi; ++b) { const int v = d.bucket_bins[b]; local += padded(v); live += v; }
    int incl = local, incl_live = live;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o), tl = __shfl_up(incl_live, o);
        if (lane >= o) { incl += t; incl_live += tl; }
    }
    if (lane == 63) { wsum_[w] = incl; wlive_[w] = incl_live; }
    __syncthreads();
    int woff = 0;
    for (int k = 0; k < w; ++k) woff += wsum_[k];
    int run = woff + incl - local;
    for (int b = lo; b < hi; ++b) {
        const int v = d.bucket_bins[b];
        d.bucket_bins[b] = run;
        d.bucket_end[b] = run + v;
        run += padded(v);
        d.bucket_end[nb + 1 + b] = run;
    }
    if (threadIdx.x == 1023) {
        int total_live = 0;
        for (int k = 0; k < 16; ++k) total_live += wlive_[k];
        d.ctrl->n_sorted = total_live;                        // rows that take part (all but the certified ones)
        d.ctrl->n_sorted_pad = pad ? run : 0;                 // extent of the padded layout
        d.ctrl->wsort_padded = pad ? 1 : 0;
    }
}

__global__ __launch_bounds__(256) void bucket_scatter_kernel(Dev d) {
    extern __shared__ int lds[];                  // [nb] local counts, then [nb] reserved bases
    const Ctrl *c = d.ctrl;
    if (!job_is_pruned(d, c->job.mode, c->job.prune) || c->skip_sort || (d.safe_mode && c->safe_epoch_valid)) return;
    const long long base = c->job.win_base;
    const int nrows = (int)(c->job.win_hi - base);
    const int r0 = blockIdx.x * BUCKET_ROWS;
    const int nb = d.nslots + 1;
    // the pads of the padded layout (home_kernel's): dead records that carry their bin's home, so that a block reads as
    // one home from its first row to its last  (every block of the grid takes its share, also those beyond the window)
    if (d.use_home)
        for (int b = blockIdx.x; b < nb; b += gridDim.x) {
            const int e0 = d.bucket_end[b], e1 = d.bucket_end[nb + 1 + b];
            for (int k = e0 + threadIdx.x; k < e1; k += 256) {
                WRec rec;
                rec.i = -1;
                rec.home = b - 1;
                rec.home_label = -1;
                rec.mlb0 = 0.0;
                rec.u = 0.5;
                d.wrec[k] = rec;
                d.wperm[k] = 0;
            }
        }
    if (r0 >= nrows) return;
    int *cnt = lds, *res = lds + nb;
    for (int b = threadIdx.x; b < nb; b += 256) cnt[b] = 0;
    __syncthreads();
    // (a thread's four rows side by side: index loads, then label and prior loads, in flight together)
    int myb[BUCKET_ROWS / 256], myk[BUCKET_ROWS / 256];
    long long ii[BUCKET_ROWS / 256];
    double lp[BUCKET_ROWS / 256], uu[BUCKET_ROWS / 256];
#pragma unroll
    for (int t = 0; t < BUCKET_ROWS / 256; ++t) {
        const int r = r0 + threadIdx.x + t * 256;
        const bool ok = r < nrows && !(d.use_certify && d.cert[r < nrows ? r : 0]);
        const long long p = base + (r < nrows ? r : 0);
        ii[t] = ok ? (d.order ? d.order[p] : p) : -1;
        uu[t] = d.u[p];                                // (read here in storage order: a scattered read per visit in home_kernel)
    }
#pragma unroll
    for (int t = 0; t < BUCKET_ROWS / 256; ++t) {
        myb[t] = ii[t] >= 0 ? d.z[ii[t]] + 1 : -1;
        lp[t] = d.log_prior[ii[t] >= 0 ? ii[t] : 0];
    }
#pragma unroll
    for (int t = 0; t < BUCKET_ROWS / 256; ++t) {
        myk[t] = 0;
        if (myb[t] >= 0) myk[t] = atomicAdd(&cnt[myb[t]], 1);          // rank inside (block, bucket)
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
            d.wperm[k] = r;
            WRec rec;
            rec.i = ii[t];
            rec.home = myb[t] - 1;
            rec.home_label = rec.home >= 0 ? d.label_of_slot[rec.home] : -1;
            rec.mlb0 = d.log_alpha + lp[t];
            rec.u = uu[t];
            d.wrec[k] = rec;
        }
}

void launch_prune_tables(const Dev &d, hipStream_t st) {
    const unsigned ngrp = (unsigned)((d.nslots + 15) / 16);
    hipLaunchKernelGGL(prune_tables_kernel, dim3((ngrp + 15) / 16 + d.nslots), dim3(1024), 0, st, d);
    // (always together with the tables, also in sweeps that do not certify: one validity flag)
    hipLaunchKernelGGL(prune_ftable_kernel, dim3(d.nslots), dim3(256), 0, st, d);
}

void launch_bucket_rows(const Dev &d, long long max_rows, hipStream_t st) {
    // (d.bucket_bins is zero on entry: cleared at create and by certify_kernel / choice_sparse_kernel)
    const int nb = d.nslots + 1;
    const unsigned g = (unsigned)((max_rows + BUCKET_ROWS - 1) / BUCKET_ROWS);
    hipLaunchKernelGGL(bucket_count_kernel, dim3(g), dim3(256), nb * (int)sizeof(int), st, d);
    hipLaunchKernelGGL(bucket_prefix_kernel, dim3(1), dim3(1024), 0, st, d);
    hipLaunchKernelGGL(bucket_scatter_kernel, dim3(g), dim3(256), 2 * nb * (int)sizeof(int), st, d);
}

void launch_sweep_begin(const Dev &d, hipStream_t st, const Dev *group, int n_group) {
    if (group) hipLaunchKernelGGL(sweep_begin_group_kernel, dim3((unsigned)n_group), dim3(TPB), 0, st, group);
    else hipLaunchKernelGGL(sweep_begin_kernel, dim3(1), dim3(TPB), 0, st, d);
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
    if (d.short_step == 2) {
        // (the bucket sort of this step has left its offsets in the bins, and the draw kernel that clears them for the next
        // sort was not queued)
        for (int b = threadIdx.x; b < d.nslots + 2; b += TPB) d.bucket_bins[b] = 0;
    }
    if (d.short_step && c->job.mode != MODE_DONE) {
        // only home_kernel (and maybe the bucket sort) ran: the step stands iff that was all the window needed
        __shared__ int refuse;
        if (threadIdx.x == 0) {
            const bool ok = job_is_pruned(d, c->job.mode, c->job.prune) && c->tables_valid &&
                            (d.short_step == 2 || c->skip_sort) && resid_left(d, c) == 0 && c->first_mover == kNoMover;
            refuse = ok ? 0 : 1;
            if (refuse) { c->retry_full = 1; c->n_refresh = 0; c->n_resid = 0; c->first_mover = kNoMover; __threadfence(); }
        }
        __syncthreads();
        if (refuse) {
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
            c->n_pairs_exact += 16ull * cnt_red[0];         // (16-visit blocks scored in full)
            c->n_bound_blocks += cnt_red[TPB];
            c->n_prune_mfma += cnt_red[2 * TPB];
            c->n_certified += cnt_red[3 * TPB];
        }
    }
    if (threadIdx.x == 0) {
        do_move = 0;
        Job &j = c->job;
        if (d.use_home && !d.lean_step && job_is_pruned(d, j.mode, j.prune) && !c->skip_apply) { c->home_in += c->n_sorted; c->home_out += resid_left(d, c); }
        c->n_resid = 0;                 // (home_kernel's list of this step has been worked through)
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
            if (!was_pruned && !d.lean_step)
                c->n_pairs_exact += (unsigned long long)((j.win_hi - j.pos) * (long long)(j.mode == MODE_FRESH ? j.K : j.n_dirty));
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
                    // (K_max reached: the reference raises from add_item with the point already taken out by
                    //  del_item -- counts, statistics and derived state are left exactly like that: the removal
                    //  is applied, the point stays unassigned, the sweep ends with the sticky error)
                    atomicCAS(&c->error, 0, rc);
                    j.mode = MODE_DONE;
                    mp.add_slot = -1; mp.add_init = 0;
                    do_move = mp.sub_slot >= 0 ? 1 : 0;
                    c->tables_valid = 0;
                    c->wsort_valid = 0;
                    c->state_epoch += 1;
                    set_refresh(d, c, mp, false);
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

// restore_component_from_stats: statistics of label `label` overwritten from device buffers; the slot goes
// on the refresh list (rebuilt from scratch by the launch behind this one)
__global__ __launch_bounds__(TPB) void set_stats_kernel(Dev d, int label, const double *__restrict__ m_in,
                                                        const double *__restrict__ S_in, int count) {
    Ctrl *c = d.ctrl;
    const int s = d.perm[label], D = d.D;
    const int DD = d.cov_type == COV_FULL ? D * D : (d.cov_type == COV_FIXED ? 2 * D : D);
    for (int a = threadIdx.x; a < D; a += TPB) d.m[(long long)s * D + a] = m_in[a];
    for (int e = threadIdx.x; e < DD; e += TPB) d.S[(long long)s * DD + e] = S_in[e];
    if (threadIdx.x == 0) {
        d.n[s] = count;
        c->tables_valid = 0; c->wsort_valid = 0; c->state_epoch += 1;
        c->n_refresh = 1; c->refresh[0] = s; c->refresh_kind[0] = REFRESH_SCRATCH;
    }
}

// The raw statistics blocks of one label, as they sit in HBM (m[D]; S: D x D, D, or -- fixed variance -- [precision_N, sum x^2])
__global__ __launch_bounds__(TPB) void raw_stats_kernel(Dev d, int label, double *__restrict__ m_out, double *__restrict__ S_out) {
    const int s = d.perm[label], D = d.D;
    const int DD = d.cov_type == COV_FULL ? D * D : (d.cov_type == COV_FIXED ? 2 * D : D);
    for (int a = threadIdx.x; a < D; a += TPB) m_out[a] = d.m[(long long)s * D + a];
    for (int e = threadIdx.x; e < DD; e += TPB) S_out[e] = d.S[(long long)s * DD + e];
}
void launch_raw_stats(const Dev &d, int label, double *m_out, double *S_out, hipStream_t st) {
    hipLaunchKernelGGL(raw_stats_kernel, dim3(1), dim3(TPB), 0, st, d, label, m_out, S_out);
}

// del_component (gaussian_components.py:188-205) as a call of its own: whoever still sits in the component becomes
// unassigned, then the swap-with-last delete of its label
__global__ __launch_bounds__(256) void del_component_members_kernel(Dev d, int label) {
    const int s = d.perm[label];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < d.N; i += (long long)gridDim.x * 256)
        if (d.z[i] == s) d.z[i] = -1;
}
__global__ void del_component_label_kernel(Dev d, int lab) {
    Ctrl *c = d.ctrl;
    const int h = d.perm[lab], last = c->job.K - 1, s_last = d.perm[last];
    d.n[h] = 0;
    d.perm[lab] = s_last; d.label_of_slot[s_last] = lab;
    d.perm[last] = h; d.label_of_slot[h] = last;
    c->job.K = last;
    c->tables_valid = 0; c->wsort_valid = 0; c->state_epoch += 1; c->n_refresh = 0;
}
void launch_del_component(const Dev &d, int label, hipStream_t st) {
    hipLaunchKernelGGL(del_component_members_kernel, dim3(1024), dim3(256), 0, st, d, label);
    hipLaunchKernelGGL(del_component_label_kernel, dim3(1), dim3(1), 0, st, d, label);
}

__global__ void set_label_kernel(Dev d, long long i, int label) {
    Ctrl *c = d.ctrl;
    d.z[i] = label < 0 ? -1 : d.perm[label];
    c->tables_valid = 0; c->wsort_valid = 0; c->state_epoch += 1; c->n_refresh = 0;
}
void launch_set_label(const Dev &d, long long i, int label, hipStream_t st) {
    hipLaunchKernelGGL(set_label_kernel, dim3(1), dim3(1), 0, st, d, i, label);
}

void launch_set_stats(const Dev &d, int label, const double *m_in, const double *S_in, int count, hipStream_t st) {
    hipLaunchKernelGGL(set_stats_kernel, dim3(1), dim3(TPB), 0, st, d, label, m_in, S_in, count);
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

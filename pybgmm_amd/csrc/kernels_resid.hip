// What home_kernel could not decide, when that is LITTLE: every (visit, component) pair exactly, in one launch.
//
// A chain at rest whose clusters sit a dozen sigma apart (C3: D = 16, K = 100) leaves home_kernel (kernels_home.hip) a
// fraction of a per cent of its visits -- those for which neither the per-home table nor the home's four neighbours
// settle the draw.  The general tools for a residual list, score_mfma_prune_kernel + choice_sparse_kernel, are built for
// long lists: two launches whose wavefronts walk chains of dependent round trips (~100 us each at C3 for 5 000 visits).
// For a short list the plain thing is cheaper by an order of magnitude: sixteen listed visits per workgroup, every label's
// exact quadratic form with v_mfma_f64_16x16x4_f64 (the labels dealt to the eight wavefronts, the factor fragments straight
// from L2), the log scores in LDS, then the reference's draw over ALL of them (igmm/crpmm.py:68-78, utils/utils.py:7-20;
// the arithmetic of choice_kernel, kernels_choice.hip: nothing is pruned, so nothing has to be argued).
// Runs iff resid_dense_takes() (bgmm_device.h): the same predicate makes the two general kernels see an empty list.
#include "score_common.h"
#include "wave_ops.h"

#define LDS_AS __attribute__((address_space(3)))

template <int NJ>
__global__ __launch_bounds__(512) void resid_dense_kernel(Dev d) {
    extern __shared__ __attribute__((aligned(16))) double tile_raw[];
    Ctrl *c = d.ctrl;
    if (!job_is_pruned(d, c->job.mode, c->job.prune) || !resid_dense_takes(d, c)) return;
    if (d.short_step && (!c->tables_valid || (d.short_step == 1 && !c->skip_sort))) return;     // (home_kernel stood aside: so does this)
    const int nres = c->n_resid;
    const int k0 = (int)blockIdx.x * 16;
    if (k0 >= nres) return;
    constexpr int NF = 2 * NJ * (NJ + 1), PF = pick_pf(NF), R = 16;
    LDS_AS double *const tile = (LDS_AS double *)tile_raw;        // [K_max + 2][16]: q, then log scores, then weights
    const int D = d.D, K = c->job.K;
    const int tid = threadIdx.x, lane = tid & 63, lr = lane & 15, lk = lane >> 4;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long win_base = c->job.win_base;
    // ---- every label's quadratic form of the sixteen rows (A fragments per wavefront, labels w, w + 8, ...)
    {
        const int k = k0 + lr < nres ? k0 + lr : nres - 1;
        const long long i = d.wrecR[k].i;
        const double *__restrict__ xrow = d.X + (i >= 0 ? i : 0) * D;
        double xf[NJ * 4];
#pragma unroll
        for (int kk = 0; kk < NJ * 4; ++kk) {
            const int l = 4 * kk + lk;
            xf[kk] = l < D ? xrow[l] : 0.0;
        }
        for (int j = w; j < K; j += 8) {
            const int s = d.perm[j];
            const double *__restrict__ wf = d.Wfrag + (long long)s * NF * 64 + lane;
            double ring[PF];
#pragma unroll
            for (int q = 0; q < PF; ++q) ring[q] = wf[q * 64];
            double qp[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int J = 0; J < NJ; ++J) {
                const double cj = d.cvec[(long long)s * d.Dp + 16 * J + lr];
                v4d acc = (v4d){cj, cj, cj, cj};
#pragma unroll
                for (int kk = 0; kk < 4 * (J + 1); ++kk) {
                    const int f = 2 * J * (J + 1) + kk;
                    const double b = ring[f % PF];
                    if (f + PF < NF) ring[f % PF] = wf[(f + PF) * 64];
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(xf[kk], b, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) qp[r] = fma(acc[r], acc[r], qp[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = row16_sum(qp[r]);
                if (lr == r) tile[j * R + lk + 4 * r] = v;
            }
        }
    }
    __syncthreads();
    // ---- the draws: wavefront w takes rows w, w + 8 (choice_kernel's three passes)
    const int NEWIDX = d.K_max + 1;
    for (int row = w; row < R; row += 8) {
        const int k = k0 + row;
        if (k >= nres) break;
        const WRec rec = d.wrecR[k];
        const int wrow = d.wpermR[k];
        const long long p = win_base + wrow;
        const long long i = rec.i;
        const int h = rec.home;
        const int nh = h >= 0 ? d.n[h] : 0;
        const bool home_live = h >= 0 && nh >= 2;
        const bool singleton = h >= 0 && nh == 1;
        const int lab_h = singleton ? d.label_of_slot[h] : -1;
        const int L = singleton ? K - 1 : K;
        double mx = -INFINITY;
        for (int j = lane; j <= L; j += 64) {
            double v;
            if (j == L) {
                v = rec.mlb0;                                   // log(alpha) + log_prior[i] (bucket_scatter_kernel)
                tile[NEWIDX * R + row] = v;
            } else {
                const int jj = (singleton && j == lab_h) ? K - 1 : j;
                const int s = d.perm[jj];
                v = slot_score_exact(d.sc[s], tile[jj * R + row], home_live && s == h);
            }
            mx = fmax(mx, v);
            if (j < L) tile[j * R + row] = v;
        }
        mx = wv_max(mx);
        // (a deleted singleton's stand-in, entry K - 1, was read above and is written below only as label lab_h's score:
        //  the passes below index by post-removal label j, the stand-in's own entry K - 1 is not among them)
        double tot = 0.0;
        for (int j = lane; j <= L; j += 64) {
            const int idx = (j == L ? NEWIDX : j) * R + row;
            const double e = exp(tile[idx] - mx);
            tile[idx] = e;
            tot += e;
        }
        tot = wv_sum(tot);
        const double u = d.u[p];
        double carry = 0.0;
        int pick = L;
        for (int j0 = 0; j0 <= L; j0 += 64) {
            const int j = j0 + lane;
            const double pj = j <= L ? tile[(j == L ? NEWIDX : j) * R + row] / tot : 0.0;
            const double cum = carry + wv_scan(pj, lane);
            const bool hit = j <= L && (u - cum) < 0.0;
            const unsigned long long m = __ballot(hit);
            if (m) { pick = j0 + __ffsll((long long)m) - 1; break; }
            carry = wv_readlane(cum, 63);
        }
        if (lane == 0) {
            d.choice[wrow] = pick;
            const bool stay = home_live && pick < L && d.perm[pick] == h;
            if (!stay) atomicMin(&c->first_mover, (unsigned long long)p);
        }
        (void)i;
    }
    if (tid == 0) atomicAdd(&c->n_pairs_exact, (unsigned long long)((nres - k0 < 16 ? nres - k0 : 16) * (long long)K));
}

int resid_dense_lds_bytes(const Dev &d) { return (d.K_max + 2) * 16 * (int)sizeof(double); }

template <int NJ>
static void launch_resid_t(const Dev &d, hipStream_t st) {
    const int lds = resid_dense_lds_bytes(d);
    static PerDeviceLds attr;
    attr.ensure((const void *)resid_dense_kernel<NJ>, lds);
    hipLaunchKernelGGL((resid_dense_kernel<NJ>), dim3(kResidDenseMax / 16), dim3(512), lds, st, d);
}

// Queued behind home_kernel when Dev::resid_dense is set (api_sweep.hip: full covariance, D <= 32, certified stays off,
// the log scores of all labels fit LDS).
void launch_resid_dense(const Dev &d, hipStream_t st) {
    if (!d.resid_dense) return;
    if (d.Dp == 16) launch_resid_t<1>(d, st);
    else if (d.Dp == 32) launch_resid_t<2>(d, st);
}

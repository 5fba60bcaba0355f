// Likelihood kernels: the batched quadratic form
//      q[r][s] = (mu_s - x_r)^T C_s^{-1} (mu_s - x_r) = || cvec_s - Winv_s x_r ||^2
// of every visit r of the current speculative window against every listed component slot s.
// This is the K*D^2 contraction of GaussianComponents.log_post_pred
// (reference gaussian_components.py:240-244, the two einsums) evaluated for a whole tile of
// visits at once against frozen component state.
//
//   score_valu_kernel : any D.  One lane per visit, x tile transposed in LDS, Winv rows
//                       fetched through the scalar cache (wave-uniform), FP64 VALU FMAs.
//   score_mfma_kernel : D padded to a multiple of 16.  v_mfma_f64_16x16x4_f64 with the x
//                       tile resident in registers as A fragments for the whole slot loop,
//                       the negated block-lower-triangular Winv streamed as pre-swizzled
//                       B fragments (one coalesced 512-B load per MFMA), the accumulator
//                       initialised with cvec so the MFMA chain directly yields
//                       y = cvec - Winv x; q = sum y^2 by a 16-lane DPP/shuffle reduction.
#include "score_common.h"

// ------------------------------------------------------------------------------------------
// VALU kernel.  q is stored slot-major: q[col * qstride + row].
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void score_valu_kernel(Dev d, const Job *__restrict__ jobp,
                                                         double *__restrict__ q, long long qstride,
                                                         int col_override, int skip_pruned_jobs) {
    extern __shared__ __attribute__((aligned(16))) double xs[];   // [D][64]
    const JobView job = load_job(jobp);
    if (job.mode == MODE_DONE || (skip_pruned_jobs && job_is_pruned(d, job.mode, job.prune))) return;
    const int chunk = blockIdx.y;
    if (chunk >= job.chunks) return;
    const long long p0 = job.pos + (long long)blockIdx.x * kValuRows;
    if (p0 >= job.win_hi) return;
    const int D = d.D;

    // stage the x tile transposed: xs[l][r]
    for (int e = threadIdx.x; e < kValuRows * D; e += 256) {
        const int r = e / D, l = e % D;
        const long long p = p0 + r;
        double v = 0.0;
        if (p < job.win_hi) {
            const long long i = d.order ? d.order[p] : p;
            v = d.X[i * D + l];
        }
        xs[l * kValuRows + r] = v;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long p = p0 + lane;
    // entries of this chunk: chunk, chunk + chunks, ...; the 4 waves take them round robin
    for (int t = chunk + job.chunks * w; t < job.nlist; t += job.chunks * 4) {
        const int s = job_slot(d, job, t);
        const double *__restrict__ W = d.Wrm + (long long)s * D * D;
        const double *__restrict__ cv = d.cvec + (long long)s * d.Dp;
        double qv = 0.0;
        for (int j = 0; j < D; ++j) {
            double acc = cv[j];
            const double *__restrict__ Wj = W + j * D;
            for (int l = 0; l <= j; ++l) acc = fma(-Wj[l], xs[l * kValuRows + lane], acc);
            qv = fma(acc, acc, qv);
        }
        if (p < job.win_hi)
            q[(long long)(col_override >= 0 ? col_override : s) * qstride + (p - job.win_base)] = qv;
    }
}

// ------------------------------------------------------------------------------------------
// MFMA kernel.  NJ = Dp/16 column blocks; each wave owns RB x 16 rows.
// Fragment conventions (v_mfma_f64_16x16x4_f64):
//   A: lane holds A[i = lane&15][k = lane>>4]   -> x[row lane&15][4*kk + (lane>>4)]
//   B: lane holds B[k = lane>>4][j = lane&15]   -> -Winv[16J + (lane&15)][4*kk + (lane>>4)]
//   C/D: lane holds rows (lane>>4) + 4*r, r = 0..3, column lane&15
// The B fragments of a slot are a flat list of NF = 2 NJ (NJ+1) 512-byte pieces.  They are
// streamed through a register ring of PF pieces: the piece consumed by MFMA number f is
// replaced at once by the load of piece f + PF (wrapping into the NEXT slot's list), so PF
// loads are always in flight ahead of the matrix pipe.
// ------------------------------------------------------------------------------------------
// Sum over the 16 lanes of a DPP row without touching the LDS crossbar: quad_perm [1,0,3,2],
// quad_perm [2,3,0,1], row_half_mirror, row_mirror -- after the four steps every lane of the
// row holds the row total.
// PROOF: the dense proof pass of a safe-stay stretch (skip_pruned_jobs 2) is an instantiation of its own -- with its label split
// as a run-time variable in the one kernel, the windows' instantiation spilled four A fragments and reloaded them in every
// slot iteration (mode `full` 52.6 -> 43.8 sweeps/s between rounds 3 and 4).
// (bx, by: the workgroup's row block and label chunk, gdy: the chunks the launch provides -- the launch grid's x / y index and y extent)
template <int NJ, int RB, bool PROOF>
__device__ __forceinline__ void score_mfma_body(const Dev &d, const Job *__restrict__ jobp, double *__restrict__ q, long long qstride,
                                                int col_override, int skip_pruned_jobs, int bx, int by, int gdy, int wg_target = 1400) {
    const JobView job = load_job(jobp);
    // (PROOF: runs whatever the window's kind, but only in front of a stretch whose proofs are to be made, kernels_safe.hip)
    if (job.mode == MODE_DONE || (!PROOF && skip_pruned_jobs == 1 && job_is_pruned(d, job.mode, job.prune)) ||
        (PROOF && skip_pruned_jobs == 2 && d.ctrl->safe_epoch_valid)) return;      // (3: a job of the look-ahead's -- DONE when idle)
    constexpr int ROWS_W_ = 16 * RB;
    const int chunk = by;
    // (the dense proof pass covers a few thousand rows: its launch brings its own, finer split of the labels -- grid.y --
    // so that every compute unit holds two or three workgroups and a wavefront's factor loads hide behind its neighbours')
    int nchunks = job.chunks;
    if (PROOF) {
        // ~1 400 workgroups whatever the stretch's length (measured: 197 ms per sweep at 0.5 % movers against 207 with 700, 240
        // with 400), at most one label chunk per grid row
        const long long rb = (job.win_hi - job.pos + 4 * ROWS_W_ - 1) / (4 * ROWS_W_);
        long long ch = rb > 0 ? (wg_target + rb - 1) / rb : 1;
        nchunks = (int)(ch < 2 ? 2 : (ch > (long long)gdy ? (long long)gdy : ch));
    }
    if (chunk >= nchunks || chunk >= job.nlist) return;
    constexpr int ROWS_W = ROWS_W_;              // rows per wave
    constexpr int NF = 2 * NJ * (NJ + 1);
    constexpr int PF = pick_pf(NF);
    const long long pb = job.pos + (long long)bx * (4 * ROWS_W);
    if (pb >= job.win_hi) return;
    const int D = d.D;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long pw = pb + w * ROWS_W;
    if (pw >= job.win_hi) return;
    const int lr = lane & 15, lk = lane >> 4;

    // A fragments of this wave's rows, resident for the whole slot loop
    double xf[RB][NJ * 4];
#pragma unroll
    for (int R = 0; R < RB; ++R) {
        const long long p = pw + R * 16 + lr;
        const bool live = p < job.win_hi;
        const long long i = live ? (d.order ? d.order[p] : p) : 0;
        const double *__restrict__ xrow = d.X + i * D;
#pragma unroll
        for (int kk = 0; kk < NJ * 4; ++kk) {
            const int l = 4 * kk + lk;
            xf[R][kk] = (live && l < D) ? xrow[l] : 0.0;
        }
    }

    const long long nfrag64 = (long long)NF * 64;
    int t = chunk;
    int s = job_slot(d, job, t);
    const double *__restrict__ wf = d.Wfrag + (long long)s * nfrag64 + lane;
    double ring[PF];
#pragma unroll
    for (int i = 0; i < PF; ++i) ring[i] = wf[i * 64];
    double cj[NJ];
#pragma unroll
    for (int J = 0; J < NJ; ++J) cj[J] = d.cvec[(long long)s * d.Dp + 16 * J + lr];

    for (;;) {
        const int tn = t + nchunks;
        const bool has_next = tn < job.nlist;
        const int sn = has_next ? job_slot(d, job, tn) : s;
        const double *__restrict__ wfn = d.Wfrag + (long long)sn * nfrag64 + lane;
        double cjn[NJ];
#pragma unroll
        for (int J = 0; J < NJ; ++J) cjn[J] = d.cvec[(long long)sn * d.Dp + 16 * J + lr];

        double qp[RB][4];
#pragma unroll
        for (int R = 0; R < RB; ++R)
#pragma unroll
            for (int r = 0; r < 4; ++r) qp[R][r] = 0.0;
#pragma unroll
        for (int J = 0; J < NJ; ++J) {
            v4d acc[RB];
#pragma unroll
            for (int R = 0; R < RB; ++R) acc[R] = (v4d){cj[J], cj[J], cj[J], cj[J]};
#pragma unroll
            for (int kk = 0; kk < 4 * (J + 1); ++kk) {
                const int f = 2 * J * (J + 1) + kk;          // folds to a constant when unrolled
                const double b = ring[f % PF];
                ring[f % PF] = (f + PF < NF) ? wf[(f + PF) * 64] : wfn[(f + PF - NF) * 64];
#pragma unroll
                for (int R = 0; R < RB; ++R)
                    acc[R] = __builtin_amdgcn_mfma_f64_16x16x4f64(xf[R][kk], b, acc[R], 0, 0, 0);
            }
#pragma unroll
            for (int R = 0; R < RB; ++R)
#pragma unroll
                for (int r = 0; r < 4; ++r) qp[R][r] = fma(acc[R][r], acc[R][r], qp[R][r]);
        }
        // sum over the 16 columns held by the 16 lanes of each (lane>>4) group, then lane
        // (lk, lr = r) stores row lk + 4r: 16 consecutive rows = one 128-byte segment
        double *__restrict__ qcol = q + (long long)(col_override >= 0 ? col_override : s) * qstride;
#pragma unroll
        for (int R = 0; R < RB; ++R) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double v = row16_sum(qp[R][r]);
                const long long p = pw + R * 16 + lk + 4 * r;
                if (lr == r && p < job.win_hi) qcol[p - job.win_base] = v;
            }
        }
        if (!has_next) break;
        t = tn;
        s = sn;
        wf = wfn;
#pragma unroll
        for (int J = 0; J < NJ; ++J) cj[J] = cjn[J];
    }
}

template <int NJ, int RB, int MINW, bool PROOF>
__global__ __launch_bounds__(256, MINW) void score_mfma_kernel(Dev d, const Job *__restrict__ jobp,
                                                            double *__restrict__ q, long long qstride,
                                                            int col_override, int skip_pruned_jobs) {
    score_mfma_body<NJ, RB, PROOF>(d, jobp, q, qstride, col_override, skip_pruned_jobs, (int)blockIdx.x, (int)blockIdx.y, (int)gridDim.y);
}
// (the dense proof pass of several chains' stretches in one launch: workgroup (x, y, c) works for chain group[c] on its own
//  open window, into its own q -- api_group.hip, the shared safe-stay steps)
// which: -1 the chain's open window, every label (skip mode 2: launch_score(.., 2, ..)); 0 .. 2 the look-ahead's job
// Dev::ah_job[which], 3 Dev::resc_job (skip mode 3; the list jobs index Dev::resc_list), 4 all three of ah_job (grid.z)
// The grid is (kProofGroupBlocks, chains): a workgroup takes the (row block, label chunk) pairs of ITS chain's job blockIdx.x,
// + gridDim.x, ... -- the job's real extent is on the device only, and a grid cut for the largest stretch a batch may reach
// ((64 row blocks, 64 chunks) per chain, 4 096 workgroups of which a re-scoring of eight touched labels uses 80) made eight
// chains' launch 32 768 workgroups: 282 us of dispatching against 27 for one chain.
static constexpr int kProofGroupBlocks = 704, kProofChunks = 64;
template <int NJ, int RB, int MINW>
__global__ __launch_bounds__(256, MINW) void score_mfma_proof_group_kernel(const Dev *__restrict__ group, int which) {
    Dev d = group[blockIdx.y];                     // (a private copy: nothing the body writes can alias it)
    if (which == 4) which = (int)blockIdx.z;       // (the look-ahead's three jobs in ONE launch: grid.z = 3)
    const Job *__restrict__ jobp = which < 0 ? &d.ctrl->job : (which == 3 ? d.resc_job : d.ah_job + which);
    if (which > 0) d.slot_list = d.resc_list;
    const int skip = which < 0 ? 2 : 3;
    // (the ~1 400 workgroups a proof launch is cut into are for ONE chain's job to fill the chip: with gridDim.y chains -- and
    //  gridDim.z jobs -- in the launch each job is cut into its share, and a workgroup keeps its rows for several labels)
    const int wg_target = max(1400 / (int)(gridDim.y * gridDim.z), 64);
    const JobView job = load_job(jobp);
    if (job.mode == MODE_DONE) return;
    const long long rows = job.win_hi - job.pos;
    if (rows <= 0) return;
    const int rb_count = (int)((rows + 4 * 16 * RB - 1) / (4 * 16 * RB));
    const long long items = (long long)rb_count * kProofChunks;       // (chunks beyond the job's own split return at once)
    for (long long it = blockIdx.x; it < items; it += gridDim.x) {
        // (chunk-major: the first gridDim.x items are the row blocks of the first chunks -- those every job has)
        score_mfma_body<NJ, RB, true>(d, jobp, d.q, d.qstride, -1, skip, (int)(it % rb_count), (int)(it / rb_count), kProofChunks, wg_target);
    }
}

// ------------------------------------------------------------------------------------------
template <int NJ>
static void launch_mfma(const Dev &d, const Job *job, double *q, long long qstride, int col_override,
                        long long max_rows, int skip_pruned_jobs, hipStream_t st) {
    const unsigned gx = (unsigned)((max_rows + kMfmaRows - 1) / kMfmaRows);
    // (skip_pruned_jobs 2 = the dense proof pass of a safe-stay stretch, a few thousand rows: up to 64 label chunks)
    const unsigned gy = skip_pruned_jobs >= 2 ? 64 : kMaxChunks;      // (the kernel picks its split from the stretch's real length)
    constexpr int W = NJ <= 4 ? 3 : (NJ <= 5 ? 2 : 1);
    // (the proof pass is ~1 400 workgroups: two per SIMD is all it fills, and at three its label split costs D = 64 32 spills)
    constexpr int WP = NJ == 4 ? 2 : W;
    if (skip_pruned_jobs >= 2)
        hipLaunchKernelGGL((score_mfma_kernel<NJ, 2, WP, true>), dim3(gx, gy), dim3(256), 0, st, d, job, q, qstride, col_override,
                           skip_pruned_jobs);
    else
        hipLaunchKernelGGL((score_mfma_kernel<NJ, 2, W, false>), dim3(gx, gy), dim3(256), 0, st, d, job, q, qstride, col_override,
                           skip_pruned_jobs);
}

void launch_score_diag(const Dev &d, const Job *job, double *q, long long qstride, int col_override,
                       long long max_rows, int skip_pruned_jobs, hipStream_t st);

template <int NJ>
static void launch_mfma_proof_group(const Dev *group, int G, long long max_rows, int which, hipStream_t st) {
    const unsigned gx = (unsigned)((max_rows + kMfmaRows - 1) / kMfmaRows);
    constexpr int W = NJ <= 4 ? 3 : (NJ <= 5 ? 2 : 1);
    constexpr int WP = NJ == 4 ? 2 : W;
    (void)gx;
    hipLaunchKernelGGL((score_mfma_proof_group_kernel<NJ, 2, WP>), dim3(kProofGroupBlocks, (unsigned)G, which == 4 ? 3 : 1), dim3(256), 0, st,
                       group, which);
}
// the exact forms of each of G chains (full covariance, D <= 128) in one launch: every pair of its open stretch (which = -1:
// launch_score(.., 2, ..) for all of them), or a job of its look-ahead (0 .. 2: Dev::ah_job, 3: Dev::resc_job)
bool launch_score_proof_group(const Dev &lead, const Dev *group, int G, long long max_rows, int which, hipStream_t st) {
    if (max_rows <= 0 || lead.cov_type != COV_FULL) return false;
    switch (lead.Dp / 16) {
        case 1: launch_mfma_proof_group<1>(group, G, max_rows, which, st); return true;
        case 2: launch_mfma_proof_group<2>(group, G, max_rows, which, st); return true;
        case 3: launch_mfma_proof_group<3>(group, G, max_rows, which, st); return true;
        case 4: launch_mfma_proof_group<4>(group, G, max_rows, which, st); return true;
        case 5: launch_mfma_proof_group<5>(group, G, max_rows, which, st); return true;
        case 6: launch_mfma_proof_group<6>(group, G, max_rows, which, st); return true;
        case 7: launch_mfma_proof_group<7>(group, G, max_rows, which, st); return true;
        case 8: launch_mfma_proof_group<8>(group, G, max_rows, which, st); return true;
        default: return false;
    }
}

void launch_score(const Dev &d, int kind, const Job *job, double *q, long long qstride, int col_override,
                  long long max_rows, int skip_pruned_jobs, hipStream_t st) {
    if (max_rows <= 0) return;
    if (d.cov_type != COV_FULL) { launch_score_diag(d, job, q, qstride, col_override, max_rows, skip_pruned_jobs, st); return; }
    if (kind == KERNEL_MFMA) {
        switch (d.Dp / 16) {
            case 1: launch_mfma<1>(d, job, q, qstride, col_override, max_rows, skip_pruned_jobs, st); return;
            case 2: launch_mfma<2>(d, job, q, qstride, col_override, max_rows, skip_pruned_jobs, st); return;
            case 3: launch_mfma<3>(d, job, q, qstride, col_override, max_rows, skip_pruned_jobs, st); return;
            case 4: launch_mfma<4>(d, job, q, qstride, col_override, max_rows, skip_pruned_jobs, st); return;
            case 5: launch_mfma<5>(d, job, q, qstride, col_override, max_rows, skip_pruned_jobs, st); return;
            case 6: launch_mfma<6>(d, job, q, qstride, col_override, max_rows, skip_pruned_jobs, st); return;
            case 7: launch_mfma<7>(d, job, q, qstride, col_override, max_rows, skip_pruned_jobs, st); return;
            case 8: launch_mfma<8>(d, job, q, qstride, col_override, max_rows, skip_pruned_jobs, st); return;
            default: break;
        }
    }
    const unsigned gx = (unsigned)((max_rows + kValuRows - 1) / kValuRows);
    const int lds = d.D * kValuRows * (int)sizeof(double);
    static PerDeviceLds attr;                             // (D > 128: the transposed tile is more than the 64 KB a launch gets unasked)
    attr.ensure((const void *)score_valu_kernel, lds);
    hipLaunchKernelGGL(score_valu_kernel, dim3(gx, kMaxChunks), dim3(256), lds, st, d, job, q, qstride,
                       col_override, skip_pruned_jobs);
}

// ------------------------------------------------------------------------------------------
// Diagonal covariance (reference gaussian_components_diag.py:231-259):
// the predictive is a product of univariate Student-t densities,
//     lp = D (lgamma((v+1)/2) - lgamma(v/2) - log(v)/2 - log(pi)/2) - log(prod var)/2
//          - (v+1)/2 sum_d log(1 + (x_d - mu_d)^2 / (v var_d)),          v = v_N,
// i.e. D logarithms per (visit, component): an FP64 VALU kernel.  Lane = visit, x tile
// transposed in LDS, means / weights of the slot wave-uniform.  It stores the log density
// itself (not a quadratic form); for the visit's own component it stores the
// one-point-removed form, rebuilt from (n-1, m-x, S-x^2) exactly as del_item does
// (gaussian_components_diag.py:178-193).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void score_diag_kernel(Dev d, const Job *__restrict__ jobp,
                                                         double *__restrict__ q, long long qstride,
                                                         int col_override, int home_correction,
                                                         int skip_pruned_jobs) {
    extern __shared__ __attribute__((aligned(16))) double xs[];   // [D][64]
    const JobView job = load_job(jobp);
    if (job.mode == MODE_DONE || (skip_pruned_jobs && job_is_pruned(d, job.mode, job.prune))) return;
    const int chunk = blockIdx.y;
    if (chunk >= job.chunks) return;
    const long long p0 = job.pos + (long long)blockIdx.x * kValuRows;
    if (p0 >= job.win_hi) return;
    const int D = d.D;
    const bool xlds = D <= kDiagLdsMaxD;          // (beyond: no tile -- every lane reads its row through the cache, XAT below)
    // (8 row-contiguous loads in flight per thread, then the transposing LDS writes)
    for (int e0 = threadIdx.x; xlds && e0 < kValuRows * D; e0 += 256 * 8) {
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = e0 + 256 * j;
            const int r = e / D, l = e % D;
            const long long p = p0 + r;
            const bool ok = e < kValuRows * D && p < job.win_hi;
            const long long i = ok ? (d.order ? d.order[p] : p) : 0;
            v[j] = d.X[i * D + (ok ? l : 0)];
            if (!ok) v[j] = 0.0;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int e = e0 + 256 * j;
            if (e < kValuRows * D) xs[(e % D) * kDiagLd + e / D] = v[j];
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long p = p0 + lane;
    const bool live = p < job.win_hi;
    const double *__restrict__ xg = d.X + (live ? (d.order ? d.order[p] : p) : 0) * D;
#define XAT(l) (xlds ? xs[(l) * kDiagLd + lane] : xg[(l)])
    // (utility jobs -- log_prior, log_post_pred -- score every slot as it is)
    int home = -1;
    if (live && home_correction) {
        const long long i = d.order ? d.order[p] : p;
        home = d.z[i];
    }
    const double hlp = 0.5 * BGMM_LOG_PI;
    for (int t = chunk + job.chunks * w; t < job.nlist; t += job.chunks * 4) {
        const int s = job_slot(d, job, t);
        const double *__restrict__ mu = d.mu + (long long)s * D;
        const double *__restrict__ dw = d.dw + (long long)s * D;
        const SlotConst *__restrict__ scp = d.sc + s;
        const bool fixed = d.cov_type == COV_FIXED;
        double acc = 0.0;
        if (fixed) {                // product of normals: sum (x - mu)^2 * predictive precision
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
        double lp = scp->A - scp->half_vd * acc;
        const int ns = d.n[s];
        if (fixed && home == s && ns >= 2) {
            // gaussian_components_fixedvar.py:164-176: numerator -= p x, precision_N -= p
            const double *__restrict__ mS = d.m + (long long)s * D;
            const double *__restrict__ SS = d.S + (long long)s * 2 * D;
            double lpp = 0.0, a1 = 0.0;
            for (int l = 0; l < D; ++l) {
                const double x = XAT(l);
                const double p = d.prior_S[D + l];
                const double mn = __dsub_rn(mS[l], __dmul_rn(p, x));
                const double pN = __dsub_rn(SS[l], p);
                const double pp = pN * p / (pN + p);
                const double dl = x - mn / pN;
                lpp += log(pp);
                a1 += (dl * dl) * pp;
            }
            lp = -0.5 * (double)D * log(2.0 * 3.14159265358979323846) + 0.5 * lpp - 0.5 * a1;
        } else if (home == s && ns >= 2) {
            // the visited point removed from its own component
            const double *__restrict__ mS = d.m + (long long)s * D;
            const double *__restrict__ SS = d.S + (long long)s * D;
            const double k1 = d.k0 + (double)(ns - 1);
            const long long v1 = d.v0 + ns - 1;
            const double scale1 = (k1 + 1.0) / (k1 * (double)v1), inv_v1 = 1.0 / (double)v1;
            double lpv = 0.0, a1 = 0.0;
            for (int l = 0; l < D; ++l) {
                const double x = XAT(l);
                const double m1 = __dsub_rn(mS[l], x);
                const double S1 = __dsub_rn(SS[l], __dmul_rn(x, x));
                const double mean = m1 / k1;
                const double var = scale1 * (S1 - k1 * (mean * mean));
                const double dl = x - mean;
                lpv += log(var);
                a1 += log(1.0 + inv_v1 * (dl * dl) * (1.0 / var));
            }
            lp = (double)D * (d.tab_lgam[v1 + 1] - d.tab_lgam[v1] - 0.5 * d.tab_log[v1] - hlp)
                 - 0.5 * lpv - 0.5 * (double)(v1 + 1) * a1;
        }
        if (live) q[(long long)(col_override >= 0 ? col_override : s) * qstride + (p - job.win_base)] = lp;
    }
#undef XAT
}

void launch_score_diag(const Dev &d, const Job *job, double *q, long long qstride, int col_override,
                       long long max_rows, int skip_pruned_jobs, hipStream_t st) {
    if (max_rows <= 0) return;
    const unsigned gx = (unsigned)((max_rows + kValuRows - 1) / kValuRows);
    const int lds = (d.D <= kDiagLdsMaxD ? d.D * kDiagLd : 1) * (int)sizeof(double);
    hipLaunchKernelGGL(score_diag_kernel, dim3(gx, kMaxChunks), dim3(256), lds, st, d, job, q, qstride,
                       col_override, job == &d.ctrl->job ? 1 : 0, skip_pruned_jobs);
}


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
#include "bgmm_device.h"

typedef double v4d __attribute__((ext_vector_type(4)));

// log score of a (visit, slot) pair from its exact quadratic form -- the formulas of the draw kernel
__device__ __forceinline__ double slot_score_exact(const SlotConst &sc, double qv, bool home_minus_one) {
    if (home_minus_one) {
        const double den = 1.0 - sc.a1 * qv;
        return sc.logseat1 + sc.A1 - 0.5 * log(den) - sc.half_vd1 * log(1.0 + sc.coef1 * qv / den);
    }
    return sc.logseat + sc.A - sc.half_vd * log(1.0 + qv * sc.inv_cv);
}

// The Job is read field by field (a by-value copy with a dynamically indexed dirty[] member
// ends up in scratch memory).
struct JobView {
    long long pos, win_base, win_hi;
    int mode, nlist, chunks, dirty0, dirty1, prune;
};
__device__ __forceinline__ JobView load_job(const Job *__restrict__ j) {
    JobView v;
    v.pos = j->pos; v.win_base = j->win_base; v.win_hi = j->win_hi;
    v.mode = j->mode; v.chunks = j->chunks;
    v.nlist = v.mode == MODE_FRESH ? j->K : j->n_dirty;
    v.dirty0 = j->dirty[0]; v.dirty1 = j->dirty[1];
    v.prune = j->prune;
    return v;
}
// entry t of the job's list -> slot
__device__ __forceinline__ int job_slot(const Dev &d, const JobView &job, int t) {
    return job.mode == MODE_FRESH ? d.perm[t] : (t == 0 ? job.dirty0 : job.dirty1);
}

// ------------------------------------------------------------------------------------------
// VALU kernel.  q is stored slot-major: q[col * qstride + row].
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void score_valu_kernel(Dev d, const Job *__restrict__ jobp,
                                                         double *__restrict__ q, long long qstride,
                                                         int col_override, int skip_pruned_jobs) {
    extern __shared__ __attribute__((aligned(16))) double xs[];   // [D][64]
    const JobView job = load_job(jobp);
    if (job.mode == MODE_DONE || (skip_pruned_jobs && job.prune)) return;
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
template <int CTRL>
__device__ __forceinline__ double dpp_mov_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, false);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, false);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row16_sum(double v) {
    v += dpp_mov_f64<0xB1>(v);
    v += dpp_mov_f64<0x4E>(v);
    v += dpp_mov_f64<0x141>(v);
    v += dpp_mov_f64<0x140>(v);
    return v;
}

constexpr int pick_pf(int nf) {
    int best = 1;
    for (int p = 1; p <= 12 && p <= nf; ++p)
        if (nf % p == 0) best = p;
    return best;
}

// MINW = waves per SIMD the register budget is planned for: 3 up to D = 64 (168 VGPRs with a
// 10..12-deep ring), 2 at D = 80, 1 for D = 96..128 (2 x 16 rows of A fragments alone are
// 96..128 VGPRs).  bgmm_api.hip sizes the grid (label chunks) to a whole number of residency rounds.
template <int NJ, int RB, int MINW>
__global__ __launch_bounds__(256, MINW) void score_mfma_kernel(Dev d, const Job *__restrict__ jobp,
                                                            double *__restrict__ q, long long qstride,
                                                            int col_override, int skip_pruned_jobs) {
    const JobView job = load_job(jobp);
    if (job.mode == MODE_DONE || (skip_pruned_jobs && job.prune)) return;
    const int chunk = blockIdx.y;
    if (chunk >= job.chunks || chunk >= job.nlist) return;
    constexpr int ROWS_W = 16 * RB;              // rows per wave
    constexpr int NF = 2 * NJ * (NJ + 1);
    constexpr int PF = pick_pf(NF);
    const long long pb = job.pos + (long long)blockIdx.x * (4 * ROWS_W);
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
        const int tn = t + job.chunks;
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

// ------------------------------------------------------------------------------------------
// MFMA kernel with EXACT pruning of negligible components (fresh windows only).
//
// For a slot s with Lambda_s >= lambda_max(S_N) (Gershgorin at every from-scratch refresh, raised by
// a |d|^2 at every rank-1 addition; slot_math.h) the quadratic form is bounded from below by the
// Euclidean distance:   q_s(x) = (mu-x)' S_N^-1 (mu-x) >= |mu - x|^2 / Lambda_s =: q_lb,   and
//     lp_ub = logseat + A - half_vd * L(q_lb * inv_cv),     L(t) <= log(1 + t)  (cheap minorant)
// is a rigorous upper bound of the component's log score for that visit.  |mu - x|^2 for 16
// slots x 32 visits costs ONE v_mfma_f64_16x16x4 per slot and 16 visits (a [rows x D].[D x 16]
// product against the means) -- 2.5 % of the full quadratic form at D = 64.
// Every visit also has a lower bound M_lb of its maximum log score: the "new table" entry
// log(alpha) + log_prior[i] (crpmm.py:74) to start with, raised by every exact score computed so
// far.  A slot whose lp_ub < M_lb - kPruneMargin for all 32 visits of the wave, and that is
// nobody's home, has weight exp(lp - max) < e^-80 ~ 2e-35 in each of those draws -- twenty orders
// of magnitude below the rounding noise of the normaliser -- and is not scored: q = +inf is
// stored, which the draw kernel turns into an exact zero weight.  The bound holds against the
// frozen state only, which is why a pruned window ends at its first move (slot_math.h).
// ------------------------------------------------------------------------------------------
static constexpr double kPruneMargin = 80.0;

// minorant of log(1 + t), t >= 0: 2t/(2+t) below 1, (e + m - 1) ln 2 above (1+t = m 2^e, 1 <= m < 2)
__device__ __forceinline__ double log1p_lower(double t) {
    if (t < 1.0) return 2.0 * t / (2.0 + t);
    int e;
    const double m = frexp(1.0 + t, &e);               // 1+t = m 2^e, 0.5 <= m < 1
    return 0.6931471805599453 * ((double)(e - 1) + (2.0 * m - 1.0));
}

// In a pruned window the visits are evaluated in the order of d.wperm (grouped by home component,
// kernels_state.hip: bucket_rows_kernel), so that the visits of one wave mostly share a home and
// need the same one or two components in full.  q is indexed by that evaluation position.
template <int NJ, int RB, int MINW>
__global__ __launch_bounds__(256, MINW) void score_mfma_prune_kernel(Dev d, const Job *__restrict__ jobp,
                                                                  double *__restrict__ q, long long qstride) {
    const JobView job = load_job(jobp);
    if (job.mode != MODE_FRESH || !job.prune) return;
    const int chunk = blockIdx.y;
    if (chunk >= job.chunks || chunk >= job.nlist) return;
    constexpr int ROWS_W = 16 * RB;
    constexpr int NF = 2 * NJ * (NJ + 1);
    const long long nrows = job.win_hi - job.pos;                 // (a pruned window starts at win_base)
    const long long kb = (long long)blockIdx.x * (4 * ROWS_W);
    if (kb >= nrows) return;
    const int D = d.D;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long kw = kb + w * ROWS_W;                         // first evaluation position of the wave
    if (kw >= nrows) return;
    const int lr = lane & 15, lk = lane >> 4;

    double xf[RB][NJ * 4];
#pragma unroll
    for (int R = 0; R < RB; ++R) {
        const long long k = kw + R * 16 + lr;
        const bool live = k < nrows;
        const long long p = live ? job.win_base + d.wperm[k] : 0;
        const long long i = live ? (d.order ? d.order[p] : p) : 0;
        const double *__restrict__ xrow = d.X + i * D;
#pragma unroll
        for (int kk = 0; kk < NJ * 4; ++kk) {
            const int l = 4 * kk + lk;
            xf[R][kk] = (live && l < D) ? xrow[l] : 0.0;
        }
    }
    // Per accumulator element (visits lk + 4r of block R): |x|^2, the lower bound of the visit's best
    // log score, its home slot (never pruned).  Dead rows can never keep a slot alive.
    double x2[RB][4], Mlb[RB][4];
    int home[RB][4];
#pragma unroll
    for (int R = 0; R < RB; ++R) {
        // |x|^2 of row lr from the A fragments (sum over kk, then over the 4 lk lanes) ...
        double part = 0.0;
#pragma unroll
        for (int kk = 0; kk < NJ * 4; ++kk) part = fma(xf[R][kk], xf[R][kk], part);
        part += __shfl_xor(part, 16);
        part += __shfl_xor(part, 32);
        // ... re-distributed to the accumulator layout: rows lk + 4r
#pragma unroll
        for (int r = 0; r < 4; ++r) x2[R][r] = __shfl(part, lk + 4 * r);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long long k = kw + R * 16 + lk + 4 * r;
            if (k < nrows) {
                const long long p = job.win_base + d.wperm[k];
                const long long i = d.order ? d.order[p] : p;
                Mlb[R][r] = d.log_alpha + d.log_prior[i];
                home[R][r] = d.z[i];
            } else {
                Mlb[R][r] = INFINITY;
                home[R][r] = -2;
            }
        }
    }

    const long long nfrag64 = (long long)NF * 64;
    unsigned n_kept = 0, n_bound = 0;
    // groups of 16 list entries: lane column lr <-> entry t0 + chunks * lr
    for (int t0 = chunk; t0 < job.nlist; t0 += 16 * job.chunks) {
        const int tl = t0 + job.chunks * lr;
        const int sg = tl < job.nlist ? job_slot(d, job, tl) : -1;
        // distances to the 16 means: G = X . Mu'  (B fragment: mu_sg[4kk + lk])
        v4d accG[RB];
#pragma unroll
        for (int R = 0; R < RB; ++R) accG[R] = (v4d){0.0, 0.0, 0.0, 0.0};
        const double *__restrict__ mup = d.mu + (long long)(sg >= 0 ? sg : 0) * D + lk;
#pragma unroll
        for (int kk = 0; kk < NJ * 4; ++kk) {
            const double bm = (sg >= 0 && 4 * kk + lk < D) ? mup[4 * kk] : 0.0;
#pragma unroll
            for (int R = 0; R < RB; ++R)
                accG[R] = __builtin_amdgcn_mfma_f64_16x16x4f64(xf[R][kk], bm, accG[R], 0, 0, 0);
        }
        bool need[RB];
        {
            const SlotConst *__restrict__ scp = d.sc + (sg >= 0 ? sg : 0);
            const double base = scp->logseat + scp->A, hvd = scp->half_vd, icv = scp->inv_cv;
            const double ilam = scp->inv_lam, mu2 = scp->mu2;
#pragma unroll
            for (int R = 0; R < RB; ++R) {
                need[R] = false;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double nrm = x2[R][r] + mu2;
                    double dist2 = nrm - 2.0 * accG[R][r] - 1e-9 * nrm;     // (rounding of the difference)
                    dist2 = dist2 > 0.0 ? dist2 : 0.0;
                    const double ub = base - hvd * log1p_lower(dist2 * ilam * icv);
                    need[R] = need[R] || (ub >= Mlb[R][r] - kPruneMargin) || (home[R][r] == sg);
                }
                need[R] = need[R] && sg >= 0;
            }
        }
        // fold the votes of the 4 lk lanes (and 4 r's) of every slot column
        unsigned keepmask[RB];
#pragma unroll
        for (int R = 0; R < RB; ++R) {
            const unsigned long long bl = __ballot(need[R]);
            keepmask[R] = (unsigned)((bl | (bl >> 16) | (bl >> 32) | (bl >> 48)) & 0xFFFFull);
        }
        // pruned (block of 16 visits, slot) pairs: exact zero weight downstream
#pragma unroll
        for (int R = 0; R < RB; ++R) {
            if (sg >= 0 && !((keepmask[R] >> lr) & 1u)) {
                double *__restrict__ qc = q + (long long)sg * qstride;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long long k = kw + R * 16 + lk + 4 * r;
                    if (k < nrows) qc[k] = INFINITY;
                }
            }
        }
        // the slots somebody needs: full quadratic form, tiles fetched in batches of 8
        unsigned todo = 0;
#pragma unroll
        for (int R = 0; R < RB; ++R) {
            todo |= keepmask[R];
            n_kept += __popc(keepmask[R]);
        }
        {
            const int left = (job.nlist - t0 + job.chunks - 1) / job.chunks;
            n_bound += RB * (left < 16 ? left : 16);
        }
        while (todo) {
            const int jbit = __ffs(todo) - 1;
            todo &= todo - 1;
            const int s = __builtin_amdgcn_readlane(sg, jbit);
            const double *__restrict__ wf = d.Wfrag + (long long)s * nfrag64 + lane;
            const double *__restrict__ cvp = d.cvec + (long long)s * d.Dp + lr;
            double qp[RB][4];
#pragma unroll
            for (int R = 0; R < RB; ++R)
#pragma unroll
                for (int r = 0; r < 4; ++r) qp[R][r] = 0.0;
            // software pipeline over the slot's NF tiles: a ring of PFK loads in flight
            constexpr int PFK = NF < 16 ? NF : 16;
            double ringk[PFK];
#pragma unroll
            for (int i = 0; i < PFK; ++i) ringk[i] = wf[i * 64];
            double cjk[NJ];
#pragma unroll
            for (int J = 0; J < NJ; ++J) cjk[J] = cvp[16 * J];
#pragma unroll
            for (int J = 0; J < NJ; ++J) {
                v4d acc[RB];
#pragma unroll
                for (int R = 0; R < RB; ++R) acc[R] = (v4d){cjk[J], cjk[J], cjk[J], cjk[J]};
#pragma unroll
                for (int kk = 0; kk < 4 * (J + 1); ++kk) {
                    const int f = 2 * J * (J + 1) + kk;          // constant after unrolling
                    const double b = ringk[f % PFK];
                    if (f + PFK < NF) ringk[f % PFK] = wf[(f + PFK) * 64];
#pragma unroll
                    for (int R = 0; R < RB; ++R)
                        acc[R] = __builtin_amdgcn_mfma_f64_16x16x4f64(xf[R][kk], b, acc[R], 0, 0, 0);
                }
#pragma unroll
                for (int R = 0; R < RB; ++R)
#pragma unroll
                    for (int r = 0; r < 4; ++r) qp[R][r] = fma(acc[R][r], acc[R][r], qp[R][r]);
            }
            const SlotConst scs = d.sc[s];
            const int ns = d.n[s];
            double *__restrict__ qcol = q + (long long)s * qstride;
#pragma unroll
            for (int R = 0; R < RB; ++R) {
                const bool kept = (keepmask[R] >> jbit) & 1u;      // else this block's column holds +inf
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double v = row16_sum(qp[R][r]);
                    const long long k = kw + R * 16 + lk + 4 * r;
                    if (kept && lr == r && k < nrows) qcol[k] = v;
                    // an exact score is a lower bound of the visit's maximum (the home component
                    // counts with its one-point-removed form, exactly as the draw kernel scores it)
                    if (k < nrows) {
                        const bool own = home[R][r] == s;
                        if (!own || ns >= 2) Mlb[R][r] = fmax(Mlb[R][r], slot_score_exact(scs, v, own));
                    }
                }
            }
        }
    }
    if (lane == 0) {
        atomicAdd(&d.ctrl->n_kept_blocks, (unsigned long long)n_kept);
        atomicAdd(&d.ctrl->n_bound_blocks, (unsigned long long)n_bound);
    }
}

template <int NJ>
static void launch_mfma_prune(const Dev &d, const Job *job, double *q, long long qstride, long long max_rows,
                              hipStream_t st) {
    const unsigned gx = (unsigned)((max_rows + kMfmaRows - 1) / kMfmaRows);
    hipLaunchKernelGGL((score_mfma_prune_kernel<NJ, 2, (NJ <= 5 ? 2 : 1)>), dim3(gx, kMaxChunks),
                       dim3(256), 0, st, d, job, q, qstride);
}

// Fresh-window scoring with pruning
bool launch_score_pruned(const Dev &d, const Job *job, double *q, long long qstride, long long max_rows,
                         hipStream_t st) {
    if (max_rows <= 0) return true;
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
template <int NJ>
static void launch_mfma(const Dev &d, const Job *job, double *q, long long qstride, int col_override,
                        long long max_rows, int skip_pruned_jobs, hipStream_t st) {
    const unsigned gx = (unsigned)((max_rows + kMfmaRows - 1) / kMfmaRows);
    hipLaunchKernelGGL((score_mfma_kernel<NJ, 2, (NJ <= 4 ? 3 : (NJ <= 5 ? 2 : 1))>), dim3(gx, kMaxChunks), dim3(256), 0, st, d, job, q,
                       qstride, col_override, skip_pruned_jobs);
}

void launch_score_diag(const Dev &d, const Job *job, double *q, long long qstride, int col_override,
                       long long max_rows, hipStream_t st);

void launch_score(const Dev &d, int kind, const Job *job, double *q, long long qstride, int col_override,
                  long long max_rows, int skip_pruned_jobs, hipStream_t st) {
    if (max_rows <= 0) return;
    if (d.cov_type != COV_FULL) { launch_score_diag(d, job, q, qstride, col_override, max_rows, st); return; }
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
                                                         int col_override, int home_correction) {
    extern __shared__ __attribute__((aligned(16))) double xs[];   // [D][64]
    const JobView job = load_job(jobp);
    if (job.mode == MODE_DONE) return;
    const int chunk = blockIdx.y;
    if (chunk >= job.chunks) return;
    const long long p0 = job.pos + (long long)blockIdx.x * kValuRows;
    if (p0 >= job.win_hi) return;
    const int D = d.D;
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
    const bool live = p < job.win_hi;
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
                const double dl = xs[l * kValuRows + lane] - mu[l];
                acc += (dl * dl) * dw[l];
            }
        } else {
            for (int l = 0; l < D; ++l) {
                const double dl = xs[l * kValuRows + lane] - mu[l];
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
                const double x = xs[l * kValuRows + lane];
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
                const double x = xs[l * kValuRows + lane];
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
}

void launch_score_diag(const Dev &d, const Job *job, double *q, long long qstride, int col_override,
                       long long max_rows, hipStream_t st) {
    if (max_rows <= 0) return;
    const unsigned gx = (unsigned)((max_rows + kValuRows - 1) / kValuRows);
    const int lds = d.D * kValuRows * (int)sizeof(double);
    hipLaunchKernelGGL(score_diag_kernel, dim3(gx, kMaxChunks), dim3(256), lds, st, d, job, q, qstride,
                       col_override, job == &d.ctrl->job ? 1 : 0);
}

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
// Per wavefront 64 rows, lane = row for everything scalar (the record, the tail).  The matrix part takes
// them 16 RB at a time: staged through a wave-private LDS tile (row-contiguous 512-byte loads in, A
// fragments out), per distinct home among them (one, at a boundary two) the inverse factor's tiles
// streamed through a register ring; quadratic forms and distances meet their rows' lanes through LDS.
#include "score_common.h"
#include "wave_ops.h"
#include "fast_math.h"

static constexpr double kHomeMargin = 80.0;          // = kPruneMargin of kernels_prune.hip

__host__ __device__ constexpr int home_row_stride(int Dp) { return ((Dp + 27) / 32) * 32 + 4; }
__host__ __device__ constexpr int home_wave_doubles(int Dp, int RB) { return 16 * home_row_stride(Dp) + 64 + 64 + 32 + 0 * RB; }

template <int NJ, int RB>
__global__ __launch_bounds__(256, 2) void home_kernel(Dev d) {
    extern __shared__ __attribute__((aligned(16))) double lds_all[];
    Ctrl *c = d.ctrl;
    if (!job_is_pruned(d, c->job.mode, c->job.prune)) return;
    constexpr int NF = 2 * NJ * (NJ + 1), NKK = NJ * 4, Ds = home_row_stride(NJ * 16);
    constexpr int PFK = pick_ring(NF, 10);
    constexpr int CH = 16 * RB, NCH = 64 / CH;                        // rows per chunk, chunks per wave
    const long long nrows = c->n_sorted;
    const long long kb = (long long)blockIdx.x * 256;
    if (kb >= nrows) return;
    const int D = d.D, K = c->job.K;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long kw = kb + w * 64;
    if (kw >= nrows) return;
    const int lr = lane & 15, lk = lane >> 4;
    double *__restrict__ xs = lds_all + w * home_wave_doubles(NJ * 16, RB);   // [16][Ds] staging tile
    double *__restrict__ sideQ = xs + 16 * Ds;                        // exact home form of row rho [64]
    double *__restrict__ sideRho = sideQ + 64;                        // |x - mu_home|^2 [64]
    int *__restrict__ sideH = (int *)(sideRho + 64);                  // home slot [64]

    // lane rho owns row kw + rho
    const long long kmine = kw + lane;
    WRec rmine;
    int wrow = 0;
    if (kmine < nrows) { rmine = d.wrec[kmine]; wrow = d.wperm[kmine]; }
    else { rmine.i = -1; rmine.home = -2; rmine.home_label = -1; rmine.mlb0 = -INFINITY; }
    const long long imine = rmine.i;
    const int hmine = rmine.home;
    sideH[lane] = hmine;
    unsigned n_mfma = 0, n_homes = 0;
    constexpr int NP = (NJ * 16 + 63) / 64;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ++ch) {
        const int r0 = ch * CH;                                      // first row of the chunk
        if (kw + r0 >= nrows) break;
        // rows -> A fragments, 16 rows at a time through the wave's tile
        double xf[RB][NKK];
#pragma unroll
        for (int R = 0; R < RB; ++R) {
            double tmp[16][NP];
#pragma unroll
            for (int row = 0; row < 16; ++row) {
                const long long i = wv_readlane_i64(imine, r0 + R * 16 + row);
                const double *__restrict__ xrow = d.X + (i >= 0 ? i : 0) * D;
#pragma unroll
                for (int pss = 0; pss < NP; ++pss) {
                    const int l = pss * 64 + lane;
                    tmp[row][pss] = xrow[l < D ? l : 0];
                }
            }
#pragma unroll
            for (int row = 0; row < 16; ++row) {
                const long long i = wv_readlane_i64(imine, r0 + R * 16 + row);
#pragma unroll
                for (int pss = 0; pss < NP; ++pss) {
                    const int l = pss * 64 + lane;
                    const double v = (i >= 0 && l < D) ? tmp[row][pss] : 0.0;
                    if (NJ * 16 >= (pss + 1) * 64 || l < NJ * 16) xs[row * Ds + l] = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (the tile is private to the wave)
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) xf[R][kk] = xs[lr * Ds + 4 * kk + lk];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        // homes of the rows this lane's accumulator rows / fragment row belong to
        int hq[RB][4], hd[RB];
#pragma unroll
        for (int R = 0; R < RB; ++R) {
#pragma unroll
            for (int r = 0; r < 4; ++r) hq[R][r] = sideH[r0 + R * 16 + lk + 4 * r];
            hd[R] = sideH[r0 + R * 16 + lr];
        }
        // ---- the homes present in the chunk, one after the other (the rows are grouped by home)
        unsigned long long pending = __ballot(lane >= r0 && lane < r0 + CH && hmine >= 0);
#pragma unroll 1
        while (pending) {
            const int first = __ffsll((long long)pending) - 1;
            const int s = __builtin_amdgcn_readlane(hmine, first);
            pending &= ~__ballot(hmine == s);
            const double *__restrict__ wf = d.Wfrag + (long long)s * (NF * 64) + lane;
            double ringk[PFK];
#pragma unroll
            for (int i = 0; i < PFK; ++i) ringk[i] = wf[i * 64];
            const double *__restrict__ cvp = d.cvec + (long long)s * d.Dp + lr;
            const double *__restrict__ mup = d.mu + (long long)s * D;
            double cjk[NJ];
#pragma unroll
            for (int J = 0; J < NJ; ++J) cjk[J] = cvp[16 * J];
            // squared distance of the rows to this home's mean, from the fragments (lane: NKK of the row's entries)
            double dpart[RB];
#pragma unroll
            for (int R = 0; R < RB; ++R) dpart[R] = 0.0;
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const int l = 4 * kk + lk;
                const double m = l < D ? mup[l] : 0.0;
#pragma unroll
                for (int R = 0; R < RB; ++R) { const double t = xf[R][kk] - m; dpart[R] = fma(t, t, dpart[R]); }
            }
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
                    const int f = 2 * J * (J + 1) + kk;
                    const double bfr = ringk[f % PFK];
                    if (f + PFK < NF) ringk[f % PFK] = wf[(f + PFK) * 64];
#pragma unroll
                    for (int R = 0; R < RB; ++R)
                        acc[R] = __builtin_amdgcn_mfma_f64_16x16x4f64(xf[R][kk], bfr, acc[R], 0, 0, 0);
                }
#pragma unroll
                for (int R = 0; R < RB; ++R)
#pragma unroll
                    for (int r = 0; r < 4; ++r) qp[R][r] = fma(acc[R][r], acc[R][r], qp[R][r]);
            }
            n_mfma += RB * NF;
            n_homes += RB;
            // row sums: quadratic forms (accumulator rows lk + 4 r, complete in all 16 lanes of the row group),
            // distances (fragment row lr: summed over the 4 lk groups) -- to the rows' own lanes through LDS
#pragma unroll
            for (int R = 0; R < RB; ++R) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const double v = row16_sum(qp[R][r]);
                    if (lr == r && hq[R][r] == s) sideQ[r0 + R * 16 + lk + 4 * r] = v;
                }
                double dd = dpart[R];
                dd += __shfl_xor(dd, 16);
                dd += __shfl_xor(dd, 32);
                if (lk == 0 && hd[R] == s) sideRho[r0 + R * 16 + lr] = dd;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    // ---- the scalar tail: lane = row
    const bool live = imine >= 0;
    bool easy = false;
    if (live && hmine >= 0) {
        const double q_t = sideQ[lane], rho2_t = sideRho[lane];
        const int nh = d.n[hmine];
        const int a = rmine.home_label;
        // per-point cache for certify_kernel (bgmm_device.h: PCache)
        PCache pc;
        pc.tag = ((long long)hmine << 32) | (unsigned int)d.mu_ver[hmine];
        pc.qhome = q_t; pc.rho2 = rho2_t; pc.pad = 0.0;
        d.pcache[imine] = pc;
        if (nh >= 2) {
            const SlotConst sc = d.sc[hmine];
            // the visited point removed from its own component (slot_math.h: home form)
            const double den = 1.0 - sc.a1 * q_t;
            const double vh = sc.logseat1 + sc.A1 - 0.5 * fm_log(den) - sc.half_vd1 * fm_log(1.0 + fm_div(sc.coef1 * q_t, den));
            const double vnew = rmine.mlb0;
            const double mx = fmax(vh, vnew);
            const double rad = sqrt(rho2_t * (1.0 + 1e-9)) * (1.0 + 1e-9);
            const double jf = rad * d.finv[a];
            if (den > 0.0 && jf < 62.0 && d.ftab[(long long)a * 64 + (int)jf + 1] < mx - kHomeMargin) {
                // two candidates: prob = exp(lp - logsumexp), u -= prob in label order (crpmm.py:75, utils.py:15-20)
                easy = true;
                const double eh = fm_exp(vh - mx), en = fm_exp(vnew - mx);
                const double lse = fm_log(eh + en) + mx;
                const long long p = c->job.win_base + wrow;
                double uu = d.u[p];
                uu -= fm_exp(vh - lse);
                const int pick = uu < 0.0 ? a : K;
                d.choice[wrow] = pick;
                PCacheExact pe;
                pe.epoch = c->state_epoch;
                // log of the alternatives' total weight relative to the home's: the new table exactly, every other
                // label below e^-80 of the best
                pe.log_alt = mx - vh + fm_log(en + (double)K * 1.8048513878454153e-35);
                d.pcache2[imine] = pe;
                if (pick != a) atomicMin(&c->first_mover, (unsigned long long)p);
            }
        }
    }
    // ---- what the table bound could not decide goes on the residual list, a contiguous run per wave
    const bool hard = live && !easy;
    const unsigned long long mhard = __ballot(hard);
    if (mhard) {
        const int fl = __ffsll((long long)mhard) - 1;
        int base = 0;
        if (lane == fl) base = atomicAdd(&c->n_resid, __popcll(mhard));
        base = __builtin_amdgcn_readlane(base, fl);
        if (hard) {
            const int pos = base + __popcll(mhard & ((1ull << lane) - 1ull));
            d.wrecR[pos] = rmine;
            d.wpermR[pos] = wrow;
        }
    }
    if (lane == 0) {
        atomicAdd(&d.pr_counts[blockIdx.x & 255], (unsigned long long)n_homes);
        atomicAdd(&d.pr_counts[512 + (blockIdx.x & 255)], (unsigned long long)n_mfma);
    }
}

template <int NJ>
static void launch_home_t(const Dev &d, long long max_rows, hipStream_t st) {
    constexpr int RB = NJ <= 5 ? 2 : 1;
    const unsigned gx = (unsigned)((max_rows + 255) / 256);
    const int lds = 4 * home_wave_doubles(NJ * 16, RB) * (int)sizeof(double);
    static PerDeviceLds attr;
    if (lds > 64 * 1024 && attr.raise(lds))
        (void)hipFuncSetAttribute((const void *)home_kernel<NJ, RB>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((home_kernel<NJ, RB>), dim3(gx), dim3(256), lds, st, d);
}

void launch_home(const Dev &d, long long max_rows, hipStream_t st) {
    if (max_rows <= 0) return;
    switch (d.Dp / 16) {
        case 1: launch_home_t<1>(d, max_rows, st); return;
        case 2: launch_home_t<2>(d, max_rows, st); return;
        case 3: launch_home_t<3>(d, max_rows, st); return;
        case 4: launch_home_t<4>(d, max_rows, st); return;
        case 5: launch_home_t<5>(d, max_rows, st); return;
        case 6: launch_home_t<6>(d, max_rows, st); return;
        case 7: launch_home_t<7>(d, max_rows, st); return;
        case 8: launch_home_t<8>(d, max_rows, st); return;
        default: return;
    }
}

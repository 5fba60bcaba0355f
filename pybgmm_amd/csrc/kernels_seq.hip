// The one-workgroup sweep for tiny dimensions (bgmm_device.h: launch_sweep_seq).
//
// Reference behaviour restated: the visit loop of igmm/crpmm.py:57-88 and igmm/pcrpmm.py:93-131 with
// del_item / add_item / log_post_pred of gaussian/gaussian_components.py:154-251 and the draw of
// utils/utils.py:7-20 -- the same arithmetic as the windowed kernels (kernels_choice.hip, slot_math.h),
// laid out for a state that fits in LDS.
#include "bgmm_device.h"
#include "slot_math.h"
#include "fast_math.h"

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

// The visit is one dependent instruction chain: the short forms of fast_math.h (about a quarter of the library
// routines' instructions, error below 1 ulp) wherever their argument is in range.
__device__ __forceinline__ double seq_log(double x) { return (x > 1e-300 && x < 1e300) ? fm_log(x) : log(x); }

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

// make_consts_from (slot_math.h) with the short division
__device__ __forceinline__ SlotConst seq_consts(const Dev &d, int n, const SlotTab &t, double logdetC, double lam,
                                                double mu2) {
    SlotConst c;
    const double Dd = (double)d.D;
    const double k_N = d.k0 + (double)n;
    const long long v = d.v0 + n - d.D + 1;
    c.logdetC = logdetC;
    c.A = t.g - 0.5 * (Dd * t.lc + logdetC);
    c.half_vd = 0.5 * (double)(v + d.D);
    c.inv_cv = fm_div(k_N, k_N + 1.0);                 // 1 / (cs v),  cs = (k_N + 1) / (k_N v)
    c.logseat = t.seat;
    c.logseat1 = t.seat1;
    c.A1 = 0.0; c.half_vd1 = 0.0; c.coef1 = 0.0; c.a1 = 0.0;
    if (n >= 2) {
        const double k1 = k_N - 1.0;
        const double a = fm_div(k_N, k1);
        c.a1 = a;
        c.A1 = t.g1 - 0.5 * (Dd * t.lc1 + logdetC);
        c.half_vd1 = 0.5 * (double)(v - 1 + d.D);
        c.coef1 = a * a * fm_div(k1, k_N);             // a^2 / (c1 v1),  c1 = k_N / (k1 v1)
    }
    c.inv_lam = (lam > 0.0 && lam < 1e300) ? fm_div(1.0, lam) : 0.0;
    c.mu2 = mu2;
    return c;
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
    const double inv_kN = fm_div(1.0, k_N);
#pragma unroll
    for (int a = 0; a < DD; ++a) mu[a] = st[a] * inv_kN;
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
        const double rpiv = fm_rsqrt(djj > 0.0 ? djj : 1.0);
#pragma unroll
        for (int i = j + 1; i < DD; ++i) A[i][j] = A[i][j] * rpiv;
        A[j][j] = rpiv;                                // (the reciprocal of the pivot: all the inverse below needs)
#pragma unroll
        for (int i = j + 1; i < DD; ++i)
#pragma unroll
            for (int l = j + 1; l <= i; ++l) A[i][l] = fma(-A[i][j], A[l][j], A[i][l]);
    }
    // logdet S_N = log of the product of the pivots: one logarithm on the chain instead of DD (the
    // sum of logs when the product leaves the comfortable range)
    double ldt;
    if (piv_prod > 1e-200 && piv_prod < 1e200) {
        ldt = fm_log(piv_prod);
    } else {
        ldt = 0.0;
#pragma unroll
        for (int j = 0; j < DD; ++j) ldt -= log(A[j][j]);
        ldt *= 2.0;
    }
    if (bad || !(ldt == ldt)) atomicCAS(&d.ctrl->error, 0, -4);
#pragma unroll
    for (int i = 0; i < DD; ++i) {                     // inverse of the factor, row by row
        const double inv_d = A[i][i];
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
    const SlotConst sc = seq_consts(d, n, tab, ldt, lam, mu2);
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
__device__ __forceinline__ void sweep_seq_body(const Dev &d, int cap) {
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
    const long long N = c->n_visits;                  // (d.N, or fewer: bgmm_set_sweep_visits)
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
                    double arg = 1.0 + fm_div(num, den);
                    double hv = homeform ? C[6 * cap] : C[2 * cap];
                    double base = homeform ? C[4 * cap] + C[5 * cap] : C[0] + C[cap];
                    if (is_aux) arg = den;
                    if (!is_lab) { hv = 0.0; base = lane == L ? d.log_alpha + lp : -INFINITY; if (!is_aux) arg = 1.0; }
                    const double lg = seq_log(arg);
                    const double lg_aux = readlane_f64(lg, L + 1);
                    if (homeform && is_lab) base = base - 0.5 * lg_aux;
                    const double v = base - hv * lg;
                    const double mx = seq_wave_max(v);
                    const double e = fm_exp(v - mx);
                    const double tot = seq_wave_sum(e);
                    const double cum = seq_wave_scan(fm_div(e, tot), lane);
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
        PF(6)
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
        PF(7)
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
        for (int a = DD; a < d.Dp; ++a) d.cvec[(long long)s * d.Dp + a] = 0.0;   // (the MFMA-shaped kernels read all Dp entries)
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
        atomicAdd(&c->n_pairs_exact, (unsigned long long)lik);
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
__global__ __launch_bounds__(64 * kSeqWaves) void sweep_seq_kernel(Dev d, int cap) { sweep_seq_body<DD>(d, cap); }

// Several chains in one launch (bgmm_group_sweep_staged): workgroup b runs the whole sweep of the chain whose device view
// is group[b] -- every chain's labels in its own workgroup's LDS, one compute unit each.
template <int DD>
__global__ __launch_bounds__(64 * kSeqWaves) void sweep_seq_group_kernel(const Dev *__restrict__ group, int cap) {
    const Dev d = group[blockIdx.x];
    sweep_seq_body<DD>(d, cap);
}

template <int DD>
static hipError_t launch_seq_t(const Dev &d, int cap, int lds, hipStream_t st, const Dev *group, int n_group) {
    if (group) {
        hipError_t e = hipFuncSetAttribute((const void *)sweep_seq_group_kernel<DD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(sweep_seq_group_kernel<DD>, dim3((unsigned)n_group), dim3(64 * kSeqWaves), lds, st, group, cap);
        return hipSuccess;
    }
    hipError_t e = hipFuncSetAttribute((const void *)sweep_seq_kernel<DD>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sweep_seq_kernel<DD>, dim3(1), dim3(64 * kSeqWaves), lds, st, d, cap);
    return hipSuccess;
}

// cap: labels (plus the next free slot) the LDS plan holds.  Returns false when D is out of range.
// group / n_group: device array of the chains' views, one workgroup each (all of dimension d.D, all planned for `cap`).
bool launch_sweep_seq(const Dev &d, int cap, hipStream_t st, const Dev *group, int n_group) {
    const int lds = sweep_seq_lds_bytes(d.D, cap);
    switch (d.D) {
        case 1: return launch_seq_t<1>(d, cap, lds, st, group, n_group) == hipSuccess;
        case 2: return launch_seq_t<2>(d, cap, lds, st, group, n_group) == hipSuccess;
        case 3: return launch_seq_t<3>(d, cap, lds, st, group, n_group) == hipSuccess;
        case 4: return launch_seq_t<4>(d, cap, lds, st, group, n_group) == hipSuccess;
        default: return false;
    }
}


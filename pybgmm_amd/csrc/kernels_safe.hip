// Safe-stay windows: the regime between "nothing moves" and "everything moves" (0.01 % .. 25 % movers).
//
// A frozen-factor window (kernels_gram.hip) walks 64 CONSECUTIVE visits whether four of them move or sixty; a
// pruned window (kernels_prune.hip, kernels_home.hip) is thrown away at its first mover.  In between lies where a
// chain spends its life.  The reference's loop (igmm/crpmm.py:57-88, igmm/pcrpmm.py:93-131) restores the cached
// statistics verbatim when a visit keeps its component (crpmm.py:82-85): a visit that stays changes nothing, so it
// need not be on the resolver's sequential chain at all -- IF it can be proven to stay not only under the frozen
// state but under every state the window can reach.  That proof:
//
//   Every change of a component inside a window is a rank-1 term of its augmented scatter matrix
//   A = [[S, m], [m', k_N]], A += sigma [y; 1][y; 1]'; with c(x, x) = [x; 1]' A^-1 [x; 1] the predictive of x is
//       not a member:  lp = seat(n) + g(n) - D/2 lc(n) - 1/2 logdet S_N - hv(n) log(k_N / (k_N + 1) (1 + c))
//       a member    :  lp = seat(n-1) + g(n-1) - D/2 lc(n-1) - 1/2 logdet S_N + (hv(n-1) - 1/2) log(k_N / (k_N - 1) (1 - c))
//   (gaussian_components.py:228-251 through slot_math.h), logdet S_N = logdet A - log k_N.  Sherman-Morrison:
//       c_t(x, x) = c_{t-1}(x, x) - sigma_t c_{t-1}(x, y_t)^2 / D_t,   D_t = 1 + sigma_t c_{t-1}(y_t, y_t),
//       logdet A_t = logdet A_{t-1} + log D_t,
//   and by Cauchy-Schwarz on the positive definite form c_{t-1}:  c_{t-1}(x, y)^2 <= c_{t-1}(x, x) c_{t-1}(y, y), hence
//       c_{t-1}(x, x) / (1 + c_{t-1}(y, y)) <= c_t(x, x) <= c_{t-1}(x, x)         (sigma = +1)
//       c_{t-1}(x, x) <= c_t(x, x) <= c_{t-1}(x, x) / (1 - c_{t-1}(y, y))         (sigma = -1)
//   i.e. with W_t = sum_{i <= t} |log D_i| (which the window resolver has at hand: it computes every D_i):
//       | log c_t(x, x) - log c_0(x, x) | <= W_t   and   | logdet A_t - logdet A_0 | <= W_t      for EVERY x.
//   The resolver ends a window right behind the move that takes a column out of its budget W <= cap, lets its count
//   drift more than dn(t) = clamp(n0 / 4, 1, kSafeDn) from the frozen one, or opens a component (kernels_gram.hip).
//   Under that budget, for a visit x whose home h keeps >= 2 members:
//       home        >= Psi_min(h) - 1/2 logdet0_h - cap/2 + hv1_max(h) log(1 - c_0(x, x) e^cap)          =: lb
//       other t     <= Phi_max(t) - 1/2 logdet0_t + cap/2 - hv_min(t) log(1 + c_lb,t(x) e^-cap),
//                      c_lb,t = |mu_t - x|^2 / Lambda_t + 1 / k_N0 from the triangle bound of the pruning kernel
//                      against the FROZEN centres (tabulated per home over the distance to the home's mean: ftabR)
//       new table   =  log alpha + log_prior[i]                                                         (exact)
//   (Phi / Psi: the count-dependent constants, extremised over the counts the budget allows.)  With R = the total
//   weight of everything but the home relative to lb, the reference's draw (utils/utils.py:7-20: u -= p_j in label
//   order) returns the home for every u in [R, 1 - R]: the labels in front of it subtract at most R, the home's own
//   probability is at least 1 - R.  Such a visit is SAFE: it stays whatever the listed visits do.
//
// A step = proof pass over the next L visits (bucket sort + home_kernel in safe mode: one exact quadratic form per
// visit, the same kernel that decides a chain at rest) -> the first 64 visits that are not SAFE, in visiting order
// (safe_compact_kernel) -> the frozen-factor kernels on exactly those rows.  L follows the density of unproven visits.
#include "score_common.h"
#include "slot_math.h"
#include "wave_ops.h"

// One thread: the open window becomes the stretch of the next proof pass (first step of a batch; later ones are
// opened by the resolver that closes the window before).
__device__ __forceinline__ void safe_open_body(const Dev &d) {
    Ctrl *c = d.ctrl;
    if (c->error != 0 || c->job.mode == MODE_DONE) return;
    for (int b = threadIdx.x; b < d.nslots + 2; b += 256) d.bucket_bins[b] = 0;      // (the bucket sort counts from zero)
    if (threadIdx.x == 0) {
        c->safe_epoch_valid = 0;
        safe_open_window(d, c);
        // (the look-ahead's ring starts empty with every batch: other kinds of windows may have changed the state in between)
        c->ah_chunk[0] = c->ah_chunk[1] = -1;
        c->ah_req_chunk = -1;
        c->ah_seq[0] = c->win_seq; c->ah_seq[1] = 0;      // ([1]: no plan has been handed to the second stream yet)
        c->ah_hseq[0] = c->ah_hseq[1] = c->win_seq;
    }
}

// Robust per-label constants (rtab[label][8]) for the frozen state:
//   0 ub0 = Phi_max - logdet0/2 + cap/2    1 hv_min    2 1/Lambda    3 1/k_N0
//   4 lb0 = Psi_min - logdet0/2 - cap/2    5 hv1_max   6 1: the label keeps >= 2 members whatever the window does
//   7 dn: how far the resolver lets the label's count drift inside a window (negative: a small label, upwards only)
// row nslots - 1: 0 e^-cap, 1 e^cap
__device__ __forceinline__ void safe_rtab_body(const Dev &d) {
    const Ctrl *c = d.ctrl;
    if (c->error != 0 || c->job.mode != MODE_FRESH || c->safe_epoch_valid) return;
    if (c->safe_epoch_built == c->state_epoch && c->safe_cap_built == safe_cap_now(d, c)) return;
    const int K = c->job.K;
    const double Dd = (double)d.D, cap = safe_cap_now(d, c);
    const int a = blockIdx.x, lane = threadIdx.x;
    if (a < K) {
        // one wavefront per label, lane = one of the counts the window may take it to.  A SMALL label (fewer than
        // kSafeSmall members) proves nothing for its own members and asks nothing of the removals it takes: its count may
        // fall to 1, its scatter matrix never falls below the prior's (logdet S_N >= logdet S_0), and only the points
        // that JOIN it count against its budget (they alone can lower c(x, x)).
        const int s = d.perm[a];
        const int n0 = d.n[s];
        const bool small = n0 < kSafeSmall;
        const double kN0 = d.k0 + (double)n0, L0 = small ? d.sc[d.K_max].logdetC : d.sc[s].logdetC;
        int dn = n0 / 4;
        dn = dn < 1 ? 1 : (dn > kSafeDn ? kSafeDn : dn);
        const int n = small ? 1 + lane : n0 - dn + lane;
        double phi = -INFINITY, psi = INFINITY, hvmin = INFINITY, hv1max = 0.0;
        if (n >= 1 && n <= n0 + dn) {
            const SlotTab t = load_slot_tab(d, n);
            const double kN = d.k0 + (double)n;
            const long long v = d.v0 + n - d.D + 1;
            const double hv = 0.5 * (double)(v + d.D);
            const double lk = small ? 0.0 : -0.5 * log(kN0 / kN);
            phi = t.seat + t.g - 0.5 * Dd * t.lc + lk + hv * log((kN + 1.0) / kN);
            hvmin = hv;
            if (n >= 2) {
                hv1max = 0.5 * (double)(v - 1 + d.D) - 0.5;
                psi = t.seat1 + t.g1 - 0.5 * Dd * t.lc1 + lk + hv1max * log(kN / (kN - 1.0));
            }
        }
        phi = wv_max(phi); psi = -wv_max(-psi); hvmin = -wv_max(-hvmin); hv1max = wv_max(hv1max);
        if (lane == 0) {
            double *g = d.rtab + (long long)a * 8;
            g[0] = phi - 0.5 * L0 + (small ? 0.0 : 0.5 * cap) + 1e-9 * fabs(phi - 0.5 * L0);
            g[1] = hvmin;
            g[2] = d.sc[s].inv_lam;
            g[3] = 1.0 / kN0;
            g[4] = psi - 0.5 * L0 - 0.5 * cap - 1e-9 * fabs(psi - 0.5 * L0);
            g[5] = hv1max;
            g[6] = (!small && n0 - dn >= 2) ? 1.0 : 0.0;
            g[7] = small ? -(double)dn : (double)dn;          // (negative: a small label -- removals are free)
            // the bound constants of the pruning kernel (pr_const, written by prune_tables_kernel and already consumed by
            // prune_ftable_kernel): their robust twins -- ub <= ub0 - hv_min log(1 + |mu - x|^2 e^-cap / Lambda)
            double *pc = d.pr_const + (long long)(a >> 4) * 128 + (a & 15);
            pc[0] = g[0];
            pc[16] = hvmin;
            pc[32] = d.sc[s].inv_lam * (exp(-cap) * (1.0 - 1e-12));
        }
    }
    if (a == 0 && lane == 0) {
        double *g = d.rtab + (long long)(d.nslots - 1) * 8;
        g[0] = exp(-cap) * (1.0 - 1e-12);
        g[1] = exp(cap) * (1.0 + 1e-12);
    }
}

// log(1 + x) from BELOW in single precision: the bounds carry explicit slack for the 1e-6 relative error of the
// hardware logarithm and the rounding of 1 + x (bounds, not scores: a nat in a thousand does not matter, the
// 100-instruction double logarithm per (home, label, radius) did -- 120 us per table)
__device__ __forceinline__ double log1p_below(double x) {
    const float L = __logf(1.0f + (float)x);
    const double v = (double)L * (1.0 - 2e-4) - 3e-7;
    return v > 0.0 ? v : 0.0;
}
// exp(x) from ABOVE in single precision, for x <= 0.7 or so (weights relative to a lower bound of the home's)
__device__ __forceinline__ double exp_above(double x) {
    if (x < -80.0) return 1.9e-35;
    return (double)__expf((float)x + 1e-5f) * (1.0 + 1e-5) + 1e-37;
}

// ftabR[a][j]: upper bound, valid under the budget, of the log of the TOTAL weight of the labels t != a for a visit
// whose home is a and whose distance to a's (frozen) mean is at most j / finv[a] -- the robust twin of
// prune_ftable_kernel (same radii, same centre distances, built right behind it).  The labels' constants are staged
// in LDS once per workgroup.
__global__ __launch_bounds__(256) void safe_ftab_kernel(Dev d) {
    extern __shared__ double lt[];                        // [K][4]: ub0, hv_min, e^-cap / Lambda, e^-cap / k_N0; [K] distances
    __shared__ double red[4][64];
    const Ctrl *c = d.ctrl;
    if (c->error != 0 || c->job.mode != MODE_FRESH || c->safe_epoch_valid) return;
    if (c->safe_epoch_built == c->state_epoch && c->safe_cap_built == safe_cap_now(d, c)) return;
    // thread = (radius j, quarter of the labels): the 64 radii of home a, the other labels dealt to four wavefronts
    const int K = c->job.K, a = blockIdx.x, j = threadIdx.x & 63, part = threadIdx.x >> 6;
    if (a >= K) return;
    const double emcap = d.rtab[(long long)(d.nslots - 1) * 8 + 0];
    for (int t = threadIdx.x; t < K; t += 256) {
        const double *__restrict__ g = d.rtab + (long long)t * 8;
        lt[4 * t] = g[0]; lt[4 * t + 1] = g[1]; lt[4 * t + 2] = g[2] * emcap; lt[4 * t + 3] = g[3] * emcap;
    }
    double *dcs = lt + 4 * K;                             // centre distances from a
    const double *__restrict__ dc = d.pr_dcc + (long long)a * d.nslots;
    for (int t = threadIdx.x; t < K; t += 256) dcs[t] = dc[t] * (1.0 - 1e-9);
    __syncthreads();
    const double finv = d.finv[a];
    const double step = finv > 0.0 ? 1.0 / finv : 0.0;
    const double rj = (double)j * step * (1.0 + 1e-9);
    auto ub = [&](int t) {
        double dl = dcs[t] - rj;
        dl = dl > 0.0 ? dl : 0.0;
        return lt[4 * t] - lt[4 * t + 1] * log1p_below(dl * dl * lt[4 * t + 2] + lt[4 * t + 3]);
    };
    double f = -INFINITY;
    for (int t = part; t < K; t += 4)
        if (t != a) f = fmax(f, ub(t));
    red[part][j] = f;
    __syncthreads();
    f = fmax(fmax(red[0][j], red[1][j]), fmax(red[2][j], red[3][j]));
    __syncthreads();
    // log of the sum: the labels within 45 nats of the largest bound are added up, the rest (each below e^-45 of it)
    // are covered by K e^-45  (the order of the additions differs from a serial loop's by rounding only, and the
    // result is an UPPER bound with 1e-12 to spare)
    double sum = 0.0;
    for (int t = part; t < K; t += 4) {
        if (t == a) continue;
        const double v = ub(t);
        if (v > f - 45.0) sum += exp_above(v - f);
    }
    red[part][j] = sum;
    __syncthreads();
    if (part != 0) return;
    sum = (double)K * 2.9e-20 + ((red[0][j] + red[1][j]) + (red[2][j] + red[3][j]));
    f += log(sum) * (1.0 + 1e-12) + 1e-12;
    d.ftabR[(long long)a * 64 + j] = (K > 1 && finv > 0.0) ? f : (K > 1 ? INFINITY : -INFINITY);
}

// The visits of the stretch [win_base, win_hi) that the proof pass did not mark SAFE (cert[row] == 1: SAFE), in
// visiting order -- at most kSafeList of them: the stretch ends at the next one -- and the start of an epoch: every
// label's budget account opened (Dev::ep_state), the first window's rows set.  One workgroup: every thread counts a
// contiguous run, one scan, the threads in front of the cut emit.
__device__ __forceinline__ void safe_compact_body(const Dev &d) {
    __shared__ int wsum[16];
    Ctrl *c = d.ctrl;
    if (c->error != 0 || c->job.mode != MODE_FRESH || c->safe_epoch_valid) return;
    const long long base = c->job.win_base;
    const int nrows = (int)(c->job.win_hi - base);
    // With the look-ahead's ring a proof pass is a verdict kernel, not a likelihood launch, and what a step costs is its
    // window's fixed part (cross forms, weights, finish, ten launches): the stretch then ends right behind the visits that
    // FILL one window -- the list holds kGramRows of them, the next stretch starts at the one after -- instead of carrying
    // a list whose last window is a partial one.
    const int cap = (d.safe_dense && d.ahead_C > 0) ? kGramRows : kSafeList;
    const int per = ((nrows + 1023) / 1024 + 15) & ~15;             // (16-byte pieces)
    const int lo = threadIdx.x * per, hi = lo + per < nrows ? lo + per : nrows;
    int cnt = 0;
    for (int r = lo; r < hi; r += 16) {
        if (r + 16 <= hi) {
            const uint4 v = *(const uint4 *)(d.cert + r);
            // bytes are 0 or 1: the unproven ones are the zero bytes
            cnt += 16 - (__popc(v.x & 0x01010101u) + __popc(v.y & 0x01010101u) + __popc(v.z & 0x01010101u) + __popc(v.w & 0x01010101u));
        } else {
            for (int q = r; q < hi; ++q) cnt += d.cert[q] ? 0 : 1;
        }
    }
    int incl = cnt;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(incl, o); if (lane >= o) incl += t; }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int off = 0;
    for (int k = 0; k < w; ++k) off += wsum[k];
    int rank = off + incl - cnt;                                    // unproven visits in front of this thread's run
    for (int b = threadIdx.x; b < d.nslots + 2; b += 1024) d.bucket_bins[b] = 0;      // (left at zero for the next sort)
    // every label's account for this stretch: nothing used, the counts the robust tables were built for
    const int K = c->job.K;
    for (int a = threadIdx.x; a < K; a += 1024) {
        const double dn_tab = d.rtab[(long long)a * 8 + 7];
        SafeCol e;
        e.w = 0.0f; e.wm = 0.0f; e.pad = 0;
        e.hi = (short)(int)fabs(dn_tab);
        e.lo = dn_tab < 0.0 ? (short)32767 : e.hi;
        d.ep_state[d.perm[a]] = e;
    }
    int total = 0;
    for (int k = 0; k < 16; ++k) total += wsum[k];
    if (threadIdx.x == 0) {
        c->gl_total = total < cap ? total : cap;
        if (total <= cap) c->gl_stretch_end = base + nrows;
        c->gl_off = 0;
        c->safe_epoch_pos0 = base;
        c->safe_scanned += nrows;
        c->n_resid = 0;                 // (the residual list of this proof pass has been worked through)
        // A dense proof pass builds neither the pruning tables nor ftabR, and leaves pr_const to whoever wrote it last:
        // it must not claim them for this epoch -- a table pass at an unchanged state_epoch (the host switches kinds between
        // batches) would skip safe_rtab / safe_ftab and prove with the bounds of an older state.
        if (!d.safe_dense) c->tables_valid = 1;
        c->safe_epoch_built = d.safe_dense ? -1 : c->state_epoch;
        c->safe_cap_built = safe_cap_now(d, c);         // (the budget the resolver enforces: kernels_gram.hip)
    }
    if (cnt > 0 && rank <= cap) {
        for (int r = lo; r < hi && rank <= cap; ++r) {
            if (d.cert[r]) continue;
            if (rank < cap) d.glist[rank] = base + r;
            else c->gl_stretch_end = base + r;                        // the first visit the list has no room for
            ++rank;
        }
    }
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
        c->safe_epoch_valid = 1;
        safe_next_window(d, c);
    }
}

// The residual list of the proof pass (visits home_kernel's table bound could not prove): every label the pruning
// kernel could not exclude under the budget has its exact frozen quadratic form in the block-sparse q lines.  Four
// threads per visit (the mask words dealt round robin): robust upper bounds of the kept labels from their exact c_0,
// the home from below, the excluded labels below e^-80 of max(home, new table) each.
__global__ __launch_bounds__(256) void safe_choice_kernel(Dev d) {
    Ctrl *c = d.ctrl;
    if (c->error != 0 || c->job.mode != MODE_FRESH || c->safe_epoch_valid) return;
    const long long nrows = c->n_resid;
    if (blockIdx.x == 0 && threadIdx.x == 0) {            // (what the host decides between the two kinds of proof pass on)
        c->safe_resid_sum += nrows; c->safe_sorted_sum += c->n_sorted;
    }
    if (nrows <= kSafeResidSkip) return;                  // (the pruning kernel stood aside: these visits stay unproven)
    const long long k = ((long long)blockIdx.x * 256 + threadIdx.x) >> 2;
    const int part = threadIdx.x & 3;
    if (k >= nrows) return;                               // (whole quads leave together)
    const int K = c->job.K;
    const int wrow = d.wpermR[k];
    const WRec rec = d.wrecR[k];
    const long long b = k >> 4;
    const unsigned long long *__restrict__ mask = d.keep64 + b * d.keep_stride;
    const double *__restrict__ qline = d.q + (b * (long long)d.nslots) * 16 + (k & 15);
    const double *__restrict__ gg = d.rtab + (long long)(d.nslots - 1) * 8;
    const double emcap = gg[0], ecap = gg[1];
    const int a = rec.home_label;
    bool ok = false;
    double lb = 0.0, R = 0.0;
    if (rec.home >= 0 && a >= 0 && a < K && ((mask[a >> 6] >> (a & 63)) & 1ull)) {
        const double *__restrict__ rh = d.rtab + (long long)a * 8;
        const double chi = (qline[(long long)a * 16] + rh[3]) * ecap;
        if (rh[6] > 0.5 && chi < 1.0) {
            ok = true;
            lb = rh[4] + rh[5] * log(1.0 - chi);
            const int nw = (K + 63) >> 6;
            for (int wi = part; wi < nw; wi += 4) {
                unsigned long long m = mask[wi];
                if (wi == nw - 1 && (K & 63)) m &= (1ull << (K & 63)) - 1ull;
                while (m) {
                    const int t = wi * 64 + __ffsll((long long)m) - 1;
                    m &= m - 1;
                    if (t == a) continue;
                    const double *__restrict__ rt = d.rtab + (long long)t * 8;
                    const double qv = qline[(long long)t * 16];
                    const double clb = ((qv > 0.0 ? qv * (1.0 - 1e-9) : 0.0) + rt[3]) * emcap;
                    R += exp_above(rt[0] - rt[1] * log1p_below(clb) - lb);
                }
            }
        }
    }
    R += __shfl_xor(R, 1);
    R += __shfl_xor(R, 2);
    if (part != 0 || !ok) return;
    const double mstar = fmax(lb, rec.mlb0);
    R += exp(rec.mlb0 - lb) + (double)K * exp(mstar - 80.0 - lb);
    R *= 1.0 + 1e-6;
    const double u = d.u[c->job.win_base + wrow];
    if (R < 0.25 && u >= R + 1e-12 && u <= 1.0 - R - 1e-12) d.cert[wrow] = 1;
}

// The DENSE proof pass (Dev::safe_dense): the plain likelihood kernel has left the exact frozen quadratic form of every
// (visit, label) pair of the stretch in q[slot][row]; four threads per visit sum the robust upper bounds of ALL other labels
// against the home's robust lower bound -- the same bounds as above, with nothing excluded and nothing tabulated (sixteen threads per visit).  Where the
// clusters overlap the per-home tables prove nothing and the pruning kernel keeps every pair: then this is the same arithmetic
// without the bucket sort, the three table kernels, the home pass and the work lists (3 launches instead of 11 per stretch).
__device__ __forceinline__ void safe_dense_choice_body(const Dev &d) {
    Ctrl *c = d.ctrl;
    if (c->error != 0 || c->job.mode != MODE_FRESH || c->safe_epoch_valid) return;
    const long long base = c->job.win_base, nrows = c->job.win_hi - base;
    const long long r = ((long long)blockIdx.x * 256 + threadIdx.x) >> 4;      // sixteen threads per visit
    const int part = threadIdx.x & 15;
    if (r >= nrows) return;                               // (whole groups leave together)
    const int K = c->job.K;
    const long long p = base + r;
    const long long i = d.order ? d.order[p] : p;
    // (with the look-ahead the forms of visit p sit at ring row p & (2 C - 1), whoever scored them)
    const long long rq = d.ahead_C > 0 ? (p & (2ll * d.ahead_C - 1)) : r;
    const int h = d.z[i];
    const int a = h >= 0 ? d.label_of_slot[h] : -1;
    const double *__restrict__ gg = d.rtab + (long long)(d.nslots - 1) * 8;
    const double emcap = gg[0], ecap = gg[1];
    bool ok = false;
    double lb = 0.0, R = 0.0;
    if (a >= 0 && a < K) {
        const double *__restrict__ rh = d.rtab + (long long)a * 8;
        const double qh = d.q[(long long)h * d.qstride + rq];
        const double chi = (qh + rh[3]) * ecap;
        if (rh[6] > 0.5 && qh >= 0.0 && chi < 1.0) {
            ok = true;
            lb = rh[4] + rh[5] * log(1.0 - chi);
            // (four labels' loads in flight together: slot, constants, quadratic form)
            for (int t0 = part; t0 < K; t0 += 64) {
                int sl[4];
                double qv[4], r0[4], r1[4], r3[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { const int t = t0 + 16 * k; sl[k] = t < K ? d.perm[t] : 0; }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int t = t0 + 16 * k;
                    const double *__restrict__ rt = d.rtab + (long long)(t < K ? t : 0) * 8;
                    r0[k] = rt[0]; r1[k] = rt[1]; r3[k] = rt[3];
                    qv[k] = d.q[(long long)sl[k] * d.qstride + rq];
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int t = t0 + 16 * k;
                    if (t >= K || t == a) continue;
                    const double clb = ((qv[k] > 0.0 ? qv[k] * (1.0 - 1e-9) : 0.0) + r3[k]) * emcap;
                    R += exp_above(r0[k] - r1[k] * log1p_below(clb) - lb);
                }
            }
        }
    }
    R += __shfl_xor(R, 1);
    R += __shfl_xor(R, 2);
    R += __shfl_xor(R, 4);
    R += __shfl_xor(R, 8);
    if (part != 0) return;
    bool safe = false;
    if (ok) {
        R += exp(d.log_alpha + d.log_prior[i] - lb);       // the new table, exactly (igmm/crpmm.py:74)
        R *= 1.0 + 1e-6;
        const double u = d.u[p];
        safe = R < 0.25 && u >= R + 1e-12 && u <= 1.0 - R - 1e-12;
    }
    d.cert[r] = safe ? 1 : 0;
}

// ---- the look-ahead of the dense proof pass -------------------------------------------------------------------------------
// A stretch's proof pass used to put every (visit, label) pair of the stretch through the likelihood kernel on the chain's
// own stream: ~1 000 visits x 200 labels, a launch too small for the chip (106 us at C4's shape) in front of a resolver that
// then walks a few dozen of those visits on ONE compute unit.  Nothing in those forms depends on the stretch: they depend
// on the visits (fixed for the sweep) and on the components' factors -- and a window changes the factors of the handful of
// labels it moves points between, nobody else's (gaussian_components.py:154-205: add_item / del_item touch one component).
// So the forms live in a RING of two chunks of ahead_C visits (row = visit & (2 C - 1)), kept current from a second stream:
//   * a chunk the chain is about to enter is scored there in one launch, every slot, beside the resolver;
//   * after every window the labels it touched (Dev::touch_seq: a dozen) are re-scored there for the ring's rows ahead of
//     the chain -- again beside the next resolver;
//   * on the chain's own stream a stretch re-scores its rows only for the labels of the window that closed last (whose
//     re-scoring on the second stream ran while that window was still being walked: it cannot have seen its outcome).
// What the verdict kernel reads is in every case an exact form under the state the stretch starts from: untouched since it
// was made, or re-made after the touching window's finish kernel.
//
// safe_plan_kernel (one workgroup, main stream, behind safe_rtab_kernel) is the only writer of the ring's state:
//   1. books what the second stream has done since the last step (the host made this stream wait for it): the ring is
//      valid up to the windows closed when that work was planned (Ctrl::ah_seq[0]);
//   2. lists the labels touched since (resc_list) and describes what this step's stretch re-scores for them (resc_job;
//      the whole stretch, every label, if the ring does not hold it; nothing while the stretch's proofs stand);
//   3. describes the second stream's next work: the same labels over the ring's rows ahead of the stretch (mt_job per half),
//      and a chunk to score in full (ah_job) -- the one the chain is in, from the end of the stretch at hand, if the ring
//      does not hold it; else the next one once half of this one is behind the chain.
__device__ __forceinline__ void safe_plan_body(const Dev &d) {
    __shared__ int n_dirty_s, hw_s;
    Ctrl *c = d.ctrl;
    const int tid = threadIdx.x;
    const long long C = d.ahead_C;
    if (tid == 0) { n_dirty_s = 0; hw_s = 0; }
    __syncthreads();
    if (c->error != 0 || c->job.mode == MODE_DONE) {
        if (tid == 0) { d.ah_job[0].mode = d.ah_job[1].mode = d.ah_job[2].mode = MODE_DONE; d.resc_job->mode = MODE_DONE; c->ah_req_chunk = -1; }
        return;
    }
    // (uniform reads of the state as the last step left it; thread 0 writes it back at the end)
    long long ch[2] = {c->ah_chunk[0], c->ah_chunk[1]}, lo[2] = {c->ah_lo[0], c->ah_lo[1]};
    long long ring_seq = c->ah_seq[0];
    const bool served = c->ah_seq[1] != 0;                          // (a plan was handed to the second stream last step)
    if (served) ring_seq = c->ah_req_seq;
    // (lazy: nobody keeps the ring current -- a half is as good as the state its chunk was scored under, and a stretch re-scores
    //  every label touched since.  Several chains side by side: re-scoring the touched labels over ALL the ring's rows ahead
    //  after every window is free beside one chain's resolver, and three quarters of the device's time beside eight)
    long long hs[2] = {c->ah_hseq[0], c->ah_hseq[1]};
    if (c->ah_req_chunk >= 0) {
        ch[c->ah_req_chunk & 1] = c->ah_req_chunk; lo[c->ah_req_chunk & 1] = c->ah_req_lo;
        hs[c->ah_req_chunk & 1] = c->ah_req_seq;
    }
    const Job &j = c->job;
    const int K = j.K;
    const long long a = j.win_base, b = j.win_hi;
    const bool proving = !c->safe_epoch_valid && j.mode == MODE_FRESH;
    const long long cur = (proving ? a : j.pos) / C;
    const int hc = (int)(cur & 1), hn = hc ^ 1;
    const bool covered = proving && ch[hc] == cur && a >= lo[hc];
    // the labels touched since the ring was last brought up to date; the highest slot in use (a chunk is scored for slots
    // 0 .. hw - 1)
    int hw = 0;
    for (int t = tid; t < K; t += 256) {
        const int s = d.perm[t];
        hw = s + 1 > hw ? s + 1 : hw;
        if (d.touch_seq[s] > (d.ahead_lazy ? hs[hc] : ring_seq)) d.resc_list[atomicAdd(&n_dirty_s, 1)] = s;
    }
    atomicMax(&hw_s, hw);
    __syncthreads();
    if (tid != 0) return;
    const int nd = n_dirty_s;
    auto job_of = [&](int mode, long long p0, long long p1, int nlist) {
        Job r;
        r.pos = p0; r.win_base = p0 & ~(2 * C - 1); r.win_hi = p1; r.K = nlist; r.n_dirty = nlist; r.dirty[0] = r.dirty[1] = 0;
        r.chunks = 1; r.prune = 0;
        r.mode = (p0 < p1 && nlist > 0) ? mode : MODE_DONE;
        return r;
    };
    // 2. this step's stretch
    if (!proving) *d.resc_job = job_of(MODE_DONE, 0, 0, 0);
    else if (!covered) { *d.resc_job = job_of(MODE_FRESH, a, b, K); c->ah_self += 1; }
    else { *d.resc_job = job_of(MODE_LIST, a, b, nd); c->ah_served += 1; c->ah_dirty += nd; }
    // 3. the second stream: the touched labels over the ring's rows ahead ...
    const long long from = proving ? b : j.pos;                // (what this step's stretch scores itself is not asked for)
    const long long end_c = (cur + 1) * C < c->n_visits ? (cur + 1) * C : c->n_visits;
    const long long end_n = (cur + 2) * C < c->n_visits ? (cur + 2) * C : c->n_visits;
    {
        const long long p0 = from > lo[hc] ? from : lo[hc];
        d.ah_job[1] = (ch[hc] == cur && !d.ahead_lazy) ? job_of(MODE_LIST, p0, end_c, nd) : job_of(MODE_DONE, 0, 0, 0);
        d.ah_job[2] = (ch[hn] == cur + 1 && !d.ahead_lazy) ? job_of(MODE_LIST, lo[hn] > (cur + 1) * C ? lo[hn] : (cur + 1) * C, end_n, nd)
                                                            : job_of(MODE_DONE, 0, 0, 0);
    }
    // ... and a chunk in full
    long long want = -1, want_lo = 0, want_hi = 0;
    if (ch[hc] != cur && from < end_c) { want = cur; want_lo = from; want_hi = end_c; }
    else if (ch[hn] != cur + 1 && (cur + 1) * C < c->n_visits && 2 * (end_c - from) <= C) { want = cur + 1; want_lo = (cur + 1) * C; want_hi = end_n; }
    d.ah_job[0] = want >= 0 ? job_of(MODE_SLOTS, want_lo, want_hi, hw_s) : job_of(MODE_DONE, 0, 0, 0);
    if (d.ah_job[0].mode == MODE_DONE) want = -1;
    else c->ah_chunks += 1;
    c->ah_chunk[0] = ch[0]; c->ah_chunk[1] = ch[1]; c->ah_lo[0] = lo[0]; c->ah_lo[1] = lo[1];
    c->ah_seq[0] = ring_seq; c->ah_seq[1] = 1;
    c->ah_hseq[0] = hs[0]; c->ah_hseq[1] = hs[1];
    c->ah_req_chunk = want; c->ah_req_seq = c->win_seq; c->ah_req_lo = want_lo;
}

__global__ __launch_bounds__(256) void safe_plan_kernel(Dev d) { safe_plan_body(d); }
__global__ __launch_bounds__(256) void safe_plan_group_kernel(const Dev *__restrict__ group) { const Dev d = group[blockIdx.y]; safe_plan_body(d); }
__global__ __launch_bounds__(256) void safe_open_kernel(Dev d) { safe_open_body(d); }
__global__ __launch_bounds__(64) void safe_rtab_kernel(Dev d) { safe_rtab_body(d); }
__global__ __launch_bounds__(1024) void safe_compact_kernel(Dev d) { safe_compact_body(d); }
__global__ __launch_bounds__(256) void safe_dense_choice_kernel(Dev d) { safe_dense_choice_body(d); }
// (several chains in one launch: workgroup (x, c) works for chain group[c] -- the shared safe-stay steps of api_group.hip)
__global__ __launch_bounds__(256) void safe_open_group_kernel(const Dev *__restrict__ group) { const Dev d = group[blockIdx.y]; safe_open_body(d); }
__global__ __launch_bounds__(64) void safe_rtab_group_kernel(const Dev *__restrict__ group) { const Dev d = group[blockIdx.y]; safe_rtab_body(d); }
__global__ __launch_bounds__(1024) void safe_compact_group_kernel(const Dev *__restrict__ group) { const Dev d = group[blockIdx.y]; safe_compact_body(d); }
__global__ __launch_bounds__(256) void safe_dense_choice_group_kernel(const Dev *__restrict__ group) {
    const Dev d = group[blockIdx.y];
    safe_dense_choice_body(d);
}

void launch_safe_open_group(const Dev *group, int G, hipStream_t st) {
    hipLaunchKernelGGL(safe_open_group_kernel, dim3(1, G), dim3(256), 0, st, group);
}
// One safe-stay step of G chains of one shape whose proof pass is the dense one, without the look-ahead (launch_safe_step's
// second branch), every kernel ONE launch for all of them.  max_rows / max_nslots: the largest stretch grid / slot count.
// ah (optional; every chain's view then has the same ahead_C > 0): the look-ahead's second stream and events, as in
// launch_safe_step's first branch -- plan, the second stream's three jobs beside the touched labels' re-scoring.
bool launch_safe_group_step(const Dev &lead, const Dev *group, int G, int reach, int resolve_lds, long long max_rows, int max_nslots,
                            hipStream_t st, const SafeAhead *ah) {
    hipLaunchKernelGGL(safe_rtab_group_kernel, dim3(max_nslots, G), dim3(64), 0, st, group);
    if (ah && lead.ahead_C > 0) {
        hipLaunchKernelGGL(safe_plan_group_kernel, dim3(1, G), dim3(256), 0, st, group);
        if (hipEventRecord(ah->ev_plan, st) != hipSuccess || hipStreamWaitEvent(ah->stream, ah->ev_plan, 0) != hipSuccess) return false;
        // (a chunk, every slot + the touched labels over this half of the ring + over the other: one launch, grid.z = the job;
        //  lazy: only the chunk)
        if (!launch_score_proof_group(lead, group, G, lead.ahead_C, lead.ahead_lazy ? 0 : 4, ah->stream)) return false;
        if (hipEventRecord(ah->ev_done, ah->stream) != hipSuccess) return false;
        launch_score_proof_group(lead, group, G, max_rows, 3, st);
        hipLaunchKernelGGL(safe_dense_choice_group_kernel, dim3((unsigned)((max_rows + 15) / 16), G), dim3(256), 0, st, group);
        hipLaunchKernelGGL(safe_compact_group_kernel, dim3(1, G), dim3(1024), 0, st, group);
        return launch_gram_group_step(lead, group, G, reach, resolve_lds, st);
    }
    if (!launch_score_proof_group(lead, group, G, max_rows, -1, st)) return false;
    hipLaunchKernelGGL(safe_dense_choice_group_kernel, dim3((unsigned)((max_rows + 15) / 16), G), dim3(256), 0, st, group);
    hipLaunchKernelGGL(safe_compact_group_kernel, dim3(1, G), dim3(1024), 0, st, group);
    return launch_gram_group_step(lead, group, G, reach, resolve_lds, st);
}

void launch_safe_open(const Dev &d, hipStream_t st) {
    hipLaunchKernelGGL(safe_open_kernel, dim3(1), dim3(256), 0, st, d);
}

// One safe-stay step.  The Dev of a safe batch has prune_enabled = 2, use_certify = 0, use_home = 1, safe_mode = 1.
bool launch_safe_step(const Dev &d, int resolve_lds, long long max_rows, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1,
                      const SafeAhead *ah) {
    if (d.safe_dense && ah && d.ahead_C > 0) {
        // with the look-ahead: plan -> (second stream: the next chunk, every slot) || re-score the dirty labels -> verdicts
        hipLaunchKernelGGL(safe_rtab_kernel, dim3(d.nslots), dim3(64), 0, st, d);
        hipLaunchKernelGGL(safe_plan_kernel, dim3(1), dim3(256), 0, st, d);
        if (hipEventRecord(ah->ev_plan, st) != hipSuccess || hipStreamWaitEvent(ah->stream, ah->ev_plan, 0) != hipSuccess) return false;
        Dev dl = d;
        dl.slot_list = d.resc_list;
        launch_score(d, KERNEL_MFMA, d.ah_job, d.q, d.qstride, -1, d.ahead_C, 3, ah->stream);           // a chunk, every slot
        launch_score(dl, KERNEL_MFMA, d.ah_job + 1, d.q, d.qstride, -1, d.ahead_C, 3, ah->stream);      // the touched labels, this half
        launch_score(dl, KERNEL_MFMA, d.ah_job + 2, d.q, d.qstride, -1, d.ahead_C, 3, ah->stream);      // ... the other half
        if (hipEventRecord(ah->ev_done, ah->stream) != hipSuccess) return false;
        launch_score(dl, KERNEL_MFMA, d.resc_job, d.q, d.qstride, -1, max_rows, 3, st);
        hipLaunchKernelGGL(safe_dense_choice_kernel, dim3((unsigned)((max_rows + 15) / 16)), dim3(256), 0, st, d);
        hipLaunchKernelGGL(safe_compact_kernel, dim3(1), dim3(1024), 0, st, d);
        return launch_gram_core(d, resolve_lds, st, ev0, ev1);
    }
    if (d.safe_dense) {
        hipLaunchKernelGGL(safe_rtab_kernel, dim3(d.nslots), dim3(64), 0, st, d);
        launch_score(d, KERNEL_MFMA, &d.ctrl->job, d.q, d.qstride, -1, max_rows, 2, st);   // every pair of the stretch, exactly
        hipLaunchKernelGGL(safe_dense_choice_kernel, dim3((unsigned)((max_rows + 15) / 16)), dim3(256), 0, st, d);
        hipLaunchKernelGGL(safe_compact_kernel, dim3(1), dim3(1024), 0, st, d);
        return launch_gram_core(d, resolve_lds, st, ev0, ev1);
    }
    launch_prune_tables(d, st);                                          // centre distances, radii (exit while valid)
    hipLaunchKernelGGL(safe_rtab_kernel, dim3(d.nslots), dim3(64), 0, st, d);
    hipLaunchKernelGGL(safe_ftab_kernel, dim3(d.nslots), dim3(256), d.nslots * 5 * (int)sizeof(double), st, d);
    launch_bucket_rows(d, max_rows, st);                                  // the stretch's visits grouped by home
    launch_home(d, max_rows, st);                                         // proof pass: cert[row] = SAFE
    launch_score_pruned(d, &d.ctrl->job, d.q, d.qstride, max_rows, st);   // what its table bound left open: exact forms
    hipLaunchKernelGGL(safe_choice_kernel, dim3((unsigned)((max_rows + 63) / 64)), dim3(256), 0, st, d);
    hipLaunchKernelGGL(safe_compact_kernel, dim3(1), dim3(1024), 0, st, d);
    return launch_gram_core(d, resolve_lds, st, ev0, ev1);
}

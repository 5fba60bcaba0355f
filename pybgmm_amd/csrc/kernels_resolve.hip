// In-launch mover resolver: ONE workgroup consumes every mover of a short sub-window.
//
// In the mover-dense regime (burn-in, overlapping clusters) the speculative window degenerates
// into the reference's sequential loop (igmm/crpmm.py:57-88): nearly every visit changes the
// state.  Paying kernel boundaries per mover (score -> choice -> apply -> refresh) then costs
// ~70 us per visit.  This kernel instead takes the R visits that start at the first mover of
// the window and walks them in order, entirely on one CU:
//   A  bookkeeping of the move (labels / slots / counts; the reference's del_item + add_item,
//      gaussian_components.py:154-205)
//   B  the two touched slots, one per half of the block: m, S (separately rounded), then the
//      rank-1 change of the inverse factor (or the from-scratch route every kRefreshEvery
//      steps) with the factor held in LDS, outputs written through to HBM
//   C  quadratic forms of the remaining rows against the two fresh factors, straight from LDS
//   D  the remaining rows' categorical draws from an LDS tile e[col][row] = exp(lp - M_row):
//      only the two dirty columns need new log/exp evaluations; totals and the
//      sequential-subtract scan (utils/utils.py:15-20) are re-done from the tile
// and then opens a fresh window behind the sub-window.  A step it has consumed is skipped by
// apply_kernel / refresh_ctrl_kernel.  It declines (and the per-mover kernels run) when
// movers are sparse, when D > 64, or when the tile would not fit in LDS.
#include "bgmm_device.h"
#include "slot_math.h"

typedef double v4d __attribute__((ext_vector_type(4)));

#define RT 512              // threads of the resolver block (8 waves: 256 VGPRs each)
#define HT 256              // threads per half
#define NPART (RT / 128)    // j-interleaves per slot in the re-scoring phase

struct ResolveShared {
    int active, nrows, K, ncols, err, pad0;
    int slot[2], kind[2], src[2], col[2], bad[2];
    double a[2], logdet_src[2], stot[2], inv_lam_src[2], lam_new[2];
    long long sub_lo, i_move, lik, moves;
    double scnew[2][12];    // fresh SlotConst of the two touched slots (D1 reads them from LDS)
};

// LDS arrays are reached through address-space-3 pointers: generic pointers stored in a struct
// make the compiler fall back to flat_load / flat_store (and 64-bit scalar pairs) for every access.
#define LDS_AS __attribute__((address_space(3)))
typedef LDS_AS double *lds_f64;
typedef LDS_AS long long *lds_i64;
typedef LDS_AS int *lds_i32;

struct ResolveLds {
    LDS_AS ResolveShared *S;
    lds_f64 W, vec, xs, et, qpart, rowM, rowEn, rowu, ldetL;
    lds_i64 rowi;
    lds_i32 rowq, rowhome, rowpick, rowstay, permL, loc, nL, nupdL;
};

enum { O_W, O_VEC, O_XS, O_ET, O_QP, O_RM, O_RE, O_RU, O_RI, O_RQ, O_RH, O_RP, O_RS, O_PL, O_LC, O_NL, O_NU,
       O_LD, O_COUNT };

// byte offsets of the arrays inside the dynamic LDS block; returns the total size
__host__ __device__ inline size_t resolve_offsets(int D, int R, int Kcap, int nslots, unsigned *o) {
    size_t off = 768;                       // ResolveShared
    const int ld = D + 1;
    unsigned dummy[O_COUNT];
    if (!o) o = dummy;
    auto take = [&](int which, size_t bytes) { o[which] = (unsigned)off; off += (bytes + 15) & ~(size_t)15; };
    take(O_W, sizeof(double) * 2 * D * ld);
    take(O_VEC, sizeof(double) * 21 * D);       // 2 x 6 vectors, the mover's x, 2 x 4 scan partials
    take(O_XS, sizeof(double) * D * R);
    take(O_ET, sizeof(double) * (size_t)Kcap * R);
    take(O_QP, sizeof(double) * (RT / 64) * R);
    take(O_RM, sizeof(double) * R);
    take(O_RE, sizeof(double) * R);
    take(O_RU, sizeof(double) * R);
    take(O_RI, sizeof(long long) * R);
    take(O_RQ, sizeof(int) * R);
    take(O_RH, sizeof(int) * R);
    take(O_RP, sizeof(int) * R);
    take(O_RS, sizeof(int) * R);
    take(O_PL, sizeof(int) * Kcap);
    take(O_LC, sizeof(int) * nslots);
    take(O_NL, sizeof(int) * nslots);
    take(O_NU, sizeof(int) * nslots);
    take(O_LD, sizeof(double) * nslots);
    return off;
}

__device__ __forceinline__ void resolve_carve(LDS_AS unsigned char *base, int D, int R, int Kcap, int nslots,
                                              ResolveLds &L) {
    unsigned o[O_COUNT];
    resolve_offsets(D, R, Kcap, nslots, o);
    L.S = (LDS_AS ResolveShared *)base;
    L.W = (lds_f64)(base + o[O_W]); L.vec = (lds_f64)(base + o[O_VEC]); L.xs = (lds_f64)(base + o[O_XS]);
    L.et = (lds_f64)(base + o[O_ET]); L.qpart = (lds_f64)(base + o[O_QP]); L.rowM = (lds_f64)(base + o[O_RM]);
    L.rowEn = (lds_f64)(base + o[O_RE]); L.rowu = (lds_f64)(base + o[O_RU]); L.ldetL = (lds_f64)(base + o[O_LD]);
    L.rowi = (lds_i64)(base + o[O_RI]);
    L.rowq = (lds_i32)(base + o[O_RQ]); L.rowhome = (lds_i32)(base + o[O_RH]); L.rowpick = (lds_i32)(base + o[O_RP]);
    L.rowstay = (lds_i32)(base + o[O_RS]); L.permL = (lds_i32)(base + o[O_PL]); L.loc = (lds_i32)(base + o[O_LC]);
    L.nL = (lds_i32)(base + o[O_NL]); L.nupdL = (lds_i32)(base + o[O_NU]);
}

// Barrier for phases that hand data over through LDS only: waits for this wave's LDS traffic, not
// for its in-flight global stores (a plain __syncthreads() drains vmcnt(0), i.e. a global round
// trip per phase).  Every mover iteration ends with one full __syncthreads(), which is what
// orders the global writes of an iteration (m, S, Wrm, mu, q ...) before the next one's reads.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wscan(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

// (Re)build row r of the tile from the q buffer: log scores -> M_r -> exp(lp - M_r).  Wave level.
__device__ void build_row(const Dev &d, const ResolveLds &L, int R, int r, int lane, long long qstride) {
    const int K = L.S->K;
    const long long i = L.rowi[r];
    const int h = L.rowhome[r];
    const int nh = h >= 0 ? L.nL[h] : 0;
    const double lp_new = d.log_alpha + d.log_prior[i];
    double mx = lp_new;
    for (int j = lane; j < K; j += 64) {
        const int s = L.permL[j];
        const double qv = d.q[(long long)s * qstride + L.rowq[r]];
        const SlotConst sc = d.sc[s];
        const bool own = s == h;
        const double lp = slot_log_score(sc, qv, own && nh >= 2);
        L.et[L.loc[s] * R + r] = lp;
        if (!(own && nh == 1)) mx = fmax(mx, lp);       // a singleton home is not a candidate
    }
    mx = wmax(mx);
    for (int j = lane; j < K; j += 64) {
        const int idx = L.loc[L.permL[j]] * R + r;
        L.et[idx] = exp(L.et[idx] - mx);
    }
    if (lane == 0) {
        L.rowM[r] = mx;
        L.rowEn[r] = exp(lp_new - mx);
    }
}

// Draw of row r from the tile (reference: crpmm.py:75-78 + utils.py:15-20).  Wave level.
__device__ void pick_row(const Dev &d, const ResolveLds &L, int R, int r, int lane, long long qstride) {
    const int K = L.S->K;
    const int h = L.rowhome[r];
    const int nh = h >= 0 ? L.nL[h] : 0;
    const bool home_live = h >= 0 && nh >= 2;
    const bool singleton = h >= 0 && nh == 1;
    const int lab_h = singleton ? d.label_of_slot[h] : -1;
    const int Lr = singleton ? K - 1 : K;
    double tot = 0.0;
    for (int attempt = 0; attempt < 2; ++attempt) {
        tot = 0.0;
        for (int j = lane; j <= Lr; j += 64) {
            const int jj = (singleton && j == lab_h) ? K - 1 : j;
            tot += j == Lr ? L.rowEn[r] : L.et[L.loc[L.permL[jj]] * R + r];
        }
        tot = wsum(tot);
        if (tot > 1e-200 && tot < 1e200) break;
        build_row(d, L, R, r, lane, qstride);           // reference point M_r went stale: rebuild
    }
    const double u = L.rowu[r];
    double carry = 0.0;
    int pick = Lr;
    for (int j0 = 0; j0 <= Lr; j0 += 64) {
        const int j = j0 + lane;
        double pj = 0.0;
        if (j <= Lr) {
            const int jj = (singleton && j == lab_h) ? K - 1 : j;
            pj = (j == Lr ? L.rowEn[r] : L.et[L.loc[L.permL[jj]] * R + r]) / tot;
        }
        const double cum = carry + wscan(pj, lane);
        const bool hit = j <= Lr && (u - cum) < 0.0;
        const unsigned long long m = __ballot(hit);
        if (m) { pick = j0 + __ffsll((long long)m) - 1; break; }
        carry = __shfl(cum, 63);
    }
    if (lane == 0) {
        L.rowpick[r] = pick;
        L.rowstay[r] = (home_live && pick < Lr && L.permL[pick] == h) ? 1 : 0;
    }
}

// Team-local outputs of one slot (HT threads): cvec (4 threads per row), Wrm, Wfrag, mu, consts.
__device__ __forceinline__ void write_slot_team(const Dev &d, int s, const lds_f64 W, int ld, const lds_f64 mu,
                                                double logdetC, double lam, int n_new, int ht, lds_f64 cv_lds,
                                                lds_f64 sc_lds) {
    const int D = d.D, Dp = d.Dp;
    // the scalar constants first: their table loads are in flight while the team does the rest
    SlotConst sc_new;
    if (ht < 64) {
        double m2 = 0.0;
        for (int l = ht; l < D; l += 64) m2 = fma(mu[l], mu[l], m2);
        m2 = wsum(m2);
        if (ht == 0) sc_new = make_consts(d, n_new, logdetC, lam, m2);
    }
    for (int r0 = 0; r0 < Dp; r0 += HT / 4) {
        const int r = r0 + (ht >> 2), part = ht & 3;
        double acc = 0.0;
        if (r < D) {
            const lds_f64 Wr = W + r * ld;
            for (int l0 = part; l0 <= r; l0 += 32) {
                double wv[8], mv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int l = l0 + 4 * k;
                    wv[k] = l <= r ? Wr[l] : 0.0;
                    mv[k] = l <= r ? mu[l] : 0.0;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) acc = fma(wv[k], mv[k], acc);
            }
        }
        acc += __shfl_xor(acc, 1);
        acc += __shfl_xor(acc, 2);
        if (part == 0 && r < Dp) {
            d.cvec[(long long)s * Dp + r] = acc;
            if (r < D) cv_lds[r] = acc;
        }
    }
    double *Wg = d.Wrm + (long long)s * D * D;
    for (int e = ht; e < D * D; e += HT) {
        const int a = e / D, b = e - a * D;
        Wg[e] = W[a * ld + b];                       // (upper triangle of the LDS copy is zero)
    }
    double *wf = d.Wfrag + (long long)s * d.nfrag * 64;
    for (int e = ht; e < d.nfrag * 64; e += HT) {
        const int f = e >> 6, lane = e & 63;
        int J = 0;
        while (2 * (J + 1) * (J + 2) <= f) ++J;
        const int kk = f - 2 * J * (J + 1);
        const int j = 16 * J + (lane & 15), l = 4 * kk + (lane >> 4);
        wf[e] = (j < D && l <= j) ? -W[j * ld + l] : 0.0;
    }
    for (int l = ht; l < D; l += HT) d.mu[(long long)s * D + l] = mu[l];
    if (ht == 0) {
        d.sc[s] = sc_new;
        d.mu_ver[s] += 1;                     // (as slot_math.h write_slot: per-point caches against this slot are stale)
        d.ctrl->tables_valid = 0;             // a move: the pruned-window tables and the cached bucket sort are stale
        d.ctrl->state_epoch += 1;
        d.ctrl->wsort_valid = 0;
        const double *src = (const double *)&sc_new;
#pragma unroll
        for (int k = 0; k < 12; ++k) sc_lds[k] = src[k];
    }
}

__global__ __launch_bounds__(RT) void resolve_kernel(Dev d, int R, int Kcap) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    ResolveLds L;
    resolve_carve((LDS_AS unsigned char *)lds_raw, d.D, R, Kcap, d.nslots, L);
    LDS_AS ResolveShared &S = *L.S;
    const int D = d.D, ld = D + 1, tid = threadIdx.x;
    const int half = tid / HT, ht = tid & (HT - 1);
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    Ctrl *c = d.ctrl;
    const long long qstride = d.qstride;

    if (tid == 0) {
        S.active = 0;
        const Job &j = c->job;
        if (j.mode != MODE_DONE && c->error == 0 && c->dense_mode != 1 && c->skip_apply == 0) {
            const unsigned long long fm = c->first_mover;
            if (fm != kNoMover) {
                const long long p = (long long)fm;
                const double ema = ema_after_mover(c->ema_run, (double)(p - c->last_mover));
                const bool dense = c->dense_mode == 2 || ema < 3.0 * (double)R;
                if (dense && !job_is_pruned(d, j.mode, j.prune) && j.K + R + 2 <= Kcap) {
                    S.active = 1;
                    S.sub_lo = p;
                    long long nr = j.win_hi - p;
                    S.nrows = (int)(nr < R ? nr : R);
                    S.K = j.K;
                    S.ncols = j.K;
                    S.err = 0;
                    S.lik = (p - j.pos) * (long long)j.K;     // the stays in front of the first mover
                    S.moves = 0;
                    c->n_steps += 1;
                    c->n_score_launches += 1;
                    c->n_scored += (j.win_hi - j.pos) * (long long)(j.mode == MODE_FRESH ? j.K : j.n_dirty);
                    c->n_pairs_exact += (unsigned long long)((j.win_hi - j.pos) * (long long)(j.mode == MODE_FRESH ? j.K : j.n_dirty));
                }
            }
        }
    }
    __syncthreads();
    if (!S.active) return;
    // A visiting order that names a point twice inside this sub-window: the second visit's home is
    // whatever the first left, but the rows' homes (and their tiles) are set up once, here.  Such a
    // stretch is left to the per-mover kernels, which read the labels live.  (The reference's orders
    // are permutations; the C-ABI takes any index array.)
    if (d.order) {
        const int nr = S.nrows;
        const long long mine = tid < nr ? d.order[S.sub_lo + tid] : -1;
        if (tid == 0) S.pad0 = 0;
        if (tid < nr) L.rowi[tid] = mine;
        __syncthreads();
        bool dup = false;
        for (int r2 = 0; r2 < tid && tid < nr; ++r2) dup |= L.rowi[r2] == mine;
        if (dup) S.pad0 = 1;
        __syncthreads();
        if (S.pad0) {
            if (tid == 0) {     // (undo the step this launch has just booked: apply_kernel books it again)
                const Job &j = c->job;
                c->n_steps -= 1;
                c->n_score_launches -= 1;
                c->n_scored -= (j.win_hi - j.pos) * (long long)(j.mode == MODE_FRESH ? j.K : j.n_dirty);
                c->n_pairs_exact -= (unsigned long long)((j.win_hi - j.pos) * (long long)(j.mode == MODE_FRESH ? j.K : j.n_dirty));
            }
            return;
        }
    }
    // phase clocks (build with -DBGMM_PROFILE; every probe costs a global read-modify-write)
#ifdef BGMM_PROFILE
    long long tk = clock64(), tk2;
#define PROF(i) do { if (tid == 0) { tk2 = clock64(); c->prof[i] += tk2 - tk; tk = tk2; } } while (0)
#else
#define PROF(i) do { } while (0)
#endif

    const long long win_base = c->job.win_base;
    const long long sub_lo = S.sub_lo;
    const int nrows = S.nrows;
    for (int r = tid; r < nrows; r += RT) {
        const long long p = sub_lo + r;
        const long long i = d.order ? d.order[p] : p;
        L.rowi[r] = i;
        L.rowu[r] = d.u[p];
        L.rowhome[r] = d.z[i];
        L.rowq[r] = (int)(p - win_base);
    }
    for (int j = tid; j < S.K; j += RT) {
        const int s = d.perm[j];
        L.permL[j] = s;
        L.loc[s] = j;
        L.nL[s] = d.n[s];
        L.nupdL[s] = d.nupd[s];
        L.ldetL[s] = d.sc[s].logdetC;
    }
    if (tid == 0) L.ldetL[d.K_max] = d.sc[d.K_max].logdetC;     // the prior pseudo slot
    __syncthreads();
    const lds_f64 xs = L.xs;
    for (int e = tid; e < R * D; e += RT) {
        const int r = e / D, l = e - r * D;
        xs[l * R + r] = r < nrows ? d.X[L.rowi[r] * D + l] : 0.0;
    }
    for (int r = wave; r < nrows; r += RT / 64) build_row(d, L, R, r, lane, qstride);
    __syncthreads();
    PROF(0);

    // per-half working set: factor W[D][ld], vectors mu dv pv lv tv cv, column-scan partials
    const lds_f64 Wh = L.W + half * D * ld;
    const lds_f64 vh = L.vec + half * 6 * D;
    const lds_f64 mu = vh, dv = vh + D, pv = vh + 2 * D, lv = vh + 3 * D, tv = vh + 4 * D, cv = vh + 5 * D;
    const lds_f64 xm = L.vec + 12 * D;                       // the mover's x
    const lds_f64 part = L.vec + 13 * D + half * 4 * D;      // [4][D]

    int cur = 0;
    for (; cur < nrows; ++cur) {
        // draw of the current row, once, against the state all earlier visits left behind
        if (wave == 0) pick_row(d, L, R, cur, lane, qstride);
        lds_barrier();
        PROF(5);
        if (L.rowstay[cur]) {                       // block-uniform
            if (tid == 0) S.lik += S.K;
            continue;
        }
        // ---- A: bookkeeping of the move --------------------------------------------------
        if (tid == 0) {
            const long long i = L.rowi[cur];
            const int h = L.rowhome[cur];
            const int lab = L.rowpick[cur];
            int K = S.K;
            S.i_move = i;
            S.slot[0] = S.slot[1] = -1;
            S.kind[0] = S.kind[1] = -1;
            S.bad[0] = S.bad[1] = 0;
            if (h >= 0) {
                const int nh = L.nL[h] - 1;
                L.nL[h] = nh;
                d.n[h] = nh;
                if (nh > 0) {
                    S.slot[0] = h; S.src[0] = h; S.col[0] = L.loc[h];
                    S.kind[0] = L.nupdL[h] < kRefreshEvery ? REFRESH_SUB : REFRESH_SCRATCH;
                } else {                            // swap-with-last delete of its label
                    const int labh = d.label_of_slot[h];
                    const int last = K - 1;
                    const int s_last = L.permL[last];
                    L.permL[labh] = s_last; d.perm[labh] = s_last; d.label_of_slot[s_last] = labh;
                    L.permL[last] = h; d.perm[last] = h; d.label_of_slot[h] = last;
                    K = last;
                }
            }
            S.lik += K;
            int t = -1;
            if (lab >= K) {
                if (K >= d.K_max) {
                    S.err = -3;
                } else {
                    t = d.perm[K];
                    L.permL[K] = t;
                    d.label_of_slot[t] = K;
                    L.nL[t] = 0;
                    L.nupdL[t] = 0;
                    L.loc[t] = S.ncols;
                    S.ncols += 1;
                    K += 1;
                    S.kind[1] = REFRESH_NEW;
                    S.src[1] = d.K_max;
                }
            } else {
                t = L.permL[lab];
                S.src[1] = t;
                S.kind[1] = L.nupdL[t] < kRefreshEvery ? REFRESH_ADD : REFRESH_SCRATCH;
            }
            if (t >= 0) {
                L.nL[t] += 1;
                d.n[t] = L.nL[t];
                d.z[i] = t;
                S.slot[1] = t;
                S.col[1] = L.loc[t];
            }
            S.K = K;
            S.moves += 1;
            for (int hf = 0; hf < 2; ++hf) {
                if (S.slot[hf] < 0 || S.kind[hf] == REFRESH_SCRATCH) continue;
                const int n_new = L.nL[S.slot[hf]];
                const double kb = d.k0 + (double)(hf == 0 ? n_new + 1 : n_new - 1);
                S.a[hf] = hf == 0 ? -kb / (kb - 1.0) : kb / (kb + 1.0);
                S.logdet_src[hf] = L.ldetL[S.src[hf]];
                S.inv_lam_src[hf] = d.sc[S.src[hf]].inv_lam;
            }
            const long long p = sub_lo + cur;
            c->ema_run = ema_after_mover(c->ema_run, (double)(p - c->last_mover));
            c->last_mover = p;
        }
        if (tid >= 64 && tid < 64 + D) xm[tid - 64] = d.X[L.rowi[cur] * D + (tid - 64)];
        lds_barrier();
        PROF(1);
        if (S.err < 0) break;

        // ---- B: statistics and derived state of the two slots, one per half ---------------
        const int myslot = S.slot[half], mykind = S.kind[half];
        const bool act = myslot >= 0;
        const bool act_r1 = act && mykind != REFRESH_SCRATCH, act_sc = act && mykind == REFRESH_SCRATCH;
        double k_new = 1.0;
        if (act) {
            double *m = d.m + (long long)myslot * D;
            double *Sg = d.S + (long long)myslot * D * D;
            const bool init = mykind == REFRESH_NEW;
            const int src = S.src[half];
            const double *Sin = init ? d.prior_S : Sg;
            const double *Wsrc = d.Wrm + (long long)src * D * D;
            k_new = d.k0 + (double)L.nL[myslot];
            for (int a = ht; a < D; a += HT) {
                const double mo = init ? d.prior_m[a] : m[a];
                const double mv = half == 0 ? __dsub_rn(mo, xm[a]) : __dadd_rn(mo, xm[a]);
                m[a] = mv;
                mu[a] = mv / k_new;
                if (act_r1) dv[a] = xm[a] - d.mu[(long long)src * D + a];
            }
            // 8 elements per trip (two trips at D = 64): the global loads of a trip are all
            // issued before its first dependent instruction
            for (int e0 = ht; e0 < D * D; e0 += 8 * HT) {
                double sv[8], wv[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int e = e0 + k * HT;
                    sv[k] = e < D * D ? Sin[e] : 0.0;
                    wv[k] = (e < D * D && act_r1) ? Wsrc[e] : 0.0;
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int e = e0 + k * HT;
                    if (e < D * D) {
                        const int a = e / D, b = e - a * D;
                        const double prod = __dmul_rn(xm[a], xm[b]);
                        const double nv = half == 0 ? __dsub_rn(sv[k], prod) : __dadd_rn(sv[k], prod);
                        Sg[e] = nv;
                        Wh[a * ld + b] = act_sc ? nv : wv[k];
                    }
                }
            }
        }
        lds_barrier();
        PROF(8);
        if (act_sc)
            for (int e = ht; e < D * D; e += HT) {
                const int a = e / D, b = e - a * D;
                Wh[a * ld + b] = b <= a ? Wh[a * ld + b] - k_new * (mu[a] * mu[b]) : 0.0;
            }
        const bool any_r1 = (S.slot[0] >= 0 && S.kind[0] != REFRESH_SCRATCH) || (S.slot[1] >= 0 && S.kind[1] != REFRESH_SCRATCH);
        const bool any_sc = (S.slot[0] >= 0 && S.kind[0] == REFRESH_SCRATCH) || (S.slot[1] >= 0 && S.kind[1] == REFRESH_SCRATCH);
        if (any_r1) {
            // p = W d (8 threads per row), prefix of p^2 -> l, t  (slot_math.h), then the column
            // recurrence W'[i][j] = l_i W[i][j] + t_i sum_{k<i} p_k W[k][j] as a two-pass scan
            // over 4 row chunks per column
            if (act_r1) {
                for (int r0 = 0; r0 < D; r0 += HT / 8) {
                    const int r = r0 + (ht >> 3), pt = ht & 7;
                    double acc = 0.0;
                    if (r < D) {
                        const lds_f64 Wr = Wh + r * ld;
                        for (int l = pt; l <= r; l += 8) acc = fma(Wr[l], dv[l], acc);
                    }
                    acc += __shfl_xor(acc, 1);
                    acc += __shfl_xor(acc, 2);
                    acc += __shfl_xor(acc, 4);
                    if (r < D && pt == 0) pv[r] = acc;
                }
            }
            lds_barrier();
            PROF(9);
            if (act_r1 && ht < 64) {
                const double a = S.a[half];
                const double v = ht < D ? pv[ht] * pv[ht] : 0.0;
                double P = v;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const double t = __shfl_up(P, o);
                    if (ht >= o) P += t;
                }
                double Pm1 = __shfl_up(P, 1);
                if (ht == 0) Pm1 = 0.0;
                const double num = 1.0 + a * Pm1, den = 1.0 + a * P;
                const bool badl = ht < D && (!(den > 0.0) || !(num > 0.0));
                if (ht < D) {
                    const double l = sqrt(num / den);
                    lv[ht] = l;
                    tv[ht] = -a * pv[ht] / (den * l);
                }
                const unsigned long long anybad = __ballot(badl);
                if (ht == D - 1) { S.stot[half] = P; S.bad[half] = anybad ? 1 : 0; }
            } else if (act_r1 && ht < 128) {
                double d2 = 0.0;                                  // |x - mu_old|^2 for the Lambda bound
                for (int l = ht - 64; l < D; l += 64) d2 = fma(dv[l], dv[l], d2);
                d2 = wsum(d2);
                if (ht == 64) S.lam_new[half] = lam_after_rank1(S.inv_lam_src[half], S.a[half], d2);
            }
            lds_barrier();
            PROF(10);
            const int cj = ht & 63, cc = ht >> 6;                  // column, row chunk (HT / 64 = 4 chunks)
            const int RC = (D + 3) >> 2;
            const int rlo = cc * RC, rhi = (rlo + RC < D) ? rlo + RC : D;
            const int r0c = rlo > cj ? rlo : cj;
            double wreg[16];
            if (act_r1 && cj < D) {
#pragma unroll
                for (int k = 0; k < 16; ++k) wreg[k] = (r0c + k < rhi) ? Wh[(r0c + k) * ld + cj] : 0.0;
                double sacc = 0.0;
#pragma unroll
                for (int k = 0; k < 16; ++k)
                    if (r0c + k < rhi) sacc = fma(pv[r0c + k], wreg[k], sacc);
                part[cc * D + cj] = sacc;
            }
            lds_barrier();
            if (act_r1 && cj < D) {
                double r = 0.0;
                for (int c2 = 0; c2 < cc; ++c2) r += part[c2 * D + cj];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (r0c + k < rhi) {
                        const int row = r0c + k;
                        Wh[row * ld + cj] = fma(tv[row], r, lv[row] * wreg[k]);
                        r = fma(pv[row], wreg[k], r);
                    }
                }
            }
            lds_barrier();
            PROF(11);
        }
        if (any_sc) {
            __syncthreads();
            gershgorin_bound<HT>((double *)Wh, ld, D, (double *)pv, (double *)&S.lam_new[half], ht, act_sc);
            chol_inverse<HT>((double *)Wh, ld, D, (double *)pv, (double *)&S.stot[half], (int *)&S.bad[half], ht, act_sc);
        }
        if (act) {
            const double logdetC = act_sc ? S.stot[half] : S.logdet_src[half] + log(1.0 + S.a[half] * S.stot[half]);
            write_slot_team(d, myslot, Wh, ld, mu, logdetC, S.lam_new[half], L.nL[myslot], ht, cv, (lds_f64)&S.scnew[half][0]);
            if (ht == 0) {
                const int nu = act_sc ? 0 : L.nupdL[myslot] + 1;
                L.nupdL[myslot] = nu;
                d.nupd[myslot] = nu;
                L.ldetL[myslot] = logdetC;
                if (S.bad[half]) S.err = -4;
            }
        }
        lds_barrier();
        PROF(2);

        // ---- C: quadratic forms of the remaining rows against the two fresh factors -------
        // y = cvec - Winv x as v_mfma_f64_16x16x4 tiles straight from LDS: one unit per
        // (slot, block of 16 rows, block of 16 factor rows J); fragment conventions as in
        // kernels_score.hip.  Partial sums of y^2 per J go to qpart[(slot, J)][row].
        {
            const int nJ = d.Dp >> 4, nRB = (R + 15) >> 4;
            const int lr = lane & 15, lk = lane >> 4;
            for (int u = wave; u < 2 * nRB * nJ; u += RT / 64) {
                const int hf = u / (nRB * nJ), rem = u - hf * (nRB * nJ);
                const int rb = rem / nJ, J = rem - rb * nJ;
                if (S.slot[hf] < 0) continue;
                const lds_f64 Wc = L.W + hf * D * ld;
                const lds_f64 cvh = L.vec + hf * 6 * D + 5 * D;
                const int jrow = 16 * J + lr, row = rb * 16 + lr;
                const double cjv = jrow < D ? cvh[jrow] : 0.0;
                v4d acc = (v4d){cjv, cjv, cjv, cjv};
                const bool jok = jrow < D, rok = row < R;
                const lds_f64 Wj = Wc + (jok ? jrow : 0) * ld;
                const lds_f64 xr = xs + (rok ? row : 0);
                for (int kk = 0; kk < 4 * (J + 1); ++kk) {
                    const int l = 4 * kk + lk;
                    const bool lok = l < D;
                    const double av = (rok && lok) ? xr[l * R] : 0.0;
                    const double bv = (jok && lok) ? -Wj[l] : 0.0;       // zero beyond the diagonal
                    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    double v = acc[r] * acc[r];
                    v += __shfl_xor(v, 1);
                    v += __shfl_xor(v, 2);
                    v += __shfl_xor(v, 4);
                    v += __shfl_xor(v, 8);
                    const int orow = rb * 16 + lk + 4 * r;
                    if (lr == r && orow < R) L.qpart[(hf * nJ + J) * R + orow] = v;
                }
            }
        }
        lds_barrier();
        PROF(3);
        // ---- D: the two dirty tile columns of the remaining rows ---------------------------
        for (int e = tid; e < 2 * nrows; e += RT) {
            const int hf = e / nrows, r = e - hf * nrows;
            const int sl = S.slot[hf];
            if (sl < 0 || r <= cur) continue;
            double qv = 0.0;
            for (int J = 0; J < (d.Dp >> 4); ++J) qv += L.qpart[(hf * (d.Dp >> 4) + J) * R + r];
            d.q[(long long)sl * qstride + L.rowq[r]] = qv;
            SlotConst sc;
            {
                double *dst = (double *)&sc;
#pragma unroll
                for (int k = 0; k < 12; ++k) dst[k] = S.scnew[hf][k];
            }
            const bool own = L.rowhome[r] == sl;
            const double lp = slot_log_score(sc, qv, own && L.nL[sl] >= 2);
            L.et[S.col[hf] * R + r] = exp(lp - L.rowM[r]);
        }
        __syncthreads();        // full: drains this iteration's global stores
        PROF(4);
    }

    if (tid == 0) {
        c->prof[7] += 1;
        c->lik_evals += S.lik;
        c->n_moves += S.moves;
        if (S.moves > 0) { c->tables_valid = 0; c->wsort_valid = 0; c->state_epoch += 1; }
        c->first_mover = kNoMover;
        c->n_refresh = 0;
        c->skip_apply = 1;
        c->job.K = S.K;
        if (S.err < 0) {
            atomicCAS(&c->error, 0, S.err);
            c->job.mode = MODE_DONE;
        } else {
            c->win_size = (int)window_for_rate(c);
            start_window(d, c, sub_lo + nrows);
        }
    }
}

// Largest sub-window (rows) whose working set fits in LDS for the current number of labels.
bool resolve_plan(const Dev &d, int K_now, int *R_out, int *Kcap_out, int *lds_out) {
    if (d.D > 64 || d.cov_type != COV_FULL) return false;
    for (int R = 64; R >= 8; R >>= 1) {
        const int Kcap = (K_now + 2 * R + 8 + 7) & ~7;     // room for labels opened by the caller's chunk
        const size_t bytes = resolve_offsets(d.D, R, Kcap, d.nslots, nullptr);
        if (bytes <= 156 * 1024) {
            *R_out = R; *Kcap_out = Kcap; *lds_out = (int)bytes;
            return true;
        }
    }
    return false;
}

void launch_resolve(const Dev &d, int R, int Kcap, int lds, hipStream_t st) {
    static PerDeviceLds attr;
    attr.ensure((const void *)resolve_kernel, 160 * 1024);
    (void)lds;
    hipLaunchKernelGGL(resolve_kernel, dim3(1), dim3(RT), lds, st, d, R, Kcap);
}

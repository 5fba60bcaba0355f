// Draw kernel: for every visit of the window turn the quadratic forms into the reference's
// categorical draw.  One wavefront per visit.
//
// Reference steps restated (file:line in the reference checkout):
//   seating prior  log n_k  / log(n_k ** r) ........ igmm/crpmm.py:70, igmm/pcrpmm.py:105-112
//   + Student-t posterior predictive ............... gaussian/gaussian_components.py:228-251
//   new table: log(alpha) + cached_log_prior[i] .... igmm/crpmm.py:74
//   prob = exp(lp - logsumexp(lp)) .................. igmm/crpmm.py:75
//   u -= prob[j] in label order, first u < 0 wins ... utils/utils.py:15-20
// The visited point's own component is scored with the point REMOVED (del_item, :171-186);
// here that is the closed-form rank-1 downdate of the frozen factor (SlotConst::A1 ...).
// A visit whose drawn label is its current one ("stay") leaves the reference's state
// bit-identical (crpmm.py:82-85), which is what makes evaluating a whole window against
// frozen state exact; the first visit that does not stay is published with atomicMin.
#include "bgmm_device.h"

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ double wave_scan(double v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const double t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}

// LDS tile: entry (j, r) at tile[j * R + r]; R = d.choice_rows visits per block, labels j < K,
// plus one extra line (index K_max + 1) for the "new table" entry.  The block first stages the
// q values of its R consecutive visits (slot-major q: R x 8 contiguous bytes per label), then
// wave r turns column r into log scores in place.
__global__ __launch_bounds__(64 * kChoiceRowsMax) void choice_kernel(Dev d) {
    extern __shared__ __attribute__((aligned(16))) double tile[];
    Ctrl *c = d.ctrl;
    const int mode = c->job.mode;
    if (mode == MODE_DONE || job_is_pruned(d, mode, c->job.prune)) return;      // (pruned windows: choice_sparse_kernel)
    const long long pos = c->job.pos, win_base = c->job.win_base, win_hi = c->job.win_hi;
    const int K = c->job.K;
    const int R = d.choice_rows;
    const long long k0 = (pos - win_base) + (long long)blockIdx.x * R;
    const long long kend = win_hi - win_base;
    if (k0 >= kend) return;
    for (int idx = threadIdx.x; idx < K * R; idx += blockDim.x) {
        const int j = idx / R, r = idx - j * R;
        tile[idx] = k0 + r < kend ? d.q[(long long)d.perm[j] * d.qstride + (k0 + r)] : 0.0;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (k0 + w >= kend) return;
    const long long p = win_base + k0 + w;
    const int NEWIDX = d.K_max + 1;

    const long long i = d.order ? d.order[p] : p;
    const int h = d.z[i];
    const int nh = h >= 0 ? d.n[h] : 0;
    const bool home_live = h >= 0 && nh >= 2;          // removal keeps the component
    const bool singleton = h >= 0 && nh == 1;          // removal deletes it (swap with last)
    const int lab_h = singleton ? d.label_of_slot[h] : -1;
    const int L = singleton ? K - 1 : K;               // labels after the removal

    // pass 1: log scores, in place.  Entry (j, w) is read and written by the one lane that
    // owns label j; a deleted singleton's stand-in (entry K-1) is only ever read, because the
    // new-table score has its own line NEWIDX.
    double mx = -INFINITY;
    for (int j = lane; j <= L; j += 64) {
        double v;
        if (j == L) {
            v = d.log_alpha + d.log_prior[i];
            tile[NEWIDX * R + w] = v;
        } else {
            const int jj = (singleton && j == lab_h) ? K - 1 : j;
            const int s = d.perm[jj];
            const double qv = tile[jj * R + w];
            const SlotConst sc = d.sc[s];
            if (d.cov_type != COV_FULL) {
                // the diag / fixed likelihood kernels store log densities (home row: one-point-removed form)
                v = ((home_live && s == h) ? sc.logseat1 : sc.logseat) + qv;
            } else if (home_live && s == h) {
                const double den = 1.0 - sc.a1 * qv;
                v = sc.logseat1 + sc.A1 - 0.5 * log(den) - sc.half_vd1 * log(1.0 + sc.coef1 * qv / den);
            } else {
                v = sc.logseat + sc.A - sc.half_vd * log(1.0 + qv * sc.inv_cv);
            }
        }
        mx = fmax(mx, v);
        if (j < L) tile[j * R + w] = v;
    }
    mx = wave_max(mx);
    // pass 2: exp and total
    double tot = 0.0;
    for (int j = lane; j <= L; j += 64) {
        const int idx = (j == L ? NEWIDX : j) * R + w;
        const double e = exp(tile[idx] - mx);
        tile[idx] = e;
        tot += e;
    }
    tot = wave_sum(tot);
    // pass 3: sequential-subtract scan in label order, 64 labels at a time
    const double u = d.u[p];
    double carry = 0.0;
    int pick = L;                                        // fallback: last entry (utils.py:20)
    for (int j0 = 0; j0 <= L; j0 += 64) {
        const int j = j0 + lane;
        const double pj = j <= L ? tile[(j == L ? NEWIDX : j) * R + w] / tot : 0.0;
        const double cum = carry + wave_scan(pj, lane);
        const bool hit = j <= L && (u - cum) < 0.0;
        const unsigned long long m = __ballot(hit);
        if (m) { pick = j0 + __ffsll((long long)m) - 1; break; }
        carry = __shfl(cum, 63);
    }
    if (lane == 0) {
        d.choice[p - win_base] = pick;
        const bool stay = home_live && pick < L && d.perm[pick] == h;
        if (!stay) atomicMin(&c->first_mover, (unsigned long long)p);
    }
}

// ------------------------------------------------------------------------------------------
// Draw kernel of a PRUNED window: one thread per visit, evaluation order (visit = wperm[k]).
// The 16 visits of evaluation block b share the block's label mask keep64[b][..]; the exact
// quadratic forms of a kept label are the 128-byte line qb[(b*nslots + label)*16 .. +16].
// Labels whose bit is clear were bounded below e^-80 of the visit's best weight by the pruning
// kernel and enter with weight exactly 0.  The arithmetic follows the reference literally:
//   prob = exp(lp - (max + log(sum exp(lp - max))))       crpmm.py:75 with scipy's logsumexp
//   u -= prob[j] in label order, first u < 0 wins, else the last index    utils.py:15-20
// ------------------------------------------------------------------------------------------
struct SparseVisit {
    const Dev *d;
    const unsigned long long *mask;
    const double *qline;       // + label * 16
    int K, h, lab_h;
    bool home_live, singleton;
};

// exact log score (seating weight + predictive) of old label t for this visit, or false if pruned
__device__ __forceinline__ bool sparse_score(const SparseVisit &sv, int t, double &v) {
    if (!((sv.mask[t >> 6] >> (t & 63)) & 1ull)) return false;
    const Dev &d = *sv.d;
    const int s = d.perm[t];
    const double qv = sv.qline[(long long)t * 16];
    const SlotConst *__restrict__ sc = d.sc + s;
    if (d.cov_type != COV_FULL) {
        // the diag / fixed likelihood kernels store log densities (home row: one-point-removed form)
        v = ((sv.home_live && s == sv.h) ? sc->logseat1 : sc->logseat) + qv;
    } else if (sv.home_live && s == sv.h) {
        const double a1 = sc->a1, den = 1.0 - a1 * qv;
        v = sc->logseat1 + sc->A1 - 0.5 * log(den) - sc->half_vd1 * log(1.0 + sc->coef1 * qv / den);
    } else {
        v = sc->logseat + sc->A - sc->half_vd * log(1.0 + qv * sc->inv_cv);
    }
    return true;
}

__global__ __launch_bounds__(256) void choice_sparse_kernel(Dev d) {
    Ctrl *c = d.ctrl;
    if (!job_is_pruned(d, c->job.mode, c->job.prune)) return;
    // (the bucket sort's bins are free again: cleared here for the next pruned window)
    if (blockIdx.x == 0)
        for (int b = threadIdx.x; b < d.nslots + 2; b += 256) d.bucket_bins[b] = 0;
    const long long win_base = c->job.win_base;
    const long long nrows = prune_count(d);              // (rows certified to stay or decided by home_kernel are not here)
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= nrows) return;
    const int K = c->job.K;
    const long long p = win_base + prune_rows(d)[k];
    const WRec rec = prune_list(d)[k];                   // (index, home, its label, new-table score)
    SparseVisit sv;
    sv.d = &d;
    sv.K = K;
    sv.h = rec.home;
    const int nh = sv.h >= 0 ? d.n[sv.h] : 0;
    sv.home_live = sv.h >= 0 && nh >= 2;
    sv.singleton = sv.h >= 0 && nh == 1;
    sv.lab_h = sv.singleton ? rec.home_label : -1;
    const long long b = k >> 4;
    sv.mask = d.keep64 + b * d.keep_stride;
    sv.qline = d.q + (b * (long long)d.nslots) * 16 + (k & 15);
    const int L = sv.singleton ? K - 1 : K;              // labels after the removal
    const int nw = (K + 63) >> 6;
    const double vnew = rec.mlb0;

    // Post-removal label j is old label j, except that a deleted singleton's place lab_h is taken
    // by old label K-1 (swap with last).  Without a singleton the kept labels are walked by bit
    // scan; the singleton case (rare) walks all L labels.
    double mx = vnew, tot = 0.0, v;
    double vh = -INFINITY, toth = 0.0;          // exact score of the home label; sum of exp(v - mx) over the alternatives
    int pick = L;
    if (!sv.singleton) {
        // the scores of the first four kept labels stay in registers: with at most four (the usual
        // case: the home and a neighbour or two) the second and third pass need no memory at all
        double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
        int t0 = 0, t1 = 0, t2 = 0, t3 = 0, nc = 0;
        for (int wi = 0; wi < nw; ++wi) {
            unsigned long long m = sv.mask[wi];
            if (wi == nw - 1 && (K & 63)) m &= (1ull << (K & 63)) - 1ull;
            while (m) {
                const int t = wi * 64 + __ffsll((long long)m) - 1;
                m &= m - 1;
                if (sparse_score(sv, t, v)) {
                    mx = fmax(mx, v);
                    if (t == rec.home_label) vh = v;
                    if (nc == 0) { v0 = v; t0 = t; } else if (nc == 1) { v1 = v; t1 = t; }
                    else if (nc == 2) { v2 = v; t2 = t; } else if (nc == 3) { v3 = v; t3 = t; }
                    ++nc;
                }
            }
        }
        if (nc <= 4) {
            double e;
            if (nc > 0) { e = exp(v0 - mx); tot += e; if (t0 != rec.home_label) toth += e; }
            if (nc > 1) { e = exp(v1 - mx); tot += e; if (t1 != rec.home_label) toth += e; }
            if (nc > 2) { e = exp(v2 - mx); tot += e; if (t2 != rec.home_label) toth += e; }
            if (nc > 3) { e = exp(v3 - mx); tot += e; if (t3 != rec.home_label) toth += e; }
            e = exp(vnew - mx); tot += e; toth += e;
            const double lse = log(tot) + mx;
            double uu = d.u[p];
            if (nc > 0) { uu -= exp(v0 - lse); if (uu < 0.0) pick = t0; }
            if (nc > 1 && pick == L) { uu -= exp(v1 - lse); if (uu < 0.0) pick = t1; }
            if (nc > 2 && pick == L) { uu -= exp(v2 - lse); if (uu < 0.0) pick = t2; }
            if (nc > 3 && pick == L) { uu -= exp(v3 - lse); if (uu < 0.0) pick = t3; }
        } else {
            for (int wi = 0; wi < nw; ++wi) {
                unsigned long long m = sv.mask[wi];
                if (wi == nw - 1 && (K & 63)) m &= (1ull << (K & 63)) - 1ull;
                while (m) {
                    const int t = wi * 64 + __ffsll((long long)m) - 1;
                    m &= m - 1;
                    if (sparse_score(sv, t, v)) {
                        const double e = exp(v - mx);
                        tot += e;
                        if (t != rec.home_label) toth += e;
                    }
                }
            }
            { const double e = exp(vnew - mx); tot += e; toth += e; }
            const double lse = log(tot) + mx;
            double uu = d.u[p];
            bool done = false;
            for (int wi = 0; wi < nw && !done; ++wi) {
                unsigned long long m = sv.mask[wi];
                if (wi == nw - 1 && (K & 63)) m &= (1ull << (K & 63)) - 1ull;
                while (m) {
                    const int t = wi * 64 + __ffsll((long long)m) - 1;
                    m &= m - 1;
                    if (sparse_score(sv, t, v)) {
                        uu -= exp(v - lse);
                        if (uu < 0.0) { pick = t; done = true; break; }
                    }
                }
            }
        }
        // (the new-table entry is the last index: it wins either by u < 0 or as the fallback)
    } else {
        for (int j = 0; j < L; ++j)
            if (sparse_score(sv, j == sv.lab_h ? K - 1 : j, v)) mx = fmax(mx, v);
        for (int j = 0; j < L; ++j)
            if (sparse_score(sv, j == sv.lab_h ? K - 1 : j, v)) tot += exp(v - mx);
        tot += exp(vnew - mx);
        const double lse = log(tot) + mx;
        double uu = d.u[p];
        for (int j = 0; j < L; ++j)
            if (sparse_score(sv, j == sv.lab_h ? K - 1 : j, v)) {
                uu -= exp(v - lse);
                if (uu < 0.0) { pick = j; break; }
            }
    }
    d.choice[p - win_base] = pick;
    if (sv.home_live) {
        // for certify_kernel, while nothing changes: log of the total weight of all alternatives relative
        // to the home's -- the scored ones exactly, every pruned label below e^-80 of the best score
        PCacheExact pe;
        pe.epoch = c->state_epoch;
        pe.log_alt = log(toth * exp(mx - vh) + (double)K * exp(mx - 80.0 - vh));
        d.pcache2[rec.i] = pe;
    }
    const bool stay = sv.home_live && pick < L && d.perm[pick] == sv.h;
    if (!stay) atomicMin(&c->first_mover, (unsigned long long)p);
}

void launch_choice_sparse(const Dev &d, long long max_rows, hipStream_t st) {
    if (max_rows <= 0) return;
    hipLaunchKernelGGL(choice_sparse_kernel, dim3((unsigned)((max_rows + 255) / 256)), dim3(256), 0, st, d);
}

int choice_rows_for(int K_max) {
    int R = kChoiceRowsMax;
    while (R > 1 && (long long)(K_max + 2) * R * (long long)sizeof(double) > 64 * 1024) R >>= 1;
    return R;
}

void launch_choice(const Dev &d, long long max_rows, hipStream_t st) {
    if (max_rows <= 0) return;
    const int R = d.choice_rows;
    const unsigned gx = (unsigned)((max_rows + R - 1) / R);
    const int lds = (d.K_max + 2) * R * (int)sizeof(double);
    static PerDeviceLds attr;
    attr.ensure((const void *)choice_kernel, lds);
    hipLaunchKernelGGL(choice_kernel, dim3(gx), dim3(64 * R), lds, st, d);
}

// Helpers shared by the likelihood kernels (kernels_score.hip: every pair evaluated;
// kernels_prune.hip: certified stays and pruned windows).
#pragma once
#include "bgmm_device.h"
#include <type_traits>

typedef double v4d __attribute__((ext_vector_type(4)));

// log score of a (visit, slot) pair from its exact quadratic form -- the formulas of the draw kernel
__device__ __forceinline__ double slot_score_exact(const SlotConst &sc, double qv, bool home_minus_one) {
    if (home_minus_one) {
        const double den = 1.0 - sc.a1 * qv;
        return sc.logseat1 + sc.A1 - 0.5 * log(den) - sc.half_vd1 * log(1.0 + sc.coef1 * qv / den);
    }
    return sc.logseat + sc.A - sc.half_vd * log(1.0 + qv * sc.inv_cv);
}

// Cheap LOWER bound of the same score (log(1 + t) <= t; for the home form -0.5 log(den) >= 0 because
// 0 < den <= 1): what the pruning kernel raises a visit's best-score bound with.
__device__ __forceinline__ double slot_score_lower(const SlotConst &sc, double qv, bool home_minus_one) {
    if (home_minus_one) {
        const double den = 1.0 - sc.a1 * qv;
        return den > 0.0 ? sc.logseat1 + sc.A1 - sc.half_vd1 * (sc.coef1 * qv / den) : -INFINITY;
    }
    return sc.logseat + sc.A - sc.half_vd * (qv * sc.inv_cv);
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
    if (job.mode == MODE_SLOTS) return t;
    if (job.mode == MODE_LIST) return d.slot_list[t];
    return job.mode == MODE_FRESH ? d.perm[t] : (t == 0 ? job.dirty0 : job.dirty1);
}


// x tile of the diag / fixed kernels in LDS: element (dimension l, visit r) at xs[l * kDiagLd + r];
// the odd stride keeps the transposing writes (consecutive l) off a single bank
static constexpr int kDiagLd = kValuRows + 1;
// up to this many dimensions the 64 rows of a block sit transposed in LDS (58 KB + the per-visit arrays: under the 64 KB a
// launch gets without asking); beyond, every lane reads its own row through the cache, dimension by dimension
static constexpr int kDiagLdsMaxD = 112;

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

constexpr int pick_ring(int nf, int cap) {
    int best = 1;
    for (int p = 1; p <= cap && p <= nf; ++p)
        if (nf % p == 0) best = p;
    return best;
}

constexpr int pick_pf(int nf) {
    int best = 1;
    for (int p = 1; p <= 12 && p <= nf; ++p)
        if (nf % p == 0) best = p;
    return best;
}

// MINW = waves per SIMD the register budget is planned for: 3 up to D = 64 (168 VGPRs with a
// 10..12-deep ring), 2 at D = 80, 1 for D = 96..128 (2 x 16 rows of A fragments alone are
// 96..128 VGPRs).  api_context.hip (resolve_kind) sizes the grid (label chunks) to a whole number of residency rounds.

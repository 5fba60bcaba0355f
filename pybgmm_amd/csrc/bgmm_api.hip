// Host side of libbgmm_hip.so: context, memory, the per-sweep launch schedule, and the
// extern "C" entry points declared in include/bgmm.h.
//
// Sweep schedule (device driven, no host round trip per visit):
//   sweep_begin                      reset window at visit 0, seating weights for this sweep
//   repeat "steps" (queued blindly in chunks of T; a step is a no-op once the sweep is DONE):
//     score   likelihood kernel over the window x {all labels | the <=2 slots a move touched}
//     choice  one categorical draw per visit against the frozen state; atomicMin(first mover)
//     apply   no mover: commit the window, open the next;   mover: commit the stays before
//             it, apply the move (rank-1 statistics change), continue after it
//     refresh Cholesky / inverse / constants of the <=2 touched slots
//   after each chunk the host reads the control block (one small D2H + stream sync).
#include "../../include/bgmm.h"
#include "bgmm_device.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <string>
#include <vector>

int choice_rows_for(int K_max);
void launch_contingency(const Dev &d, const long long *true_idx, int K_true, unsigned long long *table,
                        hipStream_t st);
void launch_dispersion(const Dev &d, double *out, hipStream_t st);
void launch_set_stats(const Dev &d, int label, const double *m_in, const double *S_in, int count, hipStream_t st);
void launch_set_label(const Dev &d, long long i, int label, hipStream_t st);
void launch_raw_stats(const Dev &d, int label, double *m_out, double *S_out, hipStream_t st);
void launch_del_component(const Dev &d, int label, hipStream_t st);
void launch_init_labels(const Dev &d, const long long *z_in, int K_init, hipStream_t st);
void launch_export_stats(const Dev &d, int K, double *m_out, double *S_out, double *logdet_out,
                         double *inv_out, hipStream_t st);

static thread_local std::string g_create_error;
// The permutations' worker thread (PermPipe::worker) reports through a string of its own: bgmm_ctx::err belongs to the
// thread that drives the context.  CK / fail write to err_of(ctx).
static thread_local std::string *g_err_sink = nullptr;

// Chains side by side on one GPU (bgmm_group_sweep_staged) keep one stream each busy.  The HIP runtime maps a process's
// streams onto GPU_MAX_HW_QUEUES hardware queues (default 4; streams that share a queue run one behind the other).  The
// library does NOT touch the process environment (a setenv from a static initializer races with getenv elsewhere and
// changes the queue mapping of every HIP user of the process): a caller that runs more than four chains per device
// exports GPU_MAX_HW_QUEUES=8 before the runtime's first call -- pybgmm_amd._lib does, unless told not to.

// scratch of the device permutations (perm_ensure)
struct PermPtrs {
    unsigned *dkey, *dkey_out, *dspare, *draw, *dwords, *ks, *idx, *iota;
    int *dpos_out, *dspare_pos, *J, *pred, *ptr, *changed, *flags, *cnt;
    long long n_words_cap;
};

struct bgmm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    Dev d{};
    std::string err;
    std::vector<void *> allocs;
    Ctrl *ctrl_host = nullptr;       // pinned mirror
    Ctrl *ctrl_pub = nullptr;        // pinned and device-mapped: apply_kernel publishes the control block here (lean batches)
    Job *util_job = nullptr;         // device
    double *util_q = nullptr;        // device [ldq]
    double *util_out = nullptr;      // device [nslots + 8]
    double *d_u = nullptr;
    long long *d_order = nullptr;
    bool have_order = false;
    bool assigned = false;
    double *res_u = nullptr;         // resident multi-sweep inputs
    long long *res_order = nullptr;
    int res_n = 0;
    const double *cur_u = nullptr;   // inputs of the sweep being run
    const long long *cur_order = nullptr;
    int kernel_kind = KERNEL_AUTO;
    int kind = KERNEL_VALU;          // resolved
    int win_rows = 0;                // allocated q / choice rows
    double last_move_rate = 0.0;     // movers per visit of the previous sweep
    int resolver_mode = 0;           // 0 auto, 1 off, 2 always when it fits
    bool order_is_perm = true;       // the staged visiting order visits every point exactly once (or is absent)
    int prune_mode = 0;              // 0 auto (on with the MFMA kernel), 1 off, 2 every window (tests),
                                     // 3 auto without certified stays (measurement)
    double *tabSeat = nullptr;       // seating-weight table (rebuilt when the exponent changes)
    int seat_use_power = 0;
    double seat_power = 1.0;
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev0, ev1;
    long long timed_launches = 0;
    double timed_ms = 0.0;
    long long prune_mfma = 0, certified = 0;
    // A pruned component enters a draw with probability 0 instead of < 2e-35.  The reference's
    // `u -= p` scan can tell the difference only for u == 0 exactly (it would return the first label
    // with a positive probability), so sweeps whose uniform stream contains an exact zero are run
    // unpruned.
    bool lean_ok = false;            // the previous sweep certified every visit and moved nothing
    // the visiting order drawn on the device (bgmm_stage_permutation_mt19937, kernels_perm.hip)
    long long *d_order_ahead = nullptr;   // the look-ahead's permutation (swapped with d_order when it is taken)
    hipStream_t perm_stream = nullptr;
    hipEvent_t perm_done = nullptr;
    bool perm_ahead_valid = false;
    int perm_ahead_pos_in = 0;
    long long perm_hits = 0, perm_misses = 0;
    int perm_last_rounds = 0, perm_max_rounds = 0;   // rounds of draws until the last / the slowest permutation settled
    bool order_staged = false;       // d_order holds a permutation staged for the NEXT sweep: a stage call without an order keeps it
    unsigned *perm_words = nullptr;  // [key in 624 | key out 624 | pos out 16 | spare key 624 | spare pos 16 | raw | untempered words]
    unsigned *perm_seeds = nullptr;  // the chains' seeds (its own: the uniforms' look-ahead may be running beside it)
    long long perm_n_words = 0;
    int perm_chains = 0;
    int *perm_ints = nullptr;        // J, pred, ptr [N] each, then changed
    unsigned *perm_uints = nullptr;  // ks, idx, iota [N] each
    void *perm_temp = nullptr;
    size_t perm_temp_bytes = 0;
    long long *perm_out = nullptr;   // {words consumed, ran out}
    unsigned *perm_host = nullptr;   // pinned: [key out 624 | pos out | changed | out (2 x 64 bit)]
    // Permutations in flight (look-ahead of the caller's numpy stream, "permutations in flight" below): kPermAhead generations
    // queued behind the one being handed out, each taking its place in the word stream from the one in front of it ON THE
    // DEVICE.  Three streams: the draws (perm_stream: the only serial chain from one generation to the next), the swaps
    // (fin: sort, links, assembly, verdicts), the words (rawst: chunks of one long stream, far ahead of the draws).
    struct PermPipe {
        static constexpr int kAhead = 3;
        bool built = false, valid = false;
        hipStream_t fin = nullptr, rawst = nullptr;
        hipEvent_t ev_draw[kAhead] = {}, ev_fin[kAhead] = {}, ev_raw = nullptr, ev_sweep = nullptr;
        unsigned *era_raw = nullptr;        // era_raw[k]: the k-th (untempered) output behind the state the era began at
        long long era_cap = 0;              // words the buffer holds
        long long era_gen_words = 0;        // ... that have been queued for generation
        unsigned *era_key = nullptr;        // device: [624] the state the era began at
        unsigned *era_key_host = nullptr;   // pinned
        int era_pos = 0;
        long long *goffs = nullptr;         // device ring [8]: where generation g starts in the era (g % 8); -1: failed
        int *cnt = nullptr;                 // [5 T] the segments' counts and memos
        int *pre0 = nullptr;                // [T + 1] prefix of the expected counts: round 1's starts
        int *zero[2] = {};                  // {block sums, round flags}: two blocks, alternating, each cleared by the generation in front
        int nblk_pad = 0;
        int rounds_q = 36;                  // rounds queued per generation (follows what the slowest generation so far needed)
        int rounds_floor = 0;               // ... never fewer than this (raised when a generation did not settle in rounds_q)
        int rounds_fixed = [] { const char *e = getenv("BGMM_PERM_CHAIN_ROUNDS"); const int v = e ? atoi(e) : 0;
                                return v < 0 ? 0 : (v > 60 ? 60 : v); }();   // (for the test of that repair)
        int *J[kAhead] = {};
        int *vblk[kAhead] = {};             // a generation's verdicts, laid out like perm_host: [624 key | pos | went through | - |
                                            // swaps overflowed | out (2 x 64 bit) | ... | round flags at 1280], one copy to the host
        int NB = 0;                         // the swaps by buckets of targets (kernels_perm.hip): their number (0: rocPRIM's sort),
        int *bnd = nullptr, *cursor = nullptr;          // boundaries [NB + 1], fill counts [NB]
        unsigned long long *slots = nullptr;            // [NB][perm_bucket_cap()] (target << 32 | step)
        long long *ord[kAhead] = {};
        long long *parked = nullptr;        // the order buffer released by the last call: written again one call later at the
                                            // earliest (a sweep begun and not yet ended may be redone from it: finish_pending)
        unsigned *host[kAhead] = {};        // pinned verdicts, laid out like perm_host
        long long gen_next = 0, gen_queued = 0;   // the generation the next call takes / generations queued so far
        long long off_exact = 0;            // where generation gen_next starts in the era
        long long cap_words = 0;            // words one generation may read
        uint32_t expect_key[624] = {};      // the caller's state iff it took the last permutation and drew nothing else
        int expect_pos = -1;
        // Queueing a generation is ~65 launches (237 us of host time, measured) -- more than the sweep it feeds takes on the
        // device.  A thread of the context does it: the stage call posts how many generations should be in the queues
        // (target) and goes on to queue the sweep.  mu guards target / gen_queued / gen_next / off_exact / busy / full / wrc.
        PermPtrs P = {};
        std::thread worker;
        std::mutex mu;
        std::condition_variable cv;
        long long target = 0;
        bool busy = false, quit = false, full = false;     // full: the era has no room for another generation
        int wrc = 0;
        std::string werr;                   // the worker's last error text (written under mu, never bgmm_ctx::err)
        int wfails = 0;                     // generations in a row the worker could not queue
        bool w_jump = true;                 // what the worker may read of the context, posted with the target (under mu)
        unsigned *w_coef = nullptr;
        // Set-up is all or nothing: what perm_pipe_build made so far is released when a step fails, `off` is latched and the
        // context stays on the single look-ahead of round 3 (INTEGRATION.md "memory of the permutations in flight").
        bool off = false;
        std::string off_why;
        std::vector<void *> dev_allocs;
    } pp;
    // bgmm_sweep_staged_begin / _end: a sweep whose first batch of launches is in the queue and has not been waited for
    bool async_pending = false, async_short = false;
    bool run_zero_u = false, run_order_is_perm = true;   // what the sweep being run was staged with (snapshots: a stage call between
                                                         // bgmm_sweep_staged_begin and _end describes the NEXT sweep)
    int async_rc = 0;
    bool defer_mt = false, defer_mt_hit = false, defer_perm = false;   // look-ahead launches a stage call put off meanwhile
    int defer_mt_pos = 0;
    std::vector<uint32_t> defer_mt_key;
    int grp_cap = 0;                 // bgmm_group_sweep_staged: the LDS plan phase 1 of sweep_impl chose for the one-workgroup sweep
    Dev *grp_devs = nullptr;         // device array of the chains' views (owned by the chain that leads a group launch)
    struct GramCombiner *combiner = nullptr;   // bgmm_group_sweep_staged: the rendezvous of the chains' host threads (below)
    int combiner_slot = -1;
    hipEvent_t grp_ev_in = nullptr, grp_ev_out = nullptr;   // this chain's stream has reached the batch / the shared launches are queued
    int grp_devs_cap = 0;
    long long short_stood = 0, short_refused = 0;   // short steps over the life of the context (bgmm_get_short_step_stats)
    bool short_ok = false;           // the previous sweep (certified stays off) was ONE pruned window, moved nothing and
                                     // home_kernel decided every visit: the next one tries a short step (Dev::short_step)
    long long moves_prev = -1;       // moves of the previous sweep (-1: none yet / state set from outside)
    long long *true_dev = nullptr;   // bgmm_contingency: the reference labelling, kept between calls
    unsigned long long *table_dev = nullptr;
    size_t table_cells = 0;
    unsigned *mt_words = nullptr;    // device scratch of bgmm_stage_mt19937 (layout there)
    unsigned *mt_coef = nullptr, *mt_seeds = nullptr;   // jump polynomials / seeds of the chains of a long request
    int mt_chains = 0;
    bool mt_jump_on = true;          // bgmm_set_mt_jump: false = the chains run one after the other (the r02 route, for comparison)
    // Look-ahead of the caller's stream (bgmm_set_mt_lookahead): the uniforms of the next `depth` sweeps are generated in one
    // request on a second stream, beside the running sweep, into one of two batch buffers; a bgmm_stage_mt19937 call is
    // served from the batch iff the state it is handed is bit for bit the state at that sweep boundary of the batch --
    // i.e. the caller drew nothing in between.  While the last sweep of a batch is served the next batch is started.
    bool mt_ahead_on = true;
    int mt_depth = 0;                // sweeps per batch (0: chosen from N at first use)
    hipStream_t mt_stream = nullptr;
    struct MtBatch {
        bool launched = false, synced = false;
        int next = 0;                // sweep of the batch the next hit serves
        int pos_in = 0;
        hipEvent_t done = nullptr;
        double *u = nullptr;         // [depth][N]
        unsigned *host = nullptr;    // pinned: [start key 624 | state behind sweep j: depth x 624 | their positions depth | zero flags depth]
    } mt_b[2];
    int mt_cur = -1;                 // batch being served
    unsigned *mt_words_ahead = nullptr;   // device scratch of a batch generation (layout in mt_launch_batch)
    long long mt_ahead_hits = 0, mt_ahead_misses = 0;
    bool cur_zero_u = false;
    std::vector<char> res_zero_u;
    std::vector<char> res_perm;      // per resident sweep: its order is a permutation (or absent)
    long long totals[4] = {0, 0, 0, 0};    // since the context was made: sweeps, pairs decided, moves, pairs executed (bgmm_get_totals)
    long long stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long long stats2[4] = {0, 0, 0, 0};   // pairs whose quadratic form was executed, frozen-factor windows, their rows, spare
    // frozen-factor windows (kernels_gram.hip): buffers sized for `gcols` columns, re-allocated when the labels outgrow them
    void *gram_mem[24] = {};         // [0 .. 9] the window buffers, [10 .. 19] their twins (pipelined windows), [20] gX
    // pipelined frozen-factor windows (kernels_gram.hip): a second stream for gram_finish / the cross forms of the window after
    // next, events between the two
    hipStream_t pipe_stream = nullptr;
    std::vector<hipEvent_t> pipe_ev;
    int pipe_mode = [] { const char *e = getenv("BGMM_GRAM_PIPE"); return e ? atoi(e) : 1; }();   // 0: never (plain windows)
    long long pipe_batches = 0, pipe_breaks = 0;
    int pipe_hold = 0;               // plain batches to go before pipelined ones are tried again (after a break)
    int gram_lds = 0;
    bool gram_off = false;           // this context cannot use them (their buffers failed to allocate three times)
    int gram_alloc_fail = 0;
    bool tables_robust = false;      // the pruning tables on the device carry a safe-stay batch's robust constants
    long long next_sweep_visits = 0; // bgmm_set_sweep_visits: the next sweep stops after this many visits (0: a whole sweep)
    int safe_rest = 0;               // sweeps to go without safe-stay windows: a batch of them covered fewer visits per
                                     // millisecond than the per-mover kernel chain is known to (they are tried again a sweep later)
    // safe-stay windows (kernels_safe.hip)
    // safe-stay windows: which kind of proof pass the next batch runs (Dev::safe_dense).  -1: the chain decides (dense once the
    // per-home tables left more than half of a batch's visits to the exact forms; looked at again every eighth sweep);
    // 0 / 1: pinned (BGMM_SAFE_DENSE in the environment, for experiments and tests)
    int safe_dense_pin = [] { const char *e = getenv("BGMM_SAFE_DENSE"); return e ? atoi(e) : -1; }();
    bool safe_dense_on = false;
    int safe_dense_age = 0;
    long long proof_batches[2] = {0, 0};      // batches of safe-stay windows queued with a table / a dense proof pass
    double safe_cap_user = 0.0;      // bgmm_set_safe_budget: > 0 pins the per-component budget of a window (0: it follows the chain)
    long long safe_stats[6] = {0, 0, 0, 0, 0, 0};
    // the look-ahead of the dense proof pass (kernels_safe.hip): its stream, a ring of event pairs (plan made / request served)
    int ahead_chunk = 8192;          // visits per chunk (a power of two; 0: off -- bgmm_set_proof_lookahead)
    hipStream_t ahead_stream = nullptr;
    hipEvent_t ahead_ev[2][8] = {};
    int seq_cap = 0;                 // labels the one-workgroup sweep plans LDS for (0: as many as fit)
    bool home_pass = true;           // home_kernel in front of the pruning kernel (kernels_home.hip)
    int home_retry = 0;
    int home_mode = 0;               // bgmm_set_home_pass: 0 auto, 1 always, 2 never, 3 always + a short step tried in every sweep
};

// Mean distance between movers below which the frozen-factor windows take over from the per-mover
// kernel chain: a window costs ~60 us plus ~1.5 us per mover and covers 64 visits, the chain ~190 us per mover.
constexpr double kGramRun = 192.0;
// Safe-stay windows (kernels_safe.hip) cover the regime in between: from one mover in kSafeRun visits up to one in
// four.  A safe-stay window costs three to four plain ones (the proof pass in front of it) and walks the visits it
// could not prove -- one per mover where clusters are apart, two to seven where they overlap -- so what decides is the
// share of visits it has to walk, measured batch by batch: above a quarter the rest of the sweep goes to plain
// frozen-factor windows (C4's shape: 1.6 % movers 1.2 s per sweep against 3.0 s; 11.6 % movers 5.6 s against 3.0 s).
constexpr double kSafeRun = 65536.0;
constexpr double kSafeDenseRate = 0.25;
constexpr double kSafeWalkShare = 0.25;

static inline std::string &err_of(bgmm_ctx *c) { return g_err_sink ? *g_err_sink : c->err; }
#define CK(ctx, call)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) {                                                             \
            err_of(ctx) = std::string(#call) + ": " + hipGetErrorString(e_);               \
            return BGMM_EDEVICE;                                                            \
        }                                                                                   \
    } while (0)

template <typename T>
static int dalloc(bgmm_ctx *c, T **p, size_t count) {
    void *q = nullptr;
    hipError_t e = hipMalloc(&q, count * sizeof(T) + 64);
    if (e != hipSuccess) {
        err_of(c) = std::string("hipMalloc: ") + hipGetErrorString(e);
        return BGMM_EDEVICE;
    }
    c->allocs.push_back(q);
    *p = (T *)q;
    return 0;
}
#define DALLOC(ctx, ptr, count)                         \
    do {                                                \
        int rc_ = dalloc((ctx), &(ptr), (count));       \
        if (rc_) return rc_;                            \
    } while (0)

static int finish_pending(bgmm_ctx *c);
// Entry points that read or change what a sweep left in the queue by bgmm_sweep_staged_begin is working on finish that
// sweep first (bgmm_sweep_staged_end then just reports its status).
#define SETTLE(c) do { if ((c)->async_pending) { const int rc_ = finish_pending(c); if (rc_) return rc_; } } while (0)

static int fail(bgmm_ctx *c, int code, const std::string &msg) {
    if (c) err_of(c) = msg;
    else g_create_error = msg;
    return code;
}

static const char *err_text(int code) {
    switch (code) {
        case -3: return "K_max exceeded: a new component was drawn while all K_max slots are in use";
        case -4: return "a component scatter matrix is not positive definite";
        case -1: return "invalid label";
        default: return "device-side error";
    }
}

static int check_device_error(bgmm_ctx *c) {
    // ctrl_host must be current
    if (c->ctrl_host->error != 0) {
        const int e = c->ctrl_host->error;
        return fail(c, e == -3 ? BGMM_EKMAX : e == -4 ? BGMM_ENOTPD : BGMM_EINVAL, err_text(e));
    }
    return 0;
}

static int fetch_ctrl(bgmm_ctx *c) {
    CK(c, hipMemcpyAsync(c->ctrl_host, c->d.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, c->stream));
    CK(c, hipStreamSynchronize(c->stream));
    return 0;
}

static void resolve_kind(bgmm_ctx *c) {
    int k = c->kernel_kind;
    if (c->d.cov_type != COV_FULL) k = KERNEL_VALU;      // (the diag / fixed likelihood kernel has the VALU geometry)
    if (k == KERNEL_AUTO) k = (c->d.D >= 12) ? KERNEL_MFMA : KERNEL_VALU;
    if (k == KERNEL_MFMA && c->d.Dp / 16 > 8) k = KERNEL_VALU;
    c->kind = k;
    c->d.rows_per_block = (k == KERNEL_MFMA) ? kMfmaRows : kValuRows;
    // 256 CUs x resident blocks per CU of the chosen kernel (see the MINW note in kernels_score.hip)
    const int nJ = c->d.Dp / 16;
    c->d.target_blocks = (k == KERNEL_MFMA) ? 256 * (nJ <= 4 ? 3 : (nJ <= 5 ? 2 : 1)) : 1024;
}

extern "C" const char *bgmm_version(void) { return "bgmm-hip 0.1 gfx950"; }

extern "C" const char *bgmm_last_error(const bgmm_ctx *ctx) {
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

extern "C" void bgmm_destroy(bgmm_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    // (the look-aheads may still be writing into buffers that are about to go)
    if (c->pp.worker.joinable()) {
        { std::lock_guard<std::mutex> g(c->pp.mu); c->pp.quit = true; }
        c->pp.cv.notify_all();
        c->pp.worker.join();
    }
    if (c->mt_stream) (void)hipStreamSynchronize(c->mt_stream);
    if (c->perm_stream) (void)hipStreamSynchronize(c->perm_stream);
    if (c->pp.fin) (void)hipStreamSynchronize(c->pp.fin);
    if (c->pp.rawst) (void)hipStreamSynchronize(c->pp.rawst);
    for (auto e : c->ev0) (void)hipEventDestroy(e);
    for (auto e : c->ev1) (void)hipEventDestroy(e);
    for (void *p : c->allocs) (void)hipFree(p);
    if (c->mt_stream) { (void)hipStreamSynchronize(c->mt_stream); (void)hipStreamDestroy(c->mt_stream); }
    for (auto &b : c->mt_b) {
        if (b.done) (void)hipEventDestroy(b.done);
        if (b.u) (void)hipFree(b.u);
        if (b.host) (void)hipHostFree(b.host);
    }
    if (c->mt_words_ahead) (void)hipFree(c->mt_words_ahead);
    if (c->grp_devs) (void)hipFree(c->grp_devs);
    if (c->grp_ev_in) (void)hipEventDestroy(c->grp_ev_in);
    if (c->grp_ev_out) (void)hipEventDestroy(c->grp_ev_out);
    if (c->pp.fin) { (void)hipStreamSynchronize(c->pp.fin); (void)hipStreamDestroy(c->pp.fin); }
    if (c->pp.rawst) { (void)hipStreamSynchronize(c->pp.rawst); (void)hipStreamDestroy(c->pp.rawst); }
    if (c->perm_stream) { (void)hipStreamSynchronize(c->perm_stream); (void)hipStreamDestroy(c->perm_stream); }
    for (int k = 0; k < bgmm_ctx::PermPipe::kAhead; ++k) {
        if (c->pp.ev_draw[k]) (void)hipEventDestroy(c->pp.ev_draw[k]);
        if (c->pp.ev_fin[k]) (void)hipEventDestroy(c->pp.ev_fin[k]);
        if (c->pp.host[k]) (void)hipHostFree(c->pp.host[k]);
    }
    if (c->pp.ev_raw) (void)hipEventDestroy(c->pp.ev_raw);
    if (c->pp.ev_sweep) (void)hipEventDestroy(c->pp.ev_sweep);
    if (c->pp.era_raw) (void)hipFree(c->pp.era_raw);
    if (c->pp.era_key_host) (void)hipHostFree(c->pp.era_key_host);
    for (void *p : c->pp.dev_allocs) (void)hipFree(p);
    if (c->perm_done) (void)hipEventDestroy(c->perm_done);
    if (c->perm_words) (void)hipFree(c->perm_words);
    if (c->perm_seeds) (void)hipFree(c->perm_seeds);
    if (c->perm_ints) (void)hipFree(c->perm_ints);
    if (c->perm_uints) (void)hipFree(c->perm_uints);
    if (c->perm_temp) (void)hipFree(c->perm_temp);
    if (c->perm_out) (void)hipFree(c->perm_out);
    if (c->perm_host) (void)hipHostFree(c->perm_host);
    if (c->mt_words) (void)hipFree(c->mt_words);
    if (c->mt_coef) (void)hipFree(c->mt_coef);
    if (c->mt_seeds) (void)hipFree(c->mt_seeds);
    for (void *p : c->gram_mem) if (p) (void)hipFree(p);
    if (c->pipe_stream) { (void)hipStreamSynchronize(c->pipe_stream); (void)hipStreamDestroy(c->pipe_stream); }
    if (c->ahead_stream) { (void)hipStreamSynchronize(c->ahead_stream); (void)hipStreamDestroy(c->ahead_stream); }
    for (auto &row : c->ahead_ev) for (hipEvent_t e : row) if (e) (void)hipEventDestroy(e);
    for (auto e : c->pipe_ev) (void)hipEventDestroy(e);
    if (c->true_dev) (void)hipFree(c->true_dev);
    if (c->table_dev) (void)hipFree(c->table_dev);
    if (c->res_u) (void)hipFree(c->res_u);
    if (c->res_order) (void)hipFree(c->res_order);
    if (c->ctrl_host) (void)hipHostFree(c->ctrl_host);
    if (c->ctrl_pub) (void)hipHostFree(c->ctrl_pub);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

static int create_impl(bgmm_ctx *c, int device, int64_t N, int32_t D, int32_t K_max, int32_t cov_type,
                       const double *X, const double *m_0, double k_0, int64_t v_0,
                       const double *S_0, double alpha, const double *lgamma_tab,
                       const double *log_tab) {
    c->device = device;
    CK(c, hipSetDevice(device));
    CK(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
    Dev &d = c->d;
    d.cov_type = cov_type;
    d.N = N; d.D = D; d.Dp = (D + 15) / 16 * 16; d.K_max = K_max; d.nslots = K_max + 1;
    d.nfrag = bgmm_nfrag(d.Dp); d.ldq = d.nslots;
    d.v0 = v_0; d.k0 = k_0; d.alpha = alpha; d.log_alpha = std::log(alpha);
    d.tab_len = v_0 + N + 2;
    d.use_power = 0; d.power = 1.0; d.order = nullptr; d.u = nullptr; d.prune_enabled = 0;
    d.use_certify = 0; d.lean_step = 0; d.seat_dirty = 0;
    d.batch_rows = 1 << 30;
    resolve_kind(c);

    const bool diag = cov_type != COV_FULL;                  // D-vector statistics (diag and fixed)
    const bool fixed = cov_type == COV_FIXED;
    const size_t DD = fixed ? (size_t)2 * D : diag ? (size_t)D : (size_t)D * D, ns = (size_t)d.nslots;   // second-moment block
    const size_t WW = diag ? 1 : (size_t)D * D;                                   // factor block (full only)
    double *dX, *dtl, *dtg, *dpm, *dpS, *dtG, *dtC, *dtS;
    DALLOC(c, dX, (size_t)N * D);
    DALLOC(c, d.log_prior, (size_t)N);
    DALLOC(c, d.z, (size_t)N);
    DALLOC(c, dtl, (size_t)d.tab_len);
    DALLOC(c, dtg, (size_t)d.tab_len);
    DALLOC(c, dtG, (size_t)d.tab_len);
    DALLOC(c, dtC, (size_t)N + 2);
    DALLOC(c, dtS, (size_t)N + 2);
    DALLOC(c, dpm, (size_t)D);
    DALLOC(c, dpS, DD);
    DALLOC(c, d.m, ns * D);
    DALLOC(c, d.S, ns * DD);
    DALLOC(c, d.mu, ns * D);
    DALLOC(c, d.Wrm, ns * WW);
    DALLOC(c, d.Wfrag, diag ? 64 : ns * d.nfrag * 64);
    DALLOC(c, d.dw, ns * D);
    DALLOC(c, d.cvec, ns * d.Dp);
    DALLOC(c, d.n, ns);
    DALLOC(c, d.nupd, ns);
    DALLOC(c, d.sc, ns);
    DALLOC(c, d.mu_ver, ns);
    DALLOC(c, d.pcache, (size_t)N);
    DALLOC(c, d.pcache2, (size_t)N);
    CK(c, hipMemsetAsync(d.mu_ver, 0, sizeof(int) * ns, c->stream));
    CK(c, hipMemsetAsync(d.pcache, 0xff, sizeof(PCache) * (size_t)N, c->stream));       // (tags: no slot)
    CK(c, hipMemsetAsync(d.pcache2, 0xff, sizeof(PCacheExact) * (size_t)N, c->stream)); // (epochs: none)
    DALLOC(c, d.perm, ns);
    DALLOC(c, d.label_of_slot, ns);
    DALLOC(c, d.ctrl, 1);
    DALLOC(c, c->util_job, 1);
    DALLOC(c, c->util_q, (size_t)d.ldq);
    DALLOC(c, c->util_out, ns + 8);
    DALLOC(c, c->d_u, (size_t)N);
    DALLOC(c, c->d_order, (size_t)N);
    // speculative window: 2^20 visits (up to 2^22 for larger N), q bounded by 16 GiB (288 GB of HBM per GPU).  Large
    // windows matter in the sparse-mover regime: the fixed cost of a step is paid once per window and
    // the pruned-window kernel overlaps its latency-bound phases over more workgroup rounds.
    long long rows = 1ll << 20;
    while (rows < N && rows < (1ll << 22)) rows <<= 1;       // (one window per sweep up to 4 Mi visits, memory permitting)
    while (rows > 1024 && (size_t)rows * d.nslots * sizeof(double) > ((size_t)16 << 30)) rows >>= 1;
    long long n_up = (N + kMfmaRows - 1) / kMfmaRows * kMfmaRows;
    if (rows > n_up) rows = n_up;
    c->win_rows = (int)rows;
    d.qstride = rows;
    d.choice_rows = choice_rows_for(K_max);
    DALLOC(c, d.q, (size_t)rows * d.nslots);
    DALLOC(c, d.choice, (size_t)rows);
    const size_t rows_pad = (size_t)rows + (size_t)kHomeBlock * (d.nslots + 2);   // (home_kernel's padded evaluation order)
    DALLOC(c, d.wperm, rows_pad);
    {
        const size_t ng = ((size_t)d.nslots + 15) / 16;
        DALLOC(c, d.pr_mufrag, ng * (size_t)(d.Dp / 4) * 64);
        DALLOC(c, d.pr_const, ng * 128);
        DALLOC(c, d.pr_slot, ng * 16);
        DALLOC(c, d.pr_dcc, (size_t)d.nslots * d.nslots);
        DALLOC(c, d.pr_rms, (size_t)d.nslots);
        DALLOC(c, d.wrec, rows_pad);
        DALLOC(c, d.wrecR, (size_t)rows);
        DALLOC(c, d.wpermR, (size_t)rows);
        DALLOC(c, d.pr_counts, 1024);
        CK(c, hipMemsetAsync(d.pr_counts, 0, 1024 * sizeof(unsigned long long), c->stream));
        DALLOC(c, d.cert, (size_t)rows);
        DALLOC(c, d.ftab, (size_t)d.nslots * 64);
        DALLOC(c, d.finv, (size_t)d.nslots);
        DALLOC(c, d.ftab2, (size_t)d.nslots * 64);
        DALLOC(c, d.nbr, (size_t)d.nslots * 4);
    }
    DALLOC(c, d.ah_job, 3);
    DALLOC(c, d.resc_job, 1);
    DALLOC(c, d.resc_list, ns);
    DALLOC(c, d.touch_seq, ns);
    CK(c, hipMemsetAsync(d.touch_seq, 0, sizeof(long long) * ns, c->stream));
    d.ahead_C = 0; d.slot_list = nullptr;
    DALLOC(c, d.glist, (size_t)kSafeList + 1);
    DALLOC(c, d.ep_state, ns);
    DALLOC(c, d.rtab, ns * 8);
    DALLOC(c, d.ftabR, ns * 64);
    d.safe_mode = 0; d.safe_cap = 0.0;
    d.keep_stride = (d.nslots + 63) / 64;
    DALLOC(c, d.keep64, (size_t)(rows / 16 + 1) * d.keep_stride);
    DALLOC(c, d.bucket_bins, ns + 4);
    CK(c, hipMemsetAsync(d.bucket_bins, 0, sizeof(int) * (ns + 4), c->stream));
    DALLOC(c, d.bucket_end, 2 * (ns + 4));
    CK(c, hipHostMalloc((void **)&c->ctrl_host, sizeof(Ctrl), hipHostMallocDefault));
    CK(c, hipHostMalloc((void **)&c->ctrl_pub, sizeof(Ctrl), hipHostMallocMapped));
    CK(c, hipHostGetDevicePointer((void **)&d.ctrl_pub, c->ctrl_pub, 0));
    d.publish = 0;

    d.X = dX; d.tab_lgam = dtl; d.tab_log = dtg; d.prior_m = dpm; d.prior_S = dpS;
    d.tabG = dtG; d.tabLogC = dtC; d.tabSeat = dtS;
    {   // mu_0 on the device (fixed-variance log marginal)
        double *dmu0;
        DALLOC(c, dmu0, (size_t)D);
        CK(c, hipMemcpyAsync(dmu0, m_0, sizeof(double) * D, hipMemcpyHostToDevice, c->stream));
        d.fv_mu0 = dmu0;
    }
    c->tabSeat = dtS;
    CK(c, hipMemcpyAsync(dX, X, sizeof(double) * N * D, hipMemcpyHostToDevice, c->stream));

    // tables: the reference's n = [1, 1, 2, ..., v_0+N+1] (gaussian_components.py:120-122)
    std::vector<double> tl(d.tab_len), tg(d.tab_len);
    for (long long t = 0; t < d.tab_len; ++t) {
        const double n = t == 0 ? 1.0 : (double)t;
        tl[t] = lgamma_tab ? lgamma_tab[t] : std::lgamma(n / 2.0);
        tg[t] = log_tab ? log_tab[t] : std::log(n);
    }
    CK(c, hipMemcpyAsync(dtl, tl.data(), sizeof(double) * d.tab_len, hipMemcpyHostToDevice, c->stream));
    CK(c, hipMemcpyAsync(dtg, tg.data(), sizeof(double) * d.tab_len, hipMemcpyHostToDevice, c->stream));

    // prior start of a fresh component (gaussian_components.py:161-164), rounded like numpy:
    // k_0*m_0  and  S_0 + k_0*outer(m_0, m_0)
    std::vector<double> pm(D), pS(DD);
    for (int a = 0; a < D; ++a) pm[a] = k_0 * m_0[a];
    if (fixed) {     // S_0 = [var ; var_0]: a new component starts at (precision_0 mu_0, precision_0)
        for (int a = 0; a < D; ++a) {
            const double p = 1.0 / S_0[a], p0 = 1.0 / S_0[D + a];
            pm[a] = p0 * m_0[a];
            pS[a] = p0;
            pS[D + a] = p;
        }
    } else if (diag) {      // S_0 is a D-vector: S_0 + k_0*square(m_0)   (gaussian_components_diag.py:170)
        for (int a = 0; a < D; ++a) {
            volatile double o = m_0[a] * m_0[a];
            volatile double ko = k_0 * o;
            pS[a] = S_0[a] + ko;
        }
    } else {
        for (int a = 0; a < D; ++a)
            for (int b = 0; b < D; ++b) {
                volatile double o = m_0[a] * m_0[b];
                volatile double ko = k_0 * o;
                pS[(size_t)a * D + b] = S_0[(size_t)a * D + b] + ko;
            }
    }
    CK(c, hipMemcpyAsync(dpm, pm.data(), sizeof(double) * D, hipMemcpyHostToDevice, c->stream));
    CK(c, hipMemcpyAsync(dpS, pS.data(), sizeof(double) * DD, hipMemcpyHostToDevice, c->stream));
    // pseudo slot K_max = the bare prior (n = 0): its refresh yields C = S_0, mu = m_0
    CK(c, hipMemsetAsync(d.n, 0, sizeof(int) * ns, c->stream));
    CK(c, hipMemsetAsync(d.nupd, 0, sizeof(int) * ns, c->stream));
    CK(c, hipMemcpyAsync(d.m + (size_t)K_max * D, pm.data(), sizeof(double) * D, hipMemcpyHostToDevice, c->stream));
    CK(c, hipMemcpyAsync(d.S + (size_t)K_max * DD, pS.data(), sizeof(double) * (fixed ? (size_t)D : DD), hipMemcpyHostToDevice, c->stream));
    if (fixed) CK(c, hipMemsetAsync(d.S + (size_t)K_max * DD + D, 0, sizeof(double) * D, c->stream));
    CK(c, hipMemsetAsync(d.z, 0xff, sizeof(int) * N, c->stream));
    CK(c, hipMemsetAsync(d.ctrl, 0, sizeof(Ctrl), c->stream));
    CK(c, hipStreamSynchronize(c->stream));   // host vectors go out of scope below

    Ctrl init;
    std::memset(&init, 0, sizeof(init));
    init.job.mode = MODE_DONE;
    init.first_mover = kNoMover;
    init.win_cap = c->win_rows;
    init.win_size = c->win_rows;
    init.ema_run = (double)c->win_rows * 4.0;
    init.last_mover = -1;
    init.safe_L = 4096;
    init.safe_cap_built = -1.0;
    init.safe_cap = 0.25;
    init.safe_mult = 4.0;
    CK(c, hipMemcpy(d.ctrl, &init, sizeof(Ctrl), hipMemcpyHostToDevice));
    std::vector<int> ident(ns);
    for (size_t i = 0; i < ns; ++i) ident[i] = (int)i;
    CK(c, hipMemcpy(d.perm, ident.data(), sizeof(int) * ns, hipMemcpyHostToDevice));
    CK(c, hipMemcpy(d.label_of_slot, ident.data(), sizeof(int) * ns, hipMemcpyHostToDevice));

    launch_build_tables(d, dtG, dtC, c->stream);
    launch_build_seat_table(d, dtS, c->stream);       // plain CRP weights log(n) until a sweep says otherwise
    c->seat_use_power = 0; c->seat_power = 1.0;
    // cached_log_prior: score every row against the pseudo slot, then the Student-t tail
    const int pslot = K_max;
    int *dslot;
    DALLOC(c, dslot, 1);
    CK(c, hipMemcpy(dslot, &pslot, sizeof(int), hipMemcpyHostToDevice));
    launch_refresh_list(d, dslot, 1, c->stream);
    Job job;
    std::memset(&job, 0, sizeof(job));
    job.pos = 0; job.win_base = 0; job.win_hi = N; job.mode = MODE_PARTIAL; job.K = 0;
    job.n_dirty = 1; job.dirty[0] = pslot; job.chunks = 1;
    CK(c, hipMemcpy(c->util_job, &job, sizeof(Job), hipMemcpyHostToDevice));
    double *qcol;
    DALLOC(c, qcol, (size_t)N);
    launch_score(d, c->kind, c->util_job, qcol, N, 0, N, 0, c->stream);
    launch_prior_lp(d, qcol, c->stream);
    CK(c, hipGetLastError());
    int rc = fetch_ctrl(c);
    if (rc) return rc;
    if (c->ctrl_host->error) return fail(c, BGMM_ENOTPD, "S_0 is not positive definite");
    return 0;
}

extern "C" int bgmm_create(bgmm_ctx **out, int device, int64_t N, int32_t D, int32_t K_max,
                           int32_t cov_type, const double *X, const double *m_0, double k_0,
                           int64_t v_0, const double *S_0, double alpha, const double *lgamma_tab,
                           const double *log_tab) {
    if (!out) return BGMM_EINVAL;
    *out = nullptr;
    if (cov_type != BGMM_COV_FULL && cov_type != BGMM_COV_DIAG && cov_type != BGMM_COV_FIXED)
        return fail(nullptr, BGMM_EUNSUPPORTED, "covariance_type must be full (0), diag (1) or fixed (2)");
    if (!X || !m_0 || !S_0 || N < 1 || D < 1 || K_max < 1) return fail(nullptr, BGMM_EINVAL, "bad shape or null pointer");
    if (cov_type == BGMM_COV_FULL && D > BGMM_MAX_D)
        return fail(nullptr, BGMM_EUNSUPPORTED, "full covariance supports D <= 128 (a component's D x D factor has to fit the LDS of a "
                                                "compute unit); covariance_type diag / fixed take D up to 4096");
    if (D > BGMM_MAX_D_DIAG) return fail(nullptr, BGMM_EUNSUPPORTED, "D > 4096 is not supported");
    if (N >= (1ll << 31) - 256) return fail(nullptr, BGMM_EUNSUPPORTED, "N must fit int32");
    if (v_0 < D && cov_type == BGMM_COV_FULL)
        return fail(nullptr, BGMM_EINVAL, "v_0 must be larger or equal to dimension of data");
    if (v_0 < 1) return fail(nullptr, BGMM_EINVAL, "v_0 must be positive");
    if (!(k_0 > 0) || !(alpha > 0)) return fail(nullptr, BGMM_EINVAL, "k_0 and alpha must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, BGMM_EDEVICE, "no HIP device visible: libbgmm_hip.so has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(nullptr, BGMM_EINVAL, "device index out of range");
    bgmm_ctx *c = new bgmm_ctx();
    const int rc = create_impl(c, device, N, D, K_max, cov_type, X, m_0, k_0, v_0, S_0, alpha, lgamma_tab, log_tab);
    if (rc != 0) {
        g_create_error = c->err;
        bgmm_destroy(c);
        return rc;
    }
    *out = c;
    return BGMM_OK;
}

extern "C" int bgmm_set_assignments(bgmm_ctx *c, const int64_t *z) {
    if (!c || !z) return BGMM_EINVAL;
    SETTLE(c);
    CK(c, hipSetDevice(c->device));
    const Dev &d = c->d;
    const long long N = d.N;
    long long zmax = -1;
    for (long long i = 0; i < N; ++i) {
        if (z[i] < -1) return fail(c, BGMM_EINVAL, "assignments must be -1 or >= 0");
        if (z[i] > zmax) zmax = z[i];
    }
    const int K = (int)(zmax + 1);
    if (K > d.K_max) return fail(c, BGMM_EINVAL, "initial assignments use more than K_max components");
    std::vector<long long> offsets(K + 1, 0);
    for (long long i = 0; i < N; ++i) if (z[i] >= 0) offsets[z[i] + 1] += 1;
    for (int k = 0; k < K; ++k) {
        if (offsets[k + 1] == 0) return fail(c, BGMM_EINVAL, "component labels must be consecutive from 0");
        offsets[k + 1] += offsets[k];
    }
    std::vector<int> members((size_t)(offsets[K] > 0 ? offsets[K] : 1));
    {
        std::vector<long long> cur(offsets.begin(), offsets.end() - 1);
        for (long long i = 0; i < N; ++i) if (z[i] >= 0) members[(size_t)cur[z[i]]++] = (int)i;
    }
    long long *dz = nullptr, *doff = nullptr;
    int *dmem = nullptr;
    hipError_t e1 = hipMalloc((void **)&dz, sizeof(long long) * N);
    hipError_t e2 = hipMalloc((void **)&doff, sizeof(long long) * (K + 1));
    hipError_t e3 = hipMalloc((void **)&dmem, sizeof(int) * members.size());
    if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) {
        if (e1 == hipSuccess) (void)hipFree(dz);
        if (e2 == hipSuccess) (void)hipFree(doff);
        if (e3 == hipSuccess) (void)hipFree(dmem);
        return fail(c, BGMM_EDEVICE, "hipMalloc failed");
    }
    int rc = 0;
    do {
        if (hipMemcpy(dz, z, sizeof(long long) * N, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(doff, offsets.data(), sizeof(long long) * (K + 1), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(dmem, members.data(), sizeof(int) * members.size(), hipMemcpyHostToDevice) != hipSuccess) {
            rc = fail(c, BGMM_EDEVICE, "hipMemcpy failed");
            break;
        }
        launch_init_labels(d, dz, K, c->stream);
        launch_init_stats(d, dmem, doff, K, c->stream);
        launch_refresh_list(d, nullptr, K, c->stream);
        if (hipGetLastError() != hipSuccess) { rc = fail(c, BGMM_EDEVICE, "kernel launch failed"); break; }
        rc = fetch_ctrl(c);
        if (rc) break;
        rc = check_device_error(c);
    } while (0);
    (void)hipFree(dz); (void)hipFree(doff); (void)hipFree(dmem);
    if (rc == 0) { c->assigned = true; c->moves_prev = -1; c->lean_ok = false; c->short_ok = false; }
    return rc;
}

// The sequential small-D sweep fetches z[i] ahead of the visit of i: sound only when no index
// comes twice.  Checked on the host for the shapes that can take that path.
static bool seq_shape(const bgmm_ctx *c) { return c->d.cov_type == COV_FULL && c->d.D <= 4; }
// 1: a permutation of 0 .. N-1; 0: in range but with repeats (the C-ABI takes any index array: the
// kernels that fetch a visit's home ahead of time are not used); -1: an index out of range.
static int classify_order(const int64_t *order, long long N) {
    std::vector<unsigned char> seen((size_t)N, 0);
    int kind = 1;
    for (long long p = 0; p < N; ++p) {
        const int64_t i = order[p];
        if (i < 0 || i >= N) return -1;
        if (seen[(size_t)i]) kind = 0;
        seen[(size_t)i] = 1;
    }
    return kind;
}

extern "C" int bgmm_stage_sweep_inputs(bgmm_ctx *c, const int64_t *order, const double *u) {
    if (!c || !u) return BGMM_EINVAL;
    // (nothing is touched before the order has been found valid)
    const int okind = order ? classify_order(order, c->d.N) : 1;
    if (okind < 0) return fail(c, BGMM_EINVAL, "visiting order holds an index outside 0 .. N-1");
    SETTLE(c);                                          // (this call overwrites the buffers a sweep in flight reads)
    CK(c, hipSetDevice(c->device));
    CK(c, hipMemcpyAsync(c->d_u, u, sizeof(double) * c->d.N, hipMemcpyHostToDevice, c->stream));
    c->cur_zero_u = false;
    for (long long i = 0; i < c->d.N; ++i)
        if (u[i] == 0.0) { c->cur_zero_u = true; break; }
    const bool keep = !order && c->order_staged;        // (bgmm_stage_permutation_mt19937 has put this sweep's order in place)
    c->have_order = order != nullptr || keep;
    c->order_is_perm = okind == 1;
    if (order) {
        CK(c, hipMemcpyAsync(c->d_order, order, sizeof(long long) * c->d.N, hipMemcpyHostToDevice, c->stream));
        c->order_staged = false;
    }
    CK(c, hipStreamSynchronize(c->stream));
    c->cur_u = c->d_u;
    c->cur_order = (order || keep) ? c->d_order : nullptr;
    return 0;
}

// jump polynomials / chain seeds for requests of up to `chains` chains (grown on demand, never while a generation runs)
static int perm_pipe_drain(bgmm_ctx *c);
static int mt_ensure_tables(bgmm_ctx *c, int chains) {
    if (chains < 2 || c->mt_chains >= chains) return 0;
    // (the permutations in flight read the coefficient table that is about to be replaced)
    { const int rc = perm_pipe_drain(c); if (rc) return rc; }
    std::vector<unsigned> coef;
    const bool have = mt19937_jump_coefficients(chains, coef);
    if (c->mt_coef) { (void)hipFree(c->mt_coef); c->mt_coef = nullptr; }
    if (c->mt_seeds) { (void)hipFree(c->mt_seeds); c->mt_seeds = nullptr; }
    CK(c, hipMalloc((void **)&c->mt_seeds, sizeof(unsigned) * 624 * (size_t)(chains + 1)));
    if (have) {
        CK(c, hipMalloc((void **)&c->mt_coef, sizeof(unsigned) * coef.size()));
        CK(c, hipMemcpy(c->mt_coef, coef.data(), sizeof(unsigned) * coef.size(), hipMemcpyHostToDevice));
    }
    c->mt_chains = chains;
    return 0;
}

static int mt_depth_for(long long N) {
    if (N < 4096) return 1;                       // (sweep boundaries must lie behind the request's first block)
    long long m = 4000000 / N;
    if (m < 1) m = 1;
    if (m > kMtMaxMids) m = kMtMaxMids;
    return (int)m;
}

// Starts the generation of batch `bi`: `depth` sweeps' uniforms from the generator state (key, pos), on the second stream.
static int mt_launch_batch(bgmm_ctx *c, int bi, const uint32_t *key, int pos) {
    const size_t N = (size_t)c->d.N, raw_n = (size_t)mt19937_raw_words();
    const int M = c->mt_depth;
    bgmm_ctx::MtBatch &B = c->mt_b[bi];
    const size_t host_words = 624 + (size_t)M * 624 + 2 * (size_t)M + 16;
    if (!c->mt_stream) CK(c, hipStreamCreateWithFlags(&c->mt_stream, hipStreamNonBlocking));
    if (!B.done) CK(c, hipEventCreateWithFlags(&B.done, hipEventDisableTiming));
    if (!B.u) CK(c, hipMalloc((void **)&B.u, sizeof(double) * N * (size_t)M + 64));
    if (!B.host) CK(c, hipHostMalloc((void **)&B.host, sizeof(unsigned) * host_words, hipHostMallocDefault));
    // device scratch: [key in 624 | state behind sweep j: M x 624 | positions M | zero flags M | pad to 16 | spare key 624 |
    //                  spare position 16 | raw | 2 M N tempered words]
    const size_t head = 624 + (size_t)M * 624 + 2 * (size_t)M + 16;
    if (!c->mt_words_ahead)
        CK(c, hipMalloc((void **)&c->mt_words_ahead, sizeof(unsigned) * (head + 640 + raw_n + 2 * N * (size_t)M)));
    unsigned *dkey = c->mt_words_ahead, *dmid = dkey + 624;
    int *dposmid = (int *)(dmid + (size_t)M * 624), *dflags = dposmid + M;
    unsigned *dspare = c->mt_words_ahead + head, *draw = dspare + 640, *dwords = draw + raw_n;
    int *dspare_pos = (int *)(dspare + 624);
    memcpy(B.host, key, sizeof(unsigned) * 624);
    memset(B.host + 624 + (size_t)M * 624, 0, sizeof(unsigned) * 2 * (size_t)M);
    B.pos_in = pos;
    hipStream_t as = c->mt_stream;
    CK(c, hipMemcpyAsync(dkey, B.host, sizeof(unsigned) * 624, hipMemcpyHostToDevice, as));
    CK(c, hipMemsetAsync(dposmid, 0, sizeof(int) * 2 * (size_t)M, as));
    const int chains = mt19937_chains_for(pos, (long long)(N * (size_t)M));
    launch_mt19937(dkey, pos, M == 1 ? dmid : dspare, M == 1 ? dposmid : dspare_pos, dwords, B.u, (long long)(N * (size_t)M), dflags,
                   (c->mt_jump_on && chains >= 2) ? c->mt_coef : nullptr, chains, draw, c->mt_seeds, as,
                   M, M > 1 ? dmid : nullptr, M > 1 ? dposmid : nullptr);
    CK(c, hipGetLastError());
    CK(c, hipMemcpyAsync(B.host + 624, dmid, sizeof(unsigned) * ((size_t)M * 624 + 2 * (size_t)M), hipMemcpyDeviceToHost, as));
    CK(c, hipEventRecord(B.done, as));
    B.launched = true; B.synced = false; B.next = 0;
    return 0;
}

static int mt_wait_batches(bgmm_ctx *c) {
    for (auto &b : c->mt_b)
        if (b.launched && !b.synced) { CK(c, hipEventSynchronize(b.done)); b.synced = true; }
    if (c->perm_ahead_valid) CK(c, hipEventSynchronize(c->perm_done));     // (the permutation's look-ahead reads the same tables)
    if (c->pp.built) {
        // (the words' chunks read the jump tables: nothing new from the worker, what it is at finished)
        {
            std::unique_lock<std::mutex> lk(c->pp.mu);
            c->pp.target = c->pp.gen_queued;
            c->pp.cv.wait(lk, [&] { return !c->pp.busy; });
            c->pp.target = c->pp.gen_queued;
        }
        CK(c, hipStreamSynchronize(c->pp.rawst));
    }
    return 0;
}

// What comes behind a served request: after one generated on the spot, a fresh batch from the state just handed back
// (key, pos); towards the end of a batch, the batch behind it (its generation runs beside the sweeps queued meanwhile).
static int mt_schedule(bgmm_ctx *c, bool hit, const uint32_t *key, int pos) {
    if (!c->mt_ahead_on) return 0;
    if (!hit) {
        int rc = mt_launch_batch(c, 0, key, pos);
        if (rc) return rc;
        c->mt_cur = 0;
        return 0;
    }
    bgmm_ctx::MtBatch *B = c->mt_cur >= 0 ? &c->mt_b[c->mt_cur] : nullptr;
    if (B && B->launched && B->next >= std::max(1, c->mt_depth - 2) && !c->mt_b[c->mt_cur ^ 1].launched) {
        // (the state behind this batch is known since its generation finished: the next batch is started two sweeps
        // before it is needed -- under a running sweep a generation takes about two of them)
        const int Md = c->mt_depth;
        return mt_launch_batch(c, c->mt_cur ^ 1, B->host + 624 + (size_t)(Md - 1) * 624,
                               (int)B->host[624 + (size_t)Md * 624 + (size_t)(Md - 1)]);
    }
    return 0;
}

extern "C" int bgmm_stage_mt19937(bgmm_ctx *c, const int64_t *order, uint32_t *key624, int32_t *pos) {
    if (!c || !key624 || !pos) return BGMM_EINVAL;
    if (*pos < 0 || *pos > 624) return fail(c, BGMM_EINVAL, "MT19937 position must be in 0 .. 624");
    // (the order is validated before any generator or look-ahead state is touched)
    const int okind = order ? classify_order(order, c->d.N) : 1;
    if (okind < 0) return fail(c, BGMM_EINVAL, "visiting order holds an index outside 0 .. N-1");
    if (order) SETTLE(c);                               // (an explicit order is copied into the buffer a sweep in flight may read)
    CK(c, hipSetDevice(c->device));
    const size_t N = (size_t)c->d.N;
    // device scratch of a request served on the spot: [key in 624 | key out 624 | pos out, zero flag, pad 16 | raw | 2 N
    // tempered words], and -- requests of more than one chain -- the jump polynomials' coefficient words and the chains'
    // seeds.  The seeds are shared with the look-ahead: one generation at a time.
    const size_t raw_n = (size_t)mt19937_raw_words();
    if (!c->mt_words) CK(c, hipMalloc((void **)&c->mt_words, sizeof(unsigned) * (1264 + raw_n + 2 * N)));
    unsigned *dkey = c->mt_words, *dkey_out = c->mt_words + 624, *draw = c->mt_words + 1264, *dwords = draw + raw_n;
    int *dpos = (int *)(c->mt_words + 1248), *dflag = (int *)(c->mt_words + 1249);
    if (c->mt_depth == 0) c->mt_depth = mt_depth_for((long long)N);
    const int M = c->mt_ahead_on ? c->mt_depth : 1;
    {
        const int chains_max = mt19937_chains_for(624, (long long)(N * (size_t)M));
        if (chains_max > c->mt_chains) {
            int rc = mt_wait_batches(c);
            if (rc) return rc;
            rc = mt_ensure_tables(c, chains_max);
            if (rc) return rc;
        }
    }
    // a batch that has been served to its end: the one started while its last sweep was being served takes over
    if (c->mt_cur >= 0 && c->mt_b[c->mt_cur].launched && c->mt_b[c->mt_cur].next >= c->mt_depth) {
        c->mt_b[c->mt_cur].launched = false;
        c->mt_cur ^= 1;
    }
    bool hit = false;
    bgmm_ctx::MtBatch *B = c->mt_cur >= 0 ? &c->mt_b[c->mt_cur] : nullptr;
    if (B && B->launched && c->mt_ahead_on) {
        if (!B->synced) { CK(c, hipEventSynchronize(B->done)); B->synced = true; }
        const int Md = c->mt_depth, j = B->next;
        const unsigned *exp_key = j == 0 ? B->host : B->host + 624 + (size_t)(j - 1) * 624;
        const int exp_pos = j == 0 ? B->pos_in : (int)B->host[624 + (size_t)Md * 624 + (size_t)(j - 1)];
        hit = *pos == exp_pos && memcmp(key624, exp_key, sizeof(unsigned) * 624) == 0;
        if (hit) {
            c->cur_u = B->u + (size_t)j * N;
            memcpy(key624, B->host + 624 + (size_t)j * 624, sizeof(unsigned) * 624);
            *pos = (int32_t)B->host[624 + (size_t)Md * 624 + (size_t)j];
            c->cur_zero_u = B->host[624 + (size_t)Md * 624 + (size_t)Md + (size_t)j] != 0;
            B->next = j + 1;
            c->mt_ahead_hits += 1;
        }
    }
    if (!hit) {
        // not foreseen (the first request, or the caller drew from its generator in between): generated on the spot, and
        // whatever the look-ahead holds is of no use any more.  (A sweep in flight reads d_u, or a batch buffer: finished first.)
        SETTLE(c);
        int rc = mt_wait_batches(c);
        if (rc) return rc;
        c->mt_b[0].launched = c->mt_b[1].launched = false;
        c->mt_cur = -1;
        int host_tail[2] = {0, 0};
        CK(c, hipMemcpyAsync(dkey, key624, sizeof(unsigned) * 624, hipMemcpyHostToDevice, c->stream));
        CK(c, hipMemcpyAsync(dpos, host_tail, sizeof(int) * 2, hipMemcpyHostToDevice, c->stream));
        const int chains = mt19937_chains_for(*pos, (long long)N);
        launch_mt19937(dkey, *pos, dkey_out, dpos, dwords, c->d_u, (long long)N, dflag,
                       (c->mt_jump_on && chains >= 2) ? c->mt_coef : nullptr, chains, draw, c->mt_seeds, c->stream);
        CK(c, hipGetLastError());
        CK(c, hipMemcpyAsync(key624, dkey_out, sizeof(unsigned) * 624, hipMemcpyDeviceToHost, c->stream));
        CK(c, hipMemcpyAsync(host_tail, dpos, sizeof(int) * 2, hipMemcpyDeviceToHost, c->stream));
        CK(c, hipStreamSynchronize(c->stream));
        *pos = host_tail[0];
        c->cur_zero_u = host_tail[1] != 0;
        c->cur_u = c->d_u;
        c->mt_ahead_misses += 1;
    }
    const bool keep = !order && c->order_staged;        // (bgmm_stage_permutation_mt19937 has put this sweep's order in place)
    c->have_order = order != nullptr || keep;
    c->order_is_perm = okind == 1;
    if (order) {
        CK(c, hipMemcpyAsync(c->d_order, order, sizeof(long long) * N, hipMemcpyHostToDevice, c->stream));
        CK(c, hipStreamSynchronize(c->stream));
        c->order_staged = false;
    }
    c->cur_order = (order || keep) ? c->d_order : nullptr;
    if (c->async_pending) {
        // (a sweep is in flight -- bgmm_sweep_staged_begin: it may still be reading the buffer the next generation would
        // write into; bgmm_sweep_staged_end starts it)
        c->defer_mt = true; c->defer_mt_hit = hit; c->defer_mt_pos = *pos;
        c->defer_mt_key.assign(key624, key624 + 624);
        return 0;
    }
    return mt_schedule(c, hit, key624, *pos);
}

// np.random.permutation(N) from the caller's legacy numpy generator, on the device (kernels_perm.hip).

static int perm_ensure(bgmm_ctx *c, PermPtrs &P) {
    const long long N = c->d.N;
    const size_t raw_n = (size_t)mt19937_raw_words();
    // words: rejection sampling takes 1.39 words per step on average (at most 2 while the mask's range is nearly all
    // rejected): 2 N and a block to spare, rounded so that the request ends on a block boundary whatever pos is
    P.n_words_cap = 624 * ((624 + 2 * N + 1248 + 623) / 624);
    const size_t head = 624 + 624 + 16 + 624 + 16;
    if (!c->perm_words) {
        CK(c, hipMalloc((void **)&c->perm_words, sizeof(unsigned) * (head + raw_n + (size_t)P.n_words_cap)));
        c->perm_chains = mt19937_chains_for_words(624, P.n_words_cap);
        CK(c, hipMalloc((void **)&c->perm_seeds, sizeof(unsigned) * 624 * (size_t)(c->perm_chains + 2)));
        CK(c, hipMalloc((void **)&c->perm_ints, sizeof(int) * (3 * (size_t)N + 64 + 5 * (size_t)perm_segments(P.n_words_cap))));
        // (targets left over from an earlier permutation are at least valid indices: when a generation's draws have not
        // settled, the kernels behind them run on whatever J holds before the repair queues them again)
        CK(c, hipMemset(c->perm_ints, 0, sizeof(int) * (3 * (size_t)N + 64 + 5 * (size_t)perm_segments(P.n_words_cap))));
        CK(c, hipMalloc((void **)&c->perm_uints, sizeof(unsigned) * (3 * (size_t)N + 16)));
        c->perm_temp_bytes = perm_sort_temp_bytes((int)N);
        CK(c, hipMalloc(&c->perm_temp, c->perm_temp_bytes + 256));
        CK(c, hipMalloc((void **)&c->perm_out, sizeof(long long) * 4));
        CK(c, hipHostMalloc((void **)&c->perm_host, sizeof(unsigned) * (1344 + (size_t)perm_segments(P.n_words_cap)), hipHostMallocDefault));
        { int rc = dalloc(c, &c->d_order_ahead, (size_t)N); if (rc) return rc; }      // (freed with the context's other buffers)
        CK(c, hipStreamCreateWithFlags(&c->perm_stream, hipStreamNonBlocking));
        CK(c, hipEventCreateWithFlags(&c->perm_done, hipEventDisableTiming));
        launch_perm_iota((int)N, c->perm_uints + 2 * (size_t)N, c->stream);
        CK(c, hipStreamSynchronize(c->stream));
    }
    if (c->perm_chains >= 2 && c->perm_chains > c->mt_chains) {       // (mt_ensure_tables has nothing to do for one chain)
        int rc = mt_wait_batches(c);
        if (rc) return rc;
        rc = mt_ensure_tables(c, c->perm_chains);
        if (rc) return rc;
    }
    P.dkey = c->perm_words; P.dkey_out = P.dkey + 624; P.dspare = P.dkey + 1264;
    P.draw = c->perm_words + head; P.dwords = P.draw + raw_n;
    P.dpos_out = (int *)(P.dkey + 1248); P.dspare_pos = (int *)(P.dspare + 624);
    P.J = c->perm_ints; P.pred = P.J + N; P.ptr = P.pred + N; P.changed = P.ptr + N; P.flags = P.changed + 4; P.cnt = P.changed + 64;
    P.ks = c->perm_uints; P.idx = P.ks + N; P.iota = P.idx + N;
    return 0;
}

// queues the whole generation on `st`: words, draws, swaps, the state behind them, and the copies of the verdicts into
// perm_host [key out 624 | pos out | pointer jumping still moved | the write pass ran | - | out (2 x 64 bit)]
static int perm_queue(bgmm_ctx *c, const PermPtrs &P, const unsigned *key_pinned, int pos, long long *order_dst, hipStream_t st) {
    const long long N = c->d.N;
    const long long n_words = 624 * (((long long)pos + 2 * N + 1248 + 623) / 624) - (long long)pos;
    c->perm_n_words = n_words;
    CK(c, hipMemcpyAsync(P.dkey, key_pinned, sizeof(unsigned) * 624, hipMemcpyHostToDevice, st));
    const int chains = mt19937_chains_for_words(pos, n_words);
    launch_mt19937_raw(P.dkey, pos, P.dwords, n_words, (c->mt_jump_on && chains >= 2) ? c->mt_coef : nullptr, chains, P.draw, c->perm_seeds,
                       P.dspare, P.dspare_pos, st);
    if (!launch_permutation(P.dwords, n_words, (int)N, P.dkey, pos, P.J, P.pred, P.ptr, P.cnt, (int *)(c->perm_host + 1344), P.flags, P.ks,
                            P.idx, P.iota, c->perm_temp, c->perm_temp_bytes, c->perm_out, P.changed, order_dst, P.dkey_out, P.dpos_out, st))
        return fail(c, BGMM_EDEVICE, "permutation kernels failed to launch");
    CK(c, hipGetLastError());
    return 0;
}

static int perm_queue_verdicts(bgmm_ctx *c, const PermPtrs &P, hipStream_t st) {
    unsigned *H = c->perm_host;
    CK(c, hipMemcpyAsync(H, P.dkey_out, sizeof(unsigned) * 624, hipMemcpyDeviceToHost, st));
    CK(c, hipMemcpyAsync(H + 624, P.dpos_out, sizeof(int), hipMemcpyDeviceToHost, st));
    CK(c, hipMemcpyAsync(H + 625, P.changed, sizeof(int), hipMemcpyDeviceToHost, st));
    CK(c, hipMemcpyAsync(H + 626, P.flags + perm_rounds() + 1, sizeof(int), hipMemcpyDeviceToHost, st));
    CK(c, hipMemcpyAsync(H + 628, c->perm_out, sizeof(long long) * 2, hipMemcpyDeviceToHost, st));
    CK(c, hipMemcpyAsync(H + 1280, P.flags, sizeof(int) * (size_t)(perm_rounds() + 2), hipMemcpyDeviceToHost, st));   // (statistics)
    return 0;
}

// after the stream has drained: the rare repairs (draws not settled within the queued rounds, chains of swaps longer than
// the queued rounds of pointer jumping), then the state numpy would be left in
static int perm_finish(bgmm_ctx *c, const PermPtrs &P, int pos_in, long long *order_dst, hipStream_t st, uint32_t *key624, int32_t *pos) {
    const long long N = c->d.N;
    unsigned *H = c->perm_host;
    for (int tries = 0;; ++tries) {
        if (tries > 64) return fail(c, BGMM_EDEVICE, "the permutation's draws did not settle");
        if (H[626] == 0) {               // the draws had not settled (no write pass yet): more rounds, then the rest again
            launch_permutation_draw_more(P.dwords, c->perm_n_words, (int)N, P.J, P.cnt, P.flags, c->perm_out, st);
            if (!launch_permutation_tail(P.dwords, (int)N, P.dkey, pos_in, P.J, P.pred, P.ptr, P.ks, P.idx, P.iota, c->perm_temp,
                                         c->perm_temp_bytes, c->perm_out, P.changed, order_dst, P.dkey_out, P.dpos_out, st))
                return fail(c, BGMM_EDEVICE, "permutation kernels failed to launch");
        } else {
            long long out[2];
            memcpy(out, H + 628, sizeof(out));
            if (out[1] != 0) return fail(c, BGMM_EUNSUPPORTED, "the permutation ran out of random words (draw it on the host)");
            if (H[625] == 0) break;
            launch_permutation_more((int)N, P.J, P.pred, P.ptr, P.changed, order_dst, st);    // (chains of swaps longer than 128 links)
        }
        int rc = perm_queue_verdicts(c, P, st);
        if (rc) return rc;
        CK(c, hipStreamSynchronize(st));
    }
    memcpy(key624, H, sizeof(unsigned) * 624);
    *pos = (int32_t)H[624];
    {   // the round of draws in which no count changed any more (statistics only)
        int r = 1;
        while (r <= perm_rounds() && H[1280 + r] != 0) ++r;
        c->perm_last_rounds = r;
        if (r > c->perm_max_rounds) c->perm_max_rounds = r;
    }
    return 0;
}

extern "C" int bgmm_get_permutation_stats(bgmm_ctx *c, int64_t *out4) {
    if (!c || !out4) return BGMM_EINVAL;
    out4[0] = c->perm_hits; out4[1] = c->perm_misses; out4[2] = c->perm_last_rounds; out4[3] = c->perm_max_rounds;
    return 0;
}

extern "C" int bgmm_get_permutation_pipe_state(bgmm_ctx *c, int64_t *out4) {
    if (!c || !out4) return BGMM_EINVAL;
    bgmm_ctx::PermPipe &Q = c->pp;
    std::lock_guard<std::mutex> g(Q.mu);
    out4[0] = Q.built ? 1 : 0; out4[1] = Q.off ? 1 : 0; out4[2] = Q.wfails;
    out4[3] = Q.built ? (int64_t)Q.era_cap * (int64_t)sizeof(unsigned) : 0;
    return 0;
}

// queues the look-ahead permutation from the state noted in perm_host[640 ..) / perm_ahead_pos_in
static int perm_schedule(bgmm_ctx *c, const PermPtrs &P) {
    int rc = perm_queue(c, P, c->perm_host + 640, c->perm_ahead_pos_in, c->d_order_ahead, c->perm_stream);
    if (rc == 0) rc = perm_queue_verdicts(c, P, c->perm_stream);
    if (rc) return rc;
    CK(c, hipEventRecord(c->perm_done, c->perm_stream));
    c->perm_ahead_valid = true;
    return 0;
}

// ---- permutations in flight ---------------------------------------------------------------------------------------------
// One generation is a chain of ~0.64 ms at N = 1e6 (words 100 us, 30 rounds of draws 250, the serial tail 84, the swaps 190,
// verdicts) in front of a pCRP sweep of 0.23 ms -- and the next generation needs only TWO things from it: where its words
// end, and that they were generated.  So (BGMM_PERM_PIPE=0: the single look-ahead of round 3):
//   * the words are one long stream (an "era": era_raw[k] = the k-th output behind the state the era began at), generated in
//     chunks on their own stream far ahead of the draws -- a chunk continues from the last block of the one before it, which
//     IS the generator's state there;
//   * a generation reads its words at the offset the generation in front of it leaves on the device (PermPipe::goffs) and
//     leaves its own end there: the draws of kAhead generations are queued back to back on one stream, no host in between;
//   * the swaps of generation g (sort by target, links, assembly) and its verdicts run on a third stream beside the draws
//     of generation g + 1.
// The host sees a generation again when it is handed out: verdicts (settled, words left, the state numpy would be in), the
// caller's state compared with the state the last call handed back -- anything else (a caller that drew from the stream in
// between, draws that did not settle in the queued rounds) drains the three streams and goes the old way, on the spot.
static void perm_pipe_worker(bgmm_ctx *c);

static bool perm_pipe_wanted() {
    static const bool on = [] { const char *e = getenv("BGMM_PERM_PIPE"); return !(e && atoi(e) == 0); }();
    return on;
}

static int perm_pipe_drain(bgmm_ctx *c) {
    bgmm_ctx::PermPipe &Q = c->pp;
    if (!Q.built) return 0;
    {
        std::unique_lock<std::mutex> lk(Q.mu);
        Q.target = Q.gen_queued;         // (what the worker has not begun stays unqueued)
        Q.cv.wait(lk, [&] { return !Q.busy; });
        Q.target = Q.gen_queued;
    }
    CK(c, hipStreamSynchronize(Q.rawst));
    CK(c, hipStreamSynchronize(c->perm_stream));
    CK(c, hipStreamSynchronize(Q.fin));
    std::lock_guard<std::mutex> g(Q.mu);
    Q.valid = false;
    Q.full = false;
    if (Q.wrc != 0) {
        // the worker could not queue a generation: the stage call goes the old way for this one -- and after three in a row
        // for good (a device call that keeps failing would otherwise be queued and dropped at every stage call)
        if (++Q.wfails >= 3 && !Q.off) {
            Q.off = true;
            Q.off_why = Q.werr;
        }
        Q.wrc = 0;
    }
    Q.gen_next = Q.gen_queued;          // (whatever was in flight is dropped)
    return 0;
}

// device memory of the pipe: on a list of its own (released as a whole when set-up fails, freed with the context otherwise)
template <typename T>
static int pp_alloc(bgmm_ctx *c, T **p, size_t count) {
    void *q = nullptr;
    const hipError_t e = hipMalloc(&q, count * sizeof(T) + 64);
    if (e != hipSuccess) {
        err_of(c) = std::string("hipMalloc (permutations in flight): ") + hipGetErrorString(e);
        return BGMM_EDEVICE;
    }
    c->pp.dev_allocs.push_back(q);
    *p = (T *)q;
    return 0;
}

static void perm_pipe_release(bgmm_ctx *c) {
    bgmm_ctx::PermPipe &Q = c->pp;
    constexpr int A = bgmm_ctx::PermPipe::kAhead;
    if (Q.fin) { (void)hipStreamSynchronize(Q.fin); (void)hipStreamDestroy(Q.fin); Q.fin = nullptr; }
    if (Q.rawst) { (void)hipStreamSynchronize(Q.rawst); (void)hipStreamDestroy(Q.rawst); Q.rawst = nullptr; }
    for (int k = 0; k < A; ++k) {
        if (Q.ev_draw[k]) { (void)hipEventDestroy(Q.ev_draw[k]); Q.ev_draw[k] = nullptr; }
        if (Q.ev_fin[k]) { (void)hipEventDestroy(Q.ev_fin[k]); Q.ev_fin[k] = nullptr; }
        if (Q.host[k]) { (void)hipHostFree(Q.host[k]); Q.host[k] = nullptr; }
        Q.J[k] = nullptr; Q.vblk[k] = nullptr; Q.ord[k] = nullptr;
    }
    if (Q.ev_raw) { (void)hipEventDestroy(Q.ev_raw); Q.ev_raw = nullptr; }
    if (Q.ev_sweep) { (void)hipEventDestroy(Q.ev_sweep); Q.ev_sweep = nullptr; }
    if (Q.era_raw) { (void)hipFree(Q.era_raw); Q.era_raw = nullptr; }
    if (Q.era_key_host) { (void)hipHostFree(Q.era_key_host); Q.era_key_host = nullptr; }
    for (void *q : Q.dev_allocs) (void)hipFree(q);
    Q.dev_allocs.clear();
    Q.era_key = nullptr; Q.goffs = nullptr; Q.cnt = nullptr; Q.pre0 = nullptr; Q.zero[0] = Q.zero[1] = nullptr;
    Q.parked = nullptr; Q.bnd = nullptr; Q.cursor = nullptr; Q.slots = nullptr;
    Q.NB = 0;
}

static int perm_pipe_build(bgmm_ctx *c, const PermPtrs &P) {
    bgmm_ctx::PermPipe &Q = c->pp;
    Q.P = P;                            // (the worker's copy: set before it exists, never written again)
    constexpr int A = bgmm_ctx::PermPipe::kAhead;
    const long long N = c->d.N;
    Q.cap_words = 2 * N + 1248;
    const int T = perm_segments(Q.cap_words);
    // the era: 32 generations' worth of words, within 1 GiB AND within a twentieth of the memory that is free now (32 chains
    // side by side at N = 1e6 would otherwise take 8 GB for look-ahead alone), never less than what kAhead + 2 generations
    // may read (BGMM_PERM_ERA: generations' worth, for the test that walks through several eras)
    static const int era_gens = [] { const char *e = getenv("BGMM_PERM_ERA"); const int v = e ? atoi(e) : 32; return v < 1 ? 1 : v; }();
    long long cap = era_gens * Q.cap_words;
    if (cap > (1ll << 28)) cap = 1ll << 28;
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const long long share = (long long)(free_b / 20 / sizeof(unsigned));
            if (cap > share) cap = share;
        } else {
            (void)hipGetLastError();
        }
    }
    if (cap < (A + 3) * Q.cap_words) cap = (A + 3) * Q.cap_words;
    Q.era_cap = 624 * ((cap + 623) / 624) + 1248;
    CK(c, hipMalloc((void **)&Q.era_raw, sizeof(unsigned) * (size_t)Q.era_cap));
    CK(c, hipHostMalloc((void **)&Q.era_key_host, sizeof(unsigned) * 640, hipHostMallocDefault));
    { int rc = pp_alloc(c, &Q.era_key, (size_t)640); if (rc) return rc; }
    { int rc = pp_alloc(c, &Q.goffs, (size_t)8); if (rc) return rc; }
    { int rc = pp_alloc(c, &Q.cnt, 5 * (size_t)T + 16); if (rc) return rc; }
    CK(c, hipMemset(Q.cnt, 0, sizeof(int) * 5 * (size_t)T));
    std::vector<int> pre((size_t)T + 1);
    Q.nblk_pad = perm_chain_guess(Q.cap_words, (int)N, pre.data());
    if (Q.rounds_fixed) Q.rounds_q = Q.rounds_fixed;
    { int rc = pp_alloc(c, &Q.pre0, (size_t)T + 16); if (rc) return rc; }
    CK(c, hipMemcpy(Q.pre0, pre.data(), sizeof(int) * ((size_t)T + 1), hipMemcpyHostToDevice));
    for (int k = 0; k < 2; ++k) {
        { int rc = pp_alloc(c, &Q.zero[k], (size_t)Q.nblk_pad + 64); if (rc) return rc; }
        CK(c, hipMemset(Q.zero[k], 0, sizeof(int) * ((size_t)Q.nblk_pad + 64)));
    }
    for (int k = 0; k < A; ++k) {
        { int rc = pp_alloc(c, &Q.J[k], (size_t)N + 16); if (rc) return rc; }
        CK(c, hipMemset(Q.J[k], 0, sizeof(int) * (size_t)N));
        { int rc = pp_alloc(c, &Q.vblk[k], (size_t)1344); if (rc) return rc; }
        CK(c, hipMemset(Q.vblk[k], 0, sizeof(int) * 1344));
        { int rc = pp_alloc(c, &Q.ord[k], (size_t)N); if (rc) return rc; }
        CK(c, hipHostMalloc((void **)&Q.host[k], sizeof(unsigned) * 1344, hipHostMallocDefault));
        memset(Q.host[k], 0, sizeof(unsigned) * 1344);
        const unsigned evf = getenv("BGMM_DEBUG_PERM") ? hipEventDefault : hipEventDisableTiming;
        CK(c, hipEventCreateWithFlags(&Q.ev_draw[k], evf));
        CK(c, hipEventCreateWithFlags(&Q.ev_fin[k], evf));
    }
    { int rc = pp_alloc(c, &Q.parked, (size_t)N); if (rc) return rc; }
    {
        std::vector<int> bnd;
        Q.NB = perm_bucket_bounds((int)N, bnd);
        if (Q.NB > 0) {
            { int rc = pp_alloc(c, &Q.bnd, (size_t)Q.NB + 16); if (rc) return rc; }
            { int rc = pp_alloc(c, &Q.cursor, (size_t)Q.NB + 16); if (rc) return rc; }
            { int rc = pp_alloc(c, &Q.slots, (size_t)Q.NB * (size_t)perm_bucket_cap()); if (rc) return rc; }
            CK(c, hipMemcpy(Q.bnd, bnd.data(), sizeof(int) * ((size_t)Q.NB + 1), hipMemcpyHostToDevice));
            CK(c, hipMemset(Q.cursor, 0, sizeof(int) * (size_t)Q.NB));
        }
    }
    CK(c, hipEventCreateWithFlags(&Q.ev_raw, hipEventDisableTiming));
    CK(c, hipEventCreateWithFlags(&Q.ev_sweep, hipEventDisableTiming));
    CK(c, hipStreamCreateWithFlags(&Q.fin, hipStreamNonBlocking));
    CK(c, hipStreamCreateWithFlags(&Q.rawst, hipStreamNonBlocking));
    try {
        Q.worker = std::thread(perm_pipe_worker, c);
    } catch (...) {
        return fail(c, BGMM_EDEVICE, "could not start the permutations' worker thread");
    }
    return 0;
}

// 0: the pipe stands; 1: it does not and will not (set-up failed -- as a rule: memory --, everything it had taken is
// released, PermPipe::off is latched): the caller goes on with the single look-ahead, no error.
// BGMM_PERM_PIPE_FAIL=1 makes the set-up fail after its allocations (the test of this path).
static int perm_pipe_ensure(bgmm_ctx *c, const PermPtrs &P) {
    bgmm_ctx::PermPipe &Q = c->pp;
    if (Q.built) return 0;
    if (Q.off) return 1;
    std::string why;
    g_err_sink = &why;                  // (a failed set-up is not the caller's error: bgmm_ctx::err stays what it was)
    int rc = perm_pipe_build(c, P);
    if (rc == 0 && getenv("BGMM_PERM_PIPE_FAIL")) {
        { std::lock_guard<std::mutex> g(Q.mu); Q.quit = true; }
        Q.cv.notify_all();
        if (Q.worker.joinable()) Q.worker.join();
        Q.quit = false;
        why = "BGMM_PERM_PIPE_FAIL";
        rc = BGMM_EDEVICE;
    }
    g_err_sink = nullptr;
    if (rc == 0) { Q.built = true; return 0; }
    (void)hipGetLastError();
    perm_pipe_release(c);
    Q.off = true;
    Q.off_why = why;
    return 1;
}

// a new era from a state the host knows (all three streams idle)
static int perm_pipe_start_era(bgmm_ctx *c, const uint32_t *key624, int pos) {
    bgmm_ctx::PermPipe &Q = c->pp;
    std::lock_guard<std::mutex> guard(Q.mu);        // (the worker is idle: drained, or never posted to)
    memcpy(Q.era_key_host, key624, sizeof(unsigned) * 624);
    CK(c, hipMemcpyAsync(Q.era_key, Q.era_key_host, sizeof(unsigned) * 624, hipMemcpyHostToDevice, Q.rawst));
    Q.era_pos = pos;
    Q.era_gen_words = 0;
    Q.off_exact = 0;
    Q.gen_next = Q.gen_queued;
    Q.target = Q.gen_queued;
    Q.full = false;
    CK(c, hipMemsetAsync(Q.goffs + (Q.gen_queued & 7), 0, sizeof(long long), c->perm_stream));
    Q.valid = true;
    return 0;
}

// more words of the era on the words' stream, until `upto` of them are queued
static int perm_pipe_words(bgmm_ctx *c, const PermPtrs &P, long long upto, const unsigned *coef) {
    bgmm_ctx::PermPipe &Q = c->pp;
    if (upto > Q.era_cap) upto = Q.era_cap;
    const long long chunk = 624 * ((Q.cap_words + 623) / 624);
    bool any = false;
    while (Q.era_gen_words < upto) {
        const bool first = Q.era_gen_words == 0;
        // (the first chunk ends on a block boundary of the stream; every later one starts behind the last block of the words
        //  so far -- that block is the generator's state there -- and is whole blocks long)
        const int pos = first ? Q.era_pos : 624;
        long long n_words = first ? 624 * (((long long)pos + chunk + 623) / 624) - pos : chunk;
        if (Q.era_gen_words + n_words > Q.era_cap) n_words = 624 * ((Q.era_cap - Q.era_gen_words) / 624);
        if (n_words < 624) break;
        const unsigned *key_in = first ? Q.era_key : Q.era_raw + Q.era_gen_words - 624;
        const int chains = mt19937_chains_for_words(pos, n_words);
        launch_mt19937_raw(key_in, pos, Q.era_raw + Q.era_gen_words, n_words, chains >= 2 ? coef : nullptr, chains,
                           P.draw, c->perm_seeds, P.dspare, P.dspare_pos, Q.rawst);
        CK(c, hipGetLastError());
        Q.era_gen_words += n_words;
        any = true;
    }
    if (any) CK(c, hipEventRecord(Q.ev_raw, Q.rawst));
    return 0;
}

// queues one more generation; 1: the era has no room for it
// (the worker's: g = the generation, gen_next / off_exact / rounds as the stage calls had left them when it began)
// (coef: the jump polynomials as the posting stage call saw them, nullptr = chains one after the other)
static int perm_pipe_queue_one(bgmm_ctx *c, const PermPtrs &P, long long g, long long gen_next, long long off_exact, int rounds,
                               const unsigned *coef) {
    bgmm_ctx::PermPipe &Q = c->pp;
    constexpr int A = bgmm_ctx::PermPipe::kAhead;
    const long long N = c->d.N;
    const int slot = (int)(g % A);
    // where it starts at the latest (every generation in front of it reads at most cap_words), what it may read
    const long long hi = off_exact + (g - gen_next) * Q.cap_words;
    const long long need = hi + Q.cap_words + 1248;
    if (need > Q.era_cap - 1248) return 1;
    int rc = perm_pipe_words(c, P, need + 2 * Q.cap_words, coef);
    if (rc) return rc;
    if (Q.era_gen_words < need) return 1;
    hipStream_t D = c->perm_stream;
    CK(c, hipStreamWaitEvent(D, Q.ev_raw, 0));
    int *V = Q.vblk[slot];
    if (!launch_permutation_draws_chained(Q.era_raw, Q.era_key, Q.era_pos, Q.goffs + (g & 7), Q.goffs + ((g + 1) & 7), Q.cap_words, (int)N,
                                           Q.J[slot], Q.cnt, Q.pre0, Q.zero[g & 1], Q.zero[(g + 1) & 1], Q.nblk_pad, V + 1280,
                                           (long long *)(V + 628), (unsigned *)V, V + 624, rounds, D))
        return fail(c, BGMM_EDEVICE, "permutation kernels failed to launch");
    CK(c, hipEventRecord(Q.ev_draw[slot], D));
    // the swaps beside the next generation's draws; the order buffer they fill may be the one a sweep in flight still reads
    // (it was swapped out when its permutation was taken): behind everything the sweeps' stream holds now
    hipStream_t F = Q.fin;
    CK(c, hipEventRecord(Q.ev_sweep, c->stream));
    CK(c, hipStreamWaitEvent(F, Q.ev_sweep, 0));
    CK(c, hipStreamWaitEvent(F, Q.ev_draw[slot], 0));
    unsigned *H = Q.host[slot];
    if (Q.NB > 0) {
        if (!launch_permutation_swaps_bucketed((int)N, Q.NB, Q.bnd, Q.J[slot], Q.cursor, Q.slots, V + 627, P.pred, P.ptr, Q.ord[slot], F))
            return fail(c, BGMM_EDEVICE, "permutation kernels failed to launch");
    } else {
        if (!launch_permutation_swaps((int)N, Q.J[slot], P.pred, P.ptr, P.ks, P.idx, P.iota, c->perm_temp, c->perm_temp_bytes, P.changed,
                                      Q.ord[slot], F))
            return fail(c, BGMM_EDEVICE, "permutation kernels failed to launch");
    }
    CK(c, hipMemcpyAsync(H, V, sizeof(int) * 1344, hipMemcpyDeviceToHost, F));
    CK(c, hipEventRecord(Q.ev_fin[slot], F));
    return 0;
}

static void perm_pipe_worker(bgmm_ctx *c) {
    (void)hipSetDevice(c->device);
    bgmm_ctx::PermPipe &Q = c->pp;
    std::string my_err;
    g_err_sink = &my_err;               // (CK / fail on this thread never touch bgmm_ctx::err)
    std::unique_lock<std::mutex> lk(Q.mu);
    for (;;) {
        Q.cv.wait(lk, [&] { return Q.quit || (Q.gen_queued < Q.target && !Q.full && Q.wrc == 0); });
        if (Q.quit) return;
        const long long g = Q.gen_queued, gn = Q.gen_next, off = Q.off_exact;
        const int rounds = Q.rounds_q;
        const bool jump = Q.w_jump;
        unsigned *const coef = Q.w_coef;
        Q.busy = true;
        lk.unlock();
        my_err.clear();
        const int rc = perm_pipe_queue_one(c, Q.P, g, gn, off, rounds, jump ? coef : nullptr);
        lk.lock();
        Q.busy = false;
        if (rc == 0) Q.gen_queued = g + 1;
        else if (rc == 1) Q.full = true;
        else { Q.wrc = rc; Q.werr = my_err; }
        Q.cv.notify_all();
    }
}

// kAhead generations behind the one the next call takes: posted to the worker
static int perm_pipe_fill(bgmm_ctx *c) {
    bgmm_ctx::PermPipe &Q = c->pp;
    {
        std::lock_guard<std::mutex> g(Q.mu);
        Q.target = Q.gen_next + bgmm_ctx::PermPipe::kAhead;
        Q.w_jump = c->mt_jump_on;
        Q.w_coef = c->mt_coef;
    }
    Q.cv.notify_all();
    return 0;
}

// waits until the generation the next call takes has been queued -- or will not be (the era is full, the worker failed or
// has nothing posted): true iff it has
static bool perm_pipe_wait_queued(bgmm_ctx *c) {
    bgmm_ctx::PermPipe &Q = c->pp;
    std::unique_lock<std::mutex> lk(Q.mu);
    Q.cv.wait(lk, [&] { return Q.gen_queued > Q.gen_next || Q.full || Q.wrc != 0 || (!Q.busy && Q.gen_queued >= Q.target); });
    return Q.gen_queued > Q.gen_next;
}

static void perm_note_rounds(bgmm_ctx *c, const unsigned *H, int queued) {
    int r = 1;
    while (r <= queued && H[1280 + r] != 0) ++r;
    c->perm_last_rounds = r;
    if (r > c->perm_max_rounds) c->perm_max_rounds = r;
}

extern "C" int bgmm_stage_permutation_mt19937(bgmm_ctx *c, uint32_t *key624, int32_t *pos) {
    if (!c || !key624 || !pos) return BGMM_EINVAL;
    if (*pos < 0 || *pos > 624) return fail(c, BGMM_EINVAL, "MT19937 position must be in 0 .. 624");
    const long long N = c->d.N;
    if (N < 4096) return fail(c, BGMM_EUNSUPPORTED, "device permutations are for N >= 4096 (draw it on the host)");
    CK(c, hipSetDevice(c->device));
    PermPtrs P;
    int rc = perm_ensure(c, P);
    if (rc) return rc;
    unsigned *key_in_pinned = c->perm_host + 640;                 // [640, 1264): the state a generation starts from
    bool piped = c->mt_ahead_on && perm_pipe_wanted() && !c->pp.off;
    bool hit = false;
    if (piped && c->pp.built && c->pp.valid) {
        bgmm_ctx::PermPipe &Q = c->pp;
        const bool same = *pos == Q.expect_pos && memcmp(key624, Q.expect_key, sizeof(unsigned) * 624) == 0;
        bool queued = same && perm_pipe_wait_queued(c);
        if (same && !queued) {
            bool full;
            { std::lock_guard<std::mutex> g(Q.mu); full = Q.full && Q.wrc == 0; }
            if (full) {
                // (the era ran out of room and the generations in it have all been taken: the next one from here)
                rc = perm_pipe_drain(c);
                if (rc == 0) rc = perm_pipe_start_era(c, key624, *pos);
                if (rc == 0) rc = perm_pipe_fill(c);
                if (rc) return rc;
                queued = perm_pipe_wait_queued(c);
            }
        }
        if (queued) {
            const int slot = (int)(Q.gen_next % bgmm_ctx::PermPipe::kAhead);
            CK(c, hipEventSynchronize(Q.ev_fin[slot]));
            const unsigned *H = Q.host[slot];
            long long out[2];
            memcpy(out, H + 628, sizeof(out));
            if (getenv("BGMM_DEBUG_PERM") && Q.gen_next > 8) {
                // (device time from the end of the previous generation's draws to the end of this one's, and to its swaps')
                static double sum_d = 0.0, sum_f = 0.0; static long cnt = 0;
                const int prev = (int)((Q.gen_next - 1) % bgmm_ctx::PermPipe::kAhead);
                float md = 0.f, mf = 0.f;
                const hipError_t e1 = hipEventElapsedTime(&md, Q.ev_draw[prev], Q.ev_draw[slot]);
                const hipError_t e2 = hipEventElapsedTime(&mf, Q.ev_draw[slot], Q.ev_fin[slot]);
                if (e1 != hipSuccess || e2 != hipSuccess) {
                    static int said = 0;
                    if (said++ < 3) fprintf(stderr, "perm pipe: elapsed time failed (%d %d)\n", (int)e1, (int)e2);
                    (void)hipGetLastError();
                } else {
                    sum_d += md; sum_f += mf; cnt += 1;
                    if (cnt % 50 == 0) fprintf(stderr, "perm pipe: draws %.1f us per generation, swaps %.1f us behind them (%ld generations)\n",
                                                1e3 * sum_d / cnt, 1e3 * sum_f / cnt, cnt);
                }
            }
            std::lock_guard<std::mutex> g(Q.mu);
            if (H[625] == 1 && H[627] == 0 && out[1] == 0 && out[0] > 0) {
                memcpy(key624, H, sizeof(unsigned) * 624);
                *pos = (int32_t)H[624];
                perm_note_rounds(c, H, 60);
                // rounds queued per generation from here on: what the slowest generation so far needed (the round that
                // changed nothing + the write pass behind it) and four to spare; one that needs more is repaired the old way
                {
                    int want = c->perm_max_rounds + 1 + 4;
                    if (want < Q.rounds_floor) want = Q.rounds_floor;
                    Q.rounds_q = Q.rounds_fixed ? Q.rounds_fixed : (want < 10 ? 10 : (want > 60 ? 60 : want));
                }
                std::swap(c->d_order, Q.ord[slot]);
                std::swap(Q.ord[slot], Q.parked);
                Q.off_exact += out[0];
                Q.gen_next += 1;
                c->perm_hits += 1;
                Q.wfails = 0;
                hit = true;
            } else {
                // (it did not get through -- as a rule: not settled within the queued rounds; this one goes the old way)
                Q.rounds_floor = Q.rounds_q + 8 > 60 ? 60 : Q.rounds_q + 8;
                if (!Q.rounds_fixed) Q.rounds_q = Q.rounds_floor;
            }
        }
        if (!hit && getenv("BGMM_DEBUG_PERM")) {
            const unsigned *H = Q.host[Q.gen_next % bgmm_ctx::PermPipe::kAhead];
            long long out[2];
            memcpy(out, H + 628, sizeof(out));
            fprintf(stderr, "perm pipe miss: same %d queued %d full %d wrc %d gen_next %lld gen_queued %lld target %lld | went through %u changed %u out %lld %lld rounds_q %d\n",
                    (int)same, (int)queued, (int)Q.full, Q.wrc, Q.gen_next, Q.gen_queued, Q.target, H[625], H[627], out[0], out[1], Q.rounds_q);
        }
        if (!hit) { rc = perm_pipe_drain(c); if (rc) return rc; }
    } else if (!piped && c->perm_ahead_valid) {
        // The permutation BEHIND the last one was started when that one was handed out (look-ahead, as for the uniforms): taken
        // iff the caller's generator is exactly where that call left it.
        CK(c, hipEventSynchronize(c->perm_done));
        c->perm_ahead_valid = false;
        hit = c->mt_ahead_on && *pos == c->perm_ahead_pos_in && memcmp(key624, key_in_pinned, sizeof(unsigned) * 624) == 0;
        if (hit) {
            rc = perm_finish(c, P, c->perm_ahead_pos_in, c->d_order_ahead, c->perm_stream, key624, pos);
            if (rc) return rc;
            std::swap(c->d_order, c->d_order_ahead);
            c->perm_hits += 1;
        }
    }
    if (!hit) {
        SETTLE(c);                                      // (generated on the spot into d_order: a sweep in flight may read it)
        if (c->pp.built) { rc = perm_pipe_drain(c); if (rc) return rc; }
        const int pos_in = *pos;
        memcpy(key_in_pinned, key624, sizeof(unsigned) * 624);
        rc = perm_queue(c, P, key_in_pinned, pos_in, c->d_order, c->stream);
        if (rc == 0) rc = perm_queue_verdicts(c, P, c->stream);
        if (rc) return rc;
        CK(c, hipStreamSynchronize(c->stream));
        rc = perm_finish(c, P, pos_in, c->d_order, c->stream, key624, pos);
        if (rc) return rc;
        c->perm_misses += 1;
    }
    c->order_staged = true;
    c->order_is_perm = true;
    c->have_order = true;
    c->cur_order = c->d_order;
    if (piped && c->pp.off) piped = false;               // (the worker kept failing: latched by the drain above)
    if (piped) {
        // the generations behind this one, from the state just handed back; a pipe that cannot be set up (memory) is not an
        // error: the single look-ahead below serves from here on
        rc = perm_pipe_ensure(c, P);
        if (rc < 0) return rc;
        if (rc == 1) piped = false;
    }
    if (piped) {
        bgmm_ctx::PermPipe &Q = c->pp;
        if (!Q.valid) { rc = perm_pipe_start_era(c, key624, *pos); if (rc) return rc; }
        memcpy(Q.expect_key, key624, sizeof(unsigned) * 624);
        Q.expect_pos = *pos;
        return perm_pipe_fill(c);
    }
    if (c->mt_ahead_on) {
        // the next permutation, from the state just handed back, into the other buffer, beside the sweep about to be queued
        memcpy(key_in_pinned, key624, sizeof(unsigned) * 624);
        c->perm_ahead_pos_in = *pos;
        if (c->async_pending) { c->defer_perm = true; return 0; }      // (that buffer is the running sweep's order: after it)
        return perm_schedule(c, P);
    }
    return 0;
}

extern "C" int bgmm_get_staged_order(bgmm_ctx *c, int64_t *order_out) {
    if (!c || !order_out) return BGMM_EINVAL;
    if (!c->cur_order) return fail(c, BGMM_EINVAL, "no visiting order staged (the next sweep visits 0 .. N-1)");
    CK(c, hipSetDevice(c->device));
    CK(c, hipMemcpy(order_out, c->cur_order, sizeof(long long) * c->d.N, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int bgmm_get_totals(bgmm_ctx *c, int64_t *out4) {
    if (!c || !out4) return BGMM_EINVAL;
    SETTLE(c);
    for (int k = 0; k < 4; ++k) out4[k] = c->totals[k];
    return 0;
}

extern "C" int bgmm_get_short_step_stats(bgmm_ctx *c, int64_t *out2) {
    if (!c || !out2) return BGMM_EINVAL;
    SETTLE(c);
    out2[0] = c->short_stood;
    out2[1] = c->short_refused;
    return 0;
}

extern "C" int bgmm_set_mt_lookahead(bgmm_ctx *c, int32_t sweeps) {
    if (!c) return BGMM_EINVAL;
    SETTLE(c);
    if (sweeps < -1 || sweeps > kMtMaxMids) return fail(c, BGMM_EINVAL, "look-ahead depth must be -1 (auto), 0 (off) or 1 .. 8 sweeps");
    CK(c, hipSetDevice(c->device));
    // (buffers and batches in flight belong to the old depth)
    int rc = mt_wait_batches(c);
    if (rc) return rc;
    for (auto &b : c->mt_b) {
        // (uniforms staged out of a batch and not swept yet move into the context's own buffer before the batch goes)
        if (b.u && c->cur_u >= b.u && c->cur_u < b.u + (size_t)c->d.N * (size_t)(c->mt_depth > 0 ? c->mt_depth : 1)) {
            CK(c, hipMemcpy(c->d_u, c->cur_u, sizeof(double) * (size_t)c->d.N, hipMemcpyDeviceToDevice));
            c->cur_u = c->d_u;
        }
        b.launched = false;
        if (b.u) { (void)hipFree(b.u); b.u = nullptr; }
        if (b.host) { (void)hipHostFree(b.host); b.host = nullptr; }
    }
    if (c->mt_words_ahead) { (void)hipFree(c->mt_words_ahead); c->mt_words_ahead = nullptr; }
    c->mt_cur = -1;
    c->mt_ahead_on = sweeps != 0;
    const int auto_depth = mt_depth_for(c->d.N);
    c->mt_depth = sweeps <= 0 ? auto_depth : (c->d.N < 4096 ? 1 : sweeps);
    return 0;
}

extern "C" int bgmm_get_mt_lookahead_stats(bgmm_ctx *c, int64_t *out2) {
    if (!c || !out2) return BGMM_EINVAL;
    out2[0] = c->mt_ahead_hits + c->perm_hits;
    out2[1] = c->mt_ahead_misses + c->perm_misses;
    return 0;
}

extern "C" int bgmm_get_staged_uniforms(bgmm_ctx *c, double *u_out) {
    if (!c || !u_out) return BGMM_EINVAL;
    if (!c->cur_u) return fail(c, BGMM_EINVAL, "no sweep inputs staged");
    CK(c, hipSetDevice(c->device));
    CK(c, hipMemcpy(u_out, c->cur_u, sizeof(double) * c->d.N, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int bgmm_upload_streams(bgmm_ctx *c, int32_t n_sweeps, const double *u_all, const int64_t *order_all) {
    if (!c || !u_all || n_sweeps < 1) return BGMM_EINVAL;
    SETTLE(c);
    CK(c, hipSetDevice(c->device));
    const size_t N = (size_t)c->d.N;
    std::vector<char> perm_kind((size_t)n_sweeps, 1);
    if (order_all)
        for (int32_t t = 0; t < n_sweeps; ++t) {
            const int k = classify_order(order_all + (size_t)t * N, (long long)N);
            if (k < 0) return fail(c, BGMM_EINVAL, "visiting order holds an index outside 0 .. N-1");
            perm_kind[(size_t)t] = (char)k;
        }
    if (c->res_u) { (void)hipFree(c->res_u); c->res_u = nullptr; }
    if (c->res_order) { (void)hipFree(c->res_order); c->res_order = nullptr; }
    c->res_n = 0;
    CK(c, hipMalloc((void **)&c->res_u, sizeof(double) * N * n_sweeps));
    CK(c, hipMemcpy(c->res_u, u_all, sizeof(double) * N * n_sweeps, hipMemcpyHostToDevice));
    if (order_all) {
        CK(c, hipMalloc((void **)&c->res_order, sizeof(long long) * N * n_sweeps));
        CK(c, hipMemcpy(c->res_order, order_all, sizeof(long long) * N * n_sweeps, hipMemcpyHostToDevice));
    }
    c->res_n = n_sweeps;
    c->res_zero_u.assign((size_t)n_sweeps, 0);
    for (int32_t t = 0; t < n_sweeps; ++t)
        for (size_t i = 0; i < N; ++i)
            if (u_all[(size_t)t * N + i] == 0.0) { c->res_zero_u[(size_t)t] = 1; break; }
    c->res_perm = perm_kind;
    return 0;
}

// Buffers and LDS plan of the frozen-factor windows for K labels now (room for the labels a batch of
// windows may open).  Returns false when no plan fits (the classic kernels carry on).
static void gram_point(bgmm_ctx *c, int par);

static bool ensure_gram(bgmm_ctx *c, int K) {
    Dev &d = c->d;
    if (c->gram_off) return false;
    int cols = 0, T = 0, lds = 0;
    if (!gram_plan_for(K, &cols, &T, &lds)) return false;      // (too many labels NOW: asked again at the next batch)
    if (d.gcols != cols || !c->gram_mem[0]) {
        (void)hipStreamSynchronize(c->stream);
        for (void *&p : c->gram_mem) { if (p) (void)hipFree(p); p = nullptr; }
        const size_t sz[10] = {sizeof(double) * (size_t)cols * kGramRows * kGramRows, sizeof(double) * (size_t)cols * kGramRows,
                               sizeof(double) * (size_t)cols * kGramRows, sizeof(double) * (size_t)cols * kGramRows,
                               sizeof(GramMove) * (size_t)kGramMaxTerms, sizeof(int) * (size_t)kGramMaxTerms,
                               sizeof(double) * 2 * kGramRows, sizeof(double) * (size_t)cols * 40,
                               sizeof(int) * (16 + kGramMaxTerms), sizeof(GramXp)};
        for (int t = 0; t < 22; ++t) {
            const size_t bytes = t < 20 ? sz[t % 10] : sz[0];         // (two sets of window buffers, then gX twice)
            if (hipMalloc(&c->gram_mem[t], bytes + 64) != hipSuccess) {
                for (void *&p : c->gram_mem) { if (p) (void)hipFree(p); p = nullptr; }
                d.gcols = 0;
                if (++c->gram_alloc_fail >= 3) c->gram_off = true;      // (latched only when memory keeps failing)
                return false;
            }
            if (t % 10 >= 8 && t < 20) (void)hipMemset(c->gram_mem[t], 0, bytes);
        }
        gram_point(c, 0);
        d.gX = (double *)c->gram_mem[20];
        d.pipe = 0; d.pipe_pos = 0; d.xp_in = nullptr;
        d.gcols = cols;
        d.gram_terms = T;
        c->gram_lds = lds;
        gram_configure(d, lds);
    }
    return true;
}

// the window buffers of set `par` into a device view (pipelined windows alternate between the two sets; plain ones use set 0)
static void gram_point_view(bgmm_ctx *c, Dev &v, int par) {
    void **m = c->gram_mem + 10 * par;
    v.gC = (double *)m[0]; v.gq0 = (double *)m[1]; v.glp0 = (double *)m[2]; v.ge0 = (double *)m[3];
    v.gmoves = (GramMove *)m[4]; v.gtouched = (int *)m[5]; v.gM = (double *)m[6]; v.gcc = (double *)m[7];
    v.gfin = (int *)m[8]; v.xp_out = (unsigned char *)m[9];
    v.gX = (double *)c->gram_mem[20 + par];
}

// A batch of T PIPELINED frozen-factor windows from visit `pos` on (kernels_gram.hip, "Pipelined windows"): window k starts
// at pos + 64 k and works in buffer set k & 1.
//   main stream   cross(0)  resolve(0)  carry(1) resolve(1)  carry(2) resolve(2) ...
//   second stream     cross(1)      finish(0) cross(2)   finish(1) cross(3) ...
// cross(k) is made against the factors as finish(k - 2) left them -- the state at the start of window k - 1 --, carry(k)
// applies window k - 1's terms; finish(k) needs resolve(k), resolve(k) needs cross(k) (+ carry).  If the chain breaks on the
// device (Ctrl::pipe_break) the rest of the batch stands still; the caller reads the control block and goes on from there.
static int gram_pipe_batch(bgmm_ctx *c, int T, long long pos) {
    Dev &d = c->d;
    hipStream_t M = c->stream;
    if (!c->pipe_stream) CK(c, hipStreamCreateWithFlags(&c->pipe_stream, hipStreamNonBlocking));
    hipStream_t S = c->pipe_stream;
    while (c->pipe_ev.size() < (size_t)(2 * T + 2)) {
        hipEvent_t e;
        CK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->pipe_ev.push_back(e);
    }
    auto evG = [&](int k) { return c->pipe_ev[(size_t)(2 + 2 * k)]; };
    auto evR = [&](int k) { return c->pipe_ev[(size_t)(3 + 2 * k)]; };
    Dev v[2] = {d, d};
    for (int p = 0; p < 2; ++p) { gram_point_view(c, v[p], p); v[p].pipe = 1; }
    auto view = [&](int k) -> Dev & {
        Dev &x = v[k & 1];
        x.pipe = k == 0 ? 2 : 1;
        x.pipe_pos = pos + (long long)kGramRows * k;
        x.xp_in = v[(k + 1) & 1].xp_out;            // what window k - 1 exported
        return x;
    };
    CK(c, hipMemsetAsync(&d.ctrl->pipe_break, 0, sizeof(int), M));
    if (!launch_gram_cross(view(0), false, M)) return fail(c, BGMM_EDEVICE, "frozen-factor window launch failed");
    CK(c, hipEventRecord(c->pipe_ev[0], M));
    CK(c, hipStreamWaitEvent(S, c->pipe_ev[0], 0));
    if (T > 1) {
        launch_gram_cross(view(1), true, S);
        CK(c, hipEventRecord(evG(1), S));
    }
    for (int k = 0; k < T; ++k) {
        if (k > 0) {
            CK(c, hipStreamWaitEvent(M, evG(k), 0));
            launch_gram_carry(view(k), M);
        }
        launch_gram_resolve_only(view(k), c->gram_lds, M);
        CK(c, hipEventRecord(evR(k), M));
        CK(c, hipStreamWaitEvent(S, evR(k), 0));
        launch_gram_finish(view(k), S);
        if (k + 2 < T) {
            launch_gram_cross(view(k + 2), true, S);
            CK(c, hipEventRecord(evG(k + 2), S));
        }
    }
    CK(c, hipEventRecord(c->pipe_ev[1], S));
    CK(c, hipStreamWaitEvent(M, c->pipe_ev[1], 0));
    CK(c, hipGetLastError());
    c->pipe_batches += 1;
    return 0;
}
static void gram_point(bgmm_ctx *c, int par) { gram_point_view(c, c->d, par); }

static int ensure_events(bgmm_ctx *c, size_t n) {
    while (c->ev0.size() < n) {
        hipEvent_t a, b;
        CK(c, hipEventCreate(&a));
        CK(c, hipEventCreate(&b));
        c->ev0.push_back(a);
        c->ev1.push_back(b);
    }
    return 0;
}

// Chains of one group call that are in the frozen-factor regime TOGETHER (burn-in from a random start) share their
// launches: the hardware runs about four kernels of different streams side by side, whatever the number of streams, so
// eight chains with four small launches per window each queue up behind one another -- while one launch whose grid is
// (x, chain) runs the eight resolvers truly side by side (kernels_gram.hip: *_group_kernel).  The chains' host threads meet
// here.  Every thread declares, once per batch of its sweep loop, either "a batch of frozen-factor windows" (submit: it
// waits) or "something else" (pass: the others do not wait for it); when nobody is undeclared, one of the waiting threads
// is made leader and queues the batch for all waiting chains of its shape on its own stream, behind an event of each
// member's stream; the members' streams wait for the leader's.  Same kernels, same per-chain control blocks: the
// trajectories are those of separate sweeps.
struct GramCombiner {
    enum { UNKNOWN = 0, WAITING = 1, BUSY = 2, DONE = 3 };
    struct Slot { int state = UNKNOWN; bgmm_ctx *c = nullptr; int T = 0; int result = 0; hipEvent_t ev = nullptr; };
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Slot> slots;
    int leader = -1;
    long long shared_batches = 0, shared_members = 0;

    void elect_locked() {
        if (leader >= 0) return;
        int first = -1;
        for (size_t k = 0; k < slots.size(); ++k) {
            if (slots[k].state == UNKNOWN) return;
            if (slots[k].state == WAITING && first < 0) first = (int)k;
        }
        if (first >= 0) { leader = first; cv.notify_all(); }
    }
    void declare(int i, int state) {
        std::lock_guard<std::mutex> lk(mu);
        slots[(size_t)i].state = state;
        elect_locked();
    }
};

static int gram_group_launch(GramCombiner &G, const std::vector<int> &members, int T, std::vector<hipEvent_t> &ev_of);

// Returns 0: the batch has been queued with the group's (the chain's stream waits for it); 1: queue it yourself; < 0: error.
static int combiner_submit(bgmm_ctx *c, int T) {
    GramCombiner &G = *c->combiner;
    const int me = c->combiner_slot;
    // (a chain that cannot take part queues its batch itself -- and says so, or the others would wait for its declaration)
    if (!c->grp_ev_in) {
        if (hipEventCreateWithFlags(&c->grp_ev_in, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&c->grp_ev_out, hipEventDisableTiming) != hipSuccess) {
            G.declare(me, GramCombiner::BUSY);
            return 1;
        }
    }
    if (hipEventRecord(c->grp_ev_in, c->stream) != hipSuccess) { G.declare(me, GramCombiner::BUSY); return 1; }
    std::unique_lock<std::mutex> lk(G.mu);
    GramCombiner::Slot &S = G.slots[(size_t)me];
    S.state = GramCombiner::WAITING; S.T = T; S.result = 1; S.ev = nullptr;
    G.elect_locked();
    G.cv.wait(lk, [&] { return S.state != GramCombiner::WAITING || G.leader == me; });
    if (S.state == GramCombiner::WAITING) {
        // leader: the waiting chains of this chain's shape (device, D, column plan)
        std::vector<int> members;
        int Tmax = 0;
        for (size_t k = 0; k < G.slots.size(); ++k) {
            const GramCombiner::Slot &o = G.slots[k];
            if (o.state != GramCombiner::WAITING) continue;
            if (o.c->device == c->device && o.c->d.D == c->d.D && o.c->d.gcols == c->d.gcols && o.c->gram_lds == c->gram_lds) {
                members.push_back((int)k);
                if (o.T > Tmax) Tmax = o.T;
            }
        }
        int rc = 1;
        std::vector<hipEvent_t> ev_of;                  // per member: the event its stream waits for (its sub-group's)
        if (members.size() >= 2) {
            lk.unlock();
            rc = gram_group_launch(G, members, Tmax, ev_of);
            lk.lock();
            if (rc == 0) { G.shared_batches += 1; G.shared_members += (long long)members.size(); }
        }
        // everybody who waited goes on: the members with the shared batch (or, if it could not be queued, on their own),
        // the chains of other shapes on their own
        for (size_t k = 0; k < G.slots.size(); ++k) {
            GramCombiner::Slot &o = G.slots[k];
            if (o.state != GramCombiner::WAITING) continue;
            const auto it = std::find(members.begin(), members.end(), (int)k);
            const bool member = it != members.end() && members.size() >= 2;
            o.result = member ? rc : 1;
            o.ev = (member && rc == 0) ? ev_of[(size_t)(it - members.begin())] : nullptr;
            o.state = GramCombiner::UNKNOWN;
        }
        G.leader = -1;
        G.cv.notify_all();
    }
    const int result = S.result;
    hipEvent_t ev = S.ev;
    lk.unlock();
    if (result == 0 && ev && ev != c->grp_ev_out) {       // (a sub-group's leader queued the batch on its own stream)
        if (hipStreamWaitEvent(c->stream, ev, 0) != hipSuccess) return BGMM_EDEVICE;
    }
    return result;
}

// The shared batch is queued as a few SUB-GROUPS, each on the stream of its first member: the one-workgroup resolvers of
// one sub-group run beside the wide kernels (cross forms, rebuilds) of the others -- with every chain in ONE launch
// sequence the chip idles through each window's resolver phase and the resolvers wait through its wide phases (eight
// C4 chains: 201 + 132 us per window whatever runs beside them).  One stream per chain, the other extreme, keeps only
// about four kernels in flight and stretches every one of them.  BGMM_GROUP_SPLIT
// overrides the number of sub-groups (1: one launch sequence for all).
static int gram_group_launch(GramCombiner &G, const std::vector<int> &members, int T, std::vector<hipEvent_t> &ev_of) {
    bgmm_ctx *lead = G.slots[(size_t)members[0]].c;
    const int m = (int)members.size();
    ev_of.assign((size_t)m, nullptr);
    if (hipSetDevice(lead->device) != hipSuccess) return 1;
    if (lead->grp_devs_cap < m) {
        if (lead->grp_devs) (void)hipFree(lead->grp_devs);
        lead->grp_devs = nullptr; lead->grp_devs_cap = 0;
        if (hipMalloc((void **)&lead->grp_devs, sizeof(Dev) * (size_t)m) != hipSuccess) return 1;
        lead->grp_devs_cap = m;
    }
    static const int split_env = [] { const char *e = getenv("BGMM_GROUP_SPLIT"); return e ? atoi(e) : 0; }();
    int n_sub = split_env > 0 ? split_env : (m >= 4 ? 2 : 1);
    if (n_sub > m / 2) n_sub = m / 2 > 0 ? m / 2 : 1;
    std::vector<Dev> views((size_t)m);
    std::vector<int> reach_of((size_t)n_sub, 0), lo_of((size_t)n_sub + 1, 0);
    for (int g = 0; g <= n_sub; ++g) lo_of[(size_t)g] = (int)((long long)m * g / n_sub);
    for (int g = 0; g < n_sub; ++g) {
        bgmm_ctx *sl = G.slots[(size_t)members[(size_t)lo_of[(size_t)g]]].c;
        for (int k = lo_of[(size_t)g]; k < lo_of[(size_t)g + 1]; ++k) {
            bgmm_ctx *o = G.slots[(size_t)members[(size_t)k]].c;
            views[(size_t)k] = o->d;
            const int r = o->d.gram_K + o->d.gram_terms / 2 + 2 + 32;
            if (r > reach_of[(size_t)g]) reach_of[(size_t)g] = r;
            // (what the member has queued on its own stream -- the sweep's opening, rebuilt factors -- comes first)
            if (o != sl && hipStreamWaitEvent(sl->stream, o->grp_ev_in, 0) != hipSuccess) return 1;
            ev_of[(size_t)k] = sl->grp_ev_out;
        }
    }
    // (a blocking copy: the views are host memory of this call; the array's last readers -- the shared batch before this one --
    // have been waited for by every one of its members)
    if (hipMemcpy(lead->grp_devs, views.data(), sizeof(Dev) * (size_t)m, hipMemcpyHostToDevice) != hipSuccess) return 1;
    // (from here on a failure is an error for every member, not a reason to queue their batches separately: part of the shared
    // batch may already be in the queue, and separate launches would run beside it on the same chains)
    // window by window across the sub-groups, so that the host queues them at the same pace
    for (int t = 0; t < T; ++t)
        for (int g = 0; g < n_sub; ++g) {
            bgmm_ctx *sl = G.slots[(size_t)members[(size_t)lo_of[(size_t)g]]].c;
            if (!launch_gram_group_step(sl->d, lead->grp_devs + lo_of[(size_t)g], lo_of[(size_t)g + 1] - lo_of[(size_t)g],
                                        reach_of[(size_t)g], sl->gram_lds, sl->stream)) return BGMM_EDEVICE;
        }
    if (hipGetLastError() != hipSuccess) return BGMM_EDEVICE;
    for (int g = 0; g < n_sub; ++g) {
        bgmm_ctx *sl = G.slots[(size_t)members[(size_t)lo_of[(size_t)g]]].c;
        if (hipEventRecord(sl->grp_ev_out, sl->stream) != hipSuccess) return BGMM_EDEVICE;
    }
    return 0;
}

// One sweep.  phase 0: all of it.  Phases 1 and 2 split it for bgmm_group_sweep_staged, which opens the sweeps of several
// chains and runs their one-workgroup sweeps (kernels_seq.hip) in ONE launch each: phase 1 = everything in front of
// sweep_begin; returns 1 if the chain can take the one-workgroup sweep (bgmm_ctx::grp_cap = its LDS plan; the caller
// launches, fills ctrl_host and comes back with phase 2), otherwise carries on as phase 0.  Phase 2 = what follows.
static int sweep_impl(bgmm_ctx *c, int32_t use_power, double power, int phase) {
    if (!c) return BGMM_EINVAL;
    if (!c->assigned) return fail(c, BGMM_EINVAL, "bgmm_set_assignments has not been called");
    if (c->async_pending && phase != 4) return fail(c, BGMM_EINVAL, "a sweep is in flight: bgmm_sweep_staged_end first");
    CK(c, hipSetDevice(c->device));
    Dev &d = c->d;
    // phase 3 (bgmm_sweep_staged_begin): as phase 0, but a first batch that is a lean or a short step -- a chain at rest --
    // is left in the queue (returns 2); phase 4 (bgmm_sweep_staged_end, which has waited for it and read the control
    // block) carries on behind it like phase 2 does behind a group launch
    const bool resume = phase == 2 || phase == 4;
    if (!resume) {
        d.use_power = use_power ? 1 : 0;
        d.power = use_power ? power : 1.0;
        if (!c->cur_u) return fail(c, BGMM_EINVAL, "no sweep inputs staged");
        d.u = c->cur_u;
        d.order = c->cur_order;
        d.order_perm = c->order_is_perm ? 1 : 0;
        c->order_staged = false;            // (a staged permutation serves one sweep)
        d.sweep_visits = c->next_sweep_visits;
        c->next_sweep_visits = 0;
        c->run_zero_u = c->cur_zero_u;
        c->run_order_is_perm = c->order_is_perm;
    }
    const bool partial = d.sweep_visits > 0 && d.sweep_visits < d.N;
    resolve_kind(c);
    const bool use_prune = c->prune_mode != 1 && (c->kind == KERNEL_MFMA || d.cov_type != COV_FULL) && !c->run_zero_u;
    d.prune_enabled = use_prune ? 1 : 0;        // (sweep_begin opens the first window under the device's rule)
    // (certify_kernel runs in front of every pruned window: on data it can do nothing for it costs
    // a tenth of the pruning kernel behind it; a rule that left it out after a poor yield misjudged
    // cold caches for hopeless data twice and was dropped)
    const bool use_certify = use_prune && c->prune_mode != 3;
    d.use_certify = use_certify ? 1 : 0;
    // (a lean step looks at the whole sweep in storage order: not for a sweep that stops early)
    bool lean = use_certify && c->lean_ok && c->prune_mode != 2 && !partial;
    // (certified stays off -- prune_mode 3 -- and the chain at rest: home_kernel between sweep_begin and apply, nothing else)
    // (bgmm_set_home_pass(3) tries one in EVERY sweep: the refusal path under test)
    bool short_step = use_prune && !use_certify && (c->short_ok || c->home_mode == 3) && c->prune_mode != 2 && !partial &&
                      !c->tables_robust && c->resolver_mode == 0;
    hipStream_t st = c->stream;
    if (!resume) {
        if (c->safe_dense_on && (++c->safe_dense_age & 7) == 0) c->safe_dense_on = false;     // (the tables get another look)
        d.seat_dirty = 0;
        if (d.use_power != c->seat_use_power || (d.use_power && d.power != c->seat_power)) {
            d.seat_dirty = 1;
            launch_build_seat_table(d, c->tabSeat, st);
            c->seat_use_power = d.use_power;
            c->seat_power = d.power;
        }
    }
    // Launch grids follow the window scale: sized for twice the device's current window (at least
    // 4096 rows, at most the allocation), never below the window that is already open.
    auto rows_for = [&](long long win_now, long long open_rows, long long grow = 2) -> int {
        long long r = 4096;
        while (r < grow * win_now && r < c->win_rows) r <<= 1;
        while (r < open_rows && r < c->win_rows) r <<= 1;
        if (r > c->win_rows) r = c->win_rows;
        return (int)r;
    };
    if (!resume) {
        d.batch_rows = rows_for(c->ctrl_host->win_size > 0 ? c->ctrl_host->win_size : c->win_rows, 0);
        if (c->moves_prev != 0 && d.cov_type == COV_FULL && use_prune)
            launch_refresh_stale(d, c->ctrl_host->job.K, st);   // (tight bounds again after a sweep with moves)
    }
    // Tiny dimensions: one workgroup walks the visits in order with the labels' state in LDS
    // (kernels_seq.hip: sweep_seq_kernel).  It leaves the sweep DONE, or -- when the labels outgrow
    // its LDS plan -- a window open at the visit it stopped at, and the loop below carries on.
    int seq_plan = 0;                      // labels the one-workgroup sweep would plan LDS for (0: not for this sweep)
    if (phase == 2) {
        seq_plan = c->grp_cap;
    } else if (phase == 4) {
        seq_plan = 0;
    } else if (seq_shape(c) && c->kernel_kind == KERNEL_AUTO && c->resolver_mode == 0 && c->prune_mode != 2 &&
               c->run_order_is_perm) {
        int cap = 2;
        while (sweep_seq_lds_bytes(d.D, cap + 16) <= 150 * 1024) cap += 16;
        if (c->seq_cap >= 2 && c->seq_cap < cap) cap = c->seq_cap;     // (bgmm_set_seq_plan)
        if (cap > d.K_max + 1) cap = d.K_max + 1;
        if (c->ctrl_host->job.K + 1 <= cap) seq_plan = cap;
    }
    if (phase == 1 && seq_plan > 0) {
        c->grp_cap = seq_plan;
        return 1;
    }
    // (sweep_begin opens the first window, and whether the kept bucket sort can serve it depends on the layout the
    //  batch wants -- padded for the home pass: the view it gets must already say so)
    d.use_home = (d.cov_type == COV_FULL && c->kind == KERNEL_MFMA && c->home_pass) ? 1 : 0;
    if (!resume) launch_sweep_begin(d, st);
    long long steps_done = 0;
    bool seq_ran = false;
    if (phase == 4) {                      // (behind a first batch that was waited for elsewhere: nothing lean or short any more)
        seq_ran = true;
        steps_done = c->ctrl_host->n_steps;
        lean = false;
        short_step = false;
    }
    if (seq_plan > 0) {
        if (phase != 2) {
            if (!launch_sweep_seq(d, seq_plan, st)) return fail(c, BGMM_EDEVICE, "sequential sweep kernel launch failed");
            CK(c, hipGetLastError());
            int rc = fetch_ctrl(c);
            if (rc) return rc;
        }
        seq_ran = true;
        steps_done = c->ctrl_host->n_steps;
    }
    // Steps are queued blindly; a step issued after the sweep is DONE is a (cheap) no-op.
    // Lower bound on the steps still needed: one per remaining window.  On top of that,
    // one step per expected mover, estimated from the rate observed so far in this sweep
    // (first chunk: from the previous sweep).
    const long long N = partial ? d.sweep_visits : d.N;
    long long pos = 0;
    int win = c->ctrl_host->win_size > 0 ? c->ctrl_host->win_size : c->win_rows;
    double rate = c->last_move_rate;
    // movers per visit over the last batch of steps (first batch: over the previous sweep): what decides
    // between the per-mover kernel chain and the frozen-factor windows while the device's running mean
    // is still catching up with a change of regime
    double recent_rate = c->last_move_rate;
    long long batch_pos0 = 0, batch_moves0 = 0;
    bool first_batch = true;               // (sweep_begin has just opened a fresh window at visit 0)
    bool gram_skip = false;                // frozen-factor windows made no progress in this sweep: not queued again
    bool safe_skip = c->safe_rest > 0;     // the same for safe-stay windows (or they did poorly a sweep ago: bgmm_ctx::safe_rest)
    if (c->safe_rest > 0) c->safe_rest -= 1;
    if (seq_ran) {
        first_batch = false;
        pos = c->ctrl_host->job.pos;
    }
    for (;;) {
        if (seq_ran && (c->ctrl_host->error != 0 || c->ctrl_host->job.mode == MODE_DONE)) break;
        const long long remaining = N - pos;
        long long lb = (remaining + win - 1) / win;
        long long extra = (long long)std::ceil(rate * (double)remaining * 1.1);
        if (extra > 2048) extra = 2048;
        // (a clean window doubles the device's window, but the launch grids of this batch were sized for
        // the current one: while nothing moves a few steps per batch are all that can be used)
        if (rate == 0.0 && lb > 8) lb = 8;
        long long Tl = lb + extra;
        if (Tl < 1) Tl = 1;
        if (Tl > 4096) Tl = 4096;
        // (the first batch of a sweep runs on the previous sweep's mover rate: keep it short, the next
        // one is planned on what this sweep has shown -- a step queued behind the end of the sweep is a
        // dozen empty launches)
        if (first_batch && Tl > 8) Tl = 8;
        // Mover-dense stretches (burn-in, overlapping clusters): frozen-factor windows (kernels_gram.hip).
        // Four launches per window of 64 visits, no per-mover kernel chain.  resolver_mode 3 forces them.
        bool use_gram = false, gram_possible = false, use_safe = false;
        d.safe_mode = 0;
        const bool rm_gram = c->resolver_mode == 0 || c->resolver_mode >= 3;      // (3 / 4 force a kind, 5: never safe-stay)
        if (d.cov_type == COV_FULL && rm_gram && c->run_order_is_perm &&
            c->prune_mode != 2 && d.Dp / 16 <= 8 && (c->resolver_mode == 3 || c->kernel_kind != KERNEL_VALU)) {
            const Ctrl &hc = *c->ctrl_host;
            const bool safe_ok = c->resolver_mode != 3 && c->resolver_mode != 5 && c->kind == KERNEL_MFMA &&
                                 c->prune_mode != 1 && !safe_skip;
            const bool very_dense = recent_rate > kSafeDenseRate || hc.ema_run < 1.0 / kSafeDenseRate;
            // (a stretch of kSafeRun visits without a mover behind us: the chain has come to rest -- the pruned windows take
            // over, whose first pass also leaves the per-point caches the certificates of the next sweep are made from)
            // -- and so do they after a sweep in which nothing moved (the control block on the host still carries that
            // sweep's running mean: sweep_begin resets it on the device)
            const bool quiet = (double)(pos - hc.last_mover) > kSafeRun || (first_batch && c->moves_prev == 0);
            const bool moderate = (hc.ema_run < kSafeRun || recent_rate * kSafeRun > 1.0) && !quiet;
            const bool want_safe = c->resolver_mode == 4 || (safe_ok && moderate && !very_dense);
            if (want_safe && !safe_skip) use_safe = ensure_gram(c, hc.job.K);
            const bool dense = c->resolver_mode == 3 || hc.ema_run < kGramRun || recent_rate * kGramRun > 1.0;
            if (!use_safe && dense && !gram_skip && c->resolver_mode != 4) use_gram = ensure_gram(c, hc.job.K);
            gram_possible = !c->gram_off;
        }
        // (beside other chains of a group call: a batch of frozen-factor windows is queued together with theirs -- declared
        // when it is submitted; any other kind of batch is this chain's own business, and nobody waits for it meanwhile)
        if (c->combiner && !(use_gram && !c->timing)) c->combiner->declare(c->combiner_slot, GramCombiner::BUSY);
        if (use_safe) {
            const Ctrl &hc = *c->ctrl_host;
            // windows still needed: from the visits a window has covered on average so far in this sweep
            double vpw = hc.safe_windows > 0 ? (double)(pos > 0 ? pos : 1) / (double)hc.safe_windows : (double)hc.safe_L;
            if (vpw < 64.0) vpw = 64.0;
            long long Tg = (long long)std::ceil((double)remaining / vpw) + 1;
            if (first_batch && Tg > 8) Tg = 8;
            if (Tg > 256) Tg = 256;
            if (recent_rate == 0.0 && Tg > 4) Tg = 4;         // (nothing has moved lately: look again soon, the chain may be at rest)
            if (c->timing) { int rc = ensure_events(c, (size_t)Tg); if (rc) return rc; }
            d.safe_mode = 1; d.lean_step = 0; d.publish = 0; d.prune_enabled = 2; d.use_certify = 0; d.use_home = 1;
            d.resid_dense = 0;
            d.safe_dense = c->safe_dense_pin >= 0 ? (c->safe_dense_pin ? 1 : 0) : (c->safe_dense_on ? 1 : 0);
            c->proof_batches[d.safe_dense] += 1;
            const long long resid0 = hc.safe_resid_sum, sorted0 = hc.safe_sorted_sum;
            d.safe_cap = c->safe_cap_user;
            d.gram_K = hc.job.K;
            // (a dense proof pass takes its forms from the look-ahead's ring: a second stream scores them a chunk at a time
            //  beside the resolver -- kernels_safe.hip "look-ahead"; not while the launches are being timed one by one)
            const bool ahead = d.safe_dense && c->ahead_chunk > 0 && !c->timing && d.qstride >= 2ll * c->ahead_chunk &&
                               d.cov_type == COV_FULL && c->kind == KERNEL_MFMA;
            d.ahead_C = ahead ? c->ahead_chunk : 0;
            if (ahead && !c->ahead_stream) {
                CK(c, hipStreamCreateWithFlags(&c->ahead_stream, hipStreamNonBlocking));
                for (auto &row : c->ahead_ev) for (hipEvent_t &e : row) CK(c, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            }
            {   // launch grids: room for the stretch to double twice inside the batch
                long long r = 4096;
                while (r < 4ll * hc.safe_L && r < c->win_rows) r <<= 1;
                if (ahead && r > c->ahead_chunk) r = c->ahead_chunk > 4096 ? c->ahead_chunk : 4096;     // (stretches end with their chunk)
                if (r > c->win_rows) r = c->win_rows;
                d.batch_rows = (int)r;
            }
            lean = false;
            first_batch = false;
            const long long w0 = hc.safe_windows, mv0 = hc.n_moves, rows0 = hc.safe_rows;
            const auto t_batch0 = std::chrono::steady_clock::now();
            launch_safe_open(d, st);
            for (int t = 0; t < (int)Tg; ++t) {
                SafeAhead ah{c->ahead_stream, c->ahead_ev[0][t & 7], c->ahead_ev[1][t & 7]};
                // (the request made by the step before has been served before this step's plan books it)
                if (ahead && t > 0) CK(c, hipStreamWaitEvent(st, c->ahead_ev[1][(t - 1) & 7], 0));
                if (!launch_safe_step(d, c->gram_lds, d.batch_rows, st, c->timing ? c->ev0[t] : nullptr, c->timing ? c->ev1[t] : nullptr,
                                      ahead ? &ah : nullptr))
                    return fail(c, BGMM_EDEVICE, "safe-stay window launch failed");
            }
            if (ahead) CK(c, hipStreamWaitEvent(st, c->ahead_ev[1][((int)Tg - 1) & 7], 0));       // (the second stream is idle when the batch ends)
            CK(c, hipGetLastError());
            int rc = fetch_ctrl(c);
            if (rc) return rc;
            const Ctrl &h = *c->ctrl_host;
            steps_done = h.n_steps;
            const bool stalled = h.gram_stall != 0;
            if (stalled) {
                c->ctrl_host->gram_stall = 0;
                CK(c, hipMemcpy(&d.ctrl->gram_stall, &c->ctrl_host->gram_stall, sizeof(int), hipMemcpyHostToDevice));
            }
            if (h.error != 0 || h.job.mode == MODE_DONE) { d.safe_mode = 0; d.ahead_C = 0; d.use_certify = use_certify ? 1 : 0; break; }
            if (h.job.pos == pos && !stalled) safe_skip = true;
            if (c->resolver_mode == 0 && h.safe_windows - w0 >= 16 && h.job.pos > pos) {
                // Did these windows pay?  Where movers are few and far between, the per-mover kernel chain (~0.2 ms per
                // mover at D = 64, pruned windows in between) is the yardstick: a stretch in which the proofs keep failing
                // (components a handful of nats apart, or too small to vouch for their members) is better left to it.
                const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_batch0).count();
                const double visits = (double)(h.job.pos - pos), mrate = (double)(h.n_moves - mv0) / visits;
                const double dscale = d.Dp > 64 ? (double)d.Dp / 64.0 : 1.0;
                const double rate_chain = 1.0 / (mrate * 0.2 * dscale + 3e-4);
                if (mrate < 2e-3 && visits / ms < 0.6 * rate_chain) { safe_skip = true; c->safe_rest = 1; }
                if ((double)(h.safe_rows - rows0) > kSafeWalkShare * visits) safe_skip = true;      // (too little proven: plain windows)
            }
            // (where the clusters overlap the per-home tables prove nothing and every visit of a stretch goes to the exact
            // forms: the dense proof pass does the same arithmetic in three launches instead of eleven)
            if (!d.safe_dense && h.safe_sorted_sum - sorted0 >= 1024 &&
                2 * (h.safe_resid_sum - resid0) > h.safe_sorted_sum - sorted0) { c->safe_dense_on = true; c->safe_dense_age = 0; }
            pos = h.job.pos;
            win = h.win_size > 0 ? h.win_size : win;
            rate = pos > 0 ? (double)h.n_moves / (double)pos : rate;
            if (pos > batch_pos0) recent_rate = (double)(h.n_moves - batch_moves0) / (double)(pos - batch_pos0);
            batch_pos0 = pos; batch_moves0 = h.n_moves;
            d.safe_mode = 0;
            d.safe_dense = 0;
            d.ahead_C = 0;
            d.use_certify = use_certify ? 1 : 0;      // (the safe batch ran without certificates: what follows does not)
            c->tables_robust = true;
            continue;
        }
        // (the frozen-factor windows take over once the movers prove dense: look again soon)
        if (gram_possible && rate > 0.0 && Tl > 24) Tl = 24;
        int T = (int)Tl;
        if (use_gram) {
            const Ctrl &hc = *c->ctrl_host;
            // windows still needed: from the rows a window has consumed on average so far in this sweep
            double rpw = hc.gram_windows > 0 ? (double)hc.gram_rows_total / (double)hc.gram_windows : 32.0;
            if (rpw < 8.0) rpw = 8.0;
            long long Tg = (long long)std::ceil((double)remaining / rpw) + 1;
            if (first_batch && Tg > 16) Tg = 16;
            if (Tg > 512) Tg = 512;
            // (beside other chains of a group call: shorter batches, so that a chain that has fallen out of step with the
            // others -- it queued a batch of its own while they were busy -- meets them again soon)
            if (c->combiner && Tg > 128) Tg = 128;
            if (c->timing) { int rc = ensure_events(c, (size_t)Tg); if (rc) return rc; }
            d.lean_step = 0; d.publish = 0; d.prune_enabled = 0;
            d.gram_K = hc.job.K;
            lean = false;
            const bool was_first = first_batch;
            first_batch = false;
            // (chains of a group call that are here together share the launches: GramCombiner above)
            int own = 1;
            if (c->combiner && !c->timing) {
                own = combiner_submit(c, (int)Tg);
                if (own < 0) return fail(c, BGMM_EDEVICE, "shared frozen-factor launch failed");
            }
            // (a chain on its own, far inside the mover-dense regime: the windows pipelined -- gram_finish and the next cross
            //  forms on a second stream beside the resolver; after a break of the chain a couple of plain batches first)
            bool piped = false;
            if (own && c->pipe_mode && !c->combiner && !c->timing && c->resolver_mode != 1 && remaining >= 4 * kGramRows &&
                (was_first || hc.job.pos == pos)) {
                if (c->pipe_hold > 0) c->pipe_hold -= 1;
                else {
                    long long Tp = remaining / kGramRows;
                    if (Tp > Tg) Tp = Tg;
                    if (Tp > 128) Tp = 128;
                    if (Tp >= 4) {
                        gram_point(c, 0);
                        const int rcp = gram_pipe_batch(c, (int)Tp, pos);
                        if (rcp) return rcp;
                        piped = true;
                    }
                }
            }
            if (own && !piped) {
                gram_point(c, 0);
                d.pipe = 0;
                for (int t = 0; t < (int)Tg; ++t)
                    if (!launch_gram_step(d, c->gram_lds, st, c->timing ? c->ev0[t] : nullptr, c->timing ? c->ev1[t] : nullptr))
                        return fail(c, BGMM_EDEVICE, "frozen-factor window launch failed");
            }
            CK(c, hipGetLastError());
            int rc = fetch_ctrl(c);
            if (rc) return rc;
            const Ctrl &h = *c->ctrl_host;
            if (c->timing) {
                const long long worked = h.n_steps - steps_done;
                for (long long t = 0; t < worked && t < Tg; ++t) {
                    float ms = 0.f;
                    CK(c, hipEventElapsedTime(&ms, c->ev0[(size_t)t], c->ev1[(size_t)t]));
                    c->timed_ms += (double)ms;
                    c->timed_launches += 1;
                }
            }
            steps_done = h.n_steps;
            const bool stalled = h.gram_stall != 0;
            if (h.gram_stall) {            // the labels outgrew the columns: larger buffers, or the classic kernels
                c->ctrl_host->gram_stall = 0;
                CK(c, hipMemcpy(&d.ctrl->gram_stall, &c->ctrl_host->gram_stall, sizeof(int), hipMemcpyHostToDevice));
                // (the plan -- columns, terms, the draw wave's width -- is re-picked for the labels there are now)
            }
            if (piped && h.pipe_break) { c->pipe_breaks += 1; c->pipe_hold = 2; }
            if (h.error != 0 || h.job.mode == MODE_DONE) break;
            // (a batch of windows that consumed no visit and asked for no new plan would be queued again forever:
            // the classic kernels take the rest of this sweep)
            if (h.job.pos == pos && !stalled && !piped) gram_skip = true;
            pos = h.job.pos;
            win = h.win_size > 0 ? h.win_size : win;
            rate = pos > 0 ? (double)h.n_moves / (double)pos : rate;
            if (pos > batch_pos0) recent_rate = (double)(h.n_moves - batch_moves0) / (double)(pos - batch_pos0);
            batch_pos0 = pos; batch_moves0 = h.n_moves;
            continue;
        }
        // the resolver's LDS plan depends on the number of labels: re-planned every chunk
        int res_R = 0, res_Kcap = 0, res_lds = 0;
        const bool use_resolver = c->resolver_mode == 2 &&
                                  resolve_plan(d, c->ctrl_host->job.K, &res_R, &res_Kcap, &res_lds);
        if (c->timing) { int rc = ensure_events(c, (size_t)T); if (rc) return rc; }
        // Which kernel set this batch of steps needs (bgmm_device.h: Dev::prune_enabled).  Far inside
        // the sparse-mover regime only the pruned-window kernels are queued, far inside the dense
        // one only the dense ones; in between both, and the device picks per window.
        int pmode = 0;
        if (use_prune) {
            const Ctrl &hc = *c->ctrl_host;
            const bool fresh = first_batch || hc.job.mode == MODE_FRESH;
            pmode = (hc.ema_run >= 4.0 * kPruneMinRun && fresh) ? 2 : (hc.ema_run < 0.5 * kPruneMinRun ? 0 : 1);
            if (c->prune_mode == 2) pmode = 2;       // (every window pruned: exact whatever the regime, for tests)
        }
        {
            const Ctrl &hc = *c->ctrl_host;
            const long long open_rows = first_batch ? (long long)d.batch_rows : hc.job.win_hi - hc.job.win_base;
            // (while nothing moves every clean window doubles the next: room for four doublings per batch)
            d.batch_rows = rows_for(win, open_rows, rate == 0.0 ? 16 : 8);
        }
        const long long grid_rows = d.batch_rows;
        if (pmode >= 1 && c->tables_robust) {
            // (a safe-stay batch left its robust bound constants in the pruning tables: valid, but looser)
            CK(c, hipMemsetAsync(&d.ctrl->tables_valid, 0, sizeof(int), st));
            c->tables_robust = false;
        }
        if (pmode != 2) lean = false;
        d.lean_step = lean ? 1 : 0;
        d.use_home = (d.cov_type == COV_FULL && c->kind == KERNEL_MFMA && c->home_pass) ? 1 : 0;
        // (a short residual list -- D <= 32: clusters a dozen sigma apart leave home_kernel a fraction of a per cent -- is settled
        //  by one dense launch; with certified stays on the sparse draw kernel also feeds the certificates, so not then)
        d.use_certify = use_certify ? 1 : 0;
        d.resid_dense = (d.use_home && !use_certify && d.Dp <= 32 && resid_dense_lds_bytes(d) <= 150 * 1024) ? 1 : 0;
        if ((pmode != 2 && !(pmode == 1 && c->home_mode == 3)) || !first_batch || !d.use_home || lean) short_step = false;
        d.short_step = short_step ? (d.order ? 2 : 1) : 0;
        d.publish = (lean || short_step) ? 1 : 0;
        if (short_step) T = 1;        // (one window is the whole sweep; a refused step is queued again in full)
        first_batch = false;
        d.prune_enabled = pmode;
        // (a forced batch cannot fall back to the dense kernels: keep it short while moves are seen)
        if (pmode == 2 && c->prune_mode != 2 && rate > 0.0 && T > lb + 64) T = (int)(lb + 64);
        for (int t = 0; t < T; ++t) {
            // With pruning on, fresh windows are scored by the pruning kernel and the plain kernel
            // only serves the re-scoring after a move; the events bracket the one that works in
            // the steady state.
            if (short_step) {
                // (the events bracket the kernels that stream the rows -- what bench.py's roofline names --, not the sort of
                //  a fresh visiting order in front of them)
                if (d.short_step == 2) launch_bucket_rows(d, grid_rows, st);
                if (c->timing) CK(c, hipEventRecord(c->ev0[t], st));
                launch_home(d, grid_rows, st);
                launch_resid_dense(d, st);
                if (c->timing) CK(c, hipEventRecord(c->ev1[t], st));
                launch_apply(d, st);
                continue;
            }
            if (pmode == 1) launch_score(d, c->kind, &d.ctrl->job, d.q, d.qstride, -1, grid_rows, 1, st);
            if (pmode >= 1 && !lean) launch_prune_tables(d, st);
            if (pmode >= 1 && !lean && !use_certify) launch_bucket_rows(d, grid_rows, st);
            if (c->timing) CK(c, hipEventRecord(c->ev0[t], st));
            if (pmode >= 1 && use_certify) launch_certify(d, grid_rows, st);
            if (pmode >= 1 && !lean && use_certify) launch_bucket_rows(d, grid_rows, st);
            if (pmode >= 1 && !lean && d.use_home) { launch_home(d, grid_rows, st); launch_resid_dense(d, st); }
            if (pmode >= 1) { if (!lean) launch_score_pruned(d, &d.ctrl->job, d.q, d.qstride, grid_rows, st); }
            else launch_score(d, c->kind, &d.ctrl->job, d.q, d.qstride, -1, grid_rows, 0, st);
            if (c->timing) CK(c, hipEventRecord(c->ev1[t], st));
            if (pmode <= 1) launch_choice(d, grid_rows, st);
            if (pmode >= 1 && !lean) launch_choice_sparse(d, grid_rows, st);
            if (use_resolver && pmode <= 1) launch_resolve(d, res_R, res_Kcap, res_lds, st);
            launch_apply(d, st);
            if (!lean) launch_refresh_ctrl(d, st);        // (a lean step moves nothing: apply refuses it otherwise)
        }
        CK(c, hipGetLastError());
        if (phase == 3 && (lean || short_step) && !c->timing) {
            c->async_pending = true;
            c->async_short = short_step;
            return 2;
        }
        if (lean || short_step) {
            // (apply_kernel has left the control block in host memory: no copy in the queue)
            CK(c, hipStreamSynchronize(st));
            memcpy(c->ctrl_host, c->ctrl_pub, sizeof(Ctrl));
        } else {
            int rc = fetch_ctrl(c);
            if (rc) return rc;
        }
        const Ctrl &h = *c->ctrl_host;
        if (c->timing) {
            const long long worked = h.n_steps - steps_done;   // the first `worked` steps did work
            for (long long t = 0; t < worked && t < T; ++t) {
                float ms = 0.f;
                CK(c, hipEventElapsedTime(&ms, c->ev0[(size_t)t], c->ev1[(size_t)t]));
                c->timed_ms += (double)ms;
                c->timed_launches += 1;
            }
        }
        steps_done = h.n_steps;
        if (short_step) { if (h.retry_full) c->short_refused += 1; else c->short_stood += 1; }
        if (h.retry_full) {          // a lean step met something it could not certify (a short step: a mover, a visit
            lean = false;            // home_kernel could not decide, stale tables): full steps from here on
            short_step = false;
            c->ctrl_host->retry_full = 0;
            CK(c, hipMemcpy(&d.ctrl->retry_full, &c->ctrl_host->retry_full, sizeof(int), hipMemcpyHostToDevice));
        }
        if (h.error != 0 || h.job.mode == MODE_DONE) break;
        pos = h.job.pos;
        win = h.win_size > 0 ? h.win_size : win;
        rate = pos > 0 ? (double)h.n_moves / (double)pos : rate;
        if (pos > batch_pos0) recent_rate = (double)(h.n_moves - batch_moves0) / (double)(pos - batch_pos0);
        batch_pos0 = pos; batch_moves0 = h.n_moves;
    }
    if (getenv("BGMM_DEBUG_PIPE"))
        fprintf(stderr, "[bgmm] pipelined batches so far %lld, chains broken %lld; this sweep %lld windows\n", c->pipe_batches, c->pipe_breaks,
                (long long)c->ctrl_host->gram_windows);
    c->last_move_rate = (double)c->ctrl_host->n_moves / (double)(N > 0 ? N : 1);
    const Ctrl &h = *c->ctrl_host;
    c->stats[0] = h.lik_evals; c->stats[1] = h.n_moves; c->stats[2] = h.n_windows;
    c->stats[3] = h.n_steps; c->stats[4] = h.n_score_launches; c->stats[5] = h.n_scored;
    c->stats[6] = (long long)h.n_kept_blocks; c->stats[7] = (long long)h.n_bound_blocks;
    c->prune_mfma = (long long)h.n_prune_mfma;
    c->certified = (long long)h.n_certified;
    c->stats2[0] = (long long)h.n_pairs_exact; c->stats2[1] = h.gram_windows; c->stats2[2] = h.gram_rows_total;
    c->stats2[3] = h.home_in - h.home_out;          // visits home_kernel decided on its own
    c->totals[0] += 1; c->totals[1] += h.lik_evals; c->totals[2] += h.n_moves; c->totals[3] += (long long)h.n_pairs_exact;
    c->safe_stats[0] = h.safe_windows; c->safe_stats[1] = h.safe_scanned; c->safe_stats[2] = h.safe_rows;
    c->safe_stats[3] = h.safe_cuts; c->safe_stats[4] = (long long)(1e6 * (c->safe_cap_user > 0.0 ? c->safe_cap_user : h.safe_cap));
    c->safe_stats[5] = h.safe_L;
    // home_kernel pays while the table bound decides most visits (well separated components); when it had to
    // pass most of them on, the next sweep goes straight to the pruning kernel -- and tries again every 64th sweep
    if (c->home_mode) c->home_pass = c->home_mode != 2;
    else if (h.home_in > 0) c->home_pass = 2 * h.home_out < h.home_in;
    else if (!c->home_pass && (++c->home_retry & 63) == 0) c->home_pass = true;
    c->moves_prev = h.n_moves;
    c->lean_ok = use_certify && !partial && h.n_moves == 0 && h.n_certified == (unsigned long long)N;
    c->short_ok = use_prune && !use_certify && !partial && h.n_moves == 0 && h.n_steps == 1 && h.n_windows == 1 &&
                  h.home_in == (long long)N && h.home_out == 0;
    return check_device_error(c);
}

extern "C" int bgmm_sweep_staged(bgmm_ctx *c, int32_t use_power, double power) { return sweep_impl(c, use_power, power, 0); }

// The staged sweep in two halves, so that a driver can prepare the NEXT sweep's inputs (bgmm_stage_* calls: host work, a
// look-ahead hit is a memcmp) while this one runs.  _begin queues the sweep; when its first batch of launches is all a
// chain at rest needs (a lean step with certified stays, a short step without), it returns without waiting.  _end waits,
// and finishes whatever is left (a refused step is redone in full) exactly as bgmm_sweep_staged would have.  Every other
// kind of sweep runs to its end inside _begin.  Between the two only bgmm_stage_* calls are allowed; the look-ahead
// generations they would start are started by _end (the running sweep may still read the buffers they write).
extern "C" int bgmm_sweep_staged_begin(bgmm_ctx *c, int32_t use_power, double power) {
    if (!c) return BGMM_EINVAL;
    const int rc = sweep_impl(c, use_power, power, 3);
    c->async_rc = rc == 2 ? 0 : rc;
    return c->async_rc;
}

// Waits for the sweep bgmm_sweep_staged_begin left in the queue and finishes it: a refused step is redone in full FIRST
// (from the inputs the sweep was begun with), and only then are the look-ahead generations started that the stage calls of
// the meantime put off -- they write buffers the redo may still read.
static int finish_pending(bgmm_ctx *c) {
    if (!c->async_pending) return c->async_rc;
    CK(c, hipSetDevice(c->device));
    CK(c, hipStreamSynchronize(c->stream));
    memcpy(c->ctrl_host, c->ctrl_pub, sizeof(Ctrl));     // (lean and short steps publish the control block to host memory)
    c->async_pending = false;
    if (c->async_short) { if (c->ctrl_host->retry_full) c->short_refused += 1; else c->short_stood += 1; }
    if (c->ctrl_host->retry_full) {
        c->ctrl_host->retry_full = 0;
        CK(c, hipMemcpy(&c->d.ctrl->retry_full, &c->ctrl_host->retry_full, sizeof(int), hipMemcpyHostToDevice));
    }
    const int rs = sweep_impl(c, c->d.use_power, c->d.power, 4);
    int rc = 0;
    if (c->defer_mt) {
        c->defer_mt = false;
        rc = mt_schedule(c, c->defer_mt_hit, c->defer_mt_key.data(), c->defer_mt_pos);
    }
    if (rc == 0 && c->defer_perm) {
        c->defer_perm = false;
        PermPtrs P;
        rc = perm_ensure(c, P);
        if (rc == 0) rc = perm_schedule(c, P);
    }
    c->async_rc = rs ? rs : rc;
    return c->async_rc;
}

extern "C" int bgmm_sweep_staged_end(bgmm_ctx *c) {
    if (!c) return BGMM_EINVAL;
    return finish_pending(c);
}

// Sweeps of several chains that live on ONE device, side by side.  Chains that can take the one-workgroup sweep (D <= 4,
// full covariance, automatic tuning, labels within the LDS plan) are opened and swept by two launches for all of them
// -- one workgroup, one compute unit per chain -- instead of two launches and a host round trip each; every other chain
// is swept on its own as bgmm_sweep_staged would.  Same trajectories as separate calls.
extern "C" int bgmm_group_sweep_staged(bgmm_ctx *const *ctxs, int32_t n, const int32_t *use_power, const double *power,
                                       int32_t *rc_out) {
    if (!ctxs || n < 1 || !rc_out) return BGMM_EINVAL;
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return BGMM_EINVAL;
        rc_out[i] = 0;
        for (int j = 0; j < i; ++j)
            if (ctxs[j] == ctxs[i]) return fail(ctxs[i], BGMM_EINVAL, "a context appears twice in the group");
    }
    std::vector<int> deferred;
    int worst = 0;
    // Chains that can never take the one-workgroup sweep (D > 4, diagonal / fixed covariance) run their whole sweep as
    // bgmm_sweep_staged would -- but CONCURRENTLY, each on its own stream, each driven by its own host thread (a context is
    // one host thread's at a time; distinct contexts share nothing).  What bounds such a sweep while the chain still moves is
    // a latency chain that keeps ONE workgroup busy (kernels_gram.hip: the resolver); G chains side by side keep G of them
    // busy, and the wide kernels of one chain (cross forms, rebuilds) run beside the resolvers of the others.
    std::vector<std::thread> workers;
    std::vector<int> threaded;
    for (int i = 0; i < n; ++i)
        if (!seq_shape(ctxs[i])) threaded.push_back(i);
    GramCombiner comb;
    if (threaded.size() >= 2) {
        comb.slots.resize(threaded.size());
        for (size_t k = 0; k < threaded.size(); ++k) {
            comb.slots[k].c = ctxs[threaded[k]];
            ctxs[threaded[k]]->combiner_slot = (int)k;
        }
        // (std::thread's constructor may throw -- no exception may cross the C ABI, least of all with joinable workers
        //  behind it: the chains whose thread could not be started are declared DONE for the rendezvous, so that nobody
        //  waits for them, and swept on this thread)
        size_t started = 0;
        try {
            workers.reserve(threaded.size());
            for (int i : threaded) {
                const int up = use_power ? use_power[i] : 0;
                const double pw = (up && power) ? power[i] : 1.0;
                workers.emplace_back([=, &comb]() {
                    ctxs[i]->combiner = &comb;
                    rc_out[i] = sweep_impl(ctxs[i], up, pw, 0);
                    comb.declare(ctxs[i]->combiner_slot, GramCombiner::DONE);
                    ctxs[i]->combiner = nullptr;
                });
                ++started;
            }
        } catch (...) {
        }
        for (size_t k = started; k < threaded.size(); ++k) comb.declare((int)k, GramCombiner::DONE);
        for (size_t k = started; k < threaded.size(); ++k) {
            const int i = threaded[k];
            const int up = use_power ? use_power[i] : 0;
            rc_out[i] = sweep_impl(ctxs[i], up, (up && power) ? power[i] : 1.0, 0);
        }
    } else {
        threaded.clear();
    }
    for (int i = 0; i < n; ++i) {
        if (std::find(threaded.begin(), threaded.end(), i) != threaded.end()) continue;
        const int up = use_power ? use_power[i] : 0;
        const int rc = sweep_impl(ctxs[i], up, (up && power) ? power[i] : 1.0, 1);
        if (rc == 1) deferred.push_back(i);
        else { rc_out[i] = rc; if (rc < 0 && worst == 0) worst = rc; }
    }
    // one pair of launches per (device, D, LDS plan) among the chains that wait
    std::vector<char> done(deferred.size(), 0);
    for (size_t a = 0; a < deferred.size(); ++a) {
        if (done[a]) continue;
        bgmm_ctx *lead = ctxs[deferred[a]];
        std::vector<int> grp;
        for (size_t b = a; b < deferred.size(); ++b) {
            bgmm_ctx *o = ctxs[deferred[b]];
            if (!done[b] && o->device == lead->device && o->d.D == lead->d.D && o->grp_cap == lead->grp_cap) {
                grp.push_back(deferred[b]);
                done[b] = 1;
            }
        }
        const int m = (int)grp.size();
        hipError_t e = hipSetDevice(lead->device);
        if (e == hipSuccess && lead->grp_devs_cap < m) {
            if (lead->grp_devs) (void)hipFree(lead->grp_devs);
            lead->grp_devs = nullptr; lead->grp_devs_cap = 0;
            e = hipMalloc((void **)&lead->grp_devs, sizeof(Dev) * (size_t)m);
            if (e == hipSuccess) lead->grp_devs_cap = m;
        }
        std::vector<Dev> views((size_t)m);
        for (int k = 0; k < m && e == hipSuccess; ++k) {
            views[(size_t)k] = ctxs[grp[(size_t)k]]->d;
            // (what phase 1 queued on the chain's own stream -- a new seating table, stale factors rebuilt -- has to be there)
            e = hipStreamSynchronize(ctxs[grp[(size_t)k]]->stream);
        }
        hipStream_t st = lead->stream;
        if (e == hipSuccess) e = hipMemcpyAsync(lead->grp_devs, views.data(), sizeof(Dev) * (size_t)m, hipMemcpyHostToDevice, st);
        if (e == hipSuccess) {
            launch_sweep_begin(lead->d, st, lead->grp_devs, m);
            if (!launch_sweep_seq(lead->d, lead->grp_cap, st, lead->grp_devs, m)) e = hipErrorLaunchFailure;
        }
        if (e == hipSuccess) e = hipGetLastError();
        for (int k = 0; k < m && e == hipSuccess; ++k) {
            bgmm_ctx *o = ctxs[grp[(size_t)k]];
            e = hipMemcpyAsync(o->ctrl_host, o->d.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, st);
        }
        if (e == hipSuccess) e = hipStreamSynchronize(st);      // (views is pageable: the copy above has been staged by now)
        for (int k = 0; k < m; ++k) {
            bgmm_ctx *o = ctxs[grp[(size_t)k]];
            int rc;
            if (e != hipSuccess) {
                o->err = std::string("group sweep: ") + hipGetErrorString(e);
                rc = BGMM_EDEVICE;
            } else {
                rc = sweep_impl(o, o->d.use_power, o->d.power, 2);
            }
            rc_out[grp[(size_t)k]] = rc;
            if (rc < 0 && worst == 0) worst = rc;
        }
    }
    for (auto &w : workers) w.join();
    if (getenv("BGMM_DEBUG_GROUP") && !threaded.empty())
        fprintf(stderr, "[bgmm] group sweep: %zu chains on threads, %lld shared batches of frozen-factor windows, %.1f chains each\n",
                threaded.size(), comb.shared_batches, comb.shared_batches ? (double)comb.shared_members / (double)comb.shared_batches : 0.0);
    for (int i : threaded)
        if (rc_out[i] < 0 && worst == 0) worst = rc_out[i];
    return worst;
}

extern "C" int bgmm_sweep_resident(bgmm_ctx *c, int32_t index, int32_t use_power, double power) {
    if (!c) return BGMM_EINVAL;
    if (index < 0 || index >= c->res_n) return fail(c, BGMM_EINVAL, "resident sweep index out of range");
    c->cur_u = c->res_u + (size_t)index * c->d.N;
    c->cur_zero_u = c->res_zero_u[(size_t)index] != 0;
    c->order_is_perm = c->res_perm[(size_t)index] != 0;
    c->cur_order = c->res_order ? c->res_order + (size_t)index * c->d.N : nullptr;
    return bgmm_sweep_staged(c, use_power, power);
}

extern "C" int bgmm_sweep(bgmm_ctx *c, const int64_t *order, const double *u, int32_t use_power, double power) {
    int rc = bgmm_stage_sweep_inputs(c, order, u);
    if (rc) return rc;
    return bgmm_sweep_staged(c, use_power, power);
}

extern "C" int bgmm_get_K(bgmm_ctx *c, int32_t *K) {
    if (!c || !K) return BGMM_EINVAL;
    SETTLE(c);
    CK(c, hipSetDevice(c->device));
    int rc = fetch_ctrl(c);
    if (rc) return rc;
    *K = c->ctrl_host->job.K;
    return 0;
}

extern "C" int bgmm_get_assignments(bgmm_ctx *c, int64_t *z_out) {
    if (!c || !z_out) return BGMM_EINVAL;
    SETTLE(c);
    CK(c, hipSetDevice(c->device));
    long long *dz;
    CK(c, hipMalloc((void **)&dz, sizeof(long long) * c->d.N));
    launch_labels(c->d, dz, nullptr, c->stream);
    hipError_t e = hipMemcpyAsync(z_out, dz, sizeof(long long) * c->d.N, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dz);
    CK(c, e);
    return 0;
}

extern "C" int bgmm_get_counts(bgmm_ctx *c, int64_t *counts_out) {
    if (!c || !counts_out) return BGMM_EINVAL;
    SETTLE(c);
    CK(c, hipSetDevice(c->device));
    int rc = fetch_ctrl(c);
    if (rc) return rc;
    const int K = c->ctrl_host->job.K;
    long long *dc;
    CK(c, hipMalloc((void **)&dc, sizeof(long long) * (c->d.K_max + 1)));
    launch_labels(c->d, nullptr, dc, c->stream);
    hipError_t e = hipMemcpyAsync(counts_out, dc, sizeof(long long) * K, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dc);
    CK(c, e);
    return 0;
}

extern "C" int bgmm_get_stats(bgmm_ctx *c, double *m_out, double *S_out, double *logdet_out, double *inv_out) {
    if (!c) return BGMM_EINVAL;
    SETTLE(c);
    CK(c, hipSetDevice(c->device));
    int rc = fetch_ctrl(c);
    if (rc) return rc;
    const int K = c->ctrl_host->job.K, D = c->d.D;
    if (K == 0) return 0;
    const size_t DD = c->d.cov_type != COV_FULL ? (size_t)D : (size_t)D * D;
    double *dm = nullptr, *dS = nullptr, *dl = nullptr, *di = nullptr;
    hipError_t e = hipSuccess;
    if (m_out && e == hipSuccess) e = hipMalloc((void **)&dm, sizeof(double) * K * D);
    if (S_out && e == hipSuccess) e = hipMalloc((void **)&dS, sizeof(double) * K * DD);
    if (logdet_out && e == hipSuccess) e = hipMalloc((void **)&dl, sizeof(double) * K);
    if (inv_out && e == hipSuccess) e = hipMalloc((void **)&di, sizeof(double) * K * DD);
    if (e == hipSuccess) {
        launch_export_stats(c->d, K, dm, dS, dl, di, c->stream);
        if (dm) e = hipMemcpyAsync(m_out, dm, sizeof(double) * K * D, hipMemcpyDeviceToHost, c->stream);
        if (dS && e == hipSuccess) e = hipMemcpyAsync(S_out, dS, sizeof(double) * K * DD, hipMemcpyDeviceToHost, c->stream);
        if (dl && e == hipSuccess) e = hipMemcpyAsync(logdet_out, dl, sizeof(double) * K, hipMemcpyDeviceToHost, c->stream);
        if (di && e == hipSuccess) e = hipMemcpyAsync(inv_out, di, sizeof(double) * K * DD, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    (void)hipFree(dm); (void)hipFree(dS); (void)hipFree(dl); (void)hipFree(di);
    CK(c, e);
    return 0;
}

extern "C" int bgmm_get_log_prior(bgmm_ctx *c, double *out) {
    if (!c || !out) return BGMM_EINVAL;
    CK(c, hipSetDevice(c->device));
    CK(c, hipMemcpyAsync(out, c->d.log_prior, sizeof(double) * c->d.N, hipMemcpyDeviceToHost, c->stream));
    CK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int bgmm_log_marg(bgmm_ctx *c, double *out) {
    if (!c || !out) return BGMM_EINVAL;
    SETTLE(c);
    CK(c, hipSetDevice(c->device));
    launch_log_marg(c->d, c->util_out, c->util_out + 8, c->stream);
    CK(c, hipMemcpyAsync(out, c->util_out, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int bgmm_log_marg_k(bgmm_ctx *c, int32_t k, double *out) {
    if (!c || !out) return BGMM_EINVAL;
    SETTLE(c);
    CK(c, hipSetDevice(c->device));
    int rc = fetch_ctrl(c);
    if (rc) return rc;
    if (k < 0 || k >= c->ctrl_host->job.K) return fail(c, BGMM_EINVAL, "component index out of range");
    launch_log_marg(c->d, c->util_out, c->util_out + 8, c->stream);
    CK(c, hipMemcpyAsync(out, c->util_out + 8 + k, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    CK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int bgmm_contingency(bgmm_ctx *c, const int64_t *true_idx, int32_t K_true, int64_t *table_out) {
    if (!c || !table_out || K_true < 1) return BGMM_EINVAL;
    SETTLE(c);
    CK(c, hipSetDevice(c->device));
    if (!true_idx && !c->true_dev) return fail(c, BGMM_EINVAL, "no reference labelling uploaded yet");
    int rc = fetch_ctrl(c);
    if (rc) return rc;
    const int K = c->ctrl_host->job.K;
    if (K == 0) return 0;
    const size_t cells = (size_t)K_true * K;
    // the reference labelling stays on the device between calls (true_idx == NULL: the same as last time)
    if (!c->true_dev) CK(c, hipMalloc((void **)&c->true_dev, sizeof(long long) * c->d.N));
    if (true_idx)
        CK(c, hipMemcpyAsync(c->true_dev, true_idx, sizeof(long long) * c->d.N, hipMemcpyHostToDevice, c->stream));
    if (cells > c->table_cells) {
        if (c->table_dev) (void)hipFree(c->table_dev);
        c->table_dev = nullptr; c->table_cells = 0;
        CK(c, hipMalloc((void **)&c->table_dev, sizeof(unsigned long long) * cells));
        c->table_cells = cells;
    }
    CK(c, hipMemsetAsync(c->table_dev, 0, sizeof(unsigned long long) * cells, c->stream));
    launch_contingency(c->d, c->true_dev, K_true, c->table_dev, c->stream);
    CK(c, hipMemcpyAsync(table_out, c->table_dev, sizeof(long long) * cells, hipMemcpyDeviceToHost, c->stream));
    CK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int bgmm_cluster_dispersion(bgmm_ctx *c, double *out) {
    if (!c || !out) return BGMM_EINVAL;
    SETTLE(c);
    CK(c, hipSetDevice(c->device));
    int rc = fetch_ctrl(c);
    if (rc) return rc;
    const int K = c->ctrl_host->job.K;
    if (K == 0) return 0;
    launch_dispersion(c->d, c->util_out, c->stream);
    CK(c, hipMemcpyAsync(out, c->util_out, sizeof(double) * K, hipMemcpyDeviceToHost, c->stream));
    CK(c, hipStreamSynchronize(c->stream));
    return 0;
}

extern "C" int bgmm_log_post_pred(bgmm_ctx *c, int64_t i, double *out) {
    if (!c || !out) return BGMM_EINVAL;
    SETTLE(c);
    CK(c, hipSetDevice(c->device));
    if (i < 0 || i >= c->d.N) return fail(c, BGMM_EINVAL, "data index out of range");
    int rc = fetch_ctrl(c);
    if (rc) return rc;
    const int K = c->ctrl_host->job.K;
    if (K == 0) return 0;
    Job job;
    std::memset(&job, 0, sizeof(job));
    job.pos = i; job.win_base = i; job.win_hi = i + 1; job.mode = MODE_FRESH; job.K = K;
    job.chunks = K < kMaxChunks ? K : kMaxChunks;
    CK(c, hipMemcpyAsync(c->util_job, &job, sizeof(Job), hipMemcpyHostToDevice, c->stream));
    CK(c, hipStreamSynchronize(c->stream));
    Dev d = c->d;
    d.order = nullptr;
    launch_score(d, c->kind, c->util_job, c->util_q, 1, -1, 1, 0, c->stream);
    launch_post_pred(d, c->util_q, c->util_out, c->stream);
    CK(c, hipMemcpyAsync(out, c->util_out, sizeof(double) * K, hipMemcpyDeviceToHost, c->stream));
    CK(c, hipStreamSynchronize(c->stream));
    return 0;
}

static int item_op(bgmm_ctx *c, int op, int64_t i, int32_t k) {
    if (!c) return BGMM_EINVAL;
    SETTLE(c);
    CK(c, hipSetDevice(c->device));
    if (i < 0 || i >= c->d.N) return fail(c, BGMM_EINVAL, "data index out of range");
    launch_item_op(c->d, op, i, k, c->stream);
    launch_refresh_ctrl(c->d, c->stream);
    int rc = fetch_ctrl(c);
    if (rc) return rc;
    rc = check_device_error(c);
    if (rc) {   // clear the sticky flag: the state was left untouched by a rejected op
        c->ctrl_host->error = 0;
        (void)hipMemcpy(&c->d.ctrl->error, &c->ctrl_host->error, sizeof(int), hipMemcpyHostToDevice);
    }
    c->assigned = true;
    c->moves_prev = -1;          // (the state changed behind the sweeps' back: the next sweep's caches are cold)
    c->short_ok = false;
    return rc;
}

extern "C" int bgmm_set_stats(bgmm_ctx *c, int32_t k, const double *m, const double *S, int64_t count) {
    if (!c || !m || !S) return BGMM_EINVAL;
    SETTLE(c);
    CK(c, hipSetDevice(c->device));
    int rc = fetch_ctrl(c);
    if (rc) return rc;
    if (k < 0 || k >= c->ctrl_host->job.K) return fail(c, BGMM_EINVAL, "component index out of range");
    if (count < 1 || count > c->d.N) return fail(c, BGMM_EINVAL, "count must be in 1 .. N");
    const int D = c->d.D;
    const size_t DD = c->d.cov_type == COV_FULL ? (size_t)D * D : (c->d.cov_type == COV_FIXED ? (size_t)2 * D : (size_t)D);
    double *dm = nullptr;
    CK(c, hipMalloc((void **)&dm, sizeof(double) * (D + DD)));
    hipError_t e = hipMemcpyAsync(dm, m, sizeof(double) * D, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dm + D, S, sizeof(double) * DD, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        launch_set_stats(c->d, k, dm, dm + D, (int)count, c->stream);
        launch_refresh_ctrl(c->d, c->stream);
        e = hipStreamSynchronize(c->stream);
    }
    (void)hipFree(dm);
    CK(c, e);
    rc = fetch_ctrl(c);
    if (rc) return rc;
    rc = check_device_error(c);
    if (rc) {   // (a matrix that is not positive definite: the flag is cleared, the statistics stay as given)
        c->ctrl_host->error = 0;
        (void)hipMemcpy(&c->d.ctrl->error, &c->ctrl_host->error, sizeof(int), hipMemcpyHostToDevice);
    }
    c->moves_prev = -1;
    c->lean_ok = false;
    c->short_ok = false;
    return rc;
}

extern "C" int bgmm_get_raw_stats(bgmm_ctx *c, int32_t k, double *m_out, double *S_out) {
    if (!c || !m_out || !S_out) return BGMM_EINVAL;
    SETTLE(c);
    CK(c, hipSetDevice(c->device));
    int rc = fetch_ctrl(c);
    if (rc) return rc;
    if (k < 0 || k >= c->ctrl_host->job.K) return fail(c, BGMM_EINVAL, "component index out of range");
    const int D = c->d.D;
    const size_t DD = c->d.cov_type == COV_FULL ? (size_t)D * D : (c->d.cov_type == COV_FIXED ? (size_t)2 * D : (size_t)D);
    double *dm = nullptr;
    CK(c, hipMalloc((void **)&dm, sizeof(double) * (D + DD)));
    launch_raw_stats(c->d, k, dm, dm + D, c->stream);
    hipError_t e = hipMemcpyAsync(m_out, dm, sizeof(double) * D, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(S_out, dm + D, sizeof(double) * DD, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(dm);
    CK(c, e);
    return 0;
}

extern "C" int bgmm_del_component(bgmm_ctx *c, int32_t k) {
    if (!c) return BGMM_EINVAL;
    SETTLE(c);
    CK(c, hipSetDevice(c->device));
    int rc = fetch_ctrl(c);
    if (rc) return rc;
    if (k < 0 || k >= c->ctrl_host->job.K) return fail(c, BGMM_EINVAL, "component index out of range");
    launch_del_component(c->d, k, c->stream);
    CK(c, hipGetLastError());
    CK(c, hipStreamSynchronize(c->stream));
    c->moves_prev = -1;
    c->lean_ok = false;
    c->short_ok = false;
    return 0;
}

extern "C" int bgmm_set_sweep_visits(bgmm_ctx *c, int64_t n_visits) {
    if (!c) return BGMM_EINVAL;
    SETTLE(c);
    if (n_visits < 0 || n_visits > c->d.N) return fail(c, BGMM_EINVAL, "n_visits must be in 0 .. N");
    c->next_sweep_visits = n_visits;
    return 0;
}

extern "C" int bgmm_set_label(bgmm_ctx *c, int64_t i, int32_t k) {
    if (!c) return BGMM_EINVAL;
    SETTLE(c);
    CK(c, hipSetDevice(c->device));
    if (i < 0 || i >= c->d.N) return fail(c, BGMM_EINVAL, "data index out of range");
    int rc = fetch_ctrl(c);
    if (rc) return rc;
    if (k < -1 || k >= c->ctrl_host->job.K) return fail(c, BGMM_EINVAL, "component index out of range");
    launch_set_label(c->d, i, k, c->stream);
    CK(c, hipStreamSynchronize(c->stream));
    c->moves_prev = -1;
    c->lean_ok = false;
    c->short_ok = false;
    return 0;
}

extern "C" int bgmm_add_item(bgmm_ctx *c, int64_t i, int32_t k) { return item_op(c, 1, i, k); }
extern "C" int bgmm_del_item(bgmm_ctx *c, int64_t i) { return item_op(c, 0, i, 0); }

extern "C" int bgmm_get_phase_clocks(bgmm_ctx *c, int64_t *out16) {
    if (!c || !out16) return BGMM_EINVAL;
    SETTLE(c);
    CK(c, hipSetDevice(c->device));
    int rc = fetch_ctrl(c);
    if (rc) return rc;
    for (int t = 0; t < 16; ++t) out16[t] = c->ctrl_host->prof[t];
    return 0;
}

extern "C" int bgmm_get_sweep_stats(bgmm_ctx *c, int64_t *out8) {
    if (!c || !out8) return BGMM_EINVAL;
    SETTLE(c);
    for (int t = 0; t < 8; ++t) out8[t] = c->stats[t];
    return 0;
}

extern "C" int bgmm_get_prune_stats(bgmm_ctx *c, int64_t *out4) {
    if (!c || !out4) return BGMM_EINVAL;
    SETTLE(c);
    out4[0] = c->stats[6]; out4[1] = c->stats[7]; out4[2] = c->prune_mfma; out4[3] = c->certified;
    return 0;
}

extern "C" int bgmm_get_path_stats(bgmm_ctx *c, int64_t *out4) {
    if (!c || !out4) return BGMM_EINVAL;
    SETTLE(c);
    for (int t = 0; t < 4; ++t) out4[t] = c->stats2[t];
    return 0;
}

extern "C" int bgmm_get_safe_stats(bgmm_ctx *c, int64_t *out6) {
    if (!c || !out6) return BGMM_EINVAL;
    SETTLE(c);
    for (int t = 0; t < 6; ++t) out6[t] = c->safe_stats[t];
    return 0;
}

extern "C" int bgmm_get_proof_pass_stats(bgmm_ctx *c, int64_t *out2) {
    if (!c || !out2) return BGMM_EINVAL;
    SETTLE(c);
    out2[0] = c->proof_batches[0];
    out2[1] = c->proof_batches[1];
    return 0;
}

extern "C" int bgmm_mt19937_chain_blocks(void) { return mt19937_chain_blocks(); }

extern "C" int bgmm_mt19937_jump_poly(int32_t chain, uint32_t *coef624) {
    if (chain < 1 || chain > 4096 || !coef624) return BGMM_EINVAL;
    std::vector<unsigned> coef;
    if (!mt19937_jump_coefficients(chain + 1, coef)) return fail(nullptr, BGMM_EUNSUPPORTED, "the generator's characteristic polynomial could not be established");
    memcpy(coef624, coef.data() + (size_t)chain * 624, sizeof(unsigned) * 624);
    return 0;
}

extern "C" int bgmm_set_mt_jump(bgmm_ctx *c, int32_t enabled) {
    if (!c) return BGMM_EINVAL;
    SETTLE(c);
    c->mt_jump_on = enabled != 0;
    return 0;
}

extern "C" int bgmm_set_window_pipeline(bgmm_ctx *c, int32_t enabled) {
    if (!c) return BGMM_EINVAL;
    SETTLE(c);
    c->pipe_mode = enabled ? 1 : 0;
    return 0;
}

extern "C" int bgmm_get_window_pipeline_stats(bgmm_ctx *c, int64_t *out4) {
    if (!c || !out4) return BGMM_EINVAL;
    out4[0] = c->pipe_batches; out4[1] = c->pipe_breaks; out4[2] = c->pipe_mode; out4[3] = c->pipe_hold;
    return 0;
}

extern "C" int bgmm_set_proof_lookahead(bgmm_ctx *c, int32_t chunk_visits) {
    if (!c || chunk_visits < 0) return BGMM_EINVAL;
    SETTLE(c);
    int v = 0;
    if (chunk_visits > 0) { v = 1024; while (v < chunk_visits && v < (1 << 20)) v <<= 1; }
    c->ahead_chunk = v;
    return 0;
}

extern "C" int bgmm_get_proof_lookahead_stats(bgmm_ctx *c, int64_t *out4) {
    if (!c || !out4) return BGMM_EINVAL;
    SETTLE(c);
    CK(c, hipSetDevice(c->device));
    int rc = fetch_ctrl(c);
    if (rc) return rc;
    const Ctrl &h = *c->ctrl_host;
    out4[0] = h.ah_served; out4[1] = h.ah_self; out4[2] = h.ah_dirty; out4[3] = h.ah_chunks;
    return 0;
}

extern "C" int bgmm_set_proof_pass(bgmm_ctx *c, int32_t kind) {
    if (!c || kind < -1 || kind > 1) return BGMM_EINVAL;
    SETTLE(c);
    c->safe_dense_pin = kind;
    return 0;
}

extern "C" int bgmm_set_safe_budget(bgmm_ctx *c, double cap) {
    if (!c || !(cap >= 0.0) || cap > 8.0) return BGMM_EINVAL;
    SETTLE(c);
    c->safe_cap_user = cap;
    return 0;
}

extern "C" int bgmm_set_kernel_timing(bgmm_ctx *c, int32_t enabled) {
    if (!c) return BGMM_EINVAL;
    SETTLE(c);
    c->timing = enabled != 0;
    c->timed_launches = 0;
    c->timed_ms = 0.0;
    return 0;
}

extern "C" int bgmm_get_kernel_timing(bgmm_ctx *c, int64_t *n_launches, double *total_ms) {
    if (!c) return BGMM_EINVAL;
    if (n_launches) *n_launches = c->timed_launches;
    if (total_ms) *total_ms = c->timed_ms;
    return 0;
}

extern "C" int bgmm_set_tuning(bgmm_ctx *c, int32_t max_window, int32_t kernel_kind, int32_t resolver_mode,
                               int32_t prune_mode) {
    if (!c) return BGMM_EINVAL;
    SETTLE(c);
    CK(c, hipSetDevice(c->device));
    if (kernel_kind < 0 || kernel_kind > 2) return fail(c, BGMM_EINVAL, "kernel_kind must be 0, 1 or 2");
    if (resolver_mode < 0 || resolver_mode > 5) return fail(c, BGMM_EINVAL, "resolver_mode must be 0 .. 5");
    if (prune_mode < 0 || prune_mode > 3) return fail(c, BGMM_EINVAL, "prune_mode must be 0 .. 3");
    c->kernel_kind = kernel_kind;
    c->resolver_mode = resolver_mode;
    c->prune_mode = prune_mode;
    CK(c, hipMemcpy(&c->d.ctrl->dense_mode, &resolver_mode, sizeof(int), hipMemcpyHostToDevice));
    resolve_kind(c);
    if (max_window > 0) {
        int rc = fetch_ctrl(c);
        if (rc) return rc;
        int cap = max_window < 64 ? 64 : max_window;
        if (cap > c->win_rows) cap = c->win_rows;
        c->ctrl_host->win_cap = cap;
        if (c->ctrl_host->win_size > cap) c->ctrl_host->win_size = cap;
        CK(c, hipMemcpy(&c->d.ctrl->win_cap, &c->ctrl_host->win_cap, sizeof(int), hipMemcpyHostToDevice));
        CK(c, hipMemcpy(&c->d.ctrl->win_size, &c->ctrl_host->win_size, sizeof(int), hipMemcpyHostToDevice));
    }
    return 0;
}

extern "C" int bgmm_set_home_pass(bgmm_ctx *c, int32_t mode) {
    if (!c || mode < 0 || mode > 3) return BGMM_EINVAL;
    SETTLE(c);
    c->home_mode = mode;
    c->home_pass = mode != 2;
    return 0;
}

extern "C" int bgmm_set_seq_plan(bgmm_ctx *c, int32_t max_labels) {
    if (!c || max_labels < 0) return BGMM_EINVAL;
    SETTLE(c);
    c->seq_cap = max_labels;
    return 0;
}

extern "C" int bgmm_synchronize(bgmm_ctx *c) {
    if (!c) return BGMM_EINVAL;
    SETTLE(c);
    CK(c, hipSetDevice(c->device));
    CK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// ------------------------------------------------------------------------------------------
// Final label gather of independent chains (include/bgmm.h): RCCL through dlopen, so that the
// library carries no link-time dependency on it (and shares the copy a host process already loaded).
// ------------------------------------------------------------------------------------------
struct Id128 { char b[128]; };                          // ncclUniqueId (passed by value)
namespace {
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id128, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
}  // namespace
static Rccl g_rccl;
static std::mutex g_rccl_mutex;          // (chains driven from threads meet here on first use)

static int rccl_load() {
    std::lock_guard<std::mutex> guard(g_rccl_mutex);
    if (g_rccl.lib) return 0;
    // The copy that belongs to THIS library's HIP runtime (the one next to the libamdhip64 we are linked against): a
    // process may carry another RCCL built against another runtime (PyTorch bundles both), and streams and device
    // pointers of one runtime mean nothing to the other.
    void *h = nullptr;
    Dl_info info;
    if (dladdr((void *)&hipGetDeviceCount, &info) && info.dli_fname) {
        std::string dir(info.dli_fname);
        const size_t slash = dir.rfind('/');
        if (slash != std::string::npos) h = dlopen((dir.substr(0, slash) + "/librccl.so.1").c_str(), RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return fail(nullptr, BGMM_EDEVICE, "librccl.so.1 not found (multi-chain gather needs RCCL)");
    Rccl r;
    r.lib = h;
    r.GetUniqueId = (int (*)(void *))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (int (*)(void **, int, Id128, int))dlsym(h, "ncclCommInitRank");
    r.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(h, "ncclAllGather");
    r.CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
    r.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.AllGather || !r.CommDestroy)
        return fail(nullptr, BGMM_EDEVICE, "librccl.so.1 lacks the nccl* entry points");
    g_rccl = r;
    return 0;
}

static int rccl_fail(bgmm_ctx *c, const char *what, int code) {
    std::string msg = std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(code) : "RCCL error");
    return fail(c, BGMM_EDEVICE, msg.c_str());
}

extern "C" int bgmm_comm_unique_id(void *id128_out) {
    if (!id128_out) return BGMM_EINVAL;
    int rc = rccl_load();
    if (rc) return rc;
    const int e = g_rccl.GetUniqueId(id128_out);
    return e == 0 ? 0 : rccl_fail(nullptr, "ncclGetUniqueId", e);
}

extern "C" int bgmm_comm_create(int32_t rank, int32_t world_size, const void *id128, int32_t device, void **comm_out) {
    if (!id128 || !comm_out || world_size < 1 || rank < 0 || rank >= world_size) return BGMM_EINVAL;
    int rc = rccl_load();
    if (rc) return rc;
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, BGMM_EDEVICE, "hipSetDevice failed");
    (void)hipGetLastError();                            // (RCCL reports a stale error of the thread as its own)
    Id128 id;
    memcpy(id.b, id128, sizeof(id.b));
    void *comm = nullptr;
    const int e = g_rccl.CommInitRank(&comm, world_size, id, rank);
    if (e != 0) return rccl_fail(nullptr, "ncclCommInitRank", e);
    *comm_out = comm;
    return 0;
}

extern "C" int bgmm_gather_labels(bgmm_ctx *c, void *comm, int32_t world_size, int64_t *z_all_out) {
    if (!c || !comm || !z_all_out || world_size < 1) return BGMM_EINVAL;
    SETTLE(c);
    int rc = rccl_load();
    if (rc) return rc;
    CK(c, hipSetDevice(c->device));
    const size_t N = (size_t)c->d.N;
    long long *dz = nullptr, *dall = nullptr;
    CK(c, hipMalloc((void **)&dz, sizeof(long long) * N));
    hipError_t e = hipMalloc((void **)&dall, sizeof(long long) * N * (size_t)world_size);
    if (e != hipSuccess) { (void)hipFree(dz); CK(c, e); }
    launch_labels(c->d, dz, nullptr, c->stream);
    (void)hipGetLastError();
    const int ne = g_rccl.AllGather(dz, dall, N, /* ncclInt64 */ 4, comm, c->stream);
    if (ne == 0) {
        e = hipMemcpyAsync(z_all_out, dall, sizeof(long long) * N * (size_t)world_size, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    (void)hipFree(dz); (void)hipFree(dall);
    if (ne != 0) return rccl_fail(c, "ncclAllGather", ne);
    CK(c, e);
    return 0;
}

extern "C" int bgmm_comm_destroy(void *comm) {
    if (!comm) return BGMM_EINVAL;
    int rc = rccl_load();
    if (rc) return rc;
    const int e = g_rccl.CommDestroy(comm);
    return e == 0 ? 0 : rccl_fail(nullptr, "ncclCommDestroy", e);
}

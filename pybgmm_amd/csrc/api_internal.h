// Host side of libbgmm_hip.so, shared by its parts (round 6: bgmm_api.hip split by concern):
//   api_context.hip   the context: create / destroy, state in and out, tuning, statistics, the method-level entry points
//   api_inputs.hip    a sweep's inputs: staged / resident streams, the caller's Mersenne Twister continued on the device
//   api_perm.hip      np.random.permutation on the device: one generation, the single look-ahead, generations in flight
//   api_sweep.hip     the per-sweep launch schedule (sweep_impl), frozen-factor batches, the staged sweep in two halves
//   api_group.hip     chains side by side: the rendezvous of their host threads, shared launches, bgmm_group_sweep_staged
//   api_comm.hip      the final label gather (RCCL, loaded on first use)
// Everything here is internal (the shared object exports the extern "C" entry points of include/bgmm.h and nothing else).
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
#pragma once
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
long long refresh_ws_doubles(int D);
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

extern thread_local std::string g_create_error;
// The permutations' worker thread (PermPipe::worker) reports through a string of its own: bgmm_ctx::err belongs to the
// thread that drives the context.  CK / fail write to err_of(ctx).
extern thread_local std::string *g_err_sink;

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

// The device copy of X may serve several contexts (bgmm_create_shared: chains side by side over one data set): it is freed by
// whoever lets go of it last.
struct SharedX {
    double *p = nullptr;
    std::atomic<int> refs{1};
};

struct bgmm_ctx {
    int device = 0;
    SharedX *xshare = nullptr;       // the data matrix on the device (owned through its reference count)
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
        int rounds_fixed = [] { const int v = bgmm_dev_option("perm_chain_rounds", 0);
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
    long long grp_stats[4] = {0, 0, 0, 0};   // bgmm_get_group_stats
    Dev *grp_pdevs = nullptr;        // ... and of the views of a pipelined shared batch (two per chain: buffer sets 0 / 1)
    int grp_pdevs_cap = 0;
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
    int pipe_mode = 1;               // 0: never (plain windows) -- bgmm_set_window_pipeline
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
    // 0 / 1: pinned (bgmm_set_proof_pass, for experiments and tests)
    int safe_dense_pin = -1;         // (bgmm_set_proof_pass)
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
inline int dalloc(bgmm_ctx *c, T **p, size_t count) {
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

int finish_pending(bgmm_ctx *c);
// Entry points that read or change what a sweep left in the queue by bgmm_sweep_staged_begin is working on finish that
// sweep first (bgmm_sweep_staged_end then just reports its status).
#define SETTLE(c) do { if ((c)->async_pending) { const int rc_ = finish_pending(c); if (rc_) return rc_; } } while (0)

inline int fail(bgmm_ctx *c, int code, const std::string &msg) {
    if (c) err_of(c) = msg;
    else g_create_error = msg;
    return code;
}

inline const char *err_text(int code) {
    switch (code) {
        case -3: return "K_max exceeded: a new component was drawn while all K_max slots are in use";
        case -4: return "a component scatter matrix is not positive definite";
        case -1: return "invalid label";
        default: return "device-side error";
    }
}

inline int check_device_error(bgmm_ctx *c) {
    // ctrl_host must be current
    if (c->ctrl_host->error != 0) {
        const int e = c->ctrl_host->error;
        return fail(c, e == -3 ? BGMM_EKMAX : e == -4 ? BGMM_ENOTPD : BGMM_EINVAL, err_text(e));
    }
    return 0;
}

inline int fetch_ctrl(bgmm_ctx *c) {
    CK(c, hipMemcpyAsync(c->ctrl_host, c->d.ctrl, sizeof(Ctrl), hipMemcpyDeviceToHost, c->stream));
    CK(c, hipStreamSynchronize(c->stream));
    return 0;
}

inline void resolve_kind(bgmm_ctx *c) {
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

// ---- shared between the parts ---------------------------------------------------------------------------------------------
inline bool seq_shape(const bgmm_ctx *c) { return c->d.cov_type == COV_FULL && c->d.D <= 4; }
int sweep_impl(bgmm_ctx *c, int32_t use_power, double power, int phase);            // api_sweep.hip
int ensure_events(bgmm_ctx *c, size_t n);
int mt_ensure_tables(bgmm_ctx *c, int chains);                                      // api_inputs.hip
int mt_wait_batches(bgmm_ctx *c);
int mt_schedule(bgmm_ctx *c, bool hit, const uint32_t *key, int pos);
int perm_pipe_drain(bgmm_ctx *c);                                                   // api_perm.hip
int perm_ensure(bgmm_ctx *c, PermPtrs &P);
int perm_schedule(bgmm_ctx *c, const PermPtrs &P);
struct GramCombiner;                                                                // api_group.hip
// (0: queued with the group's plain windows, 2: with its pipelined windows, 1: queue it yourself, < 0: error)
// kind 0: a batch of frozen-factor windows (pipe_T of them could be pipelined from visit pos on), 1: of safe-stay steps with
// the dense proof pass and no look-ahead
int combiner_submit(bgmm_ctx *c, int T, int pipe_T, long long pos, int kind);
void gram_point_view(bgmm_ctx *c, Dev &v, int par);      // api_sweep.hip: the window buffers of set `par` into a view
void combiner_declare_busy(bgmm_ctx *c);

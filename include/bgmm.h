/*
 * libbgmm_hip.so -- C-ABI of the MI355X (gfx950) collapsed-Gibbs reassignment path for the
 * CRP / pCRP Gaussian mixture model with a normal-inverse-Wishart prior.
 *
 * The reference (junlulocky/PyBGMM) is pure Python and has no FFI layer; the interface this
 * library replaces is the Python class surface of the hot path.  Each entry point below cites
 * the reference code it stands in for (file:line relative to the reference checkout).  A host
 * binds it with ctypes; pybgmm_amd/_lib.py is that binding.
 *
 * Conventions
 *   - plain C, no exceptions; every call returns 0 on success or a negative BGMM_E* code,
 *     with a human-readable message available from bgmm_last_error().
 *   - all floating point is IEEE float64; labels / counts cross the boundary as int64
 *     (the reference's platform int), data indices as int64.
 *   - host pointers are borrowed only for the duration of the call.
 *   - one context = one chain on one GPU; a context is not thread safe, distinct contexts
 *     are independent (one per GPU for the multi-chain mode).
 *   - `v_0` must be integer valued: the reference indexes its log / lgamma tables with it
 *     (pybgmm/gaussian/gaussian_components.py:120-122, 238, 248).
 */
#ifndef BGMM_H
#define BGMM_H

#include <stdint.h>

/* The library is built with -fvisibility=hidden: only the entry points declared here are exported. */
#define BGMM_API __attribute__((visibility("default")))

#ifdef __cplusplus
extern "C" {
#endif

typedef struct bgmm_ctx bgmm_ctx;

enum {
    BGMM_OK = 0,
    BGMM_EINVAL = -1,    /* bad argument (shape, label vector, v_0 < D, ...)            */
    BGMM_EDEVICE = -2,   /* HIP runtime error (message holds hipGetErrorString)          */
    BGMM_EKMAX = -3,     /* a new component would exceed K_max (the reference raises
                            IndexError at the same point, gaussian_components.py:161-164) */
    BGMM_ENOTPD = -4,    /* a component scatter matrix lost positive definiteness         */
    BGMM_EUNSUPPORTED = -5
};

enum { BGMM_COV_FULL = 0,      /* covariance_type="full"  (igmm.py:104-105, gaussian/gaussian_components.py)      */
       BGMM_COV_DIAG = 1 };    /* covariance_type="diag"  (igmm.py:106-107, gaussian/gaussian_components_diag.py):
                                  S_0 is a D-vector; per-slot S / inverse blocks are D-vectors too; `logdet_out`
                                  of bgmm_get_stats is log_prod_vars, `inv_out` is inv_vars                       */
enum { BGMM_COV_FIXED = 2 };   /* covariance_type="fixed" (igmm.py:108-109, gaussian/gaussian_components_fixedvar.py):
                                  FixedVarPrior(var, mu_0, var_0) is passed as m_0 = mu_0, S_0 = [var[D] ; var_0[D]]
                                  (k_0, v_0 ignored: pass 1, 1); bgmm_get_stats returns mu_N_numerators,
                                  precision_Ns, log_prod_precision_preds, precision_preds                           */

/* Library / build identification, e.g. "bgmm-hip 0.1 gfx950". */
BGMM_API const char *bgmm_version(void);

/* Message of the last failing call on this context (or of a failed bgmm_create when ctx==NULL). */
BGMM_API const char *bgmm_last_error(const bgmm_ctx *ctx);

/*
 * GaussianComponents.__init__ + _cache (gaussian_components.py:75-127) without the
 * assignments: uploads X[N,D] (row major), the NIW prior (prior/niw.py:10-23) and alpha,
 * allocates K_max component slots, and evaluates the prior predictive of every point
 * (`cached_log_prior`, :125-127 via :207-214) on the device.
 *   lgamma_tab / log_tab: optional host tables of length v_0+N+2 holding lgamma(n/2) and
 *   log(n) for n = [1, 1, 2, ..., v_0+N+1] (exactly the reference's `_cached_gammaln_by_2`
 *   and `_cached_log_v`); pass NULL to have the library fill them with libm.
 */
BGMM_API int bgmm_create(bgmm_ctx **out, int device, int64_t N, int32_t D, int32_t K_max, int32_t cov_type,
                const double *X, const double *m_0, double k_0, int64_t v_0, const double *S_0,
                double alpha, const double *lgamma_tab, const double *log_tab);
/* A context over the SAME data matrix as `parent` (same device, N, D, covariance type): it borrows the parent's device copy
 * of X instead of uploading one -- chains side by side over one data set (the reference builds every sampler object over
 * the caller's one X array, igmm/igmm.py:68-76) hold it once: 8 chains at C5's shape 2 GB instead of 16.  Prior and K_max are
 * its own.  The copy lives until the last context using it is destroyed, in any order. */
BGMM_API int bgmm_create_shared(bgmm_ctx **out, bgmm_ctx *parent, int32_t K_max, const double *m_0, double k_0, int64_t v_0,
                                const double *S_0, double alpha, const double *lgamma_tab, const double *log_tab);

BGMM_API void bgmm_destroy(bgmm_ctx *ctx);

/*
 * Initial component assignment (gaussian_components.py:96-111): labels must be -1 or cover
 * 0..max consecutively (the reference asserts this, :103-105).  Sufficient statistics are
 * accumulated in the reference's order (k ascending, i ascending, starting from
 * S_0 + k_0 m_0 m_0^T) so that counts / m_N_numerators / S_N_partials are bit-identical.
 */
BGMM_API int bgmm_set_assignments(bgmm_ctx *ctx, const int64_t *z);

/*
 * One full Gibbs sweep = the body of `for i_iter` in CRPMM.collapsed_gibbs_sampler
 * (igmm/crpmm.py:57-88) / PCRPMM.collapsed_gibbs_sampler (igmm/pcrpmm.py:93-131).
 *   order : visiting order (np.random.permutation, pcrpmm.py:86-91) or NULL for 0..N-1
 *   u     : one uniform per visit, in visiting order (random.random() of utils/utils.py:15)
 *   power : pCRP exponent r; seating weight log(pow(n_k, r)) (pcrpmm.py:105-108).  Pass
 *           use_power=0 for the plain CRP weight log(n_k) (crpmm.py:70, pcrpmm.py:109-112).
 * Result: identical assignment trajectory to the reference for identical (order, u).
 */
BGMM_API int bgmm_sweep(bgmm_ctx *ctx, const int64_t *order, const double *u, int32_t use_power, double power);

/* Same sweep split in two so that the H2D copy of (order, u) can sit outside a timed region. */
BGMM_API int bgmm_stage_sweep_inputs(bgmm_ctx *ctx, const int64_t *order, const double *u);
BGMM_API int bgmm_sweep_staged(bgmm_ctx *ctx, int32_t use_power, double power);
/*
 * The staged sweep in two halves, for a driver that prepares the NEXT sweep's inputs while this one runs (a sweep of a
 * chain at rest is shorter than the host's way round the loop).  _begin queues the sweep; when its first batch of launches
 * is all a chain at rest needs -- a lean step with certified stays, a short step without -- it returns without waiting,
 * otherwise the sweep runs to its end inside _begin.  _end waits and finishes what is left (a refused step is redone in
 * full, from the inputs the sweep was begun with) exactly as bgmm_sweep_staged would have: begin + end ==
 * bgmm_sweep_staged, same trajectory.  Between the two, bgmm_stage_* calls stage the sweep AFTER this one: a request the
 * look-ahead can serve costs a comparison and returns at once (the generations it would start are started by _end, behind
 * the redo: the running sweep may still read the buffers they write); a request that has to be generated on the spot, host
 * inputs (bgmm_stage_sweep_inputs) and every other entry point that reads or changes the chain's state first finish the
 * sweep in flight (what _end would have done; _end then reports its status).  A second sweep call is refused.
 */
BGMM_API int bgmm_sweep_staged_begin(bgmm_ctx *ctx, int32_t use_power, double power);
BGMM_API int bgmm_sweep_staged_end(bgmm_ctx *ctx);

/*
 * Many chains per GPU (SURVEY.md section 5 `chains=`; 8e: chains are replicas, they exchange nothing).  The staged sweeps of
 * n contexts that live on one device, side by side: what bgmm_sweep_staged(ctxs[i], use_power[i], power[i]) would do for
 * each, with the same trajectories -- but the chains that can take the one-workgroup sweep of small dimensions (D <= 4, full
 * covariance, automatic tuning, a permutation as visiting order, labels within the LDS plan) are opened and swept by TWO
 * launches for all of them, one workgroup and one compute unit per chain, where separate calls pay two launches and a host
 * round trip per chain and leave 255 of the 256 compute units idle.  Every other chain of the group (D > 4, diagonal /
 * fixed covariance) runs its sweep as bgmm_sweep_staged would, CONCURRENTLY with the others: one stream and one host thread
 * of this call per chain (a sweep of a chain that still moves is one workgroup's latency chain: G of them keep G busy), and
 * chains of one shape that are in the frozen-factor regime at the same time -- burn-in from the reference's "rand" start --
 * share their launches (workgroup (x, chain) grids).  8 chains of BASELINE's C4 shape from "rand": 5.3 x one chain's moves/s.
 * use_power / power may be NULL (plain CRP weights).  rc_out[i] = the status of chain i; returns the first failure, or 0.
 * Contexts are not thread-safe: while the call runs, nobody else may touch the contexts handed to it.
 */
BGMM_API int bgmm_group_sweep_staged(bgmm_ctx *const *ctxs, int32_t n, const int32_t *use_power, const double *power,
                                     int32_t *rc_out);

/*
 * The sweep's uniforms continued ON THE DEVICE from the caller's Mersenne Twister: replaces the
 * N calls of random.random() (utils/utils.py:13-16) plus the upload.  key624 / pos are the 624 state
 * words and the position of random.getstate()[1]; on return they hold the state N calls of
 * random.random() leave behind (feed them to random.setstate()).  The doubles are bit-identical to
 * CPython's genrand_res53.  `order` as in bgmm_stage_sweep_inputs (NULL = identity).  Follow with
 * bgmm_sweep_staged.
 */
BGMM_API int bgmm_stage_mt19937(bgmm_ctx *ctx, const int64_t *order, uint32_t *key624, int32_t *pos);
/* The N uniforms currently staged for the next sweep (whichever way they got there). */
/* bgmm_stage_mt19937 cuts a long request into chains of 256 blocks (159 744 words) that run side by side from
 * jumped-ahead generator states (the state J words ahead is a GF(2) convolution of the next 20 560 words with the
 * coefficients of t^J modulo the generator's characteristic polynomial; the polynomials are built on the host once
 * per process).  0 = run the chains one after the other instead -- same doubles, same final state. */
BGMM_API int bgmm_set_mt_jump(bgmm_ctx *ctx, int32_t enabled);
/* Look-ahead (on by default): behind a bgmm_stage_mt19937 request the library generates, on a second stream and beside the
 * sweep that follows, the uniforms of the NEXT `sweeps` sweeps in one go (one request of sweeps x N doubles: the fixed
 * cost of a generation -- the chains' jumped-ahead seeds, a dozen launches -- is paid once per batch), and notes the
 * generator state at every sweep boundary inside it.  A later bgmm_stage_mt19937 call is served from the batch iff the
 * (key624, pos) it is handed equal, bit for bit, the state at the boundary the batch has reached, i.e. the caller drew
 * nothing from its generator since the previous call (the sampler loops of crpmm.py:57-88 / pcrpmm.py:93-131 never do);
 * otherwise the batch is thrown away and the request is generated on the spot as if there had been no look-ahead.  Same
 * doubles, same states handed back either way; costs two buffers of sweeps x N doubles.  sweeps: -1 = on, depth chosen
 * from N (about 8e6 doubles per batch, at most 8 sweeps; the default), 0 = off, 1 .. 8 = that many sweeps per batch.
 * out2 = {requests served by the look-ahead, requests generated on the spot}. */
BGMM_API int bgmm_set_mt_lookahead(bgmm_ctx *ctx, int32_t sweeps);
BGMM_API int bgmm_get_mt_lookahead_stats(bgmm_ctx *ctx, int64_t *out2);
/* The coefficient bits (19 937 of them, bit i = word i / 32, bit i % 32) of t^(chain * 159 744) modulo the characteristic
 * polynomial of MT19937: host arithmetic only, no device needed (what the CPU tests check against numpy's generator). */
BGMM_API int bgmm_mt19937_jump_poly(int32_t chain, uint32_t *coef624);
/* 624-word blocks per chain (the J of the polynomials above is this many blocks). */
BGMM_API int bgmm_mt19937_chain_blocks(void);
BGMM_API int bgmm_get_staged_uniforms(bgmm_ctx *ctx, double *u_out);
/*
 * The visiting order of a pCRP sweep, `np.random.permutation(range(N))` (pcrpmm.py:86-91), drawn ON THE DEVICE from the
 * caller's legacy numpy generator: key624 / pos are the 624 state words and the position of
 * np.random.get_state()[1:3] (or a RandomState's); on return they hold the state the call of permutation() would have
 * left (feed them to set_state() together with the untouched Gaussian cache fields).  The permutation is bit-identical
 * to numpy's (legacy shuffle: for i = N-1 .. 1, j = random_interval(i) by masked rejection, swap) and becomes the
 * visiting order of the NEXT sweep: follow with bgmm_stage_mt19937 / bgmm_stage_sweep_inputs with order = NULL (which
 * then means "the staged one", once) and bgmm_sweep_staged.  N >= 4096; BGMM_EUNSUPPORTED (state untouched) otherwise, or
 * in the astronomically unlikely case that 2 N + 1248 words do not suffice -- draw it on the host then.
 * bgmm_get_staged_order: the order the next sweep will use (tests).
 */
BGMM_API int bgmm_stage_permutation_mt19937(bgmm_ctx *ctx, uint32_t *key624, int32_t *pos);
BGMM_API int bgmm_get_staged_order(bgmm_ctx *ctx, int64_t *order_out);
/* out4 = {permutations served by the look-ahead, generated on the spot, rounds of draws the last one took to settle, the
 * most any took so far} (kernels_perm.hip: the rounds are queued blindly, a fixed number at a time). */
BGMM_API int bgmm_get_permutation_stats(bgmm_ctx *ctx, int64_t *out4);
/* The permutations in flight (three generations of np.random.permutation queued ahead on three streams): out4 = {set up,
 * switched off for this context, generations in a row its worker thread could not queue, bytes of its word stream}.  Its
 * memory (the word stream: 32 generations of 2 N + 1248 words, at most 1 GiB and at most a twentieth of the memory free
 * when it is set up; 3 target arrays + 4 order buffers of N; pinned verdict blocks) is taken at the first staged
 * permutation.  When that fails -- or the worker fails three times in a row -- everything it took is released, the state
 * is latched "off" and the stage calls go on with the single look-ahead (one permutation ahead, one stream): a slower
 * route to the same bits, not an error. */
BGMM_API int bgmm_get_permutation_pipe_state(bgmm_ctx *ctx, int64_t *out4);

/* Bench / multi-sweep form: make the inputs of n_sweeps sweeps resident in HBM at once
 * (u_all[n_sweeps][N]; order_all[n_sweeps][N] or NULL), then run sweep `index` of them with no
 * host-to-device traffic inside the call. */
BGMM_API int bgmm_upload_streams(bgmm_ctx *ctx, int32_t n_sweeps, const double *u_all, const int64_t *order_all);
BGMM_API int bgmm_sweep_resident(bgmm_ctx *ctx, int32_t index, int32_t use_power, double power);

/* IGMM.log_marg (igmm/igmm.py:199-215) = CRP log P(z) + sum_k log_marg_k
 * (gaussian_components.py:253-289). */
BGMM_API int bgmm_log_marg(bgmm_ctx *ctx, double *out);
/* GaussianComponents.log_marg_k(k) (gaussian_components.py:253-276). */
BGMM_API int bgmm_log_marg_k(bgmm_ctx *ctx, int32_t k, double *out);

/* components.K / .assignments / .counts[:K] (labels in the reference's numbering, i.e.
 * after its swap-with-last deletes, gaussian_components.py:188-205). */
BGMM_API int bgmm_get_K(bgmm_ctx *ctx, int32_t *K);
BGMM_API int bgmm_get_assignments(bgmm_ctx *ctx, int64_t *z_out);
BGMM_API int bgmm_get_counts(bgmm_ctx *ctx, int64_t *counts_out /* K entries */);

/* components.m_N_numerators[:K], S_N_partials[:K], logdet_covars[:K], inv_covars[:K]
 * (gaussian_components.py:86-89) in label order; any pointer may be NULL. */
BGMM_API int bgmm_get_stats(bgmm_ctx *ctx, double *m_out, double *S_out, double *logdet_out, double *inv_out);

/* components.cached_log_prior (gaussian_components.py:125-127), N entries. */
BGMM_API int bgmm_get_log_prior(bgmm_ctx *ctx, double *out);

/* GaussianComponents.log_post_pred(i) (gaussian_components.py:228-251): K entries. */
BGMM_API int bgmm_log_post_pred(bgmm_ctx *ctx, int64_t i, double *out);

/* GaussianComponents.add_item(i, k) / del_item(i) (gaussian_components.py:154-186). */
BGMM_API int bgmm_add_item(bgmm_ctx *ctx, int64_t i, int32_t k);
BGMM_API int bgmm_del_item(bgmm_ctx *ctx, int64_t i);

/*
 * GaussianComponents.restore_component_from_stats (gaussian_components.py:144-152): overwrite the
 * sufficient statistics of component k (label numbering of bgmm_get_stats) with m_N_numerator[D],
 * S_N_partial[D*D] (diag: [D]) and count.  The reference also copies the cached logdet / inverse; here
 * they are rebuilt from the statistics (bit-identical m / S, logdet / inverse to rounding).  Assignments are
 * not touched -- as in the reference, the caller keeps them consistent (cache, del_item, restore).
 * Not offered for covariance_type="fixed".
 */
BGMM_API int bgmm_set_stats(bgmm_ctx *ctx, int32_t k, const double *m, const double *S, int64_t count);
/* `components.assignments[i] = k` (igmm/crpmm.py:85): the label of point i alone, no statistics touched
 * (k = -1: unassigned).  With bgmm_set_stats it completes the reference's cache / del_item / restore idiom. */
BGMM_API int bgmm_set_label(bgmm_ctx *ctx, int64_t i, int32_t k);
/* The raw sufficient statistics of component k as the library keeps them -- what bgmm_set_stats takes back bit for bit
 * (checkpoint / resume): m[D] and S: D x D (full), D (diag), or for fixed-variance components
 * [precision_N[D] ; sum of x^2[D]] (the reference's class keeps no second moment: gaussian_components_fixedvar.py:75-90;
 * this library's log marginal needs it, so bgmm_set_stats takes 2 D values there). */
BGMM_API int bgmm_get_raw_stats(bgmm_ctx *ctx, int32_t k, double *m_out, double *S_out);
/* GaussianComponents.del_component(k) (gaussian_components.py:188-205) as a call of its own: label k is deleted by the
 * reference's swap with the last label.  The reference only ever calls it on a component that has just lost its last
 * member (:179-181); called on a component that still has members it would leave them labelled k -- i.e. silently
 * re-homed into what was the last component, whose statistics do not know them.  Here those members become unassigned
 * (-1) instead, so that counts, statistics and labels stay consistent. */
BGMM_API int bgmm_del_component(bgmm_ctx *ctx, int32_t k);
/* The `n_visits` of SURVEY.md 8b's bgmm_sweep: the NEXT sweep call (bgmm_sweep, bgmm_sweep_staged, bgmm_sweep_resident)
 * stops after the first n_visits visits of its order (0 or N: a whole sweep); later calls are whole sweeps again.  The
 * staged uniforms / order are indexed by visit as always: a following call that should continue needs its own inputs. */
BGMM_API int bgmm_set_sweep_visits(bgmm_ctx *ctx, int64_t n_visits);

/*
 * Per-sweep clustering metrics of the record dict (gmm/gmm.py:85-104), SURVEY.md 8f rank 2.
 *   bgmm_contingency: table[t * K + k] = #{i : true_idx[i] == t and label(i) == k}, the K_true x K
 *     contingency table from which mutual_information / normalized_mutual_information /
 *     information_variation (infopy/infopy.py:31-119) follow; true_idx holds 0..K_true-1 and is kept
 *     on the device: NULL means "the labelling of the previous call".
 *   bgmm_cluster_dispersion: out[k] = sum_{i in k} |x_i - mean_k|^2 from the component's sufficient
 *     statistics -- what utils.cluster_loss_inertia (utils/utils.py:31-88) takes the square root of.
 */
BGMM_API int bgmm_contingency(bgmm_ctx *ctx, const int64_t *true_idx, int32_t K_true, int64_t *table_out);
BGMM_API int bgmm_cluster_dispersion(bgmm_ctx *ctx, double *out /* K entries */);

/*
 * Measurement hooks.
 *   sweep_stats: counters of the last sweep --
 *     [0] lik_evals = sum over visits of K at that visit, [1] visits that changed component,
 *     [2] speculative windows evaluated, [3] kernel steps issued, [4] likelihood-kernel launches
 *     that did work, [5] rows x components scored (incl. re-scores after moves; pruned pairs
 *     count: they are decided, just not by the full quadratic form), [6] (16-visit block,
 *     component) pairs the pruning kernel evaluated in full, [7] pairs it only bounded.
 *   prune_stats: of the last sweep -- [0] = sweep_stats[6], [1] = sweep_stats[7], [2] the
 *     v_mfma_f64_16x16x4_f64 instructions (2048 flop each) the pruning kernel issued for its
 *     distance bounds and exact quadratic forms, [3] visits decided by certify_kernel (they provably
 *     keep their component: nothing scored, X not read).
 *   kernel timing: when enabled, every likelihood-kernel launch is bracketed by HIP events on
 *     the context's own stream; get returns the number of timed launches that did work and the
 *     sum of their durations in milliseconds since the last reset.
 */
BGMM_API int bgmm_get_sweep_stats(bgmm_ctx *ctx, int64_t *out8);
BGMM_API int bgmm_get_prune_stats(bgmm_ctx *ctx, int64_t *out4);
/*   path_stats: of the last sweep -- [0] (visit, component) pairs whose quadratic form was EXECUTED (dense
 *     windows: every pair; pruned windows: the pairs scored in full; frozen-factor windows: rows x
 *     (components + the prior); certified visits contribute nothing), [1] frozen-factor windows
 *     (the mover-dense path), [2] visits they consumed, [3] reserved. */
BGMM_API int bgmm_get_path_stats(bgmm_ctx *ctx, int64_t *out4);
/*   phase clocks: shader-clock ticks the one-workgroup kernels of the mover-dense path spent per phase,
 *     accumulated since bgmm_create -- all zero unless the library was built with -DBGMM_PROFILE
 *     (a development aid: every probe costs the kernel a global read-modify-write). */
BGMM_API int bgmm_get_phase_clocks(bgmm_ctx *ctx, int64_t *out16);
/*   safe_stats: of the last sweep -- safe-stay windows (crpmm.py:82-85: a visit that keeps its component changes
 *     nothing, so visits PROVEN to stay under the frozen state plus a budget of logged rank-1 terms are left off the
 *     sequential chain): [0] windows, [1] visits the proof pass examined, [2] unproven visits walked in order,
 *     [3] windows ended because a component ran out of budget, [4] the budget in force x 1e6, [5] visits the next
 *     proof pass will examine. */
BGMM_API int bgmm_get_safe_stats(bgmm_ctx *ctx, int64_t *out6);
/* Batches of safe-stay windows queued so far with [0] the proof pass through the per-home bound tables, [1] the DENSE proof
 * pass (every (visit, label) pair of a stretch through the likelihood kernel: chains whose clusters overlap, where the tables
 * prove nothing).  Which one runs never changes the chain, only its cost. */
BGMM_API int bgmm_get_proof_pass_stats(bgmm_ctx *ctx, int64_t *out2);
/* Frozen-factor windows of a chain on its own are PIPELINED by default (gram_finish of window w - 1 and the cross forms of
 * window w + 1 on a second stream beside the walk of window w, window w - 1's rank-1 terms carried into window w's cross
 * forms -- kernels_gram.hip): the same chain as plain windows, label for label (the reference's loop, igmm/crpmm.py:57-88,
 * knows neither).  bgmm_set_window_pipeline(ctx, 0) runs plain windows (tests A/B the two); the stats: out4 = {pipelined
 * batches queued, chains broken on the device (a window that ended early, opened or deleted a component, or failed: the
 * host reads the control block and runs two plain batches first), the mode, plain batches still to go}. */
BGMM_API int bgmm_set_window_pipeline(bgmm_ctx *ctx, int32_t enabled);
BGMM_API int bgmm_get_window_pipeline_stats(bgmm_ctx *ctx, int64_t *out4);
/* What this chain's batches were queued as inside bgmm_group_sweep_staged calls (chains of one shape that are in the same
 * regime together share their launches -- one launch per kernel for all of them, workgroup (x, chain)): out4 = {batches of
 * frozen-factor windows shared, of those pipelined, batches of safe-stay steps shared, batches of either kind this chain
 * queued on its own although it was in a group (nobody of its shape was there with it)}. */
BGMM_API int bgmm_get_group_stats(bgmm_ctx *ctx, int64_t *out4);
/* The dense proof pass of the safe-stay windows takes the exact quadratic forms of its (visit, label) pairs from a
 * LOOK-AHEAD: a second stream scores them a chunk of `chunk_visits` visits at a time (a power of two, default 8192; 0: off
 * -- every stretch scores its own pairs on the chain's stream), every slot, beside the resolver; a stretch re-scores only
 * the labels that took a rank-1 term since its chunk was requested (gaussian_components.py:154-205: a move touches two
 * components).  Same proofs, same chain; stats: out4 = {stretches served from the ring, stretches that scored themselves in
 * full, labels re-scored over the former, chunks requested} since the context was made. */
BGMM_API int bgmm_set_proof_lookahead(bgmm_ctx *ctx, int32_t chunk_visits);
BGMM_API int bgmm_get_proof_lookahead_stats(bgmm_ctx *ctx, int64_t *out4);
/* Which proof pass the next batches of safe-stay windows run: -1 = the chain decides (the default; BGMM_SAFE_DENSE in the
 * environment sets the initial value), 0 = always the per-home tables, 1 = always the dense pass.  May be changed between
 * sweeps (the tests switch kinds on one chain); it never changes the trajectory. */
BGMM_API int bgmm_set_proof_pass(bgmm_ctx *ctx, int32_t kind);
/* Budget of a safe-stay window per component: the sum over the rank-1 terms it takes of |log |D_t|| (D_t = the
 * Sherman-Morrison denominator of the term).  0 = follows the chain (the default); > 0 pins it.  A larger budget
 * lets a window absorb more (or more eccentric) moves and proves fewer visits; it never changes the trajectory. */
BGMM_API int bgmm_set_safe_budget(bgmm_ctx *ctx, double cap);
BGMM_API int bgmm_set_kernel_timing(bgmm_ctx *ctx, int32_t enabled);
BGMM_API int bgmm_get_kernel_timing(bgmm_ctx *ctx, int64_t *n_launches, double *total_ms);

/* Tuning knobs (0 keeps the default): cap on the speculative window; forced likelihood
 * kernel (0 auto, 1 VALU, 2 MFMA); mover path (0 auto: safe-stay windows while between one visit in
 * 65 536 and one in four moves, plain frozen-factor windows above that, 1 never: the per-mover kernel
 * chain, 2 the one-workgroup resolver that updates both factors in LDS whenever it fits (D <= 64),
 * 3 plain frozen-factor windows in every regime, 4 safe-stay windows in every regime, 5 as 0 without
 * safe-stay windows); exact pruning of components whose weight in a draw is provably
 * below e^-80 (0 auto: on while movers are sparse, plus certified stays in converged chains; 1 off;
 * 2 in every window whatever the regime -- slow when movers are dense, meant for tests; 3 as 0
 * but without certified stays, for measurements).  None of them changes the sampled trajectory, in this
 * sense: counts and sufficient statistics are bit-identical to the reference's in every mode, while the
 * floats of the predictive (rank-1 factor updates with a from-scratch rebuild every 64 steps: log
 * determinant to 1e-8 relative, inverse to 1e-7; wave-order sums; components dropped below e^-80 / e^-38 of
 * the best entering a draw as 0) differ from the reference's LAPACK route at the 1e-13 level, so a uniform
 * that lands within that distance of a boundary of the draw's cumulative distribution can pick the
 * neighbouring label -- in one mode and not in another, or against the reference.  That is an event of
 * probability ~1e-13 per visit; the golden trajectories, the soak runs (default configuration against plain
 * full evaluation) and the full-size tests have never met one.  prune_mode 1 with kernel_kind 1 (every pair
 * through the VALU kernel, dense draw kernel) is the configuration closest to the reference's arithmetic.
 * With everything on auto, full-covariance problems of D <= 4 whose staged visiting order is a
 * permutation (or absent) are swept by one workgroup that keeps the labels' state in LDS
 * (sweep_seq_kernel); forcing a kernel or a resolver mode, or prune_mode 2, selects the windowed
 * kernels instead.  In that path sweep_stats [2], [3] and [4] are 1. */
BGMM_API int bgmm_set_tuning(bgmm_ctx *ctx, int32_t max_window, int32_t kernel_kind, int32_t resolver_mode,
                    int32_t prune_mode);

/* Labels (components + 1) the one-workgroup small-D sweep keeps in LDS before it hands the sweep over to
 * the windowed kernels; 0 = as many as fit (the default).  A smaller plan changes where the hand-over
 * happens, never the trajectory (the tests use it to exercise the hand-over). */
BGMM_API int bgmm_set_seq_plan(bgmm_ctx *ctx, int32_t max_labels);

/* The first pass of a pruned window (home_kernel: visits whose only live candidates are the home component
 * and a new one).  0 = the context decides sweep by sweep from how many visits the pass settled in the last
 * sweep it ran (the default), 1 = always, 2 = never; 3 = always, and with certified stays off every sweep first tries a
 * short step (below) whether or not the last sweep suggests it will stand -- for the tests of its refusal.  It never
 * changes the trajectory. */
BGMM_API int bgmm_set_home_pass(bgmm_ctx *ctx, int32_t mode);
/* Sweeps with certified stays off (prune_mode 3, what bench.py times) on a chain at rest: when the previous sweep was one
 * pruned window, moved nothing and home_kernel decided every visit on its own, the next sweep queues a SHORT step --
 * home_kernel between sweep_begin and apply, none of the table / bucket / pruning / draw launches that would find nothing
 * to do.  apply_kernel lets it stand only if nothing moved, nothing was left on the residual list and the tables and the
 * sort it relied on were still valid; otherwise the same window is queued again with the full kernel set.
 * out2 = {short steps that stood, short steps refused} over the life of the context. */
BGMM_API int bgmm_get_short_step_stats(bgmm_ctx *ctx, int64_t *out2);
/* Since the context was made: out4 = {sweeps, (visit, component) pairs decided, moves, pairs whose quadratic form was
 * executed} -- what bgmm_get_sweep_stats / bgmm_get_path_stats report per sweep, summed (a measurement loop reads it once
 * at each end instead of after every sweep). */
BGMM_API int bgmm_get_totals(bgmm_ctx *ctx, int64_t *out4);

/* Blocks until all work queued on the context's stream has finished. */
BGMM_API int bgmm_synchronize(bgmm_ctx *ctx);

/*
 * Multi-chain final label gather (the reference has no counterpart: its chains would be
 * separate processes).  Chains are replicas -- nothing is exchanged during sampling; after the last sweep ONE
 * RCCL all-gather (over xGMI between the GPUs of a node) collects the final labels.  RCCL is loaded on first
 * use (dlopen of the librccl.so.1 that sits next to the HIP runtime this library is linked against -- not a copy
 * another runtime in the process may have brought along, e.g. PyTorch's bundled one), so single-chain users never
 * touch it.
 *   bgmm_comm_unique_id   128 bytes from ncclGetUniqueId: rank 0 makes them, the caller ships them to the other
 *                         ranks by whatever it has (a file, MPI, torch.distributed ...)
 *   bgmm_comm_create      ncclCommInitRank on `device` (the device of this rank's chain)
 *   bgmm_gather_labels    z_all[r * N + i] = label of point i in rank r's chain (int64, label numbering of
 *                         bgmm_get_assignments, -1 = unassigned), on every rank, in host memory
 *   bgmm_comm_destroy
 * Errors: BGMM_EDEVICE with the RCCL message in bgmm_last_error(ctx) (or (NULL) for the calls without a context).
 */
BGMM_API int bgmm_comm_unique_id(void *id128_out);
BGMM_API int bgmm_comm_create(int32_t rank, int32_t world_size, const void *id128, int32_t device, void **comm_out);
BGMM_API int bgmm_gather_labels(bgmm_ctx *ctx, void *comm, int32_t world_size, int64_t *z_all_out);
BGMM_API int bgmm_comm_destroy(void *comm);

#ifdef __cplusplus
}
#endif
#endif /* BGMM_H */

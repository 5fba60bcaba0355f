// Shared device/host definitions of libbgmm_hip.so (gfx950 only).
//
// Data layout in HBM (all float64 unless noted), one chain per context:
//   X[N][D]            row major data matrix, read-only after upload
//   log_prior[N]       prior predictive of every point ("open a new table" likelihood)
//   z[N]      int32    SLOT id of every point (-1 = unassigned).  Slots are stable storage
//                      locations; the reference's LABELS (which change on swap-with-last
//                      deletes) are the positions of `perm`: label j  <->  slot perm[j].
//   per slot s (K_max component slots + 1 pseudo slot K_max that holds the bare prior):
//     n[s]     int32   count
//     m[s][D]          k_0 m_0 + sum x           (reference m_N_numerators)
//     S[s][D][D]       S_0 + k_0 m_0 m_0^T + sum x x^T   (reference S_N_partials)
//     mu[s][D]         m / (k_0 + n)
//     Wrm[s][D][D]     inverse Cholesky factor of C = S - k_N mu mu^T (lower, row major)
//     Wfrag[s][nfrag][64]  the same matrix NEGATED, pre-swizzled into v_mfma_f64_16x16x4
//                      B-operand fragments (block-lower-triangular, zero padded to Dp)
//     cvec[s][Dp]      Winv * mu  (so that  Winv (mu - x) = cvec - Winv x)
//     sc[s]            scalar constants of the Student-t predictive (SlotConst)
//   q[nslots][Wmax]    quadratic forms (mu_s - x)^T C_s^{-1} (mu_s - x) of the current window,
//                      slot major so that both kernels touch it in full 64/128-byte segments
//                      A PRUNED window (kernels_prune.hip) reuses the same buffer block-sparse:
//                      qb[(block * nslots + label) * 16 + v] for the 16 visits of an evaluation
//                      block, only the (block, label) lines the pruning kernel kept are written,
//   keep64[Wmax/16][keep_stride]  bit `label` of a block's mask = "that line holds exact scores"
//   choice[Wmax] int32 drawn label of every visit of the current window
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <mutex>
#include <vector>

#define BGMM_MAX_D 256            /* full covariance.  Up to kFastMaxD = 128 a component's factor fits the LDS of a compute unit: the MFMA
                                     kernels, pruning, certified stays, frozen-factor and safe-stay windows; beyond it (round 6) the
                                     general route only: the VALU likelihood kernel, the per-mover kernel chain, rebuilds through a
                                     workspace in global memory (Dev::big_ws) -- correct, slow                                      */
#define BGMM_FAST_MAX_D 128
#define BGMM_MAX_D_DIAG 4096      /* diag / fixed: the state is a D-vector; above kDiagLdsMaxD the rows are read through the
                                     cache instead of an LDS tile (score_diag*_kernel)                                       */
#define BGMM_LOG_PI 1.1447298858494001741434273513530587116472948129153

enum { MODE_FRESH = 0, MODE_PARTIAL = 1, MODE_DONE = 2,
       MODE_SLOTS = 3,        // the likelihood kernel's list is the slots 0 .. n_dirty - 1 themselves (the proof pass's look-ahead)
       MODE_LIST = 4 };       // ... is Dev::slot_list[0 .. n_dirty)  (a stretch re-scored for the labels touched since)
enum { KERNEL_AUTO = 0, KERNEL_VALU = 1, KERNEL_MFMA = 2 };
enum { COV_FULL = 0, COV_DIAG = 1, COV_FIXED = 2 };
// how a slot's derived state (mu, Winv, cvec, constants) follows a change of (n, m, S)
enum { REFRESH_SCRATCH = 0,   // Cholesky + inverse of S_N from scratch, O(D^3)
       REFRESH_ADD = 1,       // rank-1 update of Winv: point refresh_i was added
       REFRESH_SUB = 2,       // rank-1 downdate: point refresh_i was removed
       REFRESH_NEW = 3 };     // new component: rank-1 update of the prior's factor (pseudo slot)
static constexpr int kRefreshEvery = 64;   // rank-1 steps per slot between from-scratch rebuilds
// ... in the frozen-factor windows (gram_finish_kernel: ~100 slots take a term or two per window, and the kernel lasts as
// long as its slowest workgroup -- a rebuild).  A window in which some slot is due rebuilds every touched slot beyond half
// of the interval with it, so that rebuilds come in bunches instead of one in every window.  (Each step is a product with
// a triangular factor close to the identity: errors add up, ~1e-16 per step.)
static constexpr int kGramRefreshEvery = 512;

static constexpr unsigned long long kNoMover = ~0ull;
static constexpr int kMaxChunks = 8;       // grid.y of the likelihood kernels
static constexpr int kValuRows = 64;       // rows (visits) per block, VALU likelihood kernel
static constexpr int kMfmaRows = 128;      // rows per block, MFMA likelihood kernel (4 waves x 2 x 16)
static constexpr int kChoiceRowsMax = 8;   // rows per block of the draw kernel (one wave per row)
// frozen-factor windows (kernels_gram.hip): rows per window, and the most terms (rank-1 changes of a
// component: one per removal, one per addition) a window can carry
static constexpr int kGramRows = 64;
static constexpr int kGramMaxTerms = 128;
static constexpr int kGramColSlack = kGramMaxTerms / 2 + 2;   // columns a window may open (new components + the prior)
// safe-stay windows: how far a column's count may drift from its frozen value inside one window (the bounds of
// the proof pass hold for every count in that range)
static constexpr int kHomeNbr = 4;        // neighbours of a home that home_kernel scores exactly (D <= 32)
static constexpr int kHomeBlock = 256;    // rows per workgroup step of home_kernel (4 waves x 64); its evaluation order is padded to this per home
static constexpr int kSafeDn = 16;
static constexpr int kSafeList = 4096;   // unproven visits a proof pass lists (a stretch ends at the next one)
static constexpr int kSafeResidSkip = 64; // a proof pass that leaves at most this many visits to the pruning kernel skips it (they are walked)
static constexpr int kSafeSmall = 8;     // labels with fewer members prove nothing for them (their bounds: kernels_safe.hip)

// Per-slot scalar constants.  (Diagonal covariance uses A = D*(lgamma terms) - 0.5 log prod var,
// half_vd = (v_N+1)/2, logdetC = sum log S_N,d, A1 = log prod var, the seating weights; rest 0.)
// As-is predictive of a point under slot s:
//   lp = A - half_vd * log(1 + q * inv_cv)
// "Home" predictive (the visited point removed from its own component; Sherman-Morrison
// on the frozen factor, with s = q the as-is quadratic form):
//   lp = A1 - 0.5*log(1 - a1*s) - half_vd1 * log(1 + coef1 * s / (1 - a1*s))
struct SlotConst {
    double A, half_vd, inv_cv;
    double A1, half_vd1, coef1, a1;
    double logdetC;
    double logseat, logseat1;   // log(n^r), log((n-1)^r)   (r = 1 for the plain CRP)
    double inv_lam;             // 1 / Lambda, Lambda >= lambda_max(S_N) (0: no bound known -> never pruned)
    double mu2;                 // |mu|^2
};

// What a likelihood / draw kernel works on.  First member of Ctrl so that (const Job*)ctrl
// is the live sweep job; utility calls (log_prior, log_post_pred) use a private Job.
struct Job {
    long long pos;        // first visit (or row) to process
    long long win_base;   // q/choice row r holds visit win_base + r
    long long win_hi;     // end (exclusive)
    int mode;             // MODE_FRESH: all K active labels; MODE_PARTIAL: dirty[] only
    int K;                // active labels
    int n_dirty;
    int dirty[2];
    int chunks;           // label chunks per row block (<= kMaxChunks)
    int prune;            // fresh window scored by the pruning kernel (valid only while nothing moves)
};

// A pruned window is thrown away at its first mover and pays ~13 launches per step, a dense one is
// re-scored for the two changed components only (~5 launches): pruning pays when the windows are
// mostly mover-free AND long (dense evaluation costs ~17 ns per row at K = 200, D = 64, the pruned one
// ~1 ns but 100-300 us more per step).  Mean distance between movers from which fresh windows are
// pruned (windows are about half of it):
constexpr double kPruneMinRun = 16384.0;

struct Ctrl {
    Job job;
    long long n_visits;
    unsigned long long first_mover;
    int n_refresh;
    int refresh[2];
    int refresh_kind[2];  // REFRESH_* : how the derived state of refresh[b] is rebuilt
    long long refresh_i;  // data index of the point whose move caused the refresh
    int win_size;         // current adaptive window size
    int win_cap;          // upper bound (host tuning; <= allocated rows of q)
    int error;
    int skip_apply;       // set by the resolver when it has consumed the step
    int dense_mode;       // tuning: 0 auto, 1 never use the resolver, 2 always when it fits
    double ema_run;       // running mean distance between movers
    long long last_mover;
    // counters of the current sweep
    long long lik_evals, n_moves, n_windows, n_steps, n_score_launches, n_scored;
    unsigned long long n_kept_blocks;   // (16-visit block, slot) pairs the pruning kernel scored in full
    unsigned long long n_bound_blocks;  // (16-visit block, slot) pairs it bounded
    int tables_valid;     // the pruned-window tables (pr_*) match the current means / labels / seating weights
    // the evaluation order (wperm, wrec) of window [wsort_base, wsort_hi) is still the bucket sort of
    // the current state (nothing moved since, same visiting order); skip_sort: the open window is that one
    int wsort_valid, skip_sort;
    int retry_full;       // a lean step (certify only) met a visit it could not certify: queue full steps
    long long state_epoch;  // bumped by every change of the sampler's state (move, rebuild, new seating weights)
    int n_sorted;         // rows of the open pruned window that went through the bucket sort (the uncertified ones)
    // home_kernel's evaluation order is PADDED (kernels_state.hip: bucket_prefix_kernel): every home's run of rows starts
    // at a multiple of kHomeBlock, the slots behind its last row hold dead records (i = -1) -- no block of kHomeBlock
    // rows holds two homes.  n_sorted_pad = extent of that layout (0: the sort was made unpadded); wsort_padded: the kept
    // sort (wsort_valid) is a padded one -- reusable only by a batch that wants the same layout
    int n_sorted_pad, wsort_padded;
    long long home_in, home_out;   // this sweep: rows home_kernel looked at / rows it had to pass on
    int n_resid;          // of those, the rows home_kernel could not decide (the pruning kernel's work list; kernels_home.hip)
    long long wsort_base, wsort_hi;
    unsigned long long n_prune_mfma;    // v_mfma_f64_16x16x4 instructions the pruning kernel issued
    unsigned long long n_certified;     // visits decided by certify_kernel (provably stay, nothing scored)
    long long prof[16];    // resolver phase clocks (setup, A, B, C, D1, D2, tail, calls), clock64 ticks
    // frozen-factor windows (kernels_gram.hip)
    int gram_nmoves;       // moves the last window logged (GramMove records)
    int gram_ntouched;     // live slots whose statistics / factor the finish kernel has to bring up to date
    int gram_stall;        // 1: the window needs more columns than are allocated (host reallocates)
    int pipe_break;        // pipelined windows: 1 = the chain of carried windows broke (a window ended early, a component was
                           // opened or deleted, an error): the rest of the batch stands still, the host goes on unpipelined
    int gram_rebuild;      // 1: a slot of the last window is due for a from-scratch rebuild (the others past half the interval join it)
    unsigned long long n_pairs_exact;   // (visit, component) pairs whose quadratic form was executed this sweep
    long long gram_rows_total, gram_windows;   // rows consumed by / number of frozen-factor windows this sweep
    // safe-stay windows (kernels_safe.hip): frozen-factor windows over the visits that cannot be PROVEN to stay
    int gl_n;              // rows of the open window (<= kGramRows): glist[gl_off .. gl_off + gl_n)
    int gl_off, gl_total;  // first listed visit not yet walked / listed visits of the stretch (<= kSafeList)
    int safe_epoch_valid;  // the stretch's proofs still stand: no proof pass in front of the next window.  They lapse when a
                           // component leaves its budget (cumulative over the stretch's windows: Dev::ep_state), drifts too far
                           // from the count it had at the proof pass, or is opened; at the end of the stretch; with a new sweep
    int safe_epoch_pad;
    long long gl_stretch_end, safe_epoch_pos0;   // the stretch the proof pass vouched for ends here / began here
    int safe_L;            // visits the next proof pass looks at (adapts to the density of unproven visits)
    long long gl_end;      // the window reaches up to (not including) this visit: everything before it that is not listed stays
    double safe_cap;       // the budget per component and window; follows the chain (gram_resolve_kernel) unless Dev::safe_cap pins it
    // the budget's controller (gram_resolve_kernel): what counts is how far a window gets -- a small budget ends windows
    // early, a large one proves fewer visits.  Phases of a dozen windows at the current budget alternate with half a
    // dozen at twice / half of it; the trial value is kept when its windows covered 15 % more visits each.
    // The budget is kept as a multiple of the mean |log |D_t|| of the terms seen (safe_wbar): that puts it at the scale of
    // the chain at hand ((D + 1) / n for a typical member that moves, near 1 for an outlier that leaves).
    double safe_mult, safe_mult_base, safe_wbar, safe_adv_sum, safe_base_rate;
    int safe_phase_cnt, safe_try, safe_next_dir, safe_pad;
    double safe_cap_built; // the budget the robust tables were built for -- the one the resolver enforces ...
    long long safe_epoch_built;   // ... and the state epoch: valid while both still match
    long long safe_resid_sum, safe_sorted_sum;   // this sweep, over the proof passes through the per-home tables: visits they had to leave to the exact forms / visits they looked at
    long long safe_windows, safe_scanned, safe_rows, safe_cuts;   // this sweep: windows, visits examined by the proof pass,
                                                                  // rows walked by the resolver, windows ended by the budget
    // Look-ahead of the DENSE proof pass (kernels_safe.hip, "look-ahead"): the exact quadratic forms of every (visit, slot)
    // pair are made a chunk of Dev::ahead_C visits at a time on a second stream, beside the resolver, into a ring of two
    // chunks (q row = visit & (2 C - 1)); a stretch then re-scores only the labels that took a rank-1 term since its
    // chunk's request.  Only the main stream's plan kernel writes these fields.
    long long win_seq;                 // frozen-factor windows closed since the context was made
    long long ah_chunk[2];             // per half of the ring: the chunk whose forms it holds (-1: none)
    long long ah_seq[2];               // [0] the ring is valid up to the windows closed by then (win_seq): a slot with touch_seq beyond
                                       // it is dirty; [1] 1: a plan has been handed to the second stream in this batch
    long long ah_lo[2];                // ... first visit of the chunk that was scored
    long long ah_req_chunk, ah_req_seq, ah_req_lo;   // the request the second stream is serving (-1: none)
    long long ah_hseq[2];              // (Dev::ahead_lazy) per half of the ring: the windows closed when its chunk was scored
    long long ah_served, ah_self, ah_dirty, ah_chunks;   // this sweep: stretches re-scored from the ring / scored in full,
                                                         // labels re-scored over the former, chunks requested
};

// One reassignment logged by a frozen-factor window, in visiting order: the finish kernel replays
// them slot by slot (bit-identical m, S: the same roundings in the same order as apply_rank1).
struct GramMove {
    long long i;          // data index
    int sub_slot;         // slot x_i leaves (-1: none, or the component was deleted with it)
    int add_slot;         // slot x_i joins (-1: none -- K_max exceeded)
    int add_init;         // 1: add_slot is a component opened by this move (starts from the prior)
    int pad;
};

// One row of a pruned window in evaluation order: everything the pruning kernel needs to start on
// it comes with a single 32-byte load (instead of order -> label -> prior chains).
struct WRec {
    long long i;        // data index
    int home;           // home slot (-1: unassigned)
    int home_label;     // its label in this window's frozen state
    double mlb0;        // log(alpha) + log_prior[i]: the "new table" score, first lower bound of the best score
    double u;           // the visit's uniform, as of the sweep the record was written in (bucket_scatter_kernel): good for
                        // the launches behind that scatter only -- an order kept across sweeps (skip_sort, safe-stay
                        // epochs) keeps the records and reads d.u
};

// What score_mfma_prune_kernel leaves per DATA POINT for certify_kernel: its exact quadratic form
// under its home component and its squared distance to that component's mean, tagged with
// (home slot << 32 | version of that slot's state).
struct PCache {
    long long tag;
    double qhome, rho2, pad;
};

// What choice_sparse_kernel leaves per data point: the log of the total weight of all alternatives
// relative to the home component's (the labels it scored exactly, each pruned one below e^-80 of the
// best score) as of state epoch `epoch` -- valid as long as NOTHING has changed since.
struct PCacheExact {
    long long epoch;
    double log_alt;
};

// Safe-stay windows: what a component has used of its budget since the last proof pass, and the counts between which
// that pass's proofs hold (kernels_safe.hip); per slot in global memory between windows, per column in LDS inside one.
// TWO accounts (round 4): log c_t(x, x) - log c_0(x, x) lies in [-W+, +W-] and logdet A_t - logdet A_0 = W+ - W- with W+ / W- the
// sums of |log |D_t|| over the terms that ADDED a member / REMOVED one.  A label's upper bound as somebody's alternative takes
// +W-/2 on the log-determinant and e^-W+ on the quadratic form, a home's lower bound -W+/2 and e^+W-: every bound uses one
// of the two, so each may run up to the budget on its own -- twice the movers per stretch when joins and leaves are mixed.
struct SafeCol {
    float w;               // W+: sum of |log |D_t|| over the terms that added a member (rounded up)
    short lo, hi;          // members it may still lose / gain (32767: no limit)
    float wm;              // W-: the same over the terms that removed one
    int pad;
};

struct Dev {
    long long N;
    int cov_type;                // 0 full covariance, 1 diagonal (S and dw are D-vectors per slot),
                                 // 2 fixed variance (m = mu_N numerators, S = [precision_N[D], sum x^2[D]],
                                 //   prior_m = precision_0 mu_0, prior_S = [precision_0[D], precision[D]])
    const double *fv_mu0;        // fixed variance: mu_0
    int D, Dp, K_max, nslots, nfrag, ldq;
    long long qstride;           // q[slot * qstride + window row]
    int choice_rows;             // visits per block of the draw kernel
    int rows_per_block;          // of the active likelihood kernel (chunk policy)
    int target_blocks;           // blocks that fill the chip at the kernel's occupancy
    long long tab_len, v0;
    double k0, alpha, log_alpha;
    const double *X;
    double *log_prior;
    int *z;
    const double *tab_lgam, *tab_log;
    // derived tables (device built): tabG[v] = Student-t normaliser for v degrees of freedom,
    // tabLogC[n] = log((k_N+1)/(k_N v_n)), tabSeat[n] = seating weight of a table with n guests
    const double *tabG, *tabLogC, *tabSeat;
    const double *prior_m, *prior_S;
    double *m, *S, *mu, *Wrm, *Wfrag, *cvec;
    double *dw;                  // diag: per-dimension weights 1 / (v_N var_d) of the univariate Student-t
    int *n;
    int *nupd;                   // rank-1 updates since the slot's last from-scratch refresh
    SlotConst *sc;
    // certified stays (kernels_prune.hip: certify_kernel): per slot a version of its derived state
    // (mean, factor); per data point the cached squared distance to its home's mean and its exact
    // quadratic form under its home, tagged (home slot << 32 | version)
    int *mu_ver;
    PCache *pcache;
    PCacheExact *pcache2;
    unsigned char *cert;         // pruned windows, per window row: 1 = certify_kernel proved that the visit stays
    double *ftab, *finv;         // per home label a: ftab[a][j] = upper bound of every other component's score for
                                 // a visit of a at distance <= j / finv[a] from a's mean, j = 0 .. 63
    // home_kernel scores the home's nearest neighbours EXACTLY where the table above cannot exclude them (clusters a
    // dozen sigma apart, D <= 32): nbr[a][0 .. kHomeNbr) = the labels whose bound at the largest tabulated radius is
    // highest, ascending (-1: none), ftab2[a][j] = the bound of ftab over every label but a and those
    int *nbr;
    double *ftab2;
    int seat_dirty;              // this sweep's seating weights differ from the last sweep's (exponent changed)
    int use_certify;             // 1: certify_kernel runs in front of the bucket sort (which then skips its rows)
    int lean_step;               // 1: this batch queues certify_kernel WITHOUT the pruning and draw kernels (the
                                 // previous sweep certified every visit); apply_kernel refuses the step otherwise
    int short_step;              // 1 / 2: this batch queues home_kernel (2: with the bucket sort in front) and apply_kernel only
                                 // -- no table, pruning or draw kernels: the previous sweep moved nothing and home_kernel decided
                                 // every visit on its own.  Stands iff the tables and the sort are still valid, nothing is left
                                 // on the residual list and nobody moves; apply_kernel refuses the step otherwise (retry_full)
    int publish;                 // 1: apply_kernel leaves a copy of the control block in host memory (ctrl_pub), so that
                                 // the host reads the outcome of a lean batch without a device-to-host copy in the queue
    Ctrl *ctrl_pub;              // host-pinned, device-mapped
    int *perm, *label_of_slot;
    Ctrl *ctrl;
    double *q;
    int *choice;
    int *bucket_bins;            // nslots + 2 counters of the per-window bucket sort
    int *bucket_end;             // [2][nslots + 2] of the last sort: end of bin b's live rows / start of the next bin (the pad between them)
    unsigned long long *keep64;  // pruned windows: per 16-visit block, bitmask over labels of the kept q lines
    int keep_stride;             // 64-bit words per block of keep64
    // pruned windows: per group of 16 labels (label-ordered, rebuilt at the start of every pruned
    // window by prune_tables_kernel): the means pre-swizzled into MFMA B fragments
    // pr_mufrag[G][Dp/4][64] (lane (lk, lr) of fragment kk = mu[perm[16G + lr]][4kk + lk]), the bound
    // constants pr_const[G][4][16] = {logseat + A, half_vd, inv_lam * inv_cv, |mu|^2} and the slot
    // ids pr_slot[G][16] (-1 beyond the last label)
    double *pr_mufrag, *pr_const;
    double *pr_dcc;              // pr_dcc[a * nslots + b] = |mu_a - mu_b| between LABELS a, b (coarse triangle bound)
    double *pr_rms;              // pr_rms[a] = sqrt(tr S_N / n) of label a (radius grid of its bound table)
    int *pr_slot;
    struct WRec *wrec;           // pruned windows: the k-th row in evaluation order (one 32-byte record)
    unsigned long long *pr_counts;  // 4 x 256 spread counters (kept, bound, MFMA instructions, certified visits) of the pruned-window kernels
    int *wperm;                  // pruned windows: k-th row in evaluation order -> window row (grouped by home)
    struct WRec *wrecR;          // the residual list home_kernel leaves (same record / row format)
    int *wpermR;
    int use_home;                // 1: home_kernel runs in front of the pruning kernel, which then works on the residual list
    int resid_dense;             // 1: resid_dense_kernel (kernels_resid.hip) is queued behind home_kernel: a SHORT residual list is
                                 // settled there, every label exactly, and the pruning / sparse draw kernels see an empty list
    const double *u;
    const long long *order;      // may be null (identity)
    int order_perm;              // 1: the visiting order visits every point exactly once (null, or a permutation)
    long long sweep_visits;      // visits of the sweep being queued (0 or N: all of them; bgmm_set_sweep_visits)
    int use_power;
    double power;
    int batch_rows;              // rows the launches of this batch of steps are sized for: no window
                                 // opened during the batch is longer (the host raises it batch by batch)
    // frozen-factor windows (kernels_gram.hip): per column c (label c of the window's frozen state; column K
    // = the bare prior) gC[c][r][r'] = a_c(x_r) . a_c(x_r') with a_c(x) = Winv_c (x - mu_c), r' >= r by
    // 16 x 16 tiles; gq0[c][r] = |a_c(x_r)|^2; glp0[r][c] / ge0[r][c] = frozen log score / exp(lp0 - M_r)
    double *gC, *gq0, *glp0, *ge0;
    double *gM;                  // [2][64]: the rows' reference points M_r, their new-table weights exp(lp_new - M_r)
    double *gcc;                 // [gcols][5][8]: per column, for the counts n0-2 .. n0+2 it can reach with up to two terms:
                                 // {1/k_N0, 1/k_N, k_N/(k_N+1), (v+D)/2, seat + Student-t constant - logdet0/2 - log(k_N0/k_N)/2,
                                 //  the count, logdet0, -} (kernels_gram.hip: what an update needs besides its loads)
    int gcols;                   // columns allocated (also the leading dimension of glp0 / ge0)
    GramMove *gmoves;            // [kGramMaxTerms]
    int *gtouched;               // [kGramMaxTerms]
    // What a window's resolver leaves for gram_finish_kernel: gfin[0] slots on gtouched, [1] moves on gmoves, [2] 1: a slot
    // is due for a rebuild (the others past half the interval join it); gfin[16 + k]: the count of slot gtouched[k] behind
    // the window.  (Per launch: pipelined windows keep two sets.)
    int *gfin;
    // PIPELINED frozen-factor windows (kernels_gram.hip, "Pipelined windows"): window w's cross forms are made against the
    // factors as they stood at the start of window w - 1 -- while window w - 1 is still being walked -- together with the
    // cross forms between its rows and that window's (gX); gram_carry_kernel then applies window w - 1's logged terms to
    // them (a rank-k change of a 64 x 64 matrix per touched column), so that the resolver starts from the true state.
    // gram_finish / the next cross forms run on a second stream beside the resolver.
    int pipe;                    // 1: this launch belongs to a pipelined batch; 2: ... and its window is the batch's first
                                 // (the components' counts are in Dev::n; later windows take them from the window before:
                                 // gram_finish, which brings Dev::n up to date, runs beside the next resolver)
    long long pipe_pos;          // the visit the launch's window starts at (the host's prediction: 64 visits per window)
    double *gX;                  // [gcols][64][64]: c_0(new row r, old row r') - 1/k_N under the column's frozen factor
    unsigned char *xp_out;       // what the resolver exports for the carry of the NEXT window (GramXp layout below)
    const unsigned char *xp_in;  // ... and what the window before left for this one
    int gram_terms;              // terms the resolver's LDS plan holds (<= kGramMaxTerms)
    int gram_K;                  // labels when the current batch of windows was queued (host side: picks the draw wave's width)
    // safe-stay windows (kernels_safe.hip)
    int safe_mode;               // 1: this batch of steps runs them (home_kernel classifies instead of drawing, the
                                 // frozen-factor kernels take their rows from glist)
    int safe_dense;              // 1: the proof pass of this batch of safe-stay windows is DENSE -- every (visit, label) pair of the stretch
                                 // through the plain likelihood kernel and a verdict per visit from all K exact forms; no bucket sort,
                                 // no per-home tables, no home pass.  For chains whose clusters overlap: there the table bound proves
                                 // nothing and the pruning kernel keeps every pair anyway (kernels_safe.hip)
    double safe_cap;             // > 0: pins the budget per column and window (sum of |log |D_t|| over the rank-1 terms it
                                 // takes); 0: Ctrl::safe_cap, which follows the chain
    long long *glist;            // [kSafeList + 1] visit positions of the stretch's unproven visits, ascending
    struct SafeCol *ep_state;    // [nslots] per SLOT, since the proof pass: budget used, the counts the proofs allow
    double *big_ws;              // D > BGMM_FAST_MAX_D: per-workgroup scratch of the rebuild / rank-1 kernels in global memory instead
    long long big_ws_stride;     // of LDS (doubles per workgroup; nslots + 1 of them)
    int ahead_C;                 // > 0: this batch's dense proof pass takes its forms from the look-ahead ring (a power of two;
                                 // stretches end at multiples of it)
    Job *ah_job, *resc_job;      // device: ah_job[3] = what the second stream scores next (a chunk in full; the touched labels over
                                 // either half of the ring) / what the stretch at hand re-scores
    int *resc_list;              // [nslots] the dirty slots of resc_job
    const int *slot_list;        // MODE_LIST: the list this launch's job indexes
    int ahead_lazy;              // 1: the ring's chunks are never re-scored -- a stretch re-scores every label touched since ITS chunk was made
    long long *touch_seq;        // [nslots] per slot: win_seq of the last window that changed it
    double *rtab;                // [nslots][8] per label of the frozen state: robust constants (kernels_safe.hip)
    double *ftabR;               // [nslots][64] per home label: robust upper bound of every other label's score
    int prune_enabled;           // exact pruning of negligible components in fresh windows, per batch of
                                 // queued steps: 0 never (only the dense kernels are launched), 1 the device
                                 // decides per window (job.prune; both kernel sets are launched), 2 every
                                 // window (only the pruned-window kernels are launched)
};

// A window's export for the next one's carry (pipelined windows).  Sized for the larger column plan.
struct GramXp {
    static constexpr int KCmax = 1024, T = kGramMaxTerms;
    int hdr[16];                 // [0] touched columns listed, [1] terms, [2] 1: the window closed cleanly (all 64 rows walked, no
                                 // component opened or deleted, no error): the next window may be carried from it
    int tcol[T];                 // the touched columns (aligned with gtouched)
    int termCol[T], termRow[T], termPrev[T], termSigma[T];
    double termInvD[T];
    int colLast[KCmax], colN[KCmax], colN0[KCmax], colSlot[KCmax];
    double colRCF[KCmax];
    double wv[T * kGramRows];    // the terms' w vectors over the window's rows
};

__device__ inline double safe_cap_now(const Dev &d, const Ctrl *c) { return d.safe_cap > 0.0 ? d.safe_cap : c->safe_cap; }

// The list the pruning kernel and the sparse draw kernel work through
__host__ __device__ inline const WRec *prune_list(const Dev &d) { return d.use_home ? d.wrecR : d.wrec; }
__host__ __device__ inline const int *prune_rows(const Dev &d) { return d.use_home ? d.wpermR : d.wperm; }
static constexpr int kResidDenseMax = 16384;     // residual visits resid_dense_kernel takes (its grid: one workgroup per sixteen)
__host__ __device__ inline bool resid_dense_takes(const Dev &d, const Ctrl *c) {
    return d.use_home && d.resid_dense && !d.safe_mode && c->n_resid > 0 && c->n_resid <= kResidDenseMax;
}
// rows of the residual list that are left to the general kernels (and count as "not decided by the home pass")
__device__ inline long long resid_left(const Dev &d, const Ctrl *c) { return resid_dense_takes(d, c) ? 0 : c->n_resid; }
__device__ inline long long prune_count(const Dev &d) { return d.use_home ? resid_left(d, d.ctrl) : d.ctrl->n_sorted; }

// Is the (fresh) window described by (mode, prune flag) evaluated by the pruned-window kernels?
__host__ __device__ inline bool job_is_pruned(const Dev &d, int mode, int prune_flag) {
    return mode == MODE_FRESH && (d.prune_enabled == 2 || (d.prune_enabled == 1 && prune_flag != 0));
}

__host__ __device__ constexpr int bgmm_nfrag(int Dp) { return 2 * (Dp / 16) * (Dp / 16 + 1); }

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT DEVICE only: what a launcher has
// already asked for is remembered per device (contexts on different GPUs of one process are independent).
struct PerDeviceLds {
    std::atomic<int> have[64] = {};     // (chains of one process are driven from several host threads: bgmm_group_sweep_staged)
    std::mutex mu;
    // Makes sure kernel `fn` may be launched with `lds` bytes of dynamic LDS on the current device.  The new size is
    // PUBLISHED only after hipFuncSetAttribute has returned (under the lock): a thread that reads have >= lds on the
    // fast path may launch at once -- it never overtakes the thread that is still setting the attribute.
    void ensure(const void *fn, int lds) {
        if (lds <= 64 * 1024) return;
        int dev = 0;
        const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
        if (known && have[dev].load(std::memory_order_acquire) >= lds) return;
        std::lock_guard<std::mutex> lock(mu);
        if (known && have[dev].load(std::memory_order_relaxed) >= lds) return;
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (known) have[dev].store(lds, std::memory_order_release);
    }
};

// The ONE gateway to the process environment (api_context.hip): development switches of the tests and tools, read once per
// process from BGMM_DEV_OPTIONS="name=value,name=value" -- perm_pipe, perm_era, perm_chain_rounds, perm_pipe_fail, perm_rounds, mt_chain_blocks, mt_batch_doubles, mt_lead,
// group_pipe, group_safe, group_host_wait, group_carry_first, group_lazy,
// perm_tail_log2, group_split.  Everything a user may want to set has an entry point (include/bgmm.h: bgmm_set_*).
int bgmm_dev_option(const char *name, int dflt);

// ---- host-side launchers (each defined next to its kernels) -------------------------------
void launch_init_stats(const Dev &d, const int *members, const long long *offsets, int K_init,
                       hipStream_t st);
void launch_refresh_list(const Dev &d, const int *slots, int n, hipStream_t st);  // explicit slots
void launch_refresh_ctrl(const Dev &d, hipStream_t st);                          // ctrl->refresh[]
bool launch_sweep_seq(const Dev &d, int cap, hipStream_t st, const Dev *group = nullptr, int n_group = 0);                    // D <= 4: one wave, visit by visit
int sweep_seq_lds_bytes(int D, int cap);
void launch_refresh_stale(const Dev &d, int K, hipStream_t st);                  // live slots with rank-1 steps
void launch_sweep_begin(const Dev &d, hipStream_t st, const Dev *group = nullptr, int n_group = 0);
void launch_build_tables(const Dev &d, double *tabG, double *tabLogC, hipStream_t st);
void launch_build_seat_table(const Dev &d, double *tabSeat, hipStream_t st);
void launch_apply(const Dev &d, hipStream_t st);
void launch_item_op(const Dev &d, int op, long long i, int label, hipStream_t st); // add/del item
void launch_log_marg(const Dev &d, double *out_total, double *out_per_label, hipStream_t st);
void launch_labels(const Dev &d, long long *z_out, long long *counts_out, hipStream_t st);
void launch_prior_lp(const Dev &d, const double *qcol, hipStream_t st);
void launch_post_pred(const Dev &d, const double *qrow, double *out, hipStream_t st);

void launch_score(const Dev &d, int kind, const Job *job, double *q, long long qstride,
                  int col_override, long long max_rows, int skip_pruned_jobs, hipStream_t st);
bool launch_score_pruned(const Dev &d, const Job *job, double *q, long long qstride, long long max_rows,
                         hipStream_t st);
void launch_certify(const Dev &d, long long max_rows, hipStream_t st);
void launch_prune_tables(const Dev &d, hipStream_t st);
void launch_home(const Dev &d, long long max_rows, hipStream_t st);            // kernels_home.hip
void launch_resid_dense(const Dev &d, hipStream_t st);                         // kernels_resid.hip
int resid_dense_lds_bytes(const Dev &d);
void launch_choice(const Dev &d, long long max_rows, hipStream_t st);
void launch_choice_sparse(const Dev &d, long long max_rows, hipStream_t st);   // pruned windows
void launch_bucket_rows(const Dev &d, long long max_rows, hipStream_t st);
bool resolve_plan(const Dev &d, int K_now, int *R_out, int *Kcap_out, int *lds_out);
void launch_resolve(const Dev &d, int R, int Kcap, int lds, hipStream_t st);
int refresh_lds_bytes(int D, int cov_type = COV_FULL);
bool gram_plan_for(int K, int *gcols, int *terms, int *lds);   // LDS plan of the window resolver for K labels
bool launch_gram_step(const Dev &d, int resolve_lds, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);   // gram + weights + resolve + finish
void gram_configure(const Dev &d, int resolve_lds);      // per-device kernel attributes (once per context and plan)
void launch_gram_finish(const Dev &d, hipStream_t st);
void launch_gram_finish_group(const Dev &lead, const Dev *group, int G, hipStream_t st);
bool launch_gram_group_step(const Dev &lead, const Dev *group, int G, int reach, int resolve_lds, hipStream_t st);   // G chains, shared launches
// the pipelined windows of several chains in shared launches (v0 / v1: the chains' views with buffer set 0 / 1, window k)
bool launch_gram_cross_pgroup(const Dev &lead, const Dev *v0, const Dev *v1, int G, int k, bool with_previous, int max_K, hipStream_t st);
void launch_gram_carry_pgroup(const Dev *v0, const Dev *v1, int G, int k, hipStream_t st);
void launch_gram_resolve_pgroup(const Dev &lead, const Dev *v0, const Dev *v1, int G, int k, int reach, int resolve_lds, hipStream_t st);
void launch_gram_finish_pgroup(const Dev &lead, const Dev *v0, const Dev *v1, int G, int k, hipStream_t st);
void launch_safe_open(const Dev &d, hipStream_t st);                     // kernels_safe.hip
void launch_safe_open_group(const Dev *group, int G, hipStream_t st);
struct SafeAhead;
bool launch_safe_group_step(const Dev &lead, const Dev *group, int G, int reach, int resolve_lds, long long max_rows, int max_nslots,
                            hipStream_t st, const SafeAhead *ah);
bool launch_score_proof_group(const Dev &lead, const Dev *group, int G, long long max_rows, int which, hipStream_t st);   // kernels_score.hip
// (the look-ahead of a dense proof pass: its stream, the events "this step's plan is made" / "its request is served")
struct SafeAhead { hipStream_t stream; hipEvent_t ev_plan, ev_done; };
bool launch_safe_step(const Dev &d, int resolve_lds, long long max_rows, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1,
                      const SafeAhead *ah = nullptr);
bool launch_gram_core(const Dev &d, int resolve_lds, hipStream_t st, hipEvent_t ev0, hipEvent_t ev1);   // gram + weights + resolve + finish
// pipelined windows: the pieces, queued by the host on two streams (api_sweep.hip: gram_pipe_batch)
bool launch_gram_cross(const Dev &d, bool with_previous, hipStream_t st);   // cross forms (+ those with the window before) + weights
void launch_gram_carry(const Dev &d, hipStream_t st);
bool launch_gram_resolve_only(const Dev &d, int resolve_lds, hipStream_t st);
// kernels_rng.hip: the caller's MT19937 continued on the device (chains of 256 blocks from jumped-ahead states)
int mt19937_chains_for(long long pos, long long n);
int mt19937_raw_words();
int mt19937_chain_blocks();
bool mt19937_jump_coefficients(int n_chains, std::vector<unsigned> &out);
void launch_mt19937_raw(const unsigned *key_in, int pos, unsigned *words, long long n_words, const unsigned *coef_dev, int n_chains,
                        unsigned *raw, unsigned *seeds, unsigned *spare_key, int *spare_pos, hipStream_t st);
int mt19937_chains_for_words(long long pos, long long n_words);
// kernels_perm.hip: np.random.permutation(n) from the caller's legacy numpy stream, on the device
size_t perm_sort_temp_bytes(int n);
int perm_segments(long long n_avail);
int perm_rounds();
bool launch_permutation(const unsigned *raw, long long n_avail, int n, const unsigned *key_in, int pos, int *J, int *pred, int *ptr,
                        int *cnt, int *guess_pinned, int *flags, unsigned *ks, unsigned *idx, unsigned *iota, void *temp, size_t temp_bytes,
                        long long *out, int *changed, long long *order, unsigned *key_out, int *pos_out, hipStream_t st);
bool launch_permutation_tail(const unsigned *raw, int n, const unsigned *key_in, int pos, int *J, int *pred, int *ptr, unsigned *ks,
                             unsigned *idx, unsigned *iota, void *temp, size_t temp_bytes, long long *out, int *changed, long long *order,
                             unsigned *key_out, int *pos_out, hipStream_t st);
void launch_permutation_draw_more(const unsigned *raw, long long n_avail, int n, int *J, int *cnt, int *flags, long long *out,
                                  hipStream_t st);
void launch_permutation_more(int n, const int *J, const int *pred, int *ptr, int *changed, long long *order, hipStream_t st);
void launch_perm_iota(int n, unsigned *iota, hipStream_t st);
bool launch_permutation_swaps(int n, int *J, int *pred, int *ptr, unsigned *ks, unsigned *idx, unsigned *iota, void *temp,
                              size_t temp_bytes, int *changed, long long *order, hipStream_t st);
int perm_chain_guess(long long n_avail, int n, int *pre0_host);
int perm_bucket_cap();
int perm_bucket_bounds(int n, std::vector<int> &bnd);
bool launch_permutation_swaps_bucketed(int n, int NB, const int *bnd, const int *J, int *cursor, unsigned long long *slots, int *overflow,
                                       int *pred, int *ptr, long long *order, hipStream_t st);
bool launch_permutation_draws_chained(const unsigned *era_raw, const unsigned *era_key, int era_pos, const long long *goff_in,
                                       long long *goff_out, long long n_avail, int n, int *J, int *cnt, const int *pre0, int *zero,
                                       int *next_zero, int nblk_pad, int *flags_out, long long *out, unsigned *key_out, int *pos_out,
                                       int rounds, hipStream_t st);
static constexpr int kMtMaxMids = 8;      // sweeps a look-ahead request of the uniform generator may span
struct MtMids { long long nb[kMtMaxMids]; int pos[kMtMaxMids]; int m; };
void launch_mt19937(const unsigned *key_in, int pos, unsigned *key_out, int *pos_out, unsigned *words, double *u, long long n,
                    int *zero_flag, const unsigned *coef_dev, int n_chains, unsigned *raw, unsigned *seeds, hipStream_t st,
                    int n_sweeps = 1, unsigned *key_mid = nullptr, int *pos_mid = nullptr);

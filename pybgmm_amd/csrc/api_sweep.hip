// The per-sweep launch schedule (the body of `for i_iter`, igmm/crpmm.py:57-88, igmm/pcrpmm.py:93-131): which kind of
// window a batch of steps queues, frozen-factor batches plain and pipelined, the staged sweep whole and in two halves.
#include "api_internal.h"

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
void gram_point_view(bgmm_ctx *c, Dev &v, int par) {
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

int ensure_events(bgmm_ctx *c, size_t n) {
    while (c->ev0.size() < n) {
        hipEvent_t a, b;
        CK(c, hipEventCreate(&a));
        CK(c, hipEventCreate(&b));
        c->ev0.push_back(a);
        c->ev1.push_back(b);
    }
    return 0;
}

// One sweep.  phase 0: all of it.  Phases 1 and 2 split it for bgmm_group_sweep_staged, which opens the sweeps of several
// chains and runs their one-workgroup sweeps (kernels_seq.hip) in ONE launch each: phase 1 = everything in front of
// sweep_begin; returns 1 if the chain can take the one-workgroup sweep (bgmm_ctx::grp_cap = its LDS plan; the caller
// launches, fills ctrl_host and comes back with phase 2), otherwise carries on as phase 0.  Phase 2 = what follows.
int sweep_impl(bgmm_ctx *c, int32_t use_power, double power, int phase) {
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
        // (... and so is a batch of safe-stay steps whose proof pass is the dense one: round 6)
        static const bool group_safe_on = bgmm_dev_option("group_safe", 1) != 0;
        const bool safe_dense_next = c->safe_dense_pin >= 0 ? c->safe_dense_pin != 0 : c->safe_dense_on;
        const bool share_safe = use_safe && c->combiner && !c->timing && safe_dense_next && group_safe_on;
        if (c->combiner && !((use_gram && !c->timing) || share_safe)) combiner_declare_busy(c);
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
            // (chains of a group call that are here together share their steps' launches, look-ahead included: GramCombiner)
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
            int own_safe = 1;
            if (share_safe) {
                gram_point(c, 0);
                d.pipe = 0;
                own_safe = combiner_submit(c, (int)Tg, 0, pos, 1);
                if (own_safe < 0) return fail(c, BGMM_EDEVICE, "shared safe-stay launch failed");
                c->grp_stats[own_safe == 0 ? 2 : 3] += 1;
            }
            if (own_safe == 1) {
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
            }
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
            bool piped = false;
            if (c->combiner && !c->timing) {
                // (far inside the mover-dense regime the group's windows are pipelined like a single chain's: every chain says
                //  how many it could take -- 0: none, e.g. right after a break of its chain)
                long long Tp = 0;
                if (c->pipe_mode && c->resolver_mode != 1 && remaining >= 4 * kGramRows && (was_first || hc.job.pos == pos)) {
                    if (c->pipe_hold > 0) c->pipe_hold -= 1;
                    else {
                        Tp = remaining / kGramRows;
                        if (Tp > Tg) Tp = Tg;
                        if (Tp < 4) Tp = 0;
                    }
                }
                own = combiner_submit(c, (int)Tg, (int)Tp, pos, 0);
                if (own < 0) return fail(c, BGMM_EDEVICE, "shared frozen-factor launch failed");
                if (own == 2) { piped = true; own = 0; c->pipe_batches += 1; c->grp_stats[1] += 1; }
                c->grp_stats[own == 0 ? 0 : 3] += 1;
            }
            // (a chain on its own, far inside the mover-dense regime: the windows pipelined -- gram_finish and the next cross
            //  forms on a second stream beside the resolver; after a break of the chain a couple of plain batches first)
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
int finish_pending(bgmm_ctx *c) {
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

// np.random.permutation(N) of the caller's legacy numpy generator on the device (igmm/pcrpmm.py:86-91): one generation on
// the spot, the single look-ahead, and the generations in flight on three streams.
#include "api_internal.h"

// np.random.permutation(N) from the caller's legacy numpy generator, on the device (kernels_perm.hip).

int perm_ensure(bgmm_ctx *c, PermPtrs &P) {
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
int perm_schedule(bgmm_ctx *c, const PermPtrs &P) {
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
// end, and that they were generated.  So (BGMM_DEV_OPTIONS perm_pipe=0: the single look-ahead of round 3):
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
    static const bool on = bgmm_dev_option("perm_pipe", 1) != 0;
    return on;
}

int perm_pipe_drain(bgmm_ctx *c) {
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
    // may read (BGMM_DEV_OPTIONS perm_era: generations' worth, for the test that walks through several eras)
    static const int era_gens = [] { const int v = bgmm_dev_option("perm_era", 32); return v < 1 ? 1 : v; }();
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
        CK(c, hipEventCreateWithFlags(&Q.ev_draw[k], hipEventDisableTiming));
        CK(c, hipEventCreateWithFlags(&Q.ev_fin[k], hipEventDisableTiming));
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
// BGMM_DEV_OPTIONS perm_pipe_fail=1 makes the set-up fail after its allocations (the test of this path).
static int perm_pipe_ensure(bgmm_ctx *c, const PermPtrs &P) {
    bgmm_ctx::PermPipe &Q = c->pp;
    if (Q.built) return 0;
    if (Q.off) return 1;
    std::string why;
    g_err_sink = &why;                  // (a failed set-up is not the caller's error: bgmm_ctx::err stays what it was)
    int rc = perm_pipe_build(c, P);
    if (rc == 0 && bgmm_dev_option("perm_pipe_fail", 0)) {
        { std::lock_guard<std::mutex> g(Q.mu); Q.quit = true; }
        Q.cv.notify_all();
        if (Q.worker.joinable()) Q.worker.join();
        Q.quit = false;
        why = "perm_pipe_fail (BGMM_DEV_OPTIONS)";
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

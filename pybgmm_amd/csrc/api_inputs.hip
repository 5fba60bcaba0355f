// A sweep's inputs: staged and resident streams; random.random() x N of the caller's Mersenne Twister continued on the
// device with its look-ahead (utils/utils.py:13-16 is where the reference draws them).
#include "api_internal.h"

// The sequential small-D sweep fetches z[i] ahead of the visit of i: sound only when no index
// comes twice.  Checked on the host for the shapes that can take that path.
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
int mt_ensure_tables(bgmm_ctx *c, int chains) {
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
    long long m = (long long)bgmm_dev_option("mt_batch_doubles", 8000000) / N;
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

int mt_wait_batches(bgmm_ctx *c) {
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
int mt_schedule(bgmm_ctx *c, bool hit, const uint32_t *key, int pos) {
    if (!c->mt_ahead_on) return 0;
    if (!hit) {
        int rc = mt_launch_batch(c, 0, key, pos);
        if (rc) return rc;
        c->mt_cur = 0;
        return 0;
    }
    bgmm_ctx::MtBatch *B = c->mt_cur >= 0 ? &c->mt_b[c->mt_cur] : nullptr;
    static const int lead = bgmm_dev_option("mt_lead", 8);
    if (B && B->launched && B->next >= std::max(1, c->mt_depth - lead) && !c->mt_b[c->mt_cur ^ 1].launched) {
        // (the state behind this batch is known since its generation finished, the other batch's buffer is free since its
        // last sweep was: the next batch is started as soon as this one is being served -- `mt_lead` sweeps before it is
        // needed at the latest; round 5 waited until two were left, and a generation that ran long held the sweep up)
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

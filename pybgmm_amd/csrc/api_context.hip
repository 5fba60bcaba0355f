// The context: create / destroy, state in and out (labels, statistics, log marginal, metrics), tuning and statistics
// getters, the method-level entry points (add_item / del_item / del_component / set_stats: gaussian_components.py:129-205).
#include "api_internal.h"

thread_local std::string g_create_error;
thread_local std::string *g_err_sink = nullptr;

int bgmm_dev_option(const char *name, int dflt) {
    // parsed once (C++11 static initialisation is thread safe); unknown names are ignored, malformed values read as 0
    static const std::vector<std::pair<std::string, int>> opts = [] {
        std::vector<std::pair<std::string, int>> v;
        const char *e = getenv("BGMM_DEV_OPTIONS");
        std::string s = e ? e : "";
        size_t i = 0;
        while (i < s.size()) {
            size_t j = s.find(',', i);
            if (j == std::string::npos) j = s.size();
            const std::string item = s.substr(i, j - i);
            const size_t eq = item.find('=');
            if (eq != std::string::npos && eq > 0) v.emplace_back(item.substr(0, eq), atoi(item.c_str() + eq + 1));
            i = j + 1;
        }
        return v;
    }();
    for (const auto &kv : opts)
        if (kv.first == name) return kv.second;
    return dflt;
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
    if (c->xshare && c->xshare->refs.fetch_sub(1) == 1) {       // (the last context that used this copy of X)
        if (c->xshare->p) (void)hipFree(c->xshare->p);
        delete c->xshare;
    }
    c->xshare = nullptr;
    if (c->mt_stream) { (void)hipStreamSynchronize(c->mt_stream); (void)hipStreamDestroy(c->mt_stream); }
    for (auto &b : c->mt_b) {
        if (b.done) (void)hipEventDestroy(b.done);
        if (b.u) (void)hipFree(b.u);
        if (b.host) (void)hipHostFree(b.host);
    }
    if (c->mt_words_ahead) (void)hipFree(c->mt_words_ahead);
    if (c->grp_devs) (void)hipFree(c->grp_devs);
    if (c->grp_pdevs) (void)hipFree(c->grp_pdevs);
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

// (share: a context whose device copy of X this one borrows instead of uploading its own -- X is then not read)
static int create_impl(bgmm_ctx *c, int device, int64_t N, int32_t D, int32_t K_max, int32_t cov_type,
                       const double *X, const double *m_0, double k_0, int64_t v_0,
                       const double *S_0, double alpha, const double *lgamma_tab,
                       const double *log_tab, bgmm_ctx *share = nullptr) {
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
    if (share) {
        c->xshare = share->xshare;
        c->xshare->refs.fetch_add(1);
        dX = c->xshare->p;
    } else {
        void *q = nullptr;
        CK(c, hipMalloc(&q, sizeof(double) * (size_t)N * D + 64));
        c->xshare = new SharedX();
        c->xshare->p = dX = (double *)q;
    }
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
    d.big_ws = nullptr; d.big_ws_stride = 0;
    if (cov_type == BGMM_COV_FULL && D > BGMM_FAST_MAX_D) {
        d.big_ws_stride = refresh_ws_doubles(D);
        DALLOC(c, d.big_ws, (size_t)d.big_ws_stride * (ns + 1));
    }
    DALLOC(c, d.ah_job, 3);
    DALLOC(c, d.resc_job, 1);
    DALLOC(c, d.resc_list, ns);
    DALLOC(c, d.touch_seq, ns);
    CK(c, hipMemsetAsync(d.touch_seq, 0, sizeof(long long) * ns, c->stream));
    d.ahead_C = 0; d.slot_list = nullptr; d.ahead_lazy = 0;
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
    if (!share) CK(c, hipMemcpyAsync(dX, X, sizeof(double) * N * D, hipMemcpyHostToDevice, c->stream));

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
        return fail(nullptr, BGMM_EUNSUPPORTED, "full covariance supports D <= 256 (up to 128 a component's D x D factor fits the LDS of a "
                                                "compute unit: the fast kernels; 129 .. 256 take the general route through a workspace in "
                                                "global memory); covariance_type diag / fixed take D up to 4096");
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

extern "C" int bgmm_create_shared(bgmm_ctx **out, bgmm_ctx *parent, int32_t K_max, const double *m_0, double k_0, int64_t v_0,
                                  const double *S_0, double alpha, const double *lgamma_tab, const double *log_tab) {
    if (!out) return BGMM_EINVAL;
    *out = nullptr;
    if (!parent || !parent->xshare || !m_0 || !S_0 || K_max < 1) return fail(nullptr, BGMM_EINVAL, "bad shape or null pointer");
    const Dev &pd = parent->d;
    if (v_0 < pd.D && pd.cov_type == COV_FULL) return fail(nullptr, BGMM_EINVAL, "v_0 must be larger or equal to dimension of data");
    if (v_0 < 1) return fail(nullptr, BGMM_EINVAL, "v_0 must be positive");
    if (!(k_0 > 0) || !(alpha > 0)) return fail(nullptr, BGMM_EINVAL, "k_0 and alpha must be positive");
    {   // (whatever the parent still has in its queue -- its own upload of X among it -- comes first)
        if (hipSetDevice(parent->device) != hipSuccess || hipStreamSynchronize(parent->stream) != hipSuccess)
            return fail(nullptr, BGMM_EDEVICE, "the parent context's device is not available");
    }
    bgmm_ctx *c = new bgmm_ctx();
    const int rc = create_impl(c, parent->device, pd.N, pd.D, K_max, pd.cov_type, nullptr, m_0, k_0, v_0, S_0, alpha, lgamma_tab, log_tab,
                               parent);
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

extern "C" int bgmm_get_group_stats(bgmm_ctx *c, int64_t *out4) {
    if (!c || !out4) return BGMM_EINVAL;
    for (int k = 0; k < 4; ++k) out4[k] = c->grp_stats[k];
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

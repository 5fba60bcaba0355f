/*
 * ORACLE (test infrastructure, not product code) -- plain-C restatement of the
 * reference's collapsed-Gibbs reassignment path (CRP / pCRP Gaussian mixture,
 * NIW prior; full covariance, and -- SURVEY 8f rank 1 -- diagonal covariance).  Scalar, one visit at a time (the K evaluations of
 * a visit optionally shared out over threads, see go_set_threads),
 * with the reference's algorithmic structure: component statistics cached and
 * restored around every visit, covariance log-determinant and inverse rebuilt
 * FROM SCRATCH (LU with partial pivoting) on every removal and every move, a
 * K * D^2 contraction for the Student-t predictive, one uniform per visit.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this file's shared object.  The product (pybgmm_amd + libbgmm_hip.so) never
 * links, loads or calls it.
 *
 * Parity status: PINNED (tests/test_oracle_c.py): integer trajectories equal to
 * the reference's own 2014 known-answer vectors and to the trajectories captured
 * from the reference in tests/golden/; log marginals within 1e-9 relative.  Its
 * floats are NOT bit-identical to numpy's (unblocked LU, different summation
 * order); the numpy oracle (gibbs_numpy.py) is the bit-identical one.
 *
 * Reference file:line each function follows
 *   go_create / tables / log_prior ... pybgmm/gaussian/gaussian_components.py:75-127, 207-214
 *   go_set_assignments ............... :96-111   (k ascending, i ascending)
 *   seat / unseat / drop_component ... :154-205
 *   refresh_cov ...................... :319-331
 *   predictive_all ................... :228-251
 *   go_sweep ......................... pybgmm/igmm/crpmm.py:57-88, pybgmm/igmm/pcrpmm.py:93-131
 *   draw (inside go_sweep) ........... pybgmm/utils/utils.py:15-20
 *   go_log_marg ...................... pybgmm/igmm/igmm.py:199-215, gaussian_components.py:253-289
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>
#include <sched.h>
#include <stdatomic.h>
#include <unistd.h>

/* Threads (round 6).  A visit's K predictive evaluations are independent of each other, and so are the D columns of
 * the inverse: both loops are shared out over go_set_threads(T) threads.  Every component's (and every column's)
 * arithmetic is the scalar code below, untouched, on its own scratch -- the floats are bit-identical for every T; the
 * maximum, the log-sum-exp and the u -= p scan stay serial in label order.  T = 1 (the default of the library; the
 * loader picks more for the tests) is the single-threaded port bench.py's cpu_baseline times.
 *
 * The team is a pool of its own: workers that SPIN on a generation counter between the regions of a visit (two or three
 * per visit, tens of microseconds each -- an OpenMP runtime whose workers sleep between regions, which is what a
 * process that has already loaded torch's gets, made the D = 128 cases slower on a 256-core box, not faster), static
 * contiguous shares; once nothing has come for a while a worker polls every 50 us, after 10 ms every millisecond. */
#define GO_MAX_THREADS 64
typedef void (*go_job_fn)(void *arg, int64_t lo, int64_t hi, int tid);
/* The three hot loops are compiled twice, for AVX2 and for the baseline ISA, and picked at load time (the shared object is
 * built in one container and run in another).  Wider vectors hold more INDEPENDENT sums side by side; no sum is reordered
 * (-fno-fast-math) and no product is fused into its sum (-ffp-contract=off, and the clones do not enable FMA). */
#if defined(__x86_64__) && defined(__GNUC__) && !defined(__clang__)
#define GO_HOT __attribute__((target_clones("avx2", "default")))
#else
#define GO_HOT
#endif
static int go_threads = 1;
static struct {
    pthread_t th[GO_MAX_THREADS];
    int started;                        /* workers that exist (ids 1 .. started) */
    uint64_t seen0[GO_MAX_THREADS];
    atomic_flag busy;                   /* one region at a time: a second caller (another oracle on another thread) runs its own serially */
    _Atomic uint64_t gen;
    _Atomic int done;
    go_job_fn fn;
    void *arg;
    int64_t n;
    int nt;
} go_pool;

static void *go_worker(void *p) {
    const int tid = (int)(intptr_t)p;
    uint64_t seen = go_pool.seen0[tid];       /* (the generation when it was made: regions before that are not its business) */
    for (;;) {
        uint64_t g;
        int idle = 0;
        while ((g = atomic_load_explicit(&go_pool.gen, memory_order_acquire)) == seen) {
            if (++idle <= (1 << 16)) { if ((idle & 1023) == 0) sched_yield(); else __builtin_ia32_pause(); }
            else if (idle <= (1 << 16) + 200) usleep(50);
            else { usleep(1000); idle = (1 << 16) + 200; }           /* (nothing for 10 ms: the sweep is over) */
        }
        seen = g;
        if (tid < go_pool.nt) {
            const int64_t lo = go_pool.n * tid / go_pool.nt, hi = go_pool.n * (tid + 1) / go_pool.nt;
            if (hi > lo) go_pool.fn(go_pool.arg, lo, hi, tid);
        }
        /* EVERY worker answers every region, with or without a share: a worker that merely skipped one could otherwise still
         * be looking at the region's description when the next one is written over it, take a share of that one, and take it
         * again when it notices the new generation (seen as labels that differed from run to run while this was so). */
        atomic_fetch_add_explicit(&go_pool.done, 1, memory_order_release);
    }
    return NULL;
}

/* fn over [0, n) in nt contiguous shares; share 0 on the calling thread.  Returns when all of them are done. */
static void go_parallel_for(go_job_fn fn, void *arg, int64_t n, int nt) {
    if (nt > go_pool.started + 1) nt = go_pool.started + 1;
    if (nt > n) nt = (int)n;
    if (nt <= 1 || atomic_flag_test_and_set_explicit(&go_pool.busy, memory_order_acquire)) { fn(arg, 0, n, 0); return; }
    go_pool.fn = fn; go_pool.arg = arg; go_pool.n = n; go_pool.nt = nt;
    atomic_store_explicit(&go_pool.done, 0, memory_order_relaxed);
    atomic_fetch_add_explicit(&go_pool.gen, 1, memory_order_release);
    fn(arg, 0, n / nt, 0);
    for (int spin = 0; atomic_load_explicit(&go_pool.done, memory_order_acquire) < go_pool.started; ++spin) {
        if ((spin & 1023) == 1023) sched_yield();
        else __builtin_ia32_pause();
    }
    atomic_flag_clear_explicit(&go_pool.busy, memory_order_release);
}

void go_set_threads(int t) {
    if (t < 1) t = 1;
    if (t > GO_MAX_THREADS) t = GO_MAX_THREADS;
    {   /* never more threads than cores this process may run on: the team spins, an oversubscribed team crawls */
        cpu_set_t set;
        int cores = 1;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) cores = CPU_COUNT(&set);
        if (cores < 1) cores = 1;
        if (t > cores) t = cores;
    }
    while (atomic_flag_test_and_set_explicit(&go_pool.busy, memory_order_acquire)) sched_yield();     /* (no region meanwhile) */
    while (go_pool.started + 1 < t) {
        const int tid = go_pool.started + 1;
        go_pool.seen0[tid] = atomic_load_explicit(&go_pool.gen, memory_order_acquire);
        if (pthread_create(&go_pool.th[tid], NULL, go_worker, (void *)(intptr_t)tid) != 0) break;
        pthread_detach(go_pool.th[tid]);
        go_pool.started = tid;
    }
    go_threads = t < go_pool.started + 1 ? t : go_pool.started + 1;
    atomic_flag_clear_explicit(&go_pool.busy, memory_order_release);
}
int go_get_threads(void) { return go_threads; }

typedef struct {
    int64_t N, D, K_max, K;
    int diag;                   /* 1: covariance_type="diag" (gaussian_components_diag.py), 2: "fixed"
                                   (gaussian_components_fixedvar.py); S, inv are D-vectors in both */
    double *prec, *prec0;       /* fixed: 1/var and 1/var_0 (D each); m0 holds mu_0 */
    int64_t SD;                 /* entries of one second-moment block: D*D or D */
    const double *X;            /* borrowed, N x D row major */
    double *m0, *S0;
    double k0, alpha;
    int64_t v0;
    double *tab_lgam, *tab_log; /* index n -> lgamma(n/2), log(n); slot 0 dud (n=1) */
    int64_t tab_len;
    double *prior_m, *prior_S;  /* k0*m0 and S0 + k0*m0 m0^T */
    double *m, *S, *logdet, *inv;
    int64_t *n, *z;
    double *log_prior;
    /* scratch */
    double *lu, *col, *lp, *save_m, *save_S, *save_inv, *delta, *tmp;
    int64_t *piv;
} go_t;

#define LOG_PI 1.1447298858494001741434273513530587116472948129153

/* LU with partial pivoting in place; returns sign, fills piv.  The rows below the pivot are independent of each other
 * within a column step: from n = 64 on they are shared out (every row's update is the serial loop's arithmetic on that
 * row: bit-identical for any thread count); the pivot search and the row swap stay with the calling thread. */
typedef struct { double *a; int64_t n, j; double d; } lu_job_t;
GO_HOT static void lu_rows(void *argp, int64_t lo, int64_t hi, int tid) {
    (void)tid;
    const lu_job_t *J = (const lu_job_t *)argp;
    double *a = J->a; const int64_t n = J->n, j = J->j; const double d = J->d;
    const double *restrict rj = a + j * n;
    for (int64_t i = j + 1 + lo; i < j + 1 + hi; ++i) {
        double *restrict ri = a + i * n;
        double l = ri[j] / d;
        ri[j] = l;
        if (l != 0.0)
            for (int64_t c = j + 1; c < n; ++c) ri[c] -= l * rj[c];
    }
}
static int lu_factor(double *a, int64_t n, int64_t *piv) {
    int sign = 1;
    for (int64_t j = 0; j < n; ++j) {
        int64_t p = j;
        double best = fabs(a[j * n + j]);
        for (int64_t i = j + 1; i < n; ++i) {
            double v = fabs(a[i * n + j]);
            if (v > best) { best = v; p = i; }
        }
        piv[j] = p;
        if (p != j) {
            for (int64_t c = 0; c < n; ++c) {
                double t = a[j * n + c]; a[j * n + c] = a[p * n + c]; a[p * n + c] = t;
            }
            sign = -sign;
        }
        double d = a[j * n + j];
        if (d == 0.0) continue;
        if (n < 8) {            /* (tiny n: in place, no hand-over) */
            for (int64_t i = j + 1; i < n; ++i) {
                double l = a[i * n + j] / d;
                a[i * n + j] = l;
                if (l != 0.0)
                    for (int64_t c = j + 1; c < n; ++c) a[i * n + c] -= l * a[j * n + c];
            }
            continue;
        }
        lu_job_t J = {a, n, j, d};
        const int64_t rows = n - j - 1;
        int nt = n >= 64 ? go_threads : 1;
        while (nt > 1 && (rows / nt) * rows < 1500) --nt;        /* (a share worth a hand-over) */
        go_parallel_for(lu_rows, &J, rows, nt);
    }
    return sign;
}

static double lu_logabsdet(const double *lu, int64_t n) {
    double s = 0.0;
    for (int64_t j = 0; j < n; ++j) s += log(fabs(lu[j * n + j]));
    return s;
}

/* inverse from the factorisation: solve A x = e_c for every column c.  All right-hand sides of a share [lo, hi) go through
 * the substitutions TOGETHER, row by row of the factor: Y[i][c] -= lu[i][t] * Y[t][c] for t ascending -- per entry the
 * very subtractions, in the very order, of a column-at-a-time solve (s = col[i]; s -= lu[i][t] * col[t], t ascending), so
 * the floats are those of the scalar loop; but c is the fastest index, contiguous in `out` (out[i * n + c] IS Y[i][c]). */
typedef struct { const double *lu; const int64_t *piv; int64_t n; double *out; } inv_job_t;
GO_HOT static void lu_inverse_cols(void *argp, int64_t lo, int64_t hi, int tid) {
    (void)tid;
    const inv_job_t *J = (const inv_job_t *)argp;
    const double *lu = J->lu; const int64_t *piv = J->piv; const int64_t n = J->n; double *Y = J->out;
    const int64_t w = hi - lo;
    for (int64_t i = 0; i < n; ++i)
        for (int64_t c = lo; c < hi; ++c) Y[i * n + c] = (i == c) ? 1.0 : 0.0;
    for (int64_t j = 0; j < n; ++j) {
        int64_t p = piv[j];
        if (p != j)
            for (int64_t c = lo; c < hi; ++c) { double t = Y[j * n + c]; Y[j * n + c] = Y[p * n + c]; Y[p * n + c] = t; }
    }
    for (int64_t i = 0; i < n; ++i) {
        double *restrict yi = Y + i * n + lo;
        for (int64_t t = 0; t < i; ++t) {
            const double l = lu[i * n + t];
            const double *restrict yt = Y + t * n + lo;
            for (int64_t c = 0; c < w; ++c) yi[c] -= l * yt[c];
        }
    }
    for (int64_t i = n - 1; i >= 0; --i) {
        double *restrict yi = Y + i * n + lo;
        for (int64_t t = i + 1; t < n; ++t) {
            const double l = lu[i * n + t];
            const double *restrict yt = Y + t * n + lo;
            for (int64_t c = 0; c < w; ++c) yi[c] -= l * yt[c];
        }
        const double d = lu[i * n + i];
        for (int64_t c = 0; c < w; ++c) yi[c] = yi[c] / d;
    }
}
static void lu_inverse(const double *lu, const int64_t *piv, int64_t n, double *out, double *col) {
    if (n < 8) {                /* (tiny n: a column at a time -- the same subtractions in the same order) */
        for (int64_t c = 0; c < n; ++c) {
            for (int64_t i = 0; i < n; ++i) col[i] = (i == c) ? 1.0 : 0.0;
            for (int64_t j = 0; j < n; ++j) {
                int64_t p = piv[j];
                if (p != j) { double t = col[j]; col[j] = col[p]; col[p] = t; }
            }
            for (int64_t i = 0; i < n; ++i) {
                double s = col[i];
                for (int64_t t = 0; t < i; ++t) s -= lu[i * n + t] * col[t];
                col[i] = s;
            }
            for (int64_t i = n - 1; i >= 0; --i) {
                double s = col[i];
                for (int64_t t = i + 1; t < n; ++t) s -= lu[i * n + t] * col[t];
                col[i] = s / lu[i * n + i];
            }
            for (int64_t i = 0; i < n; ++i) out[i * n + c] = col[i];
        }
        return;
    }
    inv_job_t J = {lu, piv, n, out};
    /* (a column is ~2 n^2 flop: shared out only where a share is worth a hand-over) */
    int nt = go_threads;
    while (nt > 1 && (n / nt) * n * n < 20000) --nt;
    go_parallel_for(lu_inverse_cols, &J, n, nt);
}

static double slogdet_of(go_t *g, const double *a) {
    int64_t D = g->D;
    memcpy(g->lu, a, sizeof(double) * D * D);
    lu_factor(g->lu, D, g->piv);
    return lu_logabsdet(g->lu, D);
}

static void refresh_cov(go_t *g, int64_t k) {
    int64_t D = g->D;
    if (g->diag == 2) {         /* gaussian_components_fixedvar.py:282-291 */
        double lp = 0.0;
        for (int64_t a = 0; a < D; ++a) {
            double pN = g->S[k * D + a];
            double pp = pN * g->prec[a] / (pN + g->prec[a]);
            lp += log(pp);
            g->inv[k * D + a] = pp;
        }
        g->logdet[k] = lp;
        return;
    }
    if (g->diag) {              /* gaussian_components_diag.py:325-338 */
        double k_N = g->k0 + (double)g->n[k];
        double v_N = (double)(g->v0 + g->n[k]);
        double scale = (k_N + 1.) / (k_N * v_N);
        const double *m = g->m + k * D, *S = g->S + k * D;
        double lp = 0.0;
        for (int64_t a = 0; a < D; ++a) {
            double mean = m[a] / k_N;
            double var = scale * (S[a] - k_N * (mean * mean));
            lp += log(var);
            g->inv[k * D + a] = 1. / var;
        }
        g->logdet[k] = lp;
        return;
    }
    double k_N = g->k0 + (double)g->n[k];
    double v_N = (double)(g->v0 + g->n[k]);
    double scale = (k_N + 1.) / (k_N * (v_N - (double)D + 1.));
    const double *m = g->m + k * D, *S = g->S + k * D * D;
    for (int64_t a = 0; a < D; ++a) g->tmp[a] = m[a] / k_N;
    for (int64_t a = 0; a < D; ++a)
        for (int64_t b = 0; b < D; ++b)
            g->lu[a * D + b] = scale * (S[a * D + b] - k_N * (g->tmp[a] * g->tmp[b]));
    lu_factor(g->lu, D, g->piv);
    g->logdet[k] = lu_logabsdet(g->lu, D);
    lu_inverse(g->lu, g->piv, D, g->inv + k * D * D, g->col);
}

static void seat(go_t *g, int64_t i, int64_t k) {
    int64_t D = g->D;
    const double *x = g->X + i * D;
    double *m = g->m + k * D, *S = g->S + k * g->SD;
    if (k == g->K) {
        g->K += 1;
        memcpy(m, g->prior_m, sizeof(double) * D);
        memcpy(S, g->prior_S, sizeof(double) * g->SD);
    }
    if (g->diag == 2) {         /* gaussian_components_fixedvar.py:146-162 */
        for (int64_t a = 0; a < D; ++a) { double px = g->prec[a] * x[a]; m[a] += px; S[a] += g->prec[a]; }
        g->n[k] += 1;
        refresh_cov(g, k);
        g->z[i] = k;
        return;
    }
    for (int64_t a = 0; a < D; ++a) m[a] += x[a];
    if (g->diag) {
        for (int64_t a = 0; a < D; ++a) { double o = x[a] * x[a]; S[a] += o; }
    } else {
        for (int64_t a = 0; a < D; ++a)
            for (int64_t b = 0; b < D; ++b) {
                double o = x[a] * x[b];
                S[a * D + b] += o;
            }
    }
    g->n[k] += 1;
    refresh_cov(g, k);
    g->z[i] = k;
}

static void drop_component(go_t *g, int64_t k) {
    int64_t D = g->D;
    g->K -= 1;
    int64_t last = g->K;
    if (k != last) {
        memcpy(g->m + k * D, g->m + last * D, sizeof(double) * D);
        memcpy(g->S + k * g->SD, g->S + last * g->SD, sizeof(double) * g->SD);
        g->logdet[k] = g->logdet[last];
        memcpy(g->inv + k * g->SD, g->inv + last * g->SD, sizeof(double) * g->SD);
        g->n[k] = g->n[last];
        for (int64_t i = 0; i < g->N; ++i) if (g->z[i] == last) g->z[i] = k;
    }
    memset(g->m + last * D, 0, sizeof(double) * D);
    memset(g->S + last * g->SD, 0, sizeof(double) * g->SD);
    g->logdet[last] = 0.;
    memset(g->inv + last * g->SD, 0, sizeof(double) * g->SD);
    g->n[last] = 0;
}

static void unseat(go_t *g, int64_t i) {
    int64_t D = g->D;
    int64_t k = g->z[i];
    if (k == -1) return;
    g->n[k] -= 1;
    g->z[i] = -1;
    if (g->n[k] == 0) { drop_component(g, k); return; }
    const double *x = g->X + i * D;
    double *m = g->m + k * D, *S = g->S + k * g->SD;
    if (g->diag == 2) {
        for (int64_t a = 0; a < D; ++a) { double px = g->prec[a] * x[a]; m[a] -= px; S[a] -= g->prec[a]; }
        refresh_cov(g, k);
        return;
    }
    for (int64_t a = 0; a < D; ++a) m[a] -= x[a];
    if (g->diag) {
        for (int64_t a = 0; a < D; ++a) { double o = x[a] * x[a]; S[a] -= o; }
    } else {
        for (int64_t a = 0; a < D; ++a)
            for (int64_t b = 0; b < D; ++b) {
                double o = x[a] * x[b];
                S[a * D + b] -= o;
            }
    }
    refresh_cov(g, k);
}

GO_HOT static double student_t_wide(const go_t *g, const double *x, const double *mu_num, double k_N,
                        double logdet, const double *inv, int64_t nu, double *delta) {
    int64_t D = g->D;
    if (g->diag == 2) {         /* product of univariate normals: gaussian_components_fixedvar.py:293-303;
                                   mu_num / k_N is the mean (callers pass k_N = 1 and a ready mean) */
        double acc = 0.0;
        for (int64_t a = 0; a < D; ++a) {
            double dl = x[a] - mu_num[a] / k_N;
            acc += (dl * dl) * inv[a];
        }
        return -0.5 * (double)D * log(2. * 3.14159265358979323846) + 0.5 * logdet - 0.5 * acc;
    }
    if (g->diag) {              /* product of univariate Student-t: gaussian_components_diag.py:340-354 */
        double acc = 0.0;
        for (int64_t a = 0; a < D; ++a) {
            double dl = x[a] - mu_num[a] / k_N;
            acc += log(1. + 1. / (double)nu * (dl * dl) * inv[a]);
        }
        return (double)D * (g->tab_lgam[nu + 1] - g->tab_lgam[nu] - 0.5 * g->tab_log[nu] - 0.5 * LOG_PI)
               - 0.5 * logdet - ((double)nu + 1.) / 2. * acc;
    }
    /* q = sum_a (sum_b delta[b] inv[b][a]) delta[a] -- the reference's einsum pair (gaussian_components.py:240-244).
     * The inner sums run over b ascending for every a, as a scalar loop over a would have them: with b outermost the SAME
     * additions happen in the SAME order per a (r[a] += delta[b] * inv[b][a], product and sum rounded separately), but
     * the memory is walked row by row and the compiler may keep several a in one vector register.  delta[D .. 2D) is r. */
    double q = 0.0;
    if (D < 8) {                /* (tiny D: the plain column loop -- the same sums; vector set-up would cost more than it saves) */
        for (int64_t a = 0; a < D; ++a) delta[a] = mu_num[a] / k_N - x[a];
        for (int64_t a = 0; a < D; ++a) {
            double r = 0.0;
            for (int64_t b = 0; b < D; ++b) r += delta[b] * inv[b * D + a];
            q += r * delta[a];
        }
    } else {
        double *restrict r = delta + D;
        for (int64_t a = 0; a < D; ++a) { delta[a] = mu_num[a] / k_N - x[a]; r[a] = 0.0; }
        for (int64_t b = 0; b < D; ++b) {
            const double db = delta[b];
            const double *restrict row = inv + b * D;
            for (int64_t a = 0; a < D; ++a) r[a] += db * row[a];
        }
        for (int64_t a = 0; a < D; ++a) q += r[a] * delta[a];
    }
    double hd = (double)D / 2.;
    return g->tab_lgam[nu + D] - g->tab_lgam[nu] - hd * g->tab_log[nu] - hd * LOG_PI
           - 0.5 * logdet - (double)(nu + D) / 2. * log(1 + 1. / (double)nu * q);
}

static double student_t_plain(const go_t *g, const double *x, const double *mu_num, double k_N,
                        double logdet, const double *inv, int64_t nu, double *delta) {
    int64_t D = g->D;
    if (g->diag == 2) {         /* product of univariate normals: gaussian_components_fixedvar.py:293-303;
                                   mu_num / k_N is the mean (callers pass k_N = 1 and a ready mean) */
        double acc = 0.0;
        for (int64_t a = 0; a < D; ++a) {
            double dl = x[a] - mu_num[a] / k_N;
            acc += (dl * dl) * inv[a];
        }
        return -0.5 * (double)D * log(2. * 3.14159265358979323846) + 0.5 * logdet - 0.5 * acc;
    }
    if (g->diag) {              /* product of univariate Student-t: gaussian_components_diag.py:340-354 */
        double acc = 0.0;
        for (int64_t a = 0; a < D; ++a) {
            double dl = x[a] - mu_num[a] / k_N;
            acc += log(1. + 1. / (double)nu * (dl * dl) * inv[a]);
        }
        return (double)D * (g->tab_lgam[nu + 1] - g->tab_lgam[nu] - 0.5 * g->tab_log[nu] - 0.5 * LOG_PI)
               - 0.5 * logdet - ((double)nu + 1.) / 2. * acc;
    }
    /* q = sum_a (sum_b delta[b] inv[b][a]) delta[a] -- the reference's einsum pair (gaussian_components.py:240-244).
     * The inner sums run over b ascending for every a, as a scalar loop over a would have them: with b outermost the SAME
     * additions happen in the SAME order per a (r[a] += delta[b] * inv[b][a], product and sum rounded separately), but
     * the memory is walked row by row and the compiler may keep several a in one vector register.  delta[D .. 2D) is r. */
    double q = 0.0;
    if (D < 8) {                /* (tiny D: the plain column loop -- the same sums; vector set-up would cost more than it saves) */
        for (int64_t a = 0; a < D; ++a) delta[a] = mu_num[a] / k_N - x[a];
        for (int64_t a = 0; a < D; ++a) {
            double r = 0.0;
            for (int64_t b = 0; b < D; ++b) r += delta[b] * inv[b * D + a];
            q += r * delta[a];
        }
    } else {
        double *restrict r = delta + D;
        for (int64_t a = 0; a < D; ++a) { delta[a] = mu_num[a] / k_N - x[a]; r[a] = 0.0; }
        for (int64_t b = 0; b < D; ++b) {
            const double db = delta[b];
            const double *restrict row = inv + b * D;
            for (int64_t a = 0; a < D; ++a) r[a] += db * row[a];
        }
        for (int64_t a = 0; a < D; ++a) q += r[a] * delta[a];
    }
    double hd = (double)D / 2.;
    return g->tab_lgam[nu + D] - g->tab_lgam[nu] - hd * g->tab_log[nu] - hd * LOG_PI
           - 0.5 * logdet - (double)(nu + D) / 2. * log(1 + 1. / (double)nu * q);
}

/* (the vector clones pay from D = 8 on; below, the baseline build of the same function) */
static inline double student_t(const go_t *g, const double *x, const double *mu_num, double k_N,
                               double logdet, const double *inv, int64_t nu, double *delta) {
    return g->D < 8 ? student_t_plain(g, x, mu_num, k_N, logdet, inv, nu, delta)
                    : student_t_wide(g, x, mu_num, k_N, logdet, inv, nu, delta);
}

/* ------------------------------------------------------------------------- */
void *go_create(int64_t N, int64_t D, int64_t K_max, const double *X, const double *m0,
                double k0, int64_t v0, const double *S0, double alpha,
                const double *tab_lgam, const double *tab_log, int diag) {
    go_t *g = (go_t *)calloc(1, sizeof(go_t));
    g->N = N; g->D = D; g->K_max = K_max; g->X = X; g->k0 = k0; g->v0 = v0; g->alpha = alpha;
    g->diag = diag; g->SD = diag ? D : D * D;
    g->m0 = (double *)malloc(sizeof(double) * D); memcpy(g->m0, m0, sizeof(double) * D);
    g->S0 = (double *)malloc(sizeof(double) * (D * D + 2 * D)); memcpy(g->S0, S0, sizeof(double) * (diag == 2 ? 2 * D : g->SD));
    g->tab_len = v0 + N + 2;
    if (g->tab_len < 4) g->tab_len = 4;
    g->tab_lgam = (double *)malloc(sizeof(double) * g->tab_len);
    g->tab_log = (double *)malloc(sizeof(double) * g->tab_len);
    for (int64_t t = 0; t < g->tab_len; ++t) {
        double n = (t == 0) ? 1.0 : (double)t;
        g->tab_lgam[t] = tab_lgam ? tab_lgam[t] : lgamma(n / 2.);
        g->tab_log[t] = tab_log ? tab_log[t] : log(n);
    }
    g->prior_m = (double *)malloc(sizeof(double) * D);
    g->prior_S = (double *)malloc(sizeof(double) * D * D);
    for (int64_t a = 0; a < D; ++a) g->prior_m[a] = k0 * m0[a];
    if (diag == 2) {            /* S0 = [var ; var_0]; a new component starts at (precision_0 mu_0, precision_0) */
        g->prec = (double *)malloc(sizeof(double) * D);
        g->prec0 = (double *)malloc(sizeof(double) * D);
        for (int64_t a = 0; a < D; ++a) {
            g->prec[a] = 1. / S0[a];
            g->prec0[a] = 1. / S0[D + a];
            g->prior_m[a] = g->prec0[a] * m0[a];
            g->prior_S[a] = g->prec0[a];
        }
    } else if (diag) {
        for (int64_t a = 0; a < D; ++a) { double o = m0[a] * m0[a]; double ko = k0 * o; g->prior_S[a] = S0[a] + ko; }
    } else {
        for (int64_t a = 0; a < D; ++a)
            for (int64_t b = 0; b < D; ++b) {
                double o = m0[a] * m0[b];
                double ko = k0 * o;
                g->prior_S[a * D + b] = S0[a * D + b] + ko;
            }
    }
    g->m = (double *)calloc(K_max * D, sizeof(double));
    g->S = (double *)calloc(K_max * D * D, sizeof(double));
    g->inv = (double *)calloc(K_max * D * D, sizeof(double));
    g->logdet = (double *)calloc(K_max, sizeof(double));
    g->n = (int64_t *)calloc(K_max, sizeof(int64_t));
    g->z = (int64_t *)malloc(sizeof(int64_t) * N);
    for (int64_t i = 0; i < N; ++i) g->z[i] = -1;
    g->log_prior = (double *)malloc(sizeof(double) * N);
    g->lu = (double *)malloc(sizeof(double) * D * D);
    g->col = (double *)malloc(sizeof(double) * D * GO_MAX_THREADS);
    g->piv = (int64_t *)malloc(sizeof(int64_t) * D);
    g->lp = (double *)malloc(sizeof(double) * (K_max + 1));
    g->save_m = (double *)malloc(sizeof(double) * D);
    g->save_S = (double *)malloc(sizeof(double) * D * D);
    g->save_inv = (double *)malloc(sizeof(double) * D * D);
    g->delta = (double *)malloc(sizeof(double) * 2 * D * GO_MAX_THREADS);
    g->tmp = (double *)malloc(sizeof(double) * D * GO_MAX_THREADS);

    if (diag == 2) {            /* gaussian_components_fixedvar.py:205-212: N(mu_0, precision_0) */
        double lp = 0.0;
        for (int64_t a = 0; a < D; ++a) lp += log(g->prec0[a]);
        for (int64_t i = 0; i < N; ++i)
            g->log_prior[i] = student_t(g, X + i * D, m0, 1.0, lp, g->prec0, 0, g->delta);
        return g;
    }
    if (diag) {                 /* gaussian_components_diag.py:205-212 */
        double sc = (k0 + 1.) / (k0 * (double)v0), lp = 0.0;
        for (int64_t a = 0; a < D; ++a) { double var = sc * S0[a]; lp += log(var); g->save_inv[a] = 1. / var; }
        for (int64_t i = 0; i < N; ++i)
            g->log_prior[i] = student_t(g, X + i * D, m0, 1.0, lp, g->save_inv, v0, g->delta);
        return g;
    }
    /* prior predictive of every point */
    int64_t nu0 = v0 - D + 1;
    double scale = (k0 + 1) / (k0 * (double)nu0);
    double *cov = g->save_S, *iv = g->save_inv;
    for (int64_t t = 0; t < D * D; ++t) cov[t] = scale * S0[t];
    memcpy(g->lu, cov, sizeof(double) * D * D);
    lu_factor(g->lu, D, g->piv);
    double ld0 = lu_logabsdet(g->lu, D);
    lu_inverse(g->lu, g->piv, D, iv, g->col);
    for (int64_t i = 0; i < N; ++i)
        g->log_prior[i] = student_t(g, X + i * D, m0, 1.0, ld0, iv, nu0, g->delta);
    return g;
}

void go_destroy(void *h) {
    go_t *g = (go_t *)h;
    if (!g) return;
    free(g->m0); free(g->S0); free(g->tab_lgam); free(g->tab_log); free(g->prior_m);
    free(g->prior_S); free(g->m); free(g->S); free(g->inv); free(g->logdet); free(g->n);
    free(g->z); free(g->log_prior); free(g->lu); free(g->col); free(g->piv); free(g->lp);
    free(g->save_m); free(g->save_S); free(g->save_inv); free(g->delta); free(g->tmp);
    free(g->prec); free(g->prec0);
    free(g);
}

/* z: labels 0..Kinit-1 (consecutive) or -1; returns 0, or -1 on an invalid vector */
int go_set_assignments(void *h, const int64_t *z) {
    go_t *g = (go_t *)h;
    int64_t zmax = -1;
    for (int64_t i = 0; i < g->N; ++i) { if (z[i] < -1) return -1; if (z[i] > zmax) zmax = z[i]; }
    if (zmax >= g->K_max) return -1;
    for (int64_t k = 0; k <= zmax; ++k) {
        int found = 0;
        for (int64_t i = 0; i < g->N; ++i) if (z[i] == k) { found = 1; break; }
        if (!found) return -1;
    }
    for (int64_t k = 0; k <= zmax; ++k)
        for (int64_t i = 0; i < g->N; ++i)
            if (z[i] == k) seat(g, i, k);
    return 0;
}

/* labels [lo, hi) of a visit's predictive (gaussian_components.py:228-251 + the seating weight), on thread tid's scratch */
typedef struct { go_t *g; const double *x; int use_power; double power; } score_job_t;
static void score_labels(void *argp, int64_t lo, int64_t hi, int tid) {
    const score_job_t *J = (const score_job_t *)argp;
    go_t *g = J->g;
    const int64_t D = g->D;
    double *delta = g->delta + (int64_t)tid * 2 * D, *tmp = g->tmp + (int64_t)tid * D;
    for (int64_t k = lo; k < hi; ++k) {
        double w = J->use_power ? log(pow((double)g->n[k], J->power)) : log((double)g->n[k]);
        int64_t nu = g->diag ? g->v0 + g->n[k] : g->v0 + g->n[k] - D + 1;
        if (g->diag == 2) {
            for (int64_t a = 0; a < D; ++a) tmp[a] = g->m[k * D + a] / g->S[k * D + a];
            g->lp[k] = w + student_t(g, J->x, tmp, 1.0, g->logdet[k], g->inv + k * D, 0, delta);
        } else
            g->lp[k] = w + student_t(g, J->x, g->m + k * D, g->k0 + (double)g->n[k], g->logdet[k],
                                     g->inv + k * g->SD, nu, delta);
    }
}

/* One sweep.  order==NULL: visit 0..N-1.  use_power: weights log(pow(n, power)).
 * n_visits <= N visits are performed (u[t] consumed at visit t).
 * lik_evals (may be NULL) accumulates sum over visits of K_at_visit.
 * Returns 0, or -2 if a new component would exceed K_max. */
int go_sweep(void *h, const int64_t *order, const double *u, int use_power, double power,
             int64_t n_visits, int64_t *lik_evals) {
    go_t *g = (go_t *)h;
    int64_t D = g->D;
    double log_alpha = log(g->alpha);
    for (int64_t t = 0; t < n_visits; ++t) {
        int64_t i = order ? order[t] : t;
        int64_t k_old = g->z[i], K_old = g->K;
        double save_logdet = 0.; int64_t save_n = 0;
        if (k_old >= 0) {
            memcpy(g->save_m, g->m + k_old * D, sizeof(double) * D);
            memcpy(g->save_S, g->S + k_old * g->SD, sizeof(double) * g->SD);
            memcpy(g->save_inv, g->inv + k_old * g->SD, sizeof(double) * g->SD);
            save_logdet = g->logdet[k_old]; save_n = g->n[k_old];
        }
        unseat(g, i);
        int64_t K = g->K;
        if (lik_evals) *lik_evals += K;
        const double *x = g->X + i * D;
        double top = -INFINITY;
        {
            score_job_t J = {g, x, use_power, power};
            /* (a component is ~2 SD flop: shared out only where a share is worth a hand-over) */
            int nt = go_threads;
            while (nt > 1 && (K / nt) * g->SD < 12000) --nt;
            go_parallel_for(score_labels, &J, K, nt);
        }
        for (int64_t k = 0; k < K; ++k) if (g->lp[k] > top) top = g->lp[k];
        g->lp[K] = log_alpha + g->log_prior[i];
        if (g->lp[K] > top) top = g->lp[K];
        double s = 0.0;
        for (int64_t k = 0; k <= K; ++k) s += exp(g->lp[k] - top);
        double lse = log(s) + top;
        double r = u[t];
        int64_t k_new = K;
        for (int64_t k = 0; k <= K; ++k) {
            r = r - exp(g->lp[k] - lse);
            if (r < 0) { k_new = k; break; }
        }
        if (k_new == k_old && g->K == K_old) {
            memcpy(g->m + k_old * D, g->save_m, sizeof(double) * D);
            memcpy(g->S + k_old * g->SD, g->save_S, sizeof(double) * g->SD);
            memcpy(g->inv + k_old * g->SD, g->save_inv, sizeof(double) * g->SD);
            g->logdet[k_old] = save_logdet; g->n[k_old] = save_n;
            g->z[i] = k_old;
        } else {
            if (k_new >= g->K_max) return -2;
            seat(g, i, k_new);
        }
    }
    return 0;
}

/* Diagnostic for the parity harness ("report the CDF margin at first divergence"): the log_prob_z vector
 * the NEXT visit of point i would normalise (crpmm.py:60-76 / pcrpmm.py:96-118) -- point i unseated, every label scored with
 * its seating weight, the new table last.  DESTRUCTIVE: i stays unseated (z[i] = -1, an emptied component dropped), so
 * the oracle is good for nothing else afterwards.  Returns K after the removal; out holds K + 1 values. */
int64_t go_probe_visit(void *h, int64_t i, int use_power, double power, double *out) {
    go_t *g = (go_t *)h;
    int64_t D = g->D;
    unseat(g, i);
    int64_t K = g->K;
    const double *x = g->X + i * D;
    for (int64_t k = 0; k < K; ++k) {
        double w = use_power ? log(pow((double)g->n[k], power)) : log((double)g->n[k]);
        int64_t nu = g->diag ? g->v0 + g->n[k] : g->v0 + g->n[k] - D + 1;
        if (g->diag == 2) {
            for (int64_t a = 0; a < D; ++a) g->tmp[a] = g->m[k * D + a] / g->S[k * D + a];
            out[k] = w + student_t(g, x, g->tmp, 1.0, g->logdet[k], g->inv + k * D, 0, g->delta);
        } else
        out[k] = w + student_t(g, x, g->m + k * D, g->k0 + (double)g->n[k], g->logdet[k], g->inv + k * g->SD, nu, g->delta);
    }
    out[K] = log(g->alpha) + g->log_prior[i];
    return K;
}

double go_log_marg(void *h) {
    go_t *g = (go_t *)h;
    int64_t D = g->D, K = g->K;
    double sum_n = 0., sum_lf = 0.;
    for (int64_t k = 0; k < K; ++k) {
        sum_n += (double)g->n[k];
        if (g->n[k] > 0) sum_lf += lgamma((double)g->n[k]);
    }
    double log_pz = (double)(K - 1) * log(g->alpha) + lgamma(g->alpha) - lgamma(sum_n + g->alpha) + sum_lf;
    double hd = (double)D / 2.;
    double log_px = 0.;
    if (g->diag == 2) {         /* gaussian_components_fixedvar.py:248-270: sums over the members of k */
        for (int64_t k = 0; k < K; ++k) {
            double Nk = (double)g->n[k];
            for (int64_t a = 0; a < D; ++a) {
                double sx = 0., sxx = 0.;
                for (int64_t i = 0; i < g->N; ++i)
                    if (g->z[i] == k) { double x = g->X[i * D + a]; sx += x; sxx += x * x; }
                double p = g->prec[a], p0 = g->prec0[a], mu0 = g->m0[a];
                double den = Nk / p0 + 1. / p;
                log_px += (Nk - 1.) / 2. * log(p) - 0.5 * Nk * log(2. * 3.14159265358979323846)
                          - 0.5 * log(den) - 0.5 * p * sxx - 0.5 * p0 * (mu0 * mu0)
                          + 0.5 * ((sx * sx) * p / p0 + (mu0 * mu0) * p0 / p + 2. * sx * mu0) / den;
            }
        }
        return log_pz + log_px;
    }
    if (g->diag) {              /* gaussian_components_diag.py:261-284 */
        double lS0 = 0.;
        for (int64_t a = 0; a < D; ++a) lS0 += log(g->S0[a]);
        for (int64_t k = 0; k < K; ++k) {
            double k_N = g->k0 + (double)g->n[k];
            int64_t v_N = g->v0 + g->n[k];
            double lSN = 0.;
            for (int64_t a = 0; a < D; ++a) {
                double mean = g->m[k * D + a] / k_N;
                lSN += log(g->S[k * D + a] - k_N * (mean * mean));
            }
            log_px += -(double)g->n[k] * hd * LOG_PI + hd * log(g->k0) - hd * log(k_N)
                      + (double)g->v0 / 2. * lS0 - (double)v_N / 2. * lSN
                      + (double)D * (g->tab_lgam[v_N] - g->tab_lgam[g->v0]);
        }
        return log_pz + log_px;
    }
    double ld_S0 = slogdet_of(g, g->S0);
    for (int64_t k = 0; k < K; ++k) {
        double k_N = g->k0 + (double)g->n[k];
        int64_t v_N = g->v0 + g->n[k];
        const double *m = g->m + k * D, *S = g->S + k * D * D;
        for (int64_t a = 0; a < D; ++a) g->tmp[a] = m[a] / k_N;
        for (int64_t a = 0; a < D; ++a)
            for (int64_t b = 0; b < D; ++b)
                g->save_S[a * D + b] = S[a * D + b] - k_N * (g->tmp[a] * g->tmp[b]);
        double ld = slogdet_of(g, g->save_S);
        double gs = 0.;
        for (int64_t j = 1; j <= D; ++j) gs += g->tab_lgam[v_N + 1 - j] - g->tab_lgam[g->v0 + 1 - j];
        log_px += -(double)g->n[k] * hd * LOG_PI + hd * log(g->k0) - hd * log(k_N)
                  + (double)g->v0 / 2. * ld_S0 - (double)v_N / 2. * ld + gs;
    }
    return log_pz + log_px;
}

int64_t go_K(void *h) { return ((go_t *)h)->K; }
void go_get_assignments(void *h, int64_t *out) { go_t *g = (go_t *)h; memcpy(out, g->z, sizeof(int64_t) * g->N); }
void go_get_counts(void *h, int64_t *out) { go_t *g = (go_t *)h; memcpy(out, g->n, sizeof(int64_t) * g->K); }
void go_get_log_prior(void *h, double *out) { go_t *g = (go_t *)h; memcpy(out, g->log_prior, sizeof(double) * g->N); }
void go_get_stats(void *h, double *m, double *S, double *logdet, double *inv) {
    go_t *g = (go_t *)h; int64_t D = g->D, K = g->K;
    if (m) memcpy(m, g->m, sizeof(double) * K * D);
    if (S) memcpy(S, g->S, sizeof(double) * K * g->SD);
    if (logdet) memcpy(logdet, g->logdet, sizeof(double) * K);
    if (inv) memcpy(inv, g->inv, sizeof(double) * K * g->SD);
}
/* Student-t predictive of X[i] under every current component (no removal). */
void go_log_post_pred(void *h, int64_t i, double *out) {
    go_t *g = (go_t *)h; int64_t D = g->D;
    for (int64_t k = 0; k < g->K; ++k)
        if (g->diag == 2) {
            for (int64_t a = 0; a < D; ++a) g->tmp[a] = g->m[k * D + a] / g->S[k * D + a];
            out[k] = student_t(g, g->X + i * D, g->tmp, 1.0, g->logdet[k], g->inv + k * D, 0, g->delta);
        } else
        out[k] = student_t(g, g->X + i * D, g->m + k * D, g->k0 + (double)g->n[k], g->logdet[k],
                           g->inv + k * g->SD, g->diag ? g->v0 + g->n[k] : g->v0 + g->n[k] - D + 1, g->delta);
}

"""
ORACLE loader (test infrastructure): ctypes binding of oracle/gibbs_oracle.c.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libgibbs_oracle.so")
_lib = None

_f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


def build(force=False):
    src = os.path.join(_HERE, "gibbs_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, "-B", "_build/libgibbs_oracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _SO


_lock = threading.Lock()


def lib():
    """The loaded library (built first if its source is newer).  Thread safe: the heavy GPU tests' oracles start together,
    each on a thread of its own (tests/oracle_pool.py)."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is None:
            _lib = _load()
    return _lib


def _load():
    L = ctypes.CDLL(build())
    L.go_create.restype = ctypes.c_void_p
    L.go_create.argtypes = [ctypes.c_int64, ctypes.c_int64, ctypes.c_int64, _f64p, _f64p,
                            ctypes.c_double, ctypes.c_int64, _f64p, ctypes.c_double,
                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    L.go_destroy.argtypes = [ctypes.c_void_p]
    L.go_set_assignments.argtypes = [ctypes.c_void_p, _i64p]
    L.go_sweep.argtypes = [ctypes.c_void_p, ctypes.c_void_p, _f64p, ctypes.c_int,
                           ctypes.c_double, ctypes.c_int64, ctypes.c_void_p]
    L.go_log_marg.restype = ctypes.c_double
    L.go_log_marg.argtypes = [ctypes.c_void_p]
    L.go_K.restype = ctypes.c_int64
    L.go_K.argtypes = [ctypes.c_void_p]
    L.go_get_assignments.argtypes = [ctypes.c_void_p, _i64p]
    L.go_get_counts.argtypes = [ctypes.c_void_p, _i64p]
    L.go_get_log_prior.argtypes = [ctypes.c_void_p, _f64p]
    L.go_get_stats.argtypes = [ctypes.c_void_p] + [ctypes.c_void_p] * 4
    L.go_log_post_pred.argtypes = [ctypes.c_void_p, ctypes.c_int64, _f64p]
    L.go_probe_visit.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double, _f64p]
    L.go_probe_visit.restype = ctypes.c_int64
    L.go_set_threads.argtypes = [ctypes.c_int]
    L.go_get_threads.restype = ctypes.c_int
    L.go_set_threads(default_threads())
    return L


def default_threads():
    """Threads a visit's K loop (and the LU / inverse of a rebuild) is shared over.  ONE unless GIBBS_ORACLE_THREADS says
    otherwise: the floats are the same for any count, but a hand-over between cores costs ~20 us on the GPU box's 256-core
    host (measured, profiles/r06/oracle_threads.txt: D = 64, K = 200 407 us per visit on one thread, 271 on 32; D = 128 no
    gain at all) -- the tests get their parallelism from running the heavy cases' oracles side by side instead
    (tests/oracle_pool.py)."""
    env = os.environ.get("GIBBS_ORACLE_THREADS")
    return max(1, int(env)) if env else 1


def set_threads(t):
    """Process-wide (the library keeps one setting).  Returns the count in force."""
    L = lib()
    L.go_set_threads(int(t))
    return int(L.go_get_threads())


def get_threads():
    return int(lib().go_get_threads())


def host_tables(v_0, N):
    """lgamma(n/2) and log(n) for n = [1, 1, 2, ..., v_0+N+1] with scipy/numpy, i.e.
    the very values the reference caches (gaussian_components.py:120-122)."""
    from scipy.special import gammaln
    n = np.concatenate([[1], np.arange(1, int(v_0) + N + 2)]).astype(np.float64)
    return np.ascontiguousarray(gammaln(n / 2.)), np.ascontiguousarray(np.log(n))


class COracle(object):
    def __init__(self, X, m_0, k_0, v_0, S_0, alpha, z_init, K_max=None, scipy_tables=True,
                 cov_type="full"):
        self.X = np.ascontiguousarray(X, dtype=np.float64)
        self.N, self.D = self.X.shape
        self.K_max = self.N if K_max is None else int(K_max)
        m_0 = np.ascontiguousarray(m_0, dtype=np.float64)
        S_0 = np.ascontiguousarray(S_0, dtype=np.float64)
        self.diag = cov_type in ("diag", "fixed")           # D-vector statistics
        self.cov_code = {"full": 0, "diag": 1, "fixed": 2}[cov_type]
        assert S_0.shape == {"full": (self.D, self.D), "diag": (self.D,), "fixed": (2 * self.D,)}[cov_type]
        L = lib()
        if scipy_tables:
            self._tabs = host_tables(v_0, self.N)
            tl, tg = self._tabs[0].ctypes.data, self._tabs[1].ctypes.data
        else:
            tl = tg = None
        self.h = L.go_create(self.N, self.D, self.K_max, self.X, m_0, float(k_0), int(v_0), S_0,
                             float(alpha), tl, tg, self.cov_code)
        rc = L.go_set_assignments(self.h, np.ascontiguousarray(z_init, dtype=np.int64))
        assert rc == 0, "invalid initial assignment vector"
        self.lik_evals = ctypes.c_int64(0)

    def __del__(self):
        if getattr(self, "h", None):
            lib().go_destroy(self.h)
            self.h = None

    def sweep(self, u, order=None, power=None, n_visits=None):
        u = np.ascontiguousarray(u, dtype=np.float64)
        n_visits = len(u) if n_visits is None else n_visits
        if order is not None:
            order = np.ascontiguousarray(order, dtype=np.int64)
        rc = lib().go_sweep(self.h, None if order is None else order.ctypes.data, u,
                            0 if power is None else 1, 1.0 if power is None else float(power),
                            n_visits, ctypes.byref(self.lik_evals))
        if rc != 0:
            raise RuntimeError("oracle sweep failed (%d): K_max exceeded" % rc)

    @property
    def K(self):
        return int(lib().go_K(self.h))

    @property
    def z(self):
        out = np.empty(self.N, dtype=np.int64)
        lib().go_get_assignments(self.h, out)
        return out

    @property
    def counts(self):
        out = np.empty(self.K, dtype=np.int64)
        lib().go_get_counts(self.h, out)
        return out

    @property
    def log_prior(self):
        out = np.empty(self.N, dtype=np.float64)
        lib().go_get_log_prior(self.h, out)
        return out

    def log_marg(self):
        return float(lib().go_log_marg(self.h))

    def stats(self):
        K, D = self.K, self.D
        blk = (D,) if self.diag else (D, D)
        m, S = np.empty((K, D)), np.empty((K,) + blk)
        ld, iv = np.empty(K), np.empty((K,) + blk)
        lib().go_get_stats(self.h, m.ctypes.data, S.ctypes.data, ld.ctypes.data, iv.ctypes.data)
        return m, S, ld, iv

    def log_post_pred(self, i):
        out = np.empty(self.K, dtype=np.float64)
        lib().go_log_post_pred(self.h, int(i), out)
        return out

    def probe_visit(self, i, power=None):
        """log_prob_z of the next visit of point i (K + 1 values, the new table last).  Destructive: i stays unseated."""
        out = np.empty(self.K + 1, dtype=np.float64)
        K = lib().go_probe_visit(self.h, int(i), 0 if power is None else 1, 1.0 if power is None else float(power), out)
        return out[:K + 1]


def run_chain(g, n_iter=None, scipy_tables=True):
    """Run a Golden case through the C oracle; returns (oracle, per-sweep dict)."""
    o = COracle(g.X, g.m_0, g.k_0, g.v_0, g.S_0, g.alpha, g.z_init, g.K_max, scipy_tables,
                cov_type=g.cov_type)
    out = {"z": [], "K": [], "counts": [], "log_marg": []}
    for it in range(g.n_iter if n_iter is None else n_iter):
        o.sweep(g.u[it], g.sweep_order(it), g.sweep_power(it))
        out["z"].append(o.z)
        out["K"].append(o.K)
        out["counts"].append(o.counts)
        out["log_marg"].append(o.log_marg())
    return o, out

"""
ctypes binding of libbgmm_hip.so (the C-ABI of include/bgmm.h).

There is NO CPU fallback: if the shared object is missing, cannot be built, or
no HIP device is visible, the calls raise.  ``oracle/`` is never imported here.
"""
import ctypes
import threading
import os

# Chains side by side on one GPU keep one stream each busy, and the HIP runtime maps a process's streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4).  Eight, unless the user has chosen a value or opted out with
# BGMM_KEEP_HW_QUEUES=1.  Only effective when it happens before the process's first HIP call.
if os.environ.get("BGMM_KEEP_HW_QUEUES", "0") in ("", "0"):
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np

from . import _build

_f64 = ctypes.POINTER(ctypes.c_double)
_i64 = ctypes.POINTER(ctypes.c_int64)
_vp = ctypes.c_void_p

_lib = None

ERRORS = {-1: "BGMM_EINVAL", -2: "BGMM_EDEVICE", -3: "BGMM_EKMAX", -4: "BGMM_ENOTPD",
          -5: "BGMM_EUNSUPPORTED"}

# every symbol include/bgmm.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "bgmm_version": (ctypes.c_char_p, []),
    "bgmm_last_error": (ctypes.c_char_p, [_vp]),
    "bgmm_create": (ctypes.c_int, [ctypes.POINTER(_vp), ctypes.c_int, ctypes.c_int64, ctypes.c_int32,
                                   ctypes.c_int32, ctypes.c_int32, _vp, _vp, ctypes.c_double,
                                   ctypes.c_int64, _vp, ctypes.c_double, _vp, _vp]),
    "bgmm_destroy": (None, [_vp]),
    "bgmm_create_shared": (ctypes.c_int, [ctypes.POINTER(_vp), _vp, ctypes.c_int32, _vp, ctypes.c_double, ctypes.c_int64, _vp, ctypes.c_double, _vp, _vp]),
    "bgmm_set_assignments": (ctypes.c_int, [_vp, _vp]),
    "bgmm_sweep": (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int32, ctypes.c_double]),
    "bgmm_stage_sweep_inputs": (ctypes.c_int, [_vp, _vp, _vp]),
    "bgmm_stage_mt19937": (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    "bgmm_set_mt_jump": (ctypes.c_int, [_vp, ctypes.c_int32]),
    "bgmm_get_short_step_stats": (ctypes.c_int, [_vp, _vp]),
    "bgmm_get_totals": (ctypes.c_int, [_vp, _vp]),
    "bgmm_set_mt_lookahead": (ctypes.c_int, [_vp, ctypes.c_int32]),
    "bgmm_get_mt_lookahead_stats": (ctypes.c_int, [_vp, _vp]),
    "bgmm_mt19937_jump_poly": (ctypes.c_int, [ctypes.c_int32, _vp]),
    "bgmm_mt19937_chain_blocks": (ctypes.c_int, []),
    "bgmm_get_staged_uniforms": (ctypes.c_int, [_vp, _vp]),
    "bgmm_stage_permutation_mt19937": (ctypes.c_int, [_vp, _vp, _vp]),
    "bgmm_get_staged_order": (ctypes.c_int, [_vp, _vp]),
    "bgmm_get_permutation_stats": (ctypes.c_int, [_vp, _vp]),
    "bgmm_get_permutation_pipe_state": (ctypes.c_int, [_vp, _vp]),
    "bgmm_set_window_pipeline": (ctypes.c_int, [_vp, ctypes.c_int32]),
    "bgmm_set_proof_lookahead": (ctypes.c_int, [_vp, ctypes.c_int32]),
    "bgmm_get_proof_lookahead_stats": (ctypes.c_int, [_vp, _vp]),
    "bgmm_get_window_pipeline_stats": (ctypes.c_int, [_vp, _vp]),
    "bgmm_get_group_stats": (ctypes.c_int, [_vp, _vp]),
    "bgmm_sweep_staged": (ctypes.c_int, [_vp, ctypes.c_int32, ctypes.c_double]),
    "bgmm_group_sweep_staged": (ctypes.c_int, [_vp, ctypes.c_int32, _vp, _vp, _vp]),
    "bgmm_sweep_staged_begin": (ctypes.c_int, [_vp, ctypes.c_int32, ctypes.c_double]),
    "bgmm_sweep_staged_end": (ctypes.c_int, [_vp]),
    "bgmm_upload_streams": (ctypes.c_int, [_vp, ctypes.c_int32, _vp, _vp]),
    "bgmm_sweep_resident": (ctypes.c_int, [_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_double]),
    "bgmm_log_marg": (ctypes.c_int, [_vp, _f64]),
    "bgmm_log_marg_k": (ctypes.c_int, [_vp, ctypes.c_int32, _f64]),
    "bgmm_get_K": (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int32)]),
    "bgmm_get_assignments": (ctypes.c_int, [_vp, _vp]),
    "bgmm_get_counts": (ctypes.c_int, [_vp, _vp]),
    "bgmm_get_stats": (ctypes.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "bgmm_get_log_prior": (ctypes.c_int, [_vp, _vp]),
    "bgmm_log_post_pred": (ctypes.c_int, [_vp, ctypes.c_int64, _vp]),
    "bgmm_add_item": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int32]),
    "bgmm_del_item": (ctypes.c_int, [_vp, ctypes.c_int64]),
    "bgmm_set_stats": (ctypes.c_int, [_vp, ctypes.c_int32, _vp, _vp, ctypes.c_int64]),
    "bgmm_set_label": (ctypes.c_int, [_vp, ctypes.c_int64, ctypes.c_int32]),
    "bgmm_get_raw_stats": (ctypes.c_int, [_vp, ctypes.c_int32, _vp, _vp]),
    "bgmm_del_component": (ctypes.c_int, [_vp, ctypes.c_int32]),
    "bgmm_set_sweep_visits": (ctypes.c_int, [_vp, ctypes.c_int64]),
    "bgmm_contingency": (ctypes.c_int, [_vp, _vp, ctypes.c_int32, _vp]),
    "bgmm_cluster_dispersion": (ctypes.c_int, [_vp, _vp]),
    "bgmm_get_sweep_stats": (ctypes.c_int, [_vp, _vp]),
    "bgmm_get_prune_stats": (ctypes.c_int, [_vp, _vp]),
    "bgmm_get_path_stats": (ctypes.c_int, [_vp, _vp]),
    "bgmm_get_phase_clocks": (ctypes.c_int, [_vp, _vp]),
    "bgmm_get_safe_stats": (ctypes.c_int, [_vp, _vp]),
    "bgmm_get_proof_pass_stats": (ctypes.c_int, [_vp, _vp]),
    "bgmm_set_safe_budget": (ctypes.c_int, [_vp, ctypes.c_double]),
    "bgmm_set_proof_pass": (ctypes.c_int, [_vp, ctypes.c_int32]),
    "bgmm_set_kernel_timing": (ctypes.c_int, [_vp, ctypes.c_int32]),
    "bgmm_get_kernel_timing": (ctypes.c_int, [_vp, _i64, _f64]),
    "bgmm_set_tuning": (ctypes.c_int, [_vp, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32]),
    "bgmm_set_seq_plan": (ctypes.c_int, [_vp, ctypes.c_int32]),
    "bgmm_set_home_pass": (ctypes.c_int, [_vp, ctypes.c_int32]),
    "bgmm_comm_unique_id": (ctypes.c_int, [_vp]),
    "bgmm_comm_create": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int32, _vp, ctypes.c_int32, _vp]),
    "bgmm_gather_labels": (ctypes.c_int, [_vp, _vp, ctypes.c_int32, _vp]),
    "bgmm_comm_destroy": (ctypes.c_int, [_vp]),
    "bgmm_synchronize": (ctypes.c_int, [_vp]),
}


class BGMMError(RuntimeError):
    def __init__(self, code, message):
        RuntimeError.__init__(self, "%s (%d): %s" % (ERRORS.get(code, "BGMM_E?"), code, message))
        self.code = code


def lib_path():
    return _build.LIB


def load(build_if_missing=True):
    """Load (building first if the .so is absent and hipcc is present)."""
    global _lib
    if _lib is not None:
        return _lib
    path = _build.LIB
    if not os.path.exists(path):
        if not build_if_missing:
            raise OSError("libbgmm_hip.so not built: run `python __graft_entry__.py` "
                          "(there is no CPU fallback)")
        _build.build()
    L = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data


def mt19937_jump_poly(chain):
    """Coefficient words (uint32[624]) of t^(chain * 39936) mod the characteristic polynomial of MT19937 (host only)."""
    L = load()
    out = np.zeros(624, dtype=np.uint32)
    rc = L.bgmm_mt19937_jump_poly(int(chain), _ptr(out))
    if rc != 0:
        raise BGMMError(rc, (L.bgmm_last_error(None) or b"").decode())
    return out


_share_default = threading.local()


class share_x_with(object):
    """``with share_x_with(ctx): ...`` -- contexts made inside the block over the very array ``ctx`` was made over borrow its
    device copy of X (chains.run_chains_on_device builds its chains' model objects this way: the classes keep the
    reference's signatures)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        self.prev = getattr(_share_default, "ctx", None)
        _share_default.ctx = self.ctx
        return self

    def __exit__(self, *exc):
        _share_default.ctx = self.prev
        return False


class Context(object):
    """One chain on one GPU.  Thin, argument-checked wrapper over the C-ABI."""

    def __init__(self, X, m_0, k_0, v_0, S_0, alpha, K_max, device=0, tables=None, cov_type="full", share_with=None):
        """``share_with``: a Context over the same X (same device and covariance type) whose device copy of the data this
        one borrows (bgmm_create_shared) -- chains side by side hold X once."""
        L = load()
        self.L = L
        self.X = np.ascontiguousarray(X, dtype=np.float64)
        self.N, self.D = self.X.shape
        self.K_max = int(K_max)
        m_0 = np.ascontiguousarray(m_0, dtype=np.float64)
        S_0 = np.ascontiguousarray(S_0, dtype=np.float64)
        self.cov_type = cov_type
        assert cov_type in ("full", "diag", "fixed")
        self.diag = cov_type != "full"                      # per-slot blocks are D-vectors
        want = {"full": (self.D, self.D), "diag": (self.D,), "fixed": (2 * self.D,)}[cov_type]
        assert S_0.shape == want, "S_0 has the wrong shape"
        if int(v_0) != v_0:
            raise ValueError("v_0 must be integer valued (the reference indexes its "
                             "log/gammaln tables with it)")
        tl = tg = None
        if tables is not None:
            tl = np.ascontiguousarray(tables[0], dtype=np.float64)
            tg = np.ascontiguousarray(tables[1], dtype=np.float64)
            assert tl.shape == (int(v_0) + self.N + 2,) and tg.shape == tl.shape
        h = _vp()
        if share_with is None:
            dflt = getattr(_share_default, "ctx", None)
            if (dflt is not None and getattr(dflt, "h", None) and dflt.cov_type == cov_type and dflt.X.shape == self.X.shape
                    and np.shares_memory(dflt.X, self.X)):
                share_with = dflt
        if share_with is not None:
            assert share_with.X.shape == self.X.shape and share_with.cov_type == cov_type, "shared contexts are over one data set"
            rc = L.bgmm_create_shared(ctypes.byref(h), share_with.h, self.K_max, _ptr(m_0), float(k_0), int(v_0), _ptr(S_0),
                                      float(alpha), _ptr(tl), _ptr(tg))
        else:
            rc = L.bgmm_create(ctypes.byref(h), int(device), self.N, self.D, self.K_max,
                               {"full": 0, "diag": 1, "fixed": 2}[cov_type],
                               _ptr(self.X), _ptr(m_0), float(k_0), int(v_0), _ptr(S_0), float(alpha),
                               _ptr(tl), _ptr(tg))
        if rc != 0:
            raise BGMMError(rc, (L.bgmm_last_error(None) or b"").decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.bgmm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc):
        if rc != 0:
            raise BGMMError(rc, (self.L.bgmm_last_error(self.h) or b"").decode())

    # -- state ---------------------------------------------------------------
    def set_assignments(self, z):
        z = np.ascontiguousarray(z, dtype=np.int64)
        assert z.shape == (self.N,)
        self._ck(self.L.bgmm_set_assignments(self.h, _ptr(z)))

    def sweep(self, u, order=None, power=None):
        u = np.ascontiguousarray(u, dtype=np.float64)
        assert u.shape == (self.N,)
        if order is not None:
            order = np.ascontiguousarray(order, dtype=np.int64)
            assert order.shape == (self.N,)
        self._ck(self.L.bgmm_sweep(self.h, _ptr(order), _ptr(u), 0 if power is None else 1,
                                   1.0 if power is None else float(power)))

    def stage(self, u, order=None):
        u = np.ascontiguousarray(u, dtype=np.float64)
        if order is not None:
            order = np.ascontiguousarray(order, dtype=np.int64)
        self._ck(self.L.bgmm_stage_sweep_inputs(self.h, _ptr(order), _ptr(u)))

    def stage_mt19937(self, key624, pos, order=None):
        """Uniforms of the next sweep generated on the device from an MT19937 state (624 uint32 words +
        position, as in ``random.getstate()[1]``).  Returns the advanced ``(key624, pos)``."""
        key = np.ascontiguousarray(key624, dtype=np.uint32).copy()
        assert key.shape == (624,)
        p = ctypes.c_int32(int(pos))
        if order is not None:
            order = np.ascontiguousarray(order, dtype=np.int64)
            assert order.shape == (self.N,)
        self._ck(self.L.bgmm_stage_mt19937(self.h, _ptr(order), _ptr(key), ctypes.byref(p)))
        return key, int(p.value)

    def set_mt_jump(self, on=True):
        self._ck(self.L.bgmm_set_mt_jump(self.h, 1 if on else 0))

    def totals(self):
        out = np.zeros(4, dtype=np.int64)
        self._ck(self.L.bgmm_get_totals(self.h, _ptr(out)))
        return {"sweeps": int(out[0]), "lik_evals": int(out[1]), "moves": int(out[2]), "pairs_executed": int(out[3])}

    def short_step_stats(self):
        out = np.zeros(2, dtype=np.int64)
        self._ck(self.L.bgmm_get_short_step_stats(self.h, _ptr(out)))
        return {"stood": int(out[0]), "refused": int(out[1])}

    def set_mt_lookahead(self, sweeps=-1):
        """-1 / True: on, depth chosen from N; 0 / False: off; 1 .. 8: that many sweeps per look-ahead batch."""
        sweeps = -1 if sweeps is True else (0 if sweeps is False else int(sweeps))
        self._ck(self.L.bgmm_set_mt_lookahead(self.h, sweeps))

    def mt_lookahead_stats(self):
        out = np.zeros(2, dtype=np.int64)
        self._ck(self.L.bgmm_get_mt_lookahead_stats(self.h, _ptr(out)))
        return {"hits": int(out[0]), "misses": int(out[1])}

    def stage_permutation_mt19937(self, key624, pos):
        """``np.random.permutation(N)`` drawn on the device from a legacy numpy MT19937 state as the next sweep's visiting
        order (``bgmm_stage_permutation_mt19937``).  Returns the advanced ``(key624, pos)``, or None if the library
        leaves this one to the host (N < 4096)."""
        key = np.ascontiguousarray(key624, dtype=np.uint32).copy()
        assert key.shape == (624,)
        p = ctypes.c_int32(int(pos))
        rc = self.L.bgmm_stage_permutation_mt19937(self.h, _ptr(key), ctypes.byref(p))
        if rc == -5:                    # BGMM_EUNSUPPORTED
            return None
        self._ck(rc)
        return key, int(p.value)

    def permutation_stats(self):
        out = np.zeros(4, dtype=np.int64)
        self._ck(self.L.bgmm_get_permutation_stats(self.h, _ptr(out)))
        return {"lookahead_hits": int(out[0]), "generated_on_the_spot": int(out[1]), "rounds_last": int(out[2]), "rounds_max": int(out[3])}

    def set_proof_lookahead(self, chunk_visits):
        self._ck(self.L.bgmm_set_proof_lookahead(self.h, int(chunk_visits)))

    def proof_lookahead_stats(self):
        out = np.zeros(4, dtype=np.int64)
        self._ck(self.L.bgmm_get_proof_lookahead_stats(self.h, _ptr(out)))
        return {"stretches_from_the_ring": int(out[0]), "stretches_scored_in_full": int(out[1]),
                "labels_rescored": int(out[2]), "chunks_requested": int(out[3])}

    def set_window_pipeline(self, enabled):
        self._ck(self.L.bgmm_set_window_pipeline(self.h, 1 if enabled else 0))

    def window_pipeline_stats(self):
        out = np.zeros(4, dtype=np.int64)
        self._ck(self.L.bgmm_get_window_pipeline_stats(self.h, _ptr(out)))
        return {"batches": int(out[0]), "breaks": int(out[1]), "mode": int(out[2]), "hold": int(out[3])}

    def group_stats(self):
        out = np.zeros(4, dtype=np.int64)
        self._ck(self.L.bgmm_get_group_stats(self.h, _ptr(out)))
        return {"shared_frozen_factor_batches": int(out[0]), "of_them_pipelined": int(out[1]), "shared_safe_stay_batches": int(out[2]),
                "batches_on_its_own_in_a_group": int(out[3])}

    def permutation_pipe_state(self):
        out = np.zeros(4, dtype=np.int64)
        self._ck(self.L.bgmm_get_permutation_pipe_state(self.h, _ptr(out)))
        return {"built": bool(out[0]), "off": bool(out[1]), "worker_failures_in_a_row": int(out[2]), "word_stream_bytes": int(out[3])}

    def staged_order(self):
        o = np.empty(self.N, dtype=np.int64)
        self._ck(self.L.bgmm_get_staged_order(self.h, _ptr(o)))
        return o

    def staged_uniforms(self):
        u = np.empty(self.N, dtype=np.float64)
        self._ck(self.L.bgmm_get_staged_uniforms(self.h, _ptr(u)))
        return u

    def sweep_staged(self, power=None):
        self._ck(self.L.bgmm_sweep_staged(self.h, 0 if power is None else 1,
                                          1.0 if power is None else float(power)))

    def sweep_staged_begin(self, power=None):
        """First half of ``sweep_staged``: queues the sweep and, for a chain at rest, returns without waiting -- stage the
        NEXT sweep's inputs, then ``sweep_staged_end()``."""
        self._ck(self.L.bgmm_sweep_staged_begin(self.h, 0 if power is None else 1, 1.0 if power is None else float(power)))

    def sweep_staged_end(self):
        self._ck(self.L.bgmm_sweep_staged_end(self.h))

    def upload_streams(self, u_all, order_all=None):
        u_all = np.ascontiguousarray(u_all, dtype=np.float64)
        assert u_all.ndim == 2 and u_all.shape[1] == self.N
        if order_all is not None:
            order_all = np.ascontiguousarray(order_all, dtype=np.int64)
            assert order_all.shape == u_all.shape
        self._ck(self.L.bgmm_upload_streams(self.h, u_all.shape[0], _ptr(u_all), _ptr(order_all)))

    def sweep_resident(self, index, power=None):
        self._ck(self.L.bgmm_sweep_resident(self.h, int(index), 0 if power is None else 1,
                                            1.0 if power is None else float(power)))

    @property
    def K(self):
        k = ctypes.c_int32(0)
        self._ck(self.L.bgmm_get_K(self.h, ctypes.byref(k)))
        return int(k.value)

    def assignments(self):
        z = np.empty(self.N, dtype=np.int64)
        self._ck(self.L.bgmm_get_assignments(self.h, _ptr(z)))
        return z

    def counts(self):
        c = np.empty(max(self.K, 1), dtype=np.int64)
        self._ck(self.L.bgmm_get_counts(self.h, _ptr(c)))
        return c[:self.K]

    def stats(self, want_inv=True):
        K, D = self.K, self.D
        blk = (D,) if self.diag else (D, D)
        m, S = np.empty((K, D)), np.empty((K,) + blk)
        ld = np.empty(K)
        iv = np.empty((K,) + blk) if want_inv else None
        self._ck(self.L.bgmm_get_stats(self.h, _ptr(m), _ptr(S), _ptr(ld), _ptr(iv)))
        return m, S, ld, iv

    def log_prior(self):
        out = np.empty(self.N, dtype=np.float64)
        self._ck(self.L.bgmm_get_log_prior(self.h, _ptr(out)))
        return out

    def log_marg(self):
        v = ctypes.c_double(0.0)
        self._ck(self.L.bgmm_log_marg(self.h, ctypes.byref(v)))
        return float(v.value)

    def log_marg_k(self, k):
        v = ctypes.c_double(0.0)
        self._ck(self.L.bgmm_log_marg_k(self.h, int(k), ctypes.byref(v)))
        return float(v.value)

    def log_post_pred(self, i):
        out = np.empty(max(self.K, 1), dtype=np.float64)
        self._ck(self.L.bgmm_log_post_pred(self.h, int(i), _ptr(out)))
        return out[:self.K]

    def add_item(self, i, k):
        self._ck(self.L.bgmm_add_item(self.h, int(i), int(k)))

    def del_item(self, i):
        self._ck(self.L.bgmm_del_item(self.h, int(i)))

    def _raw_shape(self):
        return {"full": (self.D, self.D), "diag": (self.D,), "fixed": (2 * self.D,)}[self.cov_type]

    def set_stats(self, k, m, S, count):
        m = np.ascontiguousarray(m, dtype=np.float64)
        S = np.ascontiguousarray(S, dtype=np.float64)
        assert m.shape == (self.D,) and S.shape == self._raw_shape()
        self._ck(self.L.bgmm_set_stats(self.h, int(k), _ptr(m), _ptr(S), int(count)))

    def raw_stats(self, k):
        """(m[D], S) of component k exactly as stored: what ``set_stats`` takes back bit for bit."""
        m, S = np.empty(self.D), np.empty(self._raw_shape())
        self._ck(self.L.bgmm_get_raw_stats(self.h, int(k), _ptr(m), _ptr(S)))
        return m, S

    def del_component(self, k):
        self._ck(self.L.bgmm_del_component(self.h, int(k)))

    def set_sweep_visits(self, n_visits):
        self._ck(self.L.bgmm_set_sweep_visits(self.h, int(n_visits)))

    def set_label(self, i, k):
        self._ck(self.L.bgmm_set_label(self.h, int(i), int(k)))

    # -- record-dict metrics --------------------------------------------------
    def contingency(self, true_idx, K_true):
        """K_true x K table of (true class, current label) counts.  ``true_idx=None``: the labelling
        uploaded by the previous call (it stays on the device)."""
        if true_idx is not None:
            true_idx = np.ascontiguousarray(true_idx, dtype=np.int64)
            assert true_idx.shape == (self.N,)
        K = self.K
        table = np.zeros((int(K_true), max(K, 1)), dtype=np.int64)
        self._ck(self.L.bgmm_contingency(self.h, _ptr(true_idx), int(K_true), _ptr(table)))
        return table[:, :K]

    def cluster_dispersion(self):
        out = np.empty(max(self.K, 1), dtype=np.float64)
        self._ck(self.L.bgmm_cluster_dispersion(self.h, _ptr(out)))
        return out[:self.K]

    # -- measurement ---------------------------------------------------------
    def sweep_stats(self):
        out = np.zeros(8, dtype=np.int64)
        self._ck(self.L.bgmm_get_sweep_stats(self.h, _ptr(out)))
        keys = ("lik_evals", "moves", "windows", "steps", "score_launches", "scored", "kept_blocks",
                "bound_blocks")
        return dict(zip(keys, (int(v) for v in out)))

    def prune_stats(self):
        out = np.zeros(4, dtype=np.int64)
        self._ck(self.L.bgmm_get_prune_stats(self.h, _ptr(out)))
        return {"kept_blocks": int(out[0]), "bound_blocks": int(out[1]), "mfma_instructions": int(out[2]),
                "certified_visits": int(out[3])}

    def path_stats(self):
        out = np.zeros(4, dtype=np.int64)
        self._ck(self.L.bgmm_get_path_stats(self.h, _ptr(out)))
        return {"pairs_executed": int(out[0]), "frozen_windows": int(out[1]), "frozen_window_visits": int(out[2]),
                "home_decided": int(out[3])}

    def safe_stats(self):
        out = np.zeros(6, dtype=np.int64)
        self._ck(self.L.bgmm_get_safe_stats(self.h, _ptr(out)))
        return {"windows": int(out[0]), "visits_examined": int(out[1]), "unproven_walked": int(out[2]),
                "budget_cuts": int(out[3]), "budget": out[4] * 1e-6, "next_stretch": int(out[5])}

    def proof_pass_stats(self):
        out = np.zeros(2, dtype=np.int64)
        self._ck(self.L.bgmm_get_proof_pass_stats(self.h, _ptr(out)))
        return {"table_batches": int(out[0]), "dense_batches": int(out[1])}

    def set_proof_pass(self, kind=-1):
        self._ck(self.L.bgmm_set_proof_pass(self.h, int(kind)))

    def set_safe_budget(self, cap=0.0):
        self._ck(self.L.bgmm_set_safe_budget(self.h, float(cap)))

    def phase_clocks(self):
        out = np.zeros(16, dtype=np.int64)
        self._ck(self.L.bgmm_get_phase_clocks(self.h, _ptr(out)))
        return [int(v) for v in out]

    def set_kernel_timing(self, on):
        self._ck(self.L.bgmm_set_kernel_timing(self.h, 1 if on else 0))

    def kernel_timing(self):
        n = ctypes.c_int64(0)
        ms = ctypes.c_double(0.0)
        self._ck(self.L.bgmm_get_kernel_timing(self.h, ctypes.byref(n), ctypes.byref(ms)))
        return int(n.value), float(ms.value)

    def set_tuning(self, max_window=0, kernel_kind=0, resolver_mode=0, prune_mode=0):
        self._ck(self.L.bgmm_set_tuning(self.h, int(max_window), int(kernel_kind), int(resolver_mode),
                                        int(prune_mode)))

    def set_seq_plan(self, max_labels=0):
        self._ck(self.L.bgmm_set_seq_plan(self.h, int(max_labels)))

    def set_home_pass(self, mode=0):
        self._ck(self.L.bgmm_set_home_pass(self.h, int(mode)))

    def gather_labels(self, comm, world_size):
        """Final labels of every chain of the communicator (``Comm``): int64[world_size, N] on every rank."""
        out = np.empty((int(world_size), self.N), dtype=np.int64)
        self._ck(self.L.bgmm_gather_labels(self.h, comm.h, int(world_size), _ptr(out)))
        return out

    def synchronize(self):
        self._ck(self.L.bgmm_synchronize(self.h))


def group_sweep_staged(ctxs, powers=None, raise_errors=True):
    """``bgmm_group_sweep_staged``: the staged sweeps of several contexts of ONE device side by side (small-D chains: one
    workgroup each, two launches for all).  ``powers[i]``: chain i's pCRP exponent or None.  Raises for the first
    chain that failed -- or, with ``raise_errors=False``, returns the chains' status codes (a caller that treats them
    chain by chain: BGMM_EKMAX of a chain whose slots grow on demand)."""
    n = len(ctxs)
    L = ctxs[0].L
    handles = (_vp * n)(*[c.h for c in ctxs])
    powers = [None] * n if powers is None else list(powers)
    up = np.array([0 if p is None else 1 for p in powers], dtype=np.int32)
    pw = np.array([1.0 if p is None else float(p) for p in powers], dtype=np.float64)
    rcs = np.zeros(n, dtype=np.int32)
    rc_all = L.bgmm_group_sweep_staged(handles, n, _ptr(up), _ptr(pw), _ptr(rcs))
    if raise_errors:
        for c, rc in zip(ctxs, rcs):
            c._ck(int(rc))
    if rc_all != 0 and (raise_errors or not np.any(rcs)):       # (refused as a whole: e.g. a context twice in the group)
        raise BGMMError(rc_all, "; ".join(filter(None, ((L.bgmm_last_error(c.h) or b"").decode() for c in ctxs))) or "group sweep refused")
    return [int(rc) for rc in rcs]


class Comm(object):
    """RCCL communicator of the final label gather (include/bgmm.h: bgmm_comm_*).  ``unique_id()`` on rank 0,
    ship the 128 bytes to the other ranks, then ``Comm(rank, world_size, id_bytes, device)`` on every rank."""

    @staticmethod
    def unique_id():
        L = load()
        buf = ctypes.create_string_buffer(128)
        rc = L.bgmm_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p))
        if rc != 0:
            raise BGMMError(rc, (L.bgmm_last_error(None) or b"").decode())
        return buf.raw

    def __init__(self, rank, world_size, id_bytes, device=0):
        self.L = load()
        self.world_size = int(world_size)
        h = ctypes.c_void_p()
        buf = ctypes.create_string_buffer(bytes(id_bytes), 128)
        rc = self.L.bgmm_comm_create(int(rank), int(world_size), ctypes.cast(buf, ctypes.c_void_p), int(device),
                                     ctypes.byref(h))
        if rc != 0:
            raise BGMMError(rc, (self.L.bgmm_last_error(None) or b"").decode())
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.bgmm_comm_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

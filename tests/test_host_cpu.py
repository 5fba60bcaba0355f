"""CPU-side checks: host logic, the C-ABI surface, loud failure without a GPU."""
import ctypes
import os
import random
import re

import numpy as np
import numpy.testing as npt
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_are_exported_and_bound():
    from pybgmm_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "bgmm.h")).read()
    declared = set(re.findall(r"\b(bgmm_[A-Za-z_0-9]+)\s*\(", hdr))
    declared.discard("bgmm_ctx")
    assert declared == set(_lib.SIGNATURES), declared.symmetric_difference(_lib.SIGNATURES)
    L = _lib.load()
    for name in declared:
        assert hasattr(L, name), "libbgmm_hip.so does not export %s" % name
    assert L.bgmm_version().decode().startswith("bgmm-hip")


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from pybgmm_amd import _lib
    X = np.zeros((4, 2))
    with pytest.raises(_lib.BGMMError) as ei:
        _lib.Context(X, np.zeros(2), 1.0, 3, np.eye(2), 1.0, 4)
    assert ei.value.code == -2


def test_argument_validation_before_device():
    from pybgmm_amd import _lib
    L = _lib.load()
    h = ctypes.c_void_p()
    X = np.zeros((4, 2))
    m, S = np.zeros(2), np.eye(2)
    # v_0 < D
    rc = L.bgmm_create(ctypes.byref(h), 0, 4, 2, 4, 0, X.ctypes.data, m.ctypes.data, 1.0, 1,
                       S.ctypes.data, 1.0, None, None)
    assert rc == -1 and b"v_0" in L.bgmm_last_error(None)
    rc = L.bgmm_create(ctypes.byref(h), 0, 4, 2, 4, 7, X.ctypes.data, m.ctypes.data, 1.0, 3,
                       S.ctypes.data, 1.0, None, None)
    assert rc == -5
    rc = L.bgmm_create(ctypes.byref(h), 0, 4, 300, 4, 0, X.ctypes.data, m.ctypes.data, 1.0, 300,
                       S.ctypes.data, 1.0, None, None)
    assert rc == -5


def test_product_never_imports_oracle():
    """The shipped path must not import, load or link anything under oracle/."""
    pkg = os.path.join(ROOT, "pybgmm_amd")
    pat = re.compile(r"^\s*(from|import)\s+[\w.]*oracle|CDLL\([^)]*oracle|#include\s+[\"<][^\">]*oracle")
    for dirpath, _dirs, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                for line in open(os.path.join(dirpath, f)):
                    assert not pat.search(line), "%s refers to oracle/: %s" % (f, line)


def test_take_uniforms_is_the_python_stream():
    from pybgmm_amd.utils.rng import take_uniforms
    random.seed(11)
    a = [random.random() for _ in range(501)]
    nxt = random.random()
    random.seed(11)
    b = take_uniforms(501)
    npt.assert_array_equal(a, b)
    assert random.random() == nxt
    r1, r2 = random.Random(3), random.Random(3)
    npt.assert_array_equal([r1.random() for _ in range(64)], take_uniforms(64, r2))
    assert r1.random() == r2.random()


def test_chain_rngs_equal_global_streams():
    from pybgmm_amd.chains import chain_rngs
    rng, nprng = chain_rngs(5, 2)
    random.seed(7)
    np.random.seed(7)
    assert rng.random() == random.random()
    npt.assert_array_equal(nprng.permutation(20), np.random.permutation(20))
    npt.assert_array_equal(nprng.randint(0, 5, 9), np.random.randint(0, 5, 9))


def test_compact_labels_equals_shift_down_loop():
    from pybgmm_amd.igmm.igmm import compact_labels
    rs = np.random.RandomState(0)
    for _ in range(200):
        K = rs.randint(1, 12)
        z = rs.randint(0, K, rs.randint(1, 30))
        want = z.copy()
        for k in range(want.max()):            # the reference's loop (igmm.py:89-94), restated
            while len(np.nonzero(want == k)[0]) == 0:
                want[np.where(want > k)] -= 1
            if want.max() == k:
                break
        npt.assert_array_equal(compact_labels(z), want)


def test_rand_init_matches_golden_init():
    """'rand' initialisation consumes np.random exactly like the reference (igmm.py:86-94)."""
    from golden_util import Golden
    from pybgmm_amd.igmm.igmm import compact_labels
    for case, K in (("c2twin_crpmm_2d", 20), ("general_prior_3d", 4), ("c1_crpmm_1d", 3)):
        g = Golden(case)
        np.random.seed(int(g.d["seed_numpy"]))
        npt.assert_array_equal(compact_labels(np.random.randint(0, K, g.N)), g.z_init)


def test_error_conventions():
    from pybgmm_amd.igmm import CRPMM
    from pybgmm_amd.prior import NIW
    with pytest.raises(AssertionError):
        NIW(np.zeros(3), 1.0, 2, np.eye(3))                 # v_0 < D (niw.py:21)
    prior = NIW(np.zeros(2), 1.0, 3, np.eye(2))
    with pytest.raises(ValueError):
        CRPMM(np.zeros(5), prior, 1.0, None)                 # 1-D X (igmm.py:75-76)
    with pytest.raises(AssertionError):
        CRPMM(np.zeros((5, 2)), prior, 1.0, None, covariance_type="banana")
    with pytest.raises(AssertionError):                       # gaussian_components_diag.py:92
        CRPMM(np.zeros((5, 2)), prior, 1.0, None, covariance_type="diag")


def test_synth_recipes_are_deterministic():
    from pybgmm_amd.utils import gendata
    X1, z1 = gendata.synth_mixture(300, 5, 7, seed=3)
    X2, z2 = gendata.synth_mixture(300, 5, 7, seed=3)
    npt.assert_array_equal(X1, X2)
    npt.assert_array_equal(z1, z2)
    assert X1.flags["C_CONTIGUOUS"] and X1.dtype == np.float64
    assert np.bincount(z1).min() >= 300 // 7


def test_short_log_exp_against_libm(tmp_path):
    """pybgmm_amd/csrc/fast_math.h (the log / exp / log1p of the mover-dense path's update waves) compiled for
    the host and compared with libm in long double: below one ulp."""
    import ctypes
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    src = tmp_path / "fm.cpp"
    hdr = os.path.join(ROOT, "pybgmm_amd", "csrc", "fast_math.h")
    src.write_text('#include "%s"\n#include <cmath>\n'
                   'static double ulps(double got, long double ref) { double r = (double)ref; '
                   'return std::fabs((double)((long double)got - ref)) / std::fabs(std::nextafter(r, 1e300) - r); }\n'
                   'extern "C" double err_log(double x) { return ulps(fm_log(x), logl((long double)x)); }\n'
                   'extern "C" double err_exp(double x) { return ulps(fm_exp(x), expl((long double)x)); }\n'
                   'extern "C" double err_log1p(double x) { return ulps(fm_log1p_small(x), log1pl((long double)x)); }\n'
                   'extern "C" double val_exp(double x) { return fm_exp(x); }\n' % hdr)
    lib = tmp_path / "fm.so"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(lib), str(src)], check=True)
    L = ctypes.CDLL(str(lib))
    for f in (L.err_log, L.err_exp, L.err_log1p, L.val_exp):
        f.restype = ctypes.c_double
        f.argtypes = [ctypes.c_double]
    rs = np.random.RandomState(3)
    worst = {"log": 0.0, "exp": 0.0, "log1p": 0.0}
    for x in np.concatenate([np.exp(rs.uniform(-30, 30, 20000)), 1.0 + rs.uniform(-0.2, 0.2, 20000)]):
        worst["log"] = max(worst["log"], L.err_log(float(x)))
    for x in np.concatenate([rs.uniform(-690, 690, 20000), rs.uniform(-2, 2, 20000)]):
        worst["exp"] = max(worst["exp"], L.err_exp(float(x)))
    for x in np.concatenate([rs.uniform(-0.28, 0.28, 20000), rs.uniform(-1e-4, 1e-4, 20000)]):
        worst["log1p"] = max(worst["log1p"], L.err_log1p(float(x)))
    assert worst["log"] < 1.0 and worst["exp"] < 1.0 and worst["log1p"] < 1.0, worst
    assert L.val_exp(-800.0) == 0.0 and L.val_exp(0.0) == 1.0


def test_wishart_draws_consume_the_streams_like_the_reference():
    """prior/wishart.py:16-32 restated: row r of the Bartlett factor takes r normals from np.random, then one
    random.gammavariate(0.5 (v_0 - D + 1), 2.0); the factor is float32."""
    import random
    from pybgmm_amd.prior import wishart
    D, v0 = 4, 9
    sigma = np.diag([1.0, 2.0, 0.5, 3.0])
    random.seed(5)
    np.random.seed(6)
    W = wishart.wishrnd(sigma, v0)
    random.seed(5)
    np.random.seed(6)
    a = np.zeros((D, D), dtype=np.float32)
    for r in range(D):
        if r:
            a[r, :r] = np.random.normal(size=(r,))
        a[r, r] = np.sqrt(random.gammavariate(0.5 * (v0 - D + 1), 2.0))
    C = np.linalg.cholesky(sigma)
    # (the reference's association and its solve(sample, I): a draw is the reference's bit for bit)
    npt.assert_array_equal(W, C.dot(a).dot(a.T).dot(C.T))
    after = (random.random(), np.random.random_sample())
    random.seed(5)
    np.random.seed(6)
    Wi = wishart.iwishrnd(sigma, v0, C)
    npt.assert_array_equal(Wi, np.linalg.solve(W, np.eye(D)))
    assert (random.random(), np.random.random_sample()) == after


def test_mt19937_jump_polynomials_against_numpy():
    """bgmm_mt19937_jump_poly: t^(chain * J), J = bgmm_mt19937_chain_blocks() * 624 words, modulo the characteristic polynomial of MT19937 (Berlekamp-Massey +
    shift / multiply-and-reduce on the host, no device).  The state J words ahead must be the GF(2) convolution of the
    next 19937 + 623 words with the coefficient bits -- checked against numpy's generator for two chains."""
    from pybgmm_amd import _lib
    rs = np.random.RandomState(2014)
    key = rs.get_state()[1].astype(np.uint32)
    cb = _lib.load().bgmm_mt19937_chain_blocks()          # blocks per chain: J = cb * 624 words
    n_blocks = cb * 2 + 34

    def blocks(mt, nb):          # untempered words: the recurrence, vectorised in its three independent runs
        out = [mt.copy()]
        for _ in range(nb):
            new = mt.copy()
            for lo, hi in ((0, 227), (227, 454), (454, 623)):
                k = np.arange(lo, hi)
                y = (new[k] & np.uint32(0x80000000)) | (new[k + 1] & np.uint32(0x7fffffff))
                new[k] = new[(k + 397) % 624] ^ (y >> np.uint32(1)) ^ np.where(y & np.uint32(1), np.uint32(0x9908b0df), np.uint32(0))
            y = (new[623] & np.uint32(0x80000000)) | (new[0] & np.uint32(0x7fffffff))
            new[623] = new[396] ^ (y >> np.uint32(1)) ^ (np.uint32(0x9908b0df) if (y & np.uint32(1)) else np.uint32(0))
            mt = new
            out.append(mt.copy())
        return np.concatenate(out)

    x = blocks(key, n_blocks)
    # (sanity of the restated recurrence: numpy's own next block, tempered, is what random_sample consumes)
    for chain in (1, 2):
        g = _lib.mt19937_jump_poly(chain)
        bits = np.unpackbits(g.view(np.uint8), bitorder="little")
        assert not bits[19937:].any()
        idx = np.nonzero(bits[:19937])[0]
        J = chain * cb * 624
        acc = np.zeros(624, dtype=np.uint32)
        for i in idx:
            acc ^= x[i:i + 624]
        np.testing.assert_array_equal(acc[1:], x[J + 1:J + 624])
        assert ((acc[0] ^ x[J]) >> np.uint32(31)) == 0


def test_oracle_pool_runs_cases_side_by_side_and_serves_a_test_run_on_its_own():
    """tests/oracle_pool.py (round 6): a decorated test's case function is called with the test's own parameters (the ones it
    names), its sweeps go through the C oracle on a thread of their own, the result is handed out once -- and a test whose
    case was never prefetched (pytest -k, ORACLE_POOL=0) gets the same through the same path.  Items with a case sort behind
    the others, the cheapest oracle first."""
    import numpy as np
    import oracle_pool
    from pybgmm_amd.utils import gendata

    def case(N, D):
        X, zt = gendata.synth_mixture(N, D, 3, seed=N)
        us = np.random.RandomState(D).random_sample((2, N))
        return {"X": X, "prior": gendata.demo_prior_params(D), "z0": zt, "K_max": 12, "sweeps": [(us[0],), (us[1], None, None, N // 2)]}
    case.cost = lambda N, D: N * D

    class Spec(object):
        def __init__(self, params):
            self.params = params

    class Item(object):
        def __init__(self, nodeid, fn, params):
            self.nodeid, self.function, self.callspec = nodeid, fn, Spec(params)

    @oracle_pool.with_oracle(case)
    def decorated():
        pass

    def plain():
        pass
    items = [Item("t::big", decorated, {"N": 300, "D": 3, "budget": 1.0}), Item("t::plain", plain, {}),
             Item("t::small", decorated, {"N": 120, "D": 2, "budget": 0.0})]
    assert oracle_pool.cost_of(items[0]) == 900 and oracle_pool.cost_of(items[2]) == 240
    oracle_pool.prefetch(items)
    c_big, r_big = oracle_pool.result_for(items[0])
    c_small, r_small = oracle_pool.result_for(items[2])
    assert c_big["X"].shape == (300, 3) and c_small["X"].shape == (120, 2)
    assert len(r_big) == 2 and r_big[0]["z"].shape == (300,) and isinstance(r_big[1]["log_marg"], float)
    # handed out once; asked again (or never prefetched), the case is run on the spot with the same outcome
    c2, r2 = oracle_pool.result_for(items[2])
    assert np.array_equal(r2[1]["z"], r_small[1]["z"]) and r2[1]["log_marg"] == r_small[1]["log_marg"]
    assert any("case" in what for _, what in oracle_pool.TIMES)

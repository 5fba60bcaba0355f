"""
Parity of the HIP path (through the C-ABI) with the reference.

  * every golden trajectory captured from the reference (tests/golden), label for label,
    sweep by sweep, with both likelihood kernels;
  * the reference's own known-answer values;
  * the C oracle on fresh seeded problems;
  * at BASELINE sizes, size-independent properties (window-size independence,
    kernel independence, determinism, statistics consistent with a from-scratch rebuild).
Tolerances: integer trajectory bit-exact; log marginal 1e-6 relative (north_star), observed
~1e-12; m / S bit-exact; logdet / inverse 1e-8.
"""
import os

import numpy as np
import numpy.testing as npt
import pytest

from oracle_pool import with_oracle

from golden_util import ALL_CASES, API_CASES, DIAG_CASES, FIXED_CASES, GOLDEN_DIR, Golden
from pybgmm_amd.gaussian.gaussian_components import reference_tables

pytestmark = pytest.mark.gpu

LM_RTOL = 1e-6


def make_ctx(g, kind=0, window=0, tables=True, resolver=0, prune=0, home=0):
    from pybgmm_amd import _lib
    ctx = _lib.Context(g.X, g.m_0, g.k_0, g.v_0, g.S_0, g.alpha, g.K_max,
                       tables=reference_tables(g.v_0, g.N) if tables else None, cov_type=g.cov_type)
    ctx.set_tuning(max_window=window, kernel_kind=kind, resolver_mode=resolver, prune_mode=prune)
    ctx.set_home_pass(home)
    ctx.set_assignments(g.z_init)
    return ctx


@pytest.mark.parametrize("kind", [0, 1, 2], ids=["auto", "valu", "mfma"])
@pytest.mark.parametrize("case", ALL_CASES)
def test_golden_trajectory(case, kind):
    # ("auto": D <= 4 full-covariance cases take the sequential one-wavefront sweep, the rest the
    # windowed kernels the host would pick by itself)
    g = Golden(case)
    ctx = make_ctx(g, kind)
    npt.assert_allclose(ctx.log_prior()[:4096], g.d["cached_log_prior"], rtol=1e-11, atol=1e-11)
    for it in range(g.n_iter):
        ctx.sweep(g.u[it], g.sweep_order(it), g.sweep_power(it))
        z = ctx.assignments()
        bad = np.nonzero(z != g.z[it])[0]
        assert bad.size == 0, "sweep %d: %d labels differ, first at i=%d" % (it, bad.size, bad[0])
        assert ctx.K == g.K[it]
        npt.assert_array_equal(ctx.counts(), g.counts_at(it))
        lm = ctx.log_marg()
        assert abs(lm - g.log_marg[it]) <= LM_RTOL * abs(g.log_marg[it])
        assert abs(lm - g.log_marg[it]) <= 1e-9 * abs(g.log_marg[it]), "observed accuracy regressed"
    if "final_S" in g.d.files:
        m, S, ld, iv = ctx.stats()
        npt.assert_array_equal(m, g.d["final_m"])       # bit-exact sufficient statistics
        npt.assert_array_equal(S, g.d["final_S"])
        npt.assert_allclose(ld, g.d["final_logdet"], rtol=1e-8, atol=1e-8)
        npt.assert_allclose(iv, g.d["final_inv"], rtol=1e-7, atol=1e-9)
    ctx.close()


@pytest.mark.parametrize("resolver", [1, 2, 3, 4], ids=["per-mover-kernels", "in-launch-resolver", "frozen-factor-windows",
                                                        "safe-stay-windows"])
@pytest.mark.parametrize("case", ALL_CASES)
def test_golden_trajectory_both_mover_paths(case, resolver):
    """Force the per-mover kernel chain (1), the in-launch resolver (2), the frozen-factor windows
    (3: every visit of every sweep goes through gram_kernel / gram_resolve_kernel) and the safe-stay windows (4: a
    proof pass in front of every window, only the visits it cannot prove to stay are walked): same chain."""
    g = Golden(case)
    ctx = make_ctx(g, 0, 0, resolver=resolver)
    for it in range(g.n_iter):
        ctx.sweep(g.u[it], g.sweep_order(it), g.sweep_power(it))
        z = ctx.assignments()
        bad = np.nonzero(z != g.z[it])[0]
        assert bad.size == 0, "sweep %d: %d labels differ, first at i=%d" % (it, bad.size, bad[0])
        npt.assert_array_equal(ctx.counts(), g.counts_at(it))
        lm = ctx.log_marg()
        assert abs(lm - g.log_marg[it]) <= 1e-9 * abs(g.log_marg[it])
    if "final_S" in g.d.files:
        m, S, ld, iv = ctx.stats()
        npt.assert_array_equal(m, g.d["final_m"])
        npt.assert_array_equal(S, g.d["final_S"])
        npt.assert_allclose(ld, g.d["final_logdet"], rtol=1e-8, atol=1e-8)
        npt.assert_allclose(iv, g.d["final_inv"], rtol=1e-7, atol=1e-9)
    ctx.close()


@pytest.mark.parametrize("prune", [0, 1], ids=["pruned", "unpruned"])
@pytest.mark.parametrize("case", ["c4twin_crpmm_64d", "c4rand_crpmm_64d", "c3twin_pcrpmm_16d", "c3rand_pcrpmm_16d"])
def test_pruning_does_not_change_trajectory(case, prune):
    """Exact pruning of negligible components (MFMA kernel, forced also at D=16) on and off."""
    g = Golden(case)
    ctx = make_ctx(g, kind=2, prune=prune, resolver=1)
    for it in range(g.n_iter):
        ctx.sweep(g.u[it], g.sweep_order(it), g.sweep_power(it))
        npt.assert_array_equal(ctx.assignments(), g.z[it])
        assert abs(ctx.log_marg() - g.log_marg[it]) <= 1e-9 * abs(g.log_marg[it])
    ctx.close()


@pytest.mark.parametrize("case", ["crpmm_12d", "c3twin_pcrpmm_16d", "c3rand_pcrpmm_16d", "c4twin_crpmm_64d",
                                  "c4rand_crpmm_64d"])
@pytest.mark.parametrize("home", [1, 2], ids=["home-pass", "no-home-pass"])
def test_every_window_pruned(case, home):
    """prune_mode 2: the pruned-window kernels (home pass or not, bucket sort, bounds, block-sparse scores,
    sparse draw) evaluate EVERY window, also in the mover-dense sweeps of these fixtures -- singleton
    homes, deletions and new components included."""
    g = Golden(case)
    ctx = make_ctx(g, kind=2, prune=2, home=home)
    for it in range(g.n_iter):
        ctx.sweep(g.u[it], g.sweep_order(it), g.sweep_power(it))
        z = ctx.assignments()
        bad = np.nonzero(z != g.z[it])[0]
        assert bad.size == 0, "sweep %d: %d labels differ, first at i=%d" % (it, bad.size, bad[0])
        assert abs(ctx.log_marg() - g.log_marg[it]) <= 1e-9 * abs(g.log_marg[it])
        assert ctx.sweep_stats()["bound_blocks"] > 0 or ctx.prune_stats()["certified_visits"] > 0
    ctx.close()


@pytest.mark.parametrize("N,D,K,sep,label", [
    (6000, 12, 1500, 4.0, "many tiny clusters: more than four homes per wave, K > 1024"),
    (20000, 16, 40, 0.9, "overlapping clusters: survivors of every bound level, long work lists"),
    (8000, 48, 300, 4.0, "K > 256: several coarse passes"),
    (4000, 128, 20, 4.0, "D = 128"),
    (12000, 40, 30, 4.0, "D = 40: padded columns in the rows' 16-byte pieces (home_kernel's general loads)"),
    (9000, 32, 25, 4.0, "D = 32: four tiles in flight per wavefront"),
])
@pytest.mark.parametrize("home,prune", [(1, 2), (2, 2), (0, 3)], ids=["home-pass", "no-home-pass", "benchmarked-mode"])
def test_every_window_pruned_against_c_oracle(N, D, K, sep, label, home, prune):
    """prune_mode 2 = every window through the pruned kernels (with / without the home pass in front); prune_mode 3 = the
    mode bench.py times (`evaluated`: the schedule of the default configuration with certified stays off)."""
    from divergence import assert_same_labels, first_divergence
    from oracle import c_oracle
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    X, zt = gendata.synth_mixture(N, D, K, seed=900 + D, mu_scale=sep)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    rs = np.random.RandomState(K)
    z0 = zt.copy()
    z0[rs.rand(N) < 0.02] = -1                      # a few unassigned visits ("one-by-one")
    us = rs.random_sample((2, N))
    order = rs.permutation(N)
    K_max = min(N, 2 * K + 64)
    o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, z0, K_max)
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, K_max, tables=reference_tables(v_0, N))
    ctx.set_tuning(kernel_kind=2, prune_mode=prune)
    ctx.set_home_pass(home)
    ctx.set_assignments(z0)

    def fresh_ctx():
        c = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, K_max, tables=reference_tables(v_0, N))
        c.set_tuning(kernel_kind=2, prune_mode=prune)
        c.set_home_pass(home)
        c.set_assignments(z0)
        return c
    orders, powers = [None, order], [None, 1.03]
    for it in range(2):
        power = powers[it]
        o.sweep(us[it], orders[it], power)
        ctx.sweep(us[it], orders[it], power)
        assert_same_labels(ctx.assignments(), o.z, "%s: sweep %d" % (label, it), lambda: first_divergence(
            fresh_ctx, lambda: c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, z0, K_max), us, orders, powers, it))
        lo = o.log_marg()
        assert abs(ctx.log_marg() - lo) <= 1e-9 * abs(lo)
    ctx.close()


@pytest.mark.parametrize("window", [256, 1024])
@pytest.mark.parametrize("case", ["c2twin_crpmm_2d", "c3rand_pcrpmm_16d", "each_in_own_50"])
def test_window_size_does_not_change_trajectory(case, window):
    g = Golden(case)
    ctx = make_ctx(g, 1 if g.D <= 4 else 0, window)      # (kind 0 would take the window-less sequential sweep)
    for it in range(g.n_iter):
        ctx.sweep(g.u[it], g.sweep_order(it), g.sweep_power(it))
        npt.assert_array_equal(ctx.assignments(), g.z[it])
    ctx.close()


def test_libm_tables():
    g = Golden("kat1_igmm_2d")
    ctx = make_ctx(g, tables=False)
    for it in range(g.n_iter):
        ctx.sweep(g.u[it])
    npt.assert_array_equal(ctx.assignments(), g.z[-1])
    ctx.close()


def test_first_visit_probabilities():
    """prob_z of the very first visit (reference probe) from log_post_pred + del_item."""
    g = Golden("kat1_igmm_2d")
    ctx = make_ctx(g)
    p_ref, _u, _k = g.probes()[0]
    ctx.del_item(0)
    K = ctx.K
    lp = np.zeros(K + 1)
    lp[:K] = np.log(ctx.counts()) + ctx.log_post_pred(0)
    lp[-1] = np.log(g.alpha) + ctx.log_prior()[0]
    p = np.exp(lp - np.logaddexp.reduce(lp))
    npt.assert_allclose(p, p_ref, rtol=1e-10, atol=1e-14)
    ctx.close()


def test_log_post_pred_unvectorized_is_an_independent_check():
    """gaussian_components.py:355-363 / SURVEY 4.2: the vectorised predictive against the per-component scalar path.  Here
    the scalar path runs on the host (LAPACK slogdet / inverse of every component's covariance from the downloaded raw
    statistics), the vectorised one is the device kernel: after two sweeps of a captured chain, for points inside,
    between and far from the components, the two agree to 1e-9 -- and on the reference's own 2014 known answer."""
    from pybgmm_amd.gaussian.gaussian_components import GaussianComponents, log_post_pred_unvectorized
    from pybgmm_amd.prior import NIW
    g = Golden("crpmm_12d")
    comp = GaussianComponents(g.X, NIW(g.m_0, g.k_0, g.v_0, g.S_0), assignments=g.z_init, K_max=g.K_max)
    for it in range(2):
        comp._ctx.sweep(g.u[it], g.sweep_order(it), g.sweep_power(it))
    for i in (0, 1, g.N // 2, g.N - 1):
        npt.assert_allclose(comp.log_post_pred(i), log_post_pred_unvectorized(comp, i), rtol=1e-9, atol=1e-9)
    # pybgmm/tests/test_gaussian_components.py:104 (expected value)
    X = np.array([[1.2, 0.9], [-0.1, 0.8], [0.5, 0.4]])
    comp = GaussianComponents(X, NIW(np.zeros(2), 2., 5, 5. * np.eye(2)), assignments=[0, 0, -1], K_max=3)
    npt.assert_almost_equal(log_post_pred_unvectorized(comp, 2)[0], -2.07325364088)
    npt.assert_almost_equal(comp.log_post_pred_k(2, 0), -2.07325364088)


# ---- the reference's component known answers (pybgmm/tests/test_gaussian_components.py) ----
def test_component_known_answers():
    from pybgmm_amd.gaussian import GaussianComponents
    from pybgmm_amd.prior import NIW
    X = np.array([[-0.3406, -0.0593, -0.0686]])
    gmm = GaussianComponents(X, NIW(np.zeros(3), 0.05, 4, 0.001 * np.eye(3)))
    npt.assert_almost_equal(gmm.log_prior(0), -0.472067277015)

    X = np.array([[-0.3406, -0.3593, -0.0686], [-0.3381, 0.2993, 0.925], [-0.5, -0.101, 0.75]])
    gmm = GaussianComponents(X, NIW(np.zeros(3), 0.05, 6, 0.5 * np.eye(3)), [0, 0, 0])
    npt.assert_almost_equal(gmm.log_marg_k(0), -8.42365141729)

    prior = NIW(m_0=np.array([0.0, 0.0]), k_0=2., v_0=5, S_0=5. * np.eye(2))
    gmm = GaussianComponents(np.array([[1.2, 0.9], [-0.1, 0.8], [0.5, 0.4]]), prior)
    gmm.add_item(0, 0)
    gmm.add_item(1, 0)
    npt.assert_almost_equal(gmm.log_post_pred_k(2, 0), -2.07325364088)
    mu, sigma = gmm.map(0)
    npt.assert_almost_equal(mu, [0.275, 0.425])
    npt.assert_almost_equal(sigma, [[0.55886364, 0.04840909], [0.04840909, 0.52068182]])


def test_add_del_item_roundtrip_and_delete_swap():
    from pybgmm_amd.gaussian import GaussianComponents
    from pybgmm_amd.prior import NIW
    rs = np.random.RandomState(2)
    X = rs.rand(11, 4)
    prior = NIW(X.mean(axis=0), 0.05, 14, 0.5 * np.eye(4))
    gmm = GaussianComponents(X, prior, [0, 0, 0, 1, 0, 1, 3, 4, 3, 2, -1])
    assert gmm.K == 5
    gmm.del_item(9)                       # label 2 was a singleton: label 4 moves into 2
    assert gmm.K == 4
    npt.assert_array_equal(gmm.assignments, [0, 0, 0, 1, 0, 1, 3, 2, 3, -1, -1])
    npt.assert_array_equal(gmm.counts[:4], [4, 2, 1, 2])
    gmm.add_item(10, 4)                   # opens a new component
    assert gmm.K == 5 and gmm.assignments[10] == 4


# ---- class level: seeds -> same trajectory as the reference classes ------------------------
def test_crpmm_class_reproduces_reference_kat():
    import random
    from pybgmm_amd.igmm import CRPMM
    from pybgmm_amd.prior import NIW
    from pybgmm_amd.utils import gendata
    g = Golden("kat1_igmm_2d")
    random.seed(1)
    np.random.seed(1)
    X, z_true = gendata.demo_mixture(100, 2, 4, rs=np.random)
    npt.assert_array_equal(X, g.X)
    prior = NIW(*gendata.demo_prior_params(2, v_0=5))
    mm = CRPMM(X, prior, 1.0, None, assignments="rand", K=3)
    npt.assert_array_equal(mm.components.assignments, g.z_init)
    record, dist = mm.collapsed_gibbs_sampler(10, z_true, num_saved=0)
    npt.assert_array_equal(mm.components.assignments, g.z[-1])
    npt.assert_allclose(record["log_marg"], g.log_marg, rtol=1e-9)
    assert record["components"] == list(g.K)
    # record-dict metrics through the device contingency / dispersion kernels
    npt.assert_allclose(record["nmi"], g.d["rec_nmi"], rtol=1e-12)
    npt.assert_allclose(record["mi"], g.d["rec_mi"], rtol=1e-12)
    npt.assert_allclose(record["vi"], g.d["rec_vi"], rtol=1e-11, atol=1e-12)
    assert [int(v) for v in record["loss"]] == [int(v) for v in g.d["rec_loss"]]
    assert [int(v) for v in record["bic"]] == [int(v) for v in g.d["rec_bic"]]
    assert [str(s) for s in record["nk"]] == [str(s) for s in g.d["rec_nk"]]
    npt.assert_almost_equal(mm.log_marg(), -411.811711231)
    npt.assert_allclose(mm.log_marg_host(), mm.log_marg(), rtol=1e-12)


@pytest.mark.parametrize("case,K", [("c1_crpmm_1d", 3), ("general_prior_3d", 4), ("c2twin_crpmm_2d", 20)])
def test_record_metrics_match_reference(case, K):
    """nmi / mi / vi / loss of every sweep, device route, against what the reference recorded."""
    import random
    from pybgmm_amd.igmm import CRPMM
    from pybgmm_amd.prior import NIW
    g = Golden(case)
    random.seed(int(g.d["seed_random"]))
    np.random.seed(int(g.d["seed_numpy"]))
    mm = CRPMM(g.X, NIW(*g.prior), g.alpha, None, assignments="rand", K=K, K_max=g.K_max)
    record, _ = mm.collapsed_gibbs_sampler(g.n_iter, g.d["true_assignments"], num_saved=0)
    npt.assert_array_equal(mm.components.assignments, g.z[-1])
    npt.assert_allclose(record["nmi"], g.d["rec_nmi"], rtol=1e-11, atol=1e-13)
    npt.assert_allclose(record["mi"], g.d["rec_mi"], rtol=1e-11, atol=1e-13)
    npt.assert_allclose(record["vi"], g.d["rec_vi"], rtol=1e-10, atol=1e-11)
    assert [int(v) for v in record["loss"]] == [int(v) for v in g.d["rec_loss"]]


def test_adapcrpmm_class_reproduces_reference():
    """per-sweep exponent from the share of small clusters."""
    import random
    from pybgmm_amd.igmm import ADAPCRPMM
    from pybgmm_amd.prior import NIW
    g = Golden("adap_2d")
    random.seed(11)
    np.random.seed(11)
    mm = ADAPCRPMM(g.X, NIW(*g.prior), g.alpha, None, assignments="rand", K=12, K_max=g.K_max)
    npt.assert_array_equal(mm.components.assignments, g.z_init)
    record, _ = mm.collapsed_gibbs_sampler(g.n_iter, g.d["true_assignments"], r_up=1.4,
                                           adapcrp_perct=0.08, adapcrp_burnin=-1, num_saved=0)
    npt.assert_array_equal(mm.components.assignments, g.z[-1])
    assert record["components"] == list(g.K)
    npt.assert_allclose(record["log_marg"], g.log_marg, rtol=1e-9)
    npt.assert_allclose(record["nmi"], g.d["rec_nmi"], rtol=1e-11)
    mm2 = ADAPCRPMM(g.X, NIW(*g.prior), g.alpha, None, assignments=g.z_init, K_max=g.K_max)
    with pytest.raises(UnboundLocalError):            # the reference's default burn-in does the same
        mm2.collapsed_gibbs_sampler(1, g.d["true_assignments"], num_saved=0)


def test_pcrpmm_class_reproduces_reference():
    import random
    from pybgmm_amd.igmm import PCRPMM
    from pybgmm_amd.prior import NIW
    g = Golden("pcrp_burnin_2d")
    random.seed(6)
    np.random.seed(6)
    mm = PCRPMM(g.X, NIW(*g.prior), g.alpha, None, assignments="rand", K=6, K_max=g.K_max)
    npt.assert_array_equal(mm.components.assignments, g.z_init)
    record, _ = mm.collapsed_gibbs_sampler(g.n_iter, g.d["true_assignments"], n_power=1.5,
                                           power_burnin=1, num_saved=0)
    npt.assert_array_equal(mm.components.assignments, g.z[-1])
    npt.assert_allclose(record["log_marg"], g.log_marg, rtol=1e-9)


def test_k_max_overflow_is_an_error():
    from pybgmm_amd import _lib
    g = Golden("each_in_own_50")
    ctx = _lib.Context(g.X, g.m_0, g.k_0, g.v_0, g.S_0, 1e300, 3)  # huge alpha: always a new table
    ctx.set_assignments(np.zeros(50, dtype=np.int64))
    with pytest.raises(_lib.BGMMError) as ei:
        ctx.sweep(np.full(50, 0.999999))
    assert ei.value.code == -3 and "K_max" in str(ei.value)
    ctx.close()


def test_k_max_overflow_leaves_a_consistent_state():
    """The reference raises from add_item (gaussian_components.py:158-160) with the point already taken out by
    del_item: after BGMM_EKMAX the counts, statistics and derived state are those of the labelling the getters
    return (the visited point unassigned), not a half-applied move."""
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    N, D = 80, 6
    X, _ = gendata.synth_mixture(N, D, 2, seed=3)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1e300, 3, tables=reference_tables(v_0, N))   # always a new table
    ctx.set_tuning(kernel_kind=2)
    ctx.set_assignments(np.zeros(N, dtype=np.int64))
    with pytest.raises(_lib.BGMMError) as ei:
        ctx.sweep(np.full(N, 0.999999))
    assert ei.value.code == -3
    z = ctx.assignments()
    assert np.count_nonzero(z < 0) == 1 and ctx.K == 3
    counts = ctx.counts()
    assert counts.sum() == N - 1
    m, S, logdet, _ = ctx.stats(want_inv=False)
    for k in range(3):
        idx = np.nonzero(z == k)[0]
        assert counts[k] == idx.size
        np.testing.assert_allclose(m[k], k_0 * m_0 + X[idx].sum(axis=0), rtol=1e-12, atol=1e-12)
        S_ref = S_0 + k_0 * np.outer(m_0, m_0) + X[idx].T @ X[idx]
        np.testing.assert_allclose(S[k], S_ref, rtol=1e-11, atol=1e-11)
        k_N = k_0 + idx.size
        v_N = v_0 + idx.size
        covar = (S_ref - np.outer(m[k], m[k]) / k_N) * (k_N + 1.0) / (k_N * (v_N - D + 1.0))     # gaussian_components.py:319-331
        assert abs(logdet[k] - np.linalg.slogdet(covar)[1]) <= 1e-8 * max(1.0, abs(logdet[k]))
    ctx.close()


# ---- oracle on fresh seeded problems --------------------------------------------------------
@pytest.mark.parametrize("N,D,K,init", [(6000, 16, 40, "rand"), (3000, 32, 10, "rand"),
                                        (5000, 64, 50, "true"), (1500, 128, 6, "rand"),
                                        (4000, 5, 9, "rand")])
def test_against_c_oracle(N, D, K, init):
    from oracle import c_oracle
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    X, z_true = gendata.synth_mixture(N, D, K, seed=100 + D)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    rs = np.random.RandomState(D)
    z0 = z_true if init == "true" else np.unique(rs.randint(0, K, N), return_inverse=True)[1]
    us = rs.random_sample((2, N))
    order = rs.permutation(N)
    Kmax = 4 * K
    o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, z0, Kmax)
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, Kmax, tables=reference_tables(v_0, N))
    ctx.set_assignments(z0)
    for it in range(2):
        power = 1.01 if it == 1 else None
        o.sweep(us[it], order, power)
        ctx.sweep(us[it], order, power)
        npt.assert_array_equal(ctx.assignments(), o.z)
        lo = o.log_marg()
        assert abs(ctx.log_marg() - lo) <= 1e-9 * abs(lo)
    assert ctx.sweep_stats()["lik_evals"] > 0
    ctx.close()


@pytest.mark.parametrize("init", ["true", "rand", "one"])
def test_c2_full_size_against_c_oracle(init):
    """BASELINE config C2 at its full size (CRPMM, N=1e5, D=2, K~20; a quarter of the visits move per
    sweep), seed-locked against the C port of the reference: identical labels after every sweep."""
    from oracle import c_oracle
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    N, D, K = 100000, 2, 20
    X, z_true = gendata.synth_mixture(N, D, K, seed=1)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    rs = np.random.RandomState(42)
    if init == "true":
        z0 = z_true
    elif init == "rand":
        z0 = np.unique(rs.randint(0, K, N), return_inverse=True)[1]
    else:
        z0 = np.zeros(N, dtype=np.int64)
    Kmax = 8 * K
    o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, z0, Kmax)
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, Kmax, tables=reference_tables(v_0, N))
    ctx.set_assignments(z0)
    for it in range(3):
        u = rs.random_sample(N)
        o.sweep(u, None, None)
        ctx.sweep(u, None, None)
        z = ctx.assignments()
        bad = np.nonzero(z != o.z)[0]
        assert bad.size == 0, "sweep %d: %d labels differ, first at i=%d" % (it, bad.size, bad[0])
        lo = o.log_marg()
        assert abs(ctx.log_marg() - lo) <= 1e-9 * abs(lo)
        st = ctx.sweep_stats()
        assert st["steps"] == 1                                # (the one-workgroup sweep)
        assert st["moves"] > 1000 or init == "one"             # (the chain really moves)
    ctx.close()


# ---- BASELINE sizes: size-independent properties -------------------------------------------
@pytest.mark.parametrize("N,D,K,pcrp", [(100000, 2, 20, False), (1000000, 16, 100, True), (1000000, 64, 200, False),
                                        (2000000, 128, 200, True)],
                         ids=["C2", "C3", "C4", "C5"])
def test_full_size_properties(N, D, K, pcrp):
    """C3 and C5 are PCRPMM configurations: a fresh permutation every sweep and the powered seating
    weights from the second sweep on (pcrpmm.py:86-112) -- the storage-order certificates of a lean step
    against the permuted visiting order of the full evaluation."""
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    X, z_true = gendata.synth_mixture(N, D, K, seed=1)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    rs = np.random.RandomState(1)
    n_sw = 4 if pcrp else 3
    us = rs.random_sample((n_sw, N))
    orders = [rs.permutation(N).astype(np.int64) for _ in range(n_sw)] if pcrp else [None] * n_sw
    tabs = reference_tables(v_0, N)
    # perturb the true labelling so that the sweep has real moves to make
    z0 = z_true.copy()
    flip = rs.choice(N, size=2000, replace=False)
    z0[flip] = rs.randint(0, K, size=flip.size)
    results = []
    # default configuration / every pair evaluated in small windows / the benchmarked mode (bench.py's `evaluated`:
    # prune_mode 3 = pruned windows with certified stays off, every visit's row read and scored every sweep)
    for kind, window, prune in ((0, 0, 0), (1 if D >= 24 else 2, 2048, 1), (0, 0, 3)):
        ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K, tables=tabs)
        ctx.set_tuning(max_window=window, kernel_kind=kind, prune_mode=prune)
        ctx.set_assignments(z0)
        moves = 0
        certified = 0
        for it in range(n_sw):     # (the later sweeps of the default configuration run on certified stays)
            ctx.sweep(us[it], orders[it], 1.01 if (pcrp and it > 0) else None)
            moves += ctx.sweep_stats()["moves"]
            certified += ctx.prune_stats()["certified_visits"]
        z = ctx.assignments()
        c = ctx.counts()
        lm = ctx.log_marg()
        results.append((z, c, lm, moves))
        if kind == 0 and prune == 0 and D >= 64:   # (C3's clusters are too close for certificates at the 2^-53 level)
            assert certified > 0
        if prune == 3:
            assert certified == 0
        assert c.sum() == N and z.min() >= 0 and z.max() == len(c) - 1
        npt.assert_array_equal(np.bincount(z, minlength=len(c)), c)
        if kind == 0 and prune == 0:
            # incremental statistics agree with a from-scratch rebuild of the same labelling
            ctx2 = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K, tables=tabs)
            ctx2.set_assignments(z)
            assert abs(ctx2.log_marg() - lm) <= 1e-9 * abs(lm)
            ctx2.close()
        ctx.close()
    za, ca, lma, mva = results[0]
    assert mva > 0
    for zb, cb, lmb, mvb in results[1:]:
        npt.assert_array_equal(za, zb)      # kernel kind, window size and pruning mode do not change the chain
        assert mva == mvb
        assert abs(lma - lmb) <= 1e-9 * abs(lma)


@pytest.mark.parametrize("N,D,K,pcrp", [(1000000, 64, 200, False), (1000000, 16, 100, True)], ids=["C4-rand", "C3-rand"])
def test_full_size_burnin_two_mover_paths(N, D, K, pcrp):
    """The reference's default start ("rand", igmm.py:86-94) at BASELINE size: nearly every visit of the first
    sweep moves.  The default configuration (frozen-factor windows, kernels_gram.hip) against the
    one-workgroup resolver that updates both factors in LDS per move (kernels_resolve.hip), pruning off,
    windows of 2048: two independent implementations of the mover path, same labels."""
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    X, _ = gendata.synth_mixture(N, D, K, seed=1)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    rs = np.random.RandomState(2)
    z0 = np.unique(rs.randint(0, K, N), return_inverse=True)[1]
    u = rs.random_sample(N)
    order = rs.permutation(N).astype(np.int64) if pcrp else None
    tabs = reference_tables(v_0, N)
    out = []
    for resolver, window, prune in ((0, 0, 0), (2, 2048, 1)):
        ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K, tables=tabs)
        ctx.set_tuning(max_window=window, resolver_mode=resolver, prune_mode=prune)
        ctx.set_assignments(z0)
        ctx.sweep(u, order, None)
        st, ps = ctx.sweep_stats(), ctx.path_stats()
        out.append((ctx.assignments(), ctx.counts(), ctx.log_marg(), st["moves"], ps["frozen_windows"]))
        ctx.close()
    (za, ca, lma, mva, fa), (zb, cb, lmb, mvb, fb) = out
    assert fa > N // 80 and fb == 0             # (the first went through the frozen-factor windows, the second did not)
    assert mva > 0.9 * N
    bad = np.nonzero(za != zb)[0]
    assert bad.size == 0, "%d labels differ, first at i=%d" % (bad.size, bad[0])
    npt.assert_array_equal(ca, cb)
    assert mva == mvb and abs(lma - lmb) <= 1e-9 * abs(lma)


def _case_burnin(N, D, K, sep):
    from pybgmm_amd.utils import gendata
    X, _ = gendata.synth_mixture(N, D, K, seed=31 + D, mu_scale=sep)
    rs = np.random.RandomState(D)
    z0 = np.unique(rs.randint(0, K, N), return_inverse=True)[1]
    us = [rs.random_sample(N) for _ in range(2)]
    return {"X": X, "prior": gendata.demo_prior_params(D), "z0": z0, "K_max": 4 * K, "sweeps": [(u,) for u in us]}


_case_burnin.cost = lambda N, D, K, sep: 2 * N * (K + 2 * D) * D * D


@pytest.mark.parametrize("N,D,K,sep", [(30000, 64, 200, 4.0), (10000, 128, 60, 4.0), (60000, 16, 100, 1.0)],
                         ids=["D64-K200", "D128", "D16-overlapping"])
@with_oracle(_case_burnin)
def test_burnin_against_c_oracle(N, D, K, sep, oracle_ref):
    """Frozen-factor windows against the C port of the reference, two sweeps from a random start at the
    BASELINE dimensions (every visit of the first sweep moves; components die and are born)."""
    from pybgmm_amd import _lib
    case, ref = oracle_ref
    m_0, k_0, v_0, S_0 = case["prior"]
    ctx = _lib.Context(case["X"], m_0, k_0, v_0, S_0, 1.0, 4 * K)
    ctx.set_assignments(case["z0"])
    for it in range(2):
        ctx.sweep(case["sweeps"][it][0])
        z = ctx.assignments()
        bad = np.nonzero(z != ref[it]["z"])[0]
        assert bad.size == 0, "sweep %d: %d labels differ, first at i=%d" % (it, bad.size, bad[0])
        lo = ref[it]["log_marg"]
        assert abs(ctx.log_marg() - lo) <= 1e-9 * abs(lo)
        if it == 0:
            assert ctx.path_stats()["frozen_windows"] > N // 80
    ctx.close()


@pytest.mark.parametrize("N,D,K_true,K_init", [(24000, 64, 12, 60), (12000, 128, 8, 40)], ids=["D64", "D128"])
def test_pipelined_windows_against_plain_ones(N, D, K_true, K_init):
    """the pipelined frozen-factor windows (cross forms of window w made against the factors of window w - 2,
    window w - 1's terms carried in -- on by default for a chain on its own) A/B against plain windows on ONE chain: a
    burn-in from a random start with five times too many components -- windows that end early, open a component or
    delete one break the pipeline on the device.  Labels, counts and log marginal equal after every
    sweep (the reference's loop, igmm/crpmm.py:57-88, knows neither kind), pipelined batches AND broken chains both
    counted (bgmm_get_window_pipeline_stats)."""
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    X, _ = gendata.synth_mixture(N, D, K_true, seed=5 + D, mu_scale=3.0)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    rs = np.random.RandomState(D + 1)
    z0 = np.unique(rs.randint(0, K_init, N), return_inverse=True)[1]
    us = rs.random_sample((3, N))
    runs = []
    for piped in (True, False):
        ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K_init)
        ctx.set_window_pipeline(piped)
        ctx.set_assignments(z0)
        per_sweep = []
        for it in range(3):
            ctx.sweep(us[it])
            per_sweep.append((ctx.assignments(), ctx.counts(), ctx.log_marg(), ctx.K))
        st = ctx.window_pipeline_stats()
        runs.append((per_sweep, st))
        ctx.close()
    (a, sa), (b, sb) = runs
    assert sa["batches"] > 0 and sa["breaks"] > 0, sa
    assert sb["batches"] == 0 and sb["mode"] == 0, sb
    for it in range(3):
        bad = np.nonzero(a[it][0] != b[it][0])[0]
        assert bad.size == 0, "sweep %d: %d labels differ, first at i=%d" % (it, bad.size, bad[0])
        npt.assert_array_equal(a[it][1], b[it][1])
        assert a[it][3] == b[it][3]
        assert abs(a[it][2] - b[it][2]) <= 1e-9 * abs(b[it][2])


@pytest.mark.parametrize("D,mixed", [(64, False), (16, True)], ids=["D64", "D16-one-chain-plain"])
def test_chains_side_by_side_share_pipelined_windows(D, mixed):
    """Round 6: chains of one shape that burn in TOGETHER share their frozen-factor windows' launches (api_group.hip) -- and
    those windows are pipelined like a single chain's (gram_group_pipe_launch: every kernel of the schedule one launch for all
    chains, workgroup (x, chain); cross forms of window w + 1 and the finish of window w - 1 on a second stream beside the
    resolvers).  Four chains from random starts with five times too many components -- windows that delete a component break
    ONE chain's pipeline on the device while the others carry on -- against the same group with the pipeline switched off for
    every chain (bgmm_set_window_pipeline(0): shared PLAIN windows), and, `mixed`, with it switched off for one chain only
    (that one gets plain windows beside the others' pipelined ones): labels, counts, K, log marginal after every sweep."""
    import random
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    N, K_true, K_init, G = 16000, 10, 50, 4
    X, _ = gendata.synth_mixture(N, D, K_true, seed=9 + D, mu_scale=3.0)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)

    def run(pipe_of):
        chains_ = []
        for c in range(G):
            ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K_init, share_with=chains_[0][0] if chains_ else None)
            ctx.set_window_pipeline(pipe_of(c))
            ctx.set_assignments(np.unique(np.random.RandomState(70 + c).randint(0, K_init, N), return_inverse=True)[1])
            _, key, _ = random.Random(900 + c).getstate()
            chains_.append([ctx, np.asarray(key[:-1], dtype=np.uint32), int(key[-1])])
        out = []
        for it in range(3):
            for ch in chains_:
                ch[1], ch[2] = ch[0].stage_mt19937(ch[1], ch[2], None)
            _lib.group_sweep_staged([ch[0] for ch in chains_], [None] * G)
            out.append([(ch[0].assignments(), ch[0].counts(), ch[0].K, ch[0].log_marg()) for ch in chains_])
        stats = [ch[0].window_pipeline_stats() for ch in chains_]
        for ch in reversed(chains_):
            ch[0].close()
        return out, stats
    piped, sp = run((lambda c: c != 1) if mixed else (lambda c: True))
    plain, sq = run(lambda c: False)
    assert all(s_["batches"] == 0 for s_ in sq), sq
    assert all(s_["batches"] > 0 for c, s_ in enumerate(sp) if not (mixed and c == 1)), sp
    if mixed:
        assert sp[1]["batches"] == 0, sp
    assert sum(s_["breaks"] for s_ in sp) > 0, "components die in these sweeps: some chain's pipeline is meant to break"
    for it in range(3):
        for c in range(G):
            bad = np.nonzero(piped[it][c][0] != plain[it][c][0])[0]
            assert bad.size == 0, "sweep %d chain %d: %d labels differ, first at i=%d" % (it, c, bad.size, bad[0])
            npt.assert_array_equal(piped[it][c][1], plain[it][c][1])
            assert piped[it][c][2] == plain[it][c][2]
            assert abs(piped[it][c][3] - plain[it][c][3]) <= 1e-9 * abs(plain[it][c][3])
    assert len({piped[2][c][0].tobytes() for c in range(G)}) == G, "chains with different seeds must differ"


@pytest.mark.parametrize("N,D,K,sep,pcrp", [(60000, 64, 40, 0.5, False), (40000, 128, 20, 0.28, False), (100000, 16, 100, 1.0, True)],
                         ids=["D64", "D128", "D16-pcrp"])
def test_chains_side_by_side_share_safe_stay_steps(N, D, K, sep, pcrp):
    """Round 6: chains of one shape whose clusters overlap (the regime a chain lives in: 0.5 % of the visits move at
    equilibrium) share the launches of their safe-stay steps as well -- the dense proof pass of every chain's stretch, the
    verdicts, the lists, the frozen-factor kernels on the listed rows: one launch each for all of them (api_group.hip kind 1,
    kernels_safe.hip launch_safe_group_step).  Four chains with their own generators over one data set, four sweeps from the
    truth, against the same chains swept one by one (which take the look-ahead's route on two streams of their own): labels,
    counts, moves and log marginal after every sweep, and the group did share safe-stay batches."""
    import random
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    G = 4
    X, zt = gendata.synth_mixture(N, D, K, seed=77 + D, mu_scale=sep)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)

    def build():
        out = []
        for c in range(G):
            ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K, share_with=out[0][0] if out else None)
            ctx.set_assignments(zt)
            _, key, _ = random.Random(40 + c).getstate()
            out.append([ctx, np.asarray(key[:-1], dtype=np.uint32), int(key[-1])])
        return out
    grouped, solo = build(), build()
    rs = np.random.RandomState(D)
    moved = 0
    for it in range(4):
        # (pcrp: a fresh visiting order per chain and sweep and the seating power of pcrpmm.py:100-104)
        orders = [rs.permutation(N).astype(np.int64) if pcrp else None for _ in range(G)]
        powers = [1.0 + 0.01 * it if pcrp else None for _ in range(G)]
        for chains_ in (grouped, solo):
            for c, ch in enumerate(chains_):
                ch[1], ch[2] = ch[0].stage_mt19937(ch[1], ch[2], orders[c])
        _lib.group_sweep_staged([ch[0] for ch in grouped], powers)
        for c, ch in enumerate(solo):
            ch[0].sweep_staged(powers[c])
        for c in range(G):
            bad = np.nonzero(grouped[c][0].assignments() != solo[c][0].assignments())[0]
            assert bad.size == 0, "sweep %d chain %d: %d labels differ, first at i=%d" % (it, c, bad.size, bad[0])
            npt.assert_array_equal(grouped[c][0].counts(), solo[c][0].counts())
            assert grouped[c][0].sweep_stats()["moves"] == solo[c][0].sweep_stats()["moves"]
            lg, ls = grouped[c][0].log_marg(), solo[c][0].log_marg()
            assert abs(lg - ls) <= 1e-12 * abs(ls)
        moved += grouped[0][0].sweep_stats()["moves"]
    assert moved > 100, "the clusters are meant to overlap"
    gs = [ch[0].group_stats() for ch in grouped]
    assert sum(g_["shared_safe_stay_batches"] for g_ in gs) >= G, gs
    assert len({ch[0].assignments().tobytes() for ch in grouped}) == G, "chains with different seeds must differ"
    for ch in reversed(grouped + solo):
        ch[0].close()


def _case_safe_stay(N, D, K, sep, flip):
    from pybgmm_amd.utils import gendata
    X, zt = gendata.synth_mixture(N, D, K, seed=77 + D, mu_scale=sep)
    rs = np.random.RandomState(D + flip)
    z0 = zt.copy()
    if flip:
        idx = rs.choice(N, size=flip, replace=False)
        z0[idx] = rs.randint(0, K, size=flip)
    us = [rs.random_sample(N) for _ in range(3)]
    return {"X": X, "prior": gendata.demo_prior_params(D), "z0": z0, "K_max": 4 * K, "sweeps": [(u,) for u in us]}


_case_safe_stay.cost = lambda N, D, K, sep, flip: 3 * N * K * D * D


@pytest.mark.parametrize("N,D,K,sep,flip,budget", [(100000, 16, 100, 1.0, 0, 0.0), (100000, 16, 100, 1.0, 0, 1.0),
                                                   (60000, 64, 40, 0.5, 0, 0.0), (60000, 64, 40, 4.0, 200, 0.0),
                                                   (10000, 128, 20, 0.28, 0, 0.0)],
                         ids=["D16-5pct-movers", "D16-5pct-movers-budget-1.0", "D64-0.6pct-movers", "D64-200-wrong-labels",
                              "D128-overlapping"])
@with_oracle(_case_safe_stay)
def test_safe_stay_windows_against_c_oracle(N, D, K, sep, flip, budget, oracle_ref):
    """The regime between "nothing moves" and "everything moves" (VERDICT r2 #1): clusters that overlap so that 0.5 - 5 %
    of the visits move at equilibrium, and a chain at the truth with wrong labels sprinkled in.  Three sweeps against the
    C port of the reference, labels identical; the safe-stay windows must have run and must have left most visits off
    the resolver's chain."""
    from pybgmm_amd import _lib
    case, ref = oracle_ref
    m_0, k_0, v_0, S_0 = case["prior"]
    ctx = _lib.Context(case["X"], m_0, k_0, v_0, S_0, 1.0, 4 * K)
    ctx.set_safe_budget(budget)
    ctx.set_assignments(case["z0"])
    windows = walked = moved = 0
    for it in range(3):
        ctx.sweep(case["sweeps"][it][0])
        moved += ctx.sweep_stats()["moves"]
        z = ctx.assignments()
        bad = np.nonzero(z != ref[it]["z"])[0]
        assert bad.size == 0, "sweep %d: %d labels differ, first at i=%d" % (it, bad.size, bad[0])
        lo = ref[it]["log_marg"]
        assert abs(ctx.log_marg() - lo) <= 1e-9 * abs(lo)
        ss = ctx.safe_stats()
        windows += ss["windows"]
        walked += ss["unproven_walked"]
    assert moved >= 50, "the case is meant to have movers"
    assert windows > 0, "the safe-stay windows never ran"
    # (how much the proof pass leaves on the resolver's chain: under half at D <= 64; D = 128 clusters at mu_scale 0.28 sit
    # a few nats apart and most visits really are undecided -- 73 % walked, measured -- so that case only asks for "some")
    assert walked < (0.6 if D <= 64 else 0.9) * 3 * N, "the proof pass proved next to nothing"
    ctx.close()


@pytest.mark.parametrize("chunk", [2048, 8192], ids=["chunks-of-2048", "chunks-of-8192"])
def test_proof_lookahead_gives_the_same_chain(chunk):
    """Round 6: the dense proof pass takes its exact forms from a LOOK-AHEAD -- a second stream scores a chunk of visits
    for every slot beside the resolver, a stretch re-scores only the labels that took a rank-1 term since (kernels_safe.hip).
    A/B on one overlapping chain with the proof pass pinned dense: look-ahead off (every stretch scores its own pairs) against
    on -- labels, counts, log marginal, windows, walked visits and budget cuts identical after every sweep (the same proofs
    from the same forms: which stream made them changes the cost, never the chain, igmm/crpmm.py:57-88), and the stats
    show that stretches were served from the ring and labels were re-scored."""
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    N, D, K = 60000, 64, 40
    X, zt = gendata.synth_mixture(N, D, K, seed=141, mu_scale=0.5)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    us = np.random.RandomState(8).random_sample((3, N))
    out = []
    for ahead in (0, chunk):
        ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
        ctx.set_proof_pass(1)
        ctx.set_proof_lookahead(ahead)
        ctx.set_assignments(zt)
        per = []
        for it in range(3):
            ctx.sweep(us[it])
            st = ctx.safe_stats()
            per.append((ctx.assignments(), ctx.counts(), ctx.log_marg(), st["windows"], st["unproven_walked"], st["budget_cuts"],
                        ctx.sweep_stats()["moves"]))
        out.append((per, ctx.proof_lookahead_stats(), ctx.proof_pass_stats()))
        ctx.close()
    (a, sa, pa), (b, sb, pb) = out
    assert pa["dense_batches"] > 0 and pb["dense_batches"] > 0 and pa["table_batches"] == 0
    assert sum(x[6] for x in a) > 50, "the case is meant to have movers"
    assert sa["stretches_from_the_ring"] == 0 and sa["chunks_requested"] == 0, sa
    assert sb["stretches_from_the_ring"] >= 5 and sb["labels_rescored"] > 0 and sb["chunks_requested"] >= 3, sb
    for it in range(3):
        bad = np.nonzero(a[it][0] != b[it][0])[0]
        assert bad.size == 0, "sweep %d: %d labels differ, first at i=%d" % (it, bad.size, bad[0])
        npt.assert_array_equal(a[it][1], b[it][1])
        assert abs(a[it][2] - b[it][2]) <= 1e-12 * abs(a[it][2])
        # (the look-ahead ends stretches at chunk boundaries: more windows of the same walk)
        assert a[it][6] == b[it][6]


def test_dense_and_table_proof_passes_give_the_same_chain():
    """Safe-stay windows prove which visits stay in two ways: through the per-home bound tables + the pruning kernel's exact
    forms, or DENSELY (every (visit, label) pair of a stretch through the likelihood kernel, bgmm_get_proof_pass_stats).
    On overlapping clusters the chain picks the dense pass by itself after its first batch; pinned to either kind
    (bgmm_set_proof_pass) the labels, the log marginal, the windows and the visits left on the
    resolver's chain are the same -- which kind runs changes the cost, never the chain."""
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    N, D, K = 60000, 64, 40
    X, zt = gendata.synth_mixture(N, D, K, seed=141, mu_scale=0.5)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    us = np.random.RandomState(8).random_sample((3, N))
    out = {}
    for pin in ("0", "1", None):
        ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
        ctx.set_proof_pass(-1 if pin is None else int(pin))
        ctx.set_assignments(zt)
        zs, ss = [], []
        for it in range(3):
            ctx.sweep(us[it])
            zs.append(ctx.assignments())
            st = ctx.safe_stats()
            ss.append((st["windows"], st["unproven_walked"], st["budget_cuts"], ctx.sweep_stats()["moves"]))
        out[pin] = (zs, ss, ctx.log_marg(), ctx.proof_pass_stats())
        ctx.close()
    assert sum(s[0] for s in out["0"][1]) > 0, "the case is meant to run safe-stay windows"
    assert sum(s[3] for s in out["0"][1]) > 50, "the case is meant to have movers"
    assert out["0"][3]["dense_batches"] == 0 and out["0"][3]["table_batches"] > 0
    assert out["1"][3]["table_batches"] == 0 and out["1"][3]["dense_batches"] > 0
    assert out[None][3]["dense_batches"] > 0, "overlapping clusters: the chain is meant to go over to the dense proof pass"
    for pin in ("1", None):
        for it in range(3):
            npt.assert_array_equal(out[pin][0][it], out["0"][0][it], err_msg="pin %s sweep %d" % (pin, it))
        assert out[pin][2] == out["0"][2]
    # (the proofs of the two kinds are not the same proofs -- the dense one bounds every label from its exact form -- so
    #  the visits left to the resolver may differ; the moves may not)
    assert [s[3] for s in out["1"][1]] == [s[3] for s in out["0"][1]]
    # the kind may change on ONE chain (the host looks at the tables again every eighth sweep); a dense pass
    # builds no tables and must not claim them for the epoch it ran at.  dense / table / dense / table on one context,
    # chosen through the C-ABI, against the pinned runs; and a table pass right behind a dense one on a chain that has
    # stopped moving (nothing bumps the state epoch in between)
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
    ctx.set_assignments(zt)
    for it in range(3):
        ctx.set_proof_pass(1 if it % 2 == 0 else 0)
        ctx.sweep(us[it])
        npt.assert_array_equal(ctx.assignments(), out["0"][0][it], err_msg="alternating kinds, sweep %d" % it)
    st = ctx.proof_pass_stats()
    assert st["dense_batches"] > 0 and st["table_batches"] > 0
    assert ctx.log_marg() == out["0"][2]
    ctx.close()


def _case_benchmarked_at_scale(N, D, K, flip, tail):
    from pybgmm_amd.utils import gendata
    X, zt = gendata.synth_mixture(N, D, K, seed=31 + D)
    rs = np.random.RandomState(D + K)
    z0 = zt.copy()
    idx = rs.choice(N, size=flip, replace=False)
    z0[idx] = rs.randint(0, K, size=flip)
    n_sw = 2 if tail else 1
    us = rs.random_sample((n_sw, N))
    return {"X": X, "prior": gendata.demo_prior_params(D), "z0": z0, "K_max": 2 * K,
            "sweeps": [(us[it], None, None, N if it == 0 else tail) for it in range(n_sw)]}


_case_benchmarked_at_scale.cost = lambda N, D, K, flip, tail: (N + tail) * K * D * D


@pytest.mark.slow
@pytest.mark.parametrize("N,D,K,flip,tail", [(100000, 64, 200, 2000, 8000), (16000, 128, 40, 200, 0), (8000, 128, 200, 200, 0)],
                         ids=["D64-K200-2000-wrong-labels", "D128-K40-200-wrong-labels", "D128-K200-200-wrong-labels"])
@with_oracle(_case_benchmarked_at_scale)
def test_benchmarked_mode_against_c_oracle_at_scale(N, D, K, flip, tail, oracle_ref):
    """the largest problems the C port of the reference finishes in about a minute per sweep, at
    BASELINE's D and K, the truth with wrong labels sprinkled in (the sweep repairs them: movers one per ~50 visits, then
    a chain at rest).  The default configuration AND the mode bench.py times (prune_mode 3) against ONE oracle run --
    a whole sweep, then `tail` visits of a second one (the hand-back to the at-rest path) -- labels identical after
    each; a mismatch reports the CDF margin at the first diverging visit.  The oracle costs 0.4 - 1.2 ms per visit at
    D = 64, K = 200 (1.2 - 3.5 at D = 128, K = 40): about a minute per case on the GPU box's host.  VERDICT r4 #7(i): C5's
    D AND K together (D = 128, K = 200: 7.6 ms per oracle visit, N = 8000)."""
    from divergence import assert_same_labels, first_divergence
    from oracle import c_oracle
    from pybgmm_amd import _lib
    case, ref = oracle_ref
    X, z0 = case["X"], case["z0"]
    m_0, k_0, v_0, S_0 = case["prior"]
    n_sw = len(case["sweeps"])
    us = np.stack([sw[0] for sw in case["sweeps"]])
    orders, powers = [None] * n_sw, [None] * n_sw
    mk_oracle = lambda: c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, z0, 2 * K, scipy_tables=False)

    def mk_ctx(prune):
        c = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 2 * K)
        c.set_tuning(prune_mode=prune)
        c.set_assignments(z0)
        return c
    ctxs = {prune: mk_ctx(prune) for prune in (0, 3)}
    moved = 0
    for it in range(n_sw):
        n_vis = case["sweeps"][it][3]
        zo, lo = ref[it]["z"], ref[it]["log_marg"]
        for prune, ctx in ctxs.items():
            ctx.set_sweep_visits(n_vis)
            ctx.sweep(us[it])
            assert_same_labels(ctx.assignments(), zo, "prune_mode %d, sweep %d" % (prune, it),
                               lambda: first_divergence(lambda: mk_ctx(prune), mk_oracle, us, orders, powers, it))
            assert abs(ctx.log_marg() - lo) <= 1e-9 * abs(lo)
        moved += ctxs[3].sweep_stats()["moves"]
    assert moved >= flip // 2, "the case is meant to repair its wrong labels"
    assert ctxs[3].prune_stats()["certified_visits"] == 0
    for ctx in ctxs.values():
        ctx.close()


def test_crpmm_class_each_in_own_reproduces_reference_kat3():
    """The reference's third known-answer test through the CLASS (pybgmm/tests/test_igmm.py:106-146:
    ``CRPMM(X, prior, 1.0, None, assignments="each-in-own", K=3)``, one sweep, N = 20): the literal 2014 labels, from the
    caller's seeded global streams -- the init string handled by IGMM.__init__ (igmm/igmm.py:98-99), not a z vector."""
    import random
    from pybgmm_amd.igmm import CRPMM
    from pybgmm_amd.prior import NIW
    from pybgmm_amd.utils import gendata
    g = Golden("kat3_each_in_own")
    random.seed(1)
    np.random.seed(1)
    X, z_true = gendata.demo_mixture(20, 2, 4, rs=np.random)
    npt.assert_array_equal(X, g.X)
    mm = CRPMM(X, NIW(*gendata.demo_prior_params(2, v_0=5)), 1.0, None, assignments="each-in-own", K=3)
    npt.assert_array_equal(mm.components.assignments, np.arange(20))
    npt.assert_array_equal(mm.components.assignments, g.z_init)
    assert mm.components.K == 20
    record, _ = mm.collapsed_gibbs_sampler(1, z_true, num_saved=0)
    expected = np.array([5, 2, 4, 3, 2, 7, 2, 7, 1, 0, 4, 6, 4, 1, 6, 4, 1, 7, 1, 0])      # test_igmm.py:143
    npt.assert_array_equal(mm.components.assignments, expected)
    npt.assert_array_equal(mm.components.assignments, g.z[-1])
    npt.assert_allclose(record["log_marg"], g.log_marg, rtol=1e-9)
    assert record["components"] == list(g.K)


@pytest.mark.parametrize("case,mode,seed", [("each_in_own_50", "each-in-own", 4), ("one_by_one_50", "one-by-one", 5)])
def test_crpmm_class_init_strings_reproduce_reference(case, mode, seed):
    """``assignments="each-in-own"`` / ``"one-by-one"`` (igmm/igmm.py:95-99) through the class: the initial labels the
    reference builds (arange(N); all -1 but z[0] = 0), then three sweeps from the caller's seeded ``random`` stream --
    labels, K, counts and log marginal of every sweep as captured from the reference."""
    import random
    from pybgmm_amd.igmm import CRPMM
    from pybgmm_amd.prior import NIW
    g = Golden(case)
    assert int(g.d["seed_random"]) == seed
    random.seed(seed)
    np.random.seed(seed)
    mm = CRPMM(g.X, NIW(*g.prior), g.alpha, None, assignments=mode, K=1)
    npt.assert_array_equal(mm.components.assignments, g.z_init)
    zs, lms, Ks = [], [], []
    for it in range(g.n_iter):                           # (sweep by sweep: the labels after EVERY sweep are pinned)
        record, _ = mm.collapsed_gibbs_sampler(1, g.d["true_assignments"], num_saved=0)
        zs.append(mm.components.assignments.copy())
        lms.append(record["log_marg"][0])
        Ks.append(record["components"][0])
        npt.assert_array_equal(mm.components.counts[:Ks[-1]], g.counts_at(it))
    npt.assert_array_equal(np.array(zs), g.z)
    npt.assert_allclose(lms, g.log_marg, rtol=1e-9)
    assert Ks == list(g.K)


def _case_wrong_labels(N, D, K, flip, seed, rs_seed, n_sweeps, pcrp):
    """The truth with `flip` wrong labels; n_sweeps sweeps, pCRP ones with a fresh permutation each and powered weights from
    the second on (igmm/pcrpmm.py:86-131)."""
    from pybgmm_amd.utils import gendata
    X, zt = gendata.synth_mixture(N, D, K, seed=seed)
    rs = np.random.RandomState(rs_seed)
    z0 = zt.copy()
    idx = rs.choice(N, size=flip, replace=False)
    z0[idx] = rs.randint(0, K, size=flip)
    us = rs.random_sample((n_sweeps, N))
    orders = [rs.permutation(N).astype(np.int64) if pcrp else None for _ in range(n_sweeps)]
    powers = [1.01 if (pcrp and it > 0) else None for it in range(n_sweeps)]
    return {"X": X, "prior": gendata.demo_prior_params(D), "z0": z0, "K_max": 2 * K,
            "sweeps": [(us[it], orders[it], powers[it]) for it in range(n_sweeps)], "flip": flip}


def _case_c4_full():
    return _case_wrong_labels(1000000, 64, 200, 2000, seed=1, rs_seed=64, n_sweeps=1, pcrp=False)


_case_c4_full.cost = lambda: 1000000 * 200 * 64 * 64


def _case_c5_part():
    case = _case_wrong_labels(200000, 128, 200, 800, seed=2, rs_seed=128, n_sweeps=2, pcrp=True)
    u, order, power = case["sweeps"][1]
    case["sweeps"][1] = (u, order, power, 40000)          # (the second sweep -- powered weights -- for its first 40 000 visits)
    return case


_case_c5_part.cost = lambda: 240000 * 200 * 128 * 128


def _case_c3_full():
    return _case_wrong_labels(1000000, 16, 100, 2000, seed=1, rs_seed=16, n_sweeps=2, pcrp=True)


_case_c3_full.cost = lambda: 2 * 1000000 * 100 * 16 * 16


def _wrong_labels_against_the_oracle(case, ref):
    from divergence import assert_same_labels
    from pybgmm_amd import _lib
    m_0, k_0, v_0, S_0 = case["prior"]
    for prune in (0, 3):
        ctx = _lib.Context(case["X"], m_0, k_0, v_0, S_0, 1.0, case["K_max"])
        ctx.set_tuning(prune_mode=prune)
        ctx.set_assignments(case["z0"])
        moves = 0
        for it, sw in enumerate(case["sweeps"]):
            u, order, power, n_vis = (tuple(sw) + (None,))[:4]
            if n_vis is not None:
                ctx.set_sweep_visits(n_vis)
            ctx.sweep(u, order, power)
            assert_same_labels(ctx.assignments(), ref[it]["z"], "prune_mode %d, sweep %d" % (prune, it))
            assert abs(ctx.log_marg() - ref[it]["log_marg"]) <= 1e-9 * abs(ref[it]["log_marg"])
            moves += ctx.sweep_stats()["moves"]
        assert moves >= case["flip"] // 2
        ctx.close()


@pytest.mark.slow
@with_oracle(_case_c4_full)
def test_c4_full_size_against_c_oracle(oracle_ref):
    """BASELINE's C4 at FULL size (N = 1e6, D = 64, K = 200) against the C port of the reference: the truth
    with 2 000 wrong labels, one whole sweep in the default configuration and in the benchmarked mode, every label and the
    log marginal.  The oracle needs ~3 minutes of one host core for it (0.17 ms per visit since its loops walk the 6.5 MB of
    inverse covariances row-wise); since round 6 the heavy cases' oracles all run side by side from the start of the
    session, each on a core of its own (tests/oracle_pool.py), and this one is simply among the last to finish.  Rounds
    3 - 5 stopped at N = 1.2e5 - 2.5e5."""
    _wrong_labels_against_the_oracle(*oracle_ref)


@pytest.mark.slow
@with_oracle(_case_c3_full)
def test_c3_full_size_against_c_oracle(oracle_ref):
    """BASELINE's C3 at FULL size against the C port of the reference: PCRPMM semantics, N = 1e6, D = 16,
    K = 100, the truth with 2 000 wrong labels, two sweeps (a fresh permutation each, powered weights in the second:
    pcrpmm.py:86-112).  The default configuration AND the mode bench.py times against ONE oracle run: labels identical
    after each sweep, log marginal to 1e-9.  The oracle costs ~12 us per visit at this shape: ~25 s of host time."""
    _wrong_labels_against_the_oracle(*oracle_ref)


@pytest.mark.slow
@with_oracle(_case_c5_part)
def test_c5_shape_against_c_oracle(oracle_ref):
    """BASELINE's C5 shape (PCRPMM, D = 128, K = 200, full covariance) at N = 2e5 against the C port of the reference
    (rounds 4 - 5: N = 8 000 - 16 000): the truth with 800 wrong labels, a whole pCRP sweep and the first
    40 000 visits of a second one -- a fresh permutation each, powered seating weights in the second
    (igmm/pcrpmm.py:86-131) -- in the default configuration and in the benchmarked mode.  0.78 ms per oracle visit:
    ~3 minutes of a host core, side by side with the others."""
    _wrong_labels_against_the_oracle(*oracle_ref)


def test_first_divergence_diagnostic_finds_a_planted_divergence():
    """VERDICT r2 #8(iii) / SURVEY 7.3.2: the diagnostic itself.  The device is handed a uniform stream that differs from
    the oracle's at ONE visit (moved across a CDF boundary of that visit), so the chains part exactly there; the
    diagnostic must name the visit, rebuild both CDFs to rounding level and call it "not a tie"."""
    from divergence import _cdf, first_divergence
    from oracle import c_oracle
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    N, D, K = 6000, 16, 8
    X, zt = gendata.synth_mixture(N, D, K, seed=5, mu_scale=0.8)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    rs = np.random.RandomState(0)
    us = rs.random_sample((2, N))
    mk_oracle = lambda: c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, zt, 4 * K, scipy_tables=False)

    def mk_ctx():
        c = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
        c.set_assignments(zt)
        return c
    # a visit of sweep 1 whose draw can be flipped: run the oracle up to it, look at its CDF
    p = 3210
    o = mk_oracle()
    o.sweep(us[0])
    o.sweep(us[1], n_visits=p)
    prob, cdf = _cdf(o.probe_visit(p))
    j = int(np.argmax(prob))                          # the likeliest label and a uniform that picks another one
    us_dev = us.copy()
    lo_edge = cdf[j] - prob[j]
    inside = lo_edge + 0.5 * prob[j]
    outside = cdf[j] + 0.5 * (1.0 - cdf[j]) if cdf[j] < 1.0 - 1e-9 else 0.5 * lo_edge
    us[1][p], us_dev[1][p] = inside, outside
    rep = first_divergence(mk_ctx, mk_oracle, us, [None, None], [None, None], 1, us_dev=us_dev)
    assert rep["visit"] == p and rep["point"] == p, rep["text"]
    assert rep["oracle_draw"] == j and rep["device_draw"] != j, rep["text"]
    assert rep["max_abs_dlogp"] < 1e-9 and rep["max_abs_dcdf"] < 1e-11, rep["text"]     # (both sides' scores agree)
    assert rep["verdict"].startswith("NOT a tie"), rep["text"]
    # and with equal streams there is nothing to report
    rep = first_divergence(mk_ctx, mk_oracle, us, [None, None], [None, None], 1)
    assert rep["visit"] is None


def test_full_size_safe_stay_windows_equal_the_other_mover_paths():
    """C4's shape with 2 000 wrong labels: the default schedule (safe-stay windows) and the schedule without them
    (per-mover kernels / plain frozen-factor windows) walk the same chain."""
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    N, D, K = 1000000, 64, 200
    X, zt = gendata.synth_mixture(N, D, K, seed=11)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    rs = np.random.RandomState(3)
    z0 = zt.copy()
    idx = rs.choice(N, size=2000, replace=False)
    z0[idx] = rs.randint(0, K, size=2000)
    us = rs.random_sample((2, N))
    out = []
    for mode in (0, 5):
        ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
        ctx.set_tuning(resolver_mode=mode)
        ctx.set_assignments(z0)
        for it in range(2):
            ctx.sweep(us[it])
            if it == 0 and mode == 0:
                assert ctx.safe_stats()["windows"] > 0
        out.append((ctx.assignments(), ctx.counts(), ctx.log_marg()))
        ctx.close()
    npt.assert_array_equal(out[0][0], out[1][0])
    npt.assert_array_equal(out[0][1], out[1][1])
    assert abs(out[0][2] - out[1][2]) <= 1e-9 * abs(out[0][2])


# ---- covariance_type="diag"  -------------------------------------------
@pytest.mark.parametrize("window", [0, 64])
@pytest.mark.parametrize("case", DIAG_CASES)
def test_diag_golden_trajectory(case, window):
    g = Golden(case)
    ctx = make_ctx(g, window=window)
    npt.assert_allclose(ctx.log_prior()[:4096], g.d["cached_log_prior"], rtol=1e-11, atol=1e-11)
    for it in range(g.n_iter):
        ctx.sweep(g.u[it], g.sweep_order(it), g.sweep_power(it))
        z = ctx.assignments()
        bad = np.nonzero(z != g.z[it])[0]
        assert bad.size == 0, "sweep %d: %d labels differ, first at i=%d" % (it, bad.size, bad[0])
        npt.assert_array_equal(ctx.counts(), g.counts_at(it))
        assert abs(ctx.log_marg() - g.log_marg[it]) <= 1e-9 * abs(g.log_marg[it])
    m, S, lpv, iv = ctx.stats()
    npt.assert_array_equal(m, g.d["final_m"])
    npt.assert_array_equal(S, g.d["final_S"])
    npt.assert_allclose(lpv, g.d["final_logdet"], rtol=1e-10, atol=1e-10)
    npt.assert_allclose(iv, g.d["final_inv"], rtol=1e-10)
    ctx.close()


def test_diag_class_reproduces_reference():
    import random
    from pybgmm_amd.igmm import CRPMM
    from pybgmm_amd.prior import NIW
    from pybgmm_amd.utils import gendata
    g = Golden("diag_kat_2d")
    random.seed(1)
    np.random.seed(1)
    X, z_true = gendata.demo_mixture(100, 2, 4, rs=np.random)
    mm = CRPMM(X, NIW(*g.prior), 1.0, None, assignments="rand", K=3, covariance_type="diag")
    record, _ = mm.collapsed_gibbs_sampler(10, z_true, num_saved=0)
    npt.assert_array_equal(mm.components.assignments, g.z[-1])
    npt.assert_allclose(record["log_marg"], g.log_marg, rtol=1e-9)
    K = mm.components.K
    npt.assert_allclose(mm.components.log_prod_vars[:K], g.d["final_logdet"], rtol=1e-10)
    npt.assert_allclose(mm.components.inv_vars[:K], g.d["final_inv"], rtol=1e-10)
    assert mm.components.S_N_partials.shape == (mm.components.K_max, 2)


def test_diag_components_match_univariate_student_t():
    """The analytic checks of pybgmm/tests/test_gaussian_components_diag.py (prior predictive,
    posterior predictive, del_item, log_marg_k) against scipy's formulas, on the device."""
    from scipy.special import gammaln
    from pybgmm_amd.gaussian import GaussianComponentsDiag
    from pybgmm_amd.prior import NIW

    def t_logpdf(x, mu, var, v):
        c = gammaln((v + 1) / 2.) - gammaln(v / 2.) - 0.5 * (np.log(v) + np.log(np.pi) + np.log(var))
        return c - (v + 1) / 2. * np.log(1 + 1. / v * (x - mu) ** 2 / var)

    rs = np.random.RandomState(1)
    N, D = 10, 3
    X = 5 * rs.rand(N, D) - 1
    m_0, k_0, v_0, S_0 = 5 * rs.rand(D) - 2, float(rs.randint(15) + 1), 4, 2 * rs.rand(D) + 3
    gmm = GaussianComponentsDiag(X, NIW(m_0, k_0, v_0, S_0), np.zeros(N, dtype=int))
    var0 = S_0 * (k_0 + 1) / (k_0 * v_0)
    npt.assert_almost_equal(gmm.log_prior(0), np.sum(t_logpdf(X[0], m_0, var0, v_0)))

    def posterior(Xs):
        n = len(Xs)
        k_N, v_N = k_0 + n, v_0 + n
        m_N = (k_0 * m_0 + Xs.sum(axis=0)) / k_N
        S_N = S_0 + k_0 * m_0 ** 2 + (Xs ** 2).sum(axis=0) - k_N * m_N ** 2
        return k_N, v_N, m_N, S_N

    k_N, v_N, m_N, S_N = posterior(X)
    npt.assert_almost_equal(gmm.log_post_pred_k(0, 0),
                            np.sum(t_logpdf(X[0], m_N, S_N * (k_N + 1) / (k_N * v_N), v_N)))
    lm = (-N * D / 2. * np.log(np.pi) + D / 2. * np.log(k_0) - D / 2. * np.log(k_N)
          + v_0 / 2. * np.log(S_0).sum() - v_N / 2. * np.log(S_N).sum()
          + D * (gammaln(v_N / 2.) - gammaln(v_0 / 2.)))
    npt.assert_almost_equal(gmm.log_marg_k(0), lm)
    gmm.del_item(N - 1)
    k_N, v_N, m_N, S_N = posterior(X[:-1])
    npt.assert_almost_equal(gmm.log_post_pred_k(0, 0),
                            np.sum(t_logpdf(X[0], m_N, S_N * (k_N + 1) / (k_N * v_N), v_N)))


@pytest.mark.parametrize("N,D,K", [(5000, 16, 30), (3000, 64, 12), (2000, 128, 8), (1500, 300, 6)])
def test_diag_against_c_oracle(N, D, K):
    from oracle import c_oracle
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    X, _ = gendata.synth_mixture(N, D, K, seed=300 + D)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    S_0 = np.ascontiguousarray(np.diag(S_0))
    rs = np.random.RandomState(D)
    z0 = np.unique(rs.randint(0, K, N), return_inverse=True)[1]
    us = rs.random_sample((2, N))
    order = rs.permutation(N)
    o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, z0, 4 * K, cov_type="diag")
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K, tables=reference_tables(v_0, N), cov_type="diag")
    ctx.set_assignments(z0)
    for it in range(2):
        power = 1.05 if it == 1 else None
        o.sweep(us[it], order, power)
        ctx.sweep(us[it], order, power)
        npt.assert_array_equal(ctx.assignments(), o.z)
        lo = o.log_marg()
        assert abs(ctx.log_marg() - lo) <= 1e-9 * abs(lo)
    ctx.close()


# ---- covariance_type="fixed"  ------------------------------------------
@pytest.mark.parametrize("window", [0, 64])
@pytest.mark.parametrize("case", FIXED_CASES)
def test_fixed_golden_trajectory(case, window):
    g = Golden(case)
    ctx = make_ctx(g, window=window, tables=False)
    npt.assert_allclose(ctx.log_prior()[:4096], g.d["cached_log_prior"], rtol=1e-12, atol=1e-12)
    for it in range(g.n_iter):
        ctx.sweep(g.u[it], g.sweep_order(it), g.sweep_power(it))
        z = ctx.assignments()
        bad = np.nonzero(z != g.z[it])[0]
        assert bad.size == 0, "sweep %d: %d labels differ, first at i=%d" % (it, bad.size, bad[0])
        npt.assert_array_equal(ctx.counts(), g.counts_at(it))
        assert abs(ctx.log_marg() - g.log_marg[it]) <= 1e-9 * abs(g.log_marg[it])
    m, pN, lpp, pp = ctx.stats()
    npt.assert_array_equal(m, g.d["final_m"])
    npt.assert_array_equal(pN, g.d["final_S"])
    npt.assert_allclose(lpp, g.d["final_logdet"], rtol=1e-11, atol=1e-11)
    npt.assert_allclose(pp, g.d["final_inv"], rtol=1e-12)
    ctx.close()


def test_fixed_class_reproduces_reference():
    import random
    from pybgmm_amd.gaussian import FixedVarPrior
    from pybgmm_amd.igmm import CRPMM
    g = Golden("fixed_2d")
    random.seed(12)
    np.random.seed(12)
    D = g.D
    prior = FixedVarPrior(g.S_0[:D], g.m_0, g.S_0[D:])
    mm = CRPMM(g.X, prior, g.alpha, None, assignments="rand", K=6, K_max=g.K_max, covariance_type="fixed")
    npt.assert_array_equal(mm.components.assignments, g.z_init)
    record, _ = mm.collapsed_gibbs_sampler(g.n_iter, g.d["true_assignments"], num_saved=0)
    npt.assert_array_equal(mm.components.assignments, g.z[-1])
    npt.assert_allclose(record["log_marg"], g.log_marg, rtol=1e-9)
    npt.assert_allclose(record["nmi"], g.d["rec_nmi"], rtol=1e-11)
    assert [int(v) for v in record["loss"]] == [int(v) for v in g.d["rec_loss"]]
    K = mm.components.K
    npt.assert_allclose(mm.components.precision_preds[:K], g.d["final_inv"], rtol=1e-12)
    npt.assert_array_equal(mm.components.mu_N_numerators[:K], g.d["final_m"])


def test_fixed_components_match_univariate_normals():
    """The analytic checks of pybgmm/tests/test_gaussian_components_fixedvar.py (prior predictive,
    posterior predictive per component, del_item, log_marg_k) against closed-form normals."""
    from pybgmm_amd.gaussian import FixedVarPrior, GaussianComponentsFixedVar

    def n_logpdf(x, mean, var):
        return -0.5 * np.log(2 * np.pi * var) - (x - mean) ** 2 / (2 * var)

    rs = np.random.RandomState(1)
    D, sizes = 10, (10, 5, 5)
    N = sum(sizes)
    X = 5 * rs.rand(N, D) - 1
    var, mu_0, var_0 = rs.rand(D) + 0.1, 5 * rs.rand(D) - 2, 2 * rs.rand(D) + 0.1
    z = np.repeat(np.arange(3), sizes)
    gmm = GaussianComponentsFixedVar(X, FixedVarPrior(var, mu_0, var_0), z)
    npt.assert_almost_equal(gmm.log_prior(0), np.sum(n_logpdf(X[0], mu_0, var_0)))

    def predictive(Xs, x):
        n = len(Xs)
        var_N = 1. / (1. / var_0 + n / var)
        mu_N = var_N * (mu_0 / var_0 + Xs.sum(axis=0) / var)
        return np.sum(n_logpdf(x, mu_N, var_N + var))

    for k in range(3):
        npt.assert_almost_equal(gmm.log_post_pred_k(3, k), predictive(X[z == k], X[3]))
    npt.assert_allclose(gmm.log_post_pred(7), [predictive(X[z == k], X[7]) for k in range(3)], rtol=1e-12)
    # marginal of a component = chain rule over its points
    for k in range(3):
        Xs = X[z == k]
        chain = sum(predictive(Xs[:j], Xs[j]) for j in range(len(Xs)))
        npt.assert_almost_equal(gmm.log_marg_k(k), chain)
    gmm.del_item(N - 1)
    npt.assert_almost_equal(gmm.log_post_pred_k(0, 2), predictive(X[z == 2][:-1], X[0]))
    gmm.add_item(N - 1, 0)
    npt.assert_almost_equal(gmm.log_post_pred_k(0, 0), predictive(np.vstack([X[z == 0], X[N - 1:]]), X[0]))


@pytest.mark.parametrize("N,D,K", [(5000, 16, 30), (3000, 64, 12)])
def test_fixed_against_c_oracle(N, D, K):
    from oracle import c_oracle
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    X, _ = gendata.synth_mixture(N, D, K, seed=500 + D)
    rs = np.random.RandomState(D)
    m_0 = np.zeros(D)
    S_0 = np.concatenate([0.5 + rs.rand(D), 20.0 + rs.rand(D)])      # [var ; var_0]
    z0 = np.unique(rs.randint(0, K, N), return_inverse=True)[1]
    us = rs.random_sample((2, N))
    order = rs.permutation(N)
    o = c_oracle.COracle(X, m_0, 1.0, 1, S_0, 1.0, z0, 4 * K, cov_type="fixed")
    ctx = _lib.Context(X, m_0, 1.0, 1, S_0, 1.0, 4 * K, cov_type="fixed")
    ctx.set_assignments(z0)
    for it in range(2):
        power = 1.05 if it == 1 else None
        o.sweep(us[it], order, power)
        ctx.sweep(us[it], order, power)
        npt.assert_array_equal(ctx.assignments(), o.z)
        lo = o.log_marg()
        assert abs(ctx.log_marg() - lo) <= 1e-9 * abs(lo)
    ctx.close()


# ---- pruned windows for diagonal / fixed-variance components --------------------------------
@pytest.mark.parametrize("prune", [1, 2], ids=["unpruned", "every-window-pruned"])
@pytest.mark.parametrize("case", DIAG_CASES + FIXED_CASES)
def test_diag_fixed_pruning_does_not_change_trajectory(case, prune):
    g = Golden(case)
    ctx = make_ctx(g, prune=prune, tables=g.cov_type != "fixed")
    for it in range(g.n_iter):
        ctx.sweep(g.u[it], g.sweep_order(it), g.sweep_power(it))
        z = ctx.assignments()
        bad = np.nonzero(z != g.z[it])[0]
        assert bad.size == 0, "sweep %d: %d labels differ, first at i=%d" % (it, bad.size, bad[0])
        assert abs(ctx.log_marg() - g.log_marg[it]) <= 1e-9 * abs(g.log_marg[it])
        assert (ctx.sweep_stats()["bound_blocks"] + ctx.prune_stats()["certified_visits"] > 0) == (prune == 2)
    ctx.close()


@pytest.mark.parametrize("cov", ["diag", "fixed"])
@pytest.mark.parametrize("N,D,K,sep", [(20000, 16, 40, 4.0), (12000, 24, 300, 1.2), (6000, 8, 900, 4.0), (4000, 200, 12, 1.0)])
def test_diag_fixed_pruned_against_c_oracle(cov, N, D, K, sep):
    """Steady-state start (labels at the truth, a few unassigned): the pruned-window kernels run by
    the default policy and, second context, in every window."""
    from oracle import c_oracle
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    X, zt = gendata.synth_mixture(N, D, K, seed=700 + D, mu_scale=sep)
    rs = np.random.RandomState(K + D)
    if cov == "diag":
        m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
        S_0 = np.ascontiguousarray(np.diag(S_0))
    else:
        m_0, k_0, v_0 = np.zeros(D), 1.0, 1
        S_0 = np.concatenate([0.4 + 0.2 * rs.rand(D), 15.0 + rs.rand(D)])      # [var ; var_0]
    z0 = zt.copy()
    z0[rs.rand(N) < 0.01] = -1
    us = rs.random_sample((2, N))
    order = rs.permutation(N)
    K_max = min(N, 2 * K + 64)
    o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, z0, K_max, cov_type=cov)
    ref = []
    for it in range(2):
        o.sweep(us[it], order if it == 1 else None, 1.03 if it == 1 else None)
        ref.append((o.z.copy(), o.log_marg()))
    for prune in (0, 2):
        ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, K_max, cov_type=cov,
                           tables=reference_tables(v_0, N) if cov == "diag" else None)
        ctx.set_tuning(prune_mode=prune)
        ctx.set_assignments(z0)
        for it in range(2):
            ctx.sweep(us[it], order if it == 1 else None, 1.03 if it == 1 else None)
            z = ctx.assignments()
            bad = np.nonzero(z != ref[it][0])[0]
            assert bad.size == 0, "prune %d sweep %d: %d labels differ, first at i=%d" % (prune, it, bad.size, bad[0])
            assert abs(ctx.log_marg() - ref[it][1]) <= 1e-9 * abs(ref[it][1])
        ctx.close()


def test_uniform_exactly_zero_disables_pruning():
    """u == 0.0 is the one uniform for which a pruned (probability 0 instead of < 2e-35) label is
    visible to the reference's `u -= p` scan: it returns the FIRST label.  Such sweeps run unpruned."""
    from oracle import c_oracle
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    N, D, K = 4000, 16, 12
    X, zt = gendata.synth_mixture(N, D, K, seed=77)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    u = np.random.RandomState(5).random_sample(N)
    u[[17, 2500]] = 0.0
    o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, zt, 4 * K)
    o.sweep(u)
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K, tables=reference_tables(v_0, N))
    ctx.set_tuning(kernel_kind=2)
    ctx.set_assignments(zt)
    ctx.sweep(u)
    npt.assert_array_equal(ctx.assignments(), o.z)
    assert ctx.assignments()[17] == 0 and ctx.sweep_stats()["bound_blocks"] == 0
    ctx.close()


@pytest.mark.parametrize("N,D,K,sep", [(30000, 32, 30, 4.0), (30000, 16, 40, 1.6), (20000, 64, 12, 2.2)])
def test_certified_stays_against_c_oracle(N, D, K, sep):
    """Several sweeps from the true labelling: from the second sweep on most visits are decided by
    certify_kernel from the per-point cache (well separated data) or fall back to the pruning
    kernel (overlapping data, after moves) -- the chain must not notice."""
    from oracle import c_oracle
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    X, zt = gendata.synth_mixture(N, D, K, seed=31 + D, mu_scale=sep)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    rs = np.random.RandomState(D)
    us = rs.random_sample((5, N))
    o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, zt, 4 * K)
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K, tables=reference_tables(v_0, N))
    ctx.set_tuning(kernel_kind=2)
    ctx.set_assignments(zt)
    certified = 0
    for it in range(5):
        o.sweep(us[it])
        ctx.sweep(us[it])
        z = ctx.assignments()
        bad = np.nonzero(z != o.z)[0]
        assert bad.size == 0, "sweep %d: %d labels differ, first at i=%d" % (it, bad.size, bad[0])
        lo = o.log_marg()
        assert abs(ctx.log_marg() - lo) <= 1e-9 * abs(lo)
        certified += ctx.prune_stats()["certified_visits"]
    if sep >= 4.0:
        assert certified > 3 * N          # sweeps 2..5 almost entirely certified
    ctx.close()


def test_lean_steps_with_a_fresh_permutation_every_sweep():
    """pCRP order: a new permutation per sweep.  A lean step examines the points in storage order (the
    window is the whole sweep and only the count of certified visits matters); a changed labelling makes
    it fail and fall back.  Same chain as the plain evaluation, and as the C port of the reference."""
    from oracle import c_oracle
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    N, D, K = 20000, 32, 10
    X, zt = gendata.synth_mixture(N, D, K, seed=5)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    rs = np.random.RandomState(8)
    o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, zt, 4 * K)
    ctxs = []
    for prune in (0, 1):
        c = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K, tables=reference_tables(v_0, N))
        c.set_tuning(kernel_kind=2, prune_mode=prune)
        c.set_assignments(zt)
        ctxs.append(c)
    lean_sweeps = 0
    for it in range(8):
        u = rs.random_sample(N)
        order = rs.permutation(N)
        power = 1.02 if it >= 1 else None
        if it == 4:                          # restart all three from a labelling with 30 wrong points
            z1 = np.array(o.z, dtype=np.int64)
            assert z1.max() == K - 1
            z1[500:530] = (z1[500:530] + 1) % K
            o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, z1, 4 * K)
            for c in ctxs:
                c.set_assignments(z1)
        o.sweep(u, order, power)
        for c in ctxs:
            c.sweep(u, order, power)
        npt.assert_array_equal(ctxs[0].assignments(), o.z)
        npt.assert_array_equal(ctxs[1].assignments(), o.z)
        st, ps = ctxs[0].sweep_stats(), ctxs[0].prune_stats()
        if ps["certified_visits"] == N and st["moves"] == 0:
            lean_sweeps += 1
    assert lean_sweeps >= 3                   # (the steady sweeps before and after the disturbance)
    lo = o.log_marg()
    assert abs(ctxs[0].log_marg() - lo) <= 1e-9 * abs(lo)
    for c in ctxs:
        c.close()


def test_lean_steps_fall_back_when_certification_fails():
    """After a sweep that certified every visit only certify_kernel is queued per step; when the state
    was changed behind its back (add_item / del_item) the step is refused on the device and re-queued
    with the full kernels."""
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    N, D, K = 20000, 32, 10
    X, zt = gendata.synth_mixture(N, D, K, seed=5)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    us = np.random.RandomState(3).random_sample((5, N))
    out = []
    for prune in (0, 1, 30):                 # 30: certified stays for three sweeps, then switched off
        ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K, tables=reference_tables(v_0, N))
        ctx.set_tuning(kernel_kind=2, prune_mode=prune % 10)
        ctx.set_assignments(zt)
        traj = []
        for it in range(5):
            if it == 3 and prune == 30:
                ctx.set_tuning(kernel_kind=2, prune_mode=3)
            if it == 3:                      # move 40 points to a wrong component by hand
                for i in range(100, 140):
                    ctx.del_item(i)
                    ctx.add_item(i, (int(zt[i]) + 1) % K)
            ctx.sweep(us[it])
            traj.append(ctx.assignments().copy())
            if prune == 0 and it == 2:
                assert ctx.prune_stats()["certified_visits"] == N
        out.append((traj, ctx.log_marg()))
        ctx.close()
    for a, b, c3 in zip(out[0][0], out[1][0], out[2][0]):
        npt.assert_array_equal(a, b)
        npt.assert_array_equal(c3, b)
    assert (out[0][0][3] != out[0][0][2]).sum() == 0      # the hand-moved points went back
    assert abs(out[0][1] - out[1][1]) <= 1e-9 * abs(out[1][1])


@pytest.mark.parametrize("cov", ["diag", "fixed"])
def test_certified_stays_diag_fixed_against_c_oracle(cov):
    from oracle import c_oracle
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    N, D, K = 30000, 24, 30
    X, zt = gendata.synth_mixture(N, D, K, seed=88)
    rs = np.random.RandomState(4)
    if cov == "diag":
        m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
        S_0 = np.ascontiguousarray(np.diag(S_0))
    else:
        m_0, k_0, v_0 = np.zeros(D), 1.0, 1
        S_0 = np.concatenate([np.full(D, 0.49), np.full(D, 16.0)])
    z0 = zt.copy()
    flip = rs.choice(N, size=60, replace=False)
    z0[flip] = rs.randint(0, K, size=flip.size)          # sweep 1 has moves to make; 2.. run on certificates
    us = rs.random_sample((5, N))
    o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, z0, 4 * K, cov_type=cov)
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K, cov_type=cov,
                       tables=reference_tables(v_0, N) if cov == "diag" else None)
    ctx.set_assignments(z0)
    certified = 0
    for it in range(5):
        o.sweep(us[it])
        ctx.sweep(us[it])
        z = ctx.assignments()
        bad = np.nonzero(z != o.z)[0]
        assert bad.size == 0, "sweep %d: %d labels differ, first at i=%d" % (it, bad.size, bad[0])
        lo = o.log_marg()
        assert abs(ctx.log_marg() - lo) <= 1e-9 * abs(lo)
        certified += ctx.prune_stats()["certified_visits"]
    assert certified > 2 * N
    ctx.close()


@pytest.mark.parametrize("seed,burn", [(0, 0), (12345, 1), (7, 311), (99, 623), (2014, 1247)])
def test_device_mt19937_continues_the_callers_stream(seed, burn):
    """bgmm_stage_mt19937: the device produces exactly the doubles random.random() would, from any
    position inside a block, and leaves the generator where N host calls would."""
    import random
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata, rng as _rng
    N, D, K = 5000, 2, 3
    X, zt = gendata.synth_mixture(N, D, K, seed=1)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
    ctx.set_assignments(zt)
    host, dev = random.Random(seed), random.Random(seed)
    for _ in range(burn):                       # (an odd count leaves the position odd: a double then straddles)
        host.getrandbits(32); dev.getrandbits(32)
    for it in range(3):
        expect = np.array([host.random() for _ in range(N)])
        assert _rng.stage_uniforms_on_device(ctx, None, dev)
        npt.assert_array_equal(ctx.staged_uniforms(), expect)
        assert dev.getstate() == host.getstate()
        ctx.sweep_staged(None)
    ctx.close()


@pytest.mark.parametrize("N,burn", [(70000, 0), (200000, 311), (1000003, 1247), (2600001, 5)])
def test_device_mt19937_jump_ahead_across_chains(N, burn):
    """Requests longer than one chain of bgmm_mt19937_chain_blocks() = 256 blocks (159 744 words) run as chains side by side
    from jumped-ahead states (kernels_rng.hip): 1, 3, 13 and 33 chains here, from an even and an odd position, against
    random.random() -- and against the same chains run one after the other (bgmm_set_mt_jump(0))."""
    import random
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata, rng as _rng
    X, zt = gendata.synth_mixture(N, 2, 3, seed=1)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(2)
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 12)
    ctx.set_assignments(zt)
    for jump in (1, 0):
        ctx.set_mt_jump(jump)
        host, dev = random.Random(N), random.Random(N)
        for _ in range(burn):
            host.getrandbits(32); dev.getrandbits(32)
        for it in range(2):
            rs = np.random.RandomState()
            version, key, gauss = host.getstate()
            rs.set_state(("MT19937", np.asarray(key[:-1], dtype=np.uint32), int(key[-1])))
            expect = rs.random_sample(N)                       # (the same stream at C speed: utils/rng.py)
            _, new_key, pos = rs.get_state()[:3]
            host.setstate((version, tuple(int(v) for v in new_key) + (int(pos),), gauss))
            assert _rng.stage_uniforms_on_device(ctx, None, dev)
            npt.assert_array_equal(ctx.staged_uniforms(), expect)
            assert dev.getstate() == host.getstate()
    assert host.random() == dev.random()
    ctx.close()


@pytest.mark.parametrize("N", [4096, 4097, 65536, 100003, 1000000, 2000000])
def test_device_permutation_equals_numpy(N):
    """bgmm_stage_permutation_mt19937: np.random.permutation(N) of the pCRP sweep (pcrpmm.py:89) drawn on the device from a
    legacy numpy state -- the same permutation, the same generator state afterwards; three in a row from a position
    inside a block, then through the host helper with the GLOBAL np.random stream."""
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata, rng as _rng
    X, zt = gendata.synth_mixture(N, 2, 3, seed=1)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(2)
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 12)
    ctx.set_assignments(zt)
    host = np.random.RandomState(N % 1000)
    host.random_sample(N % 500 + 7)
    key, pos = host.get_state()[1].copy(), int(host.get_state()[2])
    for it in range(3):
        expect = host.permutation(N)
        key, pos = ctx.stage_permutation_mt19937(key, pos)
        got = ctx.staged_order()
        bad = np.nonzero(got != expect)[0]
        assert bad.size == 0, "draw %d: %d entries differ, first at %d" % (it, bad.size, bad[0])
        npt.assert_array_equal(key, host.get_state()[1])
        assert pos == host.get_state()[2]
    np.random.seed(5)
    np.random.standard_normal(3)                     # (a cached Gaussian in the state: must survive the round trip)
    st = np.random.get_state()
    expect = np.random.permutation(N)
    after = np.random.get_state()
    np.random.set_state(st)
    assert _rng.take_permutation_staged(ctx, N) is _rng.STAGED
    npt.assert_array_equal(ctx.staged_order(), expect)
    now = np.random.get_state()
    npt.assert_array_equal(now[1], after[1])
    assert now[2:] == after[2:]
    ctx.close()


def _permutations_in_a_process(env, body):
    """Runs `body` (python source; `ctx_for(N)` and numpy in scope) in a process of its own: the library reads BGMM_DEV_OPTIONS when it
    is loaded."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata
def ctx_for(N):
    X, zt = gendata.synth_mixture(N, 2, 3, seed=1)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(2)
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 12)
    ctx.set_assignments(zt)
    return ctx
def in_a_row(ctx, N, calls, disturb_at=()):
    host = np.random.RandomState(N)
    host.random_sample(33)
    key, pos = host.get_state()[1].copy(), int(host.get_state()[2])
    for it in range(calls):
        if it in disturb_at:
            host.random_sample(3)          # (the caller draws something else from the stream: whatever is in flight is void)
            key, pos = host.get_state()[1].copy(), int(host.get_state()[2])
        expect = host.permutation(N)
        key, pos = ctx.stage_permutation_mt19937(key, pos)
        assert np.array_equal(ctx.staged_order(), expect), (N, it)
        assert np.array_equal(key, host.get_state()[1]) and pos == host.get_state()[2], (N, it)
%s
print("PERMUTATIONS OK")
""" % (root, body)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=600)
    assert r.returncode == 0 and "PERMUTATIONS OK" in r.stdout, r.stdout[-2000:]


def test_device_permutation_repairs_draws_that_have_not_settled():
    """The rounds of the permutation's draws are queued blindly, a fixed number at a time; when they have not settled by
    then, more rounds are queued and everything behind the draws runs again.  With the number of rounds cut to 5
    (BGMM_DEV_OPTIONS perm_rounds, read when the library is loaded: a process of its own; perm_pipe=0: the route that generates
    one permutation at a time) that repair runs for every permutation -- which must still be numpy's, generator state included."""
    _permutations_in_a_process({"BGMM_DEV_OPTIONS": "perm_rounds=5,perm_pipe=0"}, """
for N in (70000, 1000003):
    ctx = ctx_for(N)
    in_a_row(ctx, N, 3)
    st = ctx.permutation_stats()
    assert st["rounds_max"] <= 6, st          # (every batch of rounds is 5 long: the statistics count within a batch)
    ctx.close()
""")


@pytest.mark.parametrize("env,hits", [({}, True), ({"BGMM_DEV_OPTIONS": "perm_era=5"}, True), ({"BGMM_DEV_OPTIONS": "perm_chain_rounds=8"}, False)],
                         ids=["as-shipped", "eras-of-five-generations", "never-settled-in-flight"])
def test_device_permutations_in_flight(env, hits):
    """api_perm.hip "permutations in flight": generations queued behind the one handed out, each taking its place in the word
    stream from the one in front of it on the device.  Fourteen permutations in a row per N (the rings of slots, order buffers
    and offsets all wrap), a caller that draws from the stream in between (twice), eras of five generations (the stream's
    buffer starts over every few calls), and generations that cannot settle in the rounds queued for them (eight: every one
    of them is dropped and drawn the old way): numpy's permutation and numpy's state behind it, every time."""
    _permutations_in_a_process(env, """
for N in (4096, 100003, 1000000):
    ctx = ctx_for(N)
    in_a_row(ctx, N, 14, disturb_at=(5, 6))
    st = ctx.permutation_stats()
    assert (st["lookahead_hits"] >= 10) == %r or N < 50000, st
    ctx.close()
""" % hits)


def test_permutations_in_flight_that_cannot_be_set_up_leave_the_single_lookahead():
    """set-up of the permutations in flight is all or nothing.  BGMM_DEV_OPTIONS perm_pipe_fail=1 makes it fail after all its
    allocations: everything it took is released, the state is latched off (bgmm_get_permutation_pipe_state), no stage call
    returns an error, and the single look-ahead of round 3 serves from there -- numpy's permutations and states, ten in a
    row with a foreign draw, and look-ahead hits among them."""
    _permutations_in_a_process({"BGMM_DEV_OPTIONS": "perm_pipe_fail=1"}, """
for N in (4096, 200000):
    ctx = ctx_for(N)
    in_a_row(ctx, N, 10, disturb_at=(4,))
    ps = ctx.permutation_pipe_state()
    assert ps["off"] and not ps["built"] and ps["word_stream_bytes"] == 0, ps
    st = ctx.permutation_stats()
    assert st["lookahead_hits"] >= 6, st
    ctx.close()
""")
    # ... and as shipped it is set up, within its share of the free memory
    _permutations_in_a_process({}, """
ctx = ctx_for(200000)
in_a_row(ctx, 200000, 4, disturb_at=())
ps = ctx.permutation_pipe_state()
assert ps["built"] and not ps["off"] and 0 < ps["word_stream_bytes"] <= (1 << 30) + 8192, ps
ctx.close()
""")


def test_device_permutations_with_the_lookahead_switched_off_and_on_again():
    """bgmm_set_mt_lookahead between permutations: with generations in flight the switch to 0 leaves them behind (every
    permutation is then drawn on the spot), the switch back starts a new era from the caller's state; uniforms staged in
    between (the other generator's look-ahead shares the jump tables with the permutations' word stream).  numpy's
    permutations and states throughout."""
    import random
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    N = 50000
    X, zt = gendata.synth_mixture(N, 2, 3, seed=1)
    ctx = _lib.Context(X, *gendata.demo_prior_params(2), 1.0, 12)
    ctx.set_assignments(zt)
    host = np.random.RandomState(12)
    key, pos = host.get_state()[1].copy(), int(host.get_state()[2])
    r = random.Random(3)
    st = r.getstate()
    mkey, mpos = np.array(st[1][:624], dtype=np.uint32), int(st[1][624])
    for it in range(16):
        if it == 5:
            ctx.set_mt_lookahead(0)
        if it == 9:
            ctx.set_mt_lookahead(-1)
        if it == 12:
            ctx.set_mt_lookahead(2)
        expect = host.permutation(N)
        key, pos = ctx.stage_permutation_mt19937(key, pos)
        npt.assert_array_equal(ctx.staged_order(), expect, err_msg="permutation %d" % it)
        npt.assert_array_equal(key, host.get_state()[1])
        assert pos == host.get_state()[2]
        mkey, mpos = ctx.stage_mt19937(mkey, mpos, None)
        u = ctx.staged_uniforms()
        npt.assert_array_equal(u, np.array([r.random() for _ in range(N)]))
    stats = ctx.permutation_stats()
    assert stats["lookahead_hits"] >= 8 and stats["generated_on_the_spot"] >= 5, stats
    ctx.close()


def test_device_permutations_in_flight_beyond_the_bucketed_links():
    """N > 3.1e6: more than 2 048 buckets of targets -- the generations in flight sort their (target, step) pairs with
    rocPRIM instead (kernels_perm.hip perm_bucket_bounds returns 0).  Six in a row, one foreign draw."""
    _permutations_in_a_process({}, """
ctx = ctx_for(3200000)
in_a_row(ctx, 3200000, 6, disturb_at=(3,))
st = ctx.permutation_stats()
assert st["lookahead_hits"] >= 3, st
ctx.close()
""")


def test_device_permutations_in_flight_soak():
    """The same over many calls (tools/permsoak.py is the long version): 1 500 permutations of 4 097 points in a row from one
    context, eras of three generations (the word stream's buffer starts over every other call), the caller drawing from the
    stream at 45 random moments -- every permutation and every state numpy's."""
    _permutations_in_a_process({"BGMM_DEV_OPTIONS": "perm_era=3"}, """
dice = np.random.RandomState(7)
ctx = ctx_for(4097)
in_a_row(ctx, 4097, 1500, disturb_at=set(dice.randint(0, 1500, size=45).tolist()))
st = ctx.permutation_stats()
assert st["lookahead_hits"] >= 1300, st
ctx.close()
""")


def test_device_permutation_drives_the_pcrp_classes():
    """PCRPMM with the visiting order drawn on the device equals the same run with the order drawn by numpy on the host
    (N < 4096 is the host's: a twin run with a monkey-patched helper gives the host route at the same N)."""
    import random
    from pybgmm_amd.igmm import PCRPMM
    from pybgmm_amd.prior import NIW
    from pybgmm_amd.utils import gendata, rng as _rng
    N, D, K = 6000, 6, 5
    X, zt = gendata.synth_mixture(N, D, K, seed=3, mu_scale=1.5)
    prior = NIW(*gendata.demo_prior_params(D))
    out = []
    for route in ("device", "host"):
        random.seed(4); np.random.seed(4)
        mm = PCRPMM(X, prior, 1.0, None, assignments="rand", K=K, K_max=60)
        if route == "host":
            mm._draw_order = lambda mm=mm: _rng.take_permutation(mm.N, mm._nprng)
        rec, _ = mm.collapsed_gibbs_sampler(4, zt, num_saved=0)
        out.append((mm.components.assignments.copy(), rec["log_marg"], rec["components"], np.random.get_state()[1].copy(), random.random()))
    npt.assert_array_equal(out[0][0], out[1][0])
    assert out[0][1] == out[1][1] and out[0][2] == out[1][2]
    npt.assert_array_equal(out[0][3], out[1][3])         # (both global streams end where the host route leaves them)
    assert out[0][4] == out[1][4]


@pytest.mark.parametrize("pcrp", [False, True], ids=["crp", "pcrp-fresh-permutation"])
def test_short_steps_of_the_benchmarked_mode(pcrp):
    """prune_mode 3 (what bench.py times) on a chain at rest queues short steps: home_kernel between sweep_begin and apply
    (bgmm_get_short_step_stats).  (i) On well-separated data they must come about by themselves and stand, and a change
    made through the API must send the next sweep back to the full kernel set.  (ii) On overlapping data -- movers in most
    sweeps, visits home_kernel cannot decide in all of them -- a short step is FORCED in front of every sweep
    (bgmm_set_home_pass(3)) wherever the schedule opens with a pruned window: every one of them must be refused and redone
    in full.  Every sweep of both chains against the C port of the reference."""
    from oracle import c_oracle
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    power = 1.01 if pcrp else None
    # (i) N large enough for the host to queue pruned-only batches (four windows' worth of mover-free visits)
    N, D, K = 20000, 64, 8
    X, zt = gendata.synth_mixture(N, D, K, seed=13)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    rs = np.random.RandomState(2)
    o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, zt, 4 * K, scipy_tables=False)
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
    ctx.set_tuning(prune_mode=3)
    ctx.set_assignments(zt)
    stood = []
    for it in range(7):
        if it == 3:                                  # a label changed behind the sweeps' back: no short step next
            j = 4321
            ctx.del_item(j); ctx.add_item(j, (int(zt[j]) + 1) % K)
            z_mid = o.z.copy(); z_mid[j] = (int(zt[j]) + 1) % K
            o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, z_mid, 4 * K, scipy_tables=False)
        u = rs.random_sample(N)
        order = rs.permutation(N).astype(np.int64) if pcrp else None
        ctx.sweep(u, order, power)
        o.sweep(u, order, power)
        npt.assert_array_equal(ctx.assignments(), o.z, err_msg="sweep %d" % it)
        lo = o.log_marg()
        assert abs(ctx.log_marg() - lo) <= 1e-9 * abs(lo)
        stood.append(ctx.short_step_stats()["stood"])
    # sweep 0 in full, 1 and 2 short, 3 (after the API change), 4 (after 3's move: shorter windows) and 5 (the first
    # whole-sweep window again) in full, 6 short again
    assert stood == [0, 1, 2, 2, 2, 2, 3] and ctx.short_step_stats()["refused"] == 0, (stood, ctx.short_step_stats())
    ctx.close()
    # (ii)
    N, D, K = 20000, 16, 6
    X, zt = gendata.synth_mixture(N, D, K, seed=21, mu_scale=1.4)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, zt, 4 * K, scipy_tables=False)
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
    ctx.set_tuning(prune_mode=3)
    ctx.set_home_pass(3)
    ctx.set_assignments(zt)
    moves = 0
    for it in range(6):
        u = rs.random_sample(N)
        order = rs.permutation(N).astype(np.int64) if pcrp else None
        ctx.sweep(u, order, power)
        o.sweep(u, order, power)
        npt.assert_array_equal(ctx.assignments(), o.z, err_msg="forced short steps, sweep %d" % it)
        lo = o.log_marg()
        assert abs(ctx.log_marg() - lo) <= 1e-9 * abs(lo)
        moves += ctx.sweep_stats()["moves"]
    st = ctx.short_step_stats()
    assert moves > 0, "the case is meant to have movers"
    # (later sweeps of this chain open with safe-stay / frozen-factor windows, which have no short form: at least the
    # first sweep's attempt must have been made and refused, and none may ever have stood)
    assert st["refused"] >= 1 and st["stood"] == 0, st
    ctx.close()


@pytest.mark.parametrize("N,D,depth", [(50000, 16, -1), (300000, 2, 3), (20000, 16, 1), (3000, 16, 4)])
def test_device_mt19937_lookahead_serves_only_an_untouched_stream(N, D, depth):
    """bgmm_set_mt_lookahead: the uniforms of the next `depth` sweeps (-1: chosen from N; N < 4096: always one) are generated
    in one batch beside the running sweep and handed out sweep by sweep iff the caller's generator is exactly where the
    previous call left it; while a batch's last sweep is served the next batch is started.  A chain with real sweeps between
    the stage calls, with the caller drawing from its generator before some of them (those requests must NOT be served by
    the look-ahead), against random.random() value for value, and label for label against the same chain with the
    look-ahead switched off."""
    import random
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata, rng as _rng
    X, zt = gendata.synth_mixture(N, D, 6, seed=3, mu_scale=1.5)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    out = []
    for ahead in (depth, 0):
        ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 48)
        ctx.set_mt_lookahead(ahead)
        ctx.set_assignments(zt)
        host, dev = random.Random(7), random.Random(7)
        foreign = (2, 9)                                   # sweeps in front of which the caller draws three numbers itself
        for it in range(12):
            if it in foreign:
                assert [host.random() for _ in range(3)] == [dev.random() for _ in range(3)]
            expect = _rng.take_uniforms(N, host)
            assert _rng.stage_uniforms_on_device(ctx, None, dev)
            if it == 4 and ahead:
                # the depth set again between a stage call and its sweep: the batches go, the staged uniforms (served
                # out of one of them) must survive; the next request is generated on the spot
                ctx.set_mt_lookahead(ahead)
            npt.assert_array_equal(ctx.staged_uniforms(), expect)
            assert dev.getstate() == host.getstate()
            ctx.sweep_staged(None)
        st = ctx.mt_lookahead_stats()
        assert st == ({"hits": 8, "misses": 4} if ahead else {"hits": 0, "misses": 12}), st
        out.append((ctx.assignments(), ctx.log_marg()))
        ctx.close()
    npt.assert_array_equal(out[0][0], out[1][0])
    assert out[0][1] == out[1][1]


@pytest.mark.parametrize("cov", ["diag", "fixed"])
@pytest.mark.parametrize("N", [1, 3, 64, 65, 700])
def test_diag_fixed_edge_sizes_and_repeated_orders(cov, N):
    """Diagonal / fixed-variance components at edge sizes, with an index array that repeats and skips
    points as visiting order in every other sweep, against the C port of the reference."""
    from oracle import c_oracle
    from pybgmm_amd import _lib
    D = 6
    rs = np.random.RandomState(77 + N)
    X = np.ascontiguousarray(rs.randn(N, D) + (rs.randint(0, 3, size=(N, 1)) * 5.0))
    if cov == "diag":
        m_0, k_0, v_0, S_0 = np.zeros(D), 0.05, D + 3, np.ones(D)
    else:
        m_0, k_0, v_0 = np.zeros(D), 1.0, 1
        S_0 = np.concatenate([np.full(D, 0.8), np.full(D, 25.0)])
    z0 = np.zeros(N, dtype=np.int64)
    K_max = max(2, min(N, 12))
    o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.5, z0, K_max, cov_type=cov)
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.5, K_max, cov_type=cov,
                       tables=reference_tables(v_0, N) if cov == "diag" else None)
    ctx.set_assignments(z0)
    for it in range(5):
        u = rs.random_sample(N)
        order = None
        if it % 2:
            order = rs.permutation(N)
            if N > 2:
                order[rs.randint(0, N, size=max(N // 10, 1))] = order[rs.randint(0, N, size=max(N // 10, 1))]
        errs = []
        for obj in (o, ctx):
            try:
                obj.sweep(u, order, None)
                errs.append(None)
            except Exception as e:
                errs.append(e)
        assert (errs[0] is None) == (errs[1] is None), errs
        if errs[0] is not None:
            break
        npt.assert_array_equal(ctx.assignments(), o.z)
        lo = o.log_marg()
        assert abs(ctx.log_marg() - lo) <= 1e-9 * max(abs(lo), 1.0)
    ctx.close()


@pytest.mark.parametrize("N", [1, 2, 5, 17, 64, 65, 300])
@pytest.mark.parametrize("D", [12, 33])
def test_tiny_inputs_windowed_kernels(N, D):
    """Edge sizes of the windowed path (MFMA kernels at D >= 12, one visit to a few tiles): default
    configuration against the C port of the reference, started from one table."""
    from oracle import c_oracle
    from pybgmm_amd import _lib
    rs = np.random.RandomState(1000 * D + N)
    X = np.ascontiguousarray(rs.randn(N, D) * 2.0 + (rs.randint(0, 3, size=(N, 1)) * 6.0))
    m_0, k_0, v_0, S_0 = np.zeros(D), 0.05, D + 3, np.eye(D)
    z0 = np.zeros(N, dtype=np.int64)
    K_max = max(2, min(N, 10))
    o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 2.0, z0, K_max)
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 2.0, K_max, tables=reference_tables(v_0, N))
    ctx.set_assignments(z0)
    for it in range(4):
        u = rs.random_sample(N)
        order = rs.permutation(N) if it % 2 else None
        try:
            o.sweep(u, order, None)
            o_err = None
        except Exception as e:
            o_err = e
        try:
            ctx.sweep(u, order, None)
            c_err = None
        except Exception as e:
            c_err = e
        assert (o_err is None) == (c_err is None), (o_err, c_err)
        if o_err is not None:
            break
        npt.assert_array_equal(ctx.assignments(), o.z)
        lo = o.log_marg()
        assert abs(ctx.log_marg() - lo) <= 1e-9 * max(abs(lo), 1.0)
    ctx.close()


@pytest.mark.parametrize("N", [1, 2, 9, 63, 64, 65, 130, 513])
def test_sequential_sweep_tiny_inputs(N):
    """Edge sizes of the one-workgroup sweep (fewer visits than wavefronts, ring and batch boundaries,
    everything unassigned at the start, K_max reached) against the windowed VALU path."""
    from pybgmm_amd import _lib
    rs = np.random.RandomState(N)
    for D in (1, 3):
        X = np.ascontiguousarray(rs.randn(N, D) * 3.0)
        m_0, k_0, v_0, S_0 = np.zeros(D), 0.05, D + 3, np.eye(D)
        for init in ("unassigned", "one"):
            z0 = -np.ones(N, dtype=np.int64) if init == "unassigned" else np.zeros(N, dtype=np.int64)
            K_max = max(2, min(N, 12))
            ctxs = []
            for kind in (0, 1):
                c = _lib.Context(X, m_0, k_0, v_0, S_0, 2.0, K_max)
                c.set_tuning(kernel_kind=kind, resolver_mode=0 if kind == 0 else 1)
                c.set_assignments(z0)
                ctxs.append(c)
            for it in range(4):
                u = rs.random_sample(N)
                order = rs.permutation(N) if it % 2 else None
                errs = []
                for c in ctxs:
                    try:
                        c.sweep(u, order, None)
                        errs.append(None)
                    except Exception as e:            # K_max reached: both paths must refuse alike
                        errs.append(type(e).__name__ + str(e)[:40])
                assert (errs[0] is None) == (errs[1] is None), errs
                if errs[0] is not None:
                    break
                npt.assert_array_equal(ctxs[0].assignments(), ctxs[1].assignments())
                assert ctxs[0].K == ctxs[1].K
                la, lb = ctxs[0].log_marg(), ctxs[1].log_marg()
                assert abs(la - lb) <= 1e-9 * max(abs(lb), 1.0)
                assert ctxs[0].sweep_stats()["lik_evals"] == ctxs[1].sweep_stats()["lik_evals"]
            for c in ctxs:
                c.close()


@pytest.mark.parametrize("prune", [0, 2], ids=["default", "every-window-pruned"])
def test_order_with_repeats_through_the_pruned_kernels(prune):
    """An index array with repeats and gaps as visiting order (the C-ABI takes any), D = 16: certify,
    bucket sort, pruning kernel, sparse draw and the resolver against the C port of the reference."""
    from oracle import c_oracle
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    N, D, K = 9000, 16, 8
    X, zt = gendata.synth_mixture(N, D, K, seed=21, mu_scale=2.0)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    rs = np.random.RandomState(12)
    z0 = zt.copy()
    z0[rs.choice(N, size=300, replace=False)] = rs.randint(0, K, size=300)
    o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, z0, 4 * K)
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K, tables=reference_tables(v_0, N))
    ctx.set_tuning(prune_mode=prune)
    ctx.set_assignments(z0)
    for it in range(4):
        u = rs.random_sample(N)
        order = rs.permutation(N)
        order[rs.randint(0, N, size=400)] = order[rs.randint(0, N, size=400)]
        if it == 2:
            order[100:140] = order[100]                  # the same point forty times in a row
        o.sweep(u, order, None)
        ctx.sweep(u, order, None)
        z = ctx.assignments()
        bad = np.nonzero(z != o.z)[0]
        assert bad.size == 0, "sweep %d: %d labels differ, first at i=%d" % (it, bad.size, bad[0])
        lo = o.log_marg()
        assert abs(ctx.log_marg() - lo) <= 1e-9 * abs(lo)
    ctx.close()


def test_sequential_sweep_refuses_an_order_with_repeats():
    """The small-D sweep fetches a visit's home slot ahead of time, which is sound only when no index
    comes twice: a visiting order with repeats must take the windowed kernels (same trajectory as the
    forced VALU path), a permutation the sequential kernel (one 'window', one step)."""
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    N, D, K = 6000, 2, 6
    X, zt = gendata.synth_mixture(N, D, K, seed=77, mu_scale=1.5)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    from oracle import c_oracle
    rs = np.random.RandomState(3)
    ctxs = []
    for kind in (0, 1):
        c = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 64, tables=reference_tables(v_0, N))
        c.set_tuning(kernel_kind=kind, resolver_mode=0 if kind == 0 else 1)
        c.set_assignments(zt)
        ctxs.append(c)
    o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, zt, 64)
    for it in range(4):
        u = rs.random_sample(N)
        order = rs.permutation(N)
        if it % 2 == 1:
            order[rs.randint(0, N, size=200)] = order[rs.randint(0, N, size=200)]     # repeats (and gaps)
        for c in ctxs:
            c.sweep(u, order, None)
        o.sweep(u, order, None)
        npt.assert_array_equal(ctxs[0].assignments(), ctxs[1].assignments())
        npt.assert_array_equal(ctxs[0].assignments(), o.z)     # (the in-launch resolver leaves such stretches alone)
        st = ctxs[0].sweep_stats()
        if it % 2 == 0:
            assert st["steps"] == 1 and st["windows"] == 1          # the sequential kernel did the sweep
        else:
            assert st["steps"] > 1                                  # the windowed kernels did
        assert st["lik_evals"] == ctxs[1].sweep_stats()["lik_evals"]
    for c in ctxs:
        c.close()


def test_soak_sequential_small_d_against_windowed():
    """tools/soak_seq.py: 24 random D <= 4 configurations, the sequential one-wavefront sweep (with a
    small LDS plan in a third of them, so that it hands over to the windowed kernels mid-sweep) against
    the windowed VALU path; any difference in the trajectory fails."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak_seq.py"), "24", "5"],
                       capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


def test_soak_default_against_full_evaluation():
    """tools/soak.py: 24 random configurations (shape, separation, covariance type, visiting order,
    seating exponent, flipped / unassigned labels, add_item / del_item between sweeps), 10 sweeps each:
    the default configuration and the plain full evaluation walk the same chain."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "soak.py"), "24", "11"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("case", API_CASES)
def test_distribution_dict_rand_k_and_streams_match_reference(case):
    """the whole class path with the distribution dict ON (num_saved == K:
    igmm/igmm.py:128-197, crpmm.py:49-50 -- MAP estimates, label-switch ordering, the np.random Dirichlet
    draw), then rand_k of every component (gaussian_components.py:291-303, prior/wishart.py), then the next
    draws of BOTH global streams: everything as captured from the reference by tests/golden/make_golden.py."""
    import random
    from pybgmm_amd.igmm import CRPMM, PCRPMM
    from pybgmm_amd.prior import NIW
    g = np.load(os.path.join(GOLDEN_DIR, case + ".npz"), allow_pickle=False)
    random.seed(int(g["seed_random"]))
    np.random.seed(int(g["seed_numpy"]))
    cls = {"CRPMM": CRPMM, "PCRPMM": PCRPMM}[str(g["model"])]
    mm = cls(g["X"], NIW(g["m_0"], float(g["k_0"]), int(g["v_0"]), g["S_0"]), float(g["alpha"]), None,
             assignments="rand", K=int(g["K_arg"]), K_max=int(g["K_max"]))
    record, dist = mm.collapsed_gibbs_sampler(int(g["n_iter"]), g["true_assignments"], num_saved=3,
                                              weight_first=bool(g["weight_first"]))
    npt.assert_array_equal(np.array(record["components"]), g["rec_components"])
    npt.assert_allclose(np.array(record["log_marg"]), g["rec_log_marg"], rtol=1e-9)
    npt.assert_array_equal(mm.components.assignments, g["final_z"])
    for key in ("mean", "variance", "weights"):
        assert dist[key].shape == g["dist_" + key].shape
        npt.assert_allclose(dist[key], g["dist_" + key], rtol=1e-9, atol=1e-12)
    for k in range(int(g["final_K"])):
        mu, sigma = mm.components.rand_k(k)
        npt.assert_allclose(np.ravel(mu), g["rand_k_mu"][k], rtol=1e-8)
        npt.assert_allclose(np.ravel(sigma), g["rand_k_sigma"][k], rtol=1e-8)
    npt.assert_array_equal(np.array([random.random() for _ in range(4)]), g["after_random"])
    npt.assert_array_equal(np.random.random_sample(4), g["after_numpy"])


@pytest.mark.parametrize("cov", ["full", "diag", "fixed"])
def test_cache_del_restore_is_the_reference_idiom(cov):
    """igmm/crpmm.py:60-65, 82-85: cache_component_stats, del_item, restore_component_from_stats,
    assignments[i] = k_old -- afterwards the state is what it was (statistics bit-identical), and a sweep
    from there equals a sweep of an untouched twin."""
    from pybgmm_amd.gaussian.gaussian_components import GaussianComponents, GaussianComponentsDiag
    from pybgmm_amd.prior import NIW
    from pybgmm_amd.utils import gendata
    N, D, K = 600, 5, 4
    X, zt = gendata.synth_mixture(N, D, K, seed=77, mu_scale=2.0)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    if cov == "diag":
        S_0 = np.ascontiguousarray(np.diag(S_0))
    if cov == "fixed":
        from pybgmm_amd.gaussian.gaussian_components_fixedvar import FixedVarPrior, GaussianComponentsFixedVar
        comps = [GaussianComponentsFixedVar(X, FixedVarPrior(0.49, np.zeros(D), 16.0), zt.copy(), 16) for _ in range(2)]
    else:
        cls = GaussianComponents if cov == "full" else GaussianComponentsDiag
        comps = [cls(X, NIW(m_0, k_0, v_0, S_0), zt.copy(), 16) for _ in range(2)]
    a = comps[0]
    for i in (3, 250, 599):
        k_old = int(a.assignments[i])
        stats_old = a.cache_component_stats(k_old)
        a.del_item(i)
        assert int(a.assignments[i]) == -1 and int(a.counts[k_old]) == stats_old[4] - 1
        a.restore_component_from_stats(k_old, *stats_old)
        a.set_assignment(i, k_old)
    for x, y in zip(a._ctx.stats(True)[:2], comps[1]._ctx.stats(True)[:2]):
        npt.assert_array_equal(x, y)
    for x, y in zip(a._ctx.stats(True)[2:], comps[1]._ctx.stats(True)[2:]):
        npt.assert_allclose(x, y, rtol=1e-10, atol=1e-12)
    npt.assert_array_equal(a.counts, comps[1].counts)
    for k in range(a.K):                                   # (fixed variance: the sum of squares behind the log marginal too)
        for x, y in zip(a._ctx.raw_stats(k), comps[1]._ctx.raw_stats(k)):
            npt.assert_array_equal(x, y)
    assert a._ctx.log_marg() == comps[1]._ctx.log_marg()
    u = np.random.RandomState(5).random_sample(N)
    for c in comps:
        c._ctx.sweep(u)
    npt.assert_array_equal(comps[0].assignments, comps[1].assignments)
    for c in comps:
        c._ctx.close()


def test_del_component_is_the_swap_with_last():
    """GaussianComponents.del_component(k) (gaussian_components.py:188-205): the last label takes k's place -- statistics,
    count and members; K shrinks by one.  Members the deleted component still had become unassigned (include/bgmm.h)."""
    from pybgmm_amd.gaussian.gaussian_components import GaussianComponents
    from pybgmm_amd.prior import NIW
    from pybgmm_amd.utils import gendata
    N, D, K = 500, 3, 5
    X, zt = gendata.synth_mixture(N, D, K, seed=5, mu_scale=3.0)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    a = GaussianComponents(X, NIW(m_0, k_0, v_0, S_0), zt.copy(), 16)
    m0, S0 = a._ctx.stats(False)[:2]
    c0 = a.counts.copy()
    z0 = a.assignments.copy()
    a.del_component(1)
    assert a.K == K - 1
    z1 = a.assignments
    npt.assert_array_equal(z1[z0 == 1], -1)                       # its members: unassigned
    npt.assert_array_equal(z1[z0 == K - 1], 1)                    # the last label moved into its place
    for k in (0, 2, 3):
        npt.assert_array_equal(z1[z0 == k], k)
    m1, S1 = a._ctx.stats(False)[:2]
    npt.assert_array_equal(m1[1], m0[K - 1]); npt.assert_array_equal(S1[1], S0[K - 1])
    npt.assert_array_equal(a.counts[:K - 1], [c0[0], c0[K - 1], c0[2], c0[3]])
    # the unassigned points are seated again by a sweep (they never "stay"): a consistent state afterwards
    a._ctx.sweep(np.random.RandomState(1).random_sample(N))
    assert (a.assignments >= 0).all() and int(a.counts.sum()) == N
    a.del_component(a.K - 1)                                      # deleting the last label: nothing to swap
    assert int(a.counts.sum()) == N - int((a.assignments < 0).sum())
    a._ctx.close()


@pytest.mark.parametrize("D,pcrp", [(16, True), (3, False), (64, False)])
def test_partial_sweep_and_resume(D, pcrp):
    """(i) bgmm_set_sweep_visits: a sweep that stops after n visits equals the C oracle's partial sweep (SURVEY 8b's
    n_visits).  (ii) Checkpoint / resume: labels + raw statistics + counts put into a NEW context
    (bgmm_set_assignments, then bgmm_set_stats per component) continue the chain identically, statistics bit-equal."""
    from oracle import c_oracle
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    N, K = 20000, 12
    X, zt = gendata.synth_mixture(N, D, K, seed=9 + D, mu_scale=0.9 if D >= 16 else 2.0)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    rs = np.random.RandomState(D)
    z0 = np.unique(rs.randint(0, K, N), return_inverse=True)[1]
    us = rs.random_sample((5, N))
    orders = [rs.permutation(N).astype(np.int64) if pcrp else None for _ in range(5)]
    pw = lambda it: 1.01 if (pcrp and it > 0) else None
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
    ctx.set_assignments(z0)
    o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, z0, 4 * K, scipy_tables=False)
    ctx.sweep(us[0], orders[0], pw(0)); o.sweep(us[0], orders[0], pw(0))
    for n_vis in (1, 777, N // 2):                               # partial sweeps, each from the state the last one left
        ctx.set_sweep_visits(n_vis)
        ctx.sweep(us[1], orders[1], pw(1))
        o.sweep(us[1], orders[1], pw(1), n_visits=n_vis)
        npt.assert_array_equal(ctx.assignments(), o.z)
    ctx.sweep(us[2], orders[2], pw(2)); o.sweep(us[2], orders[2], pw(2))      # (and whole sweeps again afterwards)
    npt.assert_array_equal(ctx.assignments(), o.z)
    # ---- checkpoint
    z_ck, cnt_ck = ctx.assignments(), ctx.counts()
    raw = [ctx.raw_stats(k) for k in range(ctx.K)]
    # ---- resume in a new context
    ctx2 = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
    ctx2.set_assignments(z_ck)
    for k, (m, S) in enumerate(raw):
        ctx2.set_stats(k, m, S, int(cnt_ck[k]))
    for it in (3, 4):
        for c in (ctx, ctx2):
            c.sweep(us[it], orders[it], pw(it))
        npt.assert_array_equal(ctx2.assignments(), ctx.assignments())
        npt.assert_array_equal(ctx2.counts(), ctx.counts())
    for k in range(ctx.K):
        for x, y in zip(ctx2.raw_stats(k), ctx.raw_stats(k)):
            npt.assert_array_equal(x, y)
    assert abs(ctx2.log_marg() - ctx.log_marg()) <= 1e-10 * abs(ctx.log_marg())
    ctx.close(); ctx2.close()


@pytest.mark.parametrize("model", ["CRPMM", "PCRPMM"])
def test_chain_c_equals_a_global_seed_run(model):
    """chain c (its own random.Random / RandomState, pybgmm_amd/chains.py) equals a plain run
    under random.seed(s + c); np.random.seed(s + c) -- labels, K, log marginals -- and the gathered stack
    equals the per-chain outputs.  Both chains' contexts are alive on the device at the same time (the
    per-device kernel attributes and the contexts' state are independent)."""
    import random
    from pybgmm_amd import chains
    from pybgmm_amd.igmm import CRPMM, PCRPMM
    from pybgmm_amd.prior import NIW
    from pybgmm_amd.utils import gendata
    cls = {"CRPMM": CRPMM, "PCRPMM": PCRPMM}[model]
    N, D, K, s, n_iter = 1500, 6, 5, 40, 4
    X, zt = gendata.synth_mixture(N, D, K, seed=3, mu_scale=1.5)
    prior = NIW(*gendata.demo_prior_params(D))
    runs = [chains.run_chain(cls, X, prior, 1.0, n_iter, s, c, 0, true_assignments=zt, K=K, K_max=60)
            for c in range(2)]
    zs = [m.components.assignments for m, _ in runs]
    lms = [np.array(r["log_marg"]) for _, r in runs]
    assert not np.array_equal(zs[0], zs[1]), "the two chains must differ"
    for c in range(2):
        random.seed(s + c)
        np.random.seed(s + c)
        ref = cls(X, prior, 1.0, None, assignments="rand", K=K, K_max=60)
        rec, _ = ref.collapsed_gibbs_sampler(n_iter, zt, num_saved=0)
        npt.assert_array_equal(ref.components.assignments, zs[c])
        npt.assert_array_equal(np.array(rec["components"]), np.array(runs[c][1]["components"]))
        npt.assert_array_equal(np.array(rec["log_marg"]), lms[c])
    Z, LM = chains.gather_chains(zs[0], lms[0])                 # (single process: the stack of one)
    npt.assert_array_equal(Z[0], zs[0])
    npt.assert_array_equal(np.stack(zs), np.stack([Z[0], zs[1]]))


@pytest.mark.parametrize("case", ["rest-evaluated", "rest-certified", "moving", "pcrp-rest", "pcrp-moving", "wrong-labels"])
@pytest.mark.parametrize("depth", [-1, 1])
def test_pipelined_sweeps_equal_plain_sweeps(case, depth):
    """bgmm_sweep_staged_begin / _end: the driver stages sweep k + 1's inputs (device generators: uniforms, for pCRP the
    permutation) while sweep k is in the queue.  Against the plain loop (stage, sweep, stage, sweep) from twin generators:
    labels after every sweep, log marginal, generator states -- for chains at rest in both modes (short / lean steps:
    the halves really overlap), chains that move (the sweep runs to its end inside _begin), wrong labels repaired on the
    way to rest (refused steps), with look-ahead batches of one sweep (every stage call wants to start a generation into
    the buffer the running sweep reads: put off until _end) and of the default depth."""
    import random
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata, rng as _rng
    pcrp = case.startswith("pcrp")
    moving = case.endswith("moving")
    N, D, K = (30000, 64, 8) if not moving else (30000, 16, 8)
    X, zt = gendata.synth_mixture(N, D, K, seed=5, mu_scale=1.2 if moving else 4.0)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    z0 = zt.copy()
    if case == "wrong-labels":
        idx = np.random.RandomState(1).choice(N, size=60, replace=False)
        z0[idx] = (z0[idx] + 1) % K
    n_sw = 9
    out = []
    for pipelined in (True, False):
        ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 6 * K)
        ctx.set_tuning(prune_mode=0 if case == "rest-certified" else 3)
        ctx.set_mt_lookahead(depth)
        ctx.set_assignments(z0)
        r, nr = random.Random(11), np.random.RandomState(11)

        def stage():
            order = _rng.take_permutation_staged(ctx, N, nr) if pcrp else None
            assert _rng.stage_uniforms_on_device(ctx, None if order is _rng.STAGED else order, r)
        zs = []
        stage()
        for it in range(n_sw):
            power = 1.01 if (pcrp and it > 0) else None
            if pipelined:
                ctx.sweep_staged_begin(power)
                if it + 1 < n_sw:
                    stage()                          # sweep it + 1's inputs while sweep it runs
                ctx.sweep_staged_end()
            else:
                ctx.sweep_staged(power)
                if it + 1 < n_sw:
                    stage()
            zs.append(ctx.assignments())
        out.append((zs, ctx.log_marg(), r.getstate(), nr.get_state()[1].copy(), ctx.short_step_stats(), ctx.sweep_stats()["moves"]))
        if pipelined and case.startswith("rest"):
            with pytest.raises(_lib.BGMMError):      # (a chain at rest: _begin returns with the sweep in flight) a sweep
                ctx.sweep_staged_begin(None)         # call between the halves is refused
                ctx.sweep_staged(None)
            ctx.sweep_staged_end()
        ctx.close()
    for it in range(n_sw):
        npt.assert_array_equal(out[0][0][it], out[1][0][it], err_msg="sweep %d" % it)
    assert out[0][1] == out[1][1]
    assert out[0][2] == out[1][2]
    npt.assert_array_equal(out[0][3], out[1][3])
    assert out[0][4] == out[1][4], (out[0][4], out[1][4])      # the same steps stood and were refused
    if case == "rest-evaluated":
        assert out[0][4]["stood"] >= n_sw - 3


@pytest.mark.parametrize("depth", [0, 1, -1])
@pytest.mark.parametrize("case", ["forced", "flip", "pcrp-forced", "pcrp-flip", "host-inputs"])
def test_pipelined_sweeps_with_refused_steps_in_flight(case, depth):
    """The redo path of bgmm_sweep_staged_end: a short step that is REFUSED while the next sweep's inputs have already
    been staged.  The refused step must be redone from the inputs the sweep was begun with -- not from what the stage
    calls of the meantime left (on-the-spot generation into d_u / d_order with the look-ahead off, the batch buffer a
    look-ahead generation would overwrite, the permutation look-ahead's buffer after the swap).
    `forced`: bgmm_set_home_pass(3) tries a short step in every sweep of a chain that moves (refused again and again);
    `flip`: a chain at rest, one label set wrong before _begin (short steps tried in every sweep, as in `forced`): the step
    in flight meets a mover and is refused;
    `host-inputs`: the next sweep's inputs come through bgmm_stage_sweep_inputs (host arrays into the context's own
    buffers: the call finishes the sweep in flight first).  Against the plain loop, with refusals asserted."""
    import random
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata, rng as _rng
    pcrp = case.startswith("pcrp")
    forced = case.endswith("forced")
    N, D, K = 30000, 16, 8
    X, zt = gendata.synth_mixture(N, D, K, seed=5, mu_scale=1.2 if forced else 4.0)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    n_sw = 8
    flips = {3: 17, 5: 29000, 6: 12345}                  # sweep -> data index whose label is set wrong before _begin
    out = []
    for pipelined in (True, False):
        ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 6 * K)
        ctx.set_tuning(prune_mode=3)
        ctx.set_home_pass(3)                             # (a short step in front of EVERY sweep: also right after set_label)
        ctx.set_mt_lookahead(depth)
        ctx.set_assignments(zt)
        r, nr = random.Random(11), np.random.RandomState(11)
        hr = np.random.RandomState(7)

        def stage():
            if case == "host-inputs":
                ctx.stage(hr.random_sample(N), hr.permutation(N).astype(np.int64))
                return
            order = _rng.take_permutation_staged(ctx, N, nr) if pcrp else None
            assert _rng.stage_uniforms_on_device(ctx, None if order is _rng.STAGED else order, r)
        zs = []
        stage()
        for it in range(n_sw):
            power = 1.01 if (pcrp and it > 0) else None
            if not forced and it in flips:
                i = flips[it]
                ctx.set_label(i, (int(ctx.assignments()[i]) + 1) % ctx.K)
            if pipelined:
                ctx.sweep_staged_begin(power)
                if it + 1 < n_sw:
                    stage()
                ctx.sweep_staged_end()
            else:
                ctx.sweep_staged(power)
                if it + 1 < n_sw:
                    stage()
            zs.append(ctx.assignments())
        out.append((zs, ctx.log_marg(), r.getstate(), nr.get_state()[1].copy(), ctx.short_step_stats()))
        ctx.close()
    for it in range(n_sw):
        npt.assert_array_equal(out[0][0][it], out[1][0][it], err_msg="sweep %d" % it)
    assert out[0][1] == out[1][1]
    assert out[0][2] == out[1][2]
    npt.assert_array_equal(out[0][3], out[1][3])
    assert out[0][4] == out[1][4], (out[0][4], out[1][4])
    assert out[0][4]["refused"] >= 1, out[0][4]


def test_calls_between_begin_and_end_finish_the_sweep_in_flight():
    """Entry points that read or change the state while bgmm_sweep_staged_begin has left a sweep in the queue finish that
    sweep first (they used to run against the queued sweep): labels read between the halves are the labels after the sweep,
    a label set between the halves lands behind it, bgmm_set_mt_lookahead cannot free a buffer the sweep reads, and
    bgmm_sweep_staged_end then only reports.  An order outside 0 .. N-1 is refused before any generator state moves."""
    import random
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata, rng as _rng
    N, D, K = 20000, 16, 6
    X, zt = gendata.synth_mixture(N, D, K, seed=3)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    outs = []
    for pipelined in (True, False):
        ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
        ctx.set_tuning(prune_mode=3)
        ctx.set_assignments(zt)
        r = random.Random(4)
        got = []
        for it in range(6):
            assert _rng.stage_uniforms_on_device(ctx, None, r)
            if pipelined:
                ctx.sweep_staged_begin(None)
                if it == 2:
                    got.append(ctx.assignments())        # finishes the sweep in flight
                    ctx.set_label(5, (int(zt[5]) + 1) % K)
                if it == 3:
                    ctx.set_mt_lookahead(1)
                if it == 4:
                    got.append(ctx.log_marg())
                ctx.sweep_staged_end()
            else:
                ctx.sweep_staged(None)
                if it == 2:
                    got.append(ctx.assignments())
                    ctx.set_label(5, (int(zt[5]) + 1) % K)
                if it == 3:
                    ctx.set_mt_lookahead(1)
                if it == 4:
                    got.append(ctx.log_marg())
        got.append(ctx.assignments())
        # a bad order: refused, and the generator handed in is where it was
        state = r.getstate()
        bad = np.arange(N, dtype=np.int64)
        bad[7] = N
        with pytest.raises(_lib.BGMMError):
            _rng.stage_uniforms_on_device(ctx, bad, r)
        assert r.getstate() == state
        assert _rng.stage_uniforms_on_device(ctx, None, r)
        ctx.sweep_staged(None)
        got.append(ctx.assignments())
        outs.append(got)
        ctx.close()
    npt.assert_array_equal(outs[0][0], outs[1][0])
    assert outs[0][1] == outs[1][1]
    npt.assert_array_equal(outs[0][2], outs[1][2])
    npt.assert_array_equal(outs[0][3], outs[1][3])


def test_group_sweep_equals_separate_sweeps():
    """bgmm_group_sweep_staged (many chains per GPU): chains that take the one-workgroup sweep go through ONE pair of
    launches, the others (here: D = 16, and a D = 2 chain pinned to the windowed kernels) are swept on their own -- every
    chain label for label what separate bgmm_sweep_staged calls on a twin context give, from the same seeds.  One of the
    small chains gets a tiny LDS plan, so that its group sweep is handed over to the windowed kernels half way."""
    import random
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    shapes = [(20000, 2, 20, 0), (20000, 2, 20, 0), (9000, 3, 8, 0), (6000, 16, 6, 0), (20000, 2, 20, 1), (20000, 2, 20, 2),
              (20000, 2, 20, 0)]                             # (N, D, K, variant: 1 = forced VALU kernels, 2 = tiny LDS plan)
    data = {}
    for (N, D, K, _) in shapes:
        if (N, D, K) not in data:
            data[(N, D, K)] = gendata.synth_mixture(N, D, K, seed=N % 97 + D) + gendata.demo_prior_params(D)

    def build():
        out = []
        for c, (N, D, K, var) in enumerate(shapes):
            X, zt, m_0, k_0, v_0, S_0 = data[(N, D, K)]
            ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 12 * K)
            if var == 1:
                ctx.set_tuning(kernel_kind=1)
            if var == 2:
                ctx.set_seq_plan(K + 3)
            z0 = np.unique(np.random.RandomState(c).randint(0, K, N), return_inverse=True)[1] if c % 2 else zt
            ctx.set_assignments(z0)
            _, key, _ = random.Random(100 + c).getstate()
            out.append([ctx, np.asarray(key[:-1], dtype=np.uint32), int(key[-1])])
        return out
    grouped, solo = build(), build()
    rs = np.random.RandomState(9)
    for it in range(4):
        orders = [rs.permutation(sh[0]).astype(np.int64) if (it % 2 and c % 3 == 0) else None for c, sh in enumerate(shapes)]
        powers = [1.02 if (it >= 2 and c % 2 == 0) else None for c in range(len(shapes))]
        for chains_ in (grouped, solo):
            for c, ch in enumerate(chains_):
                ch[1], ch[2] = ch[0].stage_mt19937(ch[1], ch[2], orders[c])
        _lib.group_sweep_staged([ch[0] for ch in grouped], powers)
        for c, ch in enumerate(solo):
            ch[0].sweep_staged(powers[c])
        for c in range(len(shapes)):
            npt.assert_array_equal(grouped[c][0].assignments(), solo[c][0].assignments(), err_msg="sweep %d chain %d" % (it, c))
            # (pipelined and plain frozen-factor windows -- a lone chain, two chains of a group that share launches, a chain whose
            #  pipeline broke -- rebuild the factors behind the log determinants by different routes: the last digit may
            #  differ, nothing else)
            lg, ls = grouped[c][0].log_marg(), solo[c][0].log_marg()
            assert abs(lg - ls) <= 1e-13 * abs(ls)
            sg, ss = grouped[c][0].sweep_stats(), solo[c][0].sweep_stats()
            assert sg["moves"] == ss["moves"]
    assert not np.array_equal(grouped[0][0].assignments(), grouped[1][0].assignments()), "chains with different seeds must differ"
    with pytest.raises(_lib.BGMMError):                     # (a context twice in one group)
        _lib.group_sweep_staged([grouped[0][0], grouped[0][0]])
    for ch in grouped + solo:
        ch[0].close()


def test_group_sweep_runs_large_chains_concurrently_and_equal_to_solo():
    """bgmm_group_sweep_staged with chains that cannot take the one-workgroup sweep (D = 64 / 16 full covariance from the
    reference's "rand" start -- frozen-factor and safe-stay windows --, a diagonal-covariance chain, a chain at the truth):
    they run CONCURRENTLY, one stream and one host thread each (G chains at any D).  Every chain label for
    label, log marginal for log marginal, counter for counter what bgmm_sweep_staged on a twin context gives; a small-D
    chain in the same group still goes through the one-workgroup launch."""
    import random
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    shapes = [(12000, 64, 12, "full", "rand"), (12000, 64, 12, "full", "rand"), (9000, 16, 8, "full", "rand"),
              (8000, 16, 6, "diag", "rand"), (12000, 64, 12, "full", "true"), (20000, 2, 20, "full", "rand")]
    data = {}
    for (N, D, K, cov, _) in shapes:
        if (N, D, K) not in data:
            data[(N, D, K)] = gendata.synth_mixture(N, D, K, seed=N % 89 + D)

    def build():
        out = []
        for c, (N, D, K, cov, init) in enumerate(shapes):
            X, zt = data[(N, D, K)]
            m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
            if cov == "diag":
                S_0 = np.ascontiguousarray(np.diag(S_0))
            ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 8 * K, cov_type=cov)
            z0 = np.unique(np.random.RandomState(40 + c).randint(0, K, N), return_inverse=True)[1] if init == "rand" else zt
            ctx.set_assignments(z0)
            _, key, _ = random.Random(200 + c).getstate()
            out.append([ctx, np.asarray(key[:-1], dtype=np.uint32), int(key[-1])])
        return out
    grouped, solo = build(), build()
    rs = np.random.RandomState(3)
    moved = 0
    for it in range(3):
        orders = [rs.permutation(sh[0]).astype(np.int64) if (it == 1 and c % 2 == 0) else None for c, sh in enumerate(shapes)]
        powers = [1.02 if (it == 2 and c % 2 == 0) else None for c in range(len(shapes))]
        for chains_ in (grouped, solo):
            for c, ch in enumerate(chains_):
                ch[1], ch[2] = ch[0].stage_mt19937(ch[1], ch[2], orders[c])
        _lib.group_sweep_staged([ch[0] for ch in grouped], powers)
        for c, ch in enumerate(solo):
            ch[0].sweep_staged(powers[c])
        for c in range(len(shapes)):
            npt.assert_array_equal(grouped[c][0].assignments(), solo[c][0].assignments(), err_msg="sweep %d chain %d" % (it, c))
            # (pipelined and plain frozen-factor windows -- a lone chain, two chains of a group that share launches, a chain whose
            #  pipeline broke -- rebuild the factors behind the log determinants by different routes: the last digit may
            #  differ, nothing else)
            lg, ls = grouped[c][0].log_marg(), solo[c][0].log_marg()
            assert abs(lg - ls) <= 1e-13 * abs(ls)
            sg, ss = grouped[c][0].sweep_stats(), solo[c][0].sweep_stats()
            assert sg["moves"] == ss["moves"]
        moved += grouped[0][0].sweep_stats()["moves"]
    assert moved > 5000, "the D = 64 chains are meant to burn in"
    assert not np.array_equal(grouped[0][0].assignments(), grouped[1][0].assignments()), "chains with different seeds must differ"
    for ch in grouped + solo:
        ch[0].close()


@pytest.mark.parametrize("model,D", [("CRPMM", 16), ("PCRPMM", 64)])
def test_sample_chains_at_larger_dimensions_equals_solo_runs(model, D):
    """`CRPMM.sample_chains` / `PCRPMM.sample_chains` at dimensions whose chains cannot take the one-workgroup sweep: the
    group call runs them concurrently (one stream + host thread each) and, from the reference's "rand" start, shares their
    frozen-factor launches.  Chain c through the CLASSES -- its own `random.Random` / `RandomState`, device permutations for
    PCRPMM -- is the chain a solo run with seed s + c gives: labels, K per sweep, log marginals."""
    from pybgmm_amd import chains
    from pybgmm_amd.igmm import CRPMM, PCRPMM
    from pybgmm_amd.prior import NIW
    from pybgmm_amd.utils import gendata
    cls = {"CRPMM": CRPMM, "PCRPMM": PCRPMM}[model]
    N, K, s, n_iter, G = 6000, 8, 23, 3, 3
    X, zt = gendata.synth_mixture(N, D, K, seed=5 + D, mu_scale=2.0)
    prior = NIW(*gendata.demo_prior_params(D))
    runs = cls.sample_chains(X, prior, 1.0, chains=G, n_iter=n_iter, seed=s, true_assignments=zt, K=K, K_max=64)
    moved = 0
    for c in range(G):
        m_solo, rec_solo = chains.run_chain(cls, X, prior, 1.0, n_iter, s, c, 0, true_assignments=zt, K=K, K_max=64)
        npt.assert_array_equal(runs[c][0].components.assignments, m_solo.components.assignments)
        npt.assert_array_equal(np.array(runs[c][1]["components"]), np.array(rec_solo["components"]))
        # (solo chains pipeline their frozen-factor windows, grouped ones share plain windows: the last digit may differ)
        npt.assert_allclose(np.array(runs[c][1]["log_marg"]), np.array(rec_solo["log_marg"]), rtol=1e-13)
        moved += int(np.sum(runs[c][0].components.assignments != zt))
    assert not np.array_equal(runs[0][0].components.assignments, runs[1][0].components.assignments)


@pytest.mark.parametrize("model", ["CRPMM", "PCRPMM"])
def test_run_chains_on_device_equals_solo_runs(model):
    """chains.run_chains_on_device: G model objects on one GPU, their sampler loops in lockstep, every round of sweeps one
    group call.  Chain c is the chain a solo run with seed s + c gives: labels, K per sweep, log marginals."""
    from pybgmm_amd import chains
    from pybgmm_amd.igmm import CRPMM, PCRPMM
    from pybgmm_amd.prior import NIW
    from pybgmm_amd.utils import gendata
    cls = {"CRPMM": CRPMM, "PCRPMM": PCRPMM}[model]
    N, D, K, s, n_iter, G = 5000, 2, 6, 11, 4, 5        # (N >= 4096: PCRPMM's permutations are drawn on the device, one per thread)
    X, zt = gendata.synth_mixture(N, D, K, seed=3, mu_scale=2.0)
    prior = NIW(*gendata.demo_prior_params(D))
    runs = cls.sample_chains(X, prior, 1.0, chains=G, n_iter=n_iter, seed=s, true_assignments=zt, K=K, K_max=80)   # (= chains.run_chains_on_device)
    for c in range(G):
        m_solo, rec_solo = chains.run_chain(cls, X, prior, 1.0, n_iter, s, c, 0, true_assignments=zt, K=K, K_max=80)
        npt.assert_array_equal(runs[c][0].components.assignments, m_solo.components.assignments)
        npt.assert_array_equal(np.array(runs[c][1]["components"]), np.array(rec_solo["components"]))
        # (solo chains pipeline their frozen-factor windows, grouped ones share plain windows: the last digit may differ)
        npt.assert_allclose(np.array(runs[c][1]["log_marg"]), np.array(rec_solo["log_marg"]), rtol=1e-13)
    assert not np.array_equal(runs[0][0].components.assignments, runs[1][0].components.assignments)


def test_label_gather_through_the_c_abi():
    """bgmm_comm_* + bgmm_gather_labels (RCCL all-gather of the final labels).  One GPU here, so a
    communicator of one rank: the gathered stack is the chain's own labelling (unassigned points as -1)."""
    from pybgmm_amd import _lib
    g = Golden("kat1_igmm_2d")
    ctx = _lib.Context(g.X, g.m_0, g.k_0, g.v_0, g.S_0, g.alpha, g.K_max, tables=reference_tables(g.v_0, g.N))
    ctx.set_assignments(g.z_init)
    ctx.sweep(g.u[0], g.sweep_order(0), g.sweep_power(0))
    ctx.del_item(3)
    comm = _lib.Comm(0, 1, _lib.Comm.unique_id(), device=0)
    z_all = ctx.gather_labels(comm, 1)
    assert z_all.shape == (1, g.N) and z_all.dtype == np.int64
    assert np.array_equal(z_all[0], ctx.assignments()) and z_all[0, 3] == -1
    comm.close()
    ctx.close()


@pytest.mark.parametrize("init,D,alpha", [("one-by-one", 5, 1e300), ("each-in-own", 2, 1.0), ("one-by-one", 16, 1e15)],
                         ids=["one-by-one-every-visit-a-new-table", "each-in-own-N-5000", "one-by-one-D16-grows-twice"])
def test_k_max_none_means_up_to_N_components(init, D, alpha):
    """VERDICT r4 missing #4: ``K_max=None`` is "N components" in the reference (gaussian_components.py:81-83).  Here the
    device starts with max(1024, 4 K_init) slots and the sampler loop moves the chain into a context with twice the slots
    at the visit that needs one more (gaussian_components.py: resume_in_larger_context) -- no BGMM_EKMAX short of N, and
    the trajectory is the one of a chain that had all N slots from the start: the C oracle with K_max = N, two sweeps
    through the classes with the caller's ``random`` stream."""
    import random
    from oracle import c_oracle
    from pybgmm_amd.igmm import CRPMM
    from pybgmm_amd.prior import NIW
    from pybgmm_amd.utils import gendata
    N = 5000
    X, _ = gendata.synth_mixture(N, D, 6, seed=5 + D)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    random.seed(11)
    mm = CRPMM(X, NIW(m_0, k_0, v_0, S_0), alpha, None, assignments=init)
    assert mm.components.K_max_auto
    k_start = mm.components.K_max
    z0 = mm.components.assignments
    o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, alpha, z0, N, scipy_tables=False)
    host = random.Random(11)
    mm.record_metrics = False
    for it in range(2):
        mm.collapsed_gibbs_sampler(1, None, num_saved=0)
        o.sweep(np.array([host.random() for _ in range(N)]))
        bad = np.nonzero(mm.components.assignments != o.z)[0]
        assert bad.size == 0, "sweep %d: %d labels differ, first at i=%d" % (it, bad.size, bad[0])
        lo = o.log_marg()
        assert abs(mm.log_marg() - lo) <= 1e-9 * abs(lo)
    assert random.getstate() == host.getstate()
    if init == "one-by-one":
        assert mm.components.K > k_start, "the case is meant to outgrow the slots the context started with"
        assert mm.components.K_max > k_start
    else:
        assert k_start == N
    # an explicit K_max stays a hard limit, and the message names it
    from pybgmm_amd import _lib
    random.seed(11)
    mm2 = CRPMM(X[:600], NIW(m_0, k_0, v_0, S_0), 1e300, None, assignments="one-by-one", K_max=64)
    with pytest.raises(_lib.BGMMError) as ei:
        mm2.collapsed_gibbs_sampler(1, None, num_saved=0)
    assert ei.value.code == -3 and "K_max" in str(ei.value)


def test_k_max_none_grows_on_demand_for_chains_side_by_side_too():
    """the grow-on-demand of ``K_max=None`` only wrapped the solo sweep; chains.run_chains_on_device (lockstep
    rounds through bgmm_group_sweep_staged) still raised BGMM_EKMAX at max(1024, 4 K_init) slots.  Three "one-by-one"
    chains with alpha = 1e300 -- every visit opens a component, 3 000 of them against 1 024 slots -- side by side: each
    chain equals the solo run under its seeds (seed + c) and the C oracle with K_max = N."""
    import random
    from oracle import c_oracle
    from pybgmm_amd import chains
    from pybgmm_amd.igmm import CRPMM
    from pybgmm_amd.prior import NIW
    from pybgmm_amd.utils import gendata
    N, D, alpha, G = 3000, 5, 1e300, 3
    X, _ = gendata.synth_mixture(N, D, 6, seed=9)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    out = chains.run_chains_on_device(CRPMM, X, NIW(m_0, k_0, v_0, S_0), alpha, G, 2, seed=21, assignments="one-by-one",
                                      sampler_kwargs={})
    for c, (mm, rec) in enumerate(out):
        assert mm.components.K_max_auto and mm.components.K_max > 1024
        host = random.Random(21 + c)
        o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, alpha, -np.ones(N, dtype=np.int64), N, scipy_tables=False)
        for it in range(2):
            o.sweep(np.array([host.random() for _ in range(N)]))
        bad = np.nonzero(mm.components.assignments != o.z)[0]
        assert bad.size == 0, "chain %d: %d labels differ, first at i=%d" % (c, bad.size, bad[0])
        lo = o.log_marg()
        assert abs(mm.log_marg() - lo) <= 1e-9 * abs(lo)


def _case_c5_device_generators():
    import random
    from pybgmm_amd.utils import gendata, rng as _rng
    N, D, K = 10000, 128, 20
    X, zt = gendata.synth_mixture(N, D, K, seed=59)
    z0 = zt.copy()
    rs = np.random.RandomState(4)
    idx = rs.choice(N, size=300, replace=False)
    z0[idx] = rs.randint(0, K, size=300)
    host_r, host_np = random.Random(17), np.random.RandomState(17)
    sweeps = []
    for it in range(2):
        order = host_np.permutation(N)
        u = _rng.take_uniforms(N, host_r)
        sweeps.append((u, order, 1.01 if it else None))
    return {"X": X, "prior": gendata.demo_prior_params(D), "z0": z0, "K_max": 4 * K, "sweeps": sweeps,
            "host_r": host_r, "host_np": host_np}


_case_c5_device_generators.cost = lambda: 2 * 10000 * 20 * 128 * 128


@with_oracle(_case_c5_device_generators)
def test_c5_shape_pcrp_sweeps_fed_by_the_device_generators_against_the_oracle(oracle_ref):
    """C5's shape (PCRPMM, D = 128, full covariance) driven the way the classes drive it -- every sweep's
    visiting order drawn ON THE DEVICE from a numpy RandomState (bgmm_stage_permutation_mt19937) and its uniforms from a
    random.Random (bgmm_stage_mt19937) -- against numpy / random themselves value for value, and label for label against
    the C oracle fed from twin generators on the host; powered weights from the second sweep on (pcrpmm.py:98-117).
    Wrong labels sprinkled in so that the sweeps move (the D = 128 mover paths behind a device-drawn permutation)."""
    import random
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import rng as _rng
    case, ref = oracle_ref
    N, K = case["X"].shape[0], 20
    m_0, k_0, v_0, S_0 = case["prior"]
    dev_r, dev_np = random.Random(17), np.random.RandomState(17)
    ctx = _lib.Context(case["X"], m_0, k_0, v_0, S_0, 1.0, 4 * K)
    ctx.set_assignments(case["z0"])
    moved = 0
    for it in range(2):
        u, order, power = case["sweeps"][it]
        assert _rng.take_permutation_staged(ctx, N, dev_np) is _rng.STAGED
        npt.assert_array_equal(ctx.staged_order(), order, err_msg="device permutation differs from numpy's")
        assert _rng.stage_uniforms_on_device(ctx, None, dev_r)
        npt.assert_array_equal(ctx.staged_uniforms(), u, err_msg="device uniforms differ from random.random()")
        ctx.sweep_staged(power)
        moved += ctx.sweep_stats()["moves"]
        bad = np.nonzero(ctx.assignments() != ref[it]["z"])[0]
        assert bad.size == 0, "sweep %d: %d labels differ, first at i=%d" % (it, bad.size, bad[0])
        lo = ref[it]["log_marg"]
        assert abs(ctx.log_marg() - lo) <= 1e-9 * abs(lo)
    assert moved >= 150, "the case is meant to repair its wrong labels"
    assert dev_r.getstate() == case["host_r"].getstate()
    npt.assert_array_equal(dev_np.get_state()[1], case["host_np"].get_state()[1])
    assert dev_np.get_state()[2] == case["host_np"].get_state()[2]
    ctx.close()


def test_contexts_over_one_data_set_share_its_device_copy():
    """VERDICT r5 #2 (second half): bgmm_create_shared -- chains side by side hold X once.  (i) a context made over its
    parent's copy runs the same chain as one with a copy of its own (two sweeps, labels and log marginal equal), with its own
    prior and K_max; (ii) the copy outlives the parent (destroyed first) for as long as a child uses it; (iii) four
    contexts over a 102-MB X take little more device memory than one with its working buffers; (iv) ChainGroup and
    run_chains_on_device build their chains that way."""
    import ctypes
    from pybgmm_amd import _lib, chains
    from pybgmm_amd.igmm import CRPMM
    from pybgmm_amd.prior import NIW
    from pybgmm_amd.utils import gendata
    N, D, K = 200000, 64, 12
    X, zt = gendata.synth_mixture(N, D, K, seed=3)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    us = np.random.RandomState(1).random_sample((2, N))
    z0 = zt.copy(); z0[::97] = (z0[::97] + 1) % K
    _lib.load()
    # the HIP runtime the library itself is linked against -- by the path it is mapped from: a process that has imported
    # torch holds torch's own copy of libamdhip64 as well, and that one has no device open ("no device", error 100)
    with open("/proc/self/maps") as f:
        paths = [ln.split()[-1] for ln in f if "libamdhip64" in ln]
    mine = [q for q in paths if "/torch/" not in q]
    hip = ctypes.CDLL(mine[0] if mine else "libamdhip64.so")

    def free_bytes():
        f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
        return int(f.value)
    warm = _lib.Context(X[:1000], m_0, k_0, v_0, S_0, 1.0, 4 * K)      # (the runtime's own first allocations out of the way)
    warm.synchronize()
    free0 = free_bytes()
    parent = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
    parent.synchronize()
    free1 = free_bytes()
    kids = [_lib.Context(X, m_0, 2.0 * k_0, v_0 + 3, S_0, 0.7, 3 * K, share_with=parent) for _ in range(3)]
    own = _lib.Context(X, m_0, 2.0 * k_0, v_0 + 3, S_0, 0.7, 3 * K)
    for c in kids + [own]:
        c.synchronize()
    free2 = free_bytes()
    warm.close()
    x_bytes = X.nbytes
    per_own = free0 - free1                              # a context with its own copy
    assert per_own > x_bytes
    used_by_four = free1 - free2                         # three borrowers + one with its own copy
    assert used_by_four < 4 * per_own - 2.5 * x_bytes, (per_own, used_by_four, x_bytes)
    for c in (kids[0], own):
        c.set_assignments(z0)
    parent.close()                                       # (ii): the children keep the copy alive
    for it in range(2):
        kids[0].sweep(us[it]); own.sweep(us[it])
        npt.assert_array_equal(kids[0].assignments(), own.assignments())
        assert kids[0].log_marg() == own.log_marg()
    assert kids[0].sweep_stats()["moves"] + 1 > 0
    for c in kids + [own]:
        c.close()
    # (iv)
    grp = chains.ChainGroup(X[:6000], m_0, k_0, v_0, S_0, 1.0, 4 * K, n_chains=3, seed=5)
    assert all(c.h for c in grp.ctxs)
    grp.set_assignments([zt[:6000]] * 3)
    grp.sweep()
    grp.close()
    out = chains.run_chains_on_device(CRPMM, X[:6000], NIW(m_0, k_0, v_0, S_0), 1.0, 3, 2, seed=2, assignments="rand", K=K)
    assert len(out) == 3 and all(m.components.K >= 1 for m, _ in out)


def test_thirty_two_chains_burn_in_side_by_side_and_equal_their_solo_runs():
    """G = 32 chains of one shape from the reference's "rand" start in ONE group call -- their frozen-factor
    windows shared in two sub-groups of launches on two streams (api_group.hip: gram_group_launch), a host thread each --
    every chain label for label its solo run, two sweeps."""
    import random
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    N, D, K, G = 6000, 64, 10, 32
    X, zt = gendata.synth_mixture(N, D, K, seed=21)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)

    def build(c):
        ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 8 * K)
        ctx.set_assignments(np.unique(np.random.RandomState(300 + c).randint(0, K, N), return_inverse=True)[1])
        _, key, _ = random.Random(500 + c).getstate()
        return [ctx, np.asarray(key[:-1], dtype=np.uint32), int(key[-1])]
    grouped = [build(c) for c in range(G)]
    for it in range(2):
        for ch in grouped:
            ch[1], ch[2] = ch[0].stage_mt19937(ch[1], ch[2], None)
        _lib.group_sweep_staged([ch[0] for ch in grouped], [None] * G)
    zs = [ch[0].assignments() for ch in grouped]
    lms = [ch[0].log_marg() for ch in grouped]
    for ch in grouped:
        ch[0].close()
    assert len({z.tobytes() for z in zs}) == G, "chains with different seeds must differ"
    for c in range(G):
        ch = build(c)
        for it in range(2):
            ch[1], ch[2] = ch[0].stage_mt19937(ch[1], ch[2], None)
            ch[0].sweep_staged(None)
        npt.assert_array_equal(ch[0].assignments(), zs[c], err_msg="chain %d" % c)
        assert abs(ch[0].log_marg() - lms[c]) <= 1e-13 * abs(lms[c])      # (pipelined vs shared plain windows: the last digit)
        ch[0].close()


def test_dimension_limits_are_errors_that_say_so():
    """diag / fixed components take any D up to 4096 (their state is a D-vector: goldens
    diag_crpmm_256d / fixed_pcrp_256d, the oracle cases at D = 200 / 300); full covariance takes D <= 256 since round 6 (up to
    128 through the fast kernels, beyond through the general route: test_full_covariance_beyond_128_dimensions) and says so at
    construction beyond that (BGMM_EUNSUPPORTED), where the reference has no limit (gaussian_components.py:86-90)."""
    from pybgmm_amd import _lib
    D = 260
    X = np.random.RandomState(0).randn(300, D)
    with pytest.raises(_lib.BGMMError) as ei:
        _lib.Context(X, np.zeros(D), 0.03, D + 3, np.eye(D), 1.0, 16)
    assert ei.value.code == -5 and "256" in str(ei.value) and "diag" in str(ei.value)
    ctx = _lib.Context(X, np.zeros(D), 0.03, D + 3, np.ones(D), 1.0, 16, cov_type="diag")
    ctx.set_assignments(np.zeros(300, dtype=np.int64))
    ctx.sweep(np.random.RandomState(1).random_sample(300))
    assert ctx.counts().sum() == 300
    ctx.close()


@pytest.mark.parametrize("N,D,K,init,pcrp", [(1500, 160, 5, "rand", False), (1200, 192, 4, "true", True), (900, 256, 3, "rand", True),
                                             (400, 129, 3, "one-by-one", False)],
                         ids=["D160-rand", "D192-pcrp-flipped", "D256-pcrp-rand", "D129-one-by-one"])
def test_full_covariance_beyond_128_dimensions(N, D, K, init, pcrp):
    """VERDICT r5 missing #2 / next #10: the reference has no limit on D (gaussian_components.py:86-90, 319-331); here a
    component's factor had to fit LDS, D <= 128.  Round 6: 129 .. 256 take the general route -- the VALU likelihood kernel over
    every label, the per-mover kernel chain, rebuilds and rank-1 steps in a workspace in global memory.  Three sweeps against
    the C oracle from the reference's kinds of start (components deleted and opened on the way), labels, counts, log marginal;
    then the statistics and the method-level calls the classes use (raw statistics bit-equal to the oracle's, the predictive
    of a point under every component, del_item / add_item round trip)."""
    from oracle import c_oracle
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    X, zt = gendata.synth_mixture(N, D, K, seed=D, mu_scale=1.2)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    rs = np.random.RandomState(D + 1)
    if init == "rand":
        z0 = np.unique(rs.randint(0, 3 * K, N), return_inverse=True)[1]
    elif init == "one-by-one":
        z0 = -np.ones(N, dtype=np.int64)
    else:
        z0 = zt.copy()
        idx = rs.choice(N, size=N // 10, replace=False)
        z0[idx] = rs.randint(0, K, size=idx.size)
    K_max = 8 * K
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, K_max)
    ctx.set_assignments(z0)
    o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, z0, K_max, scipy_tables=False)
    moved = 0
    for it in range(3):
        u = rs.random_sample(N)
        order = rs.permutation(N).astype(np.int64) if pcrp else None
        power = 1.01 if (pcrp and it > 0) else None
        ctx.sweep(u, order, power)
        o.sweep(u, order, power)
        bad = np.nonzero(ctx.assignments() != o.z)[0]
        assert bad.size == 0, "sweep %d: %d labels differ, first at i=%d" % (it, bad.size, bad[0])
        npt.assert_array_equal(ctx.counts(), o.counts)
        lo = o.log_marg()
        assert abs(ctx.log_marg() - lo) <= 1e-9 * abs(lo)
        moved += ctx.sweep_stats()["moves"]
    assert moved > 0
    m, S, ld, iv = o.stats()
    for k in range(o.K):
        mk, Sk = ctx.raw_stats(k)
        npt.assert_array_equal(mk, m[k])
        npt.assert_array_equal(Sk, S[k])
    i = int(np.nonzero(o.z >= 0)[0][7])
    npt.assert_allclose(ctx.log_post_pred(i), o.log_post_pred(i), rtol=1e-9, atol=1e-9)
    k_i = int(o.z[i])
    ctx.del_item(i); ctx.add_item(i, k_i)
    npt.assert_array_equal(ctx.assignments(), o.z)
    assert abs(ctx.log_marg() - lo) <= 1e-9 * abs(lo)
    ctx.close()

"""The C oracle against the reference's golden vectors and captured trajectories."""
import numpy as np
import numpy.testing as npt
import pytest

from golden_util import ALL_CASES, DIAG_CASES, FIXED_CASES, Golden
from oracle import c_oracle


@pytest.mark.parametrize("case", ALL_CASES + DIAG_CASES + FIXED_CASES)
def test_trajectory(case):
    g = Golden(case)
    o, out = c_oracle.run_chain(g)
    for it in range(g.n_iter):
        npt.assert_array_equal(out["z"][it], g.z[it], err_msg="sweep %d" % it)
        assert out["K"][it] == g.K[it]
        npt.assert_array_equal(out["counts"][it], g.counts_at(it))
        assert abs(out["log_marg"][it] - g.log_marg[it]) <= 1e-9 * abs(g.log_marg[it])
    if "final_S" in g.d.files:
        m, S, ld, iv = o.stats()
        # identical trajectory + unfused accumulation => identical sufficient statistics
        npt.assert_array_equal(m, g.d["final_m"])
        npt.assert_array_equal(S, g.d["final_S"])
        npt.assert_allclose(ld, g.d["final_logdet"], rtol=1e-9, atol=1e-9)
        npt.assert_allclose(iv, g.d["final_inv"], rtol=1e-7, atol=1e-10)
    npt.assert_allclose(o.log_prior[:4096], g.d["cached_log_prior"], rtol=1e-12)


def test_libm_tables_give_same_trajectory():
    g = Golden("kat1_igmm_2d")
    _, out = c_oracle.run_chain(g, scipy_tables=False)
    npt.assert_array_equal(out["z"][-1], g.z[-1])


def test_component_kats():
    # pybgmm/tests/test_gaussian_components.py:33,104 (expected values)
    X = np.array([[-0.3406, -0.0593, -0.0686]])
    o = c_oracle.COracle(X, np.zeros(3), 0.05, 4, 0.001 * np.eye(3), 1.0, [-1], 1)
    npt.assert_almost_equal(o.log_prior[0], -0.472067277015)
    X = np.array([[1.2, 0.9], [-0.1, 0.8], [0.5, 0.4]])
    o = c_oracle.COracle(X, np.zeros(2), 2., 5, 5. * np.eye(2), 1.0, [0, 0, -1], 3)
    npt.assert_almost_equal(o.log_post_pred(2)[0], -2.07325364088)


def test_probe_visit_reproduces_the_references_first_visit_probabilities():
    """go_probe_visit (the first-divergence diagnostic's oracle side, tests/divergence.py) against prob_z of the very
    first visit as the reference computed it (the fixtures' probes), and the diagnostic's CDF / draw helpers against
    the label the reference drew there."""
    from divergence import _cdf, _draw
    for case in ("kat1_igmm_2d", "c3twin_pcrpmm_16d", "one_by_one_50", "each_in_own_50"):
        g = Golden(case)
        p_ref, u, k = g.probes()[0]
        i = 0 if g.sweep_order(0) is None else int(g.sweep_order(0)[0])
        o = c_oracle.COracle(g.X, g.m_0, g.k_0, g.v_0, g.S_0, g.alpha, g.z_init, g.K_max, cov_type=g.cov_type)
        lp = o.probe_visit(i, g.sweep_power(0))
        p, _ = _cdf(lp)
        npt.assert_allclose(p, p_ref, rtol=1e-10, atol=1e-14)
        assert _draw(p, u) == k


@pytest.mark.parametrize("case", ["c4twin_crpmm_64d", "c3rand_pcrpmm_16d", "c4rand_crpmm_64d", "diag_crpmm_256d"])
def test_threads_do_not_change_a_single_float(case):
    """Round 6: a visit's K evaluations (and the D columns of the inverse) shared out over OpenMP threads.  Every
    component is still scored by the scalar code on one thread, the maximum / log-sum-exp / u -= p scan stay serial in
    label order (gaussian_components.py:228-251, utils.py:15-20): the captured trajectories are reproduced with the
    same floats -- log marginals and final statistics compared with == between 1 and several threads."""
    from golden_util import ALL_CASES, DIAG_CASES, FIXED_CASES
    if case not in ALL_CASES + DIAG_CASES + FIXED_CASES:
        pytest.skip("no such fixture")
    g = Golden(case)
    before = c_oracle.get_threads()
    try:
        runs = []
        for t in (1, 3, 8):
            assert c_oracle.set_threads(t) == t
            o, out = c_oracle.run_chain(g)
            runs.append((out, o.stats()))
        for it in range(g.n_iter):
            npt.assert_array_equal(runs[0][0]["z"][it], g.z[it])
        for out, st in runs[1:]:
            for it in range(g.n_iter):
                npt.assert_array_equal(out["z"][it], runs[0][0]["z"][it])
                assert out["log_marg"][it] == runs[0][0]["log_marg"][it]
            for a, b in zip(st, runs[0][1]):
                npt.assert_array_equal(a, b)
    finally:
        c_oracle.set_threads(before)

"""The numpy oracle against the reference: known-answer values of the reference's
own tests and trajectories captured from the reference (tests/golden)."""
import numpy as np
import numpy.testing as npt
import pytest

from golden_util import DIAG_CASES, FIXED_CASES, SMALL_CASES, Golden
from oracle.gibbs_numpy import NumpyGibbsOracle, run_chain


@pytest.mark.parametrize("case", SMALL_CASES)
def test_trajectory_bitexact(case):
    g = Golden(case)
    o, out = run_chain(g.X, g.prior, g.alpha, g.z_init, g.K_max, g.u, g.order,
                       g.n_power, g.power_burnin, g.flag_power)
    for it in range(g.n_iter):
        npt.assert_array_equal(out["z"][it], g.z[it], err_msg="sweep %d" % it)
        assert out["K"][it] == g.K[it]
        npt.assert_array_equal(out["counts"][it], g.counts_at(it))
        # same numpy calls on the same operands: bit-identical expected
        assert out["log_marg"][it] == g.log_marg[it]
    if "final_S" in g.d.files:
        K = g.K[-1]
        npt.assert_array_equal(o.m[:K], g.d["final_m"])
        npt.assert_array_equal(o.S[:K], g.d["final_S"])
        npt.assert_array_equal(o.logdet[:K], g.d["final_logdet"])
        npt.assert_array_equal(o.inv[:K], g.d["final_inv"])
    npt.assert_array_equal(o.log_prior[:4096], g.d["cached_log_prior"])


@pytest.mark.parametrize("case", [c for c in DIAG_CASES if c != "diag_crpmm_64d"])
def test_diag_trajectory_bitexact(case):
    g = Golden(case)
    o, out = run_chain(g.X, g.prior, g.alpha, g.z_init, g.K_max, g.u, g.order,
                       g.n_power, g.power_burnin, g.flag_power, cov_type="diag")
    for it in range(g.n_iter):
        npt.assert_array_equal(out["z"][it], g.z[it], err_msg="sweep %d" % it)
        npt.assert_array_equal(out["counts"][it], g.counts_at(it))
        assert out["log_marg"][it] == g.log_marg[it]
    K = g.K[-1]
    npt.assert_array_equal(o.m[:K], g.d["final_m"])
    npt.assert_array_equal(o.S[:K], g.d["final_S"])
    npt.assert_array_equal(o.logdet[:K], g.d["final_logdet"])
    npt.assert_array_equal(o.inv[:K], g.d["final_inv"])
    npt.assert_array_equal(o.log_prior[:4096], g.d["cached_log_prior"])


@pytest.mark.parametrize("case", FIXED_CASES)
def test_fixed_trajectory_bitexact(case):
    g = Golden(case)
    o, out = run_chain(g.X, g.prior, g.alpha, g.z_init, g.K_max, g.u, g.order,
                       g.n_power, g.power_burnin, g.flag_power, cov_type="fixed")
    for it in range(g.n_iter):
        npt.assert_array_equal(out["z"][it], g.z[it], err_msg="sweep %d" % it)
        npt.assert_array_equal(out["counts"][it], g.counts_at(it))
        assert out["log_marg"][it] == g.log_marg[it]
    K = g.K[-1]
    npt.assert_array_equal(o.m[:K], g.d["final_m"])
    npt.assert_array_equal(o.S[:K], g.d["final_S"])
    npt.assert_array_equal(o.logdet[:K], g.d["final_logdet"])
    npt.assert_array_equal(o.inv[:K], g.d["final_inv"])
    npt.assert_array_equal(o.log_prior[:4096], g.d["cached_log_prior"])


def test_diag_matches_univariate_student_t():
    """The analytic cross-check of pybgmm/tests/test_gaussian_components_diag.py:17-72: the
    predictive is a product of univariate Student-t densities (Murphy, bayesGauss p. 26)."""
    from scipy.special import gammaln
    from oracle.gibbs_numpy import NumpyGibbsOracleDiag

    def t_logpdf(x, mu, var, v):
        c = gammaln((v + 1) / 2.) - gammaln(v / 2.) - 0.5 * (np.log(v) + np.log(np.pi) + np.log(var))
        return c - (v + 1) / 2. * np.log(1 + 1. / v * (x - mu) ** 2 / var)

    rs = np.random.RandomState(1)
    X = 5 * rs.rand(10, 3) - 1
    m_0, k_0, v_0, S_0 = 5 * rs.rand(3) - 2, rs.randint(15), 4, 2 * rs.rand(3) + 3
    o = NumpyGibbsOracleDiag(X, m_0, k_0 + 1, v_0, S_0, 1.0, np.zeros(10, dtype=np.int64), None)
    k0 = k_0 + 1
    var0 = S_0 * (k0 + 1) / (k0 * v_0)
    npt.assert_almost_equal(o.log_prior[0], np.sum(t_logpdf(X[0], m_0, var0, v_0)))
    k_N, v_N = k0 + 10, v_0 + 10
    m_N = (k0 * m_0 + X.sum(axis=0)) / k_N
    S_N = S_0 + k0 * m_0 ** 2 + (X ** 2).sum(axis=0) - k_N * m_N ** 2
    var = S_N * (k_N + 1) / (k_N * v_N)
    npt.assert_almost_equal(o.predictive_k(0, 0), np.sum(t_logpdf(X[0], m_N, var, v_N)))
    npt.assert_almost_equal(o.predictive_all(0)[0], o.predictive_k(0, 0))


def test_probes_first_visits():
    g = Golden("kat1_igmm_2d")
    o = NumpyGibbsOracle(g.X, g.m_0, g.k_0, g.v_0, g.S_0, g.alpha, g.z_init, g.K_max)
    probe = []
    o.sweep(g.u[0], None, None, probe)
    for (p_ref, u_ref, k_ref), (p, u, k) in zip(g.probes(), probe):
        npt.assert_array_equal(p, p_ref)
        assert u == u_ref and k == k_ref


def test_reference_kat_literals():
    # pybgmm/tests/test_igmm.py:53-59,101 / :143 / :183 (values, not code)
    g = Golden("kat1_igmm_2d")
    assert abs(g.log_marg[-1] - (-411.811711231)) < 1e-7
    g = Golden("kat3_each_in_own")
    npt.assert_array_equal(
        g.z[-1], [5, 2, 4, 3, 2, 7, 2, 7, 1, 0, 4, 6, 4, 1, 6, 4, 1, 7, 1, 0])
    g = Golden("kat4_log_marg")
    assert abs(g.log_marg[-1] - (-30.771535771)) < 1e-7


# ---- component-level known answers: pybgmm/tests/test_gaussian_components.py ---- #
def _oracle(X, m_0, k_0, v_0, S_0, z):
    return NumpyGibbsOracle(np.asarray(X, float), m_0, k_0, v_0, S_0, 1.0, z, None)


def test_log_prior_3d():
    X = [[-0.3406, -0.0593, -0.0686]]
    o = _oracle(X, np.zeros(3), 0.05, 4, 0.001 * np.eye(3), [-1])
    npt.assert_almost_equal(o.log_prior[0], -0.472067277015)


def test_log_marg_k():
    X = [[-0.3406, -0.3593, -0.0686], [-0.3381, 0.2993, 0.925], [-0.5, -0.101, 0.75]]
    o = _oracle(X, np.zeros(3), 0.05, 6, 0.5 * np.eye(3), [0, 0, 0])
    npt.assert_almost_equal(o.log_marg_k(0), -8.42365141729)


def test_log_post_pred_k():
    X = [[1.2, 0.9], [-0.1, 0.8], [0.5, 0.4]]
    o = _oracle(X, np.zeros(2), 2., 5, 5. * np.eye(2), [0, 0, -1])
    npt.assert_almost_equal(o.predictive_k(2, 0), -2.07325364088)


def test_vectorised_equals_per_component():
    rs = np.random.RandomState(2)
    X = rs.rand(11, 4)
    o = _oracle(X, X.mean(axis=0), 0.05, 14, 0.5 * np.eye(4),
                [0, 0, 0, 1, 0, 1, 3, 4, 3, 2, -1])
    npt.assert_almost_equal([o.predictive_k(10, k) for k in range(o.K)], o.predictive_all(10))

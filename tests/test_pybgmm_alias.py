"""
The reference's import path: ``from pybgmm.igmm import CRPMM, PCRPMM``, ``from pybgmm.prior import NIW``
(reference pybgmm/igmm/__init__.py:7-8, pybgmm/prior/__init__.py:2) served by pybgmm_amd -- and a reference example
script's body run unchanged under that name on the GPU.
"""
import os
import sys

import numpy as np
import numpy.testing as npt
import pytest

from golden_util import GOLDEN_DIR


def test_pybgmm_modules_are_the_pybgmm_amd_modules():
    import pybgmm
    import pybgmm_amd.gaussian
    import pybgmm_amd.igmm
    import pybgmm_amd.prior
    from pybgmm.igmm import ADAPCRPMM, CRPMM, PCRPMM
    from pybgmm.prior import NIW
    from pybgmm.gaussian.gaussian_components import GaussianComponents
    from pybgmm.utils import gendata_1d
    assert pybgmm.igmm is pybgmm_amd.igmm and sys.modules["pybgmm.igmm.crpmm"] is sys.modules["pybgmm_amd.igmm.crpmm"]
    assert CRPMM is pybgmm_amd.igmm.CRPMM and PCRPMM is pybgmm_amd.igmm.PCRPMM and ADAPCRPMM is pybgmm_amd.igmm.ADAPCRPMM
    assert NIW is pybgmm_amd.prior.NIW
    assert GaussianComponents is pybgmm_amd.gaussian.GaussianComponents
    # not re-export files: the package directory holds one __init__ and nothing else
    files = [f for f in os.listdir(os.path.dirname(pybgmm.__file__)) if f.endswith(".py")]
    assert files == ["__init__.py"], files
    # the out-of-scope samplers are absent, not stubbed
    with pytest.raises(ImportError):
        from pybgmm.igmm import CSCRPMM  # noqa: F401
    mu, X, y = gendata_1d(50)
    assert X.shape == (50, 1) and y.shape == (50,)


@pytest.mark.gpu
def test_reference_2d_demo_recipe_runs_unchanged_under_import_pybgmm():
    """The body of examples/crpmm_2d_demo.py:25-84 (the recipe written out here, plotting left out) with the reference's
    import lines: global seeds, data and prior drawn from the global streams, CRPMM from "rand" with K = 3, 40 sweeps
    with the default num_saved (K == 3 on several sweeps: distribution-dict snapshots, Dirichlet draws from np.random),
    rand_k of every component (the demo's ellipses), and what both global streams deliver afterwards -- against the run
    of the reference itself captured by tests/golden/make_golden.py (case demo2d)."""
    import random
    from pybgmm.igmm import CRPMM
    from pybgmm.prior import NIW
    g = np.load(os.path.join(GOLDEN_DIR, "demo_crpmm_2d.npz"), allow_pickle=False)

    random.seed(1)
    np.random.seed(1)
    D, N, K_true = 2, 100, 4
    alpha, K, n_iter = 1., 3, 40
    mu_scale, covar_scale = 4.0, 0.7
    z_true = np.random.randint(0, K_true, N)
    mu = np.random.randn(D, K_true) * mu_scale
    X = mu[:, z_true] + np.random.randn(D, N) * covar_scale
    X = X.T
    npt.assert_array_equal(X, g["X"])
    m_0 = np.zeros(D)
    k_0 = covar_scale ** 2 / mu_scale ** 2
    v_0 = D + 3
    S_0 = covar_scale ** 2 * v_0 * np.eye(D)
    prior = NIW(m_0, k_0, v_0, S_0)
    crpmm = CRPMM(X, prior, alpha, save_path=None, assignments="rand", K=K)
    record_dict, _dist = crpmm.collapsed_gibbs_sampler(n_iter, z_true)

    npt.assert_array_equal(np.array(record_dict["components"]), g["rec_components"])
    npt.assert_allclose(np.array(record_dict["log_marg"]), g["rec_log_marg"], rtol=1e-9)
    npt.assert_allclose(np.array(record_dict["nmi"]), g["rec_nmi"], rtol=1e-9, atol=1e-12)
    assert [str(s) for s in record_dict["nk"]] == [str(s) for s in g["rec_nk"]]
    npt.assert_array_equal(crpmm.components.assignments, g["final_z"])
    assert crpmm.components.K == int(g["final_K"])
    for k in range(crpmm.components.K):
        mu_k, sigma_k = crpmm.components.rand_k(k)
        npt.assert_allclose(np.ravel(mu_k), g["rand_k_mu"][k], rtol=1e-8)
        npt.assert_allclose(np.ravel(sigma_k), g["rand_k_sigma"][k], rtol=1e-8)
    npt.assert_array_equal(np.array([random.random() for _ in range(4)]), g["after_random"])
    npt.assert_array_equal(np.random.random_sample(4), g["after_numpy"])

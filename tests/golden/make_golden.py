#!/usr/bin/env python3
"""
Generate the golden fixtures in this directory by RUNNING THE REFERENCE.

This script only works in the build container, where the read-only reference
checkout lives at /root/reference.  It never copies reference source into the
repo: it converts a scratch copy under a temp dir with lib2to3 (the reference
is Python 2), applies a small runtime shim for removed numpy/scipy aliases,
imports it from there, runs each case, and stores INPUTS + EXPECTED OUTPUTS as
.npz data.  The fixtures travel to the GPU box; the reference does not.

What is captured per case (all from one single ``collapsed_gibbs_sampler``
call, hooks installed from the outside, the reference code itself untouched):
  * inputs:  X (or recipe + sha256 when large), prior, alpha, initial labels,
             seeds, sampler kind and its kwargs, K_max
  * streams: every ``random.random()`` value consumed by ``utils.draw`` (``u``),
             every ``np.random.permutation`` result (``order``)
  * per sweep (hook on ``update_record_dict``): labels z, K, counts,
             log_marg, and the record_dict metric values
  * first visits: (prob_z, u, k) of the first ``n_probe`` draws
  * final stats for small cases: m_N_numerators, S_N_partials, logdet_covars,
             inv_covars, cached_log_prior

Usage:  python tests/golden/make_golden.py [case ...]
"""
import os
import random
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REFERENCE = "/root/reference"
sys.path.insert(0, REPO)

from pybgmm_amd.utils import gendata  # noqa: E402  (our own recipes)


# --------------------------------------------------------------------------- #
# scratch import of the reference                                             #
# --------------------------------------------------------------------------- #
def import_reference():
    if not os.path.isdir(REFERENCE):
        raise SystemExit("reference checkout not present; fixtures can only be "
                         "regenerated in the build container")
    scratch = tempfile.mkdtemp(prefix="pybgmm_ref_py3_")
    shutil.copytree(os.path.join(REFERENCE, "pybgmm"), os.path.join(scratch, "pybgmm"))
    subprocess.run([sys.executable, "-m", "lib2to3", "-w", "-n", "pybgmm"],
                   cwd=scratch, check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    import scipy.misc
    import scipy.special
    scipy.misc.logsumexp = scipy.special.logsumexp
    np.float = float
    np.int = int
    sys.path.insert(0, scratch)
    import matplotlib
    matplotlib.use("Agg")
    import pybgmm.igmm  # noqa: F401
    import pybgmm.prior  # noqa: F401
    return scratch


class _RandomTap(object):
    """Stand-in for the ``random`` module inside pybgmm.utils.utils: forwards to
    the real global stream and logs every uniform."""

    def __init__(self):
        self.log = []

    def random(self):
        u = random.random()
        self.log.append(u)
        return u

    def __getattr__(self, name):
        return getattr(random, name)


def run_case(name, model, X, prior_params, alpha, assignments, K, K_max, n_iter,
             seeds, sampler_kwargs=None, true_assignments=None, n_probe=50,
             store_X=True, recipe=None, skip_metrics=False, cov_type="full",
             reseed_before_model=True):
    """
    Build the reference model and run ``n_iter`` sweeps with taps installed.
    ``seeds=(s_random, s_numpy)`` are applied with random.seed / np.random.seed
    right before model construction unless ``reseed_before_model`` is False (KAT
    cases seed before the data is drawn and keep the stream running).
    """
    import pybgmm.gmm.gmm as ref_gmm
    import pybgmm.utils.utils as ref_utils
    from pybgmm.igmm import ADAPCRPMM, CRPMM, PCRPMM
    from pybgmm.prior import NIW

    sampler_kwargs = dict(sampler_kwargs or {})
    if reseed_before_model:
        random.seed(seeds[0])
        np.random.seed(seeds[1])

    if cov_type == "fixed":
        from pybgmm.gaussian.gaussian_components_fixedvar import FixedVarPrior
        fv_var, fv_mu_0, fv_var_0 = prior_params
        prior = FixedVarPrior(fv_var, fv_mu_0, fv_var_0)
        # stored in the fixture through the NIW slots: m_0 = mu_0, S_0 = [var ; var_0], k_0 / v_0 unused
        m_0, k_0, v_0, S_0 = fv_mu_0, 1.0, 1, np.concatenate([fv_var, fv_var_0])
    else:
        m_0, k_0, v_0, S_0 = prior_params
        prior = NIW(m_0, k_0, v_0, S_0)
    cls = {"CRPMM": CRPMM, "PCRPMM": PCRPMM, "ADAPCRPMM": ADAPCRPMM}[model]
    init = assignments if isinstance(assignments, str) else list(assignments)
    mm = cls(X, prior, alpha, None, assignments=init, K=K, K_max=K_max,
             covariance_type=cov_type)
    comp = mm.components
    z_init = np.array(comp.assignments, dtype=np.int64)
    K_init = comp.K

    # --- taps -------------------------------------------------------------- #
    tap = _RandomTap()
    ref_utils.random = tap
    orders = []
    real_perm = np.random.permutation

    def perm_tap(x):
        out = real_perm(x)
        orders.append(np.array(out, dtype=np.int64))
        return out

    np.random.permutation = perm_tap

    probes = []
    real_draw = ref_utils.draw

    def draw_tap(p_k):
        n0 = len(tap.log)
        k = real_draw(p_k)
        if len(probes) < n_probe:
            probes.append((np.array(p_k, dtype=np.float64), tap.log[n0], k))
        return k

    ref_utils.draw = draw_tap

    snaps = {"z": [], "K": [], "counts": [], "log_marg": []}
    real_update = ref_gmm.GMM.update_record_dict

    saved_metric_fns = {}
    if skip_metrics:
        for fn in ("normalized_mutual_information", "mutual_information",
                   "information_variation"):
            saved_metric_fns[fn] = getattr(ref_gmm, fn)
            setattr(ref_gmm, fn, lambda *a, **k: float("nan"))
        saved_metric_fns["cluster_loss_inertia"] = ref_gmm.utils.cluster_loss_inertia
        ref_gmm.utils.cluster_loss_inertia = lambda *a, **k: float("nan")

    def update_tap(self, record_dict, i_iter, true_assignments, start_time):
        out = real_update(self, record_dict, i_iter, true_assignments, start_time)
        c = self.components
        snaps["z"].append(np.array(c.assignments, dtype=np.int64))
        snaps["K"].append(int(c.K))
        snaps["counts"].append(np.array(c.counts[:c.K], dtype=np.int64))
        snaps["log_marg"].append(float(out["log_marg"][-1]))
        return out

    ref_gmm.GMM.update_record_dict = update_tap
    if true_assignments is None:
        true_assignments = np.zeros(X.shape[0], dtype=np.int64)
    try:
        record, _dist = mm.collapsed_gibbs_sampler(n_iter, true_assignments,
                                                   num_saved=0, **sampler_kwargs)
    finally:
        ref_gmm.GMM.update_record_dict = real_update
        ref_utils.draw = real_draw
        ref_utils.random = random
        np.random.permutation = real_perm
        for fn, f in saved_metric_fns.items():
            if fn == "cluster_loss_inertia":
                ref_gmm.utils.cluster_loss_inertia = f
            else:
                setattr(ref_gmm, fn, f)

    N, D = X.shape
    u = np.array(tap.log, dtype=np.float64).reshape(n_iter, N)
    kmax_seen = max(len(c) for c in snaps["counts"])
    counts = -np.ones((n_iter, kmax_seen), dtype=np.int64)
    for t, c in enumerate(snaps["counts"]):
        counts[t, :len(c)] = c

    out = {
        "case": name, "model": model, "cov_type": cov_type,
        "N": N, "D": D, "alpha": float(alpha), "K_max": int(comp.K_max),
        "n_iter": n_iter,
        "m_0": np.asarray(m_0, dtype=np.float64), "k_0": float(k_0), "v_0": int(v_0),
        "S_0": np.asarray(S_0, dtype=np.float64),
        "seed_random": seeds[0], "seed_numpy": seeds[1],
        "init": init if isinstance(init, str) else "vector", "K_arg": K,
        "z_init": z_init, "K_init": K_init,
        "true_assignments": np.asarray(true_assignments, dtype=np.int64),
        "n_power": float(sampler_kwargs.get("n_power", 1.01 if model == "PCRPMM" else 1.0)),
        "power_burnin": int(sampler_kwargs.get("power_burnin", 0)),
        "flag_power": bool(sampler_kwargs.get("flag_power", model == "PCRPMM")),
        "adap_r_up": float(sampler_kwargs.get("r_up", 1.3)),
        "adap_perct": float(sampler_kwargs.get("adapcrp_perct", 0.04)),
        "adap_burnin": int(sampler_kwargs.get("adapcrp_burnin", 0)),
        "adap_flag": bool(sampler_kwargs.get("flag_adapcrp", model == "ADAPCRPMM")),
        "u": u,
        "order": (np.stack(orders) if orders else np.zeros((0, N), dtype=np.int64)),
        "z": np.stack(snaps["z"]), "K": np.array(snaps["K"], dtype=np.int64),
        "counts": counts, "log_marg": np.array(snaps["log_marg"]),
        "probe_u": np.array([p[1] for p in probes]),
        "probe_k": np.array([p[2] for p in probes], dtype=np.int64),
        "probe_len": np.array([len(p[0]) for p in probes], dtype=np.int64),
        "probe_prob": np.concatenate([p[0] for p in probes]) if probes else np.zeros(0),
        "cached_log_prior": np.array(comp.cached_log_prior[:min(N, 4096)]),
        "X_sha256": gendata.array_digest(X),
        "recipe": recipe or "",
    }
    if not skip_metrics:
        for key in ("nmi", "mi", "vi", "loss", "bic"):
            out["rec_" + key] = np.array(record[key], dtype=np.float64)
        out["rec_nk"] = np.array(record["nk"])
    if store_X:
        out["X"] = np.asarray(X, dtype=np.float64)
    if cov_type == "fixed":
        Kf = comp.K
        out["final_m"] = np.array(comp.mu_N_numerators[:Kf])
        out["final_S"] = np.array(comp.precision_Ns[:Kf])
        out["final_logdet"] = np.array(comp.log_prod_precision_preds[:Kf])
        out["final_inv"] = np.array(comp.precision_preds[:Kf])
    elif cov_type == "diag":
        Kf = comp.K
        out["final_m"] = np.array(comp.m_N_numerators[:Kf])
        out["final_S"] = np.array(comp.S_N_partials[:Kf])
        out["final_logdet"] = np.array(comp.log_prod_vars[:Kf])      # log prod of predictive variances
        out["final_inv"] = np.array(comp.inv_vars[:Kf])
    elif comp.K * D * D <= 40000:
        Kf = comp.K
        out["final_m"] = np.array(comp.m_N_numerators[:Kf])
        out["final_S"] = np.array(comp.S_N_partials[:Kf])
        out["final_logdet"] = np.array(comp.logdet_covars[:Kf])
        out["final_inv"] = np.array(comp.inv_covars[:Kf])
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-18s N=%-5d D=%-3d sweeps=%-3d K: %s  log_marg[-1]=%.12f  moved/sweep~%s  -> %s (%.0f KB)" % (
        name, N, D, n_iter, snaps["K"], snaps["log_marg"][-1],
        [int(np.sum(snaps["z"][t] != (snaps["z"][t - 1] if t else z_init))) for t in range(min(n_iter, 4))],
        os.path.basename(path), os.path.getsize(path) / 1024.0))
    return out


# --------------------------------------------------------------------------- #
# cases                                                                       #
# --------------------------------------------------------------------------- #
def kat_data(seed, D, N, K_true):
    """Data block shared by pybgmm/tests/test_igmm.py:18-38 (seeds set first,
    data drawn from the global streams, streams keep running into the model)."""
    random.seed(seed)
    np.random.seed(seed)
    X, z_true = gendata.demo_mixture(N, D, K_true, rs=np.random)
    return X, z_true


def case_kat1():
    # pybgmm/tests/test_igmm.py:16-59,62-103  (10 sweeps, rand K=3, v_0=5)
    X, z_true = kat_data(1, 2, 100, 4)
    pp = gendata.demo_prior_params(2, v_0=5)
    out = run_case("kat1_igmm_2d", "CRPMM", X, pp, 1.0, "rand", 3, None, 10, (1, 1),
                   true_assignments=z_true, reseed_before_model=False)
    expected = np.array([
        1, 2, 0, 0, 2, 1, 2, 1, 2, 0, 0, 1, 0, 2, 1, 0, 1, 1, 1, 0, 1, 1, 1, 0,
        2, 0, 1, 0, 1, 1, 1, 0, 2, 2, 1, 1, 2, 1, 0, 1, 1, 1, 1, 2, 2, 1, 1, 1,
        1, 0, 0, 1, 0, 0, 1, 2, 2, 1, 1, 0, 1, 2, 2, 1, 1, 1, 1, 2, 0, 0, 1, 2,
        0, 1, 0, 0, 1, 2, 1, 1, 2, 0, 0, 1, 2, 1, 2, 2, 1, 1, 0, 1, 1, 2, 2, 1,
        2, 1, 0, 2])
    assert np.array_equal(out["z"][-1], expected), "reference KAT 1 labels not reproduced"
    assert abs(out["log_marg"][-1] - (-411.811711231)) < 1e-7
    print("   KAT1 (2014 golden labels + log_marg) reproduced by the scratch reference")


def case_kat3():
    # pybgmm/tests/test_igmm.py:106-146  (each-in-own, N=20, 1 sweep)
    X, z_true = kat_data(1, 2, 20, 4)
    pp = gendata.demo_prior_params(2, v_0=5)
    out = run_case("kat3_each_in_own", "CRPMM", X, pp, 1.0, "each-in-own", 3, None, 1,
                   (1, 1), true_assignments=z_true, reseed_before_model=False)
    expected = np.array([5, 2, 4, 3, 2, 7, 2, 7, 1, 0, 4, 6, 4, 1, 6, 4, 1, 7, 1, 0])
    assert np.array_equal(out["z"][-1], expected), "reference KAT 3 labels not reproduced"
    print("   KAT3 reproduced")


def case_kat4():
    # pybgmm/tests/test_igmm.py:149-187  (seed 2, N=5, v_0=D+3)
    X, z_true = kat_data(2, 2, 5, 4)
    pp = gendata.demo_prior_params(2)
    out = run_case("kat4_log_marg", "CRPMM", X, pp, 1.0, "each-in-own", 3, None, 1,
                   (2, 2), true_assignments=z_true, reseed_before_model=False)
    assert abs(out["log_marg"][-1] - (-30.771535771)) < 1e-7
    print("   KAT4 reproduced")


def case_c1():
    # BASELINE config 1: gendata_1d(500), CRPMM D=1, "rand" K=3, 40 sweeps, seeds 1/1
    _mu, X, y = gendata.gendata_1d(500)
    pp = gendata.demo_prior_params(1)
    run_case("c1_crpmm_1d", "CRPMM", X, pp, 1.0, "rand", 3, None, 40, (1, 1),
             true_assignments=y, recipe="gendata_1d(500)")


def case_c2_twin():
    X, z_true = gendata.synth_mixture(2000, 2, 20, seed=7)
    pp = gendata.demo_prior_params(2)
    run_case("c2twin_crpmm_2d", "CRPMM", X, pp, 1.0, "rand", 20, 80, 5, (1, 1),
             true_assignments=z_true, recipe="synth_mixture(2000,2,20,seed=7)")


def case_c3_twin():
    X, z_true = gendata.synth_mixture(2000, 16, 100, seed=8)
    pp = gendata.demo_prior_params(16)
    run_case("c3twin_pcrpmm_16d", "PCRPMM", X, pp, 1.0, z_true, 100, 400, 3, (1, 1),
             sampler_kwargs=dict(n_power=1.01, power_burnin=0),
             true_assignments=z_true, recipe="synth_mixture(2000,16,100,seed=8)",
             store_X=True, skip_metrics=True)


def case_c3_rand():
    X, z_true = gendata.synth_mixture(1500, 16, 30, seed=9)
    pp = gendata.demo_prior_params(16)
    run_case("c3rand_pcrpmm_16d", "PCRPMM", X, pp, 1.0, "rand", 30, 120, 3, (2, 2),
             sampler_kwargs=dict(n_power=1.01, power_burnin=0),
             true_assignments=z_true, recipe="synth_mixture(1500,16,30,seed=9)",
             skip_metrics=True)


def case_c4_twin():
    X, z_true = gendata.synth_mixture(4000, 64, 200, seed=11)
    pp = gendata.demo_prior_params(64)
    run_case("c4twin_crpmm_64d", "CRPMM", X, pp, 1.0, z_true, 200, 800, 2, (1, 1),
             true_assignments=z_true, recipe="synth_mixture(4000,64,200,seed=11)",
             store_X=False, skip_metrics=True)


def case_c4_rand():
    X, z_true = gendata.synth_mixture(600, 64, 8, seed=12)
    pp = gendata.demo_prior_params(64)
    run_case("c4rand_crpmm_64d", "CRPMM", X, pp, 1.0, "rand", 8, 64, 2, (3, 3),
             true_assignments=z_true, recipe="synth_mixture(600,64,8,seed=12)",
             store_X=False, skip_metrics=True)


def case_each_in_own():
    X, z_true = gendata.synth_mixture(50, 2, 4, seed=21)
    pp = gendata.demo_prior_params(2)
    run_case("each_in_own_50", "CRPMM", X, pp, 1.0, "each-in-own", 1, None, 3, (4, 4),
             true_assignments=z_true)


def case_one_by_one():
    X, z_true = gendata.synth_mixture(50, 2, 4, seed=22)
    pp = gendata.demo_prior_params(2)
    run_case("one_by_one_50", "CRPMM", X, pp, 1.0, "one-by-one", 1, None, 3, (5, 5),
             true_assignments=z_true)


def case_pcrp_burnin():
    X, z_true = gendata.synth_mixture(300, 2, 6, seed=23)
    pp = gendata.demo_prior_params(2)
    run_case("pcrp_burnin_2d", "PCRPMM", X, pp, 1.0, "rand", 6, 60, 4, (6, 6),
             sampler_kwargs=dict(n_power=1.5, power_burnin=1),
             true_assignments=z_true)


def case_pcrp_flag_off():
    # flag_power=False: fixed visiting order and plain CRP weights (pcrpmm.py:86-112)
    X, z_true = gendata.synth_mixture(200, 3, 5, seed=24)
    pp = gendata.demo_prior_params(3)
    run_case("pcrp_flagoff_3d", "PCRPMM", X, pp, 1.0, "rand", 5, 50, 3, (7, 7),
             sampler_kwargs=dict(n_power=1.3, power_burnin=0, flag_power=False),
             true_assignments=z_true)


def case_general_prior():
    # non-zero m_0, full (non-diagonal) S_0, alpha != 1, D=3
    rs = np.random.RandomState(31)
    X, z_true = gendata.synth_mixture(240, 3, 5, seed=25)
    X = X + np.array([10.0, -3.0, 0.5])
    A = rs.randn(3, 3)
    S_0 = A.dot(A.T) + 2.0 * np.eye(3)
    m_0 = np.array([9.0, -2.5, 1.0])
    run_case("general_prior_3d", "CRPMM", X, (m_0, 0.2, 6, S_0), 2.5, "rand", 4, 60, 5,
             (8, 8), true_assignments=z_true)


def case_d12():
    # D not a multiple of 4/16 -> exercises the padded paths of the device kernels
    X, z_true = gendata.synth_mixture(400, 12, 7, seed=26)
    pp = gendata.demo_prior_params(12)
    run_case("crpmm_12d", "CRPMM", X, pp, 1.0, "rand", 7, 70, 3, (9, 9),
             true_assignments=z_true, skip_metrics=True)


def case_d192():
    # round 6: full covariance beyond D = 128 (the device's general route: rebuilds in a global-memory workspace)
    X, z_true = gendata.synth_mixture(300, 192, 4, seed=27, mu_scale=1.5)
    pp = gendata.demo_prior_params(192)
    run_case("crpmm_192d", "CRPMM", X, pp, 1.0, "rand", 6, 48, 2, (13, 13),
             true_assignments=z_true, recipe="synth_mixture(300,192,4,seed=27,mu_scale=1.5)",
             store_X=False, skip_metrics=True)


# ---- diagonal covariance: S_0 is a D-vector ------------------------- #
def diag_prior(D, v_0=None):
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D, v_0=v_0)
    return m_0, k_0, v_0, np.ascontiguousarray(np.diag(S_0))


def case_diag_kat():
    X, z_true = kat_data(1, 2, 100, 4)
    run_case("diag_kat_2d", "CRPMM", X, diag_prior(2, v_0=5), 1.0, "rand", 3, None, 10, (1, 1),
             true_assignments=z_true, reseed_before_model=False, cov_type="diag")


def case_diag_each_in_own():
    X, z_true = gendata.synth_mixture(30, 2, 3, seed=41)
    run_case("diag_each_in_own_30", "CRPMM", X, diag_prior(2), 1.0, "each-in-own", 1, None, 3, (4, 4),
             true_assignments=z_true, cov_type="diag")


def case_diag_pcrp():
    X, z_true = gendata.synth_mixture(600, 16, 10, seed=42)
    run_case("diag_pcrp_16d", "PCRPMM", X, diag_prior(16), 1.0, "rand", 10, 80, 3, (5, 5),
             sampler_kwargs=dict(n_power=1.1, power_burnin=0), true_assignments=z_true,
             skip_metrics=True, cov_type="diag")


def case_diag_general():
    X, z_true = gendata.synth_mixture(300, 5, 6, seed=43)
    X = X + np.array([4.0, -2.0, 0.0, 7.5, 1.0])
    m_0 = np.array([3.5, -1.0, 0.5, 7.0, 0.0])
    S_0 = np.array([2.0, 0.7, 1.3, 5.0, 0.9])
    run_case("diag_general_5d", "CRPMM", X, (m_0, 0.3, 7, S_0), 2.5, "rand", 5, 60, 4, (6, 6),
             true_assignments=z_true, cov_type="diag")


def case_diag_64d():
    X, z_true = gendata.synth_mixture(800, 64, 8, seed=44)
    run_case("diag_crpmm_64d", "CRPMM", X, diag_prior(64), 1.0, "rand", 8, 64, 2, (7, 7),
             true_assignments=z_true, store_X=False, skip_metrics=True, cov_type="diag",
             recipe="synth_mixture(800,64,8,seed=44)")


def case_diag_256d():
    # diagonal components beyond D = 128 (the state is a D-vector: no limit in the reference,
    # gaussian_components_diag.py:92)
    X, z_true = gendata.synth_mixture(400, 256, 6, seed=45)
    run_case("diag_crpmm_256d", "CRPMM", X, diag_prior(256), 1.0, "rand", 6, 48, 2, (8, 8),
             true_assignments=z_true, store_X=False, skip_metrics=True, cov_type="diag",
             recipe="synth_mixture(400,256,6,seed=45)")


# ---- fixed-variance components ------------------------------------------ #
def case_fixed_256d():
    X, z_true = gendata.synth_mixture(400, 256, 6, seed=65)
    prior = (0.49 * np.ones(256), np.zeros(256), 16.0 * np.ones(256))
    run_case("fixed_pcrp_256d", "PCRPMM", X, prior, 1.0, "rand", 6, 48, 2, (15, 15),
             sampler_kwargs=dict(n_power=1.1, power_burnin=0), true_assignments=z_true, store_X=False,
             skip_metrics=True, cov_type="fixed", recipe="synth_mixture(400,256,6,seed=65)")



def case_fixed_2d():
    X, z_true = gendata.synth_mixture(300, 2, 5, seed=61)
    prior = (np.array([0.49, 0.6]), np.array([0.5, -0.3]), np.array([16.0, 12.0]))   # var, mu_0, var_0
    run_case("fixed_2d", "CRPMM", X, prior, 1.0, "rand", 6, 60, 5, (12, 12),
             true_assignments=z_true, cov_type="fixed")


def case_fixed_each_in_own():
    X, z_true = gendata.synth_mixture(30, 3, 3, seed=62)
    prior = (0.49 * np.ones(3), np.zeros(3), 16.0 * np.ones(3))
    run_case("fixed_each_in_own_30", "CRPMM", X, prior, 1.5, "each-in-own", 1, None, 3, (13, 13),
             true_assignments=z_true, cov_type="fixed")


def case_fixed_pcrp_16d():
    X, z_true = gendata.synth_mixture(500, 16, 8, seed=63)
    prior = (0.49 * np.ones(16), np.zeros(16), 16.0 * np.ones(16))
    run_case("fixed_pcrp_16d", "PCRPMM", X, prior, 1.0, "rand", 8, 80, 3, (14, 14),
             sampler_kwargs=dict(n_power=1.1, power_burnin=0), true_assignments=z_true,
             skip_metrics=True, cov_type="fixed")


def case_adap():
    # the sweep exponent adapts to the share of small clusters
    # (adapcrp_burnin=-1: with the reference's default 0 its first sweep dies on an unbound local)
    X, z_true = gendata.synth_mixture(400, 2, 5, seed=51)
    run_case("adap_2d", "ADAPCRPMM", X, gendata.demo_prior_params(2), 1.0, "rand", 12, 80, 6, (11, 11),
             sampler_kwargs=dict(r_up=1.4, adapcrp_perct=0.08, adapcrp_burnin=-1),
             true_assignments=z_true)


def three_blobs_1d(n_each, seed):
    """Three well separated 1-D clusters: the chain sits at K = 3, where the reference snapshots
    its distribution dict (num_saved = 3 is the default of every sampler)."""
    rs = np.random.RandomState(seed)
    X = np.concatenate([rs.normal(-6.0, 0.5, n_each), rs.normal(0.0, 0.5, n_each), rs.normal(6.0, 0.5, n_each)])
    y = np.repeat(np.arange(3), n_each)
    p = rs.permutation(3 * n_each)
    return X[p].reshape(-1, 1), y[p]


def api_case(name, model, weight_first, seeds, n_iter=25, alpha=0.05):
    """Class-level fixture: the whole public path with the distribution dict ON (num_saved == K,
    reference igmm/igmm.py:128-197, crpmm.py:49-50), then rand_k of every component
    (gaussian_components.py:291-303, prior/wishart.py:16-32), then a few draws from both global
    streams -- what the caller would see next."""
    from pybgmm.igmm import CRPMM, PCRPMM
    from pybgmm.prior import NIW
    X, y = three_blobs_1d(50, seeds[0] + 100)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(1)
    random.seed(seeds[0])
    np.random.seed(seeds[1])
    cls = {"CRPMM": CRPMM, "PCRPMM": PCRPMM}[model]
    mm = cls(X, NIW(m_0, k_0, v_0, S_0), alpha, None, assignments="rand", K=3, K_max=40)
    record, dist = mm.collapsed_gibbs_sampler(n_iter, y, num_saved=3, weight_first=weight_first)
    K = mm.components.K
    rk_mu, rk_sigma = [], []
    for k in range(K):
        mu, sigma = mm.components.rand_k(k)
        rk_mu.append(np.asarray(mu, dtype=np.float64).ravel())
        rk_sigma.append(np.asarray(sigma, dtype=np.float64).ravel())
    after_random = np.array([random.random() for _ in range(4)])
    after_numpy = np.random.random_sample(4)
    out = {"case": name, "model": model, "weight_first": bool(weight_first), "n_iter": n_iter,
           "seed_random": seeds[0], "seed_numpy": seeds[1], "X": X, "true_assignments": y,
           "m_0": m_0, "k_0": float(k_0), "v_0": int(v_0), "S_0": S_0, "alpha": alpha, "K_arg": 3, "K_max": 40,
           "rec_components": np.array(record["components"], dtype=np.int64),
           "rec_log_marg": np.array(record["log_marg"], dtype=np.float64),
           "dist_mean": dist["mean"], "dist_variance": dist["variance"], "dist_weights": dist["weights"],
           "final_z": np.array(mm.components.assignments, dtype=np.int64), "final_K": K,
           "rand_k_mu": np.stack(rk_mu), "rand_k_sigma": np.stack(rk_sigma),
           "after_random": after_random, "after_numpy": after_numpy}
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print("%-22s %s weight_first=%d: K per sweep %s, %d snapshots, rand_k of %d components -> %s" % (
        name, model, weight_first, list(out["rec_components"]), dist["mean"].shape[1], K, os.path.basename(path)))
    assert dist["mean"].shape[1] >= 3, "the distribution dict was hardly exercised"


def case_api_crp_wf():
    api_case("api_distdict_crpmm_wf", "CRPMM", True, (21, 21))


def case_api_crp_mf():
    api_case("api_distdict_crpmm_mf", "CRPMM", False, (22, 22))


def case_api_pcrp_wf():
    api_case("api_distdict_pcrpmm_wf", "PCRPMM", True, (23, 23))


def case_demo2d():
    """The reference's 2-D demo script as a user runs it (examples/crpmm_2d_demo.py:25-84, the recipe written out here,
    the plotting left out): seeds, data and prior from the global streams, CRPMM from "rand" with K = 3, 40 sweeps with the
    default num_saved, then rand_k of every component, then what both global streams deliver next."""
    from pybgmm.igmm import CRPMM
    from pybgmm.prior import NIW
    random.seed(1)
    np.random.seed(1)
    D, N, K_true, alpha, K, n_iter = 2, 100, 4, 1., 3, 40
    mu_scale, covar_scale = 4.0, 0.7
    z_true = np.random.randint(0, K_true, N)
    mu = np.random.randn(D, K_true) * mu_scale
    X = (mu[:, z_true] + np.random.randn(D, N) * covar_scale).T
    m_0 = np.zeros(D)
    k_0 = covar_scale ** 2 / mu_scale ** 2
    v_0 = D + 3
    S_0 = covar_scale ** 2 * v_0 * np.eye(D)
    # (the demo hands a directory: with K == num_saved the reference saves a scatter plot there per snapshot)
    # its 2017 plotting helpers no longer run under today's matplotlib (Ellipse's signature): stubbed from outside --
    # they draw, they touch neither the sampler nor the random streams
    import pybgmm.igmm.igmm as ref_igmm
    ref_igmm.plot_ellipse = lambda *a, **k: None
    ref_igmm.plot_mixture_model = lambda *a, **k: None
    # and prior/wishart.py:18 asks `C == None` of an array -- a scalar False under the numpy of its day, an elementwise
    # comparison now: the factor goes in as an ndarray view whose comparison with None answers as it did then
    import pybgmm.prior.wishart as ref_wishart

    class _OldEq(np.ndarray):
        def __eq__(self, other):
            return False if other is None else np.ndarray.__eq__(self, other)
        __hash__ = None
    real_iw = ref_wishart.iwishrnd
    ref_wishart.iwishrnd = lambda sigma, v_0, C=None: real_iw(sigma, v_0, None if C is None else np.asarray(C).view(_OldEq))
    save_path = tempfile.mkdtemp(prefix="pybgmm_demo_") + "/"
    crpmm = CRPMM(X, NIW(m_0, k_0, v_0, S_0), alpha, save_path=save_path, assignments="rand", K=K)
    record_dict, _dist = crpmm.collapsed_gibbs_sampler(n_iter, z_true)
    shutil.rmtree(save_path, ignore_errors=True)
    Kf = crpmm.components.K
    rk_mu, rk_sigma = [], []
    for k in range(Kf):
        m, sg = crpmm.components.rand_k(k)
        rk_mu.append(np.asarray(m, dtype=np.float64).ravel())
        rk_sigma.append(np.asarray(sg, dtype=np.float64).ravel())
    ref_wishart.iwishrnd = real_iw
    out = {"case": "demo_crpmm_2d", "X": X, "true_assignments": np.asarray(z_true, dtype=np.int64),
           "rec_components": np.array(record_dict["components"], dtype=np.int64),
           "rec_log_marg": np.array(record_dict["log_marg"], dtype=np.float64),
           "rec_nmi": np.array(record_dict["nmi"], dtype=np.float64),
           "rec_nk": np.array(record_dict["nk"]),
           "final_z": np.array(crpmm.components.assignments, dtype=np.int64), "final_K": Kf,
           "rand_k_mu": np.stack(rk_mu), "rand_k_sigma": np.stack(rk_sigma),
           "after_random": np.array([random.random() for _ in range(4)]), "after_numpy": np.random.random_sample(4)}
    path = os.path.join(HERE, "demo_crpmm_2d.npz")
    np.savez_compressed(path, **out)
    print("demo_crpmm_2d          K per sweep %s, log_marg[-1]=%.12f -> %s" % (
        list(out["rec_components"]), out["rec_log_marg"][-1], os.path.basename(path)))


CASES = {
    "demo2d": case_demo2d,
    "api_crp_wf": case_api_crp_wf, "api_crp_mf": case_api_crp_mf, "api_pcrp_wf": case_api_pcrp_wf,
    "kat1": case_kat1, "kat3": case_kat3, "kat4": case_kat4, "c1": case_c1,
    "c2twin": case_c2_twin, "c3twin": case_c3_twin, "c3rand": case_c3_rand,
    "c4twin": case_c4_twin, "c4rand": case_c4_rand,
    "each_in_own": case_each_in_own, "one_by_one": case_one_by_one,
    "pcrp_burnin": case_pcrp_burnin, "pcrp_flagoff": case_pcrp_flag_off,
    "general_prior": case_general_prior, "d12": case_d12, "d192": case_d192,
    "diag_kat": case_diag_kat, "diag_each_in_own": case_diag_each_in_own, "diag_pcrp": case_diag_pcrp,
    "diag_general": case_diag_general, "diag_64d": case_diag_64d, "diag_256d": case_diag_256d, "adap": case_adap,
    "fixed_256d": case_fixed_256d,
    "fixed_2d": case_fixed_2d, "fixed_each_in_own": case_fixed_each_in_own, "fixed_pcrp": case_fixed_pcrp_16d,
}


def main(argv):
    scratch = import_reference()
    try:
        for name in (argv or list(CASES)):
            CASES[name]()
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main(sys.argv[1:])

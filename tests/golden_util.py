"""Load golden fixtures (tests/golden/*.npz, produced by make_golden.py)."""
import os
import re

import numpy as np

from pybgmm_amd.utils import gendata

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

_EVERY = sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.endswith(".npz"))
ALL_CASES = [c for c in _EVERY if not c.startswith(("diag_", "fixed_", "api_", "demo_"))]   # full covariance (incl. ADAPCRPMM)
DEMO_CASES = [c for c in _EVERY if c.startswith("demo_")]             # a reference example script's body, run as a user would
API_CASES = [c for c in _EVERY if c.startswith("api_")]               # class-level runs with the distribution dict on
DIAG_CASES = [c for c in _EVERY if c.startswith("diag_")]             # covariance_type="diag"
FIXED_CASES = [c for c in _EVERY if c.startswith("fixed_")]           # covariance_type="fixed"
# cases whose reference trajectory is short enough for the pure-numpy oracle
SMALL_CASES = ["kat1_igmm_2d", "kat3_each_in_own", "kat4_log_marg", "each_in_own_50",
               "one_by_one_50", "pcrp_burnin_2d", "pcrp_flagoff_3d", "general_prior_3d"]


class Golden(object):
    def __init__(self, name):
        d = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.name = name
        self.d = d
        for key in ("N", "D", "K_max", "n_iter", "v_0", "power_burnin", "K_init"):
            setattr(self, key, int(d[key]))
        for key in ("alpha", "k_0", "n_power"):
            setattr(self, key, float(d[key]))
        self.flag_power = bool(d["flag_power"])
        self.model = str(d["model"])
        self.cov_type = str(d["cov_type"])
        self.m_0, self.S_0 = d["m_0"], d["S_0"]
        self.z_init = d["z_init"]
        self.u, self.order = d["u"], d["order"]
        self.z, self.K, self.counts, self.log_marg = d["z"], d["K"], d["counts"], d["log_marg"]
        if "X" in d.files:
            self.X = d["X"]
        else:
            m = re.match(r"synth_mixture\((\d+),(\d+),(\d+),seed=(\d+)(?:,mu_scale=([0-9.]+))?\)", str(d["recipe"]))
            assert m, "fixture %s has neither X nor a known recipe" % name
            N, D, K, seed = (int(g) for g in m.groups()[:4])
            self.X, _ = gendata.synth_mixture(N, D, K, seed, **({"mu_scale": float(m.group(5))} if m.group(5) else {}))
        assert gendata.array_digest(self.X) == str(d["X_sha256"]), \
            "regenerated X does not match the fixture's sha256"

    @property
    def prior(self):
        return (self.m_0, self.k_0, self.v_0, self.S_0)

    def _adap_power(self, it):
        """ADAPCRPMM (igmm/adapcrpmm.py:100-104): 1 + (r_up - 1) * share of clusters holding at most
        N * perct points, from the counts BEFORE sweep ``it``."""
        z = self.z_init if it == 0 else self.z[it - 1]
        nk = np.bincount(z[z >= 0])
        small = np.sum(nk <= self.N * float(self.d["adap_perct"])) * 1.0 / len(nk)
        return 1.0 + (float(self.d["adap_r_up"]) - 1.0) * small

    def sweep_order(self, it):
        if self.model == "ADAPCRPMM":
            # a permutation is drawn only in the sweeps whose exponent exceeds 1
            used = [t for t in range(self.n_iter) if self._adap_power(t) > 1]
            return self.order[used.index(it)] if it in used else None
        return self.order[it] if self.order.shape[0] else None

    def sweep_power(self, it):
        """pCRP exponent active in sweep ``it`` (None = plain CRP weights)."""
        if self.model == "ADAPCRPMM":
            return self._adap_power(it) if it > int(self.d["adap_burnin"]) else None
        return self.n_power if (self.flag_power and it > self.power_burnin) else None

    def counts_at(self, it):
        return self.counts[it, :self.K[it]]

    def probes(self):
        out, off = [], 0
        for u, k, n in zip(self.d["probe_u"], self.d["probe_k"], self.d["probe_len"]):
            out.append((self.d["probe_prob"][off:off + n], float(u), int(k)))
            off += n
        return out

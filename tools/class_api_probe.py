"""End-to-end cost of the user-facing classes (host RNG, H2D, sweep, record dict) at bench size."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pybgmm_amd.igmm import CRPMM
from pybgmm_amd.prior import NIW
from pybgmm_amd.utils import gendata
N, D, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
X, zt = gendata.synth_mixture(N, D, K, seed=1)
random.seed(1); np.random.seed(1)
mm = CRPMM(X, NIW(*gendata.demo_prior_params(D)), 1.0, None, assignments=zt, K_max=4 * K)
for metrics in (True, False):
    mm.record_metrics = metrics
    t = time.time()
    rec, _ = mm.collapsed_gibbs_sampler(5, zt, num_saved=0)
    dt = (time.time() - t) / 5
    print("record_metrics=%s: %.1f ms per sweep wall (sample_time %.1f ms), nmi %.4f K %d" % (
        metrics, dt * 1e3, 1e3 * np.mean(rec["sample_time"]), rec["nmi"][-1], rec["components"][-1]))

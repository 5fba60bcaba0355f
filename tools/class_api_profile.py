import os, random, sys, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from pybgmm_amd.igmm import CRPMM
from pybgmm_amd.prior import NIW
from pybgmm_amd.utils import gendata
N, D, K = 1000000, 64, 200
X, zt = gendata.synth_mixture(N, D, K, seed=1)
random.seed(1); np.random.seed(1)
mm = CRPMM(X, NIW(*gendata.demo_prior_params(D)), 1.0, None, assignments=zt, K_max=4 * K)
mm.collapsed_gibbs_sampler(2, zt, num_saved=0)
pr = cProfile.Profile(); pr.enable()
mm.collapsed_gibbs_sampler(10, zt, num_saved=0)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)

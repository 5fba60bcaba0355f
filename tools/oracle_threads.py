"""Thread scaling of the C oracle on this host (test infrastructure):   python tools/oracle_threads.py
us per visit of oracle/gibbs_oracle.c for 1 .. 64 threads at the BASELINE shapes -- what tests/ and bench.py's threaded
cpu_baseline leg can count on.  The floats are the same for every count (tests/test_oracle_c.py)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import c_oracle                     # noqa: E402
from pybgmm_amd.utils import gendata            # noqa: E402

print("host cores available: %d" % len(os.sched_getaffinity(0)))
for (D, K, nv) in ((64, 200, 1500), (128, 200, 400), (128, 40, 600), (16, 100, 20000)):
    N = max(4 * K, nv)
    X, zt = gendata.synth_mixture(N, D, K, 3)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    u = np.random.RandomState(1).random_sample(N)
    line = []
    for T in (1, 4, 8, 16, 32, 64):
        if c_oracle.set_threads(T) != T:
            break
        o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, zt, 4 * K, scipy_tables=False)
        t = time.time()
        o.sweep(u, n_visits=nv)
        line.append("T=%d %.1f" % (T, (time.time() - t) / nv * 1e6))
    print("D=%d K=%d us/visit: %s" % (D, K, "  ".join(line)), flush=True)

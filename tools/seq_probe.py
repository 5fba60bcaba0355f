"""Timing probe of the small-D sequential sweep at the C2 shape (movers dense) and a well-separated twin.
    python tools/seq_probe.py [prof]     (prof: libbgmm_hip_prof.so of tools/build_prof_seq.sh, per-phase shader clocks)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pybgmm_amd import _build
prof = len(sys.argv) > 1 and sys.argv[1] == "prof"
if prof:
    _build.LIB = os.path.join(os.path.dirname(_build.LIB), "libbgmm_hip_prof.so")
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata
for (N, D, K, sep) in ((100000, 2, 20, None), (100000, 2, 20, 8.0), (100000, 4, 40, 3.0)):
    X, zt = gendata.synth_mixture(N, D, K, seed=1) if sep is None else gendata.synth_mixture(N, D, K, seed=1, mu_scale=sep)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    c = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 256)
    c.set_assignments(zt)
    rs = np.random.RandomState(0)
    for it in range(5):
        u = rs.random_sample(N)
        pc0 = np.array(c.phase_clocks())
        c.stage(u, None); c.synchronize()
        t0 = time.time(); c.sweep_staged(None); c.synchronize(); dt = time.time() - t0
        st = c.sweep_stats()
    print("N=%d D=%d K=%d sep=%s: %.1f ms/sweep, %.3f us/visit, moves %d, K_end %d" % (
        N, D, K, sep, dt * 1e3, dt * 1e6 / N, st["moves"], c.K))
    if prof:
        pc = np.array(c.phase_clocks()) - pc0
        names = ["ring read", "home lookup", "evaluate", "stats", "commit", "rebuild", "barrier 1", "barrier 2"]
        tot = float(pc[:8].sum())
        for k in range(8):
            print("   %-12s %6.1f %%   %9.0f cycles per wave and 1000 visits" % (names[k], 100.0 * pc[k] / tot, pc[k] / 8.0 / N * 1000))
    c.close()

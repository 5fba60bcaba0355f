"""Frozen-factor windows (resolver_mode 3) against the C port of the reference on a fresh seeded problem,
and what a sweep costs through them.
    python tools/gram_probe.py N D K [n_sweeps] [rand|true|own] [mode] [check]
mode: resolver_mode of the timed context (default 0 = auto); check = 1 compares every sweep with the C oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata
N, D, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
n_sweeps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
init = sys.argv[5] if len(sys.argv) > 5 else "rand"
mode = int(sys.argv[6]) if len(sys.argv) > 6 else 0
check = int(sys.argv[7]) if len(sys.argv) > 7 else 0
sep = float(sys.argv[8]) if len(sys.argv) > 8 else 4.0
X, zt = gendata.synth_mixture(N, D, K, seed=11, mu_scale=sep)
m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
rs = np.random.RandomState(0)
if init == "rand":
    z0 = np.unique(rs.randint(0, K, N), return_inverse=True)[1]
elif init == "own":
    z0 = np.arange(N)
else:
    z0 = zt.copy()
K_max = max(4 * K, N if init == "own" else 0)
ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, K_max)
ctx.set_tuning(resolver_mode=mode)
ctx.set_assignments(z0)
o = None
if check:
    from oracle import c_oracle
    o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, z0, K_max, scipy_tables=False)
for it in range(n_sweeps):
    u = rs.random_sample(N)
    t0 = time.time(); ctx.sweep(u, None, None); dt = time.time() - t0
    st, ps = ctx.sweep_stats(), ctx.path_stats()
    msg = ""
    if o is not None:
        o.sweep(u)
        z = ctx.assignments()
        bad = np.nonzero(z != o.z)[0]
        lm, lo = ctx.log_marg(), o.log_marg()
        msg = "  oracle: %d labels differ%s, log_marg rel %.1e" % (
            bad.size, (" (first at %d)" % bad[0]) if bad.size else "", abs(lm - lo) / abs(lo))
    print("sweep %2d: %9.3f ms  moves %7d  steps %6d  K %3d  frozen windows %6d (%7d visits, %.1f per window)  %.2f us/move%s" % (
        it, dt * 1e3, st["moves"], st["steps"], ctx.K, ps["frozen_windows"], ps["frozen_window_visits"],
        ps["frozen_window_visits"] / max(ps["frozen_windows"], 1), dt * 1e6 / max(st["moves"], 1), msg), flush=True)
pc = ctx.phase_clocks()
if any(pc):
    w = max(pc[8], 1)
    print("resolver ticks per window: prologue %.0f  draws %.0f  bookkeeping %.0f  update(home) %.0f  update(dest) %.0f  | windows %d" % (
        pc[0] / w, pc[1] / w, pc[2] / w, pc[3] / w, pc[4] / w, pc[8]))
    print("  rebuild of a slot (ticks per window, block 0): load + S_N %.0f, Gershgorin %.0f, factorisation %.0f, inverse %.0f, outputs %.0f" % tuple(pc[10 + k] / w for k in range(5)))
    print("  wave 0 behind the barrier (housekeeping + the next visit ahead): %.0f" % (pc[5] / w))

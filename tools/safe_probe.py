"""Safe-stay windows (kernels_safe.hip): parity against the C oracle and against the other mover paths, then timing.
    python tools/safe_probe.py check            small problems, forced safe-stay windows vs the C oracle
    python tools/safe_probe.py flip N D K [resolver_mode] [n_sweeps] [sep]    a chain at the truth with N/500 labels flipped
    python tools/safe_probe.py true N D K [resolver_mode] [n_sweeps] [sep]    a chain at the truth (sep < 1: overlapping clusters, movers at equilibrium)
    python tools/safe_probe.py rand N D K [resolver_mode] [n_sweeps] [sep] [pcrp]   from the reference's "rand" initialisation
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata


def run(N, D, K, init, resolver, n_sweeps, sep=4.0, pcrp=False, oracle=False, seed=11, verbose=True, budget=0.0):
    X, zt = gendata.synth_mixture(N, D, K, seed=seed, mu_scale=sep)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    rs = np.random.RandomState(seed)
    if init == "rand":
        z0 = np.unique(rs.randint(0, K, N), return_inverse=True)[1]
    elif init == "true":
        z0 = zt.copy()
    else:
        z0 = zt.copy()
        flip = rs.choice(N, size=max(N // 500, 1), replace=False)
        z0[flip] = rs.randint(0, K, size=flip.size)
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
    ctx.set_tuning(resolver_mode=resolver)
    ctx.set_safe_budget(budget)
    ctx.set_assignments(z0)
    o = None
    if oracle:
        from oracle import c_oracle
        o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, z0, 4 * K, scipy_tables=False)
    zs = []
    for it in range(n_sweeps):
        u = rs.random_sample(N)
        order = rs.permutation(N).astype(np.int64) if pcrp else None
        power = 1.01 if (pcrp and it > 0) else None
        ctx.stage(u, order); ctx.synchronize()
        t0 = time.time(); ctx.sweep_staged(power); ctx.synchronize(); dt = time.time() - t0
        st, ss, ps = ctx.sweep_stats(), ctx.safe_stats(), ctx.path_stats()
        z = ctx.assignments()
        zs.append(z)
        msg = ""
        if o is not None:
            o.sweep(u, order, power) if pcrp else o.sweep(u)
            bad = np.nonzero(z != o.z)[0]
            msg = "  ORACLE %s" % ("ok" if bad.size == 0 else "DIFFERS at %d labels, first i=%d" % (bad.size, bad[0]))
            if bad.size:
                print(msg); ctx.close(); return None
        if verbose and os.environ.get("SAFE_DEBUG"):
            print("   why (cumulative): no-home/small, chi>=1, off-table, others heavy, new-table heavy, u near end, SAFE:", ctx.phase_clocks()[1:8])
        if verbose:
            print("sweep %2d: %9.3f ms moves %7d K %3d | safe windows %6d examined %9d walked %7d cuts %5d budget %.4f L %7d | frozen %6d steps %6d certified %7d%s" % (
                it, dt * 1e3, st["moves"], ctx.K, ss["windows"], ss["visits_examined"], ss["unproven_walked"], ss["budget_cuts"],
                ss["budget"], ss["next_stretch"], ps["frozen_windows"], st["steps"], ctx.prune_stats()["certified_visits"], msg), flush=True)
    lm = ctx.log_marg()
    if os.environ.get("SAFE_DEBUG"):
        print("why (cumulative): unassigned/small-home, chi>=1, radius off the table, others too heavy, new table heavy, u near an end, SAFE:", ctx.phase_clocks()[1:8])
    ctx.close()
    return zs, lm


if __name__ == "__main__":
    what = sys.argv[1]
    if what == "check":
        ok = True
        for (N, D, K, init, sep, pcrp) in [(3000, 16, 12, "rand", 4.0, False), (20000, 16, 40, "flip", 1.6, False),
                                           (20000, 64, 20, "flip", 4.0, False), (8000, 32, 30, "rand", 1.2, True),
                                           (30000, 16, 60, "rand", 1.0, False), (6000, 128, 10, "flip", 4.0, False)]:
            for budget in (0.0, 1.0 / 64, 2.0):
                print("== N %d D %d K %d %s sep %.1f pcrp %d forced safe-stay windows, budget %s" % (N, D, K, init, sep, pcrp, budget or "auto"), flush=True)
                r = run(N, D, K, init, 4, 3, sep, pcrp, oracle=True, budget=budget)
                ok = ok and r is not None
        print("CHECK", "OK" if ok else "FAILED")
        sys.exit(0 if ok else 1)
    N, D, K = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    resolver = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    n_sweeps = int(sys.argv[6]) if len(sys.argv) > 6 else 6
    sep = float(sys.argv[7]) if len(sys.argv) > 7 else 4.0
    pcrp = len(sys.argv) > 8 and sys.argv[8] == "pcrp"
    run(N, D, K, what, resolver, n_sweeps, sep, pcrp)

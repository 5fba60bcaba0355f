"""Chains side by side, again and again: G chains of one shape from random starts (their own generators, optionally a visiting
order each -- the pCRP sweep's), `reps` identical group runs of `sweeps` sweeps; every run must give what the first one gave
(labels, log marginal), and the first one what the chains give one by one.  A race between the shared launches' streams shows
up as a run that differs, or as a device error.
    python tools/soak_group.py N D K G reps [--sweeps 1] [--orders] [--sep 4.0]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pybgmm_amd.chains import ChainGroup
from pybgmm_amd.utils import gendata

ap = argparse.ArgumentParser()
for n in ("N", "D", "K", "G", "reps"):
    ap.add_argument(n, type=int)
ap.add_argument("--sweeps", type=int, default=1)
ap.add_argument("--orders", action="store_true")
ap.add_argument("--sep", type=float, default=4.0)
ap.add_argument("--start", default="rand", choices=["rand", "true"])
ap.add_argument("--no-solo", action="store_true")
a = ap.parse_args()
X, zt = gendata.synth_mixture(a.N, a.D, a.K, seed=1, mu_scale=a.sep)
prior = gendata.demo_prior_params(a.D)
z0s = [np.unique(np.random.RandomState(100 + c).randint(0, a.K, a.N), return_inverse=True)[1] if a.start == "rand" else zt
       for c in range(a.G)]
rs = np.random.RandomState(5)
orders = [[rs.permutation(a.N).astype(np.int64) for _ in range(a.G)] for _ in range(a.sweeps)] if a.orders else None


def run(g_lo, g_hi):
    grp = ChainGroup(X, *prior, 1.0, 4 * a.K, n_chains=g_hi - g_lo, seed=1 + g_lo)
    grp.set_assignments(z0s[g_lo:g_hi])
    for it in range(a.sweeps):
        grp.sweep(None if orders is None else orders[it][g_lo:g_hi], None)
    out = [(z.copy(), ctx.log_marg()) for z, ctx in zip(grp.assignments(), grp.ctxs)]
    stats = grp.ctxs[0].group_stats() if g_hi - g_lo > 1 else None
    grp.close()
    return out, stats


t0 = time.time()
first, st = run(0, a.G)
print("group run 0: %.2f s, chain 0's batches: %s" % (time.time() - t0, st), flush=True)
bad = 0
for c in range(0 if not a.no_solo else a.G, a.G):
    one, _ = run(c, c + 1)
    ok = np.array_equal(one[0][0], first[c][0]) and abs(one[0][1] - first[c][1]) <= 1e-12 * abs(first[c][1])
    bad += not ok
    if not ok:
        print("chain %d: the group's run differs from the solo run" % c, flush=True)
for r in range(1, a.reps):
    try:
        again, _ = run(0, a.G)
    except Exception as e:                              # noqa: BLE001  (a device error is a finding, not a crash of the soak)
        print("run %d: %s" % (r, e), flush=True)
        bad += 1
        continue
    diff = [c for c in range(a.G) if not (np.array_equal(again[c][0], first[c][0]) and again[c][1] == first[c][1])]
    if diff:
        print("run %d: chains %s differ from run 0" % (r, diff), flush=True)
        bad += 1
print("SOAK_GROUP %s (%d runs of %d chains, %.0f s)" % ("OK" if bad == 0 else "FAILED: %d" % bad, a.reps, a.G, time.time() - t0))

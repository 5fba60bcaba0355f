"""Soak of the device permutations in flight against numpy: many calls in a row, the caller drawing from the stream at random
moments, every permutation and every state compared:  python tools/permsoak.py N calls [seed]
(BGMM_DEV_OPTIONS=perm_era=3 in the environment makes the word stream's buffer start over every couple of calls.)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata

N, calls = int(sys.argv[1]), int(sys.argv[2])
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 1
X, zt = gendata.synth_mixture(N, 2, 3, seed=1)
ctx = _lib.Context(X, *gendata.demo_prior_params(2), 1.0, 12)
ctx.set_assignments(zt)
host = np.random.RandomState(seed)
dice = np.random.RandomState(seed + 1000)
key, pos = host.get_state()[1].copy(), int(host.get_state()[2])
bad = 0
t0 = time.time()
for it in range(calls):
    r = dice.random_sample()
    if r < 0.03:
        host.random_sample(int(dice.randint(1, 700)))          # (the caller draws something else)
        key, pos = host.get_state()[1].copy(), int(host.get_state()[2])
    elif r < 0.05:
        time.sleep(0.002)                                       # (everything in flight has long finished)
    expect = host.permutation(N)
    key, pos = ctx.stage_permutation_mt19937(key, pos)
    ok = np.array_equal(ctx.staged_order(), expect) and np.array_equal(key, host.get_state()[1]) and pos == host.get_state()[2]
    if not ok:
        bad += 1
        print("call %d DIFFERENT" % it)
        if bad > 3:
            break
print("N %d: %d calls, %d different, %.1f s, %s" % (N, it + 1, bad, time.time() - t0, ctx.permutation_stats()))
ctx.close()
print("PERMSOAK", "OK" if bad == 0 else "FAILED")

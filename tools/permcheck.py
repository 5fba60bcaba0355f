"""Device permutations in a row against numpy (the generations in flight of api_perm.hip), with a caller that draws from the
stream in between, and the time per call:  python tools/permcheck.py [N] [calls]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get("BGMM_LIB_VARIANT"):          # (tools/build_variant.sh)
    from pybgmm_amd import _build
    _build.LIB = os.path.join(os.path.dirname(_build.LIB), "libbgmm_hip_%s.so" % os.environ["BGMM_LIB_VARIANT"])
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 14
X, zt = gendata.synth_mixture(N, 2, 3, seed=1)
ctx = _lib.Context(X, *gendata.demo_prior_params(2), 1.0, 12)
ctx.set_assignments(zt)
host = np.random.RandomState(N % 1000)
host.random_sample(N % 500 + 7)
key, pos = host.get_state()[1].copy(), int(host.get_state()[2])
bad = 0
for it in range(calls):
    if it == calls // 2:
        host.random_sample(5)                       # (the caller draws something else: the generations in flight are void)
        key, pos = host.get_state()[1].copy(), int(host.get_state()[2])
    expect = host.permutation(N)
    key, pos = ctx.stage_permutation_mt19937(key, pos)
    got = ctx.staged_order()
    ok = np.array_equal(got, expect) and np.array_equal(key, host.get_state()[1]) and pos == host.get_state()[2]
    bad += not ok
    print("call %2d: %s  %s" % (it, "equal" if ok else "DIFFERENT", ctx.permutation_stats()))
# rate: calls back to back (nothing else on the device)
t0 = time.perf_counter()
n = 200
for it in range(n):
    key, pos = ctx.stage_permutation_mt19937(key, pos)
dt = (time.perf_counter() - t0) / n
host.set_state(("MT19937", key, pos, 0, 0.0))
expect = host.permutation(N)
key, pos = ctx.stage_permutation_mt19937(key, pos)
ok = np.array_equal(ctx.staged_order(), expect)
bad += not ok
print("after %d more: %s; %.1f us per permutation; %s" % (n, "equal" if ok else "DIFFERENT", dt * 1e6, ctx.permutation_stats()))
ctx.close()
print("PERMCHECK", "OK" if bad == 0 else "FAILED")

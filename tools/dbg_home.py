import sys, time, numpy as np, random
sys.path.insert(0, '.')
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata
from pybgmm_amd.gaussian.gaussian_components import reference_tables
N, D, K = 1000000, 64, 200
X, zt = gendata.synth_mixture(N, D, K, seed=1)
m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K, tables=reference_tables(v_0, N))
ctx.set_tuning(prune_mode=3)
ctx.set_assignments(zt)
r = random.Random(5)
_, key_t, _ = r.getstate()
key, pos = np.asarray(key_t[:-1], dtype=np.uint32), int(key_t[-1])
mode = sys.argv[1]
for it in range(int(sys.argv[2])):
    if mode == "mt":
        key, pos = ctx.stage_mt19937(key, pos, None)
    else:
        ctx.stage(np.random.RandomState(it).random_sample(N))
    ctx.sweep_staged(None)
    if it % 50 == 0:
        print(it, ctx.sweep_stats()["moves"], ctx.short_step_stats(), flush=True)
print("done", mode)

"""Development probe: time of the first pruning-kernel launch of a sweep at a given shape, with
BGMM_DEBUG_FLAGS variants (parts of the kernel switched off) applied to that launch only."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata
N, D, K = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
flags = sys.argv[4:] or ["32"]
X, zt = gendata.synth_mixture(N, D, K, seed=11)
m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
rs = np.random.RandomState(0)
us = rs.random_sample((3, N))
for f in flags:
    os.environ["BGMM_DEBUG_FLAGS"] = "0"
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
    ctx.set_assignments(zt)
    for it in range(2):
        ctx.sweep(us[it], None, None)
    os.environ["BGMM_DEBUG_FLAGS"] = f
    ctx.set_kernel_timing(True)
    sys.stderr.write("flags %s: " % f)
    sys.stderr.flush()
    ctx.sweep(us[2], None, None)
    import ctypes
    out = (ctypes.c_int64 * 16)()
    ctx.L.bgmm_debug_prof.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    ctx.L.bgmm_debug_prof(ctx.h, out)
    v = list(out)
    print('probe ticks(10ns): blk0', [v[i] - v[0] for i in range(1, 5)], 'lvl1+kept+nkept', v[6], v[5], v[7], 'blk1000', [v[8 + i] - v[8] for i in range(1, 5)], v[14], v[13], v[15])
    ctx.close()

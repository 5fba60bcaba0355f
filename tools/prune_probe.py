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
    print('probe ticks(10ns): blk0 start->staged %d ->homes done %d ->end %d listed %d | blk600 %d %d %d listed %d' % (v[1]-v[0], v[2]-v[0], v[4]-v[0], v[5], v[9]-v[8], v[10]-v[8], v[12]-v[8], v[13]))
    if int(f) & 2048:
        print('block start ticks rel. to block 0: b255 %d b256 %d b511 %d b512 %d b700 %d b1023 %d; b1023 end %d' % tuple(v[i] - v[8] for i in (9, 10, 11, 12, 13, 14, 15)))
    ctx.close()

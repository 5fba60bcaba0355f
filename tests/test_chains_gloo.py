"""The N>1 path on CPU: world_size-2 gloo run of the label / log-marginal gather."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, %r)
from pybgmm_amd.chains import chain_rngs, gather_chains
dist.init_process_group("gloo")
rank, G = dist.get_rank(), dist.get_world_size()
rng, nprng = chain_rngs(100, rank)
z = nprng.randint(0, 7, 1000).astype(np.int64)          # stands in for a chain's final labels
lm = np.array([rng.random() for _ in range(5)])
Z, LM = gather_chains(z, lm)
assert Z.shape == (G, 1000) and LM.shape == (G, 5)
for c in range(G):
    r2, n2 = chain_rngs(100, c)
    assert np.array_equal(Z[c], n2.randint(0, 7, 1000))
    assert np.array_equal(LM[c], [r2.random() for _ in range(5)])
dist.barrier()
if rank == 0:
    print("GATHER_OK", G)
dist.destroy_process_group()
""" % ROOT


def test_gather_two_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
                          "--nproc-per-node=2", "--master-addr", "127.0.0.1", "--master-port",
                          "29611", str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "GATHER_OK 2" in out.stdout


def test_gather_single_process_fallback():
    from pybgmm_amd.chains import gather_chains
    Z, LM = gather_chains(np.arange(10), np.array([1.0, 2.0]))
    assert Z.shape == (1, 10) and LM.shape == (1, 2)

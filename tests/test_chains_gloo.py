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


def test_lockstep_rendezvous_of_sampler_loops_without_a_gpu(monkeypatch):
    """chains.run_chains_on_device's host logic (many chains per GPU): G sampler loops in threads meet at a barrier in
    IGMM._sweep, the last one to arrive sweeps the whole group ONCE per round with every chain's exponent, and a chain that
    fails breaks the barrier for the others instead of leaving them waiting.  The device is replaced by a stub."""
    import threading
    from pybgmm_amd import chains, _lib

    calls = []

    class Ctx(object):
        pass

    class Comp(object):
        def __init__(self):
            self._ctx = Ctx()

    class Model(object):
        def __init__(self, *a, **kw):
            self.components = Comp()
            self.rounds = 0
            self.fail_at = None

        def collapsed_gibbs_sampler(self, n_iter, true_assignments, num_saved=0):
            for it in range(n_iter):
                if self.fail_at == it:
                    raise ValueError("chain failed")
                self._lockstep.sweep(self, 1.0 + 0.01 * it)
                self.rounds += 1
            return {"rounds": self.rounds}, None

    def fake_group_sweep(ctxs, powers, raise_errors=True):
        calls.append((len(ctxs), tuple(powers), threading.current_thread().name))
        return [0] * len(ctxs)
    monkeypatch.setattr(_lib, "group_sweep_staged", fake_group_sweep)
    out = chains.run_chains_on_device(Model, None, None, 1.0, 6, 5, seed=3)
    assert len(out) == 6 and all(rec["rounds"] == 5 for _, rec in out)
    assert len(calls) == 5                                        # one group sweep per round, not one per chain
    assert [c[0] for c in calls] == [6] * 5
    assert [c[1] for c in calls] == [tuple([1.0 + 0.01 * it] * 6) for it in range(5)]
    # a failing chain: its exception comes out, nobody hangs
    made = []

    class Failing(Model):
        def __init__(self, *a, **kw):
            Model.__init__(self, *a, **kw)
            made.append(self)
            if len(made) == 3:
                self.fail_at = 2
    import pytest
    with pytest.raises(ValueError):
        chains.run_chains_on_device(Failing, None, None, 1.0, 4, 5, seed=3)

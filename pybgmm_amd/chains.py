"""
Independent chains, one per GPU ("replicas only").

A collapsed-Gibbs chain is sequential in the datapoints, so one chain does not
shard; the multi-GPU mode runs G independent chains, chain c on GPU c with seeds
``seed + c``, and exchanges nothing until the end, when ONE collective gathers the
final labels (int64[N] per chain) and the per-sweep log marginals.  With
``torch.distributed`` backend "nccl" that collective is RCCL over xGMI; the CPU
tests run the identical code over "gloo".

Many chains on ONE GPU (``chains=``; 8e): a sweep of a chain that still moves is one workgroup's
chain of dependent draws -- the whole sweep at D <= 4 (kernels_seq.hip), the window resolver at larger D (kernels_gram.hip) --
and a GPU has 256 compute units.  ``ChainGroup`` keeps G contexts on one device and sweeps them side by side through
``bgmm_group_sweep_staged`` (D <= 4: two launches for all of them; any other shape: the chains run concurrently inside the
call and share their launches while they burn in together: 8 chains of BASELINE's C4 shape from "rand" at 4.5 - 5.3 x one
chain's rate); ``run_chains_on_device``
does the same for G model objects (CRPMM / PCRPMM / ADAPCRPMM), each with its own seeded generators, whose sampler
loops run in lockstep.  Chain c is label for label the chain a solo run with seed ``seed + c`` produces.
"""
import random
import threading

import numpy as np


def chain_rngs(seed, rank):
    """Per-chain generators whose streams equal the global ones after
    ``random.seed(seed + rank); np.random.seed(seed + rank)``."""
    return random.Random(seed + rank), np.random.RandomState(seed + rank)


def gather_chains(z_local, log_marg_local, device=None):
    """
    All-gather of the final labels and per-sweep log marginals of every chain.
    Returns ``(z[G, N] int64, log_marg[G, n_iter] float64)`` as numpy arrays on every
    rank.  Falls back to a single-chain stack when ``torch.distributed`` is not
    initialised (G = 1).
    """
    import torch
    import torch.distributed as dist

    z_local = np.ascontiguousarray(z_local, dtype=np.int64)
    lm_local = np.ascontiguousarray(log_marg_local, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return z_local[None, :], lm_local[None, :]
    G = dist.get_world_size()
    dev = torch.device("cpu") if device is None else device
    zt = torch.from_numpy(z_local).to(dev)
    lt = torch.from_numpy(lm_local).to(dev)
    # flat concatenated outputs: the form every backend (RCCL and gloo) accepts
    z_all = torch.empty(G * zt.numel(), dtype=zt.dtype, device=dev)
    l_all = torch.empty(G * lt.numel(), dtype=lt.dtype, device=dev)
    dist.all_gather_into_tensor(z_all, zt.reshape(-1))
    dist.all_gather_into_tensor(l_all, lt.reshape(-1))
    return (z_all.cpu().numpy().reshape((G,) + z_local.shape),
            l_all.cpu().numpy().reshape((G,) + lm_local.shape))


def gather_labels_rccl(ctx, rank, world_size, device_index, exchange_id):
    """
    The same label gather through the C-ABI alone (``bgmm_comm_*`` / ``bgmm_gather_labels``, include/bgmm.h): for
    hosts that have no ``torch.distributed``.  ``exchange_id(id_bytes_or_None) -> id_bytes`` ships rank 0's 128
    bytes to every rank (rank 0 passes them in, the others pass None).  Returns int64[world_size, N].
    """
    from . import _lib
    ident = exchange_id(_lib.Comm.unique_id() if rank == 0 else None)
    comm = _lib.Comm(rank, world_size, ident, device=device_index)
    try:
        return ctx.gather_labels(comm, world_size)
    finally:
        comm.close()


def run_chain(model_cls, X, prior, alpha, n_iter, seed, rank, device_index, true_assignments=None,
              assignments="rand", K=1, K_max=None, sampler_kwargs=None):
    """Build chain ``rank`` on GPU ``device_index`` with its own seeded generators and
    run it; returns ``(model, record_dict)``."""
    rng, nprng = chain_rngs(seed, rank)
    model = model_cls(X, prior, alpha, None, assignments=assignments, K=K, K_max=K_max,
                      device=device_index, rng=rng, nprng=nprng)
    record, _ = model.collapsed_gibbs_sampler(n_iter, true_assignments, num_saved=0,
                                              **(sampler_kwargs or {}))
    return model, record


class ChainGroup(object):
    """G chains of one data set on one device at the C-ABI level: contexts, per-chain ``random.Random(seed + c)`` whose
    streams the device continues (``bgmm_stage_mt19937``), sweeps side by side."""

    def __init__(self, X, m_0, k_0, v_0, S_0, alpha, K_max, n_chains, seed=0, device=0, cov_type="full"):
        from . import _lib
        self._lib = _lib
        # (one copy of X on the device for the whole group: the chains after the first borrow the first one's)
        self.ctxs = []
        for _ in range(int(n_chains)):
            self.ctxs.append(_lib.Context(X, m_0, k_0, v_0, S_0, alpha, K_max, device=device, cov_type=cov_type,
                                          share_with=self.ctxs[0] if self.ctxs else None))
        self.rngs = [random.Random(seed + c) for c in range(int(n_chains))]
        self._keys = []
        for r in self.rngs:
            _, key, _ = r.getstate()
            self._keys.append([np.asarray(key[:-1], dtype=np.uint32), int(key[-1])])

    def set_assignments(self, zs):
        for ctx, z in zip(self.ctxs, zs):
            ctx.set_assignments(z)

    def sweep(self, orders=None, powers=None):
        """One sweep of every chain: chain c's uniforms are the next N values of its generator."""
        for c, ctx in enumerate(self.ctxs):
            key, pos = ctx.stage_mt19937(self._keys[c][0], self._keys[c][1], None if orders is None else orders[c])
            self._keys[c] = [key, pos]
        self._lib.group_sweep_staged(self.ctxs, powers)

    def sync_rngs(self):
        """Writes the generator states the device has reached back into ``self.rngs``."""
        for r, (key, pos) in zip(self.rngs, self._keys):
            version, _, gauss = r.getstate()
            r.setstate((version, tuple(key.tolist()) + (int(pos),), gauss))

    def assignments(self):
        return np.stack([ctx.assignments() for ctx in self.ctxs])

    def close(self):
        for ctx in self.ctxs:
            ctx.close()
        self.ctxs = []


class _Lockstep(object):
    """Rendezvous of G sampler loops: every chain stages its own sweep inputs, the last one to arrive sweeps the whole
    group, all of them carry on.  A chain that fails breaks the barrier so that nobody waits for it."""

    def __init__(self, models):
        from . import _lib
        self._lib = _lib
        self.models = list(models)
        self.powers = [None] * len(self.models)
        self.error = None
        self.rcs = [0] * len(self.models)
        self.barrier = threading.Barrier(len(self.models), action=self._sweep_all)

    def _sweep_all(self):
        try:
            self.rcs = self._lib.group_sweep_staged([m.components._ctx for m in self.models], self.powers, raise_errors=False)
        except Exception as e:          # (raised again in every chain's thread below)
            self.error = e

    def sweep(self, model, power):
        """Chain ``model``'s part of the round.  A status of its OWN sweep is raised in its own thread only (the others
        finished theirs): IGMM._sweep treats BGMM_EKMAX of a chain whose slots grow on demand as the solo path does."""
        idx = self.models.index(model)
        self.powers[idx] = power
        self.barrier.wait()
        if self.error is not None:
            raise self.error
        if self.rcs[idx] != 0:
            model.components._ctx._ck(self.rcs[idx])

    def leave(self):
        self.barrier.abort()


def run_chains_on_device(model_cls, X, prior, alpha, n_chains, n_iter, seed=0, device_index=0, true_assignments=None,
                         assignments="rand", K=1, K_max=None, covariance_type="full", sampler_kwargs=None):
    """
    ``n_chains`` independent chains of ``model_cls`` on ONE GPU, chain c seeded ``seed + c`` (its own ``random.Random``
    and ``RandomState``: the streams a solo run under ``random.seed(seed + c); np.random.seed(seed + c)`` consumes),
    their ``collapsed_gibbs_sampler`` loops in lockstep so that every round of sweeps is one group call
    (``bgmm_group_sweep_staged``).  Returns ``[(model, record_dict), ...]``; ``record_dict["sample_time"]`` of a chain is
    the time of the ROUND it took part in.  Worth it where a sweep cannot fill the GPU on its own: D <= 4 in every regime,
    any dimension while the chains still move (burn-in, overlapping clusters); chains at rest at D >= 12 fill the GPU alone.
    """
    models = []
    from . import _lib
    X = np.ascontiguousarray(X, dtype=np.float64)          # (one array, so that the chains' contexts can share its device copy)
    for c in range(int(n_chains)):
        rng, nprng = chain_rngs(seed, c)
        with _lib.share_x_with(models[0].components._ctx if models else None):
            models.append(model_cls(X, prior, alpha, None, assignments=assignments, K=K, K_max=K_max,
                                    covariance_type=covariance_type, device=device_index, rng=rng, nprng=nprng))
    step = _Lockstep(models)
    out = [None] * len(models)
    errors = []

    def work(c):
        m = models[c]
        m._lockstep = step
        try:
            rec, _ = m.collapsed_gibbs_sampler(n_iter, true_assignments, num_saved=0, **(sampler_kwargs or {}))
            out[c] = (m, rec)
        except threading.BrokenBarrierError:
            pass                        # (another chain failed: its error is the one reported)
        except Exception as e:
            errors.append(e)
            step.leave()
        finally:
            m._lockstep = None
    threads = [threading.Thread(target=work, args=(c,)) for c in range(len(models))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    return out

"""
Independent chains, one per GPU ("replicas only").

A collapsed-Gibbs chain is sequential in the datapoints, so one chain does not
shard; the multi-GPU mode runs G independent chains, chain c on GPU c with seeds
``seed + c``, and exchanges nothing until the end, when ONE collective gathers the
final labels (int64[N] per chain) and the per-sweep log marginals.  With
``torch.distributed`` backend "nccl" that collective is RCCL over xGMI; the CPU
tests run the identical code over "gloo".
"""
import random

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

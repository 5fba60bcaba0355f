"""What runs on more than one host thread, under ThreadSanitizer:   tools/build_tsan.sh, then this script with
the sanitizer's runtime preloaded -- see the header of tools/build_tsan.sh.  Three exercises, each checked for its result as
the GPU suite checks it (four since the second half of round 6: chains with few movers share their safe-stay steps):
  1. permutations in flight: the context's worker thread queues generations while the caller's thread takes them, with
     foreign draws in between (drains) and eras that run out (restarts);
  2. chains side by side from a random start: a host thread per chain, the rendezvous that shares their frozen-factor
     launches (GramCombiner), against the chains' solo runs;
  3. a chain with overlapping clusters: the dense proof pass's look-ahead on its second stream, against the look-ahead off.
"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pybgmm_amd import _build                                     # noqa: E402
_build.LIB = os.path.join(os.path.dirname(_build.LIB), "libbgmm_hip_tsan.so")
from pybgmm_amd import _lib                                       # noqa: E402
from pybgmm_amd.utils import gendata                              # noqa: E402


def permutations():
    for N in (4096, 200000):
        X, zt = gendata.synth_mixture(N, 2, 3, seed=1)
        ctx = _lib.Context(X, *gendata.demo_prior_params(2), 1.0, 12)
        ctx.set_assignments(zt)
        host = np.random.RandomState(N)
        key, pos = host.get_state()[1].copy(), int(host.get_state()[2])
        for it in range(40):
            if it in (7, 8, 23):
                host.random_sample(3)
                key, pos = host.get_state()[1].copy(), int(host.get_state()[2])
            expect = host.permutation(N)
            key, pos = ctx.stage_permutation_mt19937(key, pos)
            assert np.array_equal(ctx.staged_order(), expect), (N, it)
        print("permutations N=%d: %s %s" % (N, ctx.permutation_stats(), ctx.permutation_pipe_state()), flush=True)
        ctx.close()


def chains():
    N, D, K, G = 6000, 64, 10, 8
    X, zt = gendata.synth_mixture(N, D, K, seed=21)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)

    def build(c):
        ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 8 * K)
        ctx.set_assignments(np.unique(np.random.RandomState(300 + c).randint(0, K, N), return_inverse=True)[1])
        _, key, _ = random.Random(500 + c).getstate()
        return [ctx, np.asarray(key[:-1], dtype=np.uint32), int(key[-1])]
    solo, grp = [build(c) for c in range(G)], [build(c) for c in range(G)]
    for it in range(2):
        for s in solo:
            s[1], s[2] = s[0].stage_mt19937(s[1], s[2], None)
            s[0].sweep_staged(None)
        for g in grp:
            g[1], g[2] = g[0].stage_mt19937(g[1], g[2], None)
        _lib.group_sweep_staged([g[0] for g in grp], None)
        for c in range(G):
            assert np.array_equal(solo[c][0].assignments(), grp[c][0].assignments()), (it, c)
    print("chains side by side: %d chains equal their solo runs" % G, flush=True)
    for s in solo + grp:
        s[0].close()


def lookahead():
    N, D, K = 60000, 64, 40
    X, zt = gendata.synth_mixture(N, D, K, seed=141, mu_scale=0.5)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    us = np.random.RandomState(8).random_sample((2, N))
    out = []
    for ahead in (0, 2048):
        ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
        ctx.set_proof_pass(1)
        ctx.set_proof_lookahead(ahead)
        ctx.set_assignments(zt)
        for it in range(2):
            ctx.sweep(us[it])
        out.append(ctx.assignments())
        print("look-ahead %d: %s" % (ahead, ctx.proof_lookahead_stats()), flush=True)
        ctx.close()
    assert np.array_equal(out[0], out[1])


def chains_with_few_movers():
    """Round 6: chains whose clusters overlap share the launches of their safe-stay steps (the rendezvous' kind 1, the
    look-ahead's second stream driven by the leader)."""
    N, D, K, G = 60000, 64, 40, 4
    X, zt = gendata.synth_mixture(N, D, K, seed=141, mu_scale=0.5)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)

    def build(c):
        ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K)
        ctx.set_assignments(zt)
        _, key, _ = random.Random(40 + c).getstate()
        return [ctx, np.asarray(key[:-1], dtype=np.uint32), int(key[-1])]
    solo, grp = [build(c) for c in range(G)], [build(c) for c in range(G)]
    for it in range(3):
        for s in solo:
            s[1], s[2] = s[0].stage_mt19937(s[1], s[2], None)
            s[0].sweep_staged(None)
        for g in grp:
            g[1], g[2] = g[0].stage_mt19937(g[1], g[2], None)
        _lib.group_sweep_staged([g[0] for g in grp], None)
        for c in range(G):
            assert np.array_equal(solo[c][0].assignments(), grp[c][0].assignments()), (it, c)
    print("chains with few movers side by side: %d chains equal their solo runs, %s" % (G, grp[0][0].group_stats()), flush=True)
    for s in solo + grp:
        s[0].close()


if __name__ == "__main__":
    permutations()
    chains()
    chains_with_few_movers()
    lookahead()
    print("TSAN RUN DONE", flush=True)

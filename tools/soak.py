"""Soak test: the default configuration (certified stays + exact pruning + lean steps + resolver; every fourth case the
benchmarked mode, certified stays off with its short steps) against the plain full evaluation (prune_mode 1, resolver
off) over many sweeps, shapes, separations, visiting orders, seating exponents and hand-made state changes.  The first
context takes its inputs the way the classes do -- the uniforms of a random.Random and the permutations of a RandomState
continued ON THE DEVICE (look-ahead batches, with the caller drawing from its generators in between now and then) --
the second one from twin generators on the host.  Any difference in the label trajectory is a bug.
    python tools/soak.py [n_cases [seed]]
(`certified` counts certificates issued: rows behind a mover are examined again by the next window.)"""
import os, random, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata, rng as _rng

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 2024)
bad = 0
t0 = time.time()
for case in range(n_cases):
    D = int(rs.choice([12, 16, 32, 64, 96]))
    K = int(rs.choice([5, 20, 60]))
    N = int(rs.choice([8000, 30000, 70000]))
    sep = float(rs.choice([1.0, 1.8, 2.5, 4.0]))
    cov = str(rs.choice(["full", "full", "diag", "fixed"]))
    pcrp = bool(rs.randint(2))
    X, zt = gendata.synth_mixture(N, D, K, seed=1000 + case, mu_scale=sep)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    if cov == "diag":
        S_0 = np.ascontiguousarray(np.diag(S_0))
    elif cov == "fixed":
        m_0, k_0, v_0 = np.zeros(D), 1.0, 1
        S_0 = np.concatenate([np.full(D, 0.49), np.full(D, 16.0)])
    z0 = zt.copy()
    nflip = int(rs.choice([0, 0, 50, 2000]))
    if nflip:
        flip = rs.choice(N, size=min(nflip, N // 4), replace=False)
        z0[flip] = rs.randint(0, K, size=flip.size)
    if rs.randint(4) == 0:
        z0[rs.rand(N) < 0.01] = -1
    K_max = 4 * K + 16
    ctxs = []
    for mode in (0, 1):
        c = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, K_max, cov_type=cov)
        # (the default configuration keeps pruned windows for long mover-free stretches; every third
        # case forces them in every regime so that the pruning / certification kernels see moving chains)
        c.set_tuning(prune_mode=(2 if case % 3 == 0 else (3 if case % 4 == 1 else 0)) if mode == 0 else 1,
                     resolver_mode=1 if mode == 1 else 0)
        c.set_assignments(z0)
        ctxs.append(c)
    n_sweeps = 10
    ok = True
    cert = 0
    # twin generators: the device continues (dev_r, dev_np), the host draws from (host_r, host_np)
    dev_r, host_r = random.Random(case), random.Random(case)
    dev_np, host_np = np.random.RandomState(case), np.random.RandomState(case)
    ctxs[0].set_mt_lookahead(int(rs.choice([-1, -1, 1, 3, 0])))
    for it in range(n_sweeps):
        if rs.randint(5) == 0:                      # the caller draws from its own generators between two sweeps
            k = int(rs.randint(1, 5))
            assert [dev_r.random() for _ in range(k)] == [host_r.random() for _ in range(k)]
            assert np.array_equal(dev_np.random_sample(k), host_np.random_sample(k))
        u = _rng.take_uniforms(N, host_r)
        order = host_np.permutation(N) if pcrp else None
        power = (1.0 + 0.02 * rs.rand()) if (pcrp and it > 0) else None
        edit = None
        if it in (4, 7) and rs.randint(2):
            ii = rs.choice(N, size=20, replace=False)
            edit = [(int(i), int(rs.randint(0, ctxs[0].K))) for i in ii]
        for ci, c in enumerate(ctxs):
            if edit:
                for i, lab in edit:
                    c.del_item(i)
                    c.add_item(i, min(lab, c.K))
            if ci == 0:
                dev_order = _rng.take_permutation_staged(c, N, dev_np) if pcrp else None
                if dev_order is _rng.STAGED:
                    dev_order = None
                assert _rng.stage_uniforms_on_device(c, dev_order, dev_r)
                c.sweep_staged(power)
            else:
                c.sweep(u, order, power)
        if dev_r.getstate() != host_r.getstate() or not np.array_equal(dev_np.get_state()[1], host_np.get_state()[1]):
            print("GENERATOR STATE case %d sweep %d: the device left a generator elsewhere than the host" % (case, it))
            ok = False
            break
        za, zb = ctxs[0].assignments(), ctxs[1].assignments()
        cert += ctxs[0].prune_stats()["certified_visits"]
        if not np.array_equal(za, zb):
            d = np.nonzero(za != zb)[0]
            print("MISMATCH case %d sweep %d: %d labels differ, first at %d" % (case, it, d.size, d[0]))
            ok = False
            break
        la, lb = ctxs[0].log_marg(), ctxs[1].log_marg()
        if abs(la - lb) > 1e-9 * abs(lb):
            print("LOG_MARG case %d sweep %d: %r vs %r" % (case, it, la, lb))
            ok = False
            break
    if not ok or case % 10 == 0:
      print("case %2d  N=%6d D=%3d K=%2d sep=%.1f cov=%-5s pcrp=%d flips=%4d  K_end=%3d  certified %.0f%%  %s" % (
        case, N, D, K, sep, cov, pcrp, nflip, ctxs[0].K, 100.0 * cert / (n_sweeps * N), "ok" if ok else "FAILED"))
    bad += 0 if ok else 1
    for c in ctxs:
        c.close()
print("%d cases, %d failed, %.0f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)

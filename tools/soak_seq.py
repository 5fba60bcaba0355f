"""Soak test of the sequential small-D sweep (kernels_seq.hip: sweep_seq_kernel): default tuning (which
takes it for D <= 4, full covariance) against the windowed VALU path with the resolver off, over random
shapes, separations, visiting orders, seating exponents, unassigned points and hand-made state changes.
A third of the cases run with a small LDS plan (Context.set_seq_plan) so that the kernel hands over to the
windowed kernels mid-sweep.  Any difference in the label trajectory is a bug.
    python tools/soak_seq.py [n_cases [seed]]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pybgmm_amd import _lib
from pybgmm_amd.utils import gendata

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rs = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
bad = 0
t0 = time.time()
for case in range(n_cases):
    D = int(rs.choice([1, 2, 2, 3, 4]))
    K = int(rs.choice([3, 8, 20, 70]))
    N = int(rs.choice([500, 5000, 20000]))
    sep = float(rs.choice([0.5, 1.5, 3.0, 6.0]))
    pcrp = bool(rs.randint(2))
    X, zt = gendata.synth_mixture(N, D, K, seed=3000 + case, mu_scale=sep)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    init = int(rs.randint(4))
    if init == 0:
        z0 = zt.copy()
    elif init == 1:
        z0 = np.zeros(N, dtype=np.int64)                 # one table
    elif init == 2:
        z0 = rs.randint(0, K, size=N)
        z0[:K] = np.arange(K)
    else:
        z0 = zt.copy()
        z0[rs.rand(N) < 0.05] = -1
    small_plan = rs.randint(3) == 0
    cap = int(z0.max() + 1 + rs.randint(1, 4)) if small_plan else 0
    K_max = 4 * K + 40
    alpha = float(rs.choice([0.3, 1.0, 5.0]))
    ctxs = []
    for mode in (0, 1):
        c = _lib.Context(X, m_0, k_0, v_0, S_0, alpha, K_max)
        c.set_tuning(kernel_kind=0 if mode == 0 else 1, resolver_mode=0 if mode == 0 else 1,
                     prune_mode=0 if mode == 0 else 1)
        c.set_assignments(z0)
        ctxs.append(c)
    n_sweeps = 8
    ok = True
    for it in range(n_sweeps):
        u = rs.random_sample(N)
        order = rs.permutation(N) if pcrp else None
        power = (1.0 + 0.02 * rs.rand()) if (pcrp and it > 0) else None
        edit = None
        if it in (3, 6) and rs.randint(2):
            ii = rs.choice(N, size=10, replace=False)
            edit = [(int(i), int(rs.randint(0, ctxs[0].K))) for i in ii]
        for k, c in enumerate(ctxs):
            if edit:
                for i, lab in edit:
                    c.del_item(i)
                    c.add_item(i, min(lab, c.K))
            if k == 0 and cap:
                c.set_seq_plan(cap)
            c.sweep(u, order, power)
        za, zb = ctxs[0].assignments(), ctxs[1].assignments()
        if not np.array_equal(za, zb):
            dd = np.nonzero(za != zb)[0]
            print("MISMATCH case %d sweep %d: %d labels differ, first at %d" % (case, it, dd.size, dd[0]))
            ok = False
            break
        la, lb = ctxs[0].log_marg(), ctxs[1].log_marg()
        if abs(la - lb) > 1e-9 * abs(lb):
            print("LOG_MARG case %d sweep %d: %r vs %r" % (case, it, la, lb))
            ok = False
            break
        la, lb = ctxs[0].sweep_stats()["lik_evals"], ctxs[1].sweep_stats()["lik_evals"]
        if la != lb:
            print("LIK_EVALS case %d sweep %d: %d vs %d" % (case, it, la, lb))
            ok = False
            break
        ma, mb = ctxs[0].stats(want_inv=False), ctxs[1].stats(want_inv=False)
        if not (np.array_equal(ma[0], mb[0]) and np.array_equal(ma[1], mb[1])):
            print("STATS case %d sweep %d: sufficient statistics differ" % (case, it))
            ok = False
            break
    if not ok or case % 6 == 0:
        print("case %2d  N=%6d D=%d K=%2d sep=%.1f pcrp=%d init=%d cap=%3d  K_end=%3d  moves(last)=%d  %s" % (
            case, N, D, K, sep, pcrp, init, cap, ctxs[0].K, ctxs[0].sweep_stats()["moves"],
            "ok" if ok else "FAILED"))
    bad += 0 if ok else 1
    for c in ctxs:
        c.close()
print("%d cases, %d failed, %.0f s" % (n_cases, bad, time.time() - t0))
sys.exit(1 if bad else 0)

"""
Development probes of the HIP path, one script (run on a GPU box, e.g. through gpurun).

    python tools/probe.py chain N D K [options]     per-sweep cost of a chain and what each layer did
        --init true|rand|flip|own   the truth / the reference's "rand" start / the truth with N/500 (or --flip n) wrong
                                    labels / each point in its own component                      (default true)
        --sweeps n                  (default 6)          --sep s      mu_scale of the data (default 4.0; < 1: overlapping)
        --pcrp                      a fresh permutation per sweep, powered weights from the second sweep on
        --cov full|diag|fixed       --prune m  --resolver m  --home m  --kernel m  --window w     set_tuning / set_home_pass
        --budget b                  set_safe_budget          --seq-plan k   set_seq_plan
        --oracle                    every sweep compared with the C port of the reference (labels, log marginal)
        --timing                    HIP-event time of the dominant likelihood launches of the LAST sweep
        --prof                      load libbgmm_hip_prof.so (tools/build_prof.sh / build_prof_seq.sh): per-phase shader
                                    clocks of home_kernel / sweep_seq_kernel / the resolver
    python tools/probe.py safe-check                forced safe-stay windows against the C oracle on six small problems
    python tools/probe.py classes N D K [--profile] end-to-end cost of the user-facing classes (RNG, sweep, record dict)
    python tools/probe.py chains D G [N K]          G chains of one shape side by side on one device (ChainGroup) vs one
    python tools/probe.py staged [N D K n]          a chain at rest, certified stays off, every sweep's uniforms continued on
                                                    the device: us per stage + sweep, the stage call's share, look-ahead hits
                                                    (under `rocprofv3 --kernel-trace` + tools/timeline.py: the GPU timeline)
    python tools/probe.py perm [N ...]              np.random.permutation(N) on the device (bgmm_stage_permutation_mt19937)
                                                    against numpy on this host: ms per permutation
    python tools/probe.py gather                    torch index_select of C4's rows: what a random row gather costs

Replaces the round-1/2 scripts burnin_probe, c3_probe, cert_probe, gram_probe, home_probe, prune_probe,
recovery_probe, safe_probe, seq_probe, class_api_probe, class_api_profile, gather_probe.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def prior_for(cov, D):
    from pybgmm_amd.utils import gendata
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    if cov == "diag":
        S_0 = np.ascontiguousarray(np.diag(S_0))
    elif cov == "fixed":
        m_0, k_0, v_0 = np.zeros(D), 1.0, 1
        S_0 = np.concatenate([np.full(D, 0.49), np.full(D, 16.0)])
    return m_0, k_0, v_0, S_0


def start_labels(init, zt, K, rs, flip=0):
    N = len(zt)
    if init == "rand":
        return np.unique(rs.randint(0, K, N), return_inverse=True)[1]
    if init == "own":
        return np.arange(N)
    z0 = zt.copy()
    if init == "flip":
        idx = rs.choice(N, size=flip or max(N // 500, 1), replace=False)
        z0[idx] = rs.randint(0, K, size=idx.size)
    return z0


def chain(a, quiet=False):
    if a.prof or a.lib:
        from pybgmm_amd import _build
        _build.LIB = os.path.join(os.path.dirname(_build.LIB), "libbgmm_hip_%s.so" % (a.lib or "prof"))
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    N, D, K = a.N, a.D, a.K
    X, zt = gendata.synth_mixture(N, D, K, seed=a.seed, mu_scale=a.sep)
    m_0, k_0, v_0, S_0 = prior_for(a.cov, D)
    rs = np.random.RandomState(a.seed)
    z0 = start_labels(a.init, zt, K, rs, a.flip)
    K_max = max(4 * K, N if a.init == "own" else 0)
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, K_max, cov_type=a.cov)
    ctx.set_tuning(max_window=a.window, kernel_kind=a.kernel, resolver_mode=a.resolver, prune_mode=a.prune)
    ctx.set_home_pass(a.home)
    if a.budget:
        ctx.set_safe_budget(a.budget)
    if a.ahead >= 0:
        ctx.set_proof_lookahead(a.ahead)
    if a.seq_plan:
        ctx.set_seq_plan(a.seq_plan)
    ctx.set_assignments(z0)
    o = None
    if a.oracle:
        from oracle import c_oracle
        o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, z0, K_max, scipy_tables=False, cov_type=a.cov)
    ok = True
    pc0 = np.array(ctx.phase_clocks())
    for it in range(a.sweeps):
        u = rs.random_sample(N)
        order = rs.permutation(N).astype(np.int64) if a.pcrp else None
        power = 1.01 if (a.pcrp and it > 0) else None
        last = it == a.sweeps - 1
        if last:
            pc0 = np.array(ctx.phase_clocks())
            if a.timing:
                ctx.set_kernel_timing(True)
        ctx.stage(u, order); ctx.synchronize()
        t0 = time.time(); ctx.sweep_staged(power); ctx.synchronize(); dt = time.time() - t0
        st, ss, ps, pr = ctx.sweep_stats(), ctx.safe_stats(), ctx.path_stats(), ctx.prune_stats()
        msg = ""
        if o is not None:
            o.sweep(u, order, power)
            bad = np.nonzero(ctx.assignments() != o.z)[0]
            lm, lo = ctx.log_marg(), o.log_marg()
            msg = "  ORACLE %s, log_marg rel %.1e" % ("ok" if bad.size == 0 else "DIFFERS at %d labels, first i=%d" % (bad.size, bad[0]),
                                                      abs(lm - lo) / abs(lo))
            ok = ok and bad.size == 0
        if not quiet:
            print("sweep %2d: %9.3f ms moves %7d (%.2f us/move) K %3d | windows %6d steps %6d | safe windows %6d examined %9d walked %7d "
                  "cuts %5d L %7d | frozen %6d (%7d visits) | certified %7d home-decided %8d bounded blocks %9d kept %8d%s" % (
                      it, dt * 1e3, st["moves"], dt * 1e6 / max(st["moves"], 1), ctx.K, st["windows"], st["steps"], ss["windows"],
                      ss["visits_examined"], ss["unproven_walked"], ss["budget_cuts"], ss["next_stretch"], ps["frozen_windows"],
                      ps["frozen_window_visits"], pr["certified_visits"], ps.get("home_decided", 0), pr["bound_blocks"],
                      pr["kept_blocks"], msg), flush=True)
        if not ok:
            break
    print("proof passes of the safe-stay windows so far: %(table_batches)d batches through the per-home tables, %(dense_batches)d dense" % ctx.proof_pass_stats())
    print("look-ahead of the dense proof pass: %s" % ctx.proof_lookahead_stats())
    if a.timing:
        n, ms = ctx.kernel_timing()
        print("last sweep: %d timed launches, avg %.4f ms" % (n, ms / max(n, 1)))
    pc = np.array(ctx.phase_clocks()) - pc0
    if not a.prof and pc[13] > 0:
        print("last sweep: gram_finish: %d windows, %.1f slots each, %.2f rebuilt from scratch per window (%.2f of them for too many terms)" % (
            pc[8], pc[13] / max(pc[8], 1), pc[14] / max(pc[8], 1), pc[15] / max(pc[8], 1)))
    if a.prof and pc[15] > 0 and D >= 12 and a.init not in ("rand",):            # home_kernel's clocks (-DBGMM_HOME_PROF)
        names = ["wait+stage", "issue", "frags", "mfma+reduce", "tail", "records", "blockhead", "switch"]
        tot = float(pc[:8].sum())
        for k in range(8):
            print("  %-12s %8.0f cycles per tile-wave  %5.1f %%" % (names[k], pc[k] / (N / 16.0), 100.0 * pc[k] / tot))
        if pc[8] > 0:
            print("  log-free tail: %d of %d wavefront-blocks; rows failing: new table %d, tables / neighbours %d, any %d of %d" % (
                pc[9], pc[8], pc[10], pc[11], pc[12], pc[13]))
    elif a.prof and D <= 4:                          # sweep_seq_kernel's clocks (-DBGMM_SEQ_PROF)
        names = ["ring read", "home lookup", "evaluate", "stats", "commit", "rebuild", "barrier 1", "barrier 2"]
        tot = float(pc[:8].sum())
        for k in range(8):
            print("   %-12s %6.1f %%   %9.0f cycles per wave and 1000 visits" % (names[k], 100.0 * pc[k] / max(tot, 1), pc[k] / 8.0 / N * 1000))
    elif a.prof and any(pc):                         # the resolver's clocks
        w = max(pc[8], 1)
        print("resolver ticks per window: prologue %.0f  draws %.0f  bookkeeping %.0f  update(home) %.0f  update(dest) %.0f  | windows %d" % (
            pc[0] / w, pc[1] / w, pc[2] / w, pc[3] / w, pc[4] / w, pc[8]))
        print("   update wave of the joined column, ticks per window: metadata (LDS) %.0f | loads + constants %.0f | row scalars + term chain %.0f | "
              "D, 1/D %.0f | q, rsqrt, log1p %.0f | exp %.0f | home-form rows %.0f (taken in %.1f moves per window); whole launch %.0f" % (
                  pc[9] / w, pc[10] / w, pc[11] / w, pc[12] / w, pc[13] / w, pc[14] / w, pc[15] / w, pc[7] / w, pc[6] / w))
    ctx.close()
    return ok


def safe_check():
    ok = True
    for (N, D, K, init, sep, pcrp) in [(3000, 16, 12, "rand", 4.0, False), (20000, 16, 40, "flip", 1.6, False),
                                       (20000, 64, 20, "flip", 4.0, False), (8000, 32, 30, "rand", 1.2, True),
                                       (30000, 16, 60, "rand", 1.0, False), (6000, 128, 10, "flip", 4.0, False)]:
        for budget in (0.0, 1.0 / 64, 2.0):
            print("== N %d D %d K %d %s sep %.1f pcrp %d forced safe-stay windows, budget %s" % (N, D, K, init, sep, pcrp, budget or "auto"), flush=True)
            a = parser().parse_args(["chain", str(N), str(D), str(K), "--init", init, "--sep", str(sep), "--resolver", "4",
                                     "--sweeps", "3", "--oracle", "--budget", str(budget), "--seed", "11"] + (["--pcrp"] if pcrp else []))
            ok = chain(a, quiet=True) and ok
    print("CHECK", "OK" if ok else "FAILED")
    return ok


def classes(a):
    import cProfile
    import pstats
    import random
    from pybgmm_amd.igmm import CRPMM
    from pybgmm_amd.prior import NIW
    from pybgmm_amd.utils import gendata
    X, zt = gendata.synth_mixture(a.N, a.D, a.K, seed=1)
    random.seed(1); np.random.seed(1)
    mm = CRPMM(X, NIW(*gendata.demo_prior_params(a.D)), 1.0, None, assignments=zt, K_max=4 * a.K)
    mm.collapsed_gibbs_sampler(2, zt, num_saved=0)
    for metrics in (True, False):
        mm.record_metrics = metrics
        t = time.time()
        rec, _ = mm.collapsed_gibbs_sampler(5, zt, num_saved=0)
        dt = (time.time() - t) / 5
        print("record_metrics=%s: %.2f ms per sweep wall (sample_time %.3f ms), nmi %.4f K %d" % (
            metrics, dt * 1e3, 1e3 * np.mean(rec["sample_time"]), rec["nmi"][-1], rec["components"][-1]))
    if a.profile:
        pr = cProfile.Profile(); pr.enable()
        mm.collapsed_gibbs_sampler(10, zt, num_saved=0)
        pr.disable()
        pstats.Stats(pr).sort_stats("cumulative").print_stats(18)


def chains(a):
    from pybgmm_amd import _lib
    from pybgmm_amd.chains import ChainGroup
    from pybgmm_amd.utils import gendata
    N, D, K, G = a.N, a.D, a.K, a.G
    X, zt = gendata.synth_mixture(N, D, K, seed=1, mu_scale=a.sep) if a.sep else gendata.synth_mixture(N, D, K, seed=1)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    for g in (1, G):
        grp = ChainGroup(X, m_0, k_0, v_0, S_0, 1.0, 8 * K, n_chains=g, seed=1)
        grp.set_assignments([zt] * g)
        if a.ahead >= 0:
            for ctx in grp.ctxs:
                ctx.set_proof_lookahead(a.ahead)
        for _ in range(a.warm):
            grp.sweep()
        t0 = time.time()
        n = a.sweeps
        for _ in range(n):
            grp.sweep()
        dt = (time.time() - t0) / n
        mv = sum(ctx.sweep_stats()["moves"] for ctx in grp.ctxs)
        print("%3d chains: %.2f ms per round of sweeps, %.2f sweeps/s aggregate (%d moves in the last round)" % (g, dt * 1e3, g / dt, mv))
        if g > 1:
            gs = [ctx.group_stats() for ctx in grp.ctxs]
            print("    batches of all chains: shared frozen-factor %d (pipelined %d), shared safe-stay %d, on their own %d; proof passes %s" % (
                sum(x["shared_frozen_factor_batches"] for x in gs), sum(x["of_them_pipelined"] for x in gs),
                sum(x["shared_safe_stay_batches"] for x in gs), sum(x["batches_on_its_own_in_a_group"] for x in gs),
                [tuple(ctx.proof_pass_stats().values()) if hasattr(ctx, "proof_pass_stats") else None for ctx in grp.ctxs][:3]))
        grp.close()


def chains_burnin(a):
    """G chains of one shape from the reference's "rand" start, side by side on one device (bgmm_group_sweep_staged: chains
    that cannot take the one-workgroup sweep run concurrently, one stream and one host thread each) against one chain:
    seconds and moves per second of the first sweeps, and every chain's labels against its solo run."""
    from pybgmm_amd.chains import ChainGroup
    from pybgmm_amd.utils import gendata
    N, D, K, G = a.N, a.D, a.K, a.G
    X, zt = gendata.synth_mixture(N, D, K, seed=1)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    z0s = [start_labels("rand", zt, K, np.random.RandomState(100 + c)) for c in range(G)]
    res = {}
    solo = []
    for g in (1, G):
        grp = ChainGroup(X, m_0, k_0, v_0, S_0, 1.0, 4 * K, n_chains=g, seed=1)
        grp.set_assignments(z0s[:g])
        for ctx in grp.ctxs:
            ctx.synchronize()
        ts, mv = [], []
        for it in range(a.sweeps):
            t0 = time.time(); grp.sweep(); dt = time.time() - t0
            ts.append(dt); mv.append(sum(ctx.sweep_stats()["moves"] for ctx in grp.ctxs))
        res[g] = (ts, mv)
        print("%2d chain(s): sweeps %s s, moves %s -> first sweep %.3g moves/s aggregate" % (
            g, " / ".join("%.3f" % t for t in ts), " / ".join(str(m) for m in mv), mv[0] / ts[0]), flush=True)
        print("   pipelined batches / breaks per chain: %s" % " ".join("%d/%d" % (s_["batches"], s_["breaks"]) for s_ in
                                                                       (ctx.window_pipeline_stats() for ctx in grp.ctxs)))
        if g == 1:
            solo.append(grp.assignments()[0])
        else:
            zg = grp.assignments()
            ok0 = np.array_equal(zg[0], solo[0])
            print("   chain 0 of the group == the solo chain: %s" % ok0)
            if a.check:
                for c in range(1, G):
                    one = ChainGroup(X, m_0, k_0, v_0, S_0, 1.0, 4 * K, n_chains=1, seed=1 + c)
                    one.set_assignments([z0s[c]])
                    for it in range(a.sweeps):
                        one.sweep()
                    print("   chain %d of the group == its solo run: %s" % (c, np.array_equal(one.assignments()[0], zg[c])), flush=True)
                    one.close()
        grp.close()
    print("aggregate first-sweep speed-up with %d chains: %.2f x" % (G, (res[G][1][0] / res[G][0][0]) / (res[1][1][0] / res[1][0][0])))


def staged(a):
    import random
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    X, zt = gendata.synth_mixture(a.N, a.D, a.K, seed=1)
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(a.D)
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * a.K)
    ctx.set_tuning(prune_mode=3)
    ctx.set_assignments(zt)
    _, key_t, _ = random.Random(5).getstate()
    key, pos = np.asarray(key_t[:-1], dtype=np.uint32), int(key_t[-1])
    for it in range(20):
        key, pos = ctx.stage_mt19937(key, pos, None); ctx.sweep_staged(None)
    ctx.synchronize()
    t0 = time.time(); ts = 0.0
    for it in range(a.n):
        t1 = time.time(); key, pos = ctx.stage_mt19937(key, pos, None); ts += time.time() - t1
        ctx.sweep_staged(None)
    ctx.synchronize()
    print("stage + sweep: %.1f us per sweep, of which the stage call %.1f us" % ((time.time() - t0) / a.n * 1e6, ts / a.n * 1e6),
          ctx.mt_lookahead_stats(), ctx.short_step_stats())
    ctx.close()


def perm(a):
    from pybgmm_amd import _lib
    from pybgmm_amd.utils import gendata
    for N in a.N or [1000000, 2000000]:
        X, zt = gendata.synth_mixture(N, 2, 3, seed=1)
        m_0, k_0, v_0, S_0 = gendata.demo_prior_params(2)
        ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 12)
        ctx.set_assignments(zt)
        rs = np.random.RandomState(3)
        key, pos = rs.get_state()[1].copy(), int(rs.get_state()[2])
        for ahead in (0, -1):
            ctx.set_mt_lookahead(ahead)
            for _ in range(3):
                key, pos = ctx.stage_permutation_mt19937(key, pos)
            t0 = time.time()
            for _ in range(20):
                key, pos = ctx.stage_permutation_mt19937(key, pos)
            dt = (time.time() - t0) / 20
            print("N=%d look-ahead %s: device %.3f ms per permutation" % (N, "on" if ahead else "off", dt * 1e3), ctx.permutation_stats())
        t0 = time.time()
        for _ in range(5):
            rs.permutation(N)
        print("N=%d: numpy on this host %.3f ms" % (N, (time.time() - t0) / 5 * 1e3))
        ctx.close()


def gather():
    import torch
    N, D = 1000000, 64
    x = torch.randn(N, D, dtype=torch.float64, device="cuda")
    for name, idx in (("random", torch.randperm(N, device="cuda")), ("identity", torch.arange(N, device="cuda"))):
        for _ in range(3):
            x.index_select(0, idx)
        torch.cuda.synchronize()
        t0 = time.time()
        for _ in range(20):
            x.index_select(0, idx)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / 20
        print(name, "%.1f us" % (dt * 1e6), "read+write %.2f TB/s" % (2 * N * D * 8 / dt / 1e12))


def parser():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    sub = ap.add_subparsers(dest="cmd", required=True)
    c = sub.add_parser("chain")
    for n in ("N", "D", "K"):
        c.add_argument(n, type=int)
    c.add_argument("--init", default="true", choices=["true", "rand", "flip", "own"])
    c.add_argument("--flip", type=int, default=0)
    c.add_argument("--sweeps", type=int, default=6)
    c.add_argument("--sep", type=float, default=4.0)
    c.add_argument("--pcrp", action="store_true")
    c.add_argument("--cov", default="full", choices=["full", "diag", "fixed"])
    for n in ("prune", "resolver", "home", "kernel", "window", "seq-plan"):
        c.add_argument("--" + n, type=int, default=0)
    c.add_argument("--budget", type=float, default=0.0)
    c.add_argument("--seed", type=int, default=11)
    c.add_argument("--oracle", action="store_true")
    c.add_argument("--timing", action="store_true")
    c.add_argument("--prof", action="store_true")
    c.add_argument("--ahead", type=int, default=-1, help="bgmm_set_proof_lookahead: visits per chunk of the dense proof pass's look-ahead (0: off)")
    c.add_argument("--lib", default="", help="load libbgmm_hip_<name>.so (tools/build_variant.sh)")
    sub.add_parser("safe-check")
    k = sub.add_parser("classes")
    for n in ("N", "D", "K"):
        k.add_argument(n, type=int)
    k.add_argument("--profile", action="store_true")
    g = sub.add_parser("chains")
    g.add_argument("D", type=int)
    g.add_argument("G", type=int)
    g.add_argument("N", type=int, nargs="?", default=100000)
    g.add_argument("K", type=int, nargs="?", default=20)
    g.add_argument("--sep", type=float, default=0.0, help="mu_scale of the data (0.55: the steady_moving regime of bench.py)")
    g.add_argument("--sweeps", type=int, default=5)
    g.add_argument("--warm", type=int, default=2)
    g.add_argument("--ahead", type=int, default=-1, help="chunk of the proof pass's look-ahead for every chain (0: off; default: the library's)")
    gb = sub.add_parser("chains-burnin")
    for n in ("N", "D", "K", "G"):
        gb.add_argument(n, type=int)
    gb.add_argument("--sweeps", type=int, default=2)
    gb.add_argument("--check", action="store_true", help="every chain of the group against its solo run")
    t = sub.add_parser("staged")
    t.add_argument("N", type=int, nargs="?", default=1000000)
    t.add_argument("D", type=int, nargs="?", default=64)
    t.add_argument("K", type=int, nargs="?", default=200)
    t.add_argument("n", type=int, nargs="?", default=300)
    pm = sub.add_parser("perm")
    pm.add_argument("N", type=int, nargs="*")
    sub.add_parser("gather")
    return ap


if __name__ == "__main__":
    args = parser().parse_args()
    if args.cmd == "chain":
        sys.exit(0 if chain(args) else 1)
    elif args.cmd == "safe-check":
        sys.exit(0 if safe_check() else 1)
    elif args.cmd == "classes":
        classes(args)
    elif args.cmd == "chains":
        chains(args)
    elif args.cmd == "chains-burnin":
        chains_burnin(args)
    elif args.cmd == "staged":
        staged(args)
    elif args.cmd == "perm":
        perm(args)
    else:
        gather()

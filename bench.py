#!/usr/bin/env python3
"""
bench.py -- collapsed-Gibbs sweeps/s of the CRP Gaussian mixture on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, one independent chain per rank)

A "step" is one full Gibbs sweep (N_data reassignment visits) of BASELINE.json's headline
configuration, configs[3]: CRPMM, D=64, N=1e6, K~200, synthetic isotropic mixture
(SURVEY.md 8d recipe), chain initialised at the true labelling (the steady-state regime the
reference CPU numbers in BASELINE.md were taken in).  Inputs (X, the per-visit uniforms of all
timed sweeps) are resident in HBM before the timed region; the PCIe-inclusive rate is reported
separately in `extra`.  The chains are replicas: no data-path collective; one RCCL
all-gather of the final labels after the timed region.

The JSON line also carries
  roofline     -- the likelihood (score) kernel, timed with HIP events on the library's stream
  cpu_baseline -- the C oracle (a port of the reference algorithm, 1 thread) on a bounded sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (N, D, K_true, model)
    "C4": (1000000, 64, 200, "CRPMM"),
    "C3": (1000000, 16, 100, "PCRPMM"),
    "C2": (100000, 2, 20, "CRPMM"),
    "C5": (2000000, 128, 200, "PCRPMM"),
}
PEAK_FP64_MFMA_TFLOPS = 78.6      # MI355X dense FP64 matrix peak (spec); see DESIGN.md section 4
SUSTAINED_FP64_MFMA_TFLOPS = 49.0 # pure v_mfma_f64_16x16x4 loop measured on the box (tools/mfma_f64_peak.hip)
PEAK_HBM_GBPS = 8000.0


def flops_per_lik_eval(D):
    """per (visit, component) work of the predictive: 2D^2 + 3D + 12."""
    return 2.0 * D * D + 3.0 * D + 12.0


def prior_for(cov, D):
    """(m_0, k_0, v_0, S_0) of the demo scripts; fixed variance rides in the same slots."""
    from pybgmm_amd.utils import gendata
    m_0, k_0, v_0, S_0 = gendata.demo_prior_params(D)
    if cov == "diag":
        S_0 = np.ascontiguousarray(np.diag(S_0))
    elif cov == "fixed":
        m_0, k_0, v_0 = np.zeros(D), 1.0, 1
        S_0 = np.concatenate([np.full(D, 0.49), np.full(D, 16.0)])          # [var ; var_0]
    return m_0, k_0, v_0, S_0


def cpu_baseline(D, K, seed, budget_visits, cov="full"):
    """C oracle (oracle/gibbs_oracle.c, scalar, 1 thread) on a down-sized twin of the
    workload: same D, K, prior, init-at-truth; per-visit cost does not depend on N."""
    from oracle import c_oracle
    from pybgmm_amd.utils import gendata
    n_cpu = max(4 * K, budget_visits)
    X, z_true = gendata.synth_mixture(n_cpu, D, K, seed)
    m_0, k_0, v_0, S_0 = prior_for(cov, D)
    t0 = time.time()
    o = c_oracle.COracle(X, m_0, k_0, v_0, S_0, 1.0, z_true, 4 * K, scipy_tables=False, cov_type=cov)
    t_init = time.time() - t0
    u = np.random.RandomState(seed).random_sample(n_cpu)
    t0 = time.time()
    o.sweep(u, n_visits=budget_visits)
    dt = time.time() - t0
    return dt / budget_visits, n_cpu, t_init, int(o.lik_evals.value)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="C4", choices=sorted(WORKLOADS))
    ap.add_argument("--init", default="true", choices=["true", "rand"])
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 VALU, 2 MFMA")
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--prune", type=int, default=0,
                    help="0 auto (exact pruning + certified stays), 1 off, 3 pruning without certified stays")
    ap.add_argument("--resolver", type=int, default=0, help="0 auto, 1 off, 2 always")
    ap.add_argument("--cpu-visits", type=int, default=20000,
                    help="visits of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cov", default="full", choices=["full", "diag", "fixed"],
                    help="covariance_type (diag / fixed: SURVEY 8f rows, not BASELINE configs)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world

    from pybgmm_amd import _lib
    from pybgmm_amd.chains import gather_chains
    from pybgmm_amd.gaussian.gaussian_components import reference_tables
    from pybgmm_amd.utils import gendata

    N, D, K, model = WORKLOADS[args.workload]
    X, z_true = gendata.synth_mixture(N, D, K, seed=args.seed)          # replicated data set
    m_0, k_0, v_0, S_0 = prior_for(args.cov, D)
    n_sweeps = args.warmup + args.steps
    # chain c: its own uniform (and permutation) streams, seeds seed + c
    rs = np.random.RandomState(1000 + args.seed + rank)
    u_all = rs.random_sample((n_sweeps, N))
    order_all = None
    power = None
    if model == "PCRPMM":
        order_all = np.stack([rs.permutation(N) for _ in range(n_sweeps)]).astype(np.int64)
        power = 1.01
    if args.init == "true":
        z0 = z_true
    else:
        z0 = np.unique(rs.randint(0, K, N), return_inverse=True)[1]

    t0 = time.time()
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K, device=local_rank,
                       tables=reference_tables(v_0, N) if args.cov != "fixed" else None, cov_type=args.cov)
    ctx.set_tuning(max_window=args.window, kernel_kind=args.kernel, resolver_mode=args.resolver,
                   prune_mode=args.prune)
    ctx.set_assignments(z0)
    t_setup = time.time() - t0
    t0 = time.time()
    ctx.upload_streams(u_all, order_all)
    t_h2d = time.time() - t0

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    def sweep_power(it):
        # pcrpmm.py:105: powered weights iff i_iter > power_burnin (= 0): sweep 0 is plain CRP
        return power if (power is not None and it > 0) else None

    for it in range(args.warmup):
        ctx.sweep_resident(it, sweep_power(it))
    barrier()
    t0 = time.time()
    lik = moves = 0
    for it in range(args.warmup, n_sweeps):
        ctx.sweep_resident(it, sweep_power(it))
        st = ctx.sweep_stats()
        lik += st["lik_evals"]
        moves += st["moves"]
    barrier()
    elapsed = time.time() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        agg = torch.tensor([float(lik), float(moves)], dtype=torch.float64, device="cuda")
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        lik_total, moves_total = float(agg[0].item()), float(agg[1].item())
    else:
        lik_total, moves_total = float(lik), float(moves)
    last_stats = ctx.sweep_stats()
    last_stats.update({"certified_visits": ctx.prune_stats()["certified_visits"]})
    log_marg = ctx.log_marg()
    K_final = ctx.K

    # --- likelihood-kernel roofline: one more sweep with HIP events around every launch ---
    def kernel_roofline(prune_mode):
        """One extra sweep with the likelihood kernel bracketed by HIP events (on the library's
        stream).  prune_mode 0 = as benchmarked, 1 = force the full evaluation of every pair."""
        ctx.set_tuning(max_window=args.window, kernel_kind=args.kernel, resolver_mode=args.resolver,
                       prune_mode=prune_mode)
        ctx.sweep_resident(n_sweeps - 1, sweep_power(n_sweeps - 1))          # settle the window policy
        ctx.set_kernel_timing(True)
        ctx.sweep_resident(n_sweeps - 1, sweep_power(n_sweeps - 1))
        n_launch, ms = ctx.kernel_timing()
        st = ctx.sweep_stats()
        ctx.set_kernel_timing(False)
        if n_launch <= 0 or ms <= 0:
            return None
        if args.cov != "full" and st["bound_blocks"] == 0 and ctx.prune_stats()["certified_visits"] == 0:
            # D logarithms per (visit, component): an FP64 VALU / transcendental kernel, no MFMA
            logs = st["scored"] * float(D)
            return {"kernel": "score_diag_kernel", "bound": "valu-transcendental",
                    "achieved": round(logs / (ms * 1e-3) / 1e9, 2), "peak": None, "unit": "Glog/s",
                    "frac": None, "traffic": None, "launches": n_launch,
                    "avg_launch_ms": round(ms / n_launch, 4),
                    "lik_evals_per_launch": round(st["scored"] / n_launch, 1),
                    "note": "covariance_type=%s is a SURVEY 8f row, not a BASELINE config" % args.cov}
        flops = st["scored"] * flops_per_lik_eval(D)
        achieved = flops / (ms * 1e-3) / 1e12
        is_mfma = (args.kernel == 2 or (args.kernel == 0 and D >= 12)) and args.cov == "full"
        nJ = (D + 15) // 16
        # flops the kernel really issues per evaluation: block-lower-triangular MFMA tiles
        # (2 nJ (nJ+1) tiles of 16x16x4 per 16 rows), or the exact triangle on the VALU path
        exec_per_eval = (2 * nJ * (nJ + 1) * 2048.0 / 16.0) if is_mfma else (D * (D + 1) + 2.0 * D)
        ps = ctx.prune_stats()
        pruning = bool(st["bound_blocks"] > 0 or ps["certified_visits"] > 0)
        # per (visit, component) in this library's formulation: y = cvec - Winv x through the lower
        # triangle (D (D + 1) flop), q = |y|^2 (2 D), the Student-t tail (~ D + 12)
        flops_kernel_alg = D * (D + 1.0) + 3.0 * D + 12.0
        hbm = st["scored"] / max(K_final, 1) * (8.0 * D + 24.0) / (ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            traffic = tj.get("hbm_bytes_per_launch_pruned" if pruning else "hbm_bytes_per_launch")
        n_visits_timed = st["scored"] / max(K_final, 1)
        common = {"traffic": traffic, "launches": n_launch, "avg_launch_ms": round(ms / n_launch, 4),
                  "lik_evals_per_launch": round(st["scored"] / n_launch, 1)}
        if pruning:
            # Pruned windows.  A visit that certify_kernel proves to keep its component
            # costs 17 B while nothing at all has moved (tier 1: the 16-byte record of the draw kernel's exact
            # alternative weight, the flag 1) or 61 B after a move somewhere (tier 2: its label, the 32-byte
            # per-point cache and its prior score on top), and X is not read; any other visit is also
            # sorted (32-byte record) and streamed once by the pruning kernel: 45 + 32 B and its row and
            # bookkeeping, 8 D + 24 B, on top.  `achieved` = those bytes / the launch time of the two kernels together;
            # the same visits priced at SURVEY's 8 D + 24 B each are reported next to it.  The matrix work
            # still issued (counted in the kernel, 2048 flop per instruction) and what the same decisions
            # would cost without pruning are given as well.
            n_cert = float(ps["certified_visits"])
            cert_bytes = 17.0 if st["moves"] == 0 else 61.0
            need_bytes = n_cert * cert_bytes + (n_visits_timed - n_cert) * (8.0 * D + 24.0 + 77.0)
            need = need_bytes / (ms * 1e-3) / 1e9
            executed = ps["mfma_instructions"] * 2048.0 / (ms * 1e-3) / 1e12
            heavy = "score_mfma_prune_kernel" if args.cov == "full" else "score_diag_prune_kernel"
            out = {"kernel": ("certify_kernel + " + heavy) if n_cert > 0 else heavy,
                   "bound": "hbm",
                   "achieved": round(need, 2), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                   "frac": round(need / PEAK_HBM_GBPS, 4),
                   "algorithmic_bytes_per_visit": round(need_bytes / max(n_visits_timed, 1), 1),
                   "survey_bytes_per_visit": 8.0 * D + 24.0,
                   "survey_equivalent_gbps": round(hbm, 2),
                   "fraction_of_visits_certified_to_stay": round(n_cert / max(n_visits_timed, 1), 5),
                   "fraction_scored_in_full": round(st["kept_blocks"] / max(st["bound_blocks"], 1), 5),
                   "mfma_instructions_per_launch": round(ps["mfma_instructions"] / n_launch, 1),
                   "mfma_executed_tflops": round(executed, 3),
                   "mfma_frac_of_spec_peak": round(executed / PEAK_FP64_MFMA_TFLOPS, 4),
                   "unpruned_equivalent_tflops": round(st["scored"] * flops_kernel_alg / (ms * 1e-3) / 1e12, 2)}
            out.update(common)
            if n_cert > 0:
                out["traffic"] = tj.get("hbm_bytes_per_launch_certified") if traffic is not None else None
            if args.cov != "full":          # no matrix work on this path; the PMC file is the full-covariance kernel's
                for key in ("mfma_instructions_per_launch", "mfma_executed_tflops", "mfma_frac_of_spec_peak",
                            "unpruned_equivalent_tflops"):
                    out.pop(key)
                out["traffic"] = None
            return out
        alg = st["scored"] * flops_kernel_alg / (ms * 1e-3) / 1e12
        executed = st["scored"] * exec_per_eval / (ms * 1e-3) / 1e12
        out = {"kernel": "score_mfma_kernel" if is_mfma else "score_valu_kernel", "bound": "mfma",
               "achieved": round(alg, 3), "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
               "frac": round(alg / PEAK_FP64_MFMA_TFLOPS, 4),
               "flops_per_lik_eval": flops_kernel_alg,
               "executed_tflops_incl_tile_padding": round(executed, 3),
               "executed_frac_of_spec_peak": round(executed / PEAK_FP64_MFMA_TFLOPS, 4),
               "executed_frac_of_sustained_mfma": round(executed / SUSTAINED_FP64_MFMA_TFLOPS, 4),
               "survey_formulation_flops_per_lik_eval": flops_per_lik_eval(D),
               "survey_formulation_tflops": round(achieved, 3),
               "hbm_gbps_algorithmic": round(hbm, 2)}
        out.update(common)
        return out

    roofline = roofline_full = roofline_pruned = None
    if not args.no_kernel_timing:
        roofline = kernel_roofline(args.prune)
        if roofline and roofline["kernel"].endswith("_prune_kernel"):
            if roofline.get("fraction_of_visits_certified_to_stay", 0) > 0:
                # the same sweep without certified stays: every visit streamed by the pruning kernel
                roofline_pruned = kernel_roofline(3)
            # the same kernel family with pruning off: every (visit, component) pair through the
            # full quadratic form -- the MFMA-efficiency number
            roofline_full = kernel_roofline(1)
        ctx.set_tuning(max_window=args.window, kernel_kind=args.kernel, resolver_mode=args.resolver,
                       prune_mode=args.prune)

    # --- the same chain with the exact shortcuts switched off, whole sweeps timed the same way ---
    # (the trajectory is identical in all modes; tests/test_gpu_parity.py, tools/soak.py)
    by_mode = None
    # (Replaying a sweep's uniforms is only a fair timing when the chain is at rest: a chain that
    # moves would mostly stay on the replay.  Mover-dense workloads report the benchmarked rate only.)
    if rank == 0 and n_gpus == 1 and not args.no_kernel_timing and args.prune == 0 and moves > 0:
        by_mode = {"as_benchmarked": round(args.steps / elapsed, 2),
                   "note": "the chain moves: the shortcut-free modes are not timed by replay"}
    elif rank == 0 and n_gpus == 1 and not args.no_kernel_timing and args.prune == 0:
        by_mode = {"as_benchmarked": round(args.steps / elapsed, 2)}
        for name, mode, reps in (("certified_stays_off", 3, 5), ("pruning_off_every_pair_evaluated", 1, 3)):
            ctx.set_tuning(max_window=args.window, kernel_kind=args.kernel, resolver_mode=args.resolver,
                           prune_mode=mode)
            ctx.sweep_resident(n_sweeps - 1, sweep_power(n_sweeps - 1))
            barrier()
            t0 = time.time()
            for _ in range(reps):
                ctx.sweep_resident(n_sweeps - 1, sweep_power(n_sweeps - 1))
            barrier()
            by_mode[name] = round(reps / (time.time() - t0), 2)
        ctx.set_tuning(max_window=args.window, kernel_kind=args.kernel, resolver_mode=args.resolver,
                       prune_mode=args.prune)

    # --- the one collective: final label gather (RCCL over xGMI), outside the timed region ---
    t0 = time.time()
    z_all, lm_all = gather_chains(ctx.assignments(), np.array([log_marg]),
                                  device=torch.device("cuda", local_rank) if world > 1 else None)
    t_gather = time.time() - t0

    cpu = None
    if rank == 0 and n_gpus == 1 and args.cpu_visits > 0:
        per_visit, n_cpu, t_init, cpu_lik = cpu_baseline(D, K, args.seed + 7, args.cpu_visits, args.cov)
        cpu = {"value": round(1.0 / (per_visit * N), 8), "unit": "sweeps/s", "cores": 1,
               "kind": "port",
               "sample": "%d visits of one sweep on a N=%d twin (same D=%d, K=%d, prior, init at truth); "
                         "%.1f us/visit extrapolated to N=%d" % (args.cpu_visits, n_cpu, D, K,
                                                                 per_visit * 1e6, N),
               "us_per_visit": round(per_visit * 1e6, 2),
               "lik_evals_per_s": round(cpu_lik / (per_visit * args.cpu_visits), 1),
               "reference_python_us_per_visit_survey_container": 802.0 if args.workload == "C4" else None}

    if rank == 0:
        sweeps_total = args.steps * n_gpus
        value = sweeps_total / elapsed
        out = {
            "metric": "gibbs_sweeps_per_sec", "value": round(value, 4), "unit": "sweeps/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %s D=%d N=%d K~%d%s, one independent chain per GPU, init=%s"
                                   % (args.workload, model, D, N, K,
                                      " covariance_type=%s" % args.cov if args.cov != "full" else "", args.init),
                       "parallelism": "replica_chains_x%d" % n_gpus,
                       "certified_stays": bool(args.prune == 0),
                       "exact_pruning": bool(args.prune == 0 and (args.cov != "full" or (args.kernel != 1 and (D >= 12 or args.kernel == 2))))},
            "lik_evals_per_sec": round(lik_total / elapsed, 1),
            "us_per_visit": round(elapsed / args.steps / N * 1e6, 5),
            "roofline": roofline,
            "roofline_pruned_evaluation": roofline_pruned,
            "roofline_full_evaluation": roofline_full,
            "cpu_baseline": cpu,
            "extra": {"moves_per_sweep": moves_total / max(sweeps_total, 1),
                      "K_final": K_final, "log_marg_rank0": log_marg,
                      "last_sweep": last_stats, "setup_s": round(t_setup, 3),
                      "h2d_streams_s": round(t_h2d, 4),
                      "pcie_inclusive_sweeps_per_s": round(
                          sweeps_total / (elapsed + t_h2d * args.steps / n_sweeps), 4),
                      "sweeps_per_s_by_mode": by_mode,
                      "label_gather_s": round(t_gather, 4),
                      "gathered_shape": list(z_all.shape)},
        }
        if cpu:
            out["extra"]["gpu_over_cpu_port"] = round(value / cpu["value"], 1)
        print(json.dumps(out))
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

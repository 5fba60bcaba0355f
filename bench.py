#!/usr/bin/env python3
"""
bench.py -- collapsed-Gibbs sweeps/s of the CRP Gaussian mixture on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU, one independent chain per rank)

A "step" is one full Gibbs sweep (N_data reassignment visits) of BASELINE.json's headline
configuration, configs[3]: CRPMM, D=64, N=1e6, K~200, synthetic isotropic mixture (SURVEY.md 8d
recipe).  X is resident in HBM before the timed region; every sweep's N uniforms (and, for pCRP workloads,
its np.random.permutation(N)) are GENERATED INSIDE the timed region, on the device, as the continuation of
the chain's own MT19937 streams -- SURVEY 8(d)'s t_sweep = inputs + kernels, what the classes' sampler loops
pay per sweep.  The rate with the inputs resident ahead of time and the PCIe-inclusive rate (host uniforms
uploaded per sweep) are reported in `extra`, never as `value`.  The chains are replicas: no
data-path collective; one RCCL all-gather of the final labels after the timed region.

What `value` measures (--mode):
  evaluated (default)  the chain at the truth, every visit EVALUATED every sweep: its row of X is read,
                       the predictive under its own component computed exactly (v_mfma_f64), every other
                       component excluded by an exact bound -- home_kernel, with resid_dense_kernel (D <= 32) or score_mfma_prune_kernel behind it
                       for the visits the per-home bound table cannot decide.  Certified stays
                       (visits proven to stay from cached state, X untouched) are OFF: that shortcut
                       makes the sweep a memo lookup on well separated data and is reported in `extra` only.
  certified            the library's default configuration (certified stays on)
  full                 every (visit, component) pair through the full quadratic form -- score_mfma_kernel
The JSON line also carries
  roofline     -- the dominant likelihood kernel of the timed mode, its launches timed with HIP events on the
                  library's stream inside this run; `traffic` from rocprofv3 --pmc passes of this same
                  script (FETCH_SIZE, WRITE_SIZE, separate passes) run as child processes
  burnin       -- the mover-dense regime: the same workload from the reference's default "rand"
                  initialisation (igmm.py:86-94), sweep by sweep until fewer than 1 % of the visits move
  class_api    -- the API the north star names: CRPMM / PCRPMM(X, NIW(..), 1.0, None, assignments=z_true)
                  .collapsed_gibbs_sampler(n, z_true), median record["sample_time"] (gmm/gmm.py:76) with the per-sweep
                  clustering metrics off and on, beside the C-ABI loop that `value` times
  cpu_baseline -- the C oracle (a port of the reference algorithm, 1 thread) on a bounded sample
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
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
MODES = {"evaluated": 3, "certified": 0, "full": 1}     # -> prune_mode of bgmm_set_tuning
PEAK_FP64_MFMA_TFLOPS = 78.6      # MI355X dense FP64 matrix peak (spec); see DESIGN.md section 4
PEAK_HBM_GBPS = 8000.0
REFERENCE_US_PER_VISIT = {"C4": 802.0, "C3": 209.0, "C2": 220.0}   # BASELINE.md, measured in the survey container


def survey_bytes_per_visit(D):
    """algorithmic HBM bytes of one visit (row of X, uniform, prior score, label in + out)."""
    return 8.0 * D + 24.0


def survey_flops_per_visit(D, K):
    """K (2 D^2 + 3 D + 12) + 4 D^2 (the reference's two einsums against a dense inverse)."""
    return K * (2.0 * D * D + 3.0 * D + 12.0) + 4.0 * D * D


def kernel_flops_per_pair(D):
    """This library's formulation of one (visit, component) pair: y = cvec - Winv x through the lower
    triangle (D (D + 1) flop), q = |y|^2 (2 D), the Student-t tail (~ D + 12)."""
    return D * (D + 1.0) + 3.0 * D + 12.0


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


def cpu_baseline(D, K, seed, budget_visits, cov="full", threads=1):
    """C oracle (oracle/gibbs_oracle.c: the reference's algorithm one visit at a time, its inner loops walked row-wise so
    that gcc vectorises them -- the same additions in the same order) on ONE thread, on a down-sized twin of the workload:
    same D, K, prior, init-at-truth; per-visit cost does not depend on N.  (The oracle can share a visit's K evaluations
    over threads with the same floats, but on the GPU box's 256-core host a hand-over between cores costs ~20 us and 32
    threads run the C4 shape three times SLOWER than one -- profiles/r06/oracle_threads.txt: no threaded leg.)"""
    from oracle import c_oracle
    from pybgmm_amd.utils import gendata
    c_oracle.set_threads(threads)
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


def cpu_baseline_numpy(D, K, seed, visits):
    """the numpy restatement of the reference (oracle/gibbs_numpy.py: the reference's own numpy / scipy
    calls -- slogdet + inv from scratch per update, einsum contraction, scipy logsumexp -- bit-identical floats) timed
    on this box's host cores, on a small twin of the workload (per-visit cost does not depend on N)."""
    from oracle import gibbs_numpy
    from pybgmm_amd.utils import gendata
    n_cpu = 4 * K
    X, z_true = gendata.synth_mixture(n_cpu, D, K, seed)
    m_0, k_0, v_0, S_0 = prior_for("full", D)
    o = gibbs_numpy.NumpyGibbsOracle(X, m_0, k_0, v_0, S_0, 1.0, z_true, 4 * K)
    u = np.random.RandomState(seed).random_sample(n_cpu)
    visits = min(visits, n_cpu)
    t0 = time.time()
    for i in range(visits):
        o.visit(i, u[i])
    return (time.time() - t0) / visits, n_cpu


def steady_moving_leg(args, local_rank):
    """The regime a chain lives in when its clusters overlap: the workload's shape with the centres
    drawn at `--moving-sep` times the scale of the at-rest data set, so that of the order of 1 % of the visits move at
    equilibrium.  Chain started at the generating labels, three sweeps to settle, then five timed ones -- uniforms
    generated inside the timed region as in the headline."""
    import random as _random
    from pybgmm_amd.utils import gendata
    N, D, K, model = WORKLOADS[args.workload]
    X2, zt = gendata.synth_mixture(N, D, K, seed=args.seed + 5, mu_scale=args.moving_sep)
    ctx = make_context(args, X2, zt, local_rank, "certified")          # the library's default configuration
    rs = np.random.RandomState(4000 + args.seed)
    _, key_t, _ = _random.Random(4000 + args.seed).getstate()
    key, pos = np.asarray(key_t[:-1], dtype=np.uint32), int(key_t[-1])
    sweeps = []
    st_ = rs.get_state()
    np_key, np_pos = np.asarray(st_[1], dtype=np.uint32), int(st_[2])
    for it in range(8):
        power = 1.01 if (model == "PCRPMM" and it > 0) else None
        ctx.synchronize()
        t0 = time.time()
        order = None
        if model == "PCRPMM":                 # (the permutation is part of the sweep's time, as in the headline)
            nxt = ctx.stage_permutation_mt19937(np_key, np_pos)
            if nxt is None:
                order = rs.permutation(N).astype(np.int64)
            else:
                np_key, np_pos = nxt
        key, pos = ctx.stage_mt19937(key, pos, order)
        ctx.sweep_staged(power)
        ctx.synchronize()
        dt = time.time() - t0
        st, ss, ps = ctx.sweep_stats(), ctx.safe_stats(), ctx.path_stats()
        sweeps.append({"seconds": round(dt, 4), "moves": st["moves"], "K": ctx.K, "safe_stay_windows": ss["windows"],
                       "visits_examined_by_proof_pass": ss["visits_examined"], "unproven_visits_walked": ss["unproven_walked"],
                       "windows_ended_by_budget": ss["budget_cuts"], "budget": round(ss["budget"], 4),
                       "plain_frozen_factor_windows": ps["frozen_windows"] - ss["windows"]})
    ctx.close()
    timed = sweeps[3:]
    t = sum(x["seconds"] for x in timed)
    mv = sum(x["moves"] for x in timed)
    side = None
    if args.burnin_chains > 1 and model != "PCRPMM" and args.cov == "full":
        # the same regime with G chains side by side on this GPU (their own generators over one copy of the data): chains of
        # one shape share the launches of their safe-stay steps (api_group.hip)
        from pybgmm_amd.chains import ChainGroup
        G = args.burnin_chains
        m_0, k_0, v_0, S_0 = prior_for(args.cov, D)
        grp = ChainGroup(X2, m_0, k_0, v_0, S_0, 1.0, max(4 * K, 64), n_chains=G, seed=4100 + args.seed, device=local_rank, cov_type=args.cov)
        grp.set_assignments([zt] * G)
        for _ in range(3):
            grp.sweep()
        for c_ in grp.ctxs:
            c_.synchronize()
        t0 = time.time()
        n_rounds = 3
        for _ in range(n_rounds):
            grp.sweep()
        dtg = (time.time() - t0) / n_rounds
        gs = grp.ctxs[0].group_stats()
        mvg = sum(c_.sweep_stats()["moves"] for c_ in grp.ctxs)
        grp.close()
        one = len(timed) / t
        side = {"chains": G, "ms_per_round_of_sweeps": round(1e3 * dtg, 2), "aggregate_sweeps_per_s": round(G / dtg, 2),
                "aggregate_over_single_chain": round(G / dtg / one, 2), "moves_in_the_last_round": int(mvg),
                "shared_safe_stay_batches_of_chain_0": gs["shared_safe_stay_batches"],
                "how": "bgmm_group_sweep_staged: the chains' safe-stay steps are one launch per kernel for all of them (workgroup (x, chain)); "
                       "every chain label for label its solo run (tests/test_gpu_parity.py::test_chains_side_by_side_share_safe_stay_steps)"}
    return {"workload": "%s shape, centres at mu_scale = %.2f (at-rest data: 4.0): clusters overlap" % (args.workload, args.moving_sep),
            "chains_side_by_side": side,
            "sweeps_per_s": round(len(timed) / t, 3), "ms_per_sweep": round(1e3 * t / len(timed), 2),
            "moves_per_sweep": round(mv / len(timed), 1), "movers_fraction": round(mv / len(timed) / N, 5),
            "us_per_move": round(1e6 * t / max(mv, 1), 2),
            "unproven_visits_walked_per_move": round(sum(x["unproven_visits_walked"] for x in timed) / max(mv, 1), 2),
            "sweeps": sweeps}


def many_chains_leg(args, X, z0, local_rank, single_rate):
    """Small dimensions: one sweep is one workgroup's chain of dependent draws, so a single chain leaves 255 of the 256
    compute units idle.  `--chains` G independent chains of the workload on THIS GPU (chain c: its own generator, seed
    + c, continued on the device), every round of sweeps one bgmm_group_sweep_staged call: aggregate sweeps/s against the
    single chain of the headline."""
    from pybgmm_amd.chains import ChainGroup
    from pybgmm_amd.utils import gendata
    N, D, K, model = WORKLOADS[args.workload]
    G = args.chains
    m_0, k_0, v_0, S_0 = prior_for(args.cov, D)
    t0 = time.time()
    grp = ChainGroup(X, m_0, k_0, v_0, S_0, 1.0, max(4 * K, 64), n_chains=G, seed=7000 + args.seed, device=local_rank,
                     cov_type=args.cov)
    grp.set_assignments([z0] * G)
    t_setup = time.time() - t0
    rs = np.random.RandomState(7000 + args.seed)

    def one_round(it):
        orders = [rs.permutation(N).astype(np.int64) for _ in range(G)] if model == "PCRPMM" else None
        grp.sweep(orders, [1.01 if (model == "PCRPMM" and it > 0) else None] * G)
    for it in range(2):
        one_round(it)
    n = 8
    t0 = time.time()
    for it in range(2, 2 + n):
        one_round(it)
    dt = (time.time() - t0) / n
    moves = [ctx.sweep_stats()["moves"] for ctx in grp.ctxs]
    distinct = len({int(ctx.assignments()[:2000].sum()) for ctx in grp.ctxs[:8]})
    grp.close()
    return {"chains": G, "ms_per_round": round(1e3 * dt, 2), "aggregate_sweeps_per_s": round(G / dt, 2),
            "single_chain_sweeps_per_s": round(single_rate, 3), "aggregate_over_single_chain": round(G / dt / single_rate, 1),
            "moves_last_sweep_min_max": [int(min(moves)), int(max(moves))], "distinct_label_vectors_among_first_8": distinct,
            "setup_s": round(t_setup, 2),
            "how": "ChainGroup: G contexts on one device, uniforms from each chain's own MT19937 continued on the device, one "
                   "bgmm_group_sweep_staged call per round (sweep_begin + sweep_seq for all chains in two launches: one workgroup "
                   "= one compute unit per chain)"}


def make_context(args, X, z0, local_rank, mode):
    from pybgmm_amd import _lib
    from pybgmm_amd.gaussian.gaussian_components import reference_tables
    N, D, K, _ = WORKLOADS[args.workload]
    m_0, k_0, v_0, S_0 = prior_for(args.cov, D)
    ctx = _lib.Context(X, m_0, k_0, v_0, S_0, 1.0, 4 * K, device=local_rank,
                       tables=reference_tables(v_0, N) if args.cov != "fixed" else None, cov_type=args.cov)
    ctx.set_tuning(max_window=args.window, kernel_kind=args.kernel, resolver_mode=args.resolver,
                   prune_mode=MODES[mode])
    ctx.set_home_pass(args.home)
    ctx.set_assignments(z0)
    return ctx


def heavy_kernel_name(args, mode, D):
    if args.cov != "full":
        return "score_diag_kernel" if mode == "full" else "score_diag_prune_kernel"
    mfma = args.kernel == 2 or (args.kernel == 0 and D >= 12)
    if mode == "full":
        return "score_mfma_kernel" if mfma else "score_valu_kernel"
    return "score_mfma_prune_kernel"


def pruned_window_kernels(args, mode, D, home_decided, handled):
    """The kernels that stream the rows of a pruned window: home_kernel in front (kernels_home.hip) while it
    decides most visits, score_mfma_prune_kernel on what it passes on."""
    name = heavy_kernel_name(args, mode, D)
    if name == "score_mfma_prune_kernel" and handled > 0 and home_decided > 0.5 * handled:
        # (D <= 32 with certified stays off: a short residual list is settled by resid_dense_kernel, kernels_resid.hip, and
        #  the visits it settles count as decided by the home pass)
        behind = "resid_dense_kernel" if (D <= 32 and mode == "evaluated") else name
        return ("home_kernel", behind)
    return (name,)


def pmc_traffic(args, mode, kernel_names, timeout_s=240):
    """HBM bytes per working launch of the kernels `kernel_names` together, measured NOW: two child runs of this script under
    rocprofv3 --pmc (FETCH_SIZE; WRITE_SIZE -- they do not fit one pass), as MI355X_MICROARCH.md
    prescribes; FETCH_SIZE doubled for gfx950.  None when rocprofv3 is missing or a pass fails."""
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    out = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        tmp = tempfile.mkdtemp(prefix="bgmm_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", counter, "--kernel-trace", "-d", tmp, "-o", "pmc", "--output-format", "csv", "--",
               sys.executable, os.path.abspath(__file__), "--inner-pmc", "--workload", args.workload, "--mode", mode,
               "--cov", args.cov, "--kernel", str(args.kernel), "--seed", str(args.seed)]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE,
                               stderr=subprocess.STDOUT, timeout=timeout_s)
        except Exception as e:       # noqa: BLE001 (timeout, missing binary ...)
            shutil.rmtree(tmp, ignore_errors=True)
            return None, "rocprofv3 pass failed: %r" % (e,)
        hits = glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True)
        vals = {k: [] for k in kernel_names}
        if hits:
            with open(hits[0]) as f:
                for row in csv.DictReader(f):
                    name = row["Kernel_Name"].split("(")[0].split("<")[0].split(" ")[-1]
                    if name not in vals or row["Counter_Name"] != counter:
                        continue
                    dur_us = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
                    vals[name].append((float(row["Counter_Value"]), dur_us))
        if args.keep_pmc:
            # the evidence behind roofline.traffic: per kernel of interest every launch's raw counter value (KiB) and duration
            os.makedirs(args.keep_pmc, exist_ok=True)
            with open(os.path.join(args.keep_pmc, "pmc_%s_%s_%s.csv" % (args.workload, mode, counter)), "w") as f:
                f.write("kernel,counter,value_KiB,duration_us\n")
                for k in kernel_names:
                    for v, dur in vals[k]:
                        f.write("%s,%s,%.1f,%.2f\n" % (k, counter, v, dur))
        shutil.rmtree(tmp, ignore_errors=True)
        if not vals[kernel_names[0]]:
            return None, "no %s rows for %s (rc %d)" % (counter, kernel_names[0], r.returncode)
        # the launches of the leading kernel that did a whole window; the kernels behind it run once per such launch
        longest = max(v[1] for v in vals[kernel_names[0]])
        n_work = len([v for v in vals[kernel_names[0]] if v[1] >= 0.5 * longest])
        total = sum(v[0] for v in vals[kernel_names[0]] if v[1] >= 0.5 * longest)
        for k in kernel_names[1:]:
            rows = sorted(vals[k], key=lambda v: -v[1])[:n_work]       # (its n_work longest launches)
            total += sum(v[0] for v in rows)
        out[counter] = total / n_work
    # counters are in KiB; FETCH_SIZE counts 128-byte requests as 64 bytes on gfx950
    return int(1024 * (2.0 * out["FETCH_SIZE"] + out["WRITE_SIZE"])), \
        "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of this script, FETCH_SIZE x 2 for gfx950"


def inner_pmc(args):
    """Child of pmc_traffic: set the workload up and run a few sweeps in the requested mode."""
    from pybgmm_amd.utils import gendata
    N, D, K, model = WORKLOADS[args.workload]
    X, z_true = gendata.synth_mixture(N, D, K, seed=args.seed)
    rs = np.random.RandomState(1000 + args.seed)
    ctx = make_context(args, X, z_true, 0, args.mode)
    for it in range(3):
        order = rs.permutation(N).astype(np.int64) if model == "PCRPMM" else None
        ctx.sweep(rs.random_sample(N), order, 1.01 if (model == "PCRPMM" and it > 0) else None)
    ctx.close()


def class_api_leg(args, X, z_true, local_rank, abi_ms_per_step):
    """The API the north star names (the timing is record_dict["sample_time"], reference
    gmm/gmm.py:76): the model classes on the benchmarked workload, chain at the truth, the caller's two global generators
    seeded as a reference script seeds them.  The classes run the library's default configuration; `evaluated` sets the
    mode `value` is measured in (certified stays off) on the model's context before the sampler loop, so that the two
    medians can be compared with the C-ABI loop directly.  record_metrics off / on = without / with the per-sweep
    NMI / MI / VI / loss of the record dict (device contingency table + dispersion, gmm/gmm.py:85-104)."""
    import random as _random
    from pybgmm_amd.igmm import CRPMM, PCRPMM
    from pybgmm_amd.prior import NIW
    N, D, K, model = WORKLOADS[args.workload]
    m_0, k_0, v_0, S_0 = prior_for("full", D)
    cls = {"CRPMM": CRPMM, "PCRPMM": PCRPMM}[model]
    out = {"call": "%s(X, NIW(m_0, k_0, v_0, S_0), 1.0, None, assignments=z_true, K_max=%d).collapsed_gibbs_sampler(n, z_true, "
                   "num_saved=0)" % (model, 4 * K),
           "c_abi_loop_ms_per_step": round(abi_ms_per_step, 4)}
    n = 120 if N * D <= 64000000 else 60
    for mode in ("evaluated", "default"):
        for metrics in (False, True):
            _random.seed(args.seed)
            np.random.seed(args.seed)
            mm = cls(X, NIW(m_0, k_0, v_0, S_0), 1.0, None, assignments=z_true, K_max=4 * K, device=local_rank)
            mm.record_metrics = metrics
            if mode == "evaluated":
                mm.components._ctx.set_tuning(prune_mode=MODES["evaluated"])
            t0 = time.time()
            record, _ = mm.collapsed_gibbs_sampler(n, z_true, num_saved=0)
            wall = time.time() - t0
            st = np.array(record["sample_time"][n // 4:])
            med = float(np.median(st))
            out["%s_metrics_%s" % (mode, "on" if metrics else "off")] = {
                "sweeps": n, "sample_time_median_ms": round(1e3 * med, 4), "sample_time_mean_ms": round(1e3 * float(st.mean()), 4),
                "sweeps_per_s_from_sample_time": round(1.0 / med, 1),
                "loop_wall_ms_per_sweep_incl_record": round(1e3 * wall / n, 4),
                "moves_last_sweep": mm.components._ctx.sweep_stats()["moves"], "K": record["components"][-1],
                "log_marg": record["log_marg"][-1]}
            mm.components._ctx.close()
    out["sample_time_over_c_abi_loop"] = round(out["evaluated_metrics_off"]["sample_time_median_ms"] / abi_ms_per_step, 3)
    return out


def burnin_leg(args, X, local_rank, K_true, seed):
    """The same data from the reference's default initialisation ("rand": uniform labels over K, igmm.py:86-94),
    swept until fewer than 1 % of the visits move (at most 6 sweeps)."""
    N, D, K, model = WORKLOADS[args.workload]
    rs = np.random.RandomState(7000 + seed)
    z0 = np.unique(rs.randint(0, K, N), return_inverse=True)[1]
    ctx = make_context(args, X, z0, local_rank, "certified")          # the library's default configuration
    sweeps = []
    until = None
    for it in range(6):
        u = rs.random_sample(N)
        order = rs.permutation(N).astype(np.int64) if model == "PCRPMM" else None
        power = 1.01 if (model == "PCRPMM" and it > 0) else None
        ctx.stage(u, order)
        ctx.synchronize()
        t0 = time.time()
        ctx.sweep_staged(power)
        ctx.synchronize()
        dt = time.time() - t0
        st, ps = ctx.sweep_stats(), ctx.path_stats()
        sweeps.append({"seconds": round(dt, 4), "moves": st["moves"], "K": ctx.K,
                       "frozen_factor_windows": ps["frozen_windows"],
                       "us_per_move": round(dt * 1e6 / max(st["moves"], 1), 3)})
        if st["moves"] < 0.01 * N:
            until = it + 1
            break
    lm = ctx.log_marg()
    ctx.close()
    first = sweeps[0]
    ref = REFERENCE_US_PER_VISIT.get(args.workload)
    return {"init": "rand (K=%d)" % K, "first_sweep_s": first["seconds"], "first_sweep_moves": first["moves"],
            "first_sweep_us_per_move": first["us_per_move"],
            "sweeps_until_moves_below_1pct": until, "sweeps": sweeps, "log_marg_after": lm,
            "reference_python_first_sweep_s": round(ref * N * 1e-6, 1) if ref else None,
            "speedup_over_reference_python": round(ref * N * 1e-6 / first["seconds"], 1) if ref else None}


def burnin_chains_leg(args, X, local_rank, single_first):
    """G chains of the workload from the reference's "rand" start, side by side on THIS GPU ("the 8 XCDs could
    additionally host independent chains").  A sweep of a chain that still moves is a latency chain that keeps one
    workgroup busy; bgmm_group_sweep_staged runs such chains concurrently, one stream and one host thread each.  Chain c
    has its own generator (seed + c, continued on the device) and its own random start."""
    from pybgmm_amd.chains import ChainGroup
    N, D, K, model = WORKLOADS[args.workload]
    G = args.burnin_chains
    m_0, k_0, v_0, S_0 = prior_for(args.cov, D)
    z0s = [np.unique(np.random.RandomState(7100 + args.seed + c).randint(0, K, N), return_inverse=True)[1] for c in range(G)]
    t0 = time.time()
    grp = ChainGroup(X, m_0, k_0, v_0, S_0, 1.0, max(4 * K, 64), n_chains=G, seed=7100 + args.seed, device=local_rank, cov_type=args.cov)
    grp.set_assignments(z0s)
    for ctx in grp.ctxs:
        ctx.synchronize()
    t_setup = time.time() - t0
    rs = np.random.RandomState(7100 + args.seed)
    orders = [rs.permutation(N).astype(np.int64) for _ in range(G)] if model == "PCRPMM" else None
    t0 = time.time()
    try:
        grp.sweep(orders, None)
    except Exception:
        for c_, ctx in enumerate(grp.ctxs):          # (what every chain was doing: a failure here is a finding)
            try:
                print("chain %d: %s %s %s K=%d" % (c_, ctx.sweep_stats(), ctx.window_pipeline_stats(), ctx.group_stats(), ctx.K), file=sys.stderr)
            except Exception as e2:                  # noqa: BLE001
                print("chain %d: %s" % (c_, e2), file=sys.stderr)
        raise
    dt = time.time() - t0
    moves = [ctx.sweep_stats()["moves"] for ctx in grp.ctxs]
    grp.close()
    agg = sum(moves) / dt
    one = single_first["moves"] / single_first["seconds"]
    return {"chains": G, "first_sweep_s": round(dt, 3), "moves": int(sum(moves)), "aggregate_moves_per_s": round(agg, 1),
            "single_chain_moves_per_s": round(one, 1), "aggregate_over_single_chain": round(agg / one, 2), "setup_s": round(t_setup, 2),
            "how": "bgmm_group_sweep_staged: chains that cannot take the one-workgroup sweep run concurrently, a host thread "
                   "each; while they burn in together their frozen-factor windows are ONE pipelined sequence of shared launches "
                   "(workgroup (x, chain); api_group.hip gram_group_pipe_launch); every chain label for label its solo run "
                   "(tests/test_gpu_parity.py::test_group_sweep_runs_large_chains_concurrently_and_equal_to_solo, "
                   "::test_chains_side_by_side_share_pipelined_windows)"}


def launch_plan(gpus, env, devices_visible):
    """What `--gpus N` means for this process.  ("run", world, rank, local_rank): it is a rank (the only one for
    N = 1, or one that torch.distributed.run started); ("spawn",): N > 1 and no launcher in the environment, so this
    process must become one.  Raises SystemExit with a message when the request cannot be met: fewer devices visible
    than ranks, or a launcher whose world size is not N.  `devices_visible` None skips the device check (--launch-check)."""
    if gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if world != gpus:
            raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (gpus, world))
        rank, local_rank = int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0"))
        if devices_visible is not None and local_rank >= devices_visible:
            raise SystemExit("bench.py: rank %d wants GPU %d but only %d HIP device(s) are visible"
                             % (rank, local_rank, devices_visible))
        return ("run", world, rank, local_rank)
    if devices_visible is not None and devices_visible < gpus:
        raise SystemExit("bench.py: --gpus %d but only %d HIP device(s) are visible (one chain per GPU; no "
                         "oversubscription, no CPU fallback)" % (gpus, devices_visible))
    if gpus == 1:
        return ("run", 1, 0, 0)
    return ("spawn",)


def torchrun_argv(gpus, argv):
    """python -m torch.distributed.run ... bench.py <argv>: the command the driver itself uses for N > 1."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000, help="timed sweeps (the default keeps the timed region above 0.5 s at C4)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--min-seconds", type=float, default=1.0,
                    help="repeat the timed stretch of --steps sweeps until this much has been timed, report the median repeat (0: once)")
    ap.add_argument("--workload", default="C4", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="evaluated", choices=sorted(MODES))
    ap.add_argument("--init", default="true", choices=["true", "rand"])
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 VALU, 2 MFMA")
    ap.add_argument("--window", type=int, default=0)
    ap.add_argument("--resolver", type=int, default=0, help="0 auto, 1 per-mover kernels, 2 in-launch resolver, 3 frozen-factor windows")
    ap.add_argument("--home", type=int, default=0, help="first pass of pruned windows: 0 auto, 1 always, 2 never")
    ap.add_argument("--cpu-visits", type=int, default=20000, help="visits of the CPU baseline sample (0 = skip)")
    ap.add_argument("--cov", default="full", choices=["full", "diag", "fixed"],
                    help="covariance_type (diag / fixed: SURVEY 8f rows, not BASELINE configs)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-burnin", action="store_true")
    ap.add_argument("--burnin-chains", type=int, default=8,
                    help="chains of the burn-in leg's side-by-side measurement (D >= 12; 0 or 1: skip)")
    ap.add_argument("--no-moving", action="store_true", help="skip the steady_moving leg (overlapping clusters)")
    ap.add_argument("--no-class-api", action="store_true", help="skip the class_api leg (CRPMM / PCRPMM.collapsed_gibbs_sampler)")
    ap.add_argument("--moving-sep", type=float, default=0.55, help="mu_scale of the steady_moving leg's data set")
    ap.add_argument("--numpy-visits", type=int, default=400, help="visits of the numpy-restatement CPU baseline (0 = skip)")
    ap.add_argument("--chains", type=int, default=-1,
                    help="chains of the many_chains leg, all on this GPU (-1: 256 -- one per compute unit -- for D <= 4, none otherwise; 0: skip)")
    ap.add_argument("--pipeline", action="store_true",
                    help="stage sweep k + 1's inputs while sweep k is in the queue (bgmm_sweep_staged_begin / _end) instead of the "
                         "plain loop stage, sweep, stage, sweep -- measured: 5 400 against 5 373 sweeps/s at C4, the host's way "
                         "round the loop is the stream synchronisation and the launch latency, not the staging")
    ap.add_argument("--no-pmc", action="store_true")
    ap.add_argument("--keep-pmc", default="", metavar="DIR",
                    help="keep the per-launch FETCH_SIZE / WRITE_SIZE rows of the roofline kernels (the rocprofv3 --pmc child "
                         "passes behind roofline.traffic) as CSV files in DIR")
    ap.add_argument("--inner-pmc", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous of the --gpus N ranks over gloo and exit (no GPU work; the CPU test of the launcher)")
    args = ap.parse_args()
    if args.inner_pmc:
        inner_pmc(args)
        return

    import torch
    plan = launch_plan(args.gpus, os.environ, torch.cuda.device_count() if not args.launch_check else None)
    if plan[0] == "spawn":
        # `python bench.py --gpus N` without a launcher: become the launcher (one rank per GPU)
        os.execv(sys.executable, [sys.executable] + torchrun_argv(args.gpus, sys.argv[1:]))
    world, rank, local_rank = plan[1:]
    dist = None
    if world > 1 or args.launch_check:
        import torch.distributed as dist
        if args.launch_check:           # rendezvous only (CPU, gloo): what tests/test_bench_launch.py drives
            dist.init_process_group("gloo")
            t = torch.tensor([float(rank)], dtype=torch.float64)
            dist.all_reduce(t)
            if rank == 0:
                print(json.dumps({"launch_check": True, "n_gpus": world, "rank_sum": float(t.item())}))
            dist.barrier()
            dist.destroy_process_group()
            return
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    n_gpus = world

    from pybgmm_amd.chains import gather_chains
    from pybgmm_amd.utils import gendata

    N, D, K, model = WORKLOADS[args.workload]
    X, z_true = gendata.synth_mixture(N, D, K, seed=args.seed)          # replicated data set
    # chain c: its own generators, seeds seed + c.  The per-visit uniforms are the continuation of the chain's
    # random.Random -- produced ON THE DEVICE from its MT19937 state (bgmm_stage_mt19937: bit-identical to N calls of
    # random.random(), the state handed back), which is what CRPMM.collapsed_gibbs_sampler does every sweep; a pCRP sweep
    # also draws its permutation on the host and uploads it.  Both are INSIDE the timed region (t_sweep = the
    # sweep's inputs + kernels, the `sample_time` of gmm.py:76).
    import random as _random
    chain_rng = _random.Random(1000 + args.seed + rank)
    rs = np.random.RandomState(1000 + args.seed + rank)
    _, key_t, _ = chain_rng.getstate()
    mt_key, mt_pos = np.asarray(key_t[:-1], dtype=np.uint32), int(key_t[-1])
    power = 1.01 if model == "PCRPMM" else None
    z0 = z_true if args.init == "true" else np.unique(rs.randint(0, K, N), return_inverse=True)[1]

    t0 = time.time()
    ctx = make_context(args, X, z0, local_rank, args.mode)
    t_setup = time.time() - t0

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        ctx.synchronize()

    def sweep_power(it):
        # pcrpmm.py:105: powered weights iff i_iter > power_burnin (= 0): sweep 0 is plain CRP
        return power if (power is not None and it > 0) else None

    np_state = rs.get_state()
    np_key, np_pos = np.asarray(np_state[1], dtype=np.uint32), int(np_state[2])

    def one_sweep(it, stage_only=False):
        # what PCRPMM.collapsed_gibbs_sampler does per sweep (pybgmm_amd/igmm/pcrpmm.py): the visiting order is
        # np.random.permutation(N) from the chain's numpy stream -- drawn on the device (bgmm_stage_permutation_mt19937:
        # bit-identical, the state handed back), on the host where the library leaves it to the host
        nonlocal mt_key, mt_pos, np_key, np_pos
        order = None
        if model == "PCRPMM":
            nxt = ctx.stage_permutation_mt19937(np_key, np_pos)
            if nxt is None:
                rs.set_state(("MT19937", np_key, np_pos))
                order = rs.permutation(N).astype(np.int64)
                st_ = rs.get_state()
                np_key, np_pos = np.asarray(st_[1], dtype=np.uint32), int(st_[2])
            else:
                np_key, np_pos = nxt
        mt_key, mt_pos = ctx.stage_mt19937(mt_key, mt_pos, order)
        if not stage_only:
            ctx.sweep_staged(sweep_power(it))

    # --pipeline: sweep k is queued (bgmm_sweep_staged_begin), the inputs of sweep k + 1 are staged while it runs (host
    # work: a look-ahead hit is a memcmp), then sweep k is waited for and finished (bgmm_sweep_staged_end).  One staging
    # and one sweep per step, the same calls and the same trajectory as the plain loop
    # (tests/test_gpu_parity.py::test_pipelined_sweeps_equal_plain_sweeps).  The default is the plain loop -- what the
    # classes' sampler loops do.
    def pipelined_sweeps(it0, n):
        for it in range(it0, it0 + n):
            ctx.sweep_staged_begin(sweep_power(it))
            one_sweep(it + 1, stage_only=True)
            ctx.sweep_staged_end()

    def run_sweeps(it0, n):
        if not args.pipeline:
            for it in range(it0, it0 + n):
                one_sweep(it)
        else:
            pipelined_sweeps(it0, n)

    if args.pipeline:
        one_sweep(0, stage_only=True)            # (the first sweep's inputs; from then on every step stages the next one's)
    run_sweeps(0, args.warmup)
    barrier()
    # EXACTLY `steps` sweeps between a barrier + synchronize on both sides -- and, when that is a short stretch (the driver's
    # 20 steps are 4 ms at C4), the same measurement REPEATED until at least a second of sweeps has been timed: `value`
    # comes from the median repeat (every repeat is `steps` sweeps, bracketed the same way; one repeat when `steps` is
    # long enough on its own).
    tot0 = ctx.totals()                               # (the library sums its per-sweep counters: read at both ends)
    chunk_marks = []
    repeat_s = []
    it_next = args.warmup
    n_repeats = 1
    while len(repeat_s) < n_repeats:
        barrier()
        t0 = time.time()
        if args.steps >= 500 and not repeat_s:
            for c0 in range(0, args.steps, 250):      # (250 sweeps at a time, for the spread between stretches)
                run_sweeps(it_next + c0, min(250, args.steps - c0))
                if c0 + 250 <= args.steps:
                    chunk_marks.append(time.time())   # (every sweep ends with a stream sync: host time is device time)
        else:
            run_sweeps(it_next, args.steps)
        barrier()
        dt = time.time() - t0
        if not repeat_s:
            t_first0 = t0
            want = int(min(400, max(1, np.ceil(1.0 / max(dt, 1e-6))))) if args.min_seconds > 0 else 1
            if dist is not None:                      # (the same number of repeats on every rank)
                w_ = torch.tensor([want], dtype=torch.int64, device="cuda")
                dist.all_reduce(w_, op=dist.ReduceOp.MAX)
                want = int(w_.item())
            n_repeats = want
        repeat_s.append(dt)
        it_next += args.steps
    tot1 = ctx.totals()
    assert tot1["sweeps"] - tot0["sweeps"] == args.steps * n_repeats
    decided = (tot1["lik_evals"] - tot0["lik_evals"]) // n_repeats
    moves = (tot1["moves"] - tot0["moves"]) // n_repeats
    executed = (tot1["pairs_executed"] - tot0["pairs_executed"]) // n_repeats
    if dist is not None:                              # the MAX over ranks, repeat by repeat
        r_ = torch.tensor(repeat_s, dtype=torch.float64, device="cuda")
        mine_repeats = list(repeat_s)
        dist.all_reduce(r_, op=dist.ReduceOp.MAX)
        repeat_s = [float(v) for v in r_.tolist()]
    else:
        mine_repeats = list(repeat_s)
    elapsed_mine = float(np.median(mine_repeats))
    elapsed = float(np.median(repeat_s))
    chunk_rates = [round(250.0 / (b - a), 1) for a, b in zip([t_first0] + chunk_marks[:-1], chunk_marks)]
    # the same chain with its inputs already resident (8 sweeps' worth of uniforms / permutations uploaded ahead and
    # cycled through): what the sweep kernels alone sustain -- reported in `extra`, never the headline
    n_res = 8
    u_res = rs.random_sample((n_res, N))
    order_res = np.stack([rs.permutation(N) for _ in range(n_res)]).astype(np.int64) if model == "PCRPMM" else None
    t1 = time.time()
    ctx.upload_streams(u_res, order_res)
    t_h2d = time.time() - t1
    n_sweeps = n_res
    resident_rate = None
    if moves == 0:
        reps = max(50, min(args.steps, 400))
        for it in range(8):
            ctx.sweep_resident(it % n_res, power)
        barrier()
        t1 = time.time()
        for it in range(reps):
            ctx.sweep_resident(it % n_res, power)
        barrier()
        resident_rate = reps / (time.time() - t1)
    per_rank_rate = [round(args.steps / elapsed_mine, 3)]
    if dist is not None:
        mine = torch.tensor([elapsed_mine], dtype=torch.float64, device="cuda")
        every = torch.empty(world, dtype=torch.float64, device="cuda")
        dist.all_gather_into_tensor(every, mine)
        per_rank_rate = [round(args.steps / float(v), 3) for v in every.tolist()]
        agg = torch.tensor([float(decided), float(executed), float(moves)], dtype=torch.float64, device="cuda")
        dist.all_reduce(agg, op=dist.ReduceOp.SUM)
        decided_total, executed_total, moves_total = (float(v) for v in agg.tolist())
    else:
        decided_total, executed_total, moves_total = float(decided), float(executed), float(moves)
    mt_ahead, short_steps = ctx.mt_lookahead_stats(), ctx.short_step_stats()   # (of the warm-up and the timed sweeps)
    last_stats = ctx.sweep_stats()
    last_stats.update({"certified_visits": ctx.prune_stats()["certified_visits"],
                       "pairs_executed": ctx.path_stats()["pairs_executed"]})
    log_marg = ctx.log_marg()
    K_final = ctx.K

    def set_mode(mode):
        ctx.set_tuning(max_window=args.window, kernel_kind=args.kernel, resolver_mode=args.resolver,
                       prune_mode=MODES[mode])

    def kernel_roofline(mode, live_sweeps=0):
        """Every launch of the mode's dominant likelihood kernel bracketed by HIP events on the library's stream.
        live_sweeps > 0 (the benchmarked mode): that many sweeps EXACTLY as the timed region runs them -- inputs staged
        per sweep, the look-ahead generating the next sweeps' uniforms beside them -- so that the average is the one a
        kernel trace of this command shows; otherwise one extra sweep with resident inputs."""
        set_mode(mode)
        if live_sweeps > 0:
            # (kernel timing keeps every sweep inside _begin: the plain loop, the same kernels beside the same look-ahead)
            for it in range(4):
                one_sweep(args.warmup + args.steps + it)
            ctx.set_kernel_timing(True)
            for it in range(live_sweeps):
                one_sweep(args.warmup + args.steps + 4 + it)
        else:
            ctx.sweep_resident(n_sweeps - 1, power)          # settle the window policy
            ctx.set_kernel_timing(True)
            ctx.sweep_resident(n_sweeps - 1, power)
        n_launch, ms = ctx.kernel_timing()
        st, ps, pa = ctx.sweep_stats(), ctx.prune_stats(), ctx.path_stats()
        ctx.set_kernel_timing(False)
        if n_launch <= 0 or ms <= 0:
            return None
        name = heavy_kernel_name(args, mode, D)
        visits = float(N)
        sweeps_timed = max(live_sweeps, 1)
        avg_ms = ms / n_launch
        n_launch = n_launch / float(sweeps_timed)          # launches per sweep (the statistics below are the last sweep's)
        ms = ms / float(sweeps_timed)
        common = {"kernel": name, "launches": n_launch, "avg_launch_ms": round(avg_ms, 4),
                  "visits_per_launch": round(visits / n_launch, 1), "traffic": None,
                  "timed_over": ("%d sweeps run exactly as the timed region runs them (inputs staged per sweep, the look-ahead "
                                 "generating beside them)" % live_sweeps) if live_sweeps else "one sweep with resident inputs"}
        if mode == "full":
            pairs = float(pa["pairs_executed"])
            if args.cov != "full":
                logs = pairs * float(D)
                common.update({"bound": "valu-transcendental", "achieved": round(logs / (ms * 1e-3) / 1e9, 2),
                               "peak": None, "unit": "Glog/s", "frac": None})
                return common
            alg = pairs * kernel_flops_per_pair(D) / (ms * 1e-3) / 1e12
            nJ = (D + 15) // 16
            executed_tf = pairs * (2 * nJ * (nJ + 1) * 2048.0 / 16.0) / (ms * 1e-3) / 1e12
            common.update({"bound": "mfma", "achieved": round(alg, 3), "peak": PEAK_FP64_MFMA_TFLOPS,
                           "unit": "TFLOP/s", "frac": round(alg / PEAK_FP64_MFMA_TFLOPS, 4),
                           "flops_per_pair": kernel_flops_per_pair(D),
                           "flop_formula": "D (D + 1) + 3 D + 12 per (visit, component) pair: the inverse Cholesky "
                                           "factor's lower triangle, |y|^2, the Student-t tail",
                           "pairs_per_launch": round(pairs / n_launch, 1),
                           "executed_tflops_incl_tile_padding": round(executed_tf, 3)})
            return common
        # pruned windows: an HBM stream.  Algorithmic bytes = SURVEY 8(d): 8 D + 24 per visit the kernel handles.
        n_cert = float(ps["certified_visits"])
        handled = visits - n_cert
        need = handled * survey_bytes_per_visit(D) / (ms * 1e-3) / 1e9
        # both rooflines of the launch from EXECUTED work: algorithmic bytes against the HBM peak, the
        # v_mfma_f64_16x16x4_f64 instructions the kernels counted (2 048 flop each, tile padding included: that is what
        # occupies the pipe) against the FP64 matrix peak.  `bound` is the one the launch sits closer to.
        mfma_tf = float(ps["mfma_instructions"]) * 2048.0 / (ms * 1e-3) / 1e12 if args.cov == "full" else 0.0
        hbm_frac, mfma_frac = need / PEAK_HBM_GBPS, mfma_tf / PEAK_FP64_MFMA_TFLOPS
        if mfma_frac > hbm_frac:
            common.update({"bound": "mfma", "achieved": round(mfma_tf, 3), "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": round(mfma_frac, 4)})
        else:
            common.update({"bound": "hbm", "achieved": round(need, 2), "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                           "frac": round(hbm_frac, 4)})
        common.update({"hbm_frac": round(hbm_frac, 4), "mfma_frac": round(mfma_frac, 4),
                       "hbm": {"achieved": round(need, 2), "peak": PEAK_HBM_GBPS, "unit": "GB/s"},
                       "mfma": {"achieved": round(mfma_tf, 3), "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                                "what": "executed v_mfma_f64_16x16x4_f64 x 2048 flop (tile padding included)"},
                       "algorithmic_bytes_per_visit": survey_bytes_per_visit(D),
                       "visits_through_kernel_per_launch": round(handled / n_launch, 1),
                       "fraction_of_visits_certified_to_stay": round(n_cert / visits, 5),
                       "pairs_scored_in_full_per_visit": round(pa["pairs_executed"] / visits, 3),
                       "mfma_instructions_per_launch": round(ps["mfma_instructions"] / n_launch, 1)})
        names = pruned_window_kernels(args, mode, D, float(pa.get("home_decided", 0)), handled)
        common["kernel"] = " + ".join(names)
        common["kernels_for_traffic"] = list(names)
        if len(names) > 1:
            common["visits_decided_by_home_kernel"] = float(pa["home_decided"])
        if n_cert > 0:
            common["kernel"] = "certify_kernel + " + common["kernel"]
        return common

    roofline = None
    rates = {}
    extra_rooflines = {}
    single = rank == 0 and n_gpus == 1
    if not args.no_kernel_timing:
        roofline = kernel_roofline(args.mode, live_sweeps=200 if moves_total == 0 else 0)
        if roofline and moves_total == 0:
            # the same kernel with nothing beside it (inputs resident, no look-ahead generating on the second stream)
            alone = kernel_roofline(args.mode)
            if alone:
                roofline["alone_on_the_gpu"] = {"avg_launch_ms": alone["avg_launch_ms"], "frac": alone.get("frac"),
                                                "timed_over": alone["timed_over"]}
        if single and roofline and not args.no_pmc and args.cov == "full":
            traffic, note = pmc_traffic(args, args.mode,
                                        tuple(roofline.get("kernels_for_traffic") or [heavy_kernel_name(args, args.mode, D)]))
            roofline["traffic"] = traffic
            roofline["traffic_source"] = note
            if traffic:
                roofline["traffic_over_algorithmic"] = round(
                    traffic / (roofline.get("visits_through_kernel_per_launch", roofline["visits_per_launch"])
                               * survey_bytes_per_visit(D)), 3) if roofline["bound"] == "hbm" else None
    # --- the same chain in the other modes, whole sweeps timed the same way (identical trajectory in all
    #     of them: tests/test_gpu_parity.py); only for a chain at rest -- a replayed sweep of a moving chain
    #     would mostly stay
    if single and not args.no_kernel_timing and moves_total == 0:
        for mode, reps in (("evaluated", 5), ("certified", 20), ("full", 3)):
            if mode == args.mode:
                rates[mode] = round(resident_rate, 2) if resident_rate else None
                continue
            set_mode(mode)
            for _ in range(3):
                ctx.sweep_resident(n_sweeps - 1, power)
            barrier()
            t0 = time.time()
            for _ in range(reps):
                ctx.sweep_resident(n_sweeps - 1, power)
            barrier()
            rates[mode] = round(reps / (time.time() - t0), 2)
            if mode == "full" or (mode == "evaluated" and args.mode == "certified"):
                extra_rooflines[mode] = kernel_roofline(mode)
        set_mode(args.mode)

    # --- the one collective: final label gather (RCCL over xGMI), outside the timed region ---
    t0 = time.time()
    z_all, lm_all = gather_chains(ctx.assignments(), np.array([log_marg]),
                                  device=torch.device("cuda", local_rank) if world > 1 else None)
    t_gather = time.time() - t0
    # ... and the same gather through the C-ABI alone (bgmm_comm_* / bgmm_gather_labels: ncclAllGather of the
    # device-side labels); the two must agree
    from pybgmm_amd import _lib
    t0 = time.time()
    ident = [_lib.Comm.unique_id() if rank == 0 else None]
    if dist is not None:
        dist.broadcast_object_list(ident, src=0)
    comm = _lib.Comm(rank, world, ident[0], device=local_rank)
    t_comm_create = time.time() - t0                 # (librccl loaded + ncclCommInitRank: once per process, ~6 s)
    z_abi = ctx.gather_labels(comm, world)           # (first collective on the communicator)
    t0 = time.time()
    z_abi = ctx.gather_labels(comm, world)           # the gather itself, communicator in hand -- what a run pays per gather
    t_gather_abi = time.time() - t0
    comm.close()
    if not np.array_equal(z_abi, z_all):
        raise SystemExit("bench.py: bgmm_gather_labels and chains.gather_chains disagree on rank %d" % rank)
    ctx.close()

    burnin = None
    if single and not args.no_burnin and args.cov == "full":
        burnin = burnin_leg(args, X, local_rank, K, args.seed)
        if args.burnin_chains > 1 and D >= 12:
            burnin["chains_side_by_side"] = burnin_chains_leg(args, X, local_rank, burnin["sweeps"][0])

    class_api = None
    if single and not args.no_class_api and args.cov == "full" and args.mode == "evaluated" and args.init == "true":
        class_api = class_api_leg(args, X, z_true, local_rank, 1e3 * elapsed / args.steps)

    moving = None
    if single and not args.no_moving and args.cov == "full" and D >= 12:
        moving = steady_moving_leg(args, local_rank)

    many = None
    n_chains = args.chains if args.chains >= 0 else (256 if D <= 4 else 0)
    if single and n_chains > 1:
        args.chains = n_chains
        many = many_chains_leg(args, X, z0, local_rank, args.steps / elapsed)

    cpu = None
    if single and args.cpu_visits > 0:
        per_visit, n_cpu, t_init, cpu_lik = cpu_baseline(D, K, args.seed + 7, args.cpu_visits, args.cov)
        cpu = {"value": round(1.0 / (per_visit * N), 8), "unit": "sweeps/s", "cores": 1,
               "kind": "port",
               "sample": "%d visits of one sweep on a N=%d twin (same D=%d, K=%d, prior, init at truth), every "
                         "(visit, component) pair evaluated; %.1f us/visit extrapolated to N=%d"
                         % (args.cpu_visits, n_cpu, D, K, per_visit * 1e6, N),
               "us_per_visit": round(per_visit * 1e6, 2),
               "lik_evals_per_s": round(cpu_lik / (per_visit * args.cpu_visits), 1),
               "reference_python_us_per_visit_survey_container": REFERENCE_US_PER_VISIT.get(args.workload)}
        if args.numpy_visits > 0 and args.cov == "full":
            npv, n_np = cpu_baseline_numpy(D, K, args.seed + 7, args.numpy_visits)
            cpu["numpy_restatement"] = {
                "kind": "numpy-restatement", "value": round(1.0 / (npv * N), 8), "unit": "sweeps/s", "cores": 1,
                "us_per_visit": round(npv * 1e6, 2),
                "sample": "%d visits on a N=%d twin, oracle/gibbs_numpy.py: the reference's numpy / scipy calls (slogdet + "
                          "inv from scratch per update, einsum, logsumexp), floats bit-identical to the reference's" % (
                              min(args.numpy_visits, n_np), n_np)}

    if rank == 0:
        sweeps_total = args.steps * n_gpus
        value = sweeps_total / elapsed
        t_sweep = elapsed / args.steps
        out = {
            "metric": "gibbs_sweeps_per_sec", "value": round(value, 4), "unit": "sweeps/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(t_sweep * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %s D=%d N=%d K~%d%s, one independent chain per GPU, init=%s, mode=%s"
                                   % (args.workload, model, D, N, K,
                                      " covariance_type=%s" % args.cov if args.cov != "full" else "", args.init,
                                      args.mode),
                       "parallelism": "replica_chains_x%d" % n_gpus,
                       "mode": args.mode,
                       "certified_stays": bool(args.mode == "certified"),
                       "exact_pruning": bool(args.mode != "full")},
            # pairs whose quadratic form was EXECUTED per second / pairs DECIDED per second (sum over visits of K)
            "lik_evals_per_sec": round(executed_total / elapsed, 1),
            "lik_evals_decided_per_sec": round(decided_total / elapsed, 1),
            "us_per_visit": round(t_sweep / N * 1e6, 5),
            # whole-sweep fractions of the two rooflines under ITS accounting
            "survey_8d": {"bytes_sweep": N * survey_bytes_per_visit(D), "flops_sweep": N * survey_flops_per_visit(D, K),
                          "hbm_frac": round(N * survey_bytes_per_visit(D) / t_sweep / (PEAK_HBM_GBPS * 1e9), 4),
                          "formula": "bytes_visit = 8 D + 24; flops_visit = K (2 D^2 + 3 D + 12) + 4 D^2 is the reference's "
                                     "formulation of a sweep -- no fraction of the FP64 peak is quoted from it: exact bounds decide "
                                     "most pairs without executing them; the executed fractions are roofline.hbm_frac / "
                                     "roofline.mfma_frac (mode `full`: extra.roofline_other_modes.full)"},
            "roofline": roofline,
            "burnin": burnin,
            "steady_moving": moving,
            "class_api": class_api,
            "many_chains": many,
            "cpu_baseline": cpu,
            "extra": {"moves_per_sweep": moves_total / max(sweeps_total, 1),
                      "K_final": K_final, "log_marg_rank0": log_marg,
                      "last_sweep": last_stats, "setup_s": round(t_setup, 3),
                      "host_loop": "plain" if not args.pipeline else "pipelined: sweep k + 1's inputs staged while sweep k is in the queue "
                                   "(bgmm_sweep_staged_begin / _end)",
                      "value_is": "sweeps/s with every sweep's uniforms generated inside the timed region (the caller's MT19937 "
                                  "continued on the device, bgmm_stage_mt19937: sweep k + 1's are generated on a second stream while "
                                  "sweep k runs) and, for pCRP workloads, its np.random.permutation(N) drawn on the device as well "
                                  "(bgmm_stage_permutation_mt19937, bit-identical to numpy's) -- SURVEY 8(d)'s t_sweep",
                      "mt19937_lookahead": dict(mt_ahead, note="stage calls served by the look-ahead (the next sweep's uniforms "
                                                "generated on a second stream beside the running sweep; taken only when the caller's "
                                                "generator is exactly where the last call left it) / generated on the spot"),
                      "short_steps": dict(short_steps, note="sweeps queued as sweep_begin + home_kernel + apply (a chain at rest with "
                                          "certified stays off) that stood / were refused and redone with the full kernel set"),
                      "timed_region_s": round(elapsed, 4),
                      "timed_repeats": {"repeats": n_repeats, "sweeps_each": args.steps, "seconds_median": round(elapsed, 5),
                                        "seconds_min": round(min(repeat_s), 5), "seconds_max": round(max(repeat_s), 5),
                                        "seconds_all": round(sum(repeat_s), 3),
                                        "note": "`value` = steps / the median repeat; every repeat is exactly `steps` sweeps between "
                                                "barrier + synchronize (max over ranks), repeated until >= 1 s has been timed"},
                      "sweeps_per_s_per_250_sweeps": {"min": min(chunk_rates), "max": max(chunk_rates),
                                                      "median": sorted(chunk_rates)[len(chunk_rates) // 2]} if chunk_rates else None,
                      "resident_inputs_sweeps_per_s": round(resident_rate, 2) if resident_rate else None,
                      "h2d_of_8_sweeps_of_inputs_s": round(t_h2d, 4),
                      "host_uniforms_uploaded_per_sweep_sweeps_per_s": round(
                          1.0 / (1.0 / resident_rate + t_h2d / n_res), 2) if resident_rate else None,
                      "sweeps_per_s_by_mode_resident_inputs": rates or None,
                      "roofline_other_modes": extra_rooflines or None,
                      "label_gather_s": round(t_gather, 4),
                      "label_gather_c_abi_s": round(t_gather_abi, 4),
                      "label_gather_c_abi_comm_create_s": round(t_comm_create, 3),
                      "label_gathers_agree": True,
                      "sweeps_per_s_per_rank": per_rank_rate,
                      "gathered_shape": list(z_all.shape)},
        }
        # (librccl prints its version banner through C stdio when the label gather loads it: flushed here, so that the JSON
        # line is the LAST line this process writes)
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:       # noqa: BLE001
            pass
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

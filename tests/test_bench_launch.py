"""bench.py --gpus N: the launcher logic, driven on CPU.

`python bench.py --gpus N` without a launcher must become one (torch.distributed.run, one rank per GPU), refuse
when fewer devices are visible than ranks, and refuse a launcher whose world size is not N."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_single_rank_runs_in_process():
    assert bench.launch_plan(1, {}, 1) == ("run", 1, 0, 0)


def test_more_ranks_than_devices_is_refused():
    with pytest.raises(SystemExit) as e:
        bench.launch_plan(2, {}, 1)
    assert "only 1 HIP device" in str(e.value)
    with pytest.raises(SystemExit):
        bench.launch_plan(8, {}, 0)
    with pytest.raises(SystemExit):
        bench.launch_plan(1, {}, 0)
    with pytest.raises(SystemExit):        # a rank of a launcher on a box with too few devices
        bench.launch_plan(2, {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}, 1)


def test_launcher_world_must_match():
    with pytest.raises(SystemExit) as e:
        bench.launch_plan(8, {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"}, 8)
    assert "WORLD_SIZE=2" in str(e.value)
    assert bench.launch_plan(2, {"WORLD_SIZE": "2", "RANK": "1", "LOCAL_RANK": "1"}, 2) == ("run", 2, 1, 1)


def test_without_a_launcher_it_spawns_one():
    assert bench.launch_plan(2, {}, 2) == ("spawn",)
    argv = bench.torchrun_argv(2, ["--gpus", "2", "--steps", "3"])
    assert argv[:2] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in argv
    assert argv[argv.index("--nproc-per-node") + 1] == "2"
    assert argv[-4:] == ["--gpus", "2", "--steps", "3"] and argv[-5].endswith("bench.py")


def test_gpus_2_on_a_box_without_two_devices_exits_nonzero():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode != 0
    assert "HIP device" in r.stderr and not r.stdout.strip()


def test_gpus_2_rendezvous_over_gloo():
    """The re-exec under torch.distributed.run, world size 2, CPU: both ranks meet and rank 0 reports n_gpus 2."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out == {"launch_check": True, "n_gpus": 2, "rank_sum": 1.0}

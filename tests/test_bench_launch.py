"""``python bench.py --gpus N`` with no launcher in front of it: the script starts its
own rank processes, they meet over matchering_amd.ranks, time their steps between two barriers, take the
maximum over ranks, and rank 0 prints ONE JSON line.  ``--stand-in`` replaces the GPU workload by a 1 ms
sleep, so the whole path -- the one the driver's 2 / 4 / 8-GPU runs take -- runs here without a GPU."""

import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT


def _bench(*flags, env=None, timeout=120):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--stand-in", "--steps", "5", "--warmup", "2",
           "--spinup", "0.01", *flags]
    clean = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=dict(clean, **(env or {})))
    assert run.returncode == 0, run.stderr[-2000:]
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, run.stdout                  # one line, from rank 0 only
    return json.loads(lines[0])


@pytest.mark.parametrize("world", [2, 4, 8])
def test_self_launched_ranks_print_one_line(world):
    line = _bench("--gpus", str(world))
    assert line["n_gpus"] == world and line["launch"] == "self" and line["scaling"] == "weak"
    assert line["steps"] == 5 and line["warmup"] == 2
    assert line["rendezvous"]["ranks_seen"] == world
    assert len(line["rank_seconds"]) == world and all(0.004 < s < 5.0 for s in line["rank_seconds"])
    # value = frames of ALL ranks / the slowest rank's time
    per_rank = line["config"]["frames_per_gpu_per_step"]
    assert line["value"] == pytest.approx(world * per_rank * 5 / (line["ms_per_step"] * 5e-3) / 1e6, rel=1e-3)
    assert line["ms_per_step"] * 5e-3 >= max(line["rank_seconds"]) - 1e-3
    # which device every rank sat on (PCI address on a GPU box; a stand-in name here), and what the roofline is taken against
    assert len(line["rank_gpus"]) == world and len(set(line["rank_gpus"])) == line["physical_gpus"] == world
    assert line["shared_gpu"] is False
    assert line["pipeline_hbm_model"]["measured_bytes"] is None          # (no profiler without a GPU)
    # the N = 1 point of the SAME workload, taken by rank 0 alone after the timed region
    n1 = line["n1_same_workload"]
    assert n1["workload"] == "stand_in" and n1["unit"] == "Msamples/s"
    assert n1["value"] == pytest.approx(per_rank / (n1["ms_per_step"] * 1e-3) / 1e6, rel=1e-3)
    assert 0.5 < line["value"] / (world * n1["value"]) < 1.5              # a curve of one workload: efficiency near 1


def test_config_5_per_gpu_share_on_eight_ranks():
    """the line a driver's 8-GPU run of config #5's per-GPU share would print, without a GPU."""
    line = _bench("--gpus", "8", "--workload", "96k_16k_x16_full")
    assert line["n_gpus"] == 8 and len(line["rank_gpus"]) == 8 and line["physical_gpus"] == 8
    cfg = line["config"]
    assert cfg["pairs_per_gpu_per_step"] == 16 and cfg["frames_per_gpu_per_step"] == 16 * 240 * 96000
    assert "96k_16k_x16_full" in cfg["workload"] and cfg["parallelism"] == "pairs x128"       # batch 128 on 8 GPUs
    gathered = line["rccl_fir_allgather"]
    assert gathered["bytes_per_rank"] == 2 * 16384 * 4 == 131072 and gathered["ok"] and gathered["ranks_seen"] == 8
    assert gathered["distinct_tables"] == 8 and "rccl_init_s" in line
    n1 = line["n1_same_workload"]
    assert n1["unit"] == "Msamples/s" and 0.5 < line["value"] / (8 * n1["value"]) < 1.5
    rep = line["ms_per_step_repeats"]
    assert rep["blocks"] == 6 and rep["min"] <= rep["median"] <= rep["max"] and rep["min"] <= line["ms_per_step"] <= rep["max"]


def test_single_rank_needs_no_rendezvous():
    line = _bench("--gpus", "1")
    assert line["n_gpus"] == 1 and line["launch"] == "single" and "rendezvous" not in line
    assert line["rank_gpus"] == ["stand-in-0"] and line["physical_gpus"] == 1 and "n1_same_workload" not in line


def test_a_launcher_started_rank_does_not_launch_again():
    """Under torch.distributed.run every process IS a rank (WORLD_SIZE is set): a world of one must not spawn."""
    line = _bench("--gpus", "1", env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert line["n_gpus"] == 1


def test_world_size_mismatch_is_refused():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--stand-in", "--gpus", "2"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=60,
                         env=dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0"))
    assert run.returncode != 0 and "--gpus 2" in run.stderr

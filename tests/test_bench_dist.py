"""bench.py's multi-rank code path on CPU: `python bench.py --gpus 2 --dry-run` spawns two ranks itself (no torchrun),
forms a gloo group, shards the pairs round-robin, runs the barrier / max-over-ranks timing and gathers the per-pair
records onto rank 0 - everything the 8-GPU run does except the GPU step, which a stub replaces."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(*flags, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--dry-run", *flags], capture_output=True, text=True,
                       timeout=300, env=e)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout       # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_self_spawned_two_ranks_weak_scaling():
    out = _run("--gpus", "2", "--steps", "3", "--warmup", "1", "--pairs", "4")
    assert out["n_gpus"] == 2 and out["ranks_seen"] == [0, 1] and out["scaling"] == "weak"
    assert out["steps"] == 3 and out["records_gathered"] == 8 and out["success_rate"] == 1.0
    # pair i lives on rank i % 2 and the gather restores the global order: scene id == pair id in weak mode
    assert out["pose_tx"] == [float(i) for i in range(8)]
    assert out["record_rank"] == [float(i % 2) for i in range(8)]
    assert out["value"] > 0 and abs(out["per_rank_pairs_per_s"] * 2 - out["value"]) < 1e-6 * out["value"]


def test_fixed_total_split_is_ragged_and_complete():
    """configs[3]: 545 pairs do not divide by 8; here 11 pairs over 2 ranks in batches of 4 -> rank 0 walks 4 + 2,
    rank 1 walks 4 + 1, and all 11 records arrive in global order."""
    out = _run("--gpus", "2", "--steps", "2", "--warmup", "0", "--pairs", "4", "--total-pairs", "11", "--pool", "3")
    assert out["scaling"] == "strong" and out["records_gathered"] == 11 and out["ranks_seen"] == [0, 1]
    assert out["record_rank"] == [float(i % 2) for i in range(11)]
    # scene of pair i on rank r: r + 2 * ((i // 2) % pool)
    assert out["pose_tx"] == [float((i % 2) + 2 * ((i // 2) % 3)) for i in range(11)]
    assert out["steps"] == 2 * 2          # two passes over rank 0's two batches


def test_single_rank_and_torchrun_environment():
    out = _run("--gpus", "1", "--steps", "2", "--warmup", "1", "--pairs", "3")
    assert out["n_gpus"] == 1 and out["ranks_seen"] == [0] and out["records_gathered"] == 3
    # under a launcher the ranks come from the environment: world size 1 with the variables set behaves the same
    out = _run("--gpus", "1", "--steps", "1", "--warmup", "0", "--pairs", "2",
               env={"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29533"})
    assert out["n_gpus"] == 1 and out["records_gathered"] == 2


def test_eight_ranks_lokitti_545_split():
    """configs[3] at its real shape: 545 pairs over 8 ranks in batches of 64, one process per rank as
    `scripts/test_kitti.sh:45-75` runs one process per GPU.  Pair i lives on rank i % 8 (ranks 0 holds 69 pairs, the others
    68), every rank's ragged last batch is walked, and rank 0 gets all 545 records back in global order."""
    out = _run("--gpus", "8", "--steps", "1", "--warmup", "0", "--pairs", "64", "--total-pairs", "545")
    assert out["n_gpus"] == 8 and out["ranks_seen"] == list(range(8)) and out["scaling"] == "strong"
    assert out["records_gathered"] == 545 and out["success_rate"] == 1.0
    rr = out["record_rank"]
    assert rr == [float(i % 8) for i in range(545)]
    counts = [rr.count(float(r)) for r in range(8)]
    assert counts == [69] + [68] * 7
    assert out["steps"] == 2                                    # rank 0: one batch of 64 and a ragged one of 5
    # whole-job throughput = all 545 pairs over the slowest rank's time
    assert out["value"] > 0 and out["config"]["pairs_per_step"] == 64


def test_scene_workers_are_divided_by_the_ranks_of_the_node():
    """8 ranks on a node share its cores: each rank's scene-generator pool is cores // 8 (at least 1, at most 8)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    cores = bench.usable_cores()
    assert bench.scene_workers(8, 64) == max(1, min(8, cores // 8, 64))
    assert bench.scene_workers(1, 64) == max(1, min(8, cores, 64))
    assert bench.scene_workers(8, 1) == 1

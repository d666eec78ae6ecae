"""bench.py as the driver launches it: `--gpus N` without WORLD_SIZE in the environment starts the N ranks itself
(rank r on GPU r, rendezvous on 127.0.0.1, counts reduced over the ranks, one JSON line from rank 0)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]      # ONE line, from rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 3])
def test_gpus_flag_forks_the_ranks(n):
    """no GPU needed: --dry-run exercises launch, rendezvous (gloo) and the count reduction only"""
    out = run_bench("--gpus", str(n), "--dry-run", timeout=300)
    assert out["n_gpus"] == n and out["dry_run"] is True
    tri = n * (n + 1) // 2   # rank r contributes (r+1, 10(r+1), 100(r+1))
    assert out["feature_counts"] == [tri, 10 * tri, 100 * tri]
    assert out["max_elapsed_s"] == float(n)       # max over ranks of 1 + rank


@pytest.mark.gpu
def test_two_ranks_equal_one_rank_over_the_same_frames():
    """2 ranks x 3 frames (both on device 0: a functional check of the N>1 path on a 1-GPU box) must count what one rank
    counts on the same 6 frames"""
    common = ["--no-cpu", "--steps", "1", "--warmup", "1", "--inner", "1"]
    two = run_bench("--gpus", "2", "--share-device", "--batch", "3", *common)
    one = run_bench("--gpus", "1", "--batch", "6", *common)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["feature_counts"] == one["config"]["feature_counts"]
    assert all(v > 0 for v in one["config"]["feature_counts"].values())
    assert two["metric"] == one["metric"] and "roofline" in two


@pytest.mark.gpu
def test_stream_config_shards_frames_and_checks_a_sample():
    """configs[4] in miniature: 10 frames, batches of 4 (a ragged last batch), sharded over 2 ranks; per-frame counts are
    gathered in stream order and equal the single-rank run's; the sampled parity check is clean"""
    common = ["--config", "5", "--frames", "10", "--batch", "4", "--warmup", "1", "--no-cpu"]
    one = run_bench("--gpus", "1", *common)
    two = run_bench("--gpus", "2", "--share-device", *common)
    for r in (one, two):
        assert r["config"]["frames_total"] == 10 and r["config"]["per_frame_counts_gathered"] == 10
    assert one["config"]["per_frame_counts_checksum"] == two["config"]["per_frame_counts_checksum"]
    assert one["config"]["feature_counts"] == two["config"]["feature_counts"]
    par = one["parity"]["parity_sample"]
    assert par["frames_checked"] >= 1 and par["frames_with_different_corner_coordinates"] == 0
    assert par["frames_with_strength_rel_err_above_1e-4"] == 0 and par["canny_mismatching_pixels_total"] == 0
    assert par["frames_whose_streamed_counts_differ"] == 0

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
    # the per-rank table: one row per rank, its frames, seconds and rate, and what the NUMA pinning did
    rows = two["config"]["per_rank"]
    assert [r["rank"] for r in rows] == [0, 1] and all(r["frames"] == 3 and r["s"] > 0 and r["Mpixels_per_s"] > 0 and "pinned" in r["numa"] for r in rows)
    assert sum(r["pixels"] for r in rows) == 6 * 3840 * 2160
    assert len(one["config"]["per_rank"]) == 1 and one["config"]["delivery"] == "resident"


@pytest.mark.gpu
def test_h2d_delivery_gives_the_same_counts():
    """--h2d: the frames come from pinned host memory, every pass uploads its batch on a copy stream beside the kernels of the
    batch before -- same feature counts as the resident run, in the default workload and in the stream (ragged last batch, two
    ranks), and the line says how the frames were delivered"""
    common = ["--no-cpu", "--steps", "2", "--warmup", "1", "--inner", "3", "--batch", "3"]
    res = run_bench("--gpus", "1", *common)
    h2d = run_bench("--gpus", "1", "--h2d", *common)
    assert h2d["config"]["feature_counts"] == res["config"]["feature_counts"] and h2d["config"]["delivery"].startswith("h2d")
    s_common = ["--config", "5", "--frames", "10", "--batch", "4", "--warmup", "1", "--no-cpu"]
    s_res = run_bench("--gpus", "1", *s_common)
    s_h2d = run_bench("--gpus", "2", "--share-device", "--h2d", *s_common)
    assert s_h2d["config"]["per_frame_counts_checksum"] == s_res["config"]["per_frame_counts_checksum"]
    assert s_h2d["config"]["feature_counts"] == s_res["config"]["feature_counts"]
    assert [r["frames"] for r in s_h2d["config"]["per_rank"]] == [5, 5]


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


def test_a_crashing_rank_does_not_hang_the_launcher():
    """rank 1 of 3 exits before the rendezvous: the launcher stops the two ranks that would wait for it and returns rank 1's
    code, within seconds"""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--dry-run", "--crash-rank", "1"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert p.returncode == 3, (p.returncode, p.stderr[-1000:])
    assert "rank 1 exited with code 3" in p.stderr
    assert time.time() - t0 < 200
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]   # no result line from a broken job


@pytest.mark.gpu
def test_more_ranks_than_gpus_is_an_error_not_a_hang():
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--no-cpu", "--steps", "1", "--inner", "1", "--batch", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    assert "is more than this node has" in p.stderr


@pytest.mark.gpu
def test_nccl_world1_collectives():
    """N = 1 forms the group of N > 1 (gloo for host tensors, RCCL for device tensors): the counts are reduced over gloo, and ONE
    RCCL all_gather of the count vectors runs after every clock has stopped -- the line says how many ranks it saw.  The three
    collectives of image_amd/stream.py also run on device tensors over a plain nccl group."""
    out = run_bench("--config", "5", "--frames", "6", "--batch", "4", "--warmup", "1", "--no-cpu")
    assert out["config"]["rccl_ranks_seen"] == 1 and "ONE RCCL all_gather" in out["config"]["collectives"] and "sums equal" in out["config"]["collectives"], out["config"].get("collectives")
    assert out["config"]["per_frame_counts_gathered"] == 6
    # the same reductions directly, on device tensors
    import torch
    import torch.distributed as dist
    from image_amd import stream
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ["MASTER_PORT"] = str(29500 + os.getpid() % 2000)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        assert dist.get_backend() == "nccl"
        c = torch.tensor([3, 5, 7], dtype=torch.int64, device="cuda:0")
        tot, dt = stream.reduce_counts(c, 1.5, dist)
        assert tot.tolist() == [3, 5, 7] and dt == 1.5 and tot.is_cuda
        fc = torch.arange(10, dtype=torch.int64, device="cuda:0").reshape(2, 5)
        assert torch.equal(stream.gather_frame_counts(fc, dist), fc)
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_scale_run_sheet_on_one_device(tmp_path):
    """scripts/scale_8gpu.sh end to end in miniature: N = 1 and 2 (the two ranks share device 0, so no RCCL communicator can
    form: the sheet records 0 ranks seen), resident and --h2d, the default workload and the stream; the sheet's own assertions
    (N = 1 against a plain line, stream counts independent of the sharding) must hold and the JSON carry every run"""
    out = tmp_path / "scale.json"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    env.update(GPUS="1 2", SHARE="1", FRAMES="12", BATCH="3", STEPS="2", INNER="2")
    p = subprocess.run(["bash", os.path.join(ROOT, "scripts", "scale_8gpu.sh"), str(out)], env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    sheet = json.loads(out.read_text())
    assert len(sheet["runs"]) == 8 and {r["n_gpus"] for r in sheet["runs"]} == {1, 2} and {r["delivery"] for r in sheet["runs"]} == {"resident", "h2d"}
    assert all(len(r["per_rank"]) == r["n_gpus"] and r["value"] > 0 for r in sheet["runs"])

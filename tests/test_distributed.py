"""N>1 path on CPU: two `gloo` ranks shard a small frame stream exactly like bench.py shards it over GPUs
(per-rank contiguous blocks, no data-path collective, one sum of feature counts + one max of elapsed time)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from image_amd import stream, synth

N_FRAMES, NX, NY, SEED0 = 7, 96, 64, 50000


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _counts(first, n):
    c = np.zeros(2, np.int64)
    for f in range(first, first + n):
        img = synth.frame(stream.frame_seed(SEED0, f), NX, NY)
        c[0] += oracle.fast9(img, 20, True).shape[0]
        c[1] += oracle.harris(img.astype(np.float32)).shape[0]
    return c


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, n = stream.rank_block(N_FRAMES, rank, world)
    counts = torch.from_numpy(_counts(first, n))
    total, tmax = stream.reduce_counts(counts, elapsed_s=0.25 + rank, dist=dist)
    if rank == 0:
        q.put((total.tolist(), tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_rank_blocks_partition_the_stream():
    for n in (0, 1, 7, 8, 1250, 10000):
        for world in (1, 2, 3, 8):
            blocks = [stream.rank_block(n, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and sum(c for _, c in blocks) == n
            for (s0, c0), (s1, _) in zip(blocks, blocks[1:]):
                assert s1 == s0 + c0
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1


@pytest.mark.slow
def test_two_gloo_ranks_reproduce_the_single_process_counts():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    total, tmax = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert total == _counts(0, N_FRAMES).tolist()
    assert tmax == 1.25  # max over ranks

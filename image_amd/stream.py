"""Sharding of a frame stream over the GPUs of one node.

Every frame is an independent unit: rank r of W owns a contiguous block of the stream, keeps its inputs,
intermediates and feature lists local, and the only collective of the whole path is the reduction of a few
int64 feature counts (plus the max-over-ranks of the elapsed time for reporting).  No halo, no exchange.
"""
from __future__ import annotations


def rank_block(n_frames: int, rank: int, world: int) -> tuple[int, int]:
    """(first frame, number of frames) of rank `rank`: blocks of ceil/floor(n/W), earlier ranks take the remainder."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    q, r = divmod(int(n_frames), world)
    start = rank * q + min(rank, r)
    return start, q + (1 if rank < r else 0)


def frame_seed(seed0: int, frame: int) -> int:
    """Seed of global frame index `frame` (config 5: G(seed = 50000 + f))."""
    return int(seed0) + int(frame)


def device_backend(dist) -> str:
    """the backend that serves device tensors in the default process group ("nccl" = RCCL, or "gloo")"""
    return "nccl" if "nccl" in str(dist.get_backend()) else "gloo"


def reduce_counts(counts, elapsed_s: float, dist=None, device=None):
    """Sum the per-rank feature counts and take the max elapsed time over ranks.

    counts: 1-D int64 torch tensor on the rank's device.  Returns (total_counts tensor, max elapsed float)."""
    import torch
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=counts.device if device is None else device)
    if dist is not None and dist.is_initialized():   # one rank included: the reductions of N = 1 run through the communicator too
        if device_backend(dist) == "gloo" and counts.is_cuda:  # functional checks: gloo reduces host tensors
            c, tt = counts.cpu(), t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return c.to(counts.device), float(tt.item())
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return counts, float(t.item())


def gather_frame_counts(frame_counts, dist=None):
    """configs[4]: every rank holds a [k, n_local] int64 tensor of per-frame counts of its contiguous block of the
    stream; returns the [k, n_total] tensor in stream order on every rank (one all_gather of padded blocks)."""
    import torch
    if dist is None or not dist.is_initialized():
        return frame_counts
    world = dist.get_world_size()
    use_host = device_backend(dist) == "gloo" and frame_counts.is_cuda
    fc = frame_counts.cpu() if use_host else frame_counts
    n = torch.tensor([fc.shape[1]], dtype=torch.int64, device=fc.device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n)
    n_max = int(max(int(x) for x in ns))
    pad = torch.zeros((fc.shape[0], n_max), dtype=torch.int64, device=fc.device)
    pad[:, :fc.shape[1]] = fc
    parts = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    out = torch.cat([p[:, :int(k)] for p, k in zip(parts, ns)], dim=1)
    return out.to(frame_counts.device) if use_host else out

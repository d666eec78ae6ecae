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


def has_host_side(dist) -> bool:
    """the process group serves HOST tensors through gloo ("gloo" alone, or the mixed "cpu:gloo,cuda:nccl" group bench.py forms)"""
    return "gloo" in str(dist.get_backend())


def reduce_counts(counts, elapsed_s: float, dist=None, device=None, on_host: bool = False):
    """Sum the per-rank feature counts and take the max elapsed time over ranks.

    counts: 1-D int64 torch tensor on the rank's device.  Returns (total_counts tensor, max elapsed float).
    on_host: reduce copies of the few bytes over the group's gloo side when it has one -- no RCCL communicator has to exist
    (a live one costs the kernels beside it ~12 % on this hardware, profiles/r03); rccl_gather_counts() is the device-side
    collective north_star asks for, run once when every clock has stopped."""
    import torch
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=counts.device if device is None else device)
    if dist is not None and dist.is_initialized():   # one rank included: the reductions of N = 1 run through the group too
        if counts.is_cuda and (device_backend(dist) == "gloo" or (on_host and has_host_side(dist))):
            c, tt = counts.cpu(), t.cpu()
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            return c.to(counts.device), float(tt.item())
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return counts, float(t.item())


def rccl_gather_counts(counts, dist):
    """ONE all_gather of every rank's feature-count vector on DEVICE tensors ("RCCL only to gather feature counts"); returns the
    [world, k] tensor on the host, or None where the group has no device backend (ranks sharing one device: gloo only)."""
    import torch
    if dist is None or not dist.is_initialized() or device_backend(dist) != "nccl" or not counts.is_cuda:
        return None
    parts = [torch.zeros_like(counts) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, counts.contiguous())
    return torch.stack(parts).cpu()


def gather_frame_counts(frame_counts, dist=None, on_host: bool = False):
    """configs[4]: every rank holds a [k, n_local] int64 tensor of per-frame counts of its contiguous block of the
    stream; returns the [k, n_total] tensor in stream order on every rank (one all_gather of padded blocks).
    on_host: as reduce_counts."""
    import torch
    if dist is None or not dist.is_initialized():
        return frame_counts
    world = dist.get_world_size()
    use_host = frame_counts.is_cuda and (device_backend(dist) == "gloo" or (on_host and has_host_side(dist)))
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

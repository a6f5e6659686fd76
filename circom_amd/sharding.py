"""Multi-GPU sharding of a batch of independent instances.

Every input vector is an independent unit: rank r of W takes a contiguous slice, the schedule / constants /
R1CS are replicated, nothing is exchanged while generating or checking.  The ONE exchange is the final
gather of the per-instance status words (4 B each) and public signals (32 B per public signal) to rank 0 —
over RCCL on GPUs (`backend="nccl"`), over gloo in the CPU tests.  Full witnesses are not gathered (config 4 would need 256 GiB at the root);
each rank serves/writes its own."""
from __future__ import annotations


def shard_range(total: int, rank: int, world: int):
    """Contiguous [lo, hi) slice of `total` instances for `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_status(status, dist=None, rank: int = 0, world: int = 1, dst: int = 0):
    """status: 1-D torch tensor (same dtype/device on every rank; lengths may differ).  Returns the
    concatenation in rank order on `dst`, None elsewhere."""
    import torch
    if dist is None or world == 1:
        return status
    n = torch.tensor([status.numel()], device=status.device, dtype=torch.int64)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(sizes)
    pad = torch.zeros(m, device=status.device, dtype=status.dtype)
    pad[:status.numel()] = status
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return torch.cat([b[:k] for b, k in zip(bufs, sizes)])


def gather_rows(rows, dist=None, rank: int = 0, world: int = 1, dst: int = 0):
    """rows: torch tensor [n, ...] (e.g. the public signals [n][n_public][32] of this rank's instances; n may differ
    between ranks).  Returns the concatenation along dim 0 in rank order on `dst`, None elsewhere."""
    import torch
    if dist is None or world == 1:
        return rows
    flat = rows.reshape(rows.shape[0], -1)
    width = flat.shape[1]
    got = gather_status(flat.reshape(-1), dist, rank, world, dst)
    if got is None:
        return None
    return got.reshape((-1,) + tuple(rows.shape[1:])) if width else got.reshape((0,) + tuple(rows.shape[1:]))

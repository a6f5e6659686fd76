"""Multi-GPU sharding of a batch of independent instances.

Every input vector is an independent unit: rank r of W takes a contiguous slice, the schedule / constants /
R1CS are replicated, nothing is exchanged while generating or checking.  The ONE exchange is the final
gather of the per-instance status words (4 B each) and public signals (32 B per public signal; ONE BIT per signal when every
public signal of every instance is 0 or 1 - the digests of a 2^21-instance SHA-256 shard are 64 MB instead of 17 GB per rank)
to rank 0 — over RCCL on GPUs (`backend="nccl"`), over gloo in the CPU tests.  Full witnesses are not gathered (config 4 would need 256 GiB at the root);
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


def pack_bit_rows(rows):
    """rows: [n][k][32] uint8 field elements.  Returns [n][ceil(k / 8)] uint8 with bit j of byte i = element 8 i + j when EVERY
    element is 0 or 1 (the public signals of a bit-level circuit: SHA-256 digests), else None."""
    import torch
    if rows.numel() == 0 or bool(rows[:, :, 1:].any().item()) or bool((rows[:, :, 0] > 1).any().item()):
        return None
    n, k = rows.shape[0], rows.shape[1]
    bits = rows[:, :, 0]
    pad = (-k) % 8
    if pad:
        bits = torch.cat([bits, torch.zeros((n, pad), dtype=bits.dtype, device=bits.device)], dim=1)
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=bits.device)
    return (bits.reshape(n, -1, 8).to(torch.int32) * w).sum(dim=2).to(torch.uint8)


def unpack_bit_rows(packed, k: int):
    """inverse of pack_bit_rows: [n][k][32] uint8"""
    import torch
    n = packed.shape[0]
    sh = torch.arange(8, dtype=torch.int32, device=packed.device)
    bits = ((packed.to(torch.int32).unsqueeze(2) >> sh) & 1).reshape(n, -1)[:, :k].to(torch.uint8)
    out = torch.zeros((n, k, 32), dtype=torch.uint8, device=packed.device)
    out[:, :, 0] = bits
    return out


def gather_public(pub, dist=None, rank: int = 0, world: int = 1, dst: int = 0):
    """The public signals of this rank's instances ([n][n_public][32]) to `dst`: as packed bits when every rank's are all 0 / 1
    (agreed on with one all-reduce), as field elements otherwise.  Returns (rows on dst | None, "bits" | "elements")."""
    import torch
    packed = pack_bit_rows(pub) if pub.shape[1] else None
    ok = torch.tensor([1 if packed is not None else 0], device=pub.device, dtype=torch.int32)
    if dist is not None and world > 1:
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    if int(ok.item()):
        return gather_rows(packed, dist, rank, world, dst), "bits"
    return gather_rows(pub, dist, rank, world, dst), "elements"

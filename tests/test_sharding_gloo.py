"""The N>1 path of bench.py on CPU: two processes over gloo shard a batch and gather the status words and the
public signals (the one exchange of the job)."""
import os
import socket
import sys
from pathlib import Path

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _worker(rank, world, port, total, out_dir):
    sys.path.insert(0, str(ROOT))
    from circom_amd.sharding import shard_range, gather_status, gather_rows
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = shard_range(total, rank, world)
    # a fake per-instance status: instance id in the low bits, one deliberately failing instance per rank
    st = torch.arange(lo, hi, dtype=torch.int32) * 8
    st[0] += 1
    got = gather_status(st, dist, rank, world)
    # fake public signals: [n][2][32] bytes derived from the instance id
    ids = torch.arange(lo, hi, dtype=torch.int64)
    pub = ((ids[:, None, None] * 7 + torch.arange(2)[None, :, None] * 3 + torch.arange(32)[None, None, :]) % 251).to(torch.uint8)
    gpub = gather_rows(pub, dist, rank, world)
    # the compact exchange: public signals that are all 0 / 1 on EVERY rank travel as bits (11 signals -> 2 bytes per instance);
    # one rank with a field-sized value makes every rank send elements
    from circom_amd.sharding import gather_public
    bitpub = torch.zeros((hi - lo, 11, 32), dtype=torch.uint8)
    bitpub[:, :, 0] = ((ids[:, None] >> torch.arange(11)[None, :]) & 1).to(torch.uint8)
    gbits, form = gather_public(bitpub, dist, rank, world)
    mixed = bitpub.clone()
    if rank == 1:
        mixed[0, 3, 5] = 9
    gmixed, form2 = gather_public(mixed, dist, rank, world)
    assert (form, form2) == ("bits", "elements")
    if rank == 0:
        torch.save(got, os.path.join(out_dir, "gathered.pt"))
        torch.save(gpub, os.path.join(out_dir, "public.pt"))
        torch.save((gbits, gmixed), os.path.join(out_dir, "public_bits.pt"))
    else:
        assert got is None and gpub is None and gbits is None and gmixed is None
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges_cover_batch():
    from circom_amd.sharding import shard_range
    for total in (0, 1, 7, 8192, 65537):
        for world in (1, 2, 3, 8):
            parts = [shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_gloo_status_gather(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    total = 1001           # ragged on purpose
    mp.spawn(_worker, args=(2, port, total, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(tmp_path / "gathered.pt")
    want = torch.arange(0, total, dtype=torch.int32) * 8
    want[0] += 1
    want[501] += 1
    assert torch.equal(got, want)
    gpub = torch.load(tmp_path / "public.pt")
    ids = torch.arange(0, total, dtype=torch.int64)
    wpub = ((ids[:, None, None] * 7 + torch.arange(2)[None, :, None] * 3 + torch.arange(32)[None, None, :]) % 251).to(torch.uint8)
    assert gpub.shape == (total, 2, 32) and torch.equal(gpub, wpub)
    from circom_amd.sharding import unpack_bit_rows
    gbits, gmixed = torch.load(tmp_path / "public_bits.pt")
    assert gbits.shape == (total, 2) and gbits.dtype == torch.uint8
    full = unpack_bit_rows(gbits, 11)
    assert full.shape == (total, 11, 32) and not full[:, :, 1:].any()
    assert torch.equal(full[:, :, 0].to(torch.int64), (ids[:, None] >> torch.arange(11)[None, :]) & 1)
    assert gmixed.shape == (total, 11, 32) and gmixed[501, 3, 5] == 9 and torch.equal(gmixed[:, :, 0], full[:, :, 0])


def _cabi_worker(rank, world, port, total, shared_dir):
    """N>1 path through the C ABI: rank 0 compiles the circuit once, the others load the artefacts from the shared
    directory; every rank stages the inputs of ITS shard in a host-only batch (device = -1: the library validates and
    stages, nothing computes without a GPU), then the per-instance words are gathered on rank 0."""
    sys.path.insert(0, str(ROOT))
    from circom_amd import runtime as rt
    from circom_amd.sharding import shard_range, gather_status
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = lambda ext: os.path.join(shared_dir, "poseidon2" + ext)
    if rank == 0:
        from circom_amd.compiler import compile_program
        from circom_amd.frontend.dsl import Program
        from circom_amd.circuits.poseidon import Poseidon
        compile_program(Program(Poseidon(2)), shared_dir, "poseidon2", sym=False, strands=(1,))
    dist.barrier()
    c = rt.Circuit(p(".cwt"), p(".dat"), p(".r1cs"))             # every rank loads the same files
    lo, hi = shard_range(total, rank, world)
    b = c.batch(hi - lo, device=-1)
    for i in range(lo, hi):
        b.set_inputs_json(i - lo, '{"inputs": ["%d", "%d"]}' % (i, 7 * i + 1))
    assert all(b.remaining_inputs(k) == 0 for k in range(hi - lo))
    try:
        b.run()
        raise AssertionError("a host-only batch must not compute")
    except rt.CwError:
        pass
    # what each rank staged for its instances travels to rank 0 the way the status words do
    words = torch.tensor([b.staged_input(k, 1) % (1 << 31) for k in range(hi - lo)], dtype=torch.int32)
    got = gather_status(words, dist, rank, world)
    if rank == 0:
        torch.save(got, os.path.join(shared_dir, "staged.pt"))
    dist.barrier()
    b.close(); c.close()
    dist.destroy_process_group()


def test_two_rank_gloo_through_the_c_abi(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    total = 37
    mp.spawn(_cabi_worker, args=(2, port, total, str(tmp_path)), nprocs=2, join=True)
    got = torch.load(tmp_path / "staged.pt")
    assert got.tolist() == [(7 * i + 1) % (1 << 31) for i in range(total)]

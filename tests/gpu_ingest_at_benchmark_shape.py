#!/usr/bin/env python3
"""GPU check run by tests/test_baseline_configs.py::test_metric_workload_32_byte_ingest_at_2M in a process of its own (torch first,
then the library): 2^21 instances of the 1 020 832-constraint SHA-256 through the boundary's own input format - the canonical
32-byte image of all 2^21 x 2 048 inputs, 137 GB, built on the device as bench.py does - i.e. through `cw_bits_ingest_kernel`,
the dominant kernel of the benchmark step, at the benchmark shape.  The two reference goldens sit inside the batch (their full
32 MB `.wtns` compared through the recorded hash), every digest must equal the packed-input run of the same batch, and an
instance with ONE non-boolean input must be the only one sent to the 256-bit fallback.  Prints INGEST OK."""
import hashlib
import json
import os
import sys

import torch
torch.cuda.init()                                    # torch's HIP runtime first (it cannot initialise after the system's)
import numpy as np                                   # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
import bench                                         # noqa: E402
from circom_amd import runtime as rt                 # noqa: E402


def main(tmp):
    GOLD = json.load(open(os.path.join(HERE, "golden", "reference_wtns.json")))
    name, B = "sha256_2048", bench.JIT_BATCH
    cache = os.path.join(ROOT, "gpurun_in", "cache")
    cp, _, _ = bench.get_compiled(name, B, cache if os.path.isdir(cache) else tmp, 0, None)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    dev = torch.device("cuda", 0)
    vecs = GOLD["cases"][name]["vectors"][:2]
    at = [5, B - 3]
    rng = np.random.default_rng(9)
    G = B // 64
    masks = rng.integers(0, 1 << 63, size=(G, c.n_inputs), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(G, c.n_inputs), dtype=np.uint64)
    for pos, vec in zip(at, vecs):
        g, i = pos // 64, np.uint64(pos % 64)
        bitsv = np.array([int(v) for v in vec["inputs"]], dtype=np.uint64)
        masks[g] = (masks[g] & ~(np.uint64(1) << i)) | (bitsv << i)
    d_m = torch.from_numpy(masks.view(np.int64)).to(dev)                          # [G][n_inputs]
    # the packed run: what the digests must be
    b = c.batch(B)
    assert b.bitmode and b.jit
    b.set_inputs_bits_device(d_m.data_ptr())
    b.run(); b.check_r1cs(); b.sync()
    assert (b.status() == 0).all()
    pub = torch.empty((B, c.n_public, 32), dtype=torch.uint8, device=dev)
    b.public_signals_device(pub.data_ptr()); b.sync()
    assert not bool(pub[:, :, 1:].any().item())
    want_bits = pub[:, :, 0].clone()
    del pub
    b.close()                                                                     # (its table: the image needs the room)
    torch.cuda.empty_cache()
    # the canonical image, built on the device
    d_in = torch.zeros((B, c.n_inputs, 32), dtype=torch.uint8, device=dev)
    j = torch.arange(64, device=dev, dtype=torch.int64).view(1, 64, 1)
    for g0 in range(0, G, 1024):
        g1 = min(G, g0 + 1024)
        d_in[g0 * 64:g1 * 64, :, 0] = ((d_m[g0:g1].unsqueeze(1) >> j) & 1).to(torch.uint8).reshape((g1 - g0) * 64, c.n_inputs)
    del d_m
    b2 = c.batch(B)
    b2.set_inputs_device(d_in.data_ptr())
    b2.run(); b2.check_r1cs(); b2.sync()
    assert (b2.status() == 0).all()
    pub = torch.empty((B, c.n_public, 32), dtype=torch.uint8, device=dev)
    b2.public_signals_device(pub.data_ptr()); b2.sync()
    assert not bool(pub[:, :, 1:].any().item()) and torch.equal(pub[:, :, 0], want_bits), "32-byte ingest and packed inputs disagree"
    del pub, want_bits
    torch.cuda.empty_cache()
    for pos, vec in zip(at, vecs):
        p = os.path.join(tmp, "h%d.wtns" % pos)
        b2.write_wtns(pos, p)
        raw = open(p, "rb").read()
        os.unlink(p)
        assert len(raw) == int(vec["wtns_len"]) and hashlib.sha256(raw).hexdigest() == vec["wtns_sha256"], "golden %d" % pos
    # one instance whose input is NOT a bit in one place: the ingest must send exactly that instance to the 256-bit fallback
    d_in[B // 3, 7, 0] = 2
    b2.set_inputs_device(d_in.data_ptr())
    b2.run(); b2.check_r1cs(); b2.sync()
    st = b2.status()
    assert (np.delete(st, B // 3) == 0).all()
    del d_in
    b2.close()
    c.close()
    print("INGEST OK")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "/tmp")

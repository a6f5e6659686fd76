"""`.wtnsb` (circom_amd/wtnsb.py): the reader / expander against containers built by hand in both table layouts and both
kinds; the writer (cw_write_wtnsb) is exercised on the GPU (tests/test_bitplane.py, tests/test_gpu_parity.py)."""
import random
import struct

import numpy as np
import pytest

from circom_amd import wtnsb
from circom_amd.hip_elements.writers import wtns_bytes

Q = 21888242871839275222246405745257275088548364400416034343698204186575808495617


def _container(tmp_path, shift, batch, n_wit, bits, wide):
    """bit-plane container: `bits[i][k]` = witness element k of instance i; slots in a scrambled order"""
    r = random.Random(5)
    slots = n_wit + 7
    wslot = r.sample(range(3, slots), n_wit - 1)
    wslot = [1] + wslot                                        # element 0 = the constant 1
    groups = (batch + 63) // 64
    if shift:
        groups = (groups + (1 << shift) - 1) >> shift << shift
    table = np.zeros(groups * slots, dtype=np.uint64)

    def word(g, s):
        return (((g >> shift) * slots + s) << shift) + (g & ((1 << shift) - 1))
    for g in range(groups):
        table[word(g, 1)] = np.uint64(0xFFFFFFFFFFFFFFFF)
    for i in range(batch):
        for k in range(1, n_wit):
            if bits[i][k]:
                table[word(i >> 6, wslot[k])] |= np.uint64(1) << np.uint64(i & 63)
    p = tmp_path / ("s%d.wtnsb" % shift)
    with open(p, "wb") as f:
        f.write(b"wtnb" + struct.pack("<3I", 1, 1, 32) + Q.to_bytes(32, "little") + struct.pack("<2I", n_wit, batch))
        f.write(struct.pack("<QII", slots, shift, groups))
        f.write(np.asarray(wslot, dtype="<u4").tobytes())
        f.write(table.tobytes())
        f.write(struct.pack("<I", len(wide)))
        for inst, vals in wide.items():
            f.write(struct.pack("<I", inst) + b"".join(v.to_bytes(32, "little") for v in vals))
    return p


@pytest.mark.parametrize("shift", [0, 5])
def test_bit_plane_container_expands_to_the_reference_file_layout(tmp_path, shift):
    batch, n_wit = 150, 40
    r = random.Random(shift)
    bits = [[1] + [r.randrange(2) for _ in range(n_wit - 1)] for _ in range(batch)]
    wide = {77: [1] + [r.randrange(Q) for _ in range(n_wit - 1)]}
    w = wtnsb.load(_container(tmp_path, shift, batch, n_wit, bits, wide))
    assert (w.kind, w.batch, w.n_witness, w.prime, w.shift) == (1, batch, n_wit, Q, shift)
    for i in (0, 1, 63, 64, 77, 149):
        want = wide[i] if i in wide else bits[i]
        assert w.expand(i) == wtns_bytes(Q, want), i
    with pytest.raises(IndexError):
        w.expand(batch)


def test_field_element_container(tmp_path):
    batch, n_wit = 5, 9
    r = random.Random(1)
    vals = [[1] + [r.randrange(Q) for _ in range(n_wit - 1)] for _ in range(batch)]
    p = tmp_path / "f.wtnsb"
    with open(p, "wb") as f:
        f.write(b"wtnb" + struct.pack("<3I", 1, 0, 32) + Q.to_bytes(32, "little") + struct.pack("<2I", n_wit, batch))
        for row in vals:
            f.write(b"".join(v.to_bytes(32, "little") for v in row))
    w = wtnsb.load(p)
    for i in range(batch):
        assert w.expand(i) == wtns_bytes(Q, vals[i])


def test_damaged_containers_are_refused(tmp_path):
    p = tmp_path / "x.wtnsb"
    p.write_bytes(b"wtns" + bytes(60))
    with pytest.raises(ValueError):
        wtnsb.load(p)
    p.write_bytes(b"wtnb" + struct.pack("<3I", 9, 1, 32) + bytes(60))
    with pytest.raises(ValueError):
        wtnsb.load(p)

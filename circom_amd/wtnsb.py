"""`<name>.wtnsb`: the witnesses of a whole batch in one compact container (written by cw_write_wtnsb, csrc/cw_host.cpp).

The reference writes one `.wtns` per witness (writeBinWitness, code_producers/src/c_elements/common/main.cpp:288-334): 32
bytes per element.  For a boolean circuit that image is 256 times the information; the container keeps the bit table (1 bit
per distinct signal value and instance) plus the slot of every witness element, and `expand(i)` reproduces the reference's
file for instance i bit for bit.

  "wtnb" | u32 version = 1 | u32 kind (0 field elements, 1 bit planes) | u32 n8 | prime (n8 bytes) | u32 n_witness | u32 batch
  kind 0:  batch x n_witness x n8 bytes (canonical little-endian values, instance-major)
  kind 1:  u64 slots | u32 shift | u32 groups | n_witness x u32 slot | groups x slots x u64 table | u32 n_wide |
           n_wide x { u32 instance | n_witness x n8 bytes }
           table word of (group g, slot s) = (((g >> shift) * slots + s) << shift) + (g & ((1 << shift) - 1)); bit i = instance 64 g + i
"""
from __future__ import annotations

import struct

import numpy as np

MAGIC = b"wtnb"
VERSION = 1


class WitnessBatch:
    def __init__(self, path):
        self.path = str(path)
        with open(self.path, "rb") as f:
            head = f.read(16)
            if head[:4] != MAGIC:
                raise ValueError("not a .wtnsb file")
            version, self.kind, self.n8 = struct.unpack("<3I", head[4:16])
            if version != VERSION or self.kind not in (0, 1):
                raise ValueError("unsupported .wtnsb version / kind")
            self.prime = int.from_bytes(f.read(self.n8), "little")
            self.n_witness, self.batch = struct.unpack("<2I", f.read(8))
            at = 16 + self.n8 + 8
            if self.kind == 0:
                self._values = np.memmap(self.path, dtype=np.uint8, mode="r", offset=at, shape=(self.batch, self.n_witness, self.n8))
                return
            self.slots, self.shift, self.groups = struct.unpack("<QII", f.read(16))
            at += 16
            self.wslot = np.frombuffer(f.read(4 * self.n_witness), dtype="<u4").astype(np.int64)
            at += 4 * self.n_witness
            self._table = np.memmap(self.path, dtype="<u8", mode="r", offset=at, shape=(self.groups * self.slots,))
            f.seek(at + 8 * self.groups * self.slots)
            n_wide, = struct.unpack("<I", f.read(4))
            self.wide = {}
            row = self.n_witness * self.n8
            for _ in range(n_wide):
                inst, = struct.unpack("<I", f.read(4))
                self.wide[inst] = f.read(row)
        if (self.wslot >= self.slots).any() or self.groups * 64 < self.batch:
            raise ValueError("damaged .wtnsb: slot map / group count")

    def element_bits(self, instance: int) -> np.ndarray:
        """0/1 value of every witness element of a bit-plane instance"""
        g, i = instance >> 6, instance & 63
        sh = self.shift
        base = (((g >> sh) * self.slots) << sh) + (g & ((1 << sh) - 1))
        words = self._table[base + (self.wslot << sh)]
        return ((words >> np.uint64(i)) & np.uint64(1)).astype(np.uint8)

    def values(self, instance: int) -> bytes:
        """n_witness x n8 bytes: the canonical little-endian values of one instance"""
        if not 0 <= instance < self.batch:
            raise IndexError(instance)
        if self.kind == 0:
            return self._values[instance].tobytes()
        if instance in self.wide:
            return self.wide[instance]
        out = np.zeros((self.n_witness, self.n8), dtype=np.uint8)
        out[:, 0] = self.element_bits(instance)
        return out.tobytes()

    def expand(self, instance: int) -> bytes:
        """the `.wtns` file of one instance, as writeBinWitness lays it out (main.cpp:288-334)"""
        body = self.values(instance)
        return (b"wtns" + struct.pack("<II", 2, 2) + struct.pack("<IQ", 1, 8 + self.n8) + struct.pack("<I", self.n8)
                + self.prime.to_bytes(self.n8, "little") + struct.pack("<I", self.n_witness)
                + struct.pack("<IQ", 2, self.n8 * self.n_witness) + body)


def load(path) -> WitnessBatch:
    return WitnessBatch(path)

"""What does a vrow of the bit-plane evaluation kernel cost, and what does the circuit's operand pattern add?
    python tools/bits_shape_bench.py <dir> <name> [groups] [width]
Times cw_bits_eval_kernel (cw_bits_eval_bench: zero-filled table, results unused) on variants of the circuit's program:
  real      the program as lowered
  norows    the same records, no row loads / flushes
  aligned   every operand and result at the lane's OWN position of its row (no LDS bank conflicts), no row traffic
  idle      every lane idle (constants in, own ring entry out)
"""
import ctypes as C
import os
import struct
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (first: its HIP runtime must initialise before the library's)
from circom_amd import runtime as rt
if os.environ.get("CW_LIB"):
    from pathlib import Path
    rt.LIB_PATH = Path(os.environ["CW_LIB"]).resolve()

d, name = sys.argv[1], sys.argv[2]
groups = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
width = int(sys.argv[4]) if len(sys.argv) > 4 else 64
tape = open(os.path.join(d, name + ".cwt"), "rb").read()
c = rt.Circuit(os.path.join(d, name + ".cwt"), os.path.join(d, name + ".dat"), None)
info = c.bits_info()
n_vrows, n_slots, ring, cache = info["vrows"], info["slots_per_group"], info["ring"], info["cache"]
# the bit program is the tail of the tape: header (8 words), records, command blocks, signal map, assertion slots
n_cmd = n_vrows // 8 * 24
hdr_at = None
want = struct.pack("<5I", ring, n_vrows, n_slots & 0xFFFFFFFF, n_slots >> 32, cache)
hdr_at = tape.rfind(want)
assert hdr_at >= 0
n_asserts = struct.unpack_from("<I", tape, hdr_at + 20)[0]
recs = np.frombuffer(tape, dtype="<u4", count=n_vrows * 64 * 2, offset=hdr_at + 32).reshape(-1, 2).copy()
cmds = np.frombuffer(tape, dtype="<u4", count=n_cmd, offset=hdr_at + 32 + recs.size * 4).reshape(-1, 24).copy()
const_off = (ring + cache) * 512
lane8 = (np.arange(n_vrows * 64, dtype=np.uint32) % 64) * 8


def run(label, r, cm):
    ms = C.c_float()
    rt._chk(rt.lib().cw_bits_eval_bench(0, ring, cache, n_vrows, n_slots, r.ctypes.data_as(C.c_void_p), cm.ctypes.data_as(C.c_void_p),
                                        groups, width, 5, C.byref(ms)))
    print("SHAPE %-8s groups %5d width %2d  %8.3f ms  %7.1f ns per vrow" % (label, groups, width, ms.value, ms.value * 1e6 / n_vrows))


zero_cmds = np.zeros_like(cmds)
run("real", recs, cmds)
run("norows", recs, zero_cmds)
nf_only = cmds.copy(); nf_only[:, 0] &= 0xFF00        # flushes only
nl_only = cmds.copy(); nl_only[:, 0] &= 0x00FF        # loads only
run("flushes", recs, nf_only)
run("loads", recs, nl_only)
same = nf_only.copy()                                  # every flush onto ONE row of the table (no HBM write-back volume)
for j in range(6):
    same[:, 2 + 8 + 2 * j] = (n_slots // 64 - 1) * 512
run("flush1row", recs, same)
al = recs.copy()
w0, w1 = al[:, 0], al[:, 1]
def realign(off):
    rows = (off // 512) * 512
    return np.where(off >= const_off, off, rows + lane8).astype(np.uint32)
a = realign(w0 & 0xFFF8); b = realign(w0 >> 16); cc = realign(w1 & 0xFFFF); dd = realign(w1 >> 16)
al[:, 0] = a | (w0 & 3) | (b << 16)
al[:, 1] = cc | (dd << 16)
run("aligned", al, zero_cmds)
idle = np.zeros_like(recs)
idle[:, 0] = const_off | (const_off << 16)
vr = np.arange(n_vrows * 64, dtype=np.uint32) // 64
idle[:, 1] = const_off | ((((vr % ring) * 512) + lane8) << 16)
run("idle", idle, zero_cmds)

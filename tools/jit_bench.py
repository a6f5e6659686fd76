"""Time the emitted bit-plane code (hip_elements/bitjit.py) on a prebuilt SHA-256 tape (tools/prebuild_jit.py):
   python tools/jit_bench.py <message bits> <batch> [steps]
Packed inputs (cw_set_inputs_bits_device), so that the numbers are the evaluation's own; digests of sampled instances are
checked against hashlib, the fused R1CS flags must be clean, and one run of the stand-alone audit is timed for comparison."""
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from circom_amd import runtime as rt           # noqa: E402

nbits = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_in", "jit")
name = "sha256_%d" % nbits
fname = name + ("_" + os.environ["TAG"] if os.environ.get("TAG") else "")
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

res = {"workload": fname, "batch": B}
for eng in (os.environ.get("ENGINES", "jit,interp").split(",")):
    os.environ["CW_BITS_JIT"] = "1" if eng == "jit" else "0"
    c = rt.Circuit(os.path.join(d, fname + ".cwt"), os.path.join(d, name + ".dat"), os.path.join(d, name + ".r1cs"))   # variants share the tables
    b = c.batch(B)
    assert b.bitmode and b.jit == (eng == "jit")
    rng = np.random.default_rng(1)
    G = (B + 63) // 64
    masks = rng.integers(0, 1 << 63, size=(G, c.n_inputs), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, size=(G, c.n_inputs), dtype=np.uint64)
    p = C.c_void_p()
    assert hip.hipMalloc(C.byref(p), masks.nbytes) == 0
    assert hip.hipMemcpy(p, masks.ctypes.data, masks.nbytes, 1) == 0
    b.set_inputs_bits_device(p.value)
    b.run(); b.sync()
    t = []
    for _ in range(steps):
        t0 = time.perf_counter()
        b.run(); b.sync()
        t.append((time.perf_counter() - t0) * 1e3)
    tc = []
    for _ in range(max(1, steps // 2)):
        t0 = time.perf_counter()
        b.check_r1cs(); b.sync()
        tc.append((time.perf_counter() - t0) * 1e3)
    st = b.status()
    ok = bool((st == 0).all())
    # digests of a few instances
    bad = 0
    for i in (0, 1, 63, 64, 2047, 2048, B // 2 + 5, B - 1):
        if i >= B:
            continue
        bits = [(int(masks[i // 64, k]) >> (i % 64)) & 1 for k in range(nbits)]
        dg = hashlib.sha256(np.packbits(np.array(bits, dtype=np.uint8)).tobytes()).digest()
        want = [(dg[k // 8] >> (7 - k % 8)) & 1 for k in range(256)]
        got = [b.signal(i, 1 + k) for k in range(256)]
        bad += got != want
    audit_ms = None
    if not os.environ.get("NO_AUDIT"):
        os.environ["CW_R1CS_AUDIT"] = "1"
        t0 = time.perf_counter()
        b.check_r1cs(); b.sync()
        audit_ms = (time.perf_counter() - t0) * 1e3
        del os.environ["CW_R1CS_AUDIT"]
    ok2 = bool((b.status() == 0).all())
    res[eng] = {"run_ms": t, "run_ms_min": min(t), "check_ms": tc, "audit_ms": audit_ms, "status_clean": ok, "after_audit_clean": ok2,
                "digest_mismatches": bad, "witnesses_per_s": B / (min(t) + min(tc)) * 1e3, "table_GB": b.bits_slots * 8 * b.bits_groups / 1e9}
    print(eng, json.dumps(res[eng]), flush=True)
    b.close(); c.close()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/jit_bench_%s_%d.json" % (fname, B), "w"), indent=1)

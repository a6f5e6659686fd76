"""Emitted-code variants of one SHA-256 workload for timing experiments (gpurun_in/jit/<name>_<tag>.*):
   python tools/prebuild_jit_variants.py <message bits> tag:key=val,key=val ...   keys: prefetch, n_vgpr, n_agpr, fuse_check"""
import os, sys, time, json, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from circom_amd.frontend.dsl import Program
from circom_amd.frontend.flatten import flatten
from circom_amd.circuits.sha256 import Sha256
from circom_amd.hip_elements import bitjit, writers
from circom_amd.hip_elements.bitblast import bitblast
from circom_amd.hip_elements.bitmap import map_network
from circom_amd.hip_elements.bitsched import lower_bits
from circom_amd.hip_elements.lower import lower

nbits = int(sys.argv[1])
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_in", "jit")
os.makedirs(out, exist_ok=True)
t0 = time.time()
fc = flatten(Program(Sha256(nbits)))
net = bitblast(fc)
bt = lower_bits(map_network(net), fc)
tapes = [lower(fc, n_strands=1, mont=False)]
base = os.path.join(out, "sha256_%d" % nbits)
writers.write_dat(base + ".dat", fc)
writers.write_r1cs(base + ".r1cs", fc)
print("network ready %.0f s" % (time.time() - t0), flush=True)
for spec in sys.argv[2:]:
    tag, _, kv = spec.partition(":")
    kw = {}
    for item in filter(None, kv.split(",")):
        k, v = item.split("=")
        kw[k] = (v != "0") if k == "fuse_check" else int(v)
    t0 = time.time()
    asm_kw = {k: kw.pop(k) for k in list(kw) if k in ("nt", "nowait", "nostore", "noload", "noacc", "nopage")}
    jp = bitjit.lower_jit(net, fc, **kw)
    asm = bitjit.to_asm(jp)
    if asm_kw.get("nt") == 0:
        asm = asm.replace(" nt\n", "\n")
    # timing-only experiments (results are wrong): which instructions make a wave wait
    drop = tuple(t for k, t in (("nowait", "s_waitcnt vmcnt"), ("nostore", "buffer_store_dword"), ("noload", "buffer_load_dword"), ("noacc", "v_accvgpr_"), ("nopage", "s_mov_b32 s16,")) if asm_kw.get(k))
    if drop:
        asm = "".join(l for l in asm.splitlines(True) if not any(t in l for t in drop))
    jp.code = bitjit.assemble(asm)
    p = base + "_" + tag
    writers.write_tape(p + ".cwt", tapes, bt, jp)
    print(tag, "%.0f s" % (time.time() - t0), json.dumps(jp.stats), flush=True)

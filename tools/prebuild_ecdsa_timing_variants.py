"""Timing-only variants of the ECDSA verifier's emitted 16-strand code (results are garbage, only the clock is read): which part of a
launch is operand waits, barriers, stores, the bodies.  One lowering, four emissions (fpjit._EXP), artefacts under
gpurun_in/cache/ecdsa_verify_s16_b1_ma_r06x_<variant> (xz); run with CW_ARTEFACT_FP=r06x_<variant> python bench.py --workload ecdsa_verify
--no-parity --no-cpu-baseline --in-flight 1 --steps 3"""
import os, sys, time, json, shutil, subprocess, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from circom_amd import compiler
from circom_amd.frontend.flatten import flatten
from circom_amd.hip_elements import writers, fpjit
from circom_amd.hip_elements.lower import lower

t0 = time.time()
fc = flatten(bench.make_program("ecdsa_verify"))
tape = lower(fc, n_strands=16, mont=False)
print("lowered %.0f s: %d rows" % (time.time() - t0, len(tape.rows)), flush=True)
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_in", "cache")
base = None
for exp in sys.argv[1:] or ["nowait", "nobarrier", "nostore", "nocall"]:
    t0 = time.time()
    fpjit._EXP = "" if exp == "base" else exp
    d = os.path.join(root, "ecdsa_verify_s16_b1_ma_r06x_%s" % exp)
    os.makedirs(d, exist_ok=True)
    p = lambda ext: os.path.join(d, "ecdsa_verify" + ext)
    sp = tempfile.mkdtemp(prefix="cw_fpjit_")
    prog = fpjit.emit(tape, constraints=None, spool_path=os.path.join(sp, "k.s"))
    fpjit.assemble(prog)
    shutil.rmtree(sp, ignore_errors=True)
    rid = writers.write_r1cs(p(".r1cs"), fc)
    writers.write_tape(p(".cwt"), [tape], None, None, (prog,), r1cs_id=rid)
    writers.write_dat(p(".dat"), fc)
    json.dump({}, open(p(".jit.json"), "w"))
    json.dump([dict(prog.stats, n_strands=16, code_bytes=len(prog.code))], open(p(".fpjit.json"), "w"))
    for ext in (".cwt", ".r1cs", ".dat"):
        subprocess.run(["xz", "-T0", "-3", "-f", p(ext)], check=True)
    open(os.path.join(d, "done"), "w").write("r06x")
    print(exp, "%.0f s" % (time.time() - t0), len(prog.code), flush=True)

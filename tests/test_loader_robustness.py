"""cw_load must reject malformed artefacts with CW_EIO instead of crashing (the reference mmap()s its .dat blindly,
main.cpp:38-56; a library serving many circuits cannot)."""
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

from circom_amd import runtime as rt
from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program
from circom_amd.circuits.basic import BasicMain


def test_truncated_and_corrupt_files_are_rejected(tmp_path):
    cp = compile_program(Program(BasicMain()), str(tmp_path), "basic", strands=(1, 4))
    tape = open(cp.tape_path, "rb").read()
    dat = open(cp.dat_path, "rb").read()
    r1cs = open(cp.r1cs_path, "rb").read()
    rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path).close()           # sanity: the originals load

    def attempt(t=None, d=None, r=None):
        tp, dp, rp = tmp_path / "x.cwt", tmp_path / "x.dat", tmp_path / "x.r1cs"
        tp.write_bytes(tape if t is None else t)
        dp.write_bytes(dat if d is None else d)
        rp.write_bytes(r1cs if r is None else r)
        return rt.Circuit(tp, dp, rp)

    for cut in (0, 3, 16, 60, 100, len(tape) // 2, len(tape) - 1):
        with pytest.raises(rt.CwError):
            attempt(t=tape[:cut])
    with pytest.raises(rt.CwError):
        attempt(t=b"XXXX" + tape[4:])                                       # bad magic
    with pytest.raises(rt.CwError):
        attempt(t=tape[:4] + (99).to_bytes(4, "little") + tape[8:])         # unknown version
    # (the runtime reads the hash map and the witness list of the .dat; the constant table after them is the
    #  reference runtime's business, so only truncation inside the first two parts is an error here)
    for cut in (0, 10, 256 * 24 + 5):
        with pytest.raises(rt.CwError):
            attempt(d=dat[:cut])
    for cut in (0, 11, 40, len(r1cs) // 2, len(r1cs) - 1):
        with pytest.raises(rt.CwError):
            attempt(r=r1cs[:cut])
    with pytest.raises(rt.CwError):
        attempt(r=b"r1cx" + r1cs[4:])
    # an .r1cs of another prime is refused
    other = compile_program(Program(BasicMain(), prime="bls12381"), str(tmp_path / "o"), "basic", strands=(1,))
    with pytest.raises(rt.CwError) as e:
        rt.Circuit(cp.tape_path, cp.dat_path, other.r1cs_path)
    assert "prime" in str(e.value)


def test_small_primes_are_rejected_not_miscomputed(tmp_path):
    """The device field code assumes circom's 253..256-bit primes (the short-path sums rely on q > 2^192); Goldilocks
    (64-bit, a separate runtime in the reference) must be refused by the lowering and by the library."""
    import numpy as np
    from circom_amd.frontend.dsl import Program
    from circom_amd.frontend.flatten import flatten
    from circom_amd.hip_elements.lower import lower
    from circom_amd.circuits.basic import Multiplier2
    with pytest.raises(ValueError, match="Goldilocks"):
        lower(flatten(Program(Multiplier2(), prime="goldilocks")))
    gold = 0xFFFFFFFF00000001
    a = np.zeros((4, 32), dtype=np.uint8)
    with pytest.raises(rt.CwError, match="unsupported prime"):
        rt.fp_mul_bench(gold, a, a, 4, device=0)


def test_mutated_files_never_crash_the_loader(tmp_path):
    """Random byte flips, truncations and wild 32-bit values in .cwt / .dat / .r1cs: every index the files carry is
    validated at load time (schedule operands, destinations, term and extra tables, header shape, input hash map),
    so a damaged file is either rejected with CW_EIO or loads into something whose indices are all in range.  Run in
    a child process: a crash of the library would take the interpreter with it."""
    import subprocess
    import sys
    code = r'''
import sys, os, random
sys.path.insert(0, %r)
from circom_amd import runtime as rt
from circom_amd.compiler import compile_program
from circom_amd.frontend.dsl import Program
from circom_amd.circuits.poseidon import Poseidon
from circom_amd.circuits.basic import Num2Bits
d = %r
os.environ.pop("CW_FPJIT", None)          # (the .cwt under test carries the emitted-code section as well)
rng = random.Random(5)
n_ok = n_bad = 0
for name, prog, js in (("n2b", Num2Bits(8), '{"in": "5"}'), ("pos", Poseidon(2), '{"inputs": ["5", "6"]}')):
    cp = compile_program(Program(prog), d, name, sym=False, fpjit=True)
    assert cp.fpjit and b"FPJT" in open(cp.tape_path, "rb").read()
    files = {k: open(p, "rb").read() for k, p in (("cwt", cp.tape_path), ("dat", cp.dat_path), ("r1cs", cp.r1cs_path))}
    for it in range(700):
        which = rng.choice(["cwt", "cwt", "dat", "r1cs"])
        data = bytearray(files[which])
        r = rng.random()
        if r < 0.6:
            for _ in range(rng.randrange(1, 8)):
                data[rng.randrange(len(data))] = rng.randrange(256)
        elif r < 0.8:
            data = data[:rng.randrange(len(data))]
        else:
            pos = rng.randrange(len(data) - 4)
            data[pos:pos + 4] = rng.choice([0xFFFFFFFF, 0x7FFFFFFF, 0x80000000, 1 << 20]).to_bytes(4, "little")
        paths = {k: os.path.join(d, "f_" + k) for k in files}
        for k in files:
            open(paths[k], "wb").write(bytes(data) if k == which else files[k])
        try:
            c = rt.Circuit(paths["cwt"], paths["dat"], paths["r1cs"])
        except rt.CwError:
            n_bad += 1
            continue
        b = c.batch(4, device=-1)
        for call in (lambda: b.set_inputs_json(0, js), lambda: c.r1cs_plan_stats(100, 0, 0)):
            try:
                call()
            except rt.CwError:
                pass
        b.close(); c.close(); n_ok += 1
print("ok", n_ok, n_bad)
''' % (str(ROOT), str(tmp_path))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, "the loader crashed on a mutated file (exit %d)\n%s" % (r.returncode, r.stderr[-500:])
    out = r.stdout.split()
    assert out[0] == "ok" and int(out[2]) > 200          # most mutations are rejected, none kills the process


def test_advisor_round1_loader_cases(tmp_path):
    """Regression cases of the round-1 review: (1) a .dat hash-map entry whose signalid + size wraps in 64 bits,
    (2) a tape whose hash map is smaller than its number of input names / carries a name twice, (3) a D_LINSUM row
    whose unused operand b holds a wild index, (4) an .r1cs whose last section length is close to 2^64, a short header
    section, a constraint count the section cannot hold, a duplicated section."""
    import struct
    cp = compile_program(Program(BasicMain()), str(tmp_path), "basic", strands=(1,))
    tape = open(cp.tape_path, "rb").read()
    dat = bytearray(open(cp.dat_path, "rb").read())
    r1cs = open(cp.r1cs_path, "rb").read()

    def attempt(t=None, d=None, r=None):
        tp, dp, rp = tmp_path / "y.cwt", tmp_path / "y.dat", tmp_path / "y.r1cs"
        tp.write_bytes(tape if t is None else t)
        dp.write_bytes(bytes(dat) if d is None else d)
        rp.write_bytes(r1cs if r is None else r)
        return rt.Circuit(tp, dp, rp)

    # (1) first non-empty hash entry: signalid = 2^64 - 1, size = 2  (the sum wraps to 1)
    for i in range(256):
        h, sid, sz = struct.unpack_from("<QQQ", dat, i * 24)
        if sid:
            bad = bytearray(dat)
            struct.pack_into("<QQQ", bad, i * 24, h, (1 << 64) - 1, 2)
            with pytest.raises(rt.CwError):
                attempt(d=bytes(bad))
            bad = bytearray(dat)
            struct.pack_into("<QQQ", bad, i * 24, h, sid, (1 << 64) - 1)
            with pytest.raises(rt.CwError):
                attempt(d=bytes(bad))
            break
    else:
        raise AssertionError("no input in the hash map")
    # (2) header word 5 = n_input_names, word 6 = hashmap size (after magic/version/n64/n_variants + prime)
    hdr = 16 + 32
    words = list(struct.unpack_from("<12I", tape, hdr))
    w2 = list(words); w2[6] = 512                       # 512 slots for 2 names: not max(2^ceil(log2 n), 256)
    with pytest.raises(rt.CwError):
        attempt(t=tape[:hdr] + struct.pack("<12I", *w2) + tape[hdr + 48:])
    # (4) r1cs: walk the sections
    off, secs = 12, []
    nsec = struct.unpack_from("<I", r1cs, 8)[0]
    for _ in range(nsec):
        typ, ln = struct.unpack_from("<IQ", r1cs, off)
        secs.append((typ, off, ln))
        off += 12 + ln
    typ, o, ln = secs[-1]
    with pytest.raises(rt.CwError):
        attempt(r=r1cs[:o + 4] + struct.pack("<Q", (1 << 64) - 8) + r1cs[o + 12:])
    hd = [s for s in secs if s[0] == 1][0]
    short = r1cs[:hd[1] + 4] + struct.pack("<Q", 40) + r1cs[hd[1] + 12:hd[1] + 12 + 40] + r1cs[hd[1] + 12 + hd[2]:]
    with pytest.raises(rt.CwError):
        attempt(r=short)
    ncons_at = hd[1] + 12 + 36 + 16 + 8
    with pytest.raises(rt.CwError):
        attempt(r=r1cs[:ncons_at] + struct.pack("<I", 0xFFFFFFF0) + r1cs[ncons_at + 4:])
    dup = r1cs[:8] + struct.pack("<I", nsec + 1) + r1cs[12:] + r1cs[hd[1]:hd[1] + 12 + hd[2]]
    with pytest.raises(rt.CwError):
        attempt(r=dup)
    attempt().close()                                                        # the unmodified files still load


def test_emitted_checks_are_trusted_only_for_the_r1cs_they_were_built_from(tmp_path, monkeypatch):
    """the fused checks / covered rows are baked into the emitted code at lowering time; the only guard that the .r1cs
    given to cw_load is THAT system used to be its row count.  The tape now records CRC-32 + length of the file's constraint
    section: another file with the same rows count switches the baked-in checks off (the stand-alone kernels check every row)"""
    import struct
    from circom_amd import runtime as rt
    from circom_amd.compiler import compile_program
    from circom_amd.frontend.dsl import Program
    from circom_amd.circuits.poseidon import Poseidon
    from test_bitplane import BitGadget
    monkeypatch.setenv("CW_FPJIT", "1")
    for name, prog, kw in (("p2", Program(Poseidon(2)), dict(strands=(4,))), ("bg", Program(BitGadget(16)), dict(strands=(1,), bits=True, jit=True))):
        cp = compile_program(prog, str(tmp_path), name, sym=False, **kw)
        c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
        assert c.emitted_checks_match
        c.close()
        raw = bytearray(open(cp.r1cs_path, "rb").read())
        # constraint section = the first section of the file (type 2): flip the low byte of the first coefficient
        assert struct.unpack_from("<I", raw, 12)[0] == 2
        pos = 24                                                   # file header 12 + section header 12: nnz of the first block
        while struct.unpack_from("<I", raw, pos)[0] == 0:          # (empty linear combinations have no terms to change)
            pos += 4
        raw[pos + 4 + 4] ^= 2                                      # nnz, wire id, then the coefficient's low byte
        other = tmp_path / (name + "_other.r1cs")
        other.write_bytes(bytes(raw))
        c = rt.Circuit(cp.tape_path, cp.dat_path, other)
        assert not c.emitted_checks_match and c.n_constraints == len(cp.flat.constraints)
        c.close()

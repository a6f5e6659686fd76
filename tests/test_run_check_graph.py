"""cw_run_check: cw_run + cw_check_r1cs captured once as a HIP graph on the batch's stream and replayed (include/circom_amd.h).
The replay must be a full step - table init, ingest of whatever the input buffer holds NOW, evaluation, check - on both engines:
every round rewrites the device buffer, and statuses / first violated rows / witnesses are compared with a plain batch's and
with the oracle's."""
import numpy as np
import pytest

from circom_amd.frontend.dsl import Program
from oracle.tape_eval import check_r1cs

pytestmark = pytest.mark.gpu


def _image(q, rows):
    return np.frombuffer(b"".join(int(v % q).to_bytes(32, "little") for r in rows for v in r), dtype=np.uint8).reshape(len(rows), len(rows[0]), 32).copy()


def _rounds(c, cp, hip, B, make_rows, n_rounds=4, expect_bad=False):
    from circom_amd import runtime as rt
    imgs = [_image(c.q, make_rows(r)) for r in range(n_rounds)]
    d_in = hip.upload(imgs[0])
    g, p = c.batch(B), c.batch(B)
    g.set_inputs_device(d_in)
    p.set_inputs_device(d_in)
    seen_bad = 0
    for r in range(n_rounds):
        assert hip.h.hipMemcpy(d_in, imgs[r].ctypes.data, imgs[r].nbytes, 1) == 0
        g.run_check(); g.sync()
        assert g.graph_captured == (r >= 1), r                      # the first call is plain, the second captures, then replays
        p.run(); p.check_r1cs(); p.sync()
        st, fb = g.status(), g.r1cs_first_bad()
        if not ((st == p.status()).all() and (fb == p.r1cs_first_bad()).all()):
            same_w = [g.witness(i) == p.witness(i) for i in (0, 1, B - 1)]
            raise AssertionError("round %d: status %s / %s, first bad %s / %s, witnesses equal %s, emitted %s fused %s lanes %s" % (
                r, st[:4].tolist(), p.status()[:4].tolist(), fb[:4].tolist(), p.r1cs_first_bad()[:4].tolist(), same_w,
                getattr(g, "emitted", None), getattr(g, "fused_check", None), g.lanes))
        for i in sorted({0, 1, B // 2, B - 1, (7 * r + 3) % B}):
            w = g.witness(i)
            assert w == p.witness(i), (r, i)
            want = check_r1cs(c.q, cp.flat.constraints, w)
            assert bool(st[i] & rt.ST_R1CS_FAILED) == (want is not None), (r, i)
            if want is not None:
                assert fb[i] == want, (r, i)
                seen_bad += 1
    if expect_bad:
        assert seen_bad > 0
    # timing marks are events on the stream: not captured - the call falls back to the two plain calls, and captures again after
    g.set_timing(True)
    g.run_check(); g.sync()
    assert not g.graph_captured and g.kernel_ms()["check"] > 0
    g.set_timing(False)
    g.run_check(); g.run_check(); g.sync()
    assert g.graph_captured and (g.status() == p.status()).all()
    # inputs set on the host: the copy belongs to no graph
    rows = make_rows(0)
    g.set_inputs(rows)
    g.run_check(); g.sync()
    assert not g.graph_captured
    p.set_inputs(rows); p.run(); p.check_r1cs(); p.sync()
    assert (g.status() == p.status()).all() and g.witness(B - 1) == p.witness(B - 1)
    g.close(); p.close()


@pytest.mark.parametrize("mont", [False, True])
def test_graph_replay_on_the_256_bit_engine(tmp_path, mont, monkeypatch):
    from test_bitplane import _Hip
    from test_gpu_parity import FlakyChain, _compile
    monkeypatch.setenv("CW_MONT", "1" if mont else "0")
    cp, c = _compile(tmp_path, Program(FlakyChain(9)), "flaky%d" % mont)
    B = 333
    _rounds(c, cp, _Hip(), B, lambda r: [[(i * 2654435761 + r * 97) % (1 << 61), (i + r) % 20 if (i + r) % 3 else 9 + i] for i in range(B)], expect_bad=True)
    c.close()


def test_graph_replay_with_the_fused_check(tmp_path, monkeypatch):
    """the emitted program with the R1CS check fused in (CW_FP_FUSED=1): every round breaks a different row of every instance, the
    replayed step must name this round's rows (finding words, first-bad words and status words are all reset or rewritten inside
    the step - by kernels: a memset node of the captured graph was seen to write garbage on the second replay)"""
    from test_bitplane import _Hip
    from test_gpu_parity import FlakyChain, _compile
    monkeypatch.setenv("CW_MONT", "1")
    monkeypatch.setenv("CW_FP_FUSED", "1")
    cp, c = _compile(tmp_path, Program(FlakyChain(9)), "flakyfused")
    B = 333
    probe = c.batch(B)
    fused = probe.fused_check
    probe.close()
    assert fused, "no fused-check program for this circuit: the test would not test what it says"
    _rounds(c, cp, _Hip(), B, lambda r: [[(i * 2654435761 + r * 97) % (1 << 61), (i + r) % 20 if (i + r) % 3 else 9 + i] for i in range(B)],
            n_rounds=5, expect_bad=True)
    c.close()


def test_graph_replay_of_emitted_256_bit_code(tmp_path):
    from test_bitplane import _Hip
    from test_gpu_parity import _compile
    from circom_amd.circuits.poseidon import Poseidon
    cp, c = _compile(tmp_path, Program(Poseidon(2)), "pos2g")
    B = 700
    rng = np.random.default_rng(3)
    _rounds(c, cp, _Hip(), B, lambda r: [[int.from_bytes(rng.bytes(32), "little") % c.q, r + i] for i in range(B)], n_rounds=3)
    c.close()


def test_graph_replay_on_the_bit_plane_engine(tmp_path, monkeypatch):
    from test_bitplane import _Hip, _gpu, BitGadget
    for eng in ("0", "1"):
        monkeypatch.setenv("CW_BITS_JIT", eng)
        cp, c = _gpu(tmp_path, Program(BitGadget(16)), "bg16g" + eng)
        fc = cp.flat
        B = 300
        rng = np.random.default_rng(11)
        # round 2 has instances with an input that is not a bit: the side batch on the 256-bit schedule serves them after the replay
        def rows(r):
            m = rng.integers(0, 2, size=(B, fc.n_main_inputs)).tolist()
            if r == 2:
                m[5][0] = 7
                m[B - 1][1] = c.q - 1
            return m
        _rounds(c, cp, _Hip(), B, rows)
        c.close()


def test_graph_replay_on_the_64_bit_engine(tmp_path):
    """--prime goldilocks (csrc/cw64.hip): the same replay through the 64-bit runtime's kernels, with a witness hint that is wrong
    for odd inputs so that the check's verdict changes from round to round"""
    from test_bitplane import _Hip
    from circom_amd import runtime as rt
    from circom_amd.compiler import compile_program
    from circom_amd.frontend.dsl import template

    @template
    def BadSquare(cx):
        a = cx.input("a")
        out = cx.output("out")
        cx.hint(out, a * a + (a & 1))                             # wrong for odd a
        cx.enforce(out, a * a, runtime_check=False)
    cp = compile_program(Program(BadSquare(), prime="goldilocks"), str(tmp_path), "badsq_g", sym=False)
    c = rt.Circuit(cp.tape_path, cp.dat_path, cp.r1cs_path)
    B = 200
    _rounds(c, cp, _Hip(), B, lambda r: [[(i * 0x9E3779B97F4A7C15 + r) % c.q] for i in range(B)], expect_bad=True)
    c.close()

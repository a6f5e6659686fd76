"""How many signals of a circuit are provably small?  (Feasibility data for typed narrow value slots, NOTES.md.)

Abstract interpretation of the flat witness code, assuming the main inputs are bits (`--input-bits 1`) or below
2^k: every value gets a signed integer interval; a signal is
  * `bit`    if its interval is within [0, 1],
  * `narrow` if within [0, 2^32),
  * `wide`   otherwise (or unknown).
Plain interval arithmetic loses the correlations of boolean polynomials (a + b - 2ab has interval [-2, 2]), so an
expression whose leaves are at most 8 bit-typed signals is evaluated exactly over all assignments instead.

    python tools/range_report.py sha256_512 | poseidon2 | semaphore20 [--input-bits K]
"""
import itertools
import os
import sys
sys.setrecursionlimit(10000)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from circom_amd import opcodes as O                       # noqa: E402
from circom_amd.frontend.dsl import Program               # noqa: E402
from circom_amd.frontend.flatten import FlatCircuit       # noqa: E402

K_SIG, K_TMP, K_CONST, K_NONE = O.K_SIG, O.K_TMP, O.K_CONST, O.K_NONE
WIDE = None


def analyse(fc, input_bits=1):
    q = fc.fp.q
    half = q >> 1
    code = fc.code
    op, dk, dv = code["op"], code["dk"], code["dv"]
    ak, av, bk, bv, ck, cv = code["ak"], code["av"], code["bk"], code["bv"], code["ck"], code["cv"]
    consts = [c - q if c > half else c for c in fc.constants]       # signed representatives
    sig = {0: (1, 1)}                                               # signal -> interval or WIDE
    for k in range(fc.n_main_inputs):
        sig[fc.main_input_start + k] = (0, (1 << input_bits) - 1)
    tmp_def = {}                                                    # temp -> row index (SSA within the flat code)

    def leaf_interval(k, v):
        if k == K_CONST:
            return (consts[v], consts[v])
        if k == K_SIG:
            return sig.get(v, WIDE)
        return None

    tmp_iv = {}                                                     # temps are single-assignment: memoise

    def interval(k, v, depth=0):
        if k != K_TMP:
            return leaf_interval(k, v)
        if v in tmp_iv:
            return tmp_iv[v]
        i = tmp_def.get(v)
        if i is None or depth > 900:
            return WIDE
        r = row_interval(i, depth + 1)
        tmp_iv[v] = r
        return r

    def row_interval(i, depth=0):
        o = op[i]
        a = interval(ak[i], av[i], depth)
        if o == O.COPY:
            return a
        if o in (O.LT, O.GT, O.LEQ, O.GEQ, O.EQ, O.NEQ, O.LAND, O.LOR, O.LNOT):
            return (0, 1)
        b = interval(bk[i], bv[i], depth) if bk[i] != K_NONE else None
        if o == O.NEG:
            return WIDE if a is WIDE else (-a[1], -a[0])
        if o == O.BAND:
            cands = [x[1] for x in (a, b) if x is not WIDE and x is not None and x[0] >= 0]
            return (0, min(cands)) if cands else WIDE
        if o == O.SHR:
            if a is not WIDE and a[0] >= 0 and b is not WIDE and b is not None and b[0] == b[1] and 0 <= b[0] < 254:
                return (a[0] >> b[0], a[1] >> b[0])
            return WIDE
        if o == O.SELECT:
            c = interval(ck[i], cv[i], depth)
            if b is WIDE or c is WIDE or b is None or c is None:
                return WIDE
            return (min(b[0], c[0]), max(b[1], c[1]))
        if a is WIDE or b is WIDE or b is None:
            return WIDE
        if o == O.ADD:
            return (a[0] + b[0], a[1] + b[1])
        if o == O.SUB:
            return (a[0] - b[1], a[1] - b[0])
        if o == O.MUL:
            ps = [a[0] * b[0], a[0] * b[1], a[1] * b[0], a[1] * b[1]]
            r = (min(ps), max(ps))
            return r if max(abs(r[0]), abs(r[1])) < (1 << 200) else WIDE
        if o in (O.BOR, O.BXOR):
            if a[0] >= 0 and b[0] >= 0:
                return (0, (1 << max(a[1].bit_length(), b[1].bit_length())) - 1)
            return WIDE
        return WIDE

    def exact(i):
        """exact value set of row i over all assignments of its (bit-typed) leaf signals, or None"""
        leaves = []
        seen = set()
        stack = [i]
        rows = []
        while stack:
            r = stack.pop()
            if r in seen:
                continue
            seen.add(r)
            rows.append(r)
            for k, v in ((ak[r], av[r]), (bk[r], bv[r]), (ck[r], cv[r])):
                if k == K_TMP:
                    d = tmp_def.get(v)
                    if d is None:
                        return None
                    stack.append(d)
                elif k == K_SIG:
                    if sig.get(v, WIDE) is WIDE or sig[v][0] < 0 or sig[v][1] > 1:
                        return None
                    if v not in leaves:
                        leaves.append(v)
            if len(leaves) > 8 or len(rows) > 60:
                return None
        rows.sort()
        lo, hi = None, None
        for bits in itertools.product((0, 1), repeat=len(leaves)):
            env = dict(zip(leaves, bits))
            tv = {}

            def val(k, v):
                if k == K_CONST:
                    return consts[v]
                if k == K_SIG:
                    return env[v] if v in env else sig[v][0]
                return tv[v]
            for r in rows:
                o = op[r]
                a = val(ak[r], av[r])
                b = val(bk[r], bv[r]) if bk[r] != K_NONE else 0
                if o == O.COPY: x = a
                elif o == O.ADD: x = a + b
                elif o == O.SUB: x = a - b
                elif o == O.MUL: x = a * b
                elif o == O.NEG: x = -a
                elif o == O.BAND and a >= 0 and b >= 0: x = a & b
                elif o == O.SHR and a >= 0 and 0 <= b < 254: x = a >> b
                else: return None
                if dk[r] == K_TMP:
                    tv[dv[r]] = x
                res = x
            lo = res if lo is None else min(lo, res)
            hi = res if hi is None else max(hi, res)
        return (lo, hi)

    for i in range(len(op)):
        if op[i] in O.NO_DST:
            continue
        if dk[i] == K_TMP:
            tmp_def[dv[i]] = i
        elif dk[i] == K_SIG:
            iv = row_interval(i)
            if iv is WIDE or iv[0] < 0 or iv[1] >= (1 << 32):
                ex = exact(i)
                if ex is not None:
                    iv = ex
            sig[dv[i]] = iv
    counts = {"bit": 0, "narrow": 0, "wide": 0}
    for s in range(fc.n_signals):
        iv = sig.get(s, WIDE)
        if iv is not WIDE and iv[0] >= 0 and iv[1] <= 1:
            counts["bit"] += 1
        elif iv is not WIDE and iv[0] >= 0 and iv[1] < (1 << 32):
            counts["narrow"] += 1
        else:
            counts["wide"] += 1
    return counts


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "sha256_512"
    bits = int(sys.argv[sys.argv.index("--input-bits") + 1]) if "--input-bits" in sys.argv else 1
    if name.startswith("sha256_"):
        from circom_amd.circuits.sha256 import Sha256
        prog = Sha256(int(name.split("_")[1]))
    elif name == "poseidon2":
        from circom_amd.circuits.poseidon import Poseidon
        prog = Poseidon(2)
    else:
        from circom_amd.circuits.eddsa import SemaphoreStyle
        prog = SemaphoreStyle(20)
    fc = FlatCircuit(Program(prog))
    c = analyse(fc, bits)
    n = fc.n_signals
    print("%s: %d signals, inputs assumed < 2^%d: bit %d (%.1f%%), other < 2^32 %d (%.1f%%), wide/unknown %d (%.1f%%)"
          % (name, n, bits, c["bit"], 100.0 * c["bit"] / n, c["narrow"], 100.0 * c["narrow"] / n, c["wide"], 100.0 * c["wide"] / n))
    wide_bytes = 32 * c["wide"] + 4 * (c["bit"] + c["narrow"])
    print("value table per instance: %d B all-wide -> %d B with 4-byte narrow slots (%.1fx smaller)" % (32 * n, wide_bytes, 32.0 * n / wide_bytes))


if __name__ == "__main__":
    main()

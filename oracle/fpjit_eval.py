"""CPU replay of the IR of the emitted 256-bit code (circom_amd/hip_elements/fpjit.py).

TEST INFRASTRUCTURE ONLY.  The emitted kernel keeps the interpreter's execution model (oracle/tape_eval.py replays and
race-checks the schedule itself); what the emission adds is exactly what goes wrong silently on a GPU: a wait count that is
one too large, a body that clobbers a register another step still needs, a prefetch that lands in a register a body is
using, an accumulator that was assumed zero.  This replay executes the IR of every strand for ONE instance with

  * register groups (operand sets A/B of both parities, D, G, the accumulators) that are POISONED when a body the
    emitter calls may write them, and PENDING while a load is in flight: using either raises `JitHazard`,
  * the in-order vector-memory and LDS queues: a `wait` makes visible exactly the loads its counts cover,
  * the bodies' arithmetic restated on Python integers (oracle/field.py: the reference's Fr_* semantics, generic/fr.cpp),

and returns the signal table + status word, to be compared with `tape_eval.eval_tape` of the same schedule.
"""
from __future__ import annotations

from .field import Field, FieldError

A_E, B_E, D_REG, A_O, B_O, G_REG, ACC_REG = 0, 8, 16, 24, 32, 76, 84
_GROUPS = (A_E, B_E, D_REG, A_O, B_O, G_REG)


class JitHazard(Exception):
    pass


class _Pending:
    def __init__(self, seq, value):
        self.seq, self.value = seq, value


POISON = object()


def replay(prog_ir, body_info, q: int, n_signals: int, n_tslots: int, n_lds: int, inputs: dict, rbits: int = 261, one: int = 1,
           functions=(), consts=()):
    """prog_ir: per strand list of IR tuples; body_info: name -> (set of VGPRs the body may touch, parity)."""
    f = Field(q)
    rinv = pow(1 << rbits, -1, q)
    mem = [0] * (n_signals + max(n_tslots, 1))
    mem[0] = one
    for k, v in inputs.items():
        mem[k] = v % q
    lds = [0] * max(n_lds, 1)
    status = [0]
    fbmin = [None]                       # smallest index of a constraint the fused check found violated
    ns = len(prog_ir)
    pcs = [0] * ns
    state = []
    for s in range(ns):
        state.append({"reg": {g: POISON for g in _GROUPS}, "vm_done": 0, "lg_done": 0, "acc": None, "acc_zero": 0, "sarg": None,
                      "coef": None, "sel": False, "lin": None})
        state[s]["reg"][D_REG] = 0

    def rd(st, g, what):
        v = st["reg"][g]
        if v is POISON:
            raise JitHazard("%s reads register group v%d, which holds nothing defined" % (what, g))
        if isinstance(v, _Pending):
            raise JitHazard("%s reads register group v%d while its load is still in flight (wait count too large)" % (what, g))
        return v

    def fail(bits, seq):
        word = bits | (seq << 8)
        if status[0] == 0 or (word >> 8) < (status[0] >> 8):
            status[0] = word

    def run(s):
        """strand s up to and over its next barrier; returns the barrier kind or None at the end"""
        st = state[s]
        reg = st["reg"]
        ir = prog_ir[s]
        while pcs[s] < len(ir):
            ins = ir[pcs[s]]
            pcs[s] += 1
            k = ins[0]
            if k == "ld":
                if isinstance(reg[ins[1]], _Pending):
                    raise JitHazard("two loads in flight to register group v%d" % ins[1])
                reg[ins[1]] = _Pending(("vm", ins[3]), mem[ins[2]])
            elif k == "ldl":
                reg[ins[1]] = _Pending(("lg", ins[3]), lds[ins[2]])
            elif k == "wait":
                vm, lg = ins[1], ins[2]
                for g, v in reg.items():
                    if isinstance(v, _Pending):
                        kind, seq = v.seq
                        if (kind == "vm" and vm is not None and seq <= vm) or (kind == "lg" and lg is not None and seq <= lg):
                            reg[g] = v.value
            elif k == "st":
                mem[ins[1]] = rd(st, D_REG, "a store")
            elif k == "stl":
                lds[ins[1]] = rd(st, D_REG, "an LDS store")
            elif k == "mov":
                reg[ins[1]] = rd(st, ins[2], "a register move")
            elif k == "lit":
                reg[ins[1]] = ins[2]
            elif k == "zacc":
                if st["acc_zero"] < ins[2]:
                    raise JitHazard("the emitter skipped clearing accumulator registers %d..%d, which are not known to be zero"
                                    % (st["acc_zero"], ins[2]))
                st["acc_zero"] = max(st["acc_zero"], ins[1])
                st["acc"] = 0
                st["lin"] = 0
            elif k == "sarg":
                st["sarg"] = ins[1]
            elif k == "coef":
                st["coef"] = sum(w << (29 * j) for j, w in enumerate(ins[1]))
            elif k == "bit":
                a = rd(st, ins[1], "bit extraction")
                reg[D_REG] = (a >> ins[2]) & 1 if ins[2] < 256 else 0
            elif k == "bits":
                # D_BITS as one step: consecutive bits of the operand, one store per entry (an entry flagged `next` moves on first)
                a = rd(st, ins[1], "bit-field extraction")
                kbit = ins[2]
                for slot, nxt, _lo in ins[3]:
                    kbit += 1 if nxt else 0
                    mem[slot] = (a >> kbit) & 1 if kbit < 256 else 0
                reg[D_REG] = POISON                   # (the row has no value: D holds the last bit's words)
            elif k == "bar":
                for g, v in reg.items():
                    if isinstance(v, _Pending) and v.seq[0] == "lg":
                        reg[g] = v.value
                return "full" if ins[1] else "light"
            elif k == "heavy_done":
                pass
            elif k == "callfn":
                # D_CALL: the interpreter body reads and writes the call's register window in the value table (tape_eval's
                # restatement of the bytecode / of the native closed forms)
                from . import tape_eval as TE
                fn, slot0, seq = ins[1], ins[2], ins[3]
                n_regs, fcode, native = functions[fn]
                if native is not None and TE.USE_NATIVE:
                    from circom_amd.circuits.bigint_func import native_eval
                    kind, n_, k_, modulus = native
                    from circom_amd.circuits.bigint_func import native_n_args
                    kname = {1: "mod_inv", 2: "ec_add", 3: "ec_double", 4: "long_div"}[kind]
                    n_args = native_n_args(kname, k_, modulus)
                    try:
                        for j2, v2 in enumerate(native_eval(kname, n_, k_, modulus, [mem[slot0 + x] for x in range(n_args)])):
                            mem[slot0 + n_args + j2] = v2
                    except ZeroDivisionError:     # long_div on a divisor outside its contract: the device flags the lane
                        fail(2, seq)
                elif not TE.run_dev_function(f, fcode, mem, slot0, consts):
                    fail(2, seq)
            elif k == "call":
                name = ins[1]
                touched, parity = body_info[name]
                base = name.rsplit("_", 1)[0] if parity in ("e", "o", "h", "c", "k") else name
                ra, rb = (A_O, B_O) if parity == "o" else (A_E, B_E)
                what = "body " + name
                # a load in flight to a register the body may write would land in the middle of its arithmetic
                for g, v in reg.items():
                    if isinstance(v, _Pending) and any(r in touched for r in range(g, g + 8)) and g not in (ra, rb):
                        raise JitHazard("%s may write register group v%d while a load to it is in flight" % (what, g))
                res = None
                if base in ("add", "sub", "shl", "shr", "band", "bor", "bxor", "lt", "gt", "leq", "geq", "eq", "neq", "land", "lor", "pow"):
                    res = getattr(f, base)(rd(st, ra, what), rd(st, rb, what))
                elif base in ("idiv", "mod"):
                    a, b = rd(st, ra, what), rd(st, rb, what)
                    try:
                        res = getattr(f, base)(a, b)
                    except FieldError:
                        fail(2, st["sarg"])
                        res = 0
                elif base in ("neg", "bnot", "lnot", "inv"):
                    res = getattr(f, base)(rd(st, ra, what))
                elif base == "mmul":
                    res = rd(st, ra, what) * rd(st, rb, what) * rinv % q
                elif base == "mul2":
                    res = rd(st, ra, what) * rd(st, rb, what) % q
                elif base == "madd":
                    res = (rd(st, ra, what) * rd(st, rb, what) * rinv + rd(st, D_REG, what)) % q
                elif base.startswith("mulc"):
                    a, b = rd(st, ra, what), rd(st, rb, what)
                    res = a * b * rinv % q
                    if base[4] in "pn":
                        mag = st["sarg"]
                        c_plain = mag if base[4] == "p" else (q - mag) % q
                        if a * c_plain % q != res:
                            raise JitHazard("%s: the small constant and the scaled constant disagree" % what)
                    if base.endswith("a"):
                        res = (res + rd(st, D_REG, what)) % q
                elif base == "select":
                    st["sel"] = rd(st, ra, what) != 0
                elif base == "ext":
                    a, b = rd(st, ra, what), rd(st, rb, what)
                    res = a if st["sel"] else b
                elif base == "asserteq":
                    if rd(st, ra, what) != rd(st, rb, what):
                        fail(1, st["sarg"])
                elif base == "assertnz":
                    if rd(st, ra, what) == 0:
                        fail(1, st["sarg"])
                elif base in ("linp", "linn"):
                    if st["acc_zero"] < 12 and st["lin"] is None:
                        raise JitHazard("%s accumulates into registers that were never cleared" % what)
                    x = rd(st, ra, what)
                    rd(st, G_REG, what)
                    st["lin"] += (-st["sarg"] if base == "linn" else st["sarg"]) * x
                    st["acc_zero"] = 0
                elif base == "linfin":
                    res = (rd(st, G_REG, what) + st["lin"]) % q
                    st["lin"] = None
                elif base == "dotmac":
                    if st["acc"] is None:
                        raise JitHazard("%s accumulates into columns that were never cleared" % what)
                    st["acc"] += rd(st, ra, what) * st["coef"]
                    st["acc_zero"] = 0
                elif base in ("dotred", "dotfin"):
                    if st["acc"] is None:
                        raise JitHazard("%s reduces columns that were never cleared" % what)
                    g = (rd(st, G_REG, what) + st["acc"] * rinv) % q
                    st["acc"] = 0
                    st["acc_zero"] = 36
                    if base == "dotred":
                        reg[G_REG] = g
                    else:
                        res = g
                elif base in ("publish", "call"):
                    pass
                elif base in ("chkeq", "chkmul", "chkmul2", "chkadd", "chkdot"):
                    if base == "chkeq":
                        bad = rd(st, ra, what) != rd(st, rb, what)
                    elif base == "chkmul":
                        bad = rd(st, ra, what) * rd(st, rb, what) * rinv % q != rd(st, G_REG, what)
                    elif base == "chkmul2":
                        bad = rd(st, ra, what) * rd(st, rb, what) % q != rd(st, G_REG, what)
                    elif base == "chkadd":
                        bad = (rd(st, ra, what) + rd(st, rb, what)) % q != rd(st, G_REG, what)
                    else:
                        if st["acc"] is None:
                            raise JitHazard("%s reduces columns that were never cleared" % what)
                        bad = (rd(st, G_REG, what) + st["acc"] * rinv) % q != rd(st, ra, what)
                        st["acc"] = 0
                        st["acc_zero"] = 36
                    if bad and (fbmin[0] is None or st["sarg"] < fbmin[0]):
                        fbmin[0] = st["sarg"]
                else:
                    raise ValueError("no restatement of body %s" % name)
                # everything the body may touch is garbage afterwards, except what it defines
                for g in _GROUPS:
                    if g == G_REG and base in ("linp", "linn", "dotmac", "dotred"):
                        continue
                    if g == D_REG and base.startswith("chk"):
                        continue
                    if any(r in touched for r in range(g, g + 8)) and not isinstance(reg[g], _Pending):
                        if g == D_REG and res is None and base not in ("linfin", "dotfin"):
                            continue          # bodies without a value keep D (checked at build time: fpjit_bodies.parse_bodies)
                        reg[g] = POISON
                wacc = [r - ACC_REG for r in touched if ACC_REG <= r < ACC_REG + 36]
                if wacc and base not in ("linp", "linn", "dotmac", "dotred", "dotfin", "linfin", "chkdot"):
                    st["acc"] = st["lin"] = None
                    st["acc_zero"] = min(st["acc_zero"], min(wacc))
                if base == "linfin":
                    st["acc"] = None
                    st["acc_zero"] = 0
                if res is not None:
                    reg[D_REG] = res
            else:
                raise ValueError("IR op %r" % (k,))
        return None

    alive = True
    while alive:
        alive = False
        kinds = set()
        for s in range(ns):
            kd = run(s)
            if kd:
                alive = True
                kinds.add(kd)
        if len(kinds) > 1:
            raise JitHazard("strands disagree on the barrier kind")
    replay.first_bad = fbmin[0]
    return mem[:n_signals], status[0]


def replay_tape(tape, prog, bodies, inputs: dict):
    """convenience wrapper: Montgomery-form tapes are fed x R' and read back x R'^-1, as the runtime's ingest / egress do"""
    q = tape.q
    R = pow(2, tape.rbits, q)
    mont = bool(getattr(tape, "mont", False))
    if mont:
        inputs = {k: v % q * R % q for k, v in inputs.items()}
    info = {n: (set(b.vwritten), b.parity) for n, b in bodies.items()}
    sig, st = replay(prog.ir, info, q, tape.n_signals, tape.n_tslots, tape.n_lds, inputs, tape.rbits, R if mont else 1,
                     getattr(tape, "functions", ()), tape.consts)
    if mont:
        rinv = pow(R, -1, q)
        sig = [v * rinv % q for v in sig]
    return sig, st

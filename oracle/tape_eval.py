"""CPU oracle: evaluate a flattened circuit's op list on Python ints (one instance at a time).

TEST INFRASTRUCTURE ONLY.  Restates what the reference's emitted `<name>.cpp` does when `run(ctx)`
executes: each op is the corresponding Fr_* call (compute_bucket.rs:315-341) with the
semantics of oracle/field.py; COPY is Fr_copy (store_bucket.rs:641-643); ASSERT_EQ is the `===`
run-time check (assert_bucket.rs:70-89), reported instead of aborting.
Pure-Python loops: use for small cases only (the C oracle / oracle/_ref binary cover large ones).
"""
from __future__ import annotations

from .field import Field, FieldError

# operator numbering of circom_amd/opcodes.py (kept literal here so the oracle has no product import)
COPY, ADD, SUB, MUL, DIV, IDIV, MOD, POW, NEG = range(9)
SHL, SHR, BAND, BOR, BXOR, BNOT = range(9, 15)
LT, GT, LEQ, GEQ, EQ, NEQ, LAND, LOR, LNOT = range(15, 24)
SELECT, ASSERT_EQ, ASSERT_NZ, RUN, CALL, LOG = range(24, 30)
K_SIG, K_TMP, K_CONST, K_NONE = 0, 1, 2, 3
F_JZ, F_JMP, F_LDX, F_STX, F_RET = 100, 101, 102, 103, 104      # circom_amd/frontend/rtcode.py
CALL_STEP_LIMIT = 1 << 24
USE_NATIVE = True            # closed forms for functions that carry a `native` tag (False: always interpret the bytecode)

_BIN = {ADD: "add", SUB: "sub", MUL: "mul", DIV: "div", IDIV: "idiv", MOD: "mod", POW: "pow",
        SHL: "shl", SHR: "shr", BAND: "band", BOR: "bor", BXOR: "bxor", LT: "lt", GT: "gt",
        LEQ: "leq", GEQ: "geq", EQ: "eq", NEQ: "neq", LAND: "land", LOR: "lor"}
_UN = {NEG: "neg", BNOT: "bnot", LNOT: "lnot"}


def run_function(f: Field, fn: dict, regs: list, base: int, constants) -> bool:
    """Interpret the register bytecode of a circom function (frontend/rtcode.py) the way the reference's emitted C++
    would run it (while/if on Fr_isTrue, array addresses through Fr_toInt: loop_bucket.rs:76-91, branch_bucket.rs:100-122,
    compute_bucket.rs:361-363).  regs[base + r] = register r.  Returns False on an arithmetic error (integer division by
    zero, an array index outside its array, more than CALL_STEP_LIMIT instructions)."""
    q = f.q
    nat = fn.get("native")
    if nat is not None and USE_NATIVE:
        # a pure big-integer function with a closed form (circuits/bigint_func.py): the same values the body computes, without
        # ~10^6 interpreted steps per modular inverse; tests/test_ecdsa.py pins the two against each other and against the
        # reference runtime executing the body
        from circom_amd.circuits.bigint_func import native_eval
        kind, n, k, modulus = nat
        try:
            res = native_eval(kind, n, k, modulus, [regs[base + r] for r in range(fn["n_args"])])
        except ZeroDivisionError:
            res = None                   # long_div outside the contract of its tag: the body decides (the reference's semantics)
        if res is not None:
            for j, v in enumerate(res):
                regs[base + fn["ret_base"] + j] = v
            return True
    code = fn["code"]
    bins = {k: getattr(f, v) for k, v in _BIN.items()}
    uns = {k: getattr(f, v) for k, v in _UN.items()}

    def val(x):
        return regs[base + x[1]] if x[0] == 'r' else constants[x[1]]

    def to_int(v):              # Fr_toInt, generic/fr.cpp:1146-1170: small non-negative, or q - small
        if v < (1 << 31):
            return v
        if q - v <= (1 << 31):
            return v - q
        return None

    pc = 0
    steps = 0
    ok = True
    while True:
        steps += 1
        if steps > CALL_STEP_LIMIT:
            return False
        op, d, a, b = code[pc]
        pc += 1
        if op == F_RET:
            return ok
        if op == F_JMP:
            pc = d
        elif op == F_JZ:
            if val(a) == 0:
                pc = d
        elif op == F_LDX or op == F_STX:
            i = to_int(regs[base + b[0]])
            if i is None or not 0 <= i < b[1]:
                ok = False
                i = 0
            if op == F_LDX:
                regs[base + d] = regs[base + a + i]
            else:
                regs[base + d + i] = val(a)
        elif op == COPY:
            regs[base + d] = val(a)
        elif op in bins:
            try:
                regs[base + d] = bins[op](val(a), val(b))
            except FieldError:
                ok = False
                regs[base + d] = 0
        elif op in uns:
            regs[base + d] = uns[op](val(a))
        else:
            raise ValueError("bad function opcode %d" % op)


def eval_flat(q: int, n_signals: int, n_temps: int, constants, code, inputs: dict, functions=(), log=None, log_strings=()):
    """inputs: {signal slot: canonical value}.  Returns (signals list, failed_row or None).
    log: a list that receives the text of the log(...) statements, one entry per statement with its newline, as the
    emitted calculator prints them (log_bucket.rs:105-162: values through Fr_element2str = the canonical residue in
    decimal, arguments separated by one blank) up to the first failed check (where the reference process exits)."""
    f = Field(q)
    sig = [0] * n_signals
    sig[0] = 1
    for k, v in inputs.items():
        sig[k] = v % q
    tmp = [0] * max(n_temps, 1)
    op_, dk, dv = code["op"], code["dk"], code["dv"]
    ak, av, bk, bv, ck, cv = code["ak"], code["av"], code["bk"], code["bv"], code["ck"], code["cv"]
    bins = {k: getattr(f, v) for k, v in _BIN.items()}
    uns = {k: getattr(f, v) for k, v in _UN.items()}

    def rd(k, v):
        if k == K_SIG:
            return sig[v]
        if k == K_TMP:
            return tmp[v]
        return constants[v]

    failed = None
    line = []
    for i in range(len(op_)):
        op = int(op_[i])
        if op == LOG:                       # one argument of a log statement; dv = 1 on the row that ends it
            if int(ak[i]) != 3:             # K_NONE: a string (av = its id, -1 = no argument)
                line.append(str(rd(int(ak[i]), int(av[i])) % q))
            elif int(av[i]) >= 0:
                line.append(log_strings[int(av[i])].replace("%%", "%"))     # the string is printf's FORMAT (log_bucket.rs:126-133)
            if int(dv[i]):
                if log is not None and failed is None:
                    log.append(" ".join(line) + "\n")
                line = []
            continue
        if op == CALL:                      # a = function id, b = first register (a temporary)
            if not run_function(f, functions[int(av[i])], tmp, int(bv[i]), constants) and failed is None:
                failed = i
            continue
        a = rd(int(ak[i]), int(av[i]))
        if op in bins:
            try:
                r = bins[op](a, rd(int(bk[i]), int(bv[i])))
            except FieldError:
                if failed is None:
                    failed = i
                r = 0
        elif op == COPY:
            r = a
        elif op in uns:
            r = uns[op](a)
        elif op == SELECT:
            r = rd(int(bk[i]), int(bv[i])) if a != 0 else rd(int(ck[i]), int(cv[i]))
        elif op == ASSERT_EQ:
            if a != rd(int(bk[i]), int(bv[i])) and failed is None:
                failed = i
            continue
        elif op == ASSERT_NZ:
            if a == 0 and failed is None:
                failed = i
            continue
        else:
            raise ValueError("bad op %d" % op)
        if dk[i] == K_SIG:
            sig[int(dv[i])] = r
        else:
            tmp[int(dv[i])] = r
    return sig, failed


def check_r1cs(q: int, constraints, w) -> int | None:
    """First violated constraint index of A*B - C = 0 over witness w (wire 0 = 1), else None."""
    for i, (a, b, c) in enumerate(constraints):
        va = sum(co * w[k] for k, co in a.items()) % q
        vb = sum(co * w[k] for k, co in b.items()) % q
        vc = sum(co * w[k] for k, co in c.items()) % q
        if (va * vb - vc) % q:
            return i
    return None


# ---- lowered (device) schedule: same row format the HIP kernel consumes ---------------------------------
# D_* numbering of circom_amd/csrc/cw_tape.h
(D_COPY, D_ADD, D_SUB, D_NEG, D_MMUL, D_INV, D_IDIV, D_MOD, D_POW, D_SHL, D_SHR, D_BAND, D_BOR, D_BXOR,
 D_BNOT, D_LT, D_GT, D_LEQ, D_GEQ, D_EQ, D_NEQ, D_LAND, D_LOR, D_LNOT, D_SELECT, D_EXT, D_ASSERT_EQ,
 D_ASSERT_NZ, D_ALSO, D_BARRIER, D_MUL2, D_MADD, D_MULC, D_MADDC, D_LINSUM, D_BIT, D_DOTC, D_CALL, D_BITS) = range(39)   # D_ALSO: a spacer that does nothing
X_NEXT = 1 << 29
DF_JZ, DF_JMP, DF_LDX, DF_STX, DF_RET, DF_DIV = 100, 101, 102, 103, 104, 105     # device bytecode of circom functions (lower.py)
FN_CONST = 1 << 31
_DBIN = {D_ADD: "add", D_SUB: "sub", D_IDIV: "idiv", D_MOD: "mod", D_POW: "pow", D_SHL: "shl", D_SHR: "shr",
         D_BAND: "band", D_BOR: "bor", D_BXOR: "bxor", D_LT: "lt", D_GT: "gt", D_LEQ: "leq", D_GEQ: "geq",
         D_EQ: "eq", D_NEQ: "neq", D_LAND: "land", D_LOR: "lor"}
_DUN = {D_NEG: "neg", D_BNOT: "bnot", D_LNOT: "lnot", D_INV: "inv", D_COPY: None}


class ScheduleHazard(Exception):
    """The schedule violates the executor's memory model (race between strands / stale prefetch)."""


SH_DK, SH_AK, SH_BK, SH_NX = 8, 11, 14, 17
K_PREV, K_LDS, KD_NONE = 3, 4, 2
X_TMP, X_LDS = 1 << 31, 1 << 30


def run_dev_function(f: Field, code, regs: list, base: int, consts) -> bool:
    """the device form of a circom function (lower.py::_encode_function) executed the way the kernel's per-lane
    interpreter does; regs[base + r] = register r.  False = arithmetic error / step limit"""
    q = f.q
    bins = {k: getattr(f, v) for k, v in _DBIN.items()}
    uns = {k: (getattr(f, v) if v else (lambda x: x)) for k, v in _DUN.items()}

    def val(x):
        return consts[x & 0x7FFFFFFF] if x & FN_CONST else regs[base + x]

    pc = 0
    steps = 0
    ok = True
    while True:
        steps += 1
        if steps > CALL_STEP_LIMIT or pc >= len(code):
            return False
        op, d, a, b = (int(x) for x in code[pc])
        pc += 1
        if op == DF_RET:
            return ok
        if op == DF_JMP:
            pc = d
        elif op == DF_JZ:
            if val(a) == 0:
                pc = d
        elif op in (DF_LDX, DF_STX):
            v = regs[base + (b & 0xFFFF)]
            n = b >> 16
            i = v if v < (1 << 31) else (v - q if q - v <= (1 << 31) else None)
            if i is None or not 0 <= i < n:
                ok = False
                i = 0
            if op == DF_LDX:
                regs[base + d] = regs[base + a + i]
            else:
                regs[base + d + i] = val(a)
        elif op == D_MUL2:
            regs[base + d] = val(a) * val(b) % q
        elif op == DF_DIV:
            regs[base + d] = f.div(val(a), val(b))
        elif op in bins:
            try:
                regs[base + d] = bins[op](val(a), val(b))
            except FieldError:
                ok = False
                regs[base + d] = 0
        elif op in uns:
            regs[base + d] = uns[op](val(a))
        else:
            raise ValueError("bad device function opcode %d" % op)


def eval_rows(q: int, n_signals: int, n_tslots: int, consts, rows, inputs: dict, rbits: int = 261,
              stream_off=None, extras=None, extra_off=None, n_lds: int = 0, terms=None, term_off=None, lconsts=(),
              functions=(), one: int = 1, seqs=None, seq_off=None):
    """Evaluate a lowered schedule exactly the way cw_eval_kernel does, for one instance:
      * every strand (stream) walks its own rows; strands meet at BARRIER rows,
      * operands of row r+1 are fetched BEFORE row r stores its result (one-row-ahead prefetch), except
        kind-3 operands, which are the previous value-producing row's result held in a register,
      * a row's n_extra further destinations come from the strand's extra-destination table,
      * LDS slots hand values between strands; a LIGHT barrier orders LDS traffic only, a FULL barrier
        (dst = 1) also makes earlier global stores of other strands visible.
    Strands of one epoch are simulated one after the other; any cross-strand read-after-write or
    write-after-read inside an epoch, and any cross-strand global read not separated from its write by a
    FULL barrier, is reported as a ScheduleHazard (on the GPU it would be a race).
    Returns (signal values, status) with status = 0 | bits + (index << 8) like the kernel: `seqs` names, for every row
    that can fail (in stream order), the flat operation it comes from, and the failing check with the SMALLEST index wins
    whatever strand hits it and in whatever order - the check the reference's sequential program stops at; without
    `seqs` the schedule row of the first failure met is reported."""
    f = Field(q)
    rinv = pow(1 << rbits, -1, q)        # MMUL = a*b*R'^-1 with the schedule's radix (device: R' = 2^261)
    sig = [0] * n_signals
    sig[0] = one                         # the constant-one signal (R' when the table holds Montgomery forms)
    for k, v in inputs.items():
        sig[k] = v % q
    tmp = [0] * max(n_tslots, 1)
    lds = [0] * max(n_lds, 1)
    bins = {k: getattr(f, v) for k, v in _DBIN.items()}
    uns = {k: (getattr(f, v) if v else (lambda x: x)) for k, v in _DUN.items()}
    rows = [tuple(int(x) for x in r) for r in rows]
    if stream_off is None:
        stream_off = [0, len(rows)]
    ns = len(stream_off) - 1
    if extras is None:
        extras, extra_off = [], [0] * (ns + 1)
    extras = [int(x) for x in extras]
    xp = [int(extra_off[s]) for s in range(ns)]
    if terms is None:
        terms, term_off = [], [0] * (ns + 1)
    terms = [tuple(int(x) for x in t) for t in terms]
    tp = [int(term_off[s]) for s in range(ns)]
    pc = [int(stream_off[s]) for s in range(ns)]
    end = [int(stream_off[s + 1]) for s in range(ns)]
    prev = [0] * ns
    sel = [False] * ns
    status = [0]
    sp = [int(seq_off[s]) for s in range(ns)] if seqs is not None else None

    def fail(s, bits, r):
        if seqs is None:
            if status[0] == 0:
                status[0] = bits | (r << 8)
            return
        word = bits | (int(seqs[sp[s]]) << 8)
        if status[0] == 0 or (word >> 8) < (status[0] >> 8):
            status[0] = word

    writer = {}      # (kind, slot) -> (epoch, strand) of the last write
    reader = {}      # (kind, slot) -> set of strands that read it in the current epoch
    state = {"epoch": 0, "last_full": -1}

    def store_of(k):
        return sig if k == 0 else (tmp if k == 1 else lds)

    def mem_read(s, k, v):
        key = (k, v)
        w = writer.get(key)
        if w is not None and w[1] != s:
            if w[0] == state["epoch"]:
                raise ScheduleHazard("strand %d reads %s written by strand %d in the same epoch" % (s, key, w[1]))
            if k != K_LDS and w[0] > state["last_full"]:
                raise ScheduleHazard("strand %d reads global %s written by strand %d with no FULL barrier in between"
                                     % (s, key, w[1]))
        reader.setdefault(key, set()).add(s)
        return store_of(k)[v]

    def fetch(s, k, v):
        if k == 2:
            return consts[v]
        if k == K_PREV:
            return None               # resolved at execution time
        return mem_read(s, k, v)

    def mem_write(s, k, v, val):
        key = (k, v)
        rs = reader.get(key)
        if rs and (rs - {s}):
            raise ScheduleHazard("strand %d overwrites %s read by another strand in the same epoch" % (s, key))
        w = writer.get(key)
        if w is not None and w[0] == state["epoch"] and w[1] != s:
            raise ScheduleHazard("two strands write %s in the same epoch" % (key,))
        writer[key] = (state["epoch"], s)
        store_of(k)[v] = val

    def operands_of(s, r):
        w0, _, a_, b_ = rows[r]
        op = w0 & 0xFF
        if op == D_BARRIER or op == D_LINSUM or op == D_DOTC or op == D_CALL:
            return None, None                      # LINSUM / DOTC / CALL read their operands at execution time
        if op == D_BIT or op == D_BITS:
            return fetch(s, (w0 >> SH_AK) & 7, a_), None
        ak, bk = (w0 >> SH_AK) & 7, (w0 >> SH_BK) & 7
        a = fetch(s, ak, a_)
        b = None if op in _DUN or op in (D_ASSERT_NZ, D_SELECT) else fetch(s, bk, b_)
        return a, b

    def run_strand(s):
        """run strand s up to (and over) its next BARRIER; returns 'light'/'full' or None when the stream ended"""
        r = pc[s]
        if r >= end[s]:
            return None
        pre = operands_of(s, r)
        while r < end[s]:
            w0, dst, a_, b_ = rows[r]
            op, dk, ak, bk = w0 & 0xFF, (w0 >> SH_DK) & 7, (w0 >> SH_AK) & 7, (w0 >> SH_BK) & 7
            nx = (w0 >> SH_NX) & 0xFFF
            a, b = pre
            nxt = r + 1
            if op == D_BARRIER:            # nothing is prefetched across a barrier
                pc[s] = r + 1
                return "full" if dst == 1 else "light"
            pre = operands_of(s, nxt) if nxt < end[s] else (None, None)    # prefetch BEFORE this row's stores
            if ak == K_PREV:
                a = prev[s]
            if bk == K_PREV:
                b = prev[s]
            res = None
            if op == D_MMUL:
                res = a * b * rinv % q
            elif op == D_MUL2:
                res = a * b % q
            elif op == D_MADD:
                res = (a * b * rinv + prev[s]) % q
            elif op in (D_MULC, D_MADDC):
                # operand b = scaled constant c*R'; [b+1] = |val(c)| with flags in word 0 (checked for consistency)
                flag = (w0 >> 29) & 3
                res = a * b * rinv % q
                if flag:
                    mag = consts[b_ + 1]
                    c_plain = mag if flag == 1 else (q - mag) % q
                    if mag >= (1 << 63) or (a * c_plain) % q != res:
                        raise ScheduleHazard("constant pair of row %d is inconsistent" % r)
                if op == D_MADDC:
                    res = (res + prev[s]) % q
            elif op in bins:
                try:
                    res = bins[op](a, b)
                except FieldError:
                    fail(s, 2, r)
                    res = 0
            elif op in uns:
                res = uns[op](a)
            elif op == D_LINSUM:
                acc = consts[b_] if bk == 2 else 0
                for (tk, tv, lo, hi) in terms[tp[s]:tp[s] + a_]:
                    kind = tk & 7
                    x = prev[s] if kind == K_PREV else mem_read(s, kind, tv)
                    cf = lo | (hi << 32)
                    acc += -cf * x if tk >> 31 else cf * x
                tp[s] += a_
                res = acc % q
            elif op == D_DOTC:
                acc = consts[b_] if bk == 2 else 0
                for (tk, tv, lo, hi) in terms[tp[s]:tp[s] + a_]:
                    kind = tk & 7
                    x = prev[s] if kind == K_PREV else mem_read(s, kind, tv)
                    acc += x * lconsts[lo] * rinv           # limb-table constant = coef * R'
                tp[s] += a_
                res = acc % q
            elif op == D_BIT:
                res = (a >> b_) & 1 if b_ < 256 else 0
            elif op == D_BITS:                # consecutive bits of a from bit b: the entries of the extra table say where each goes
                bit = b_
                for e in extras[xp[s]:xp[s] + nx]:
                    if e & X_LDS:
                        raise ValueError("bit-field destinations live in the value table")
                    if e & X_NEXT:
                        bit += 1
                    if bit >= 256:
                        raise ValueError("bit-field row runs past bit 255")
                    mem_write(s, 1 if e & X_TMP else 0, e & 0x1FFFFFFF, (a >> bit) & 1)
                xp[s] += nx
                nx = 0
            elif op == D_ALSO:
                pass
            elif op == D_CALL:                # a = function id, b = first register slot (pinned temps)
                n_regs, fcode, native = functions[a_]
                for k in range(n_regs):       # the interpreter reads and writes its registers in the value table: what another
                    w_ = writer.get((1, b_ + k))   # strand stored there (the arguments) must be behind a FULL barrier
                    if w_ is not None and w_[1] != s:
                        mem_read(s, 1, b_ + k)
                for k in range(n_regs):
                    mem_write(s, 1, b_ + k, tmp[b_ + k])
                if native is not None and USE_NATIVE:      # the device computes the closed form (eval_call_native)
                    from circom_amd.circuits.bigint_func import native_eval
                    kind, n_, k_, modulus = native
                    from circom_amd.circuits.bigint_func import native_n_args
                    kname = {1: "mod_inv", 2: "ec_add", 3: "ec_double", 4: "long_div"}[kind]
                    n_args = native_n_args(kname, k_, modulus)
                    try:
                        for j, v in enumerate(native_eval(kname, n_, k_, modulus, [tmp[b_ + x] for x in range(n_args)])):
                            tmp[b_ + n_args + j] = v
                    except ZeroDivisionError:     # long_div on a divisor outside its contract: the device flags the lane
                        fail(s, 2, r)
                elif not run_dev_function(f, fcode, tmp, b_, consts):
                    fail(s, 2, r)
            elif op == D_SELECT:
                sel[s] = a != 0               # latched lane mask; no value
            elif op == D_EXT:
                res = a if sel[s] else b
            elif op == D_ASSERT_EQ:
                if a != b:
                    fail(s, 1, r)
            elif op == D_ASSERT_NZ:
                if a == 0:
                    fail(s, 1, r)
            else:
                raise ValueError("bad device op %d" % op)
            if sp is not None and op in (D_ASSERT_EQ, D_ASSERT_NZ, D_IDIV, D_MOD, D_CALL):
                sp[s] += 1
            if res is not None:
                prev[s] = res
                if dk != KD_NONE:
                    mem_write(s, dk, dst, res)
                for e in extras[xp[s]:xp[s] + nx]:
                    if e & X_LDS:
                        mem_write(s, K_LDS, e & 0x3FFFFFFF, res)
                    elif e & X_TMP:
                        mem_write(s, 1, e & 0x3FFFFFFF, res)
                    else:
                        mem_write(s, 0, e, res)
            elif nx:
                raise ValueError("extra destinations on a row without a value")
            xp[s] += nx
            r = nxt
        pc[s] = r
        return None

    alive = True
    while alive:
        alive = False
        kinds = set()
        for s in range(ns):
            k = run_strand(s)
            if k:
                alive = True
                kinds.add(k)
        if len(kinds) > 1:
            raise ScheduleHazard("strands disagree on the barrier kind closing epoch %d" % state["epoch"])
        if "full" in kinds:
            state["last_full"] = state["epoch"]
        state["epoch"] += 1
        reader.clear()
    return sig, status[0]


def eval_tape(tape, inputs: dict):
    """Convenience wrapper over a circom_amd.hip_elements.lower.Tape (duck-typed).  A tape lowered with signals in
    Montgomery form (tape.mont) is fed x R' and its signals are multiplied by R'^-1, as the runtime's ingest / egress do."""
    mont = bool(getattr(tape, "mont", False))
    R = pow(2, tape.rbits, tape.q)
    if mont:
        inputs = {k: v % tape.q * R % tape.q for k, v in inputs.items()}
    one = R if mont else 1
    if getattr(tape, "kind", 0) == 1:
        sig, st = eval_pipe(tape.q, tape.n_signals, tape.n_tslots, tape.consts, tape.rows, tape.extras, tape.terms, tape.lconsts,
                            tape.pipe, inputs, tape.rbits, one)
    else:
        sig, st = eval_rows(tape.q, tape.n_signals, tape.n_tslots, tape.consts, tape.rows, inputs, tape.rbits,
                            tape.stream_off, tape.extras, tape.extra_off, tape.n_lds, tape.terms, tape.term_off, tape.lconsts,
                            getattr(tape, "functions", ()), one, getattr(tape, "seqs", None), getattr(tape, "seq_off", None))
    if mont:
        rinv = pow(R, -1, tape.q)
        sig = [v * rinv % tape.q for v in sig]
    return sig, st


# ---- pipelined single-wave schedule (circom_amd/hip_elements/pipe.py), executed the way cw_pipe_kernel does ------------
P_NONE = 0xFFFFFFFF
PX_TMP, PX_CONST = 1 << 31, 1 << 30
P_ENTRY_NONE = 0xFF
D_NOP = 255


def eval_pipe(q: int, n_signals: int, n_tslots: int, consts, rows, loads, terms, lconsts, pipe, inputs: dict, rbits: int = 261,
              one: int = 1):
    """Replays a pipelined schedule for one instance with the kernel's timing:
      * L(k), the load list of batch k, reads the value table / constant table when batch k-1 starts (after batch k-2's
        loads have landed) and lands in the staging half k % 2 when batch k starts; L(0) and L(1) are issued up front;
      * the a/b operands of row r+1 are read from LDS BEFORE row r writes its ring entry, except for the first row of a
        batch, which reads them after its batch's loads have landed; LINSUM / DOTC terms are read when the row executes;
      * a value-producing row writes ring entry `d`, then its two store targets.
    Raises ScheduleHazard for: an LDS entry read before anything wrote it, a staging entry of the half that is being
    filled, a load of a temp slot nobody wrote, a store on a row without a value.
    Returns (signal values, status) like eval_rows."""
    f = Field(q)
    nb, nld, rr = pipe
    rinv = pow(1 << rbits, -1, q)
    sig = [0] * n_signals
    sig[0] = one
    for k, v in inputs.items():
        sig[k] = v % q
    tmp = [None] * max(n_tslots, 1)
    n_ent = rr + 2 * nld
    lds = [None] * n_ent
    rows = [tuple(int(x) for x in r) for r in rows]
    loads = [int(x) for x in loads]
    terms = [tuple(int(x) for x in t) for t in terms]
    n_rows = len(rows)
    if n_rows % nb or len(loads) != (n_rows // nb + 2) * nld:
        raise ScheduleHazard("row / load counts do not form whole batches")
    bins = {k: getattr(f, v) for k, v in _DBIN.items()}
    uns = {k: (getattr(f, v) if v else (lambda x: x)) for k, v in _DUN.items()}
    status = 0
    pending = {}

    def issue(k):
        vals = []
        for j in range(nld):
            lw = loads[k * nld + j]
            if lw == P_NONE:
                vals.append(None)
            elif lw & PX_TMP:
                v = tmp[lw & 0x3FFFFFFF]
                if v is None:
                    raise ScheduleHazard("batch %d loads temp slot %d before any row stored it" % (k, lw & 0x3FFFFFFF))
                vals.append(v)
            elif lw & PX_CONST:
                vals.append(consts[lw & 0x3FFFFFFF])
            else:
                vals.append(sig[lw])
        pending[k] = vals

    def land(k):
        base = rr + (k & 1) * nld
        for j, v in enumerate(pending.pop(k)):
            lds[base + j] = v

    def lds_read(pos, e):
        k = pos // nb
        if e >= n_ent:
            raise ScheduleHazard("row %d: LDS entry %d out of range" % (pos, e))
        if e >= rr and (e - rr) // nld != (k & 1):
            raise ScheduleHazard("row %d reads the staging half that is being filled" % pos)
        v = lds[e]
        if v is None:
            raise ScheduleHazard("row %d reads LDS entry %d, which holds nothing" % (pos, e))
        return v

    def operands(pos):
        w0, _, abd = rows[pos][:3]
        op = w0 & 0xFF
        ak, bk = (w0 >> 8) & 7, (w0 >> 11) & 7
        a = lds_read(pos, abd & 0xFF) if ak == K_LDS else None
        b = lds_read(pos, (abd >> 8) & 0xFF) if bk == K_LDS else None
        return a, b

    issue(0)
    land(0)
    issue(1)
    prev = 0
    sel = False
    tp = 0
    pre = (None, None)
    for pos in range(n_rows):
        if pos % nb == 0:
            if pos:
                land(pos // nb)
                issue(pos // nb + 1)
            pre = operands(pos)
        w0, aux, abd, st0, st1, cml, cmh, _ = rows[pos]
        op, ak, bk, flag = w0 & 0xFF, (w0 >> 8) & 7, (w0 >> 11) & 7, (w0 >> 29) & 3
        a, b = pre
        if pos + 1 < n_rows and (pos + 1) % nb:
            pre = operands(pos + 1)
        if ak == K_PREV:
            a = prev
        if bk == K_PREV:
            b = prev
        res = None
        if op == D_NOP:
            pass
        elif op == D_MMUL:
            res = a * b * rinv % q
        elif op == D_MUL2:
            res = a * b % q
        elif op == D_MADD:
            res = (a * b * rinv + prev) % q
        elif op in (D_MULC, D_MADDC):
            res = a * b * rinv % q
            if flag:
                mag = cml | (cmh << 32)
                c_plain = mag if flag == 1 else (q - mag) % q
                if mag >= (1 << 63) or (a * c_plain) % q != res:
                    raise ScheduleHazard("constant pair of row %d is inconsistent" % pos)
            if op == D_MADDC:
                res = (res + prev) % q
        elif op in bins:
            try:
                res = bins[op](a, b)
            except FieldError:
                if status == 0 or aux < (status >> 8):
                    status = 2 | (aux << 8)
                res = 0
        elif op in uns:
            res = uns[op](a)
        elif op in (D_LINSUM, D_DOTC):
            acc = b if bk else 0
            for (tk, te, lo, hi) in terms[tp:tp + aux]:
                kind = tk & 7
                x = prev if kind == K_PREV else lds_read(pos, te)
                if kind not in (K_PREV, K_LDS):
                    raise ScheduleHazard("row %d: term kind %d" % (pos, kind))
                if op == D_DOTC:
                    acc += x * lconsts[lo] * rinv
                else:
                    cf = lo | (hi << 32)
                    acc += -cf * x if tk >> 31 else cf * x
            tp += aux
            res = acc % q
        elif op == D_BIT:
            res = (a >> aux) & 1 if aux < 256 else 0
        elif op == D_SELECT:
            sel = a != 0
        elif op == D_EXT:
            res = a if sel else b
        elif op == D_ASSERT_EQ:
            if a != b and (status == 0 or aux < (status >> 8)):
                status = 1 | (aux << 8)
        elif op == D_ASSERT_NZ:
            if a == 0 and (status == 0 or aux < (status >> 8)):
                status = 1 | (aux << 8)
        else:
            raise ValueError("bad device op %d in a pipelined schedule" % op)
        d = (abd >> 16) & 0xFF
        if res is not None:
            prev = res
            if d != P_ENTRY_NONE:
                if d >= rr:
                    raise ScheduleHazard("row %d writes outside the ring" % pos)
                lds[d] = res
            for st in (st0, st1):
                if st == P_NONE:
                    continue
                if st & PX_TMP:
                    tmp[st & 0x3FFFFFFF] = res
                else:
                    sig[st] = res
        elif d != P_ENTRY_NONE or st0 != P_NONE or st1 != P_NONE:
            raise ScheduleHazard("row %d has destinations but no value" % pos)
    if tp != len(terms) - 4 and tp != len(terms):
        raise ScheduleHazard("term table not consumed exactly")
    return sig, status


# ---- bit-plane program (circom_amd/hip_elements/bitsched.py), executed the way cw_bits_eval_kernel does ----------------
B_IN_BASE = 3
B_BATCH = 8
B_MAX_LOADS = 4
B_MAX_FLUSH = 6
B_CMD_WORDS = 24
B_K_AND = 1
B_K_OR = 2


def eval_bits(recs, cmds, ring: int, cache: int, n_slots: int, input_masks: dict, width: int = 1):
    """Replays a bit-plane program on `width` instances at once (python ints as masks: bit i = instance i).
    recs: [n_vrows * 64][2] record words, cmds: [n_batches][B_CMD_WORDS] command blocks (bitsched.py); input_masks:
    bit-table slot -> mask of the main inputs (slot B_IN_BASE + k for input k).  Mirrors the kernel's timing:
      * the three LDS operands of vrow v+1 are read BEFORE vrow v writes its result entry,
      * row loads of batch b read the bit table when the batch starts (they see flushes of batches <= b-1) and land in
        their cache slot between steps 6 and 7 of batch b+1,
      * flushes of batch b copy a cache slot to the bit table DURING batch b+1: the first two entries of the command block
        are read before that batch's first vrow writes, the others after it (cw_bits.hip); their stores are issued in
        batch b+1, so a load sees flushes of batches <= its own - 2,
    LDS entries and table slots start POISONED (None): using a poisoned operand, reading a table row that was never
    written, or any offset outside the areas raises ScheduleHazard.  Returns the bit table (list of masks / None)."""
    full = (1 << width) - 1
    n_vrows = len(recs) // 64
    n_batches = n_vrows // B_BATCH
    if n_vrows % B_BATCH or len(cmds) != n_batches:
        raise ScheduleHazard("program is not a whole number of batches with one command block each")
    if n_slots % 64:
        raise ScheduleHazard("bit table is not a whole number of rows")
    T = [None] * n_slots
    T[0], T[1], T[2] = 0, full, 0
    n_in_rows = 0
    for s, m in input_masks.items():
        T[s] = m & full
        n_in_rows = max(n_in_rows, s // 64 + 1)
    for s in range(n_in_rows * 64):          # padding of the input rows: the init kernel zeroes what ingest does not write
        if T[s] is None:
            T[s] = 0
    ring_bytes = ring * 512
    const_off = (ring + cache) * 512
    lds_entries = (ring + cache) * 64 + 2
    lds = [None] * lds_entries
    lds[const_off // 8] = 0
    lds[const_off // 8 + 1] = full
    recs = [(int(r[0]), int(r[1])) for r in recs]
    cmds = [[int(x) for x in c] for c in cmds]

    def entry(v, lane, off, what):
        if off % 8 or off // 8 >= lds_entries:
            raise ScheduleHazard("vrow %d lane %d: %s offset outside the LDS areas" % (v, lane, what))
        return off // 8

    def early(v):
        out = []
        for lane in range(64):
            w0, w1 = recs[v * 64 + lane]
            vals = []
            for what, off in (("a", w0 & 0xFFF8), ("b", w0 >> 16), ("c", w1 & 0xFFFF)):
                x = lds[entry(v, lane, off, what)]
                if x is None:
                    raise ScheduleHazard("vrow %d lane %d reads LDS entry %d (operand %s) that holds no value" % (v, lane, off // 8, what))
                vals.append(x)
            out.append(vals)
        return out

    def parse_cmd(b):
        c = cmds[b]
        nl, nf = c[0] & 0xFF, (c[0] >> 8) & 0xFF
        if nl > B_MAX_LOADS or nf > B_MAX_FLUSH or c[0] >> 16:
            raise ScheduleHazard("batch %d: command counts" % b)
        out = []
        for j in range(nl + nf):
            k = j if j < nl else B_MAX_LOADS + (j - nl)
            goff, loff = c[2 + 2 * k], c[3 + 2 * k]
            if goff % 512 or goff // 8 + 64 > n_slots:
                raise ScheduleHazard("batch %d: row outside the bit table" % b)
            if loff % 512 or loff < ring_bytes or loff + 512 > const_off:
                raise ScheduleHazard("batch %d: cache slot outside the cache area" % b)
            out.append((goff // 8, loff // 8))
        return out[:nl], out[nl:]

    fetched = early(0) if n_vrows else []
    in_flight = []                           # rows requested by the previous batch: (values, LDS entry)
    pending_flush = []                       # flushes of the previous batch: (table slot, LDS entry)
    for b in range(n_batches + 1):
        if b == n_batches:                   # the kernel's drain after the last batch
            for g, l in pending_flush:
                T[g:g + 64] = lds[l:l + 64]
            break
        loads, flushes = parse_cmd(b)
        early_flush = [(g, lds[l:l + 64]) for g, l in pending_flush[:2]]      # read before this batch's first vrow writes
        late_flush = pending_flush[2:]
        requested = []
        for g, l in loads:                   # requested when the batch starts
            row = T[g:g + 64]
            if any(x is None for x in row) and all(x is None for x in row):
                raise ScheduleHazard("batch %d loads bit-table row %d before it was written" % (b, g // 64))
            requested.append((row, l))
        for k in range(B_BATCH):
            v = b * B_BATCH + k
            if k == B_BATCH - 1:             # the rows requested a batch ago land before the last step's operand reads
                for row, l in in_flight:
                    lds[l:l + 64] = row
                in_flight = []
            nxt = early(v + 1) if v + 1 < n_vrows else None
            for lane in range(64):
                w0, w1 = recs[v * 64 + lane]
                a_, b_, c_ = fetched[lane]
                u = (a_ & b_) if (w0 & B_K_AND) else (a_ ^ b_)
                r = (u | c_) if (w0 & B_K_OR) else (u ^ c_)
                if (w0 >> 2) & 1:
                    raise ScheduleHazard("vrow %d lane %d: reserved record bit" % (v, lane))
                lds[entry(v, lane, w1 >> 16, "destination")] = r
                if (w1 >> 16) >= const_off:
                    raise ScheduleHazard("vrow %d lane %d writes a constant entry" % (v, lane))
            fetched = nxt
            if k == 0:                       # the stores of the previous batch's flushes leave after this batch's first vrow
                for g, row in early_flush:
                    T[g:g + 64] = row
                for g, l in late_flush:
                    T[g:g + 64] = lds[l:l + 64]
        in_flight = requested
        for g, l in flushes:
            if g // 64 < n_in_rows:
                raise ScheduleHazard("batch %d flushes onto a constant / input row" % b)
        pending_flush = flushes
    return T

"""Run-time code of circom *functions*: loops and branches whose conditions are run-time values, arrays indexed by
run-time values ("tier 2").

In the reference such code is emitted as real C++ control flow: `while (Fr_isTrue(&cond))` (loop_bucket.rs:76-91),
`if (Fr_isTrue(...))` (branch_bucket.rs:100-122), array addresses through `Fr_toInt` (compute_bucket.rs:361-363,
generic/fr.cpp:1146-1170), function calls on a private `lvar` arena (call_bucket.rs:466-533, function.rs:91-127).  A trace
cannot unroll that: the trip count differs per input.  Here a function is a small register bytecode that

  * the oracle interprets on Python ints (oracle/tape_eval.py),
  * oracle/emit_ref_cpp.py prints as C++ over the reference's own `Fr_*` calls (so the reference RUNTIME executes it),
  * the HIP kernel interprets per lane with a per-lane program counter (csrc/cw_kernels.hip, D_CALL): lanes of a wave
    that sit at different instructions take turns (divergence, lowest program counter first), every lane stops after
    CALL_STEP_LIMIT instructions.

Registers are field elements (one 256-bit value per instance).  Instruction = (opcode, dst, a, b):
  ALU      opcode = circom_amd.opcodes (ADD .. LNOT, COPY, NEG, BNOT): dst <- a op b; operands are registers or constants
  F_JZ     jump to `dst` when register a == 0;  F_JMP  jump to `dst`
  F_LDX    dst <- reg[a + toInt(reg[b.idx])]   (b = (index register, array length): out-of-range index = arithmetic error)
  F_STX    reg[dst + toInt(reg[b.idx])] <- a
  F_RET    end
Operand encoding: ('r', register) | ('c', canonical constant value).
"""
from __future__ import annotations

from .. import opcodes as O

F_JZ, F_JMP, F_LDX, F_STX, F_RET = 100, 101, 102, 103, 104
CALL_STEP_LIMIT = 1 << 24


class RtError(Exception):
    pass


class RtVar:
    """a register of the function being built (or a constant)"""
    __slots__ = ("f", "kind", "val")

    def __init__(self, f, kind, val):
        self.f, self.kind, self.val = f, kind, val

    def _bin(self, op, other, swap=False):
        o = self.f.lift(other)
        a, b = (o, self) if swap else (self, o)
        return self.f.emit(op, a, b)

    def __add__(self, o): return self._bin(O.ADD, o)
    def __radd__(self, o): return self._bin(O.ADD, o, True)
    def __sub__(self, o): return self._bin(O.SUB, o)
    def __rsub__(self, o): return self._bin(O.SUB, o, True)
    def __mul__(self, o): return self._bin(O.MUL, o)
    def __rmul__(self, o): return self._bin(O.MUL, o, True)
    def __truediv__(self, o): return self._bin(O.DIV, o)
    def __floordiv__(self, o): return self._bin(O.IDIV, o)
    def __mod__(self, o): return self._bin(O.MOD, o)
    def __lshift__(self, o): return self._bin(O.SHL, o)
    def __rshift__(self, o): return self._bin(O.SHR, o)
    def __and__(self, o): return self._bin(O.BAND, o)
    def __or__(self, o): return self._bin(O.BOR, o)
    def __xor__(self, o): return self._bin(O.BXOR, o)
    def __neg__(self): return self.f.emit(O.NEG, self, None)
    def lt(self, o): return self._bin(O.LT, o)
    def gt(self, o): return self._bin(O.GT, o)
    def leq(self, o): return self._bin(O.LEQ, o)
    def geq(self, o): return self._bin(O.GEQ, o)
    def eq(self, o): return self._bin(O.EQ, o)
    def neq(self, o): return self._bin(O.NEQ, o)
    def land(self, o): return self._bin(O.LAND, o)
    def lor(self, o): return self._bin(O.LOR, o)
    def lnot(self): return self.f.emit(O.LNOT, self, None)

    def set(self, value):
        """`var = value;` (mutable circom `var`)"""
        if self.kind != 'r':
            raise RtError("cannot assign to a constant")
        v = self.f.lift(value)
        self.f.code.append((O.COPY, self.val, (v.kind, v.val), None))


class RtArray:
    """circom `var x[n]`: a block of registers; x[python int] is a register, x.load(v) / x.store(v, value) index it
    with a run-time value"""

    def __init__(self, f, base, n):
        self.f, self.base, self.n = f, base, n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if not 0 <= i < self.n:
            raise RtError("array index out of range")
        return RtVar(self.f, 'r', self.base + i)

    def load(self, idx) -> RtVar:
        idx = self.f.reg_of(idx)
        d = self.f.new_reg()
        self.f.code.append((F_LDX, d, self.base, (idx, self.n)))
        return RtVar(self.f, 'r', d)

    def store(self, idx, value):
        idx = self.f.reg_of(idx)
        v = self.f.lift(value)
        self.f.code.append((F_STX, self.base, (v.kind, v.val), (idx, self.n)))


class _Loop:
    def __init__(self, f):
        self.f = f
        self.breaks = []

    def __enter__(self):
        self.start = len(self.f.code)
        return self

    def break_unless(self, cond):
        """leave the loop when cond == 0 (`while (cond) { ... }` = break_unless(cond) at the top)"""
        r = self.f.reg_of(cond)
        self.breaks.append(len(self.f.code))
        self.f.code.append([F_JZ, None, ('r', r), None])

    def __exit__(self, *exc):
        if exc[0] is not None:
            return False
        self.f.code.append((F_JMP, self.start, None, None))
        end = len(self.f.code)
        for b in self.breaks:
            self.f.code[b][1] = end
        return False


class _If:
    def __init__(self, f, cond):
        self.f = f
        self.r = f.reg_of(cond)

    def __enter__(self):
        self.jz = len(self.f.code)
        self.f.code.append([F_JZ, None, ('r', self.r), None])
        return self

    def __exit__(self, *exc):
        if exc[0] is not None:
            return False
        self.f.code[self.jz][1] = len(self.f.code)
        self.f.last_if = self
        return False


class _Else:
    def __init__(self, f):
        self.f = f
        self.iff = f.last_if
        if self.iff is None or self.iff.f.code[self.iff.jz][1] != len(f.code):
            raise RtError("else_() must directly follow an if_() block")

    def __enter__(self):
        self.jmp = len(self.f.code)
        self.f.code.append([F_JMP, None, None, None])
        self.f.code[self.iff.jz][1] = len(self.f.code)      # the false branch starts here
        return self

    def __exit__(self, *exc):
        if exc[0] is not None:
            return False
        self.f.code[self.jmp][1] = len(self.f.code)
        return False


class RtFunction:
    """`function name(args) { ... return ...; }`.  build(f, *args) writes the body through the builder and returns the
    list of results.  After construction: n_args, n_ret, n_regs, code (list of 4-tuples with resolved targets)."""

    def __init__(self, name: str, n_args: int, build, fp):
        self.name = name
        self.fp = fp
        self.n_args = n_args
        self.code = []
        self.n_regs = n_args
        self.last_if = None
        args = [RtVar(self, 'r', k) for k in range(n_args)]
        res = build(self, *args)
        if isinstance(res, RtVar):
            res = [res]
        res = [self.lift(x) for x in res]
        self.n_ret = len(res)
        # results land in dedicated registers (never written by the caller: the flat code stays single-assignment)
        self.ret_base = self.n_regs
        self.n_regs += self.n_ret
        for k, x in enumerate(res):
            self.code.append((O.COPY, self.ret_base + k, (x.kind, x.val), None))
        self.code.append((F_RET, 0, None, None))
        self.code = [tuple(c) for c in self.code]
        for c in self.code:
            if c[0] in (F_JZ, F_JMP) and c[1] is None:
                raise RtError("unresolved jump")
        self.code_built, self.n_regs_built, self.ret_base_built = self.code, self.n_regs, self.ret_base   # as written (tests)
        self._optimise()
        self.id = None                # set by Program.register_function
        self.native = None            # (kind, n, k, modulus): closed form of a pure big-integer function (circuits/bigint_func.py)

    # ---- bytecode optimisation ------------------------------------------------------------------------------------
    # The builder is SSA-flavoured: every operator result and every `var` gets a fresh register and `var = expr` is an
    # operator followed by a COPY.  On the device a register is a 32-byte column of the value table and every
    # instruction is a round trip to it (csrc/cw_kernels.hip eval_call), so both the instruction count and the size of
    # the register window (does it stay in the L2?) are the cost.  Two passes, both on the finished bytecode:
    #   1. an operator whose result is only read by the COPY right behind it writes the COPY's destination itself;
    #   2. registers are renumbered by live range (backward dataflow over the jumps, then a linear scan): arguments keep
    #      their places, results follow them, blocks addressed through F_LDX / F_STX stay contiguous and live throughout.
    @staticmethod
    def _reads(c):
        """(scalar registers read, array windows (base, n) touched) of one instruction"""
        op = c[0]
        regs, wins = [], []
        if op == F_RET or op == F_JMP:
            return regs, wins
        if op == F_JZ:
            return [c[2][1]], wins
        if op == F_LDX:
            return [c[3][0]], [(c[2], c[3][1])]
        if op == F_STX:
            regs.append(c[3][0])
            if c[2][0] == 'r':
                regs.append(c[2][1])
            return regs, [(c[1], c[3][1])]
        for o in (c[2], c[3]):
            if o is not None and o[0] == 'r':
                regs.append(o[1])
        return regs, wins

    def _optimise(self):
        code = self.code
        n = len(code)
        in_block = set()                       # registers inside an indexed window (and the argument views of them)
        for c in code:
            for base, ln in self._reads(c)[1]:
                in_block.update(range(base, base + ln))
        fixed = set(range(self.n_args)) | set(range(self.ret_base, self.ret_base + self.n_ret)) | in_block
        targets = {c[1] for c in code if c[0] in (F_JZ, F_JMP)}
        uses, defs = {}, {}
        for c in code:
            for r in self._reads(c)[0]:
                uses[r] = uses.get(r, 0) + 1
            if c[0] not in (F_JZ, F_JMP, F_RET, F_STX):
                defs[c[1]] = defs.get(c[1], 0) + 1
        # pass 1: fold `t = a op b; v = t` into `v = a op b`
        keep = [True] * n
        out = list(code)
        for i in range(n - 1):
            c, d = out[i], out[i + 1]
            if c[0] in (F_JZ, F_JMP, F_RET, F_STX) or d[0] != O.COPY or d[2] != ('r', c[1]):
                continue
            t = c[1]
            if t in fixed or uses.get(t, 0) != 1 or defs.get(t, 0) != 1 or (i + 1) in targets or d[1] in in_block:
                continue
            out[i] = (c[0], d[1], c[2], c[3])
            keep[i + 1] = False
        remap, k = [], 0
        for i in range(n):
            remap.append(k)
            k += keep[i]
        remap.append(k)
        code = [((c[0], remap[c[1]], c[2], c[3]) if c[0] in (F_JZ, F_JMP) else c) for c, kp in zip(out, keep) if kp]
        n = len(code)
        # pass 2: live ranges
        succ = []
        for i, c in enumerate(code):
            if c[0] == F_RET:
                succ.append(())
            elif c[0] == F_JMP:
                succ.append((c[1],))
            elif c[0] == F_JZ:
                succ.append((i + 1, c[1]))
            else:
                succ.append((i + 1,))
        use_s = [frozenset(r for r in self._reads(c)[0] if r not in fixed) for c in code]
        def_s = [None if (c[0] in (F_JZ, F_JMP, F_RET, F_STX) or c[1] in fixed) else c[1] for c in code]
        live_in = [frozenset()] * (n + 1)
        changed = True
        while changed:
            changed = False
            for i in range(n - 1, -1, -1):
                lo = frozenset().union(*[live_in[j] for j in succ[i]]) if succ[i] else frozenset()
                li = use_s[i] | (lo - {def_s[i]} if def_s[i] is not None else lo)
                if li != live_in[i]:
                    live_in[i] = li
                    changed = True
        first, last = {}, {}
        for i in range(n):
            lo = frozenset().union(*[live_in[j] for j in succ[i]]) if succ[i] else frozenset()
            here = set(live_in[i]) | set(lo)
            if def_s[i] is not None:
                here.add(def_s[i])
            for r in here:
                first.setdefault(r, i)
                last[r] = i
        # fixed registers: arguments stay, results right behind them, then every indexed window (order kept)
        new = {r: r for r in range(self.n_args)}
        nxt = self.n_args
        for r in range(self.ret_base, self.ret_base + self.n_ret):
            new[r] = nxt
            nxt += 1
        for r in sorted(in_block):
            if r not in new:
                new[r] = nxt
                nxt += 1
        import heapq
        free, active = [], []                  # free physical registers; (last position, physical register) of live ranges
        for r in sorted(first, key=lambda r: (first[r], r)):
            while active and active[0][0] < first[r]:      # a range that ended BEFORE this position frees its register
                heapq.heappush(free, heapq.heappop(active)[1])
            if free:
                p = heapq.heappop(free)
            else:
                p = nxt
                nxt += 1
            new[r] = p
            heapq.heappush(active, (last[r], p))

        def opnd(o):
            return o if o is None or o[0] != 'r' else ('r', new[o[1]])

        res = []
        for c in code:
            op = c[0]
            if op in (F_JMP, F_RET):
                res.append(c)
            elif op == F_JZ:
                res.append((op, c[1], opnd(c[2]), None))
            elif op == F_LDX:
                res.append((op, new[c[1]], new[c[2]], (new[c[3][0]], c[3][1])))
            elif op == F_STX:
                res.append((op, new[c[1]], opnd(c[2]), (new[c[3][0]], c[3][1])))
            else:
                res.append((op, new[c[1]], opnd(c[2]), opnd(c[3])))
        self.code = res
        self.n_regs = nxt
        self.ret_base = self.n_args

    # ---- builder ------------------------------------------------------------------------------------------------
    def new_reg(self):
        r = self.n_regs
        self.n_regs += 1
        return r

    def lift(self, x) -> RtVar:
        if isinstance(x, RtVar):
            return x
        if isinstance(x, int):
            return RtVar(self, 'c', x % self.fp.q)
        raise RtError("cannot use %r as a run-time value" % (x,))

    def reg_of(self, x) -> int:
        x = self.lift(x)
        if x.kind == 'r':
            return x.val
        r = self.new_reg()
        self.code.append((O.COPY, r, ('c', x.val), None))
        return r

    def var(self, init=0) -> RtVar:
        """`var x = init;`"""
        r = self.new_reg()
        v = self.lift(init)
        self.code.append((O.COPY, r, (v.kind, v.val), None))
        return RtVar(self, 'r', r)

    def array(self, n: int, init=None) -> RtArray:
        base = self.n_regs
        self.n_regs += n
        arr = RtArray(self, base, n)
        for k in range(n):
            v = self.lift(0 if init is None else init[k])
            self.code.append((O.COPY, base + k, (v.kind, v.val), None))
        return arr

    def emit(self, op, a: RtVar, b) -> RtVar:
        d = self.new_reg()
        self.code.append((op, d, (a.kind, a.val), None if b is None else (b.kind, b.val)))
        return RtVar(self, 'r', d)

    def loop(self):
        return _Loop(self)

    def if_(self, cond):
        return _If(self, cond)

    def else_(self):
        return _Else(self)

    def args_array(self, first: int, n: int) -> RtArray:
        """view n consecutive ARGUMENT registers as an array (circom passes arrays by value)"""
        if first + n > self.n_args:
            raise RtError("argument array out of range")
        return RtArray(self, first, n)

    # ---- plain-data form for the oracle / emitters ------------------------------------------------------------------
    def as_data(self):
        return {"name": self.name, "n_args": self.n_args, "n_ret": self.n_ret, "ret_base": self.ret_base, "n_regs": self.n_regs,
                "code": [list(c) for c in self.code], "native": self.native}

"""Circuit front-end: a small embedded DSL with circom's template/component/signal model.

Why this exists: the reference's Rust front-end (parser → type analysis → constraint_generation →
dag → compiler) cannot be built in this environment (no cargo), and the north-star keeps it on the
host anyway.  This module is the stand-in that feeds the new `hip_elements` back-end: a template
body written in Python is *traced once per template instance* and yields exactly what the
reference's construction phase hands to its code producers:

  * the signal table of the instance in circom's numbering — outputs, then inputs, then
    intermediates, then each sub-component's block in (name, index-vector) order
    (constraint_generation/src/execution_data/executed_template.rs:266-305,325-345),
  * the constraints in A*B - C = 0 form with the coefficient placement of
    circom_algebra/src/algebra.rs:113-145,254-440 (`<==` adds `symbol - rhs`,
    constraint_generation/src/execute.rs:460-464),
  * the witness-computation code as a straight-line list of field operations over
    instance-relative signal offsets, with a RUN marker where a sub-component fires (when its last
    input is stored: compiler/.../store_bucket.rs:660-735; at creation if it has no inputs:
    circuit_design/template.rs:274-278).

Loops and conditionals over `var`s/parameters are ordinary Python control flow and therefore
unrolled at trace time (circom forbids constraints under signal-dependent
control flow); signal-dependent choices in `<--` code use `select`.
"""
from __future__ import annotations

from bisect import bisect_right

import numpy as np

from .. import opcodes as O
from ..field import Fp, fp_for

K_SIG, K_TMP, K_CONST = O.K_SIG, O.K_TMP, O.K_CONST
CONST_KEY = -1  # key of the constant coefficient in linear forms (the reference uses the empty symbol)


class CircuitError(Exception):
    pass


# ----------------------------------------------------------------------------------------------
# symbolic algebra (ArithmeticExpression of circom_algebra/src/algebra.rs:9-34)
#   ('n', value) | ('s', pid) | ('l', {pid: coeff}) | ('q', a, b, c) | None (NonQuadratic)
# ----------------------------------------------------------------------------------------------
def _lin_of(alg):
    k = alg[0]
    if k == 'l':
        return alg[1]
    if k == 's':
        return {alg[1]: 1}
    if k == 'n':
        return {CONST_KEY: alg[1]}
    raise AssertionError


def _lin_add(x: dict, y: dict, q: int) -> dict:
    r = dict(y)
    for k, v in x.items():
        r[k] = (r.get(k, 0) + v) % q
    return r


def _lin_scale(x: dict, s: int, q: int) -> dict:
    return {k: (v * s) % q for k, v in x.items()}


def alg_add(l, r, q):
    if l is None or r is None:
        return None
    kl, kr = l[0], r[0]
    if kl == 'q' and kr == 'q':
        return None
    if kl == 'n' and kr == 'n':
        return ('n', (l[1] + r[1]) % q)
    if kl == 'q':
        return ('q', l[1], l[2], _lin_add(_lin_of(r), l[3], q))
    if kr == 'q':
        return ('q', r[1], r[2], _lin_add(_lin_of(l), r[3], q))
    return ('l', _lin_add(_lin_of(l), _lin_of(r), q))


def alg_mul(l, r, q):
    if l is None or r is None:
        return None
    kl, kr = l[0], r[0]
    if kl == 'n' and kr == 'n':
        return ('n', (l[1] * r[1]) % q)
    if kl == 'n' or kr == 'n':
        num, oth = (l, r) if kl == 'n' else (r, l)
        v = num[1]
        if oth[0] == 'q':
            return ('q', _lin_scale(oth[1], v, q), oth[2], _lin_scale(oth[3], v, q))
        return ('l', _lin_scale(_lin_of(oth), v, q))
    if kl == 'q' or kr == 'q':
        return None
    if kl == 's' and kr == 's':
        return ('q', {l[1]: 1}, {r[1]: 1}, {})
    if kl == 's' or kr == 's':           # (Signal, Linear) | (Linear, Signal): a = linear, b = signal
        sig, lin = (l, r) if kl == 's' else (r, l)
        return ('q', dict(lin[1]), {sig[1]: 1}, {})
    return ('q', dict(l[1]), dict(r[1]), {})


def alg_sub(l, r, q):                     # algebra.rs:441-450: left + (-1)*right
    return alg_add(l, alg_mul(('n', q - 1), r, q), q)


def alg_div(l, r, fp):                    # algebra.rs:452-500: only division by a Number stays quadratic
    if l is None or r is None or r[0] != 'n':
        return None
    if l[0] == 'n':
        return ('n', fp.div(l[1], r[1]))
    return alg_mul(l, ('n', fp.inv(r[1])), fp.q)


# ----------------------------------------------------------------------------------------------
# expressions
# ----------------------------------------------------------------------------------------------
class Expr:
    """A field value at trace time: where it lives at run time (kind,val) + how to get its
    symbolic form (lazily: most `<--` expressions never need one)."""
    __slots__ = ("ctx", "kind", "val", "_alg", "_src")

    def __init__(self, ctx, kind, val, alg=None, src=None):
        self.ctx = ctx
        self.kind = kind      # K_SIG (val = pid) | K_TMP (val = temp id) | K_CONST (val = canonical int)
        self.val = val
        self._alg = alg       # cached symbolic form, or False = not yet computed
        self._src = src       # (opcode, lhs, rhs) to compute it on demand

    # -- symbolic form -------------------------------------------------------------------------
    def alg(self):
        if self._alg is not False:
            return self._alg
        # iterative post-order evaluation (sums can be thousands of terms deep)
        stack = [self]
        fp = self.ctx.fp
        q = fp.q
        while stack:
            e = stack[-1]
            if e._alg is not False:
                stack.pop()
                continue
            op, l, r = e._src
            pend = [x for x in (l, r) if x is not None and x._alg is False]
            if pend:
                stack.extend(pend)
                continue
            la = l._alg
            ra = r._alg if r is not None else None
            if op == O.ADD:
                res = alg_add(la, ra, q)
            elif op == O.SUB:
                res = alg_sub(la, ra, q)
            elif op == O.MUL:
                res = alg_mul(la, ra, q)
            elif op == O.NEG:
                res = alg_mul(la, ('n', q - 1), q)
            elif op == O.DIV:
                res = alg_div(la, ra, fp)
            else:
                res = None
            e._alg = res
            e._src = None
            stack.pop()
        return self._alg

    def is_const(self):
        return self.kind == K_CONST

    # -- operators -------------------------------------------------------------------------------
    def _bin(self, op, other, swap=False):
        ctx = self.ctx
        o = ctx.lift(other)
        a, b = (o, self) if swap else (self, o)
        return ctx.emit2(op, a, b)

    def __add__(self, o): return self._bin(O.ADD, o)
    def __radd__(self, o): return self._bin(O.ADD, o, True)
    def __sub__(self, o): return self._bin(O.SUB, o)
    def __rsub__(self, o): return self._bin(O.SUB, o, True)
    def __mul__(self, o): return self._bin(O.MUL, o)
    def __rmul__(self, o): return self._bin(O.MUL, o, True)
    def __truediv__(self, o): return self._bin(O.DIV, o)
    def __rtruediv__(self, o): return self._bin(O.DIV, o, True)
    def __floordiv__(self, o): return self._bin(O.IDIV, o)      # circom `\`
    def __mod__(self, o): return self._bin(O.MOD, o)
    def __pow__(self, o): return self._bin(O.POW, o)
    def __lshift__(self, o): return self._bin(O.SHL, o)
    def __rshift__(self, o): return self._bin(O.SHR, o)
    def __and__(self, o): return self._bin(O.BAND, o)
    def __rand__(self, o): return self._bin(O.BAND, o, True)
    def __or__(self, o): return self._bin(O.BOR, o)
    def __ror__(self, o): return self._bin(O.BOR, o, True)
    def __xor__(self, o): return self._bin(O.BXOR, o)
    def __rxor__(self, o): return self._bin(O.BXOR, o, True)
    def __rlshift__(self, o): return self._bin(O.SHL, o, True)
    def __rrshift__(self, o): return self._bin(O.SHR, o, True)
    def __neg__(self): return self.ctx.emit1(O.NEG, self)
    def __invert__(self): return self.ctx.emit1(O.BNOT, self)
    # relational operators are methods (Python's rich comparisons must return bool for dict keys etc.)
    def lt(self, o): return self._bin(O.LT, o)
    def gt(self, o): return self._bin(O.GT, o)
    def leq(self, o): return self._bin(O.LEQ, o)
    def geq(self, o): return self._bin(O.GEQ, o)
    def eq(self, o): return self._bin(O.EQ, o)
    def neq(self, o): return self._bin(O.NEQ, o)
    def land(self, o): return self._bin(O.LAND, o)
    def lor(self, o): return self._bin(O.LOR, o)
    def lnot(self): return self.ctx.emit1(O.LNOT, self)


class SigArray:
    """A (possibly multi-dimensional) block of signals; row-major like circom's."""
    __slots__ = ("ctx", "base", "shape", "stride", "owner")

    def __init__(self, ctx, base, shape, owner=None):
        self.ctx = ctx
        self.base = base
        self.shape = tuple(shape)
        n = 1
        for d in self.shape[1:]:
            n *= d
        self.stride = n
        self.owner = owner      # None = own signal, else CompRef

    def __len__(self):
        return self.shape[0]

    def __getitem__(self, i):
        if not 0 <= i < self.shape[0]:
            raise CircuitError("signal index out of range")
        b = self.base + i * self.stride
        if len(self.shape) == 1:
            return Expr(self.ctx, K_SIG, b, ('s', b))
        return SigArray(self.ctx, b, self.shape[1:], self.owner)

    def __iter__(self):
        for i in range(self.shape[0]):
            yield self[i]

    @property
    def size(self):
        return self.shape[0] * self.stride


def _prod(dims):
    n = 1
    for d in dims:
        n *= d
    return n


def _freeze(p):
    if isinstance(p, (list, tuple)):
        return tuple(_freeze(x) for x in p)
    return p


class TemplateSpec:
    """`T(params)` before instantiation."""
    __slots__ = ("name", "fn", "params", "key")

    def __init__(self, name, fn, params):
        self.name = name
        self.fn = fn
        self.params = params
        self.key = (name, _freeze(params))


def template(fn):
    """Decorator: `@template def Name(c, *params)`; `Name(*params)` then denotes the template
    applied to parameter values (circom `Name(params)`)."""
    def make(*params):
        return TemplateSpec(fn.__name__, fn, params)
    make.__name__ = fn.__name__
    make.body = fn
    return make


class CompRef:
    """A sub-component of the instance being traced."""
    __slots__ = ("ctx", "k", "name", "index", "inst", "pid0", "pending", "assigned", "ran")

    def __init__(self, ctx, k, name, index, inst, pid0):
        self.ctx = ctx
        self.k = k
        self.name = name
        self.index = index
        self.inst = inst
        self.pid0 = pid0
        self.pending = inst.n_in
        self.assigned = bytearray(inst.n_in)
        self.ran = False

    def __getitem__(self, signame):
        inst = self.inst
        try:
            off, dims, cat = inst.iface[signame]
        except KeyError:
            raise CircuitError("component %s (%s) has no input/output %r" % (self.name, inst.name, signame))
        if not dims:
            return Expr(self.ctx, K_SIG, self.pid0 + off, ('s', self.pid0 + off))
        return SigArray(self.ctx, self.pid0 + off, dims, self)


class TemplateInstance:
    """A traced (template, parameters) pair: TemplateInstance of compiler/src/hir/very_concrete_program.rs
    plus the node data of dag/src/lib.rs."""

    def __init__(self, prog, spec, tid):
        self.prog = prog
        self.name = spec.name
        self.params = spec.params
        self.id = tid
        self.header = "%s_%d" % (spec.name, tid)      # executed_template.rs:420
        self.iface = {}            # signal name -> (local offset, dims, 'o'|'i')
        self.decls = {"o": [], "i": [], "m": []}   # (name, dims, pid0)
        self.n_out = self.n_in = self.n_mid = 0
        self.children = []         # sorted: (cname, index, inst, sig_off, comp_off)
        self.n_local = 0
        self.n_total = 0
        self.n_components = 1
        self.n_temps = 0
        self.code = None           # dict of numpy arrays
        self.constraints = []      # (A, B, C) dicts over local offsets, CONST_KEY for the constant
        self.input_names = []      # for main: (name, local offset, size)


class Ctx:
    """Trace context of one template instance (the `c` argument of a template body)."""

    def __init__(self, prog, inst):
        self.prog = prog
        self.fp: Fp = prog.fp
        self.inst = inst
        self.npid = 0
        self.own = []              # (cat, name, dims, pid0, size)
        self.comps = []            # CompRef in creation order
        self.ntmp = 0
        # code columns
        self.c_op, self.c_dk, self.c_dv = [], [], []
        self.c_ak, self.c_av, self.c_bk, self.c_bv, self.c_ck, self.c_cv = [], [], [], [], [], []
        self.cons = []             # (A,B,C) over pids
        self._names = set()
        self._last_tmp_row = {}    # temp id -> code row that defines it (for store retargeting)
        self._tmp_alias = {}       # temp id -> pid of the signal its defining op now writes
        self._assigned_own = set()
        self._comp_pid0 = []       # pid0 of each component block (increasing), for bisect

    # ---- declarations ---------------------------------------------------------------------------
    def _declare(self, cat, name, dims):
        if name in self._names:
            raise CircuitError("symbol %r declared twice" % name)
        self._names.add(name)
        dims = tuple(int(d) for d in dims)
        size = _prod(dims)
        pid0 = self.npid
        self.npid += size
        self.own.append((cat, name, dims, pid0, size))
        if not dims:
            return Expr(self, K_SIG, pid0, ('s', pid0))
        return SigArray(self, pid0, dims)

    def input(self, name, *dims):
        return self._declare("i", name, dims)

    def output(self, name, *dims):
        return self._declare("o", name, dims)

    def signal(self, name, *dims):
        return self._declare("m", name, dims)

    def component(self, name, spec: TemplateSpec, index=()):
        """`component name[index] = T(params);`"""
        if not isinstance(index, tuple):
            index = (index,)
        inst = self.prog.instantiate(spec)
        ref = CompRef(self, len(self.comps), name, index, inst, self.npid)
        self._comp_pid0.append(self.npid)
        self.npid += inst.n_total
        self.comps.append(ref)
        if inst.n_in == 0:
            self._run(ref)
        return ref

    # ---- values ---------------------------------------------------------------------------------
    def const(self, v: int) -> Expr:
        v %= self.fp.q
        return Expr(self, K_CONST, v, ('n', v))

    def lift(self, x) -> Expr:
        if isinstance(x, Expr):
            return x
        if isinstance(x, int):
            return self.const(x)
        raise CircuitError("cannot use %r as a field value" % (x,))

    def _row(self, op, dk, dv, a, b=None, c=None):
        self.c_op.append(op)
        self.c_dk.append(dk); self.c_dv.append(dv)
        self.c_ak.append(a.kind); self.c_av.append(a.val)
        if b is None:
            self.c_bk.append(O.K_NONE); self.c_bv.append(0)
        else:
            self.c_bk.append(b.kind); self.c_bv.append(b.val)
        if c is None:
            self.c_ck.append(O.K_NONE); self.c_cv.append(0)
        else:
            self.c_ck.append(c.kind); self.c_cv.append(c.val)
        return len(self.c_op) - 1

    def _newtmp(self, op, a, b, sym):
        t = self.ntmp
        self.ntmp += 1
        row = self._row(op, K_TMP, t, a, b)
        self._last_tmp_row[t] = row
        if sym:
            return Expr(self, K_TMP, t, False, (op, a, b))
        return Expr(self, K_TMP, t, None)

    _FOLD = {O.ADD: "add", O.SUB: "sub", O.MUL: "mul", O.DIV: "div", O.IDIV: "idiv", O.MOD: "mod",
             O.POW: "pow", O.SHL: "shl", O.SHR: "shr", O.BAND: "band", O.BOR: "bor", O.BXOR: "bxor",
             O.LT: "lt", O.GT: "gt", O.LEQ: "leq", O.GEQ: "geq", O.EQ: "eq", O.NEQ: "neq",
             O.LAND: "land", O.LOR: "lor"}
    _SYM = (O.ADD, O.SUB, O.MUL, O.DIV)

    def emit2(self, op, a: Expr, b: Expr) -> Expr:
        if a.kind == K_CONST and b.kind == K_CONST:
            return self.const(getattr(self.fp, self._FOLD[op])(a.val, b.val))
        # identities that leave the value unchanged (no run-time op needed)
        if op == O.ADD:
            if a.kind == K_CONST and a.val == 0:
                return b
            if b.kind == K_CONST and b.val == 0:
                return a
        elif op == O.SUB:
            if b.kind == K_CONST and b.val == 0:
                return a
        elif op == O.MUL:
            if a.kind == K_CONST:
                if a.val == 1:
                    return b
                if a.val == 0:
                    return self.const(0)
            if b.kind == K_CONST:
                if b.val == 1:
                    return a
                if b.val == 0:
                    return self.const(0)
        return self._newtmp(op, a, b, op in self._SYM)

    def emit1(self, op, a: Expr) -> Expr:
        if a.kind == K_CONST:
            fp = self.fp
            v = {O.NEG: fp.neg, O.BNOT: fp.bnot, O.LNOT: fp.lnot}[op](a.val)
            return self.const(v)
        return self._newtmp(op, a, None, op == O.NEG)

    def select(self, cond, a, b) -> Expr:
        """`cond ? a : b` on a signal-dependent condition inside `<--` code (BranchBucket →
        predicated select; both sides are evaluated)."""
        cond, a, b = self.lift(cond), self.lift(a), self.lift(b)
        if cond.kind == K_CONST:
            return a if cond.val != 0 else b
        t = self.ntmp
        self.ntmp += 1
        self._row(O.SELECT, K_TMP, t, cond, a, b)
        return Expr(self, K_TMP, t, None)

    # ---- statements -----------------------------------------------------------------------------
    def _store(self, dst: Expr, src: Expr):
        if dst.kind != K_SIG:
            raise CircuitError("assignment target must be a signal")
        pid = dst.val
        row = self._last_tmp_row.pop(src.val, None) if src.kind == K_TMP else None
        if row is not None and self.c_dk[row] == K_TMP and self.c_dv[row] == src.val:
            # retarget the producing op to write the signal directly; later reads of the
            # temporary now read the signal (same value)
            self.c_dk[row] = K_SIG
            self.c_dv[row] = pid
            self._tmp_alias[src.val] = pid
            src.kind, src.val = K_SIG, pid
        else:
            self._row(O.COPY, K_SIG, pid, src)
        self._after_store(pid)

    def _after_store(self, pid):
        # own signal or a sub-component input?
        j = bisect_right(self._comp_pid0, pid) - 1
        if j >= 0:
            ref = self.comps[j]
            if pid < ref.pid0 + ref.inst.n_total:
                off = pid - ref.pid0
                lo = ref.inst.n_out
                if not lo <= off < lo + ref.inst.n_in:
                    raise CircuitError("only inputs of sub-component %s can be assigned" % ref.name)
                i = off - lo
                if ref.assigned[i]:
                    raise CircuitError("signal assigned twice: %s input %d" % (ref.name, i))
                ref.assigned[i] = 1
                ref.pending -= 1
                if ref.pending == 0:
                    self._run(ref)
                return
        if pid in self._assigned_own:
            raise CircuitError("signal assigned twice (pid %d)" % pid)
        self._assigned_own.add(pid)

    def _run(self, ref: CompRef):
        ref.ran = True
        self.c_op.append(O.RUN)
        self.c_dk.append(O.K_NONE); self.c_dv.append(0)
        self.c_ak.append(O.K_NONE); self.c_av.append(ref.k)
        self.c_bk.append(O.K_NONE); self.c_bv.append(0)
        self.c_ck.append(O.K_NONE); self.c_cv.append(0)

    def _constraint_from(self, e_alg):
        """transform_expression_to_constraint_form, algebra.rs:113-145."""
        if e_alg is None:
            raise CircuitError("Non quadratic constraints are not allowed!")
        q = self.fp.q
        k = e_alg[0]
        if k == 'q':
            a, b, c = e_alg[1], e_alg[2], e_alg[3]
        else:
            a, b, c = {}, {}, _lin_of(e_alg)
        c = _lin_scale(c, q - 1, q)
        self.cons.append((a, b, c))

    def set(self, dst, value):
        """`dst <== value`  (store + constraint `dst - value`)."""
        value = self.lift(value)
        dalg = ('s', dst.val)
        valg = value.alg()          # before the store may retarget `value`
        self._store(dst, value)
        self._constraint_from(alg_sub(dalg, valg, self.fp.q))

    def hint(self, dst, value):
        """`dst <-- value`  (store only)."""
        self._store(dst, self.lift(value))

    def enforce(self, lhs, rhs=0, runtime_check=True):
        """`lhs === rhs`  (constraint + run-time assert, assert_bucket.rs:70-89).
        runtime_check=False emits only the constraint (what `--sanity_check 0` does, assert_bucket.rs:73);
        tests use it to produce witnesses that violate the R1CS without tripping an assert."""
        lhs, rhs = self.lift(lhs), self.lift(rhs)
        self._constraint_from(alg_sub(lhs.alg(), rhs.alg(), self.fp.q))
        if not runtime_check:
            return
        if lhs.kind == K_CONST and rhs.kind == K_CONST:
            if lhs.val != rhs.val:
                raise CircuitError("constraint between constants does not hold")
            return
        self._row(O.ASSERT_EQ, O.K_NONE, 0, lhs, rhs)

    def log(self, *args):
        """`log(arg, ...)`: strings and expressions (LogBucket, log_bucket.rs:105-162: the reference prints every argument
        with printf - values through Fr_element2str, i.e. the canonical residue in decimal - separated by blanks, then a
        newline).  One LOG row per argument; the row that ends the statement carries dv = 1.  A batched run has no console
        per instance: the lowering keeps every logged value in the table and `cw_get_log` formats the text the reference
        binary prints for one instance."""
        items = list(args) or [None]
        for k, a in enumerate(items):
            last = 1 if k == len(items) - 1 else 0
            if a is None or isinstance(a, str):
                sid = -1 if a is None else self.prog.log_string_id(a)
                self._row(O.LOG, O.K_NONE, last, Expr(self, O.K_NONE, sid, None))
            else:
                self._row(O.LOG, O.K_NONE, last, self.lift(a))

    def assert_(self, cond):
        """`assert(cond)`."""
        cond = self.lift(cond)
        if cond.kind == K_CONST:
            if cond.val == 0:
                raise CircuitError("assert(false) on constants")
            return
        self._row(O.ASSERT_NZ, O.K_NONE, 0, cond)

    # ---- circom functions with run-time control flow (tier 2, frontend/rtcode.py) ---------------------------------
    def function(self, name: str, n_args: int, build, native=None):
        """`function name(...) {...}`: built once per program (build(f, *args) -> results, see rtcode.RtFunction).
        native = (kind, n, k, modulus): the function is a pure big-integer routine with this closed form
        (circuits/bigint_func.py native_eval): oracle and device may compute it directly instead of interpreting the body"""
        return self.prog.function(name, n_args, build, native)

    def call(self, fn, args):
        """`name(args)` inside `<--` code: CallBucket (call_bucket.rs:466-533).  The function's registers are a block of
        consecutive temporaries of this instance: arguments are copied into the first ones, results are read back from
        the registers the function reserves for them.  Returns the list of results."""
        args = [self.lift(a) for a in args]
        if len(args) != fn.n_args:
            raise CircuitError("function %s takes %d arguments" % (fn.name, fn.n_args))
        t0 = self.ntmp
        self.ntmp += fn.n_regs
        for k, a in enumerate(args):
            self._row(O.COPY, K_TMP, t0 + k, a)
        self.c_op.append(O.CALL)
        self.c_dk.append(O.K_NONE); self.c_dv.append(0)
        self.c_ak.append(O.K_NONE); self.c_av.append(fn.id)
        self.c_bk.append(K_TMP); self.c_bv.append(t0)
        self.c_ck.append(O.K_NONE); self.c_cv.append(0)
        return [Expr(self, K_TMP, t0 + fn.ret_base + k, None) for k in range(fn.n_ret)]

    # ---- finalisation -----------------------------------------------------------------------------
    def finalize(self):
        inst = self.inst
        for ref in self.comps:
            if not ref.ran:
                raise CircuitError("component %s of %s never received all its inputs (%d missing)"
                                   % (ref.name, inst.name, ref.pending))
        perm = np.full(self.npid, -1, dtype=np.int64)
        off = 0
        for cat in ("o", "i", "m"):
            for (c, name, dims, pid0, size) in self.own:
                if c != cat:
                    continue
                perm[pid0:pid0 + size] = np.arange(off, off + size)
                inst.decls[cat].append((name, dims, off))
                if cat != "m":
                    inst.iface[name] = (off, dims, cat)
                if cat == "i":
                    inst.input_names.append((name, off, size))
                off += size
            if cat == "o":
                inst.n_out = off
            elif cat == "i":
                inst.n_in = off - inst.n_out
        inst.n_local = off
        inst.n_mid = off - inst.n_out - inst.n_in
        # children in (name, index) order — executed_template.rs:326-335
        order = sorted(range(len(self.comps)), key=lambda k: (self.comps[k].name, self.comps[k].index))
        k2sorted = {}
        coff = 1
        for pos, k in enumerate(order):
            ref = self.comps[k]
            n = ref.inst.n_total
            perm[ref.pid0:ref.pid0 + n] = np.arange(off, off + n)
            inst.children.append((ref.name, ref.index, ref.inst, off, coff))
            k2sorted[k] = pos
            off += n
            coff += ref.inst.n_components
        inst.n_total = off
        inst.n_components = coff
        inst.n_temps = self.ntmp
        assert (perm >= 0).all()

        alias = self._tmp_alias

        def col(kinds, vals):
            if alias:
                kinds = list(kinds)
                vals = list(vals)
                for i, (kk, vv) in enumerate(zip(kinds, vals)):
                    if kk == K_TMP and vv in alias:
                        kinds[i] = K_SIG
                        vals[i] = alias[vv]
            k = np.asarray(kinds, dtype=np.uint8)
            is_sig = k == K_SIG
            is_const = k == K_CONST
            if is_const.any():
                # intern constants in the program-wide table (constant_tracking/src/lib.rs: first use order)
                cv = [self.prog.const_id(v) if c else 0 for v, c in zip(vals, is_const)]
                vals = [c if ic else v for v, c, ic in zip(vals, cv, is_const)]
            v = np.asarray(vals, dtype=np.int64)
            if is_sig.any():
                v = np.where(is_sig, perm[np.where(is_sig, v, 0)], v)
            return k, v

        op = np.asarray(self.c_op, dtype=np.uint8)
        dk, dv = col(self.c_dk, self.c_dv)
        ak, av = col(self.c_ak, self.c_av)
        bk, bv = col(self.c_bk, self.c_bv)
        ck, cv = col(self.c_ck, self.c_cv)
        run = op == O.RUN
        if run.any():
            av = av.copy()
            av[run] = [k2sorted[int(k)] for k in av[run]]
        inst.code = dict(op=op, dk=dk, dv=dv, ak=ak, av=av, bk=bk, bv=bv, ck=ck, cv=cv)

        def relin(d):
            return {(CONST_KEY if k == CONST_KEY else int(perm[k])): v for k, v in d.items() if v}
        inst.constraints = [(relin(a), relin(b), relin(c)) for a, b, c in self.cons]


class Program:
    """A whole circuit: `component main {public [...]} = T(params);` over a prime."""

    def __init__(self, main: TemplateSpec, public=(), prime="bn128"):
        self.prime = prime
        self.fp = fp_for(prime)
        self.instances = {}
        self.inst_list = []
        self.constants = []
        self._const_ids = {}
        self.public = tuple(public)
        self.functions = []            # rtcode.RtFunction, in order of first use
        self.functions_by_name = {}
        self.main = self.instantiate(main)
        for name in self.public:
            if name not in self.main.iface or self.main.iface[name][2] != "i":
                raise CircuitError("public signal %r is not an input of main" % name)
        if self.public:
            self._reorder_main_public()

    def log_string_id(self, text: str) -> int:
        """string table of the log statements (the producer's string table, c_elements/mod.rs get_string_table)"""
        if not hasattr(self, "log_strings"):
            self.log_strings, self._log_string_ids = [], {}
        i = self._log_string_ids.get(text)
        if i is None:
            i = len(self.log_strings)
            self._log_string_ids[text] = i
            self.log_strings.append(text)
        return i

    def const_id(self, v: int) -> int:
        i = self._const_ids.get(v)
        if i is None:
            i = len(self.constants)
            self._const_ids[v] = i
            self.constants.append(v)
        return i

    def function(self, name: str, n_args: int, build, native=None):
        fn = self.functions_by_name.get(name)
        if fn is None:
            from .rtcode import RtFunction
            fn = RtFunction(name, n_args, build, self.fp)
            fn.native = native
            # constants of the body live in the program-wide constant table like every other constant
            fn.code = [tuple((x[0], self.const_id(x[1])) if isinstance(x, tuple) and len(x) == 2 and x[0] == 'c' else x
                             for x in ins) for ins in fn.code]
            fn.id = len(self.functions)
            self.functions.append(fn)
            self.functions_by_name[name] = fn
        elif fn.n_args != n_args:
            raise CircuitError("function %s redefined with another arity" % name)
        return fn

    def instantiate(self, spec: TemplateSpec) -> TemplateInstance:
        inst = self.instances.get(spec.key)
        if inst is not None:
            return inst
        inst = TemplateInstance(self, spec, -1)
        ctx = Ctx(self, inst)
        spec.fn(ctx, *spec.params)
        ctx.finalize()
        inst.id = len(self.inst_list)          # ids in order of completed instantiation
        inst.header = "%s_%d" % (inst.name, inst.id)
        self.inst_list.append(inst)
        self.instances[spec.key] = inst
        return inst

    def _reorder_main_public(self):
        """Main's public inputs are numbered before its private ones (executed_template.rs:277-299).
        Declaration order is kept inside each class."""
        m = self.main
        ins = m.decls["i"]
        new_order = [d for d in ins if d[0] in self.public] + [d for d in ins if d[0] not in self.public]
        if [d[0] for d in new_order] == [d[0] for d in ins]:
            return
        remap = np.arange(m.n_total, dtype=np.int64)
        off = m.n_out
        new_decls = []
        for name, dims, old in new_order:
            size = _prod(dims)
            remap[old:old + size] = np.arange(off, off + size)
            new_decls.append((name, dims, off))
            off += size
        m.decls["i"] = new_decls
        m.input_names = [(n, o, _prod(d)) for n, d, o in new_decls]
        for n, d, o in new_decls:
            m.iface[n] = (o, d, "i")
        code = m.code
        for kk, vv in (("dk", "dv"), ("ak", "av"), ("bk", "bv"), ("ck", "cv")):
            sig = code[kk] == K_SIG
            code[vv] = np.where(sig, remap[np.where(sig, code[vv], 0)], code[vv])
        rl = lambda d: {(k if k == CONST_KEY else int(remap[k])): v for k, v in d.items()}
        m.constraints = [(rl(a), rl(b), rl(c)) for a, b, c in m.constraints]

    @property
    def n_public_inputs(self):
        return sum(_prod(d) for n, d, o in self.main.decls["i"] if n in self.public)

"""Executor of parsed `.circom` programs: walks the AST of frontend/circom_lang.py and drives the tracing front-end
(frontend/dsl.py) - so a circuit written in circom's own language reaches the same flat circuit, `.r1cs`, `.dat`, `.sym`
and lowered device program as one authored against the Python eDSL.

What it restates: the reference's construction phase, constraint_generation/src/execute.rs (execute_statement :233,
substitutions :380-480, declarations :250-330, conditionals / loops on KNOWN conditions unrolled while they run,
execute_expression, template / function calls with their own environments), with the value domain of
circom_algebra (known field elements fold through modular_arithmetic.rs = circom_amd/field.py; anything that
depends on a signal is an expression traced by dsl.Ctx).  The split the reference makes between "known at compile time"
and "unknown" is the split between Python ints and dsl.Expr here.

Run-time control flow inside `<--` code:
  * `c ? a : b` on an unknown condition is a predicated select (both sides evaluated), as in dsl.Ctx.select;
  * `if (unknown) {...} else {...}` around `var` assignments and `<--` stores is if-converted the same way (the reference
    emits a BranchBucket, branch_bucket.rs:100-122): both arms run on their own copy of the variables, what differs is merged
    through selects.  Constraints, component creation and signal declarations under an unknown condition are errors, as in
    the reference (`constraint_generation` reports them);
  * a FUNCTION called with unknown arguments is first run on an abstract domain (known / unknown only).  If its loops and
    array indices stay known it is traced inline - its operations become rows of the calling component, which is what a
    schedule wants; if a `while` condition, an array index or a `return` depends on an unknown value, the function is
    compiled to tier-2 bytecode (frontend/circom_rt.py -> rtcode.RtFunction) and called (CallBucket, call_bucket.rs:466-533).

Tags (mkdocs circom-language/tags.md): a signal assigned from a tagged signal inherits its tags and their values; an input of
a sub-component declared with tags only accepts a signal that carries them; a valued tag is frozen once its signal has a
value; a parent reads `component.out.tag`.  A tag VALUE does not flow INTO a component (its body is traced once per
parameter set, before the parent assigns its inputs).

Not supported (each raises CircuitError with the source position, never a silent difference): custom templates / `extern_c`
(the device-side slot is a native body); `while` on an unknown condition in a TEMPLATE body; a tag VALUE read
inside the component it flows into (`signal input {maxbit} x; var m = x.maxbit;` - "tag maxbit has no value": the reference
re-executes the template per tag-value set); functions that recurse on run-time values ("recursion on run-time values is not
supported"), and in the abstract pre-run a recursion under an unknown condition ends in "function calls nested too deeply"
instead of becoming a run-time function; a division by a KNOWN zero inside an untaken run-time branch aborts compilation
(the reference only fails if the branch runs).  The front-end is frozen at this state (SURVEY §2 rows 1/2/4 are out of scope).
"""
from __future__ import annotations

import sys

from .. import opcodes as O
from ..field import fp_for
from . import dsl
from .dsl import CircuitError, Expr, SigArray, TemplateSpec
from .circom_lang import Archive, parse_program, parse_text


class _Unknown:
    """the abstract domain's only non-constant value"""
    __slots__ = ()

    def __repr__(self):
        return "UNK"


UNK = _Unknown()

_BIN = {"+": ("add", O.ADD), "-": ("sub", O.SUB), "*": ("mul", O.MUL), "/": ("div", O.DIV), "\\": ("idiv", O.IDIV),
        "%": ("mod", O.MOD), "**": ("pow", O.POW), "<<": ("shl", O.SHL), ">>": ("shr", O.SHR), "&": ("band", O.BAND),
        "|": ("bor", O.BOR), "^": ("bxor", O.BXOR), "<": ("lt", O.LT), ">": ("gt", O.GT), "<=": ("leq", O.LEQ),
        ">=": ("geq", O.GEQ), "==": ("eq", O.EQ), "!=": ("neq", O.NEQ), "&&": ("land", O.LAND), "||": ("lor", O.LOR)}
_UN = {"-": ("neg", O.NEG), "!": ("lnot", O.LNOT), "~": ("bnot", O.BNOT)}


class _Return(Exception):
    def __init__(self, value):
        self.value = value


class NeedsRuntime(Exception):
    """raised by the abstract run of a function: its control flow depends on run-time values"""


# ---- environment entries ----------------------------------------------------------------------------------------------------
class VarSlot:
    __slots__ = ("value",)

    def __init__(self, value):
        self.value = value


class SigSlot:
    __slots__ = ("obj", "kind", "tags", "dims")

    def __init__(self, obj, kind, tags, dims):
        self.obj, self.kind, self.dims = obj, kind, dims
        self.tags = {t: None for t in tags}


class CompSlot:
    __slots__ = ("name", "dims", "refs")

    def __init__(self, name, dims):
        self.name, self.dims = name, dims
        self.refs = {}


class BusSlot:
    """a bus-typed signal (or array of them): fields resolved against the flattened block of signals"""
    __slots__ = ("layout", "dims", "obj", "kind", "tags")

    def __init__(self, layout, dims, obj, kind, tags):
        self.layout, self.dims, self.obj, self.kind = layout, dims, obj, kind
        self.tags = {t: None for t in tags}


class BusLayout:
    """fields of one bus instance in declaration order: name -> (offset, dims, sub-layout | None); size in signals"""

    def __init__(self, name):
        self.name = name
        self.fields = {}
        self.order = []
        self.size = 0


class TemplateCall:
    __slots__ = ("spec", "parallel")

    def __init__(self, spec, parallel=False):
        self.spec, self.parallel = spec, parallel


class BusView:
    """a bus value (or an element of a bus array) at a base signal of the traced instance"""
    __slots__ = ("ctx", "layout", "base", "owner")

    def __init__(self, ctx, layout, base, owner=None):
        self.ctx, self.layout, self.base, self.owner = ctx, layout, base, owner


def _deep_copy(v):
    if isinstance(v, list):
        return [_deep_copy(x) for x in v]
    return v


def _shape(v):
    s = []
    while isinstance(v, list):
        s.append(len(v))
        v = v[0] if v else None
    return tuple(s)


def _flat(v, out):
    if isinstance(v, list):
        for x in v:
            _flat(x, out)
    else:
        out.append(v)
    return out


def _zeros(dims):
    if not dims:
        return 0
    return [_zeros(dims[1:]) for _ in range(dims[0])]


def _freeze(v):
    if isinstance(v, list):
        return tuple(_freeze(x) for x in v)
    return v


def _thaw(v):
    if isinstance(v, tuple):
        return [_thaw(x) for x in v]
    return v


def _all_known(v):
    if isinstance(v, list):
        return all(_all_known(x) for x in v)
    return isinstance(v, int)


class Executor:
    """one template body (mode 'trace', with a dsl.Ctx) or one function body (mode 'const': every value known;
    'abstract': known / UNK; 'trace': inlined into the calling component)"""

    def __init__(self, world, mode, ctx=None, depth=0):
        self.w = world
        self.ar = world.archive
        self.fp = world.fp
        self.q = world.fp.q
        self.mode = mode
        self.ctx = ctx
        self.scopes = [{}]
        self.depth = depth
        self.in_function = False
        self.cond_stack = []       # if-conversion frames: dict(hints={pid: (dst, value)})
        self.loop_counts = {}      # while statement position -> completed iterations (anonymous component indices)
        self.loop_stack = []
        self.assigned_tagsets = set()     # id() of the tag tables of signals that already received a value
        self.underscored = set()           # pids named in `_ <== ...` (not reported by --inspect)

    # ---- errors ---------------------------------------------------------------------------------------------------------
    def fail(self, msg, pos):
        fn, ln, col = self.ar.where(pos)
        raise CircuitError("%s:%d:%d: %s" % (fn, ln, col, msg))

    # ---- scopes ---------------------------------------------------------------------------------------------------------
    def lookup(self, name, pos):
        for sc in reversed(self.scopes):
            s = sc.get(name)
            if s is not None:
                return s
        self.fail("undeclared symbol %s" % name, pos)

    def declare(self, name, slot, pos):
        for sc in self.scopes:
            if name in sc:
                # the reference rejects shadowing inside one template / function (type analysis: "symbol declared twice")
                self.fail("symbol %s declared twice" % name, pos)
        self.scopes[-1][name] = slot

    # ---- values ---------------------------------------------------------------------------------------------------------
    def known(self, v):
        return isinstance(v, int)

    def binop(self, op, a, b, pos):
        if isinstance(a, list) or isinstance(b, list):
            self.fail("operator %s on arrays" % op, pos)
        name, code = _BIN[op]
        if isinstance(a, int) and isinstance(b, int):
            if op in ("/", "\\", "%") and b == 0:
                self.fail("division by zero", pos)               # ArithmeticError::DivisionByZero, modular_arithmetic.rs:41-62
            return getattr(self.fp, name)(a, b)
        if self.mode == "abstract":
            return UNK
        if self.mode == "const":
            self.fail("value not known at compile time", pos)
        return self.ctx.emit2(code, self.ctx.lift(a), self.ctx.lift(b))

    def unop(self, op, a, pos):
        if isinstance(a, list):
            self.fail("operator %s on an array" % op, pos)
        name, code = _UN[op]
        if isinstance(a, int):
            return getattr(self.fp, name)(a)
        if self.mode == "abstract":
            return UNK
        return self.ctx.emit1(code, a)

    def select(self, c, a, b, pos):
        """c unknown: elementwise select over values of equal shape"""
        if isinstance(a, list) or isinstance(b, list):
            if _shape(a) != _shape(b):
                self.fail("the two sides of a run-time choice have different sizes", pos)
            return [self.select(c, x, y, pos) for x, y in zip(a, b)]
        if isinstance(a, int) and isinstance(b, int) and a == b:
            return a
        if a is b:
            return a
        if self.mode == "abstract":
            return UNK
        return self.ctx.select(c, a, b)

    def as_index(self, v, pos, what="array index"):
        if isinstance(v, int):
            if v >= 1 << 31:
                self.fail("%s out of bounds" % what, pos)
            return v
        if self.mode == "abstract":
            raise NeedsRuntime()
        self.fail("%s is not known at compile time" % what, pos)

    # ---- expressions ------------------------------------------------------------------------------------------------------
    def eval(self, e):
        k = e[0]
        if k == "num":
            return e[1] % self.q
        if k == "var":
            return self.read(e)
        if k == "bin":
            # a long sum is a LEFT-deep tree (every infix tier is left associative): walked with a loop, not by recursion
            chain = []
            node = e
            while node[0] == "bin":
                chain.append(node)
                node = node[2]
            acc = self.eval(node)
            for nd in reversed(chain):
                acc = self.binop(nd[1], acc, self.eval(nd[3]), nd[-1])
            return acc
        if k == "un":
            return self.unop(e[1], self.eval(e[2]), e[-1])
        if k == "tern":
            c = self.eval(e[1])
            if isinstance(c, list):
                self.fail("condition is an array", e[-1])
            if isinstance(c, int):
                return self.eval(e[2]) if c != 0 else self.eval(e[3])
            a = self.eval(e[2])
            b = self.eval(e[3])
            return self.select(c, a, b, e[-1])
        if k == "call":
            return self.call(e)
        if k == "arr":
            vals = [self.eval(x) for x in e[1]]
            s0 = _shape(vals[0])
            for v in vals[1:]:
                if _shape(v) != s0:
                    self.fail("inline array with elements of different sizes", e[-1])
            return vals
        if k == "parallel":
            v = self.eval(e[1])
            if not isinstance(v, TemplateCall):
                self.fail("parallel applies to a template call", e[-1])
            v.parallel = True
            return v
        if k == "anon":
            return self.anonymous(e)
        if k == "tuple":
            return ("tuple", [self.eval(x) for x in e[1]])
        self.fail("unexpected expression", e[-1])

    def sig_value(self, obj):
        if isinstance(obj, SigArray):
            return [self.sig_value(obj[i]) for i in range(len(obj))]
        return obj

    def bus_value(self, view: BusView):
        """every signal of a bus in field order (nested lists): what whole-bus assignments move"""
        out = []
        for fname in view.layout.order:
            off, dims, sub = view.layout.fields[fname]
            out.append(self._field_value(view, off, dims, sub))
        return out

    def _field_value(self, view, off, dims, sub):
        ctx = view.ctx
        if sub is None:
            if not dims:
                return Expr(ctx, O.K_SIG, view.base + off, ('s', view.base + off))
            return self.sig_value(SigArray(ctx, view.base + off, dims, view.owner))

        def rec(base, d):
            if not d:
                return self.bus_value(BusView(ctx, sub, base, view.owner))
            stride = sub.size
            for x in d[1:]:
                stride *= x
            return [rec(base + i * stride, d[1:]) for i in range(d[0])]
        return rec(view.base + off, dims)

    def read(self, e):
        name, access, pos = e[1], e[2], e[-1]
        if name == "_":
            self.fail("_ cannot be read", pos)
        slot = self.lookup(name, pos)
        if isinstance(slot, VarSlot):
            v = slot.value
            for a in access:
                if a[0] != "idx":
                    self.fail("a variable has no field %s" % a[1], pos)
                if not isinstance(v, list):
                    self.fail("too many indices for %s" % name, pos)
                i = self.as_index(self.eval(a[1]), pos)
                if i >= len(v):
                    self.fail("array index out of bounds: %s[%d] of %d" % (name, i, len(v)), pos)
                v = v[i]
            return v
        r = self.resolve_signal(slot, name, access, pos, for_write=False)
        if r[0] == "tag":
            v = r[1].get(r[2])
            if v is None:
                self.fail("tag %s has no value" % r[2], pos)
            return v
        if r[0] == "bus":
            return self.bus_value(r[1])
        if r[0] == "comp":
            self.fail("a component is not a value", pos)
        return self.sig_value(r[1])

    def _index_sig(self, obj, idxs, name, pos):
        for i in idxs:
            if not isinstance(obj, SigArray):
                self.fail("too many indices for signal %s" % name, pos)
            if i >= len(obj):
                self.fail("signal index out of bounds: %s[%d] of %d" % (name, i, len(obj)), pos)
            obj = obj[i]
        return obj

    def _walk_bus(self, view_or_arr, dims, layout, access, k, name, pos, tags):
        """continue an access path inside a bus: (ctx, base, owner) at `dims` of `layout`"""
        ctx, base, owner = view_or_arr
        # indices of the bus array first
        d = list(dims)
        while d and k < len(access) and access[k][0] == "idx":
            i = self.as_index(self.eval(access[k][1]), pos)
            if i >= d[0]:
                self.fail("bus index out of bounds in %s" % name, pos)
            stride = layout.size
            for x in d[1:]:
                stride *= x
            base += i * stride
            d.pop(0)
            k += 1
        if k == len(access):
            if d:
                # an array of buses as a value
                def rec(b, dd):
                    if not dd:
                        return self.bus_value(BusView(ctx, layout, b, owner))
                    stride = layout.size
                    for x in dd[1:]:
                        stride *= x
                    return [rec(b + i * stride, dd[1:]) for i in range(dd[0])]
                return ("sigval", rec(base, d))
            return ("bus", BusView(ctx, layout, base, owner))
        if d:
            self.fail("field access on an array of buses in %s" % name, pos)
        a = access[k]
        if a[0] != "field":
            self.fail("too many indices for bus %s" % name, pos)
        f = layout.fields.get(a[1])
        if f is None:
            if k == len(access) - 1 and a[1] in tags:
                return ("tag", tags, a[1])
            self.fail("bus %s has no field %s" % (layout.name, a[1]), pos)
        off, fdims, sub = f
        k += 1
        if sub is not None:
            return self._walk_bus((ctx, base + off, owner), fdims, sub, access, k, name, pos, {})
        obj = Expr(ctx, O.K_SIG, base + off, ('s', base + off)) if not fdims else SigArray(ctx, base + off, fdims, owner)
        idxs = []
        while k < len(access) and access[k][0] == "idx":
            idxs.append(self.as_index(self.eval(access[k][1]), pos))
            k += 1
        if k != len(access):
            self.fail("unexpected field access in %s" % name, pos)
        return ("sig", self._index_sig(obj, idxs, name, pos))

    def resolve_signal(self, slot, name, access, pos, for_write):
        """-> ('sig', Expr | SigArray) | ('tag', dict, tag) | ('comp', CompSlot, index tuple) | ('bus', BusView)
        | ('sigval', nested value)"""
        if isinstance(slot, SigSlot):
            idxs = []
            k = 0
            while k < len(access) and access[k][0] == "idx":
                idxs.append(self.as_index(self.eval(access[k][1]), pos))
                k += 1
            if k < len(access):
                if k != len(access) - 1 or access[k][1] not in slot.tags:
                    self.fail("signal %s has no tag %s" % (name, access[k][1]), pos)
                return ("tag", slot.tags, access[k][1])
            return ("sig", self._index_sig(slot.obj, idxs, name, pos))
        if isinstance(slot, BusSlot):
            return self._walk_bus((self.ctx, slot.obj, None), slot.dims, slot.layout, access, 0, name, pos, slot.tags)
        if isinstance(slot, CompSlot):
            idxs = []
            k = 0
            while k < len(access) and access[k][0] == "idx" and len(idxs) < len(slot.dims):
                idxs.append(self.as_index(self.eval(access[k][1]), pos))
                k += 1
            if len(idxs) != len(slot.dims):
                if k == len(access):
                    self.fail("an array of components is not a value", pos)
                self.fail("component array %s needs %d indices" % (name, len(slot.dims)), pos)
            for i, d in zip(idxs, slot.dims):
                if i >= d:
                    self.fail("component index out of bounds: %s" % name, pos)
            idx = tuple(idxs)
            if k == len(access):
                return ("comp", slot, idx)
            ref = slot.refs.get(idx)
            if ref is None:
                self.fail("component %s%s is used before it is instantiated" % (name, "".join("[%d]" % i for i in idx)), pos)
            a = access[k]
            if a[0] != "field":
                self.fail("too many indices for component %s" % name, pos)
            k += 1
            binfo = ref.inst.bus_iface.get(a[1]) if hasattr(ref.inst, "bus_iface") else None
            if not for_write and not ref.ran:
                # execute.rs:3973 / 4115: an output (signal, bus field or tag) of a component that has not received all its
                # inputs does not exist yet - the reading row would be scheduled ahead of the component's own rows
                fcat = binfo[3] if binfo is not None else ref.inst.iface.get(a[1], (0, (), "?"))[2]
                if fcat == "o":
                    self.fail("Exception caused by invalid access: trying to access to an output signal of a component with not all "
                              "its inputs initialized (%s.%s, %d inputs missing)" % (name, a[1], ref.pending), pos)
            if binfo is not None:
                off, bdims, layout, cat = binfo
                return self._walk_bus((self.ctx, ref.pid0 + off, ref), bdims, layout, access, k, name, pos, {})
            try:
                obj = ref[a[1]]
            except CircuitError as ex:
                self.fail(str(ex), pos)
            idxs = []
            while k < len(access) and access[k][0] == "idx":
                idxs.append(self.as_index(self.eval(access[k][1]), pos))
                k += 1
            if k < len(access):
                tags = getattr(ref.inst, "sig_tags", {}).get(a[1], {})
                if k != len(access) - 1 or access[k][1] not in tags:
                    self.fail("signal %s.%s has no tag %s" % (name, a[1], access[k][1]), pos)
                return ("tag", tags, access[k][1])
            return ("sig", self._index_sig(obj, idxs, name + "." + a[1], pos))
        self.fail("%s is not a signal" % name, pos)

    # ---- calls -----------------------------------------------------------------------------------------------------------
    def call(self, e):
        name, args, pos = e[1], e[2], e[-1]
        if name in self.ar.functions:
            vals = [self.eval(a) for a in args]
            return self.w.call_function(self, name, vals, pos)
        if name in self.ar.templates:
            if self.in_function:
                self.fail("a function cannot create components", pos)
            vals = [self.eval(a) for a in args]
            for v in vals:
                if not _all_known(v):
                    self.fail("template parameters must be known at compile time", pos)
            return TemplateCall(self.w.spec(name, vals, pos))
        if name in self.ar.buses:
            self.fail("a bus is not a value", pos)
        self.fail("call to an undeclared function or template %s" % name, pos)

    def anonymous(self, e):
        """T(params)(signals): the component syntax_sugar_remover.rs:418-620 declares under the name
        <T>_<line>_<offset>, inputs assigned with <== in the order of their names, outputs as the value"""
        _, tname, params, sigs, names, pos = e
        if self.in_function:
            self.fail("Functions cannot contain calls to anonymous templates", pos)
        if self.cond_stack:
            self.fail("an anonymous component cannot be created under a run-time condition", pos)
        if tname not in self.ar.templates:
            self.fail("The template %s does not exist" % tname, pos)
        vals = [self.eval(a) for a in params]
        for v in vals:
            if not _all_known(v):
                self.fail("template parameters must be known at compile time", pos)
        spec = self.w.spec(tname, vals, pos)
        cname = "%s_%d_%d" % (tname, self.ar.line_of(pos), pos[1])
        index = (self.loop_counts.get(self.loop_stack[-1], 0),) if self.loop_stack else ()
        ref = self.ctx.component(cname, spec, index=index)
        inst = ref.inst
        in_decl = [d[0] for d in inst.decl_order if d[1] == "i"]
        out_decl = [d[0] for d in inst.decl_order if d[1] == "o"]
        if names is not None:
            if len(names) != len(in_decl) or sorted(n for _, n in names) != sorted(in_decl):
                self.fail("The number of template input signals must coincide with the number of input parameters", pos)
            for op, _n in names:
                if op != "<==":
                    self.fail("Anonymous components only admit the use of the operator <==", pos)
            assign = {n: s for (_, n), s in zip(names, sigs)}
        else:
            if len(sigs) != len(in_decl):
                self.fail("The number of template input signals must coincide with the number of input parameters", pos)
            assign = dict(zip(in_decl, sigs))
        for n in sorted(assign):
            v = self.eval(assign[n])
            self.store_signals(self._comp_field(ref, n, pos), v, "<==", pos)
        outs = [self.sig_value(self._comp_field(ref, n, pos)) for n in out_decl]
        if len(outs) == 1:
            return outs[0]
        return ("tuple", outs)

    def _comp_field(self, ref, n, pos):
        binfo = getattr(ref.inst, "bus_iface", {}).get(n)
        if binfo is not None:
            off, bdims, layout, cat = binfo
            r = self._walk_bus((self.ctx, ref.pid0 + off, ref), bdims, layout, [], 0, n, pos, {})
            return r[1]
        return ref[n]

    # ---- statements -------------------------------------------------------------------------------------------------------
    def run_block(self, stmts):
        self.scopes.append({})
        try:
            for s in stmts:
                self.exec(s)
        finally:
            self.scopes.pop()

    def exec(self, s):
        k = s[0]
        if k == "block":
            self.run_block(s[1])
        elif k == "seq":
            for x in s[1]:
                self.exec(x)
        elif k == "decl":
            self.declare_symbol(s)
        elif k == "subst":
            self.substitute(s)
        elif k == "if":
            self.exec_if(s)
        elif k == "while":
            self.exec_while(s)
        elif k == "return":
            if not self.in_function:
                self.fail("return outside a function", s[-1])
            if self.cond_stack:
                if self.mode == "abstract":
                    raise NeedsRuntime()
                self.fail("return under a run-time condition", s[-1])
            v = self.eval(s[1])
            raise _Return(_deep_copy(v))
        elif k == "ceq":
            self.constraint_equality(s)
        elif k == "log":
            self.exec_log(s)
        elif k == "assert":
            self.exec_assert(s)
        elif k == "anonstmt":
            self.eval(s[1])
        else:
            self.fail("unexpected statement", s[-1])

    def _dims(self, exprs, pos):
        dims = []
        for d in exprs:
            v = self.eval(d)
            if not isinstance(v, int):
                if self.mode == "abstract":
                    raise NeedsRuntime()
                self.fail("array dimensions must be known at compile time", pos)
            if v >= 1 << 31:
                self.fail("array dimension too large", pos)
            dims.append(v)
        return dims

    def declare_symbol(self, s):
        _, xtype, name, dim_exprs, pos = s
        dims = self._dims(dim_exprs, pos)
        t = xtype[0]
        if t == "var":
            self.declare(name, VarSlot(_zeros(dims)), pos)
            return
        if self.in_function:
            self.fail("signals and components cannot be declared inside functions", pos)
        if self.cond_stack:
            self.fail("signals and components cannot be declared under a run-time condition", pos)
        if self.loop_stack:
            # scoping.md: signals, buses and components live in the top-level block of their template or, since 2.1.5, in
            # `if` blocks with known conditions - never in the block of a loop
            self.fail("%s Is outside the initial scope" % name, pos)
        if t == "component":
            self.declare(name, CompSlot(name, dims), pos)
            return
        ctx = self.ctx
        if t == "signal":
            _, kind, tags = xtype
            mk = {"input": ctx.input, "output": ctx.output, "mid": ctx.signal}[kind]
            try:
                obj = mk(name, *dims)
            except CircuitError as ex:
                self.fail(str(ex), pos)
            slot = SigSlot(obj, kind, tags, dims)
            self.declare(name, slot, pos)
            self.w.note_decl(ctx, name, {"input": "i", "output": "o", "mid": "m"}[kind], slot.tags)
            return
        if t == "bus":
            _, bname, args, kind, tags = xtype
            vals = [self.eval(a) for a in args]
            for v in vals:
                if not _all_known(v):
                    self.fail("bus parameters must be known at compile time", pos)
            layout = self.w.bus_layout(bname, vals, pos)
            mk = {"input": ctx.input, "output": ctx.output, "mid": ctx.signal}[kind]
            # every signal field is declared under its qualified name (`s.b[1].x`: what the `.sym` file and the input list
            # call it), one after the other: the bus is the contiguous block that starts at the first one
            leaves = []
            for a, _s in _accesses(dims):
                _qualified_names(layout, 0, name + a, leaves, with_dims=True)
            first = None
            try:
                for lname, _off, ldims in leaves:
                    obj = mk(lname, *ldims)
                    if first is None:
                        first = obj.base if isinstance(obj, SigArray) else obj.val
            except CircuitError as ex:
                self.fail(str(ex), pos)
            if first is None:
                self.fail("bus %s has no signals" % bname, pos)
            slot = BusSlot(layout, dims, first, kind, tags)
            self.declare(name, slot, pos)
            self.w.note_decl(ctx, name, {"input": "i", "output": "o", "mid": "m"}[kind], slot.tags, bus=(layout, dims, leaves[0][0]))
            return
        self.fail("unexpected declaration", pos)

    # .. assignments ..
    def substitute(self, s):
        _, target, op, rhe, pos = s
        if target[0] == "tuple":
            v = self.eval(rhe)
            if not (isinstance(v, tuple) and v and v[0] == "tuple"):
                self.fail("a tuple is assigned from a tuple (or from a component with several outputs)", pos)
            if len(v[1]) != len(target[1]):
                self.fail("the two tuples have different lengths", pos)
            for t, x in zip(target[1], v[1]):
                if t[1] == "_":
                    continue
                self.assign(t, op, x, pos)
            return
        if target[1] == "_":
            v = self.eval(rhe)      # evaluated for its effects (an anonymous component); `_ <== x` also tells --inspect that
            for x in (_flat(v, []) if isinstance(v, list) else [v]):     # x is left unconstrained on purpose
                if isinstance(x, Expr) and x.kind == O.K_SIG:
                    self.underscored.add(x.val)
            return
        v = self.eval(rhe)
        if op != "=" and not self.in_function:
            self._check_and_inherit_tags(target, rhe, pos)
        self.assign(target, op, v, pos)

    # .. tags (mkdocs circom-language/tags.md) ..
    def _tags_of(self, e):
        """the tags a right-hand side carries: only a plain signal reference (own signal, or input / output of a
        sub-component) has any; an expression has none"""
        if e[0] != "var" or e[1] == "_":
            return None
        slot = None
        for sc in reversed(self.scopes):
            slot = sc.get(e[1])
            if slot is not None:
                break
        if isinstance(slot, (SigSlot, BusSlot)):
            return slot.tags if all(a[0] == "idx" for a in e[2]) else None
        if isinstance(slot, CompSlot):
            fields = [a for a in e[2] if a[0] == "field"]
            if len(fields) != 1:
                return None
            for ref in slot.refs.values():
                return getattr(ref.inst, "sig_tags", {}).get(fields[0][1], {})
        return None

    def _check_and_inherit_tags(self, target, rhe, pos):
        """a signal assigned from a tagged signal inherits its tags (with their values); an input of a sub-component that is
        declared with tags only accepts a signal that carries them ("the compiler checks if the array assigned to the input
        array has the tag")"""
        slot = None
        for sc in reversed(self.scopes):
            slot = sc.get(target[1])
            if slot is not None:
                break
        src = self._tags_of(rhe)
        if isinstance(slot, (SigSlot, BusSlot)):
            if all(a[0] == "idx" for a in target[2]):
                if src:
                    for t, val in src.items():
                        if slot.tags.get(t) is None:
                            slot.tags[t] = val
                self.assigned_tagsets.add(id(slot.tags))
            return
        if isinstance(slot, CompSlot):
            fields = [a for a in target[2] if a[0] == "field"]
            if len(fields) != 1:
                return
            for ref in slot.refs.values():
                need = getattr(ref.inst, "sig_tags", {}).get(fields[0][1], {})
                for t in need:
                    if src is None or t not in src:
                        self.fail("the signal assigned to %s.%s does not carry the tag %s its declaration asks for"
                                  % (target[1], fields[0][1], t), pos)
                break

    def assign(self, target, op, v, pos):
        name, access = target[1], target[2]
        slot = self.lookup(name, pos)
        if isinstance(v, tuple) and v and v[0] == "tuple":
            self.fail("a tuple can only be assigned to a tuple", pos)
        if isinstance(slot, VarSlot):
            if op != "=":
                self.fail("a variable is assigned with =", pos)
            if isinstance(v, TemplateCall):
                self.fail("a template can only be assigned to a component", pos)
            self.assign_var(slot, name, access, v, pos)
            return
        if isinstance(slot, CompSlot) and op == "=":
            r = self.resolve_signal(slot, name, access, pos, True)
            if r[0] == "tag":
                self._set_tag(r, v, pos)
                return
            if r[0] != "comp":
                self.fail("signals are assigned with <== or <--", pos)
            if not isinstance(v, TemplateCall):
                self.fail("a component is initialised with a template", pos)
            if self.cond_stack:
                self.fail("a component cannot be created under a run-time condition", pos)
            idx = r[2]
            if idx in slot.refs:
                self.fail("component %s is instantiated twice" % name, pos)
            for other in slot.refs.values():
                if other.inst.name != v.spec.name:
                    # (the elements may differ in their PARAMETERS - a Mixed cluster - but not in their template)
                    self.fail("all components of the array %s must be instances of the same template" % name, pos)
                break
            try:
                slot.refs[idx] = self.ctx.component(name, v.spec, index=idx)
            except CircuitError as ex:
                self.fail(str(ex), pos)
            return
        if self.in_function:
            self.fail("functions cannot assign signals", pos)
        r = self.resolve_signal(slot, name, access, pos, True)
        if r[0] == "tag":
            if op != "=":
                self.fail("a tag is assigned with =", pos)
            self._set_tag(r, v, pos)
            return
        if op == "=":
            self.fail("signals are assigned with <== or <--", pos)
        if r[0] == "comp":
            self.fail("a component is initialised with =", pos)
        if isinstance(slot, (SigSlot, BusSlot)) and slot.kind == "input":
            self.fail("the input signal %s of the template cannot be assigned inside it" % name, pos)
        dst = r[1]
        self.store_signals(dst, v, op, pos)

    def _set_tag(self, r, v, pos):
        if not isinstance(v, int):
            self.fail("tag values must be known at compile time", pos)
        if id(r[1]) in self.assigned_tagsets:
            self.fail("Invalid assignment: tags cannot be assigned to a signal already initialized", pos)
        r[1][r[2]] = v

    def _weak(self, old, v, name, pos):
        """`var` arrays of different lengths (program_structure/src/utils/memory_slice.rs:129-160, 300-320: variables are not
        strict): the overlapping positions are assigned, the others keep their values; a warning is recorded
        (execute.rs:3949-3965).  Same number of dimensions only."""
        so, sv = _shape(old), _shape(v)
        if so == sv:
            return v
        if len(so) != len(sv) or not so:
            self.fail("assignee and assigned arrays of %s have different numbers of dimensions" % name, pos)
        self.w.note_typing_warning(self, so, sv, pos)

        def merge(o, x):
            if not isinstance(o, list):
                return x
            return [merge(o[i], x[i]) if i < len(x) else o[i] for i in range(len(o))]
        return merge(old, v)

    def assign_var(self, slot, name, access, v, pos):
        v = _deep_copy(v)
        if not access:
            old = slot.value
            if isinstance(old, list):
                if not isinstance(v, list):
                    self.fail("a single value is assigned to the array %s" % name, pos)
                v = self._weak(old, v, name, pos)
            slot.value = v
            return
        cur = slot.value
        for n, a in enumerate(access):
            if a[0] != "idx":
                self.fail("a variable has no field %s" % a[1], pos)
            if not isinstance(cur, list):
                self.fail("too many indices for %s" % name, pos)
            iv = self.eval(a[1])
            i = self.as_index(iv, pos)
            if i >= len(cur):
                self.fail("array index out of bounds: %s[%d] of %d" % (name, i, len(cur)), pos)
            if n == len(access) - 1:
                if isinstance(cur[i], list) != isinstance(v, list):
                    self.fail("assignee and assigned arrays of %s have different sizes" % name, pos)
                cur[i] = self._weak(cur[i], v, name, pos) if isinstance(v, list) else v
            else:
                cur = cur[i]

    def store_signals(self, dst, v, op, pos):
        """dst: Expr (one signal) | SigArray | BusView; v: a value of the same shape"""
        if isinstance(dst, BusView):
            dsts = _flat(self.bus_value(dst), [])
        elif isinstance(dst, SigArray):
            dsts = _flat(self.sig_value(dst), [])
            if not isinstance(v, list):
                self.fail("a single value is assigned to an array of %d signals" % len(dsts), pos)
        else:
            dsts = [dst]
        vals = _flat(v, []) if isinstance(v, list) else [v]
        if len(vals) != len(dsts):
            self.fail("assignee (%d signals) and assigned value (%d) have different sizes" % (len(dsts), len(vals)), pos)
        for d, x in zip(dsts, vals):
            if isinstance(x, TemplateCall):
                self.fail("a template is not a signal value", pos)
            try:
                if self.cond_stack:
                    if op == "<==":
                        self.fail("a constraint cannot be generated under a run-time condition", pos)
                    fr = self.cond_stack[-1]
                    if d.val in fr:
                        self.fail("signal assigned twice", pos)
                    fr[d.val] = (d, x)
                elif op == "<==":
                    self.ctx.set(d, x)
                else:
                    self.ctx.hint(d, x)
            except CircuitError as ex:
                if ":" in str(ex).split(" ")[0]:
                    raise
                self.fail(str(ex), pos)

    def constraint_equality(self, s):
        _, l, r, pos = s
        if self.in_function:
            self.fail("functions cannot generate constraints", pos)
        if self.cond_stack:
            self.fail("a constraint cannot be generated under a run-time condition", pos)
        a = self.eval(l)
        b = self.eval(r)
        la = _flat(a, []) if isinstance(a, list) else [a]
        lb = _flat(b, []) if isinstance(b, list) else [b]
        if len(la) != len(lb):
            self.fail("the two sides of === have different sizes", pos)
        for x, y in zip(la, lb):
            try:
                self.ctx.enforce(x, y)
            except CircuitError as ex:
                self.fail(str(ex), pos)

    def exec_log(self, s):
        _, args, pos = s
        vals = []
        for a in args:
            if a[0] == "str":
                vals.append(a[1])
            else:
                v = self.eval(a)
                if isinstance(v, list):
                    self.fail("log of an array", pos)
                vals.append(v)
        if self.in_function and self.mode != "trace":
            if self.mode == "const":
                # a function evaluated by the compiler logs while compiling (execute.rs LogCall on known values)
                self.w.compile_log.append(" ".join(str(v) for v in vals))
            return
        if self.cond_stack:
            self.fail("log under a run-time condition", pos)
        self.ctx.log(*vals)

    def exec_assert(self, s):
        _, e, pos = s
        v = self.eval(e)
        if isinstance(v, list):
            self.fail("assert on an array", pos)
        if isinstance(v, int):
            if v == 0:
                self.fail("assert failed: false", pos)               # ReportCode::RuntimeError at compile time
            return
        if self.mode == "abstract":
            return
        if self.cond_stack:
            self.fail("assert under a run-time condition", pos)
        self.ctx.assert_(v)

    # .. control flow ..
    def exec_if(self, s):
        _, cond, then, other, pos = s
        c = self.eval(cond)
        if isinstance(c, list):
            self.fail("condition is an array", pos)
        if isinstance(c, int):
            if c != 0:
                self.run_block([then])
            elif other is not None:
                self.run_block([other])
            return
        # run-time condition: if-conversion
        before = self._snapshot()
        self.cond_stack.append({})
        self.run_block([then])
        hints_t = self.cond_stack.pop()
        after_t = self._snapshot()
        self._restore(before)
        self.cond_stack.append({})
        if other is not None:
            self.run_block([other])
        hints_e = self.cond_stack.pop()
        after_e = self._snapshot()
        # merge variables
        for slot, vt, ve in zip(self._var_slots(), after_t, after_e):
            slot.value = self.select(c, vt, ve, pos)
        if set(hints_t) != set(hints_e):
            self.fail("a signal assigned under a run-time condition must be assigned in both branches", pos)
        for pid, (d, x) in hints_t.items():
            y = hints_e[pid][1]
            v = self.select(c, x, y, pos)
            if self.cond_stack:
                self.cond_stack[-1][pid] = (d, v)
            elif self.mode != "abstract":
                try:
                    self.ctx.hint(d, v)
                except CircuitError as ex:
                    self.fail(str(ex), pos)

    def _var_slots(self):
        out = []
        for sc in self.scopes:
            for s in sc.values():
                if isinstance(s, VarSlot):
                    out.append(s)
        return out

    def _snapshot(self):
        return [_deep_copy(s.value) for s in self._var_slots()]

    def _restore(self, snap):
        for s, v in zip(self._var_slots(), snap):
            s.value = _deep_copy(v)

    def exec_while(self, s):
        _, cond, body, pos = s
        self.loop_stack.append(pos)
        self.loop_counts.setdefault(pos, 0)
        try:
            n = 0
            while True:
                c = self.eval(cond)
                if isinstance(c, list):
                    self.fail("condition is an array", pos)
                if not isinstance(c, int):
                    if self.mode == "abstract":
                        raise NeedsRuntime()
                    self.fail("the condition of a loop in a template must be known at compile time", pos)
                if c == 0:
                    break
                self.run_block([body])
                self.loop_counts[pos] += 1
                n += 1
                if n > self.w.max_loop:
                    self.fail("loop does not terminate (more than %d iterations)" % self.w.max_loop, pos)
        finally:
            self.loop_stack.pop()


class World:
    """one program being built: archive + field + the caches of template specs, bus layouts and compiled functions"""

    def __init__(self, archive: Archive, prime: str):
        self.archive = archive
        self.prime = prime
        self.fp = fp_for(prime)
        self.max_loop = 1 << 26
        self.compile_log = []
        self._bus_layouts = {}
        self._rt_functions = {}
        self._fn_kind = {}
        self.prog = None
        self.inspect = False          # --inspect: collect the warnings of constraint_correctness_analysis.rs
        self.warnings = []
        self.typing_warnings = []     # arrays of different lengths assigned to variables (execute.rs:3949-3965)
        self.template_depth = 0

    # ---- templates ----------------------------------------------------------------------------------------------------------
    def spec(self, name, vals, pos):
        d = self.archive.templates[name]
        _, _, params, body, flags, tpos = d
        if "custom" in flags or "extern_c" in flags:
            fn, ln, col = self.archive.where(pos)
            raise CircuitError("%s:%d:%d: custom / extern_c templates are not supported by this front-end" % (fn, ln, col))
        if len(vals) != len(params):
            fn, ln, col = self.archive.where(pos)
            raise CircuitError("%s:%d:%d: template %s takes %d parameters" % (fn, ln, col, name, len(params)))
        frozen = tuple(_freeze(v) for v in vals)
        world = self

        def body_fn(ctx, *pvals):
            ex = Executor(world, "trace", ctx)
            for pn, pv in zip(params, pvals):
                ex.scopes[0][pn] = VarSlot(_thaw(pv))
            ctx.inst.decl_order = []
            ctx.inst.sig_tags = {}
            ctx.inst.bus_iface = {}
            ctx._bus_decls = ctx.inst._bus_decls = []
            # a template that instantiates itself without ever reaching its base case must end as an error, not as a
            # stack overflow of the interpreter
            world.template_depth += 1
            try:
                if world.template_depth > 150:
                    fn_, ln, col = world.archive.where(tpos)
                    raise CircuitError("%s:%d:%d: template %s: instantiations nested more than 150 deep" % (fn_, ln, col, name))
                ex.run_block(body[1])
            finally:
                world.template_depth -= 1
            if world.inspect:
                world.inspect_instance(ctx, ex, name, pvals)
            world.finish_instance(ctx)
        body_fn.__name__ = name
        return TemplateSpec(name, body_fn, frozen)

    def note_typing_warning(self, ex, expected, given, pos):
        if ex.mode == "abstract":
            return
        fn, ln, col = self.archive.where(pos)
        n_exp, n_giv = 1, 1
        for d in expected:
            n_exp *= d
        for d in given:
            n_giv *= d
        kind = "smaller length, the remaining positions are not modified. Initially all variables are initialized to 0." \
            if n_giv < n_exp else "greater length, the remaining positions of the expression are not assigned to the array."
        msg = "%s:%d:%d: Typing warning: Mismatched dimensions, assigning to an array an expression of %s\n  Expected length: %d, given %d" \
            % (fn, ln, col, kind, n_exp, n_giv)
        if msg not in self.typing_warnings:
            self.typing_warnings.append(msg)

    def inspect_instance(self, ctx, ex, name, pvals):
        """--inspect (dag/src/constraint_correctness_analysis.rs): signals of the instance, and inputs / outputs of its
        sub-components, that appear in no constraint of this template (and were not handed to `_`)"""
        seen = set(ex.underscored)
        for a, b, c in ctx.cons:
            seen.update(a)
            seen.update(b)
            seen.update(c)
        title = "%s(%s)" % (name, ", ".join(str(_thaw(v)).replace(" ", "") for v in pvals))

        def names_of(base, dims):
            if not dims:
                return [base]
            out = [base]
            for d in dims:
                out = [n + "[%d]" % i for n in out for i in range(d)]
            return out

        def report(kind, base, dims, pid0, size):
            missing = [k for k in range(size) if pid0 + k not in seen]
            if not missing:
                return
            nm = names_of(base, dims)
            if len(missing) == 1:
                what = "Local signal %s does not appear in any constraint" if kind == "local" else \
                    "Subcomponent input/output signal %s does not appear in any constraint of the father component"
                self.warnings.append('In template "%s": ' % title + what % nm[missing[0]])
            else:
                what = "Array of local signals %s contains a total of %d signals that do not appear in any constraint" if kind == "local" \
                    else "Array of subcomponent input/output signals %s contains a total of %d signals that do not appear in any " \
                         "constraint of the father component"
                self.warnings.append('In template "%s": ' % title + what % (base, len(missing))
                                     + " = For example: %s, %s." % (nm[missing[0]], nm[missing[1]]))
        for cat, sname, dims, pid0, size in ctx.own:
            report("local", sname, dims, pid0, size)
        for ref in ctx.comps:
            cname = ref.name + "".join("[%d]" % i for i in ref.index)
            for sname, (off, dims, cat) in ref.inst.iface.items():
                size = 1
                for d in dims:
                    size *= d
                report("io", cname + "." + sname, dims, ref.pid0 + off, size)

    def note_decl(self, ctx, name, cat, tags, bus=None):
        ctx.inst.decl_order.append((name, cat))
        ctx.inst.sig_tags[name] = tags            # the slot's own table: values set in the body are seen by the parent
        if bus is not None:
            ctx._bus_decls.append((name, cat, bus))

    def finish_instance(self, ctx):
        """called at the end of a template body, before dsl finalises the numbering: remember where bus-typed inputs and
        outputs sit so that a parent can address `component.bus.field`"""
        if not ctx._bus_decls:
            return
        decls = ctx._bus_decls
        inst = ctx.inst

        # offsets are only known after Ctx.finalize (outputs, inputs, intermediates are renumbered): resolve lazily
        class _LazyIface(dict):
            def get(self_inner, key, default=None):
                for name, cat, (layout, dims, first) in decls:
                    if name == key and cat in ("i", "o"):
                        off = inst.iface[first][0]
                        return (off, dims, layout, cat)
                return default
        inst.bus_iface = _LazyIface()

    # ---- buses --------------------------------------------------------------------------------------------------------------
    def bus_layout(self, name, vals, pos):
        """execute a bus definition: its body declares signals (and nested buses) in order; the layout is the flattening
        (constraint_generation/src/execution_data/executed_bus.rs: fields in declaration order)"""
        d = self.archive.buses.get(name)
        if d is None:
            fn, ln, col = self.archive.where(pos)
            raise CircuitError("%s:%d:%d: bus %s is not defined" % (fn, ln, col, name))
        key = (name, tuple(_freeze(v) for v in vals))
        lay = self._bus_layouts.get(key)
        if lay is not None:
            return lay
        _, _, params, body, bpos = d
        if len(vals) != len(params):
            fn, ln, col = self.archive.where(pos)
            raise CircuitError("%s:%d:%d: bus %s takes %d parameters" % (fn, ln, col, name, len(params)))
        lay = BusLayout(name)
        ex = Executor(self, "const")
        ex.in_function = True              # vars, loops and conditionals behave as in a function body
        for pn, pv in zip(params, vals):
            ex.scopes[0][pn] = VarSlot(_deep_copy(pv))
        world = self

        def declare_field(s):
            _, xtype, fname, dim_exprs, fpos = s
            dims = ex._dims(dim_exprs, fpos)
            if fname in lay.fields:
                ex.fail("field %s declared twice" % fname, fpos)
            if xtype[0] == "signal":
                if xtype[1] != "mid":
                    ex.fail("the fields of a bus are neither inputs nor outputs", fpos)
                size, sub = 1, None
            elif xtype[0] == "bus":
                bvals = [ex.eval(a) for a in xtype[2]]
                sub = world.bus_layout(xtype[1], bvals, fpos)
                size = sub.size
            else:
                return False
            for x in dims:
                size *= x
            lay.fields[fname] = (lay.size, tuple(dims), sub)
            lay.order.append(fname)
            lay.size += size
            return True
        orig = ex.declare_symbol

        def declare_symbol(s):
            if s[1][0] in ("signal", "bus"):
                declare_field(s)
            else:
                orig(s)
        ex.declare_symbol = declare_symbol
        ex.run_block(body[1])
        self._bus_layouts[key] = lay
        return lay

    # ---- functions ------------------------------------------------------------------------------------------------------------
    def call_function(self, caller: Executor, name, vals, pos):
        d = self.archive.functions[name]
        _, _, params, body, fpos = d
        if len(vals) != len(params):
            caller.fail("function %s takes %d arguments" % (name, len(params)), pos)
        for v in vals:
            if isinstance(v, (TemplateCall, tuple)):
                caller.fail("a function takes values", pos)
        if caller.depth > 200:
            caller.fail("function calls nested too deeply", pos)
        known = all(_all_known(v) for v in vals)
        if known:
            mode = "const" if caller.mode != "abstract" else "abstract"
        elif caller.mode == "abstract":
            mode = "abstract"
        else:
            mode = "trace"
            # decide between inlining and tier-2 bytecode on an abstract run
            absvals = [self._abstract(v) for v in vals]
            try:
                self._run_function(caller, name, params, body, absvals, "abstract", pos)
            except NeedsRuntime:
                from .circom_rt import call_runtime_function
                return call_runtime_function(self, caller, name, vals, pos)
        return self._run_function(caller, name, params, body, vals, mode, pos)

    def _abstract(self, v):
        if isinstance(v, list):
            return [self._abstract(x) for x in v]
        return v if isinstance(v, int) else UNK

    def _run_function(self, caller, name, params, body, vals, mode, pos):
        ex = Executor(self, mode, caller.ctx if mode == "trace" else None, caller.depth + 1)
        ex.in_function = True
        for pn, pv in zip(params, vals):
            ex.scopes[0][pn] = VarSlot(_deep_copy(pv))
        try:
            ex.run_block(body[1])
        except _Return as r:
            return r.value
        except RecursionError:
            caller.fail("function calls nested too deeply", pos)
        caller.fail("function %s ends without a return" % name, pos)


# ---- entry points ----------------------------------------------------------------------------------------------------------
def build_program(archive: Archive, prime="bn128", inspect=False):
    """archive -> dsl.Program (main component instantiated, public inputs ordered)"""
    if archive.main is None:
        raise CircuitError("No main specified in the project structure")          # ReportCode::NoMain
    sys.setrecursionlimit(max(20000, sys.getrecursionlimit()))
    from .circom_check import check_archive
    check_archive(archive)                 # the static part of type_analysis: also over code that is never executed
    world = World(archive, prime)
    world.inspect = inspect
    _, public, init, pos = archive.main
    ex = Executor(world, "const")
    if init[0] == "anon":
        ex.fail("The main component cannot contain an anonymous call", pos)
    if init[0] == "parallel":
        init = init[1]
    if init[0] != "call" or init[1] not in archive.templates:
        ex.fail("the main component is initialised with a template", pos)
    vals = [ex.eval(a) for a in init[2]]
    spec = world.spec(init[1], vals, pos)
    prog = dsl.Program.__new__(dsl.Program)
    world.prog = prog
    prog.world = world
    try:
        dsl.Program.__init__(prog, spec, public=(), prime=prime)
    except RecursionError:
        raise CircuitError("%s: expressions, calls or components nested too deeply" % archive.where(pos)[0]) from None
    # the public list names signals and buses of main: a bus stands for all its fields
    m = prog.main
    buses = {bname: bus for bname, cat, bus in getattr(m, "_bus_decls", ()) if cat == "i"}
    expanded = []
    for pname in public:
        if pname in buses:
            layout, dims, _first = buses[pname]
            leaves = []
            for a, _s in _accesses(dims):
                _qualified_names(layout, 0, pname + a, leaves)
            expanded += [l[0] for l in leaves]
        elif pname in m.iface and m.iface[pname][2] == "i":
            expanded.append(pname)
        else:
            ex.fail("public signal %s is not an input of main" % pname, pos)
    if expanded:
        prog.public = tuple(expanded)
        prog._reorder_main_public()
    _qualify_main_bus_inputs(prog)
    # the bus-field map of the `.dat` (c_code_generator.rs:740-794; filled by build.rs:601-626 get_info_buses): per bus instance -
    # in the order the layouts were completed, a nested bus before the one that holds it - its fields in declaration order:
    # (offset inside the bus, dimensions, size of ONE element, id of the field's own bus | None, name)
    ids = {id(lay): i for i, lay in enumerate(world._bus_layouts.values())}
    prog.bus_field_map = []
    for lay in world._bus_layouts.values():
        fields = []
        for fname in lay.order:
            off, dims, sub = lay.fields[fname]
            fields.append((int(off), tuple(int(d) for d in dims), int(sub.size) if sub is not None else 1, ids[id(sub)] if sub is not None else None, fname))
        prog.bus_field_map.append(fields)
    return prog


def _accesses(dims):
    """("[i][j]", element index) of every element of an array, row-major (build.rs:324-346 get_accesses)"""
    if not dims:
        return [("", 0)]
    inner = _accesses(dims[1:])
    stride = 1
    for d in dims[1:]:
        stride *= d
    return [("[%d]%s" % (i, a), i * stride + s) for i in range(dims[0]) for a, s in inner]


def _qualified_names(layout, start, prefix, out, with_dims=False):
    """one entry per signal field of a bus, named <prefix>.<field> (build.rs:348-382 get_qualified_names):
    (name, offset, size), or (name, offset, dims) with with_dims"""
    for fname in layout.order:
        off, dims, sub = layout.fields[fname]
        name = "%s.%s" % (prefix, fname)
        if sub is not None:
            for a, s in _accesses(dims):
                _qualified_names(sub, start + off + s * sub.size, name + a, out, with_dims)
        else:
            size = 1
            for d in dims:
                size *= d
            out.append((name, start + off, tuple(dims) if with_dims else size))


def _qualify_main_bus_inputs(prog):
    """Bus-typed inputs of main enter the input list (the `.dat` hash map) once per signal field under their qualified names
    - the keys main.cpp's qualify_input makes of nested JSON objects - and once more as the whole bus, behind every other
    entry (compiler/src/circuit_design/build.rs:300-425 main_input_list)."""
    m = prog.main
    tail = []
    for name, cat, (layout, dims, first) in getattr(m, "_bus_decls", ()):
        if cat != "i":
            continue
        size = layout.size
        for d in dims:
            size *= d
        tail.append((name, m.iface[first][0], size))      # (the fields themselves are declared under their qualified names)
    m.input_names = list(m.input_names) + tail


def program_from_file(path, libs=(), prime="bn128", inspect=False):
    return build_program(parse_program(path, libs), prime, inspect)


def program_from_text(text, prime="bn128", name="<text>", inspect=False):
    return build_program(parse_text(text, name), prime, inspect)

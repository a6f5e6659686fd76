"""circom FUNCTIONS whose control flow depends on run-time values, compiled from their source text to tier-2 bytecode
(frontend/rtcode.py RtFunction): the path a call takes when the abstract run of circom_exec finds a `while` condition, an
array index or a `return` that depends on an argument.

In the reference such a call is a CallBucket (call_bucket.rs:466-533) into a C++ function whose loops and branches are real
control flow over `Fr_isTrue` (loop_bucket.rs:76-91, branch_bucket.rs:100-122) and whose array addresses go through
`Fr_toInt` (compute_bucket.rs:361-363).  Here the body is PARTIALLY EVALUATED while the bytecode is written:

  * a value is a Python int (known while compiling), a register / constant of the function (RtVar), or a nested list;
    arguments that are known at the call site (limb sizes, counts) are baked into the specialisation, so the loops over
    limbs unroll and only data-dependent control flow is left as jumps;
  * before a run-time region (an `if` or `while` on a register) is entered, what the region assigns is PINNED: a scalar
    variable to a register, of an array variable the ELEMENTS the region assigns (`Cell`; the region is built once as a scout
    to learn them, the builder is rolled back, and the region is built again - `_region`), an array that is indexed by a
    run-time value to a contiguous block.  Inside and after the region assignments to a pinned location are in-place copies,
    so values flow around the loop and out of both arms without merges;
  * a pinned register bound to another variable, stored into an array or returned is copied first (a pinned register is a
    location, not a value);
  * an array read or written at a run-time index is a block access (F_LDX / F_STX; the linear index of a multi-dimensional
    array is computed at run time); an unpinned array read that way is materialised into a block first;
  * `return` copies into the function's result registers and jumps to its end - unless it is the function's only exit, whose
    value is handed back as it is; nested calls are inlined (their arguments by value, their own result registers and end label).
"""
from __future__ import annotations

from .. import opcodes as O
from .dsl import CircuitError
from .rtcode import RtVar, RtArray, F_JMP
from . import circom_exec as X


UNROLL_BUDGET = 4096       # instructions a known loop may emit before its remaining trips are left to run time


class Pinned:
    """a variable that lives in registers: a scalar (dims = ()) or a row-major block"""
    __slots__ = ("base", "dims")

    def __init__(self, base, dims):
        self.base, self.dims = base, tuple(dims)

    @property
    def size(self):
        n = 1
        for d in self.dims:
            n *= d
        return n


class Cell:
    """ONE element of an array variable that lives in a register (the elements a run-time region assigns; the others stay
    plain values)"""
    __slots__ = ("reg",)

    def __init__(self, reg):
        self.reg = reg


class _Repin(Exception):
    """a run-time region assigned an array element its scout did not see: the region is rebuilt with every element pinned"""


class _Frame:
    """one (inlined) function activation"""

    def __init__(self, name):
        self.name = name
        self.scopes = [{}]
        self.ret_shape = None
        self.ret_base = None
        self.jumps = []
        self.direct = None         # the value of a function whose only return is reached on every path (no result registers)
        self.region0 = 0           # depth of run-time regions at which the activation started


class _Stop(Exception):
    """unwinds the building of the current function after a `return` outside every run-time region"""


class RtCompiler:
    def __init__(self, world, f, pos):
        self.w = world
        self.ar = world.archive
        self.fp = world.fp
        self.q = world.fp.q
        self.f = f
        self.frames = []
        self.region = 0            # depth of run-time regions being built
        self.pinned_regs = set()
        self.call_pos = pos
        self.depth = 0
        self.scouting = False      # a region is being built once to learn which array elements it assigns
        self.scout_log = {}        # id(slot) -> set of element paths | "rt" (indexed by a run-time value)
        self.region_visible = []   # per open region: ids of the slots that existed when it was entered

    # ---- helpers ------------------------------------------------------------------------------------------------------
    def fail(self, msg, pos):
        fn, ln, col = self.ar.where(pos)
        raise CircuitError("%s:%d:%d: %s" % (fn, ln, col, msg))

    @property
    def scopes(self):
        return self.frames[-1].scopes

    def lookup(self, name, pos):
        for sc in reversed(self.scopes):
            s = sc.get(name)
            if s is not None:
                return s
        self.fail("undeclared symbol %s" % name, pos)

    def known(self, v):
        return isinstance(v, int) or (isinstance(v, RtVar) and v.kind == 'c')

    def kval(self, v):
        return v if isinstance(v, int) else v.val

    def rt(self, v):
        return v if isinstance(v, RtVar) else self.f.lift(v)

    def binop(self, op, a, b, pos):
        if isinstance(a, (list, Pinned)) or isinstance(b, (list, Pinned)):
            self.fail("operator %s on arrays" % op, pos)
        name, code = X._BIN[op]
        if self.known(a) and self.known(b):
            x, y = self.kval(a), self.kval(b)
            if op in ("/", "\\", "%") and y == 0:
                self.fail("division by zero", pos)
            return getattr(self.fp, name)(x, y)
        return self.f.emit(code, self.rt(a), self.rt(b))

    def unop(self, op, a, pos):
        name, code = X._UN[op]
        if self.known(a):
            return getattr(self.fp, name)(self.kval(a))
        return self.f.emit(code, self.rt(a), None)

    # ---- values of variables --------------------------------------------------------------------------------------------
    def _pin_value(self, v):
        """value (int | RtVar | nested list) -> Pinned, with fresh registers holding copies"""
        f = self.f
        if isinstance(v, Pinned):
            return v
        if not isinstance(v, list):
            r = f.var(self.rt(v))
            self.pinned_regs.add(r.val)
            return Pinned(r.val, ())
        dims = X._shape(v)
        leaves = X._flat(v, [])
        arr = f.array(len(leaves), init=[self.rt(x) for x in leaves])
        self.pinned_regs.update(range(arr.base, arr.base + arr.n))
        return Pinned(arr.base, dims)

    def _unpin_view(self, p: Pinned):
        """the current contents of a pinned variable as a value of register ALIASES (copied when they are bound elsewhere)"""
        if not p.dims:
            return RtVar(self.f, 'r', p.base)

        def rec(base, dims):
            if not dims:
                return RtVar(self.f, 'r', base)
            stride = 1
            for d in dims[1:]:
                stride *= d
            return [rec(base + i * stride, dims[1:]) for i in range(dims[0])]
        return rec(p.base, p.dims)

    def _view(self, v):
        """a variable's value with its pinned elements as register aliases"""
        if isinstance(v, list):
            return [self._view(x) for x in v]
        if isinstance(v, Cell):
            return RtVar(self.f, 'r', v.reg)
        if isinstance(v, Pinned):
            return self._unpin_view(v)
        return v

    def _own(self, v):
        """a value about to be bound to a variable / array element / result: aliases of pinned registers are copied"""
        if isinstance(v, list):
            return [self._own(x) for x in v]
        if isinstance(v, Pinned):
            return self._own(self._unpin_view(v))
        if isinstance(v, RtVar) and v.kind == 'r' and v.val in self.pinned_regs:
            return self.f.var(v)
        if isinstance(v, RtVar) and v.kind == 'c':
            return v.val
        return v

    def _linear_index(self, dims, idxs, pos):
        """run-time linear index register + remaining dims for an access dims[idxs...] (all dimensions consumed)"""
        if len(idxs) != len(dims):
            self.fail("an array indexed by a run-time value must be indexed down to one element", pos)
        lin = None
        for k, i in enumerate(idxs):
            stride = 1
            for d in dims[k + 1:]:
                stride *= d
            term = self.binop("*", i, stride, pos) if stride != 1 else i
            lin = term if lin is None else self.binop("+", lin, term, pos)
        return lin

    # ---- expressions ------------------------------------------------------------------------------------------------------
    def eval(self, e):
        k = e[0]
        if k == "num":
            return e[1] % self.q
        if k == "var":
            return self.read(e)
        if k == "bin":
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
            if self.known(c):
                return self.eval(e[2]) if self.kval(c) != 0 else self.eval(e[3])
            # run-time choice: a branch around the two evaluations, result in a fresh pinned value
            f = self.f
            self.region += 1
            res = None
            with f.if_(self.rt(c)):
                a = self.eval(e[2])
                res = self._pin_value(self._own(a))
            shape = res.dims
            with f.else_():
                b = self._own(self.eval(e[3]))
                if X._shape(b) != shape:
                    self.fail("the two sides of a run-time choice have different sizes", e[-1])
                self._store_pinned(res, [], b, e[-1])
            self.region -= 1
            return self._unpin_view(res)
        if k == "call":
            return self.call(e)
        if k == "arr":
            vals = [self.eval(x) for x in e[1]]
            vals = [self._unpin_view(v) if isinstance(v, Pinned) else v for v in vals]
            s0 = X._shape(vals[0])
            for v in vals[1:]:
                if X._shape(v) != s0:
                    self.fail("inline array with elements of different sizes", e[-1])
            return vals
        self.fail("this expression is not allowed inside a function evaluated at run time", e[-1])

    def read(self, e):
        name, access, pos = e[1], e[2], e[-1]
        slot = self.lookup(name, pos)
        v = slot.value
        idxs = []
        for a in access:
            if a[0] != "idx":
                self.fail("a variable has no field %s" % a[1], pos)
            idxs.append(self.eval(a[1]))
        if not idxs:
            return self._view(v)
        if all(self.known(i) for i in idxs):
            if isinstance(v, Pinned):
                v = self._unpin_view(v)
            for i in idxs:
                i = self.kval(i)
                if not isinstance(v, list):
                    self.fail("too many indices for %s" % name, pos)
                if i >= len(v):
                    self.fail("array index out of bounds: %s[%d] of %d" % (name, i, len(v)), pos)
                v = v[i]
            return self._view(v)
        # run-time index: a block access
        if isinstance(v, Pinned):
            p = v
        else:
            if not isinstance(v, list):
                self.fail("too many indices for %s" % name, pos)
            p = self._pin_value(self._view(v))  # materialised copy (the variable itself stays as it is)
        lin = self._linear_index(p.dims, idxs, pos)
        return RtArray(self.f, p.base, p.size).load(self.rt(lin))

    def call(self, e):
        name, args, pos = e[1], e[2], e[-1]
        if name not in self.ar.functions:
            self.fail("only functions can be called inside a function", pos)
        vals = [self.eval(a) for a in args]
        vals = [self._unpin_view(v) if isinstance(v, Pinned) else v for v in vals]
        return self.inline(name, vals, pos)

    # ---- functions ----------------------------------------------------------------------------------------------------------
    def _const_value(self, v):
        if isinstance(v, list):
            return [self._const_value(x) for x in v]
        return self.kval(v)

    def _all_known(self, v):
        if isinstance(v, list):
            return all(self._all_known(x) for x in v)
        return self.known(v)

    def inline(self, name, vals, pos):
        d = self.ar.functions[name]
        _, _, params, body, fpos = d
        if len(vals) != len(params):
            self.fail("function %s takes %d arguments" % (name, len(params)), pos)
        if all(self._all_known(v) for v in vals):
            ex = X.Executor(self.w, "const")
            ex.depth = self.depth
            return self.w.call_function(ex, name, [self._const_value(v) for v in vals], pos)
        if self.depth > 64 or any(fr.name == name for fr in self.frames):
            self.fail("recursion on run-time values is not supported (function %s)" % name, pos)
        fr = _Frame(name)
        for pn, pv in zip(params, vals):
            fr.scopes[0][pn] = X.VarSlot(self._own(X._deep_copy(pv)))
        self.frames.append(fr)
        self.depth += 1
        saved_region = self.region
        fr.region0 = self.region
        try:
            try:
                self.run_block(body[1])
                if not fr.jumps and fr.direct is None:
                    self.fail("function %s ends without a return" % name, pos)
            except _Stop:
                pass
        finally:
            self.region = saved_region
            self.depth -= 1
            self.frames.pop()
        if fr.direct is not None:
            return fr.direct[0]
        end = len(self.f.code)
        for j in fr.jumps:
            self.f.code[j][1] = end
        self.f.last_if = None
        return self._unpin_view(Pinned(fr.ret_base, fr.ret_shape))

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
            _, xtype, name, dim_exprs, pos = s
            if xtype[0] != "var":
                self.fail("signals and components cannot be declared inside functions", pos)
            dims = []
            for dexp in dim_exprs:
                v = self.eval(dexp)
                if not self.known(v):
                    self.fail("array dimensions must be known at compile time", pos)
                dims.append(self.kval(v))
            for sc in self.scopes:
                if name in sc:
                    self.fail("symbol %s declared twice" % name, pos)
            self.scopes[-1][name] = X.VarSlot(X._zeros(dims))
        elif k == "subst":
            _, target, op, rhe, pos = s
            if op != "=" or target[0] != "var":
                self.fail("functions assign variables with =", pos)
            v = self.eval(rhe)
            self.assign(target, v, pos)
        elif k == "if":
            self.exec_if(s)
        elif k == "while":
            self.exec_while(s)
        elif k == "return":
            self.exec_return(s)
        elif k == "assert":
            v = self.eval(s[1])
            if self.known(v):
                if self.kval(v) == 0:
                    self.fail("assert failed: false", s[-1])
            # a run-time assert inside a function has no counterpart in the bytecode: the templates that call hint
            # functions constrain their results; dropped (documented in DESIGN 3.6)
        elif k == "log":
            pass
        else:
            self.fail("this statement is not allowed inside a function", s[-1])

    def assign(self, target, v, pos):
        name, access = target[1], target[2]
        slot = self.lookup(name, pos)
        idxs = []
        for a in access:
            if a[0] != "idx":
                self.fail("a variable has no field %s" % a[1], pos)
            idxs.append(self.eval(a[1]))
        cur = slot.value
        if isinstance(cur, Pinned):
            self._store_pinned(cur, idxs, v, pos)
            return
        if not all(self.known(i) for i in idxs):
            # a run-time index on an unpinned array: the whole array becomes a block now
            if self.scouting:
                self.scout_log[id(slot)] = "rt"
            elif self._is_outer(slot):
                raise _Repin()
            slot.value = cur = self._pin_value(self._view(cur))
            self._store_pinned(cur, idxs, v, pos)
            return
        v = self._own(v)
        if not idxs:
            if isinstance(cur, list):
                self._store_elements(slot, cur, (), self._weak(self._view(cur), v, name, pos))
            else:
                slot.value = v
            return
        path = []
        for n, i in enumerate(idxs):
            i = self.kval(i)
            if not isinstance(cur, list):
                self.fail("too many indices for %s" % name, pos)
            if i >= len(cur):
                self.fail("array index out of bounds: %s[%d] of %d" % (name, i, len(cur)), pos)
            path.append(i)
            if n == len(idxs) - 1:
                if isinstance(cur[i], list):
                    self._store_elements(slot, cur[i], tuple(path), self._weak(self._view(cur[i]), v, name, pos))
                else:
                    if isinstance(v, list):
                        self.fail("an array is assigned to one element of %s" % name, pos)
                    self._store_leaf(slot, cur, i, tuple(path), v)
            else:
                cur = cur[i]

    def _is_outer(self, slot):
        return any(id(slot) in vis for vis in self.region_visible)

    def _store_leaf(self, slot, container, i, path, v):
        old = container[i]
        if isinstance(old, Cell):
            x = self.rt(v)
            if not (x.kind == 'r' and x.val == old.reg):
                self.f.code.append((O.COPY, old.reg, (x.kind, x.val), None))
            if self.scouting:
                log = self.scout_log.setdefault(id(slot), set())
                if log != "rt":
                    log.add(path)
            return
        if self._is_outer(slot):
            raise _Repin()                     # an element this region's scout did not see assigned
        container[i] = v

    def _store_elements(self, slot, container, path, v):
        """v (same shape as container, leaves already owned) into the elements of an array variable"""
        for i in range(len(container)):
            if isinstance(container[i], list):
                self._store_elements(slot, container[i], path + (i,), v[i])
            else:
                x = v[i]
                old = container[i]
                if isinstance(old, Cell) and isinstance(x, RtVar) and x.kind == 'r' and x.val == old.reg:
                    continue                   # (a position the weak rule left as it was)
                if not isinstance(old, Cell) and (x is old or (isinstance(x, int) and isinstance(old, int) and x == old)):
                    continue
                self._store_leaf(slot, container, i, path + (i,), x)

    def _weak(self, old, v, name, pos):
        """arrays of different lengths into a variable (memory_slice.rs:129-160): overlapping positions only"""
        so, sv = X._shape(old), X._shape(v)
        if so == sv:
            return v
        if len(so) != len(sv) or not so:
            self.fail("assignee and assigned arrays of %s have different numbers of dimensions" % name, pos)

        def merge(o, x):
            if not isinstance(o, list):
                return x
            return [merge(o[i], x[i]) if i < len(x) else o[i] for i in range(len(o))]
        return merge(old, v)

    def _store_pinned(self, p: Pinned, idxs, v, pos):
        f = self.f
        if isinstance(v, Pinned):
            v = self._unpin_view(v)
        if all(self.known(i) for i in idxs):
            base, dims = p.base, list(p.dims)
            for i in idxs:
                i = self.kval(i)
                if not dims:
                    self.fail("too many indices", pos)
                if i >= dims[0]:
                    self.fail("array index out of bounds", pos)
                stride = 1
                for d in dims[1:]:
                    stride *= d
                base += i * stride
                dims.pop(0)
            if X._shape(v) != tuple(dims):
                # variables are not strict: the overlapping positions are assigned, the others keep their registers
                def view(b, dd):
                    if not dd:
                        return RtVar(self.f, 'r', b)
                    st = 1
                    for d in dd[1:]:
                        st *= d
                    return [view(b + k * st, dd[1:]) for k in range(dd[0])]
                v = self._weak(view(base, dims), v, "the array", pos)
            leaves = X._flat(v, []) if isinstance(v, list) else [v]
            # an array assigned from (a view of) itself must not be overwritten while it is read: stage through temporaries
            srcs = [self.rt(x) for x in leaves]
            dst = set(range(base, base + len(leaves)))
            if any(x.kind == 'r' and x.val in dst and x.val != base + n for n, x in enumerate(srcs)):
                srcs = [f.var(x) for x in srcs]
            for n, x in enumerate(srcs):
                if x.kind == 'r' and x.val == base + n:
                    continue
                f.code.append((O.COPY, base + n, (x.kind, x.val), None))
            return
        if isinstance(v, list):
            self.fail("an array indexed by a run-time value must be indexed down to one element", pos)
        lin = self._linear_index(p.dims, idxs, pos)
        RtArray(f, p.base, p.size).store(self.rt(lin), self.rt(v))

    # .. regions ..
    def _assigned_names(self, s, out):
        k = s[0]
        if k in ("block", "seq"):
            for x in s[1]:
                self._assigned_names(x, out)
        elif k == "subst":
            if s[1][0] == "var":
                out.add(s[1][1])
        elif k == "if":
            self._assigned_names(s[2], out)
            if s[3] is not None:
                self._assigned_names(s[3], out)
        elif k == "while":
            self._assigned_names(s[2], out)
        return out

    def _assigned_slots(self, stmts):
        names = set()
        for s in stmts:
            if s is not None:
                self._assigned_names(s, names)
        out = []
        for nm in names:
            for sc in reversed(self.scopes):
                slot = sc.get(nm)
                if slot is not None:
                    out.append(slot)
                    break
        return out

    def _pin_leaves(self, slot, which):
        """which: None = every element, else a set of element paths"""
        def rec(v, path):
            for i in range(len(v)):
                if isinstance(v[i], list):
                    rec(v[i], path + (i,))
                elif not isinstance(v[i], Cell) and (which is None or path + (i,) in which):
                    r = self.f.var(self.rt(v[i]))
                    self.pinned_regs.add(r.val)
                    v[i] = Cell(r.val)
        rec(slot.value, ())

    def _snapshot(self):
        cp = lambda v: [cp(x) for x in v] if isinstance(v, list) else v
        slots = [(sl, cp(sl.value)) for fr in self.frames for sc in fr.scopes for sl in sc.values()]
        return (len(self.f.code), self.f.n_regs, self.f.last_if, set(self.pinned_regs), self.region,
                [(fr, fr.ret_shape, fr.ret_base, list(fr.jumps)) for fr in self.frames], slots)

    def _restore(self, snap):
        cp = lambda v: [cp(x) for x in v] if isinstance(v, list) else v
        n_code, n_regs, last_if, pinned, region, frames, slots = snap
        del self.f.code[n_code:]
        self.f.n_regs = n_regs
        self.f.last_if = last_if
        self.pinned_regs = set(pinned)
        self.region = region
        for fr, shape, base, jumps in frames:
            fr.ret_shape, fr.ret_base, fr.jumps = shape, base, list(jumps)
        for sl, val in slots:
            sl.value = cp(val)

    def _region(self, stmts, build):
        """Build a run-time region.  Scalars it assigns are pinned to a register; of the ARRAYS it assigns only the elements
        it assigns are (circom-ecdsa's functions carry 100-entry arrays of which a handful of entries is ever touched): the
        region is built once as a scout with every element pinned, the builder's state is rolled back, and the region is
        built again with the elements the scout saw assigned; an array indexed by a run-time value becomes a block."""
        slots = self._assigned_slots(stmts)
        arrays = [sl for sl in slots if isinstance(sl.value, list)]
        for sl in slots:
            if not isinstance(sl.value, (list, Pinned)):
                sl.value = self._pin_value(sl.value)
        if not arrays or self.scouting:
            for sl in arrays:
                self._pin_leaves(sl, None)
            build()
            return
        snap = self._snapshot()
        self.scouting, self.scout_log = True, {}
        try:
            for sl in arrays:
                self._pin_leaves(sl, None)
            build()
        finally:
            self.scouting = False
        log = self.scout_log
        self._restore(snap)
        for sl in arrays:
            seen = log.get(id(sl))
            if seen == "rt":
                sl.value = self._pin_value(self._view(sl.value))
            elif seen:
                self._pin_leaves(sl, seen)
        self.region_visible.append({id(sl) for fr in self.frames for sc in fr.scopes for sl in sc.values()})
        try:
            try:
                build()
            finally:
                self.region_visible.pop()
        except _Repin:
            self._restore(snap)
            for sl in arrays:
                self._pin_leaves(sl, None)
            build()

    def exec_if(self, s):
        _, cond, then, other, pos = s
        c = self.eval(cond)
        if isinstance(c, (list, Pinned)):
            self.fail("condition is an array", pos)
        if self.known(c):
            if self.kval(c) != 0:
                self.run_block([then])
            elif other is not None:
                self.run_block([other])
            return
        f = self.f
        cr = self.rt(c)

        def build():
            self.region += 1
            with f.if_(cr):
                self.run_block([then])
            if other is not None:
                with f.else_():
                    self.run_block([other])
            self.region -= 1
        self._region([then, other], build)

    def exec_while(self, s):
        _, cond, body, pos = s
        # a loop whose condition is known now may still become a run-time loop after its body pins the counter: decide on
        # the first evaluation, and keep unrolling only while the condition stays known
        n = 0
        start = len(self.f.code)
        while True:
            c = self.eval(cond)
            if isinstance(c, (list, Pinned)):
                self.fail("condition is an array", pos)
            if not self.known(c):
                break
            if self.kval(c) == 0:
                return
            if n and len(self.f.code) - start > UNROLL_BUDGET:
                break               # unrolling on would multiply a large body: the remaining trips run as a run-time loop
            self.run_block([body])
            n += 1
            if n > self.w.max_loop:
                self.fail("loop does not terminate", pos)
        # run-time loop: everything the body or the condition's variables assign lives in registers from here on
        f = self.f

        def build():
            self.region += 1
            with f.loop() as L:
                c = self.eval(cond)
                if self.known(c):
                    # the pinning turned the condition's variables into registers: it cannot be known any more unless it
                    # does not depend on them at all
                    if self.kval(c) == 0:
                        self.fail("unexpected constant-false loop condition", pos)
                    c = f.var(1)
                L.break_unless(self.rt(c))
                self.run_block([body])
            self.region -= 1
        self._region([body], build)

    def exec_return(self, s):
        fr = self.frames[-1]
        v = self.eval(s[1])
        if isinstance(v, Pinned):
            v = self._unpin_view(v)
        shape = X._shape(v)
        f = self.f
        top = self.region == fr.region0
        if top and fr.ret_shape is None:
            # no path left the function earlier: its value is this one, whatever registers or constants it is made of
            fr.direct = (self._view(v),)
            raise _Stop()
        if fr.ret_shape is None:
            fr.ret_shape = shape
            n = 1
            for d in shape:
                n *= d
            fr.ret_base = f.n_regs
            f.n_regs += n
        elif shape != fr.ret_shape:
            self.fail("the returns of function %s have different sizes" % fr.name, s[-1])
        leaves = X._flat(v, []) if isinstance(v, list) else [v]
        for n, x in enumerate(leaves):
            x = self.rt(x)
            f.code.append((O.COPY, fr.ret_base + n, (x.kind, x.val), None))
        fr.jumps.append(len(f.code))
        f.code.append([F_JMP, None, None, None])
        if top:
            raise _Stop()


def call_runtime_function(world, caller, name, vals, pos):
    """compile (once per specialisation) and call: returns the function's value shaped like its `return`"""
    ar = world.archive
    d = ar.functions[name]
    params = d[2]
    # the specialisation: known leaves are baked in, unknown leaves become argument registers in flattening order
    unknown = []

    def mark(v):
        if isinstance(v, list):
            return tuple(mark(x) for x in v)
        if isinstance(v, int):
            return v
        unknown.append(v)
        return None
    key = (name, tuple(mark(v) for v in vals))
    n_args = len(unknown)
    cache = world._rt_functions
    ent = cache.get(key)
    if ent is None:
        fname = "%s$%d" % (name, sum(1 for k in cache if k[0] == name))
        info = {}

        def build(f, *regs):
            it = iter(regs)

            def fill(v):
                if isinstance(v, list):
                    return [fill(x) for x in v]
                if isinstance(v, int):
                    return v
                return next(it)
            args = [fill(v) for v in vals]
            rc = RtCompiler(world, f, pos)
            rc.frames.append(_Frame("<call>"))
            out = rc.inline(name, args, pos)
            info["shape"] = X._shape(out)
            leaves = X._flat(out, []) if isinstance(out, list) else [out]
            return [rc.rt(x) for x in leaves]
        fn = caller.ctx.function(fname, n_args, build)
        ent = (fn, info["shape"])
        cache[key] = ent
    fn, shape = ent
    res = caller.ctx.call(fn, unknown)

    def shape_up(flat, dims):
        if not dims:
            return flat.pop(0)
        return [shape_up(flat, dims[1:]) for _ in range(dims[0])]
    return shape_up(list(res), list(shape))

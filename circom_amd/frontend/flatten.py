"""Flatten the traced component tree into one circuit-wide signal table, op list and constraint list.

Mirrors what the reference's construction phase exports at `--O0`:
  * global signal numbering = preorder walk of the component tree, slot 0 = the constant 1, main's
    block starts at 1 (dag/src/lib.rs:328-371, dag/src/witness_producer.rs:3-19, calcwit.cpp:34);
    at --O0 the witness list is the identity over all signals ("--O0 identity"),
  * constraints in tree preorder: a node's own constraints, then each child's subtree
    (matches the golden `basic.circom --O0` listing in mkdocs/docs/circom-language/formats/constraints-json.md),
  * the witness code in *execution* order: a child's code is spliced in at the point where the parent
    stores its last input (store_bucket.rs:660-735).
"""
from __future__ import annotations

import sys

import numpy as np

from .. import opcodes as O
from .dsl import Program, TemplateInstance, CONST_KEY, _prod

K_SIG, K_TMP = O.K_SIG, O.K_TMP


class FlatCircuit:
    def __init__(self, prog: Program):
        self.prog = prog
        self.prime = prog.prime
        self.fp = prog.fp
        m = prog.main
        self.n_signals = 1 + m.n_total
        self.n_components = m.n_components
        self.constants = prog.constants
        self.functions = [dict(f.as_data(), consts=prog.constants) for f in prog.functions]   # circom functions (rtcode.py)
        self.n_outputs = m.n_out
        self.n_pub_in = prog.n_public_inputs
        self.n_prv_in = m.n_in - self.n_pub_in
        self.main_input_start = 1 + m.n_out                   # get_main_input_signal_start, c_elements/mod.rs:156-158
        self.n_main_inputs = m.n_in
        self.inputs = [(n, 1 + off, size) for n, off, size in m.input_names]
        self.input_dims = {n: d for n, d, o in m.decls["i"]}
        # component table in preorder
        self.comp_inst = []
        self.comp_sigstart = []
        self.comp_father = []
        self.comp_name = []
        self._flatten()

    # ------------------------------------------------------------------------------------------
    def _flatten(self):
        sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
        chunks = {k: [] for k in ("op", "dk", "dv", "ak", "av", "bk", "bv", "ck", "cv")}
        cons = []
        # per-instance cached pieces
        cache = {}

        def prep(inst: TemplateInstance):
            c = inst.code
            op = c["op"]
            runs = np.nonzero(op == O.RUN)[0]
            segs = []
            start = 0
            bounds = list(runs) + [len(op)]
            for r in bounds:
                if r > start:
                    sl = slice(start, r)
                    seg = {"op": op[sl]}
                    for kk, vv in (("dk", "dv"), ("ak", "av"), ("bk", "bv"), ("ck", "cv")):
                        k = c[kk][sl]
                        seg[kk] = k
                        seg[vv] = c[vv][sl]
                        seg[vv + "_s"] = (k == K_SIG).astype(np.int64)
                        seg[vv + "_t"] = (k == K_TMP).astype(np.int64)
                    segs.append(("seg", seg))
                if r < len(op):
                    segs.append(("run", int(c["av"][r])))
                start = r + 1
            cache[inst.id] = segs
            return segs

        # Component numbering must be preorder over the *sorted* child table, while code order follows
        # execution.  Do it in two passes: (1) number + record constraints in preorder, (2) emit code.
        order = []  # (inst, base, father, name)

        def number(inst, base, father, name):
            me = len(order)
            order.append((inst, base, father, name))
            for a, b, c in inst.constraints:
                cons.append((self._reloc(a, base), self._reloc(b, base), self._reloc(c, base)))
            for cname, cidx, cinst, soff, coff in inst.children:
                assert len(order) == me + coff
                number(cinst, base + soff, me, cname + "".join("[%d]" % i for i in cidx))

        number(self.prog.main, 1, 0, "main")
        for inst, base, father, name in order:
            self.comp_inst.append(inst.id)
            self.comp_sigstart.append(base)
            self.comp_father.append(father)
            self.comp_name.append(name)
        # temp bases per component (preorder)
        tbase = np.zeros(len(order) + 1, dtype=np.int64)
        for i, (inst, _, _, _) in enumerate(order):
            tbase[i + 1] = tbase[i] + inst.n_temps
        self.n_temps = int(tbase[-1])

        # rows of the flat code that a component's whole subtree occupies: one contiguous range (a child's code is spliced in
        # where it fires).  The code emitters use it to find the instances of a repeated template (hip_elements/bitjit.py loops)
        n_emitted = [0]
        self.comp_code_range = [(0, 0)] * len(order)

        def emit(ci):
            inst, base, _, _ = order[ci]
            tb = int(tbase[ci])
            segs = cache.get(inst.id)
            if segs is None:
                segs = prep(inst)
            first = n_emitted[0]
            for kind, item in segs:
                if kind == "seg":
                    n_emitted[0] += len(item["op"])
                    chunks["op"].append(item["op"])
                    for kk, vv in (("dk", "dv"), ("ak", "av"), ("bk", "bv"), ("ck", "cv")):
                        chunks[kk].append(item[kk])
                        chunks[vv].append(item[vv] + base * item[vv + "_s"] + tb * item[vv + "_t"])
                else:
                    emit(ci + inst.children[item][4])
            self.comp_code_range[ci] = (first, n_emitted[0])

        emit(0)
        self.code = {k: (np.concatenate(v) if v else np.zeros(0, dtype=np.int64)) for k, v in chunks.items()}
        self.constraints = cons
        self.io_map = self._build_io_map()
        self.bus_field_map = list(getattr(self.prog, "bus_field_map", ()))   # (the text front-end's buses; the eDSL has none)
        self.log_strings = list(getattr(self.prog, "log_strings", ()))
        self.n_log_values = int(((self.code["op"] == O.LOG) & (self.code["ak"] != O.K_NONE)).sum())

    def log_program(self):
        return log_program(self)

    def code_with_log_copies(self):
        return code_with_log_copies(self)

    def _build_io_map(self):
        """TemplateInstanceIOMap of compiler/src/circuit_design/build.rs:488-520: the component arrays of a template whose
        elements are instances of ONE template name with DIFFERENT parameters are `Mixed` clusters (translate.rs:1017-1045):
        the reference addresses their signals through a run-time table - per template instance, per input/output signal
        (code = position among the template's wires: outputs, then inputs): local offset, array lengths, element size, bus
        id - which it reads from the tail of the `.dat` (c_code_generator.rs:681-738, main.cpp:60-92), and runs them
        through `_functionTable[templateId]` (store_bucket.rs:706-710).  Here every access is resolved at trace time, so the
        table only matters for the files: `write_dat` emits it, `cw_load` validates it, and the reference runtime executes
        the oracle's emitted C++ THROUGH it (oracle/emit_ref_cpp.py).
        Sets `inst.mixed_children` (child table positions) on every instance; returns [(template id, [(offset, dims, size,
        bus id)])] sorted by template id."""
        mixed_templates = {}
        for inst in self.prog.inst_list:
            groups = {}
            for k, (cname, cidx, cinst, soff, coff) in enumerate(inst.children):
                groups.setdefault(cname, []).append((k, cinst))
            inst.mixed_children = set()
            for cname, members in groups.items():
                if len({id(ci) for _, ci in members}) > 1:
                    for k, ci in members:
                        inst.mixed_children.add(k)
                        mixed_templates[ci.id] = ci
        out = []
        for tid in sorted(mixed_templates):
            ci = mixed_templates[tid]
            defs = []
            for cat in ("o", "i"):
                for name, dims, pid0 in ci.decls[cat]:
                    defs.append((int(pid0), tuple(int(d) for d in dims), 1, 0))
            out.append((tid, defs))
        return out

    @staticmethod
    def _reloc(d, base):
        return {(0 if k == CONST_KEY else k + base): v for k, v in d.items()}

    # ------------------------------------------------------------------------------------------
    def signal_names(self):
        """Qualified names in .sym order (constraint_writers/src/sym_writer.rs; docs formats/sym.md)."""
        names = ["one"] + [None] * (self.n_signals - 1)
        insts = self.prog.inst_list

        def arr_names(prefix, name, dims):
            if not dims:
                yield prefix + name
                return
            idx = [0] * len(dims)
            total = _prod(dims)
            for _ in range(total):
                yield prefix + name + "".join("[%d]" % i for i in idx)
                for d in range(len(dims) - 1, -1, -1):
                    idx[d] += 1
                    if idx[d] < dims[d]:
                        break
                    idx[d] = 0

        # rebuild full qualified component paths
        paths = [None] * len(self.comp_inst)
        for ci in range(len(self.comp_inst)):
            f = self.comp_father[ci]
            paths[ci] = "main" if ci == 0 else paths[f] + "." + self.comp_name[ci]
        for ci, iid in enumerate(self.comp_inst):
            inst = insts[iid]
            base = self.comp_sigstart[ci]
            pre = paths[ci] + "."
            for cat in ("o", "i", "m"):
                for name, dims, off in inst.decls[cat]:
                    for j, nm in enumerate(arr_names(pre, name, dims)):
                        names[base + off + j] = nm
        return names


def log_program(fc):
    """The `log(...)` statements in execution order: [(index of the flat operation that ends the statement, [item])],
    item = ("s", string id) | ("v", j): the j-th logged value of the program (LOG rows with an operand, in order)."""
    op, ak, av, dv = fc.code["op"], fc.code["ak"], fc.code["av"], fc.code["dv"]
    out, items, j = [], [], 0
    for i in np.nonzero(op == O.LOG)[0]:
        if ak[i] != O.K_NONE:
            items.append(("v", j))
            j += 1
        elif av[i] >= 0:
            items.append(("s", int(av[i])))
        if dv[i]:
            out.append((int(i), items))
            items = []
    assert not items
    return out


def code_with_log_copies(fc):
    """The flat code as the lowering sees it: a LOG row with an operand is a COPY of that operand into a HIDDEN signal
    (ids n_signals .. n_signals + n_log_values - 1: table slots that are no witness elements), the other LOG rows
    are RUN markers (ignored).  Row indices are unchanged (failure reports name flat operations by index)."""
    code = {k: v.copy() for k, v in fc.code.items()}
    j = 0
    for i in np.nonzero(code["op"] == O.LOG)[0]:
        if code["ak"][i] != O.K_NONE:
            code["op"][i] = O.COPY
            code["dk"][i], code["dv"][i] = K_SIG, fc.n_signals + j
            j += 1
        else:
            code["op"][i] = O.RUN
            code["dk"][i], code["dv"][i] = O.K_NONE, 0
    return code, j


def flatten(prog: Program) -> FlatCircuit:
    return FlatCircuit(prog)

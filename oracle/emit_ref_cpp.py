"""Emit a reference-style `<name>.cpp` for a traced circuit so that the *reference's own C++ runtime*
(main.cpp / calcwit.cpp / generic fr.cpp, compiled by oracle/Makefile) computes the witness.

TEST INFRASTRUCTURE ONLY.  The Rust compiler that normally emits this file cannot be built here; this is
a restatement of its emission rules (SURVEY Appendix A derives them):
  prologue + get_*() table ......... compiler/src/circuit_design/circuit.rs:420-493
  <T>_<id>_create .................. circuit_design/template.rs:177-279
  <T>_<id>_run ..................... circuit_design/template.rs:281-472 (sub-components created at the top,
                                     translate.rs:1960; children in sorted order)
  Store / sub-component trigger .... intermediate_representation/store_bucket.rs:442-851
  Compute -> Fr_* .................. compute_bucket.rs:315-341
  Assert ........................... assert_bucket.rs:70-89
  run() ............................ circuit.rs:526-561
Differences (do not change computed values): loops are already unrolled by the trace, so bodies are
straight-line; long bodies are split into chunk functions to keep g++ compile time linear.
"""
from __future__ import annotations

from pathlib import Path

# operator numbering of circom_amd/opcodes.py (literal: the oracle has no product import)
COPY, ADD, SUB, MUL, DIV, IDIV, MOD, POW, NEG = range(9)
SHL, SHR, BAND, BOR, BXOR, BNOT = range(9, 15)
LT, GT, LEQ, GEQ, EQ, NEQ, LAND, LOR, LNOT = range(15, 24)
SELECT, ASSERT_EQ, ASSERT_NZ, RUN, CALL, LOG = range(24, 30)
F_JZ, F_JMP, F_LDX, F_STX, F_RET = 100, 101, 102, 103, 104      # circom_amd/frontend/rtcode.py
K_SIG, K_TMP, K_CONST, K_NONE = 0, 1, 2, 3
SYM = {ADD: "Fr_add", SUB: "Fr_sub", MUL: "Fr_mul", DIV: "Fr_div", IDIV: "Fr_idiv", MOD: "Fr_mod", POW: "Fr_pow",
       SHL: "Fr_shl", SHR: "Fr_shr", BAND: "Fr_band", BOR: "Fr_bor", BXOR: "Fr_bxor", LT: "Fr_lt", GT: "Fr_gt",
       LEQ: "Fr_leq", GEQ: "Fr_geq", EQ: "Fr_eq", NEQ: "Fr_neq", LAND: "Fr_land", LOR: "Fr_lor"}
SYM1 = {NEG: "Fr_neg", BNOT: "Fr_bnot", LNOT: "Fr_lnot", COPY: "Fr_copy"}

CHUNK = 1500
ARGS_DECL = "Circom_CalcWit* ctx, FrElement* signalValues, FrElement* circuitConstants, FrElement* expaux, u64 mySignalStart, u32* mySubcomponents, u64 myId"
ARGS_CALL = "ctx, signalValues, circuitConstants, expaux, mySignalStart, mySubcomponents, myId"
# The 64-bit runtime (`--prime goldilocks`: goldilocks/fr.hpp + common64/{main,calcwit}.cpp) holds field elements as plain
# u64 VALUES: the reference emits `lvar[i] = Fr_add(a, b);` instead of `Fr_add(&lvar[i], &a, &b);`, constants as literals
# (value_bucket.rs:82-86) and no constant table (compute_bucket.rs:353, store_bucket.rs:575-657, call_bucket.rs:474-533).
W64 = False
ARGS_DECL64 = "Circom_CalcWit* ctx, u64* signalValues, u64* expaux, u64 mySignalStart, u32* mySubcomponents, u64 myId"
ARGS_CALL64 = "ctx, signalValues, expaux, mySignalStart, mySubcomponents, myId"
CONSTS64 = []


def _child_of(inst, off):
    """(child index in sorted table, offset inside the child) for a subtree offset >= n_local."""
    ch = inst.children
    lo, hi = 0, len(ch) - 1
    while lo < hi:
        mid = (lo + hi + 1) // 2
        if ch[mid][3] <= off:
            lo = mid
        else:
            hi = mid - 1
    return lo, off - ch[lo][3]


def _mapped(child, o):
    """(signal code, index expression) of offset `o` inside a child of a Mixed cluster: LocationRule::Mapped
    (store_bucket.rs:498-566, load_bucket.rs:264-330): code = position among the template's wires (outputs, inputs),
    multi-dimensional indices are folded with the lengths the RUNTIME holds (`cur_def->lengths[i-1]`)."""
    code = 0
    for cat in ("o", "i"):
        for name, dims, pid0 in child.decls[cat]:
            size = 1
            for d in dims:
                size *= d
            if pid0 <= o < pid0 + size:
                rem = o - pid0
                idx = []
                for d in reversed(dims):
                    idx.append(rem % d)
                    rem //= d
                idx.reverse()
                if not idx:
                    return code, None
                e = "%d" % idx[0]
                for i in range(1, len(idx)):
                    e = "(%s)*cur_def->lengths[%d]+%d" % (e, i - 1, idx[i])
                return code, e
            code += 1
    raise ValueError("offset %d is no input/output of %s" % (o, child.header))


def _ref(inst, k, v):
    if W64:
        if k == K_CONST:
            return "%dull" % CONSTS64[v]
        r = _ref_ptr(inst, k, v)
        return r[1:] if r.startswith("&") else "(*%s)" % r
    return _ref_ptr(inst, k, v)


def _ref_ptr(inst, k, v):
    if k == K_SIG:
        if v < inst.n_local:
            return "&signalValues[mySignalStart + %d]" % v
        ci, o = _child_of(inst, v)
        if ci in getattr(inst, "mixed_children", ()):
            code, idx = _mapped(inst.children[ci][2], o)
            cm = "ctx->componentMemory[mySubcomponents[%d]]" % ci
            base = "%s.signalStart + ctx->templateInsId2IOSignalInfo[%s.templateId].defs[%d].offset" % (cm, cm, code)
            if idx is None:
                return "&ctx->signalValues[%s]" % base
            # a GNU statement expression keeps the reference's shape (cur_def, then the folded index times the element size)
            return ("({ IOFieldDef *cur_def = &(ctx->templateInsId2IOSignalInfo[%s.templateId].defs[%d]); "
                    "&ctx->signalValues[%s + (%s)*cur_def->size]; })" % (cm, code, base, idx))
        return "&ctx->signalValues[ctx->componentMemory[mySubcomponents[%d]].signalStart + %d]" % (ci, o)
    if k == K_TMP:
        return "&expaux[%d]" % v
    if k == K_CONST:
        return "&circuitConstants[%d]" % v
    raise ValueError(k)


def _emit_instance(inst, out):
    h = inst.header
    code = inst.code
    n = len(code["op"])
    op = code["op"].tolist()
    dk, dv = code["dk"].tolist(), code["dv"].tolist()
    ak, av = code["ak"].tolist(), code["av"].tolist()
    bk, bv = code["bk"].tolist(), code["bv"].tolist()
    ck, cv = code["ck"].tolist(), code["cv"].tolist()
    stmts = []
    for i in range(n):
        o = op[i]
        if o == RUN:
            ci = av[i]
            child = inst.children[ci][2]
            if child.n_in == 0:
                continue        # ran at creation (template.rs:274-278)
            if ci in getattr(inst, "mixed_children", ()):       # store_bucket.rs:706-710: through the function table
                stmts.append("assert(!(ctx->componentMemory[mySubcomponents[%d]].inputCounter)); "
                             "(*_functionTable[ctx->componentMemory[mySubcomponents[%d]].templateId])(mySubcomponents[%d],ctx);" % (ci, ci, ci))
                continue
            stmts.append("assert(!(ctx->componentMemory[mySubcomponents[%d]].inputCounter)); %s_run(mySubcomponents[%d],ctx);"
                         % (ci, child.header, ci))
            continue
        if o == LOG:                # LogBucket, statement by statement as log_bucket.rs:105-162 prints it
            if ak[i] != 3:
                stmts.append("{ char* temp = Fr_element2str(%s); printf(\"%%s\",temp); delete [] temp; }" % _ref(inst, ak[i], av[i]))
            elif av[i] >= 0:
                text = inst.prog.log_strings[av[i]].replace("\\", "\\\\").replace("\"", "\\\"").replace("\n", "\\n")
                stmts.append("{ printf(\"%s\"); }" % text)
            stmts.append("{ printf(\"\\n\"); }" if dv[i] else "{ printf(\" \"); }")
            continue
        if o == CALL:               # CallBucket (call_bucket.rs:466-533): the callee works on its own lvar arena
            if W64:
                raise ValueError("run-time functions are not emitted for the 64-bit runtime")
            stmts.append("rtfn_%d(ctx,&expaux[%d]);" % (av[i], bv[i]))
            continue
        if o == ASSERT_EQ or o == ASSERT_NZ:
            if o == ASSERT_EQ and W64:
                cond = "aux_assert = Fr_eq(%s,%s);" % (_ref(inst, ak[i], av[i]), _ref(inst, bk[i], bv[i]))
                test = "aux_assert"
            elif o == ASSERT_EQ:
                cond = "Fr_eq(&aux_assert,%s,%s);" % (_ref(inst, ak[i], av[i]), _ref(inst, bk[i], bv[i]))
                test = "&aux_assert"
            else:
                cond = ""
                test = _ref(inst, ak[i], av[i])
            stmts.append("{ " + ("u64" if W64 else "FrElement") + " aux_assert; %s if (!Fr_isTrue(%s)) { std::cout << \"Failed assert in template/function \" "
                         "<< \"%s\" << \" op %d\" << std::endl; std::cout << \"Followed trace of components: \" << "
                         "ctx->getTrace(myId) << std::endl; assert(false); } }" % (cond, test, inst.name, i))
            continue
        dst = _ref(inst, dk[i], dv[i])
        if W64 and o in SYM:
            s = "%s = %s(%s,%s);" % (dst, SYM[o], _ref(inst, ak[i], av[i]), _ref(inst, bk[i], bv[i]))
        elif W64 and o == COPY:
            s = "%s = %s;" % (dst, _ref(inst, ak[i], av[i]))
        elif W64 and o in SYM1:
            s = "%s = %s(%s);" % (dst, SYM1[o], _ref(inst, ak[i], av[i]))
        elif W64 and o == SELECT:
            s = "if (Fr_isTrue(%s)) { %s = %s; } else { %s = %s; }" % (
                _ref(inst, ak[i], av[i]), dst, _ref(inst, bk[i], bv[i]), dst, _ref(inst, ck[i], cv[i]))
        elif o in SYM:
            s = "%s(%s,%s,%s);" % (SYM[o], dst, _ref(inst, ak[i], av[i]), _ref(inst, bk[i], bv[i]))
        elif o in SYM1:
            s = "%s(%s,%s);" % (SYM1[o], dst, _ref(inst, ak[i], av[i]))
        elif o == SELECT:
            s = "if (Fr_isTrue(%s)) { Fr_copy(%s,%s); } else { Fr_copy(%s,%s); }" % (
                _ref(inst, ak[i], av[i]), dst, _ref(inst, bk[i], bv[i]), dst, _ref(inst, ck[i], cv[i]))
        else:
            raise ValueError("op %d" % o)
        if dk[i] == K_SIG and dv[i] >= inst.n_local:
            ci, _ = _child_of(inst, dv[i])
            s += " ctx->componentMemory[mySubcomponents[%d]].inputCounter -= 1;" % ci   # store_bucket.rs:663-670
        stmts.append(s)

    nsub = len(inst.children)
    # ---- create (template.rs:177-279) ----
    out.append("void %s_create(uint soffset,uint coffset,Circom_CalcWit* ctx,std::string componentName,uint componentFather){" % h)
    out.append("ctx->componentMemory[coffset].templateId = %d;" % inst.id)
    out.append("ctx->componentMemory[coffset].templateName = \"%s\";" % inst.name)
    out.append("ctx->componentMemory[coffset].signalStart = soffset;")
    out.append("ctx->componentMemory[coffset].inputCounter = %d;" % inst.n_in)
    out.append("ctx->componentMemory[coffset].componentName = componentName;")
    out.append("ctx->componentMemory[coffset].idFather = componentFather;")
    out.append("ctx->componentMemory[coffset].subcomponents = new uint[%d]%s;" % (nsub, "{0}" if nsub else ""))
    if inst.n_in == 0:
        out.append("%s_run(coffset,ctx);" % h)
    out.append("}")
    # ---- body chunks ----
    chunks = [stmts[i:i + CHUNK] for i in range(0, len(stmts), CHUNK)] or [[]]
    for ci, ch in enumerate(chunks):
        out.append("static void %s_body%d(%s){" % (h, ci, ARGS_DECL64 if W64 else ARGS_DECL))
        out.extend(ch)
        out.append("}")
    # ---- run (template.rs:281-472) ----
    out.append("void %s_run(uint ctx_index,Circom_CalcWit* ctx){" % h)
    if W64:
        out.append("u64* signalValues = ctx->signalValues;")
        out.append("std::vector<u64> expaux_v(%d);" % max(inst.n_temps, 1))
        out.append("u64* expaux = expaux_v.data();")
    else:
        out.append("FrElement* circuitConstants = ctx->circuitConstants;")
        out.append("FrElement* signalValues = ctx->signalValues;")
        out.append("std::vector<FrElement> expaux_v(%d);" % max(inst.n_temps, 1))
        out.append("FrElement* expaux = expaux_v.data();")
    out.append("u64 mySignalStart = ctx->componentMemory[ctx_index].signalStart;")
    out.append("u64 myId = ctx_index;")
    out.append("u32* mySubcomponents = ctx->componentMemory[ctx_index].subcomponents;")
    for k, (cname, cidx, cinst, soff, coff) in enumerate(inst.children):
        nm = cname + "".join("[%d]" % i for i in cidx)
        out.append("mySubcomponents[%d] = ctx_index + %d; %s_create(mySignalStart + %d, ctx_index + %d, ctx, \"%s\", myId);"
                   % (k, coff, cinst.header, soff, coff, nm))
    for ci in range(len(chunks)):
        out.append("%s_body%d(%s);" % (h, ci, ARGS_CALL64 if W64 else ARGS_CALL))
    out.append("for (uint i = 0; i < %d; i++){" % nsub)
    out.append("uint index_subc = ctx->componentMemory[ctx_index].subcomponents[i];")
    out.append("if (index_subc != 0){ assert(!(ctx->componentMemory[index_subc].inputCounter)); release_memory_component(ctx,index_subc); }")
    out.append("}")
    out.append("}")


def _emit_function(fid, fn, out):
    """A circom function with run-time control flow (function.rs:91-127 emits `while (Fr_isTrue(..))` / `if`; the same
    control flow is printed here with labels and gotos): registers = the callee's lvar arena, array addresses through
    Fr_toInt exactly as compute_bucket.rs:361-363 does (no bounds check there either)."""
    def opnd(x):
        return "&lvar[%d]" % x[1] if x[0] == 'r' else "&circuitConstants[%d]" % x[1]
    out.append("static void rtfn_%d(Circom_CalcWit* ctx, FrElement* lvar){" % fid)
    out.append("FrElement* circuitConstants = ctx->circuitConstants;")
    for pc, (op, d, a, b) in enumerate(fn["code"]):
        lab = "L%d: " % pc
        if op == F_RET:
            out.append(lab + "return;")
        elif op == F_JMP:
            out.append(lab + "goto L%d;" % d)
        elif op == F_JZ:
            out.append(lab + "if (!Fr_isTrue(%s)) goto L%d;" % (opnd(a), d))
        elif op == F_LDX:
            out.append(lab + "Fr_copy(&lvar[%d],&lvar[%d + Fr_toInt(&lvar[%d])]);" % (d, a, b[0]))
        elif op == F_STX:
            out.append(lab + "Fr_copy(&lvar[%d + Fr_toInt(&lvar[%d])],%s);" % (d, b[0], opnd(a)))
        elif op in SYM:
            out.append(lab + "%s(&lvar[%d],%s,%s);" % (SYM[op], d, opnd(a), opnd(b)))
        elif op in SYM1:
            out.append(lab + "%s(&lvar[%d],%s);" % (SYM1[op], d, opnd(a)))
        else:
            raise ValueError("function opcode %d" % op)
    out.append("}")


def emit(fc, path, hashmap_size: int, n_witness=None):
    """fc: circom_amd FlatCircuit (duck-typed: .prog.inst_list, .prog.main, sizes)."""
    global W64, CONSTS64
    W64 = fc.prime == "goldilocks"
    CONSTS64 = [int(c) % fc.fp.q for c in fc.constants]
    prog = fc.prog
    insts = prog.inst_list
    out = ["#include <stdio.h>", "#include <iostream>", "#include <vector>", "#include <assert.h>",
           "#include \"circom.hpp\"", "#include \"calcwit.hpp\""] + (["#include \"fr.hpp\""] if W64 else [])
    for t in insts:
        out.append("void %s_create(uint soffset,uint coffset,Circom_CalcWit* ctx,std::string componentName,uint componentFather);" % t.header)
        out.append("void %s_run(uint ctx_index,Circom_CalcWit* ctx);" % t.header)
    out.append("Circom_TemplateFunction _functionTable[%d] = { %s };" % (len(insts), ", ".join(t.header + "_run" for t in insts)))
    out.append("Circom_TemplateFunction _functionTableParallel[%d] = { %s };" % (len(insts), ", ".join("NULL" for _ in insts)))
    m = prog.main
    out.append("uint get_main_input_signal_start() {return %d;}" % fc.main_input_start)
    out.append("uint get_main_input_signal_no() {return %d;}" % fc.n_main_inputs)
    out.append("uint get_total_signal_no() {return %d;}" % fc.n_signals)
    out.append("uint get_number_of_components() {return %d;}" % fc.n_components)
    out.append("uint get_size_of_input_hashmap() {return %d;}" % hashmap_size)
    # (--O0: every signal is a witness entry; a simplified system keeps fewer: the list itself is in the .dat)
    out.append("uint get_size_of_witness() {return %d;}" % (fc.n_signals if n_witness is None else n_witness))
    if not W64:
        out.append("uint get_size_of_constants() {return %d;}" % len(fc.constants))
    out.append("uint get_size_of_io_map() {return %d;}" % len(getattr(fc, "io_map", ())))
    out.append("uint get_size_of_bus_field_map() {return %d;}" % len(getattr(fc, "bus_field_map", ())))
    # generate_function_release_memory_component, c_code_generator.rs:914-933
    out.append("void release_memory_component(Circom_CalcWit* ctx, uint pos) {{ if (pos != 0){{ if(ctx->componentMemory[pos].subcomponents) "
               "delete []ctx->componentMemory[pos].subcomponents; ctx->componentMemory[pos].subcomponents = NULL; }} }}")
    out.append("// function declarations")
    for fid, fn in enumerate(getattr(fc, "functions", ())):
        _emit_function(fid, fn, out)
    out.append("// template declarations")
    for t in insts:
        _emit_instance(t, out)
    out.append("void run(Circom_CalcWit* ctx){")
    out.append("%s_create(1,0,ctx,\"main\",0);" % m.header)
    if m.n_in > 0:
        out.append("%s_run(0,ctx);" % m.header)
    out.append("}")
    Path(path).write_text("\n".join(out) + "\n")

"""Operator numbering shared by the front-end trace, the lowering and the device tape.

The operator set is the reference IR's `OperatorType` (compiler/src/intermediate_representation/
compute_bucket.rs:7-34) minus the address-arithmetic operators (ToAddress/MulAddress/AddAddress are
folded at trace time, as `ir_processing/reduce_stack.rs:28-50` does when indices are known), plus
COPY (StoreBucket), SELECT (a value-dependent BranchBucket lowered to predication), ASSERT_EQ /
ASSERT_NZ (AssertBucket), RUN (the point where a sub-component fires, store_bucket.rs:660-735) and CALL (CallBucket,
call_bucket.rs:466-533: a circom function with run-time control flow, frontend/rtcode.py; a = function id, b = first
of the consecutive temporaries that are the function's registers: arguments in, results out) and LOG (LogBucket,
log_bucket.rs:105-162: ONE argument of a `log(...)` statement per row - a = the value, or kind NONE with the string-table
index (-1: no argument) - and dv = 1 on the row that ends the statement; the device keeps the values, the host formats them).
"""

COPY, ADD, SUB, MUL, DIV, IDIV, MOD, POW, NEG = range(9)
SHL, SHR, BAND, BOR, BXOR, BNOT = range(9, 15)
LT, GT, LEQ, GEQ, EQ, NEQ, LAND, LOR, LNOT = range(15, 24)
SELECT, ASSERT_EQ, ASSERT_NZ, RUN, CALL, LOG = range(24, 30)

NAMES = ["copy", "add", "sub", "mul", "div", "idiv", "mod", "pow", "neg", "shl", "shr", "band", "bor",
         "bxor", "bnot", "lt", "gt", "leq", "geq", "eq", "neq", "land", "lor", "lnot", "select",
         "assert_eq", "assert_nz", "run", "call", "log"]

# reference C symbol each operator maps to (compute_bucket.rs:315-341)
C_SYMBOL = {ADD: "Fr_add", SUB: "Fr_sub", MUL: "Fr_mul", DIV: "Fr_div", IDIV: "Fr_idiv", MOD: "Fr_mod",
            POW: "Fr_pow", NEG: "Fr_neg", SHL: "Fr_shl", SHR: "Fr_shr", BAND: "Fr_band", BOR: "Fr_bor",
            BXOR: "Fr_bxor", BNOT: "Fr_bnot", LT: "Fr_lt", GT: "Fr_gt", LEQ: "Fr_leq", GEQ: "Fr_geq",
            EQ: "Fr_eq", NEQ: "Fr_neq", LAND: "Fr_land", LOR: "Fr_lor", LNOT: "Fr_lnot", COPY: "Fr_copy"}

UNARY = {COPY, NEG, BNOT, LNOT, ASSERT_NZ}
NO_DST = {ASSERT_EQ, ASSERT_NZ, RUN, CALL, LOG}

# operand kinds
K_SIG, K_TMP, K_CONST, K_NONE = 0, 1, 2, 3

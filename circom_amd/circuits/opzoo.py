"""Every operator of circom's witness-code expression language on two run-time values (`<--` accepts arbitrary
expressions, basic-operators.md): field + - * / , integer \\ % ** , shifts, bitwise, relational, boolean, unary
minus, ~, !, and the conditional.  No constraints: this circuit exists to pin the device implementations of the
`Fr_*` functions END TO END against the reference's own runtime, through the same lowering,
scheduling and kernels the real circuits use (the slow-path operators live only in the FULL_OPS kernel variant)."""
from ..frontend.dsl import template

NAMES = ["add", "sub", "mul", "div", "idiv", "mod", "pow", "shl", "shr", "band", "bor", "bxor", "lt", "gt", "leq", "geq",
         "eq", "neq", "land", "lor", "neg", "bnot", "lnot", "sel", "mix"]


@template
def OperatorZoo(c):
    a = c.input("a")
    b = c.input("b")
    out = c.output("out", len(NAMES))
    nz = c.select(b.eq(0), 1, b)           # \\ and % by zero abort the reference (GMP division by zero)
    exprs = [a + b, a - b, a * b, a / b, a // nz, a % nz, a ** (b & 255), a << b, a >> b, a & b, a | b, a ^ b,
             a.lt(b), a.gt(b), a.leq(b), a.geq(b), a.eq(b), a.neq(b), a.land(b), a.lor(b), -a, ~a, a.lnot(),
             c.select(a.lt(b), a * 3, b - 1),
             ((a >> 3) & 0xFFFF) * (b % 1000 + 1) + (a // 7 % 11) - (~b & 15)]
    for i, e in enumerate(exprs):
        c.hint(out[i], e)

"""The parser of circom source text (circom_amd/frontend/circom_lang.py) against the grammar it restates
(parser/src/lang.lalrpop): precedence and associativity of every operator tier, the statement forms, declarations with
initialisers, the syntactic sugar the reference removes before execution, and the errors the reference reports while parsing."""
import pytest

from circom_amd.frontend.circom_lang import CircomSyntaxError, Parser, Source, parse_text, tokenize


def expr(text):
    p = Parser(Source(0, "<t>", text))
    e = p.expression()
    if p.peek()[0] != "eof":
        p.err("unexpected %r after the expression" % (p.peek()[1],))
    return strip(e)


def strip(e):
    """drop source positions"""
    if isinstance(e, tuple):
        if len(e) and isinstance(e[-1], tuple) and len(e[-1]) == 2 and all(isinstance(x, int) for x in e[-1]) and e[0] in (
                "num", "var", "bin", "un", "tern", "call", "arr", "tuple", "parallel", "anon", "block", "seq", "decl", "subst", "if",
                "while", "return", "ceq", "log", "assert", "anonstmt", "template", "function", "bus", "main"):
            e = e[:-1]
        return tuple(strip(x) for x in e)
    if isinstance(e, list):
        return [strip(x) for x in e]
    return e


def V(n, *acc):
    return ("var", n, list(acc))


def N(v):
    return ("num", v)


def B(op, l, r):
    return ("bin", op, l, r)


def test_every_infix_tier_is_left_associative_including_pow():
    # lang.lalrpop:575-606: InfixOpTier<Op, Next> = InfixOpTier Op Next | Next
    assert expr("a ** b ** c") == B("**", B("**", V("a"), V("b")), V("c"))
    assert expr("a - b - c") == B("-", B("-", V("a"), V("b")), V("c"))
    assert expr("a \\ b % c") == B("%", B("\\", V("a"), V("b")), V("c"))
    assert expr("a < b == c") == B("==", B("<", V("a"), V("b")), V("c"))


def test_precedence_ladder():
    # || < && < cmp < | < ^ < & < shifts < + - < * / \ % < ** < prefix
    assert expr("a || b && c") == B("||", V("a"), B("&&", V("b"), V("c")))
    assert expr("a && b == c") == B("&&", V("a"), B("==", V("b"), V("c")))
    assert expr("a == b | c") == B("==", V("a"), B("|", V("b"), V("c")))
    assert expr("a | b ^ c") == B("|", V("a"), B("^", V("b"), V("c")))
    assert expr("a ^ b & c") == B("^", V("a"), B("&", V("b"), V("c")))
    assert expr("a & b << c") == B("&", V("a"), B("<<", V("b"), V("c")))
    assert expr("a << b + c") == B("<<", V("a"), B("+", V("b"), V("c")))
    assert expr("a + b * c") == B("+", V("a"), B("*", V("b"), V("c")))
    assert expr("a * b ** c") == B("*", V("a"), B("**", V("b"), V("c")))
    # (in >> i) & 1 needs its parentheses in C, not in circom: shifts bind tighter than &
    assert expr("in >> i & 1") == B("&", B(">>", V("in"), V("i")), N(1))


def test_prefix_binds_tighter_than_pow_and_takes_one_operator():
    assert expr("-2 ** 2") == B("**", ("un", "-", N(2)), N(2))
    assert expr("!a && ~b") == B("&&", ("un", "!", V("a")), ("un", "~", V("b")))
    assert expr("a - -b") == B("-", V("a"), ("un", "-", V("b")))
    with pytest.raises(CircomSyntaxError):
        expr("- -a")                       # PrefixOpTier<Op, Expression1>: the operand is not a prefix expression again


def test_inline_switch_takes_expression12_branches():
    assert expr("a < b ? a * 3 : b - 1") == ("tern", B("<", V("a"), V("b")), B("*", V("a"), N(3)), B("-", V("b"), N(1)))
    with pytest.raises(CircomSyntaxError):
        expr("a ? b : c ? d : e")          # a nested switch needs parentheses (Expression13 is not recursive)
    assert expr("a ? b : (c ? d : e)")[0] == "tern"


def test_accesses_calls_arrays_tuples_anonymous_components():
    assert expr("c[i].out[2]") == V("c", ("idx", V("i")), ("field", "out"), ("idx", N(2)))
    assert expr("f(1, x)") == ("call", "f", [N(1), V("x")])
    assert expr("[a, b + 1]") == ("arr", [V("a"), B("+", V("b"), N(1))])
    assert expr("(a, _, c)") == ("tuple", [V("a"), V("_"), V("c")])
    assert expr("(a)") == V("a")
    assert expr("T(3)(x, y)") == ("anon", "T", [N(3)], [V("x"), V("y")], None)
    assert expr("T()(b <== x, a <== y)") == ("anon", "T", [], [V("x"), V("y")], [("<==", "b"), ("<==", "a")])
    assert expr("parallel T(2)") == ("parallel", ("call", "T", [N(2)]))
    assert expr("0x1F + 010") == B("+", N(31), N(10))


def test_tokens():
    toks = [(k, v) for k, v, _ in tokenize(Source(0, "t", "a<==b-->c // x\n/* y\n */ d**=2; e\\=3 $_x1 ===_ \"s t\""))]
    assert toks == [("id", "a"), ("op", "<=="), ("id", "b"), ("op", "-->"), ("id", "c"), ("id", "d"), ("op", "**="), ("num", 2),
                    ("op", ";"), ("id", "e"), ("op", "\\="), ("num", 3), ("id", "$_x1"), ("op", "==="), ("id", "_"), ("str", "s t"),
                    ("eof", None)]
    with pytest.raises(CircomSyntaxError, match="unterminated"):
        tokenize(Source(0, "t", "a /* b"))
    with pytest.raises(CircomSyntaxError, match="illegal character"):
        tokenize(Source(0, "t", "a # b"))


def body(text):
    ar = parse_text("template T() { %s }" % text)
    return strip(ar.templates["T"][3])[1]


def test_statement_forms():
    s = body("signal input a; signal output {binary, maxbit} b[2]; var x = 3, y[2]; component c = A(); x += 2; x++; a ==> c.in; "
             "c.out --> b[0]; b[1] <== a * a; a * b[0] === b[1]; log(\"v\", x); log(); assert(x > 1);")
    assert s[0] == ("decl", ("signal", "input", []), "a", [])
    assert s[1] == ("decl", ("signal", "output", ["binary", "maxbit"]), "b", [N(2)])
    assert s[2] == ("seq", [("decl", ("var",), "x", []), ("subst", V("x"), "=", N(3)), ("decl", ("var",), "y", [N(2)])])
    assert s[3] == ("seq", [("decl", ("component",), "c", []), ("subst", V("c"), "=", ("call", "A", []))])
    assert s[4] == ("subst", V("x"), "=", B("+", V("x"), N(2)))
    assert s[5] == ("subst", V("x"), "=", B("+", V("x"), N(1)))
    assert s[6] == ("subst", V("c", ("field", "in")), "<==", V("a"))
    assert s[7] == ("subst", V("b", ("idx", N(0))), "<--", V("c", ("field", "out")))
    assert s[8][2] == "<==" and s[9][0] == "ceq"
    assert s[10] == ("log", [("str", "v"), V("x")]) and s[11] == ("log", []) and s[12][0] == "assert"


def test_for_is_a_block_around_a_while():
    # ast_shortcuts::for_into_while: { init; while (cond) { body; step } }
    s = body("for (var i = 0; i < 4; i++) { x = x + i; }")[0]
    assert s[0] == "block" and s[1][0] == ("seq", [("decl", ("var",), "i", []), ("subst", V("i"), "=", N(0))])
    w = s[1][1]
    assert w[0] == "while" and w[1] == B("<", V("i"), N(4))
    assert w[2][0] == "block" and w[2][1][1] == ("subst", V("i"), "=", B("+", V("i"), N(1)))


def test_else_binds_to_the_nearest_if():
    s = body("if (a) if (b) x = 1; else x = 2;")[0]
    assert s[0] == "if" and s[3] is None and s[2][0] == "if" and s[2][3] is not None


def test_signal_declaration_with_initialiser_and_both_keyword_orders():
    s = body("signal output o <== a * b; input signal p; signal q <-- 3, r <-- 4;")
    assert s[0] == ("seq", [("decl", ("signal", "output", []), "o", []), ("subst", V("o"), "<==", B("*", V("a"), V("b")))])
    assert s[1] == ("decl", ("signal", "input", []), "p", [])
    assert [x[0] for x in s[2][1]] == ["decl", "subst", "decl", "subst"]
    with pytest.raises(CircomSyntaxError, match="mix"):
        body("signal q <-- 3, r <== 4;")


def test_tuple_declarations_and_bus_declarations():
    s = body("var (a, b[2]) = (1, [2, 3]); Point(2) input {tag} p[3]; output Seg s; Point q <== r;")
    assert s[0][1][2] == ("subst", ("tuple", [V("a"), V("b")]), "=", ("tuple", [N(1), ("arr", [N(2), N(3)])]))
    assert s[1] == ("decl", ("bus", "Point", [N(2)], "input", ["tag"]), "p", [N(3)])
    assert s[2] == ("decl", ("bus", "Seg", [], "output", []), "s", [])
    assert s[3][1][0] == ("decl", ("bus", "Point", [], "mid", []), "q", [])


def test_definitions_pragmas_main():
    ar = parse_text("""pragma circom 2.1.6; pragma custom_templates;
        function f(a, b) { return a + b; }
        template parallel T(n) { signal input x; }
        template custom G() { signal input x; }
        bus P(n) { signal v[n]; }
        component main {public [x, y]} = T(3);""")
    assert ar.version == (2, 1, 6) and ar.custom_templates
    assert ar.functions["f"][2] == ["a", "b"]
    assert ar.templates["T"][4] == frozenset({"parallel"}) and ar.templates["G"][4] == frozenset({"custom"})
    assert "P" in ar.buses and ar.main[1] == ["x", "y"]


@pytest.mark.parametrize("text,msg", [
    ("template T() { signal input a }", "missing semicolon"),
    ("template T() { ++a; }", "does not admit"),
    ("template T() { a + 1; }", "illegal expression"),
    ("template T() { a = ; }", "illegal expression"),
    ("template T() { signal input a;", "unterminated block"),
    ("pragma circom 2.0; template T() {}", "unrecognized version"),
    ("pragma once; template T() {}", "unrecognized pragma"),
    ("template T() {} template T() {}", "declared twice"),
    ("template T() {} component main = T(); component main = T();", "multiple main"),
    ("template T() { (a + 1, b) <== X()(c); }", "holds variables only"),
    ("function 3f() {}", "expected function name"),
])
def test_syntax_errors_carry_file_line_column(text, msg):
    with pytest.raises(CircomSyntaxError, match=msg) as ex:
        parse_text(text, "x.circom")
    if ex.value.line is not None:
        assert str(ex.value).startswith("x.circom:%d:%d: " % (ex.value.line, ex.value.col))


def test_include_resolution(tmp_path):
    from circom_amd.frontend.circom_lang import parse_program
    (tmp_path / "lib").mkdir()
    (tmp_path / "lib" / "a.circom").write_text('include "b.circom"; template A() { signal input x; }')
    (tmp_path / "lib" / "b.circom").write_text('template Bt() { signal input x; }')
    (tmp_path / "sub").mkdir()
    (tmp_path / "sub" / "c.circom").write_text('include "../lib/b.circom"; template C() { signal input x; }')
    (tmp_path / "main.circom").write_text('include "a.circom"; include "sub/c.circom"; component main = A();')
    ar = parse_program(str(tmp_path / "main.circom"), [str(tmp_path / "lib")])
    # b.circom is reached twice (through the library path and relative to sub/) and parsed once
    assert set(ar.templates) == {"A", "Bt", "C"} and len(ar.sources) == 4
    with pytest.raises(CircomSyntaxError, match="has not been found"):
        parse_program(str(tmp_path / "main.circom"), [])

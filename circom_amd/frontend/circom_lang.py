"""The circom language as text: lexer + recursive-descent parser of `.circom` sources.

What it replaces: the reference's `parser` crate (parser/src/lang.lalrpop - the grammar this file follows production by
production, parser/src/include_logic.rs for `include`, parser/src/syntax_sugar_remover.rs for anonymous components and
tuples, which stay in the tree here and are resolved by the executor).  It is new code: a hand-written scanner and a
precedence-climbing expression parser, not a generated LR table.

Properties of the grammar that are easy to get wrong and are pinned by tests/test_circom_lang.py:
  * every infix tier is LEFT associative, `**` included (lang.lalrpop:600-606: InfixOpTier is left-recursive);
  * prefix `- ! ~` bind tighter than `**` (Expression2 below Expression3): `-2 ** 2` is `(-2) ** 2`;
  * the branches of `c ? a : b` are Expression12: a nested switch needs parentheses;
  * relational operators form one left-associative tier; `|` is looser than `^`, which is looser than `&`, then shifts,
    then `+ -`, then `* / \\ %`;
  * `for (init; cond; step) body` is `{ init; while (cond) { body; step; } }` (ast_shortcuts::for_into_while);
  * `x++` / `x += e` are substitutions (`x = x + 1`), there is no `++x`;
  * identifiers match `[$_]*[a-zA-Z][a-zA-Z$_0-9]*`; a lone `_` is the discard target; numbers are decimal or `0x` hex and
    are reduced modulo the field when the executor reads them (build_number).

AST: plain tuples, first element = node kind, last element = source position (file id, byte offset).
  expressions   ('num', v, pos) ('var', name, access, pos) with access = [('idx', expr) | ('field', name)]
                ('bin', op, l, r, pos) ('un', op, e, pos) ('tern', c, a, b, pos) ('call', name, args, pos)
                ('arr', [e], pos) ('tuple', [e], pos) ('parallel', e, pos)
                ('anon', template, params, signals, names | None, pos)
  statements    ('block', [s], pos) ('seq', [s], pos)  - seq = several statements of ONE declaration, no scope of its own
                ('decl', xtype, name, dims, pos)  xtype = ('var',) | ('component',) | ('signal', kind, tags)
                                                         | ('bus', bus name, args, kind, tags); kind = 'input' | 'output' | 'mid'
                ('subst', target, op, rhe, pos)   target = a 'var' / 'tuple' expression, op in '=' '<--' '<=='
                ('if', c, t, e | None, pos) ('while', c, body, pos) ('return', e, pos) ('ceq', l, r, pos)
                ('log', [expr | ('str', text)], pos) ('assert', e, pos) ('anonstmt', e, pos)
  definitions   ('template', name, params, body, flags, pos) ('function', name, params, body, pos)
                ('bus', name, params, body, pos)
"""
from __future__ import annotations

import os
import re


class CircomSyntaxError(Exception):
    def __init__(self, msg, file=None, line=None, col=None):
        self.msg, self.file, self.line, self.col = msg, file, line, col
        where = "%s:%s:%s: " % (file, line, col) if file is not None else ""
        super().__init__(where + msg)


KEYWORDS = {"signal", "input", "output", "public", "template", "component", "var", "function", "return", "if", "else",
            "for", "while", "do", "log", "assert", "include", "pragma", "parallel", "bus", "custom", "extern_c", "main"}
# `main`, `circom`, `custom_templates` are contextual in the reference's lexer as well (they are string literals of the
# grammar, hence reserved); `main` is kept usable as the component name only.

_OPS = ["<==", "<--", "==>", "-->", "===", "**=", "<<=", ">>=", "\\=", "+=", "-=", "*=", "/=", "%=", "&=", "|=", "^=", "++", "--",
        "**", "<<", ">>", "<=", ">=", "==", "!=", "&&", "||", "+", "-", "*", "/", "\\", "%", "&", "|", "^", "~", "!", "<", ">", "=",
        "?", ":", ";", ",", ".", "(", ")", "[", "]", "{", "}"]
_OPS.sort(key=len, reverse=True)
_TOKEN = re.compile(
    r"(?P<ws>[ \t\r\n]+)"
    r"|(?P<lc>//[^\n]*)"
    r"|(?P<bc>/\*.*?\*/)"
    r"|(?P<hex>0x[0-9A-Fa-f]*)"
    r"|(?P<num>[0-9]+)"
    r"|(?P<id>[$_]*[a-zA-Z][a-zA-Z$_0-9]*)"
    r"|(?P<under>_)"
    r"|(?P<str>\"[^\"\n]*\")"
    r"|(?P<op>" + "|".join(re.escape(o) for o in _OPS) + ")",
    re.S)


class Source:
    """one file: text, name, and the line table error messages and anonymous-component names need"""

    def __init__(self, fid, name, text):
        self.fid, self.name, self.text = fid, name, text
        self._nl = [m.start() for m in re.finditer("\n", text)]

    def line_col(self, off):
        from bisect import bisect_left
        ln = bisect_left(self._nl, off)
        start = self._nl[ln - 1] + 1 if ln else 0
        return ln + 1, off - start + 1


def tokenize(src: Source):
    text = src.text
    out = []
    i, n = 0, len(text)
    while i < n:
        m = _TOKEN.match(text, i)
        if text.startswith("/*", i) and (m is None or m.lastgroup != "bc"):
            ln, col = src.line_col(i)
            raise CircomSyntaxError("unterminated /* */ comment", src.name, ln, col)
        if m is None:
            ln, col = src.line_col(i)
            raise CircomSyntaxError("illegal character %r" % text[i], src.name, ln, col)
        k = m.lastgroup
        if k == "hex":
            out.append(("num", int(m.group()[2:] or "0", 16), i))
        elif k == "num":
            out.append(("num", int(m.group()), i))
        elif k == "id":
            out.append(("id", m.group(), i))
        elif k == "under":
            out.append(("id", "_", i))
        elif k == "str":
            out.append(("str", m.group()[1:-1], i))
        elif k == "op":
            out.append(("op", m.group(), i))
        i = m.end()
    out.append(("eof", None, n))
    return out


_BIN_TIERS = [          # loosest first; every tier left-associative (lang.lalrpop:575-606)
    ("||",), ("&&",), ("==", "!=", "<", ">", "<=", ">="), ("|",), ("^",), ("&",), ("<<", ">>"), ("+", "-"),
    ("*", "/", "\\", "%"), ("**",)]
_ASSIGN_OPS = {"\\=": "\\", "**=": "**", "+=": "+", "-=": "-", "*=": "*", "/=": "/", "%=": "%", "<<=": "<<", ">>=": ">>",
               "&=": "&", "|=": "|", "^=": "^"}


class Parser:
    def __init__(self, src: Source):
        self.src = src
        self.toks = tokenize(src)
        self.p = 0

    # ---- token helpers ------------------------------------------------------------------------------------------------
    def err(self, msg, tok=None):
        tok = tok or self.toks[self.p]
        ln, col = self.src.line_col(tok[2])
        raise CircomSyntaxError(msg, self.src.name, ln, col)

    def peek(self, k=0):
        return self.toks[min(self.p + k, len(self.toks) - 1)]

    def at_op(self, *ops):
        t = self.toks[self.p]
        return t[0] == "op" and t[1] in ops

    def at_kw(self, *kws):
        t = self.toks[self.p]
        return t[0] == "id" and t[1] in kws

    def take_op(self, op):
        t = self.toks[self.p]
        if t[0] == "op" and t[1] == op:
            self.p += 1
            return t
        self.err("expected %r, found %r" % (op, t[1] if t[0] != "eof" else "end of file"))

    def take_kw(self, kw):
        t = self.toks[self.p]
        if t[0] == "id" and t[1] == kw:
            self.p += 1
            return t
        self.err("expected %r, found %r" % (kw, t[1] if t[0] != "eof" else "end of file"))

    def take_id(self, what="identifier"):
        t = self.toks[self.p]
        if t[0] == "id" and t[1] not in KEYWORDS and t[1] != "_":
            self.p += 1
            return t[1]
        self.err("expected %s, found %r" % (what, t[1] if t[0] != "eof" else "end of file"))

    def semicolon(self):
        t = self.toks[self.p]
        if t[0] == "op" and t[1] == ";":
            self.p += 1
            return
        self.err("missing semicolon")          # ReportCode::MissingSemicolon

    def pos(self, tok=None):
        return (self.src.fid, (tok or self.toks[self.p])[2])

    # ---- file level ---------------------------------------------------------------------------------------------------
    def parse_file(self):
        """-> dict(version, custom_templates, includes, definitions, main)   (ParseAst, lang.lalrpop:93-100)"""
        version, custom = None, False
        while self.at_kw("pragma"):
            self.p += 1
            t = self.peek()
            if t[0] == "id" and t[1] == "circom":
                self.p += 1
                parts = []
                for k in range(3):
                    n = self.peek()
                    if n[0] != "num":
                        self.err("unrecognized version")
                    parts.append(n[1])
                    self.p += 1
                    if k < 2:
                        if not self.at_op("."):
                            self.err("unrecognized version")      # ReportCode::UnrecognizedVersion
                        self.p += 1
                version = tuple(parts)
            elif t[0] == "id" and t[1] == "custom_templates":
                self.p += 1
                custom = True
            else:
                self.err("unrecognized pragma")
            self.semicolon()
        includes = []
        while self.at_kw("include"):
            self.p += 1
            t = self.peek()
            if t[0] != "str":
                self.err("unrecognized include")
            includes.append((t[1], self.pos(t)))
            self.p += 1
            self.semicolon()
        defs = []
        main = None
        while self.peek()[0] != "eof":
            if self.at_kw("component") and self.peek(1)[0] == "id" and self.peek(1)[1] == "main":
                if main is not None:
                    self.err("multiple main components")
                main = self.parse_main()
                continue
            if main is not None:
                self.err("definitions must precede the main component")
            defs.append(self.parse_definition())
        return dict(version=version, custom_templates=custom, includes=includes, definitions=defs, main=main)

    def parse_main(self):
        t0 = self.take_kw("component")
        self.take_kw("main")
        public = []
        if self.at_op("{"):
            self.p += 1
            self.take_kw("public")
            self.take_op("[")
            public.append(self.take_id())
            while self.at_op(","):
                self.p += 1
                public.append(self.take_id())
            self.take_op("]")
            self.take_op("}")
        self.take_op("=")
        init = self.expression()
        self.semicolon()
        return ("main", public, init, self.pos(t0))

    def _param_names(self):
        names = []
        self.take_op("(")
        if not self.at_op(")"):
            names.append(self.take_id())
            while self.at_op(","):
                self.p += 1
                names.append(self.take_id())
        self.take_op(")")
        return names

    def parse_definition(self):
        t0 = self.peek()
        if self.at_kw("function"):
            self.p += 1
            name = self.take_id("function name")
            params = self._param_names()
            return ("function", name, params, self.block(), self.pos(t0))
        if self.at_kw("template"):
            self.p += 1
            flags = set()
            for kw in ("custom", "extern_c", "parallel"):          # in this order (lang.lalrpop:136)
                if self.at_kw(kw):
                    self.p += 1
                    flags.add(kw)
            name = self.take_id("template name")
            params = self._param_names() if self.at_op("(") else []
            return ("template", name, params, self.block(), frozenset(flags), self.pos(t0))
        if self.at_kw("bus"):
            self.p += 1
            name = self.take_id("bus name")
            params = self._param_names() if self.at_op("(") else []
            return ("bus", name, params, self.block(), self.pos(t0))
        self.err("expected a template, function or bus definition")

    # ---- statements -----------------------------------------------------------------------------------------------------
    def block(self):
        t0 = self.take_op("{")
        stmts = []
        while not self.at_op("}"):
            if self.peek()[0] == "eof":
                self.err("unterminated block", t0)
            stmts.append(self.statement3())
        self.p += 1
        return ("block", stmts, self.pos(t0))

    def _starts_declaration(self):
        t = self.peek()
        if t[0] != "id":
            return False
        if t[1] in ("var", "signal", "component"):
            return True
        if t[1] in ("input", "output"):
            return True
        # a bus declaration: `BusName [ (args) ] [input|output] [{tags}] name ...`: an identifier followed by an identifier,
        # by `input` / `output`, by a tag list + identifier, or by a parenthesised argument list + one of those
        if t[1] in KEYWORDS or t[1] == "_":
            return False
        k = 1
        n = self.peek(k)
        if n[0] == "op" and n[1] == "(":
            depth = 0
            while True:
                n = self.peek(k)
                if n[0] == "eof":
                    return False
                if n[0] == "op" and n[1] == "(":
                    depth += 1
                elif n[0] == "op" and n[1] == ")":
                    depth -= 1
                    if depth == 0:
                        break
                k += 1
            k += 1
            n = self.peek(k)
            if n[0] == "op" and n[1] == "(":        # T(params)(signals): an anonymous component, not a declaration
                return False
        if n[0] == "id" and n[1] in ("input", "output"):
            return True
        if n[0] == "op" and n[1] == "{":
            # `Bus {tag} name`: a tag list is `{ id (, id)* }` followed by an identifier
            j = k + 1
            while self.peek(j)[0] == "id" and self.peek(j + 1)[0] == "op" and self.peek(j + 1)[1] == ",":
                j += 2
            return (self.peek(j)[0] == "id" and self.peek(j + 1)[0] == "op" and self.peek(j + 1)[1] == "}"
                    and self.peek(j + 2)[0] == "id")
        return n[0] == "id" and n[1] not in KEYWORDS and n[1] != "_"

    def statement3(self):
        if self._starts_declaration():
            d = self.declaration()
            self.semicolon()
            return d
        return self.statement()

    def _tags(self):
        tags = []
        if self.at_op("{"):
            self.p += 1
            tags.append(self.take_id("tag name"))
            while self.at_op(","):
                self.p += 1
                tags.append(self.take_id("tag name"))
            self.take_op("}")
        return tags

    def _dims(self):
        dims = []
        while self.at_op("["):
            self.p += 1
            dims.append(self.expression())
            self.take_op("]")
        return dims

    def declaration(self):
        """ParseDeclaration (lang.lalrpop:292-371) -> one statement (a 'seq' when there are several symbols or initialisers)"""
        t0 = self.peek()
        pos = self.pos(t0)
        if self.at_kw("var"):
            self.p += 1
            xtype, init_ops = ("var",), ("=",)
        elif self.at_kw("component"):
            self.p += 1
            xtype, init_ops = ("component",), ("=",)
        elif self.at_kw("signal") or (self.at_kw("input", "output") and self._kw_at(1, "signal")):
            kind = "mid"
            if self.at_kw("signal"):
                self.p += 1
                if self.at_kw("input", "output"):
                    kind = self.peek()[1]
                    self.p += 1
            else:                                   # `input signal` / `output signal`
                kind = self.peek()[1]
                self.p += 1
                self.take_kw("signal")
            xtype, init_ops = ("signal", kind, self._tags()), ("<==", "<--")
        else:
            # bus: [input|output] Name [(args)] [input|output] [{tags}]
            kind = "mid"
            if self.at_kw("input", "output"):
                kind = self.peek()[1]
                self.p += 1
            bname = self.take_id("bus name")
            args = []
            if self.at_op("("):
                self.p += 1
                if not self.at_op(")"):
                    args = self.listable()
                self.take_op(")")
            if self.at_kw("input", "output"):
                kind = self.peek()[1]
                self.p += 1
            xtype, init_ops = ("bus", bname, args, kind, self._tags()), ("<==", "<--")
        stmts = []
        if self.at_op("("):
            # `var (a, b[2]) = expr;` : declarations + one tuple substitution
            self.p += 1
            syms = []
            while True:
                nt = self.peek()
                name = self.take_id()
                syms.append((name, self._dims(), self.pos(nt)))
                if self.at_op(","):
                    self.p += 1
                    continue
                break
            self.take_op(")")
            for name, dims, npos in syms:
                stmts.append(("decl", xtype, name, dims, npos))
            if self.at_op("=", "<==", "<--"):
                op = self.peek()[1]
                self.p += 1
                rhe = self.expression()
                target = ("tuple", [("var", name, [], npos) for name, dims, npos in syms], pos)
                stmts.append(("subst", target, op, rhe, pos))
            return ("seq", stmts, pos)
        first_op = None
        while True:
            nt = self.peek()
            name = self.take_id()
            dims = self._dims()
            npos = self.pos(nt)
            stmts.append(("decl", xtype, name, dims, npos))
            if self.at_op(*init_ops):
                op = self.peek()[1]
                if xtype[0] in ("signal", "bus"):
                    # one declaration uses ONE operator for all its symbols (SignalSymbol / SignalSimpleSymbol lists)
                    if first_op is not None and op != first_op:
                        self.err("a declaration cannot mix <== and <--")
                    first_op = op
                self.p += 1
                rhe = self.expression()
                stmts.append(("subst", ("var", name, [], npos), op, rhe, npos))
            if self.at_op(","):
                self.p += 1
                continue
            break
        return stmts[0] if len(stmts) == 1 else ("seq", stmts, pos)

    def _kw_at(self, k, kw):
        t = self.peek(k)
        return t[0] == "id" and t[1] == kw

    def statement(self):
        self.sdepth = getattr(self, "sdepth", 0) + 1
        try:
            if self.sdepth > self.MAX_NESTING:
                self.err("block nested too deeply")
            return self._statement()
        finally:
            self.sdepth -= 1

    def _statement(self):
        t0 = self.peek()
        pos = self.pos(t0)
        if self.at_kw("if"):
            self.p += 1
            self.take_op("(")
            cond = self.expression()
            self.take_op(")")
            then = self.statement()
            other = None
            if self.at_kw("else"):                   # binds to the nearest `if`
                self.p += 1
                other = self.statement()
            return ("if", cond, then, other, pos)
        if self.at_kw("for"):
            self.p += 1
            self.take_op("(")
            init = self.declaration() if self._starts_declaration() else self.substitution()
            self.semicolon()
            cond = self.expression()
            self.semicolon()
            step = self.substitution()
            self.take_op(")")
            body = self.statement()
            return ("block", [init, ("while", cond, ("block", [body, step], pos), pos)], pos)
        if self.at_kw("while"):
            self.p += 1
            self.take_op("(")
            cond = self.expression()
            self.take_op(")")
            return ("while", cond, self.statement(), pos)
        if self.at_kw("return"):
            self.p += 1
            e = self.expression()
            self.semicolon()
            return ("return", e, pos)
        if self.at_kw("log"):
            self.p += 1
            self.take_op("(")
            args = []
            if not self.at_op(")"):
                while True:
                    t = self.peek()
                    if t[0] == "str":
                        self.p += 1
                        args.append(("str", t[1]))
                    else:
                        args.append(self.expression())
                    if self.at_op(","):
                        self.p += 1
                        continue
                    break
            self.take_op(")")
            self.semicolon()
            return ("log", args, pos)
        if self.at_kw("assert"):
            self.p += 1
            self.take_op("(")
            e = self.expression()
            self.take_op(")")
            self.semicolon()
            return ("assert", e, pos)
        if self.at_op("{"):
            return self.block()
        if self.at_op("++", "--"):
            self.err("illegal expression: circom language does not admit the %s<var> operator, use <var>%s instead"
                     % (t0[1], t0[1]))
        s = self.substitution(allow_other=True)
        self.semicolon()
        return s

    def substitution(self, allow_other=False):
        """ParseSubstitution (lang.lalrpop:373-457); with allow_other also `lhe === rhe` and a lone anonymous component"""
        t0 = self.peek()
        pos = self.pos(t0)
        lhe = self.expression()
        if self.at_op("=", "<--", "<=="):
            op = self.peek()[1]
            self.p += 1
            rhe = self.expression()
            return ("subst", self._target(lhe, t0), op, rhe, pos)
        if self.at_op("-->", "==>"):
            op = "<--" if self.peek()[1] == "-->" else "<=="
            self.p += 1
            t1 = self.peek()
            target = self.expression()
            return ("subst", self._target(target, t1), op, lhe, pos)
        if self.at_op(*_ASSIGN_OPS):
            op = _ASSIGN_OPS[self.peek()[1]]
            self._plain_variable(lhe, t0)
            self.p += 1
            rhe = self.expression()
            return ("subst", lhe, "=", ("bin", op, lhe, rhe, pos), pos)
        if self.at_op("++", "--"):
            op = "+" if self.peek()[1] == "++" else "-"
            self._plain_variable(lhe, t0)
            self.p += 1
            return ("subst", lhe, "=", ("bin", op, lhe, ("num", 1, pos), pos), pos)
        if allow_other:
            if self.at_op("==="):
                self.p += 1
                rhe = self.expression()
                return ("ceq", lhe, rhe, pos)
            if lhe[0] == "anon":
                return ("anonstmt", lhe, pos)
        self.err("illegal expression", t0)

    def _plain_variable(self, e, tok):
        if e[0] != "var" or e[1] == "_":
            self.err("the left side of this operator must be a variable", tok)

    def _target(self, e, tok):
        if e[0] == "var":
            return e
        if e[0] == "tuple":
            for x in e[1]:
                if x[0] != "var":
                    self.err("a tuple on the left side of an assignment holds variables only", tok)
            return e
        self.err("the left side of an assignment must be a variable, a signal or a tuple of them", tok)

    # ---- expressions ----------------------------------------------------------------------------------------------------
    def listable(self):
        out = [self.expression()]
        while self.at_op(","):
            self.p += 1
            out.append(self.expression())
        return out

    MAX_NESTING = 200      # parentheses / brackets / calls inside one another, blocks inside one another

    def expression(self):
        t0 = self.peek()
        self.depth = getattr(self, "depth", 0) + 1
        try:
            if self.depth > self.MAX_NESTING:
                self.err("expression nested too deeply")
            if self.at_kw("parallel"):
                self.p += 1
                return ("parallel", self.expression1(), self.pos(t0))
            return self.expression1()
        finally:
            self.depth -= 1

    def expression1(self):
        t0 = self.peek()
        cond = self.binary(0)
        if self.at_op("?"):
            self.p += 1
            a = self.binary(0)
            self.take_op(":")
            b = self.binary(0)
            return ("tern", cond, a, b, self.pos(t0))
        return cond

    def binary(self, tier):
        if tier == len(_BIN_TIERS):
            return self.prefix()
        t0 = self.peek()
        ops = _BIN_TIERS[tier]
        lhe = self.binary(tier + 1)
        while self.at_op(*ops):
            op = self.peek()[1]
            self.p += 1
            rhe = self.binary(tier + 1)
            lhe = ("bin", op, lhe, rhe, self.pos(t0))
        return lhe

    def prefix(self):
        t0 = self.peek()
        if self.at_op("-", "!", "~"):
            self.p += 1
            # PrefixOpTier<Op, Expression1>: ONE prefix operator in front of an Expression1 (`- -x` does not parse)
            return ("un", t0[1], self.expression_1(), self.pos(t0))
        return self.expression_1()

    def expression_1(self):
        """Expression1 / Expression0: calls, anonymous components, inline arrays, tuples, variables, literals, parentheses"""
        t0 = self.peek()
        pos = self.pos(t0)
        if t0[0] == "num":
            self.p += 1
            return ("num", t0[1], pos)
        if t0[0] == "op" and t0[1] == "[":
            self.p += 1
            vals = self.listable()
            self.take_op("]")
            return ("arr", vals, pos)
        if t0[0] == "op" and t0[1] == "(":
            self.p += 1
            first = self.expression()
            if self.at_op(","):
                vals = [first]
                while self.at_op(","):
                    self.p += 1
                    vals.append(self.expression())
                self.take_op(")")
                return ("tuple", vals, pos)
            self.take_op(")")
            return first
        if t0[0] == "id":
            if t0[1] == "_":
                self.p += 1
                return ("var", "_", [], pos)
            if t0[1] in KEYWORDS:
                self.err("unexpected keyword %r" % t0[1])
            self.p += 1
            name = t0[1]
            if self.at_op("("):
                self.p += 1
                args = [] if self.at_op(")") else self.listable()
                self.take_op(")")
                if self.at_op("("):
                    # anonymous component: T(params)(signals) or T(params)(name <== e, ...)
                    self.p += 1
                    sigs, names = [], None
                    if not self.at_op(")"):
                        n1, n2 = self.peek(), self.peek(1)
                        if n1[0] == "id" and n2[0] == "op" and n2[1] in ("=", "<--", "<=="):
                            names = []
                            while True:
                                nm = self.take_id("input name")
                                if not self.at_op("=", "<--", "<=="):
                                    self.err("expected an assignment operator")
                                op = self.peek()[1]
                                self.p += 1
                                names.append((op, nm))
                                sigs.append(self.expression())
                                if self.at_op(","):
                                    self.p += 1
                                    continue
                                break
                        else:
                            sigs = self.listable()
                    self.take_op(")")
                    return ("anon", name, args, sigs, names, pos)
                return ("call", name, args, pos)
            access = []
            while True:
                if self.at_op("["):
                    self.p += 1
                    access.append(("idx", self.expression()))
                    self.take_op("]")
                elif self.at_op("."):
                    self.p += 1
                    t = self.peek()
                    if t[0] != "id" or t[1] == "_":
                        self.err("expected a name after '.'")
                    self.p += 1
                    access.append(("field", t[1]))
                else:
                    break
            return ("var", name, access, pos)
        self.err("illegal expression")


# ---- program archive (parser/src/lib.rs run_parser + include_logic.rs) ------------------------------------------------------
class Archive:
    """every definition reachable from the main file through `include`, plus the main component"""

    def __init__(self):
        self.sources = []          # Source by file id
        self.templates = {}
        self.functions = {}
        self.buses = {}
        self.main = None           # ('main', public, init, pos)
        self.version = None
        self.custom_templates = False

    def where(self, pos):
        src = self.sources[pos[0]]
        ln, col = src.line_col(pos[1])
        return src.name, ln, col

    def line_of(self, pos):
        return self.sources[pos[0]].line_col(pos[1])[0]


def _resolve_include(path, including_dir, libs):
    """include_logic.rs:28-58: relative to the including file first, then every -l directory in order"""
    cand = [os.path.join(including_dir, path)] + [os.path.join(l, path) for l in libs]
    for c in cand:
        if os.path.isfile(c):
            return os.path.realpath(c)
    raise CircomSyntaxError("The file %s to be included has not been found" % path)


def parse_text(text: str, name="<text>", archive: Archive = None) -> Archive:
    """one self-contained source text (no includes)"""
    return _load(None, [], archive, text=text, name=name)


def parse_program(path: str, libs=()) -> Archive:
    """the main file and, transitively, every file it includes (each file once)"""
    return _load(os.path.realpath(path), [os.path.realpath(l) for l in libs], None)


def _load(path, libs, archive, text=None, name=None):
    ar = archive or Archive()
    seen = set()
    queue = [(path, text, name)]
    first = True
    while queue:
        p, text, nm = queue.pop(0)
        if p is not None:
            if p in seen:
                continue
            seen.add(p)
            with open(p, "r") as f:
                text = f.read()
            nm = p
        src = Source(len(ar.sources), nm, text)
        ar.sources.append(src)
        try:
            ast = Parser(src).parse_file()
        except RecursionError:
            raise CircomSyntaxError("expression or block nested too deeply", nm) from None
        if first:
            ar.version = ast["version"]
        ar.custom_templates = ar.custom_templates or ast["custom_templates"]
        for d in ast["definitions"]:
            table = {"template": ar.templates, "function": ar.functions, "bus": ar.buses}[d[0]]
            if d[1] in ar.templates or d[1] in ar.functions or d[1] in ar.buses:
                fn, ln, col = ar.where(d[-1])
                raise CircomSyntaxError("symbol %s declared twice" % d[1], fn, ln, col)
            table[d[1]] = d
        if ast["main"] is not None:
            if ar.main is not None:
                fn, ln, col = ar.where(ast["main"][-1])
                raise CircomSyntaxError("multiple main components in the project structure", fn, ln, col)   # ReportCode::MultipleMain
            ar.main = ast["main"]
        for inc, ipos in ast["includes"]:
            if p is None:
                raise CircomSyntaxError("include needs a file on disk: use parse_program")
            queue.append((_resolve_include(inc, os.path.dirname(p), libs), None, None))
        first = False
    return ar

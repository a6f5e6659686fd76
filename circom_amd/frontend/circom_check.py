"""Static checks over EVERY definition of a parsed program, instantiated or not - the part of the reference's
`type_analysis` crate that does not need types:

  symbol_analysis.rs                                every name that is used is declared (in an enclosing block, before its use),
                                                    nothing is declared twice, called functions / templates / buses exist and
                                                    take that many arguments
  functions_free_of_template_elements.rs            no signals, components, `<==` / `<--` / `===` inside a function
  functions_all_paths_with_return_statement.rs      every path through a function ends in a `return`
  no_returns_in_template.rs                         no `return` inside a template
  buses_free_of_invalid_statements.rs               a bus body declares signals and buses (variables and loops may shape them)
  signal_declaration_analysis.rs                    signals / buses / components are declared in the top-level block of their
                                                    template or in nested `if` blocks - never in the block of a loop

The executor (circom_exec.py) finds the same errors in the code it RUNS; this pass finds them in the branches and the
definitions it never reaches, as the reference does before it executes anything.
"""
from __future__ import annotations

from .dsl import CircuitError


class _Checker:
    def __init__(self, ar):
        self.ar = ar

    def fail(self, msg, pos):
        fn, ln, col = self.ar.where(pos)
        raise CircuitError("%s:%d:%d: %s" % (fn, ln, col, msg))

    # ---- expressions ------------------------------------------------------------------------------------------------------
    def expr(self, e, scopes, kind):
        k = e[0]
        if k == "num":
            return
        if k == "var":
            if e[1] != "_" and not any(e[1] in sc for sc in scopes):
                self.fail("undeclared symbol %s" % e[1], e[-1])
            for a in e[2]:
                if a[0] == "idx":
                    self.expr(a[1], scopes, kind)
            return
        if k == "bin":
            while e[0] == "bin":                     # left-deep chains without recursion
                self.expr(e[3], scopes, kind)
                e = e[2]
            self.expr(e, scopes, kind)
        elif k == "un":
            self.expr(e[2], scopes, kind)
        elif k == "tern":
            for x in e[1:4]:
                self.expr(x, scopes, kind)
        elif k in ("arr", "tuple"):
            for x in e[1]:
                self.expr(x, scopes, kind)
        elif k == "parallel":
            self.expr(e[1], scopes, kind)
        elif k == "call":
            name, args = e[1], e[2]
            d = self.ar.functions.get(name) or self.ar.templates.get(name)
            if d is None:
                self.fail("call to an undeclared function or template %s" % name, e[-1])
            if len(d[2]) != len(args):
                self.fail("%s %s takes %d %s" % (d[0], name, len(d[2]), "arguments" if d[0] == "function" else "parameters"), e[-1])
            if d[0] == "template" and kind == "function":
                self.fail("a function cannot create components", e[-1])
            for x in args:
                self.expr(x, scopes, kind)
        elif k == "anon":
            _, tname, params, sigs, names, pos = e
            if kind == "function":
                self.fail("Functions cannot contain calls to anonymous templates", pos)
            d = self.ar.templates.get(tname)
            if d is None:
                self.fail("The template %s does not exist" % tname, pos)
            if len(d[2]) != len(params):
                self.fail("template %s takes %d parameters" % (tname, len(d[2])), pos)
            for x in list(params) + list(sigs):
                self.expr(x, scopes, kind)

    # ---- statements ---------------------------------------------------------------------------------------------------------
    def declare(self, name, what, scopes, pos):
        if any(name in sc for sc in scopes):
            self.fail("symbol %s declared twice" % name, pos)
        scopes[-1][name] = what

    def stmt(self, s, scopes, kind, in_loop):
        k = s[0]
        if k == "block":
            scopes.append({})
            for x in s[1]:
                self.stmt(x, scopes, kind, in_loop)
            scopes.pop()
        elif k == "seq":
            for x in s[1]:
                self.stmt(x, scopes, kind, in_loop)
        elif k == "decl":
            _, xtype, name, dims, pos = s
            for d in dims:
                self.expr(d, scopes, kind)
            t = xtype[0]
            if t != "var":
                if kind == "function":
                    self.fail("signals and components cannot be declared inside functions", pos)
                if kind == "bus" and t == "component":
                    self.fail("a bus cannot declare components", pos)
                if in_loop:
                    self.fail("%s Is outside the initial scope" % name, pos)
            if t == "bus":
                b = self.ar.buses.get(xtype[1])
                if b is None:
                    self.fail("bus %s is not defined" % xtype[1], pos)
                if len(b[2]) != len(xtype[2]):
                    self.fail("bus %s takes %d parameters" % (xtype[1], len(b[2])), pos)
                for x in xtype[2]:
                    self.expr(x, scopes, kind)
            self.declare(name, t, scopes, pos)
        elif k == "subst":
            _, target, op, rhe, pos = s
            self.expr(rhe, scopes, kind)
            self.expr(target, scopes, kind)
            if op != "=" and kind == "function":
                self.fail("functions cannot assign signals", pos)
            if op != "=" and kind == "bus":
                self.fail("a bus body cannot assign signals", pos)
        elif k == "if":
            self.expr(s[1], scopes, kind)
            self.stmt(("block", [s[2]], s[-1]), scopes, kind, in_loop)
            if s[3] is not None:
                self.stmt(("block", [s[3]], s[-1]), scopes, kind, in_loop)
        elif k == "while":
            self.expr(s[1], scopes, kind)
            self.stmt(("block", [s[2]], s[-1]), scopes, kind, True)
        elif k == "return":
            if kind != "function":
                self.fail("return outside a function", s[-1])
            self.expr(s[1], scopes, kind)
        elif k == "ceq":
            if kind != "template":
                self.fail("%s cannot generate constraints" % ("functions" if kind == "function" else "a bus body"), s[-1])
            self.expr(s[1], scopes, kind)
            self.expr(s[2], scopes, kind)
        elif k == "log":
            for a in s[1]:
                if a[0] != "str":
                    self.expr(a, scopes, kind)
        elif k == "assert":
            self.expr(s[1], scopes, kind)
        elif k == "anonstmt":
            self.expr(s[1], scopes, kind)

    def returns(self, s):
        """does every path through s end in a return?"""
        k = s[0]
        if k == "return":
            return True
        if k in ("block", "seq"):
            return any(self.returns(x) for x in s[1])
        if k == "if":
            return s[3] is not None and self.returns(s[2]) and self.returns(s[3])
        return False            # (a loop may run zero times)

    def run(self):
        ar = self.ar
        for name, d in ar.functions.items():
            _, _, params, body, pos = d
            scopes = [{p: "var" for p in params}]
            if len(set(params)) != len(params):
                self.fail("function %s has two parameters of the same name" % name, pos)
            self.stmt(body, scopes, "function", False)
            if not self.returns(body):
                self.fail("In function %s there are paths without return" % name, pos)
        for name, d in ar.templates.items():
            _, _, params, body, flags, pos = d
            if len(set(params)) != len(params):
                self.fail("template %s has two parameters of the same name" % name, pos)
            self.stmt(body, [{p: "var" for p in params}], "template", False)
        for name, d in ar.buses.items():
            _, _, params, body, pos = d
            self.stmt(body, [{p: "var" for p in params}], "bus", False)
        if ar.main is not None:
            _, public, init, pos = ar.main
            e = init[1] if init[0] == "parallel" else init
            if e[0] == "call" and e[1] in ar.templates:
                self.expr(e, [{}], "main")
                body = ar.templates[e[1]][3]
                inputs = set()

                def collect(s):
                    if s[0] in ("block", "seq"):
                        for x in s[1]:
                            collect(x)
                    elif s[0] == "if":
                        collect(s[2])
                        if s[3] is not None:
                            collect(s[3])
                    elif s[0] == "decl" and s[1][0] in ("signal", "bus") and (s[1][1] if s[1][0] == "signal" else s[1][3]) == "input":
                        inputs.add(s[2])
                collect(body)
                for p in public:
                    if p not in inputs:
                        self.fail("public signal %s is not an input of main" % p, pos)


def check_archive(archive):
    _Checker(archive).run()

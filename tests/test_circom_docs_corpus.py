"""The reference's own DOCUMENTATION as a corpus for the circom-text front-end: every code block of mkdocs/docs that is circom
source (not pseudo-code with `...`, not a compiler message) must parse; every complete program (one with `component main`)
must either compile to a flat circuit or - where the documentation presents it as an ERRONEOUS example - be rejected with an
error.  Which examples the documentation calls erroneous is listed below by file and by a phrase of the surrounding text.

The documentation lives in /root/reference (absent on the GPU box): the test is skipped without it.  Nothing is copied: the
snippets are read from the reference tree while the test runs."""
import glob
import os
import re

import pytest

from circom_amd.frontend.circom_exec import program_from_text
from circom_amd.frontend.circom_lang import CircomSyntaxError, parse_text
from circom_amd.frontend.dsl import CircuitError
from circom_amd.frontend.flatten import flatten

DOCS = os.path.join(os.environ.get("CIRCOM_REF", "/root/reference"), "mkdocs", "docs")
pytestmark = pytest.mark.skipif(not os.path.isdir(DOCS), reason="reference documentation not present")

# blocks that are not circom source: prose, compiler output, intentionally wrong syntax (the text around them says so)
NOT_SOURCE = {
    ("circom-language/buses.md", "Parameters of a bus must be known"),            # compiler message
    ("circom-language/variables-and-mutability.md", "if (x = 3)"),                  # "assignment as expression" = the error shown
}
# complete programs the documentation shows BECAUSE they fail to compile: (file, phrase inside the program)
ERRONEOUS = [
    ("circom-insight/compiler-messages.md", "in = 1;"),                              # signal assigned with =
    ("circom-insight/unknowns.md", "out <== array[in];"),                            # unknown index in a constraint
    ("circom-insight/unknowns.md", "var array[in];"),                                # unknown array size
    ("circom-insight/unknowns.md", "b(in) out;"),                                    # unknown bus parameter
    ("circom-insight/unknowns.md", "if (in < 0){"),                                  # constraint under an unknown condition
    ("circom-insight/unknowns.md", "for (var i = 0; i < in; i++){"),                 # constraint under an unknown loop
    ("code-quality/code-assertion.md", "component main = A(0);"),                    # false assert reached
    ("circom-language/control-flow.md", "template wrong(N1)"),                       # component under an unknown condition
    ("circom-language/scoping.md", "signal aux;"),                                   # signal declared inside a for block
    ("circom-language/scoping.md", "c3.comp2.x"),                                    # signal of a nested sub-component
    ("circom-language/signals.md", "We do not declare it as output"),                # intermediate signal of a sub-component
    ("circom-language/signals.md", "out <== comp.outA;\n}\n\ncomponent main = B();"),  # (the two programs of signals.md that end
    ("circom-language/signals.md", "while (i < in){"),                               #  like this: assigned twice / unknown loop)
    ("circom-language/templates-and-components.md", "component c = A(a,N);"),       # unknown template parameter
    ("circom-language/templates-and-components.md", "a <== N;"),                    # own input assigned inside the template
    ("circom-language/templates-and-components.md", "c.in[1] is not assigned yet"),  # output of a component read before its inputs are in
]


def _blocks():
    """(file, line, text) of every fenced block that is marked text / circom / unmarked and mentions a circom keyword; fences are
    followed line by line (some blocks of the documentation close their fence at the end of a line of text)"""
    out = []
    for f in sorted(glob.glob(os.path.join(DOCS, "**", "*.md"), recursive=True)):
        rel = os.path.relpath(f, DOCS)
        inside, fence, lang, start, buf = False, "", "", 0, []
        for n, ln in enumerate(open(f).read().split("\n"), 1):
            st = ln.strip()
            if not inside:
                m = re.match(r"^(`{3,})\s*(\w*)\s*$", st)
                if m:
                    inside, fence, lang, start, buf = True, m.group(1), m.group(2), n, []
                continue
            if st.startswith(fence) or st.endswith("```"):
                code = "\n".join(buf + ([st[:-3]] if not st.startswith(fence) else [])) + "\n"
                if lang in ("text", "circom", "") and re.search(r"\b(template|function|signal|component|var)\b", code):
                    out.append((rel, start, code))
                inside = False
                continue
            buf.append(ln)
    return out


def _is_pseudo(code):
    return any(t in code for t in ("...", ". . .", "`", "–", "error[", "warning[", "/*Code from", 'In template "')) or "(input/output)" in code


def test_every_circom_block_of_the_documentation_parses():
    blocks = _blocks()
    assert len(blocks) > 90
    parsed = failed = 0
    problems = []
    for rel, line, code in blocks:
        if _is_pseudo(code) or any(rel == f and p in code for f, p in NOT_SOURCE):
            continue
        top = re.search(r"^\s*(template|function|bus)\b", code, re.M) or re.search(r"^\s*component\s+main\b", code, re.M)
        src = code if top else "template Wrap__() {\n" + code + "\n}"
        # the documentation closes bus definitions with `};` - the grammar (lang.lalrpop:137-145) has no semicolon there
        src = re.sub(r"^\};", "}", src, flags=re.M)
        src = re.sub(r'^\s*include\s+"[^"\n]*"\s*;', "", src, flags=re.M)          # (circomlib is not in the reference tree)
        try:
            parse_text(src, rel)
            parsed += 1
        except CircomSyntaxError as ex:
            # fragments cut out of a larger program (an unbalanced closing brace) are not sources either
            if src.count("{") != src.count("}") or "contains a total of" in code:
                continue
            failed += 1
            problems.append((rel, line, str(ex)))
    assert not problems, problems
    assert parsed >= 80


def test_complete_programs_of_the_documentation_compile_or_fail_as_documented():
    compiled, rejected, problems = 0, 0, []
    for rel, line, code in _blocks():
        if not re.search(r"^\s*component\s+main\b", code, re.M) or _is_pseudo(code) or "include" in code:
            continue
        if not re.search(r"^\s*template\b", code, re.M):
            continue                                             # a lone `component main = ...;` line
        wrong = any(rel.endswith(f) and p in code for f, p in ERRONEOUS)
        try:
            fc = flatten(program_from_text(code, name=rel))
            assert fc.n_signals >= 1
            if wrong:
                problems.append((rel, line, "compiles although the documentation shows it as an error"))
            compiled += 1
        except (CircomSyntaxError, CircuitError) as ex:
            if not wrong:
                problems.append((rel, line, str(ex)))
            rejected += 1
    assert not problems, problems
    assert compiled >= 20 and rejected >= 13

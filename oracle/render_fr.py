"""Render the reference's `generic/fr.{hpp,cpp}` handlebars templates for a prime.

TEST INFRASTRUCTURE ONLY (oracle). Nothing on the product path may import this.

The reference ships its portable (`--no_asm`) field library as a handlebars
template and renders it in Rust (`code_producers/src/c_elements/
c_code_generator.rs:1076-1128`, hpp at `:1004-1013`).  The Rust toolchain is
absent here, so this module re-implements the handlebars *subset* those two
templates use and computes the parameters with the same formulas.  Input is
read from `/root/reference` where it lies; output goes to `oracle/_ref/<prime>/`
(git-ignored).  No reference source is stored in this repository.

Supported syntax: `{{name}}`, `{{ name }}`, `{{#if x}}..{{else}}..{{/if}}`
(also `{{ else }}`), `{{#each list}}..{{/each}}` with `{{@index}}`, `{{this}}`,
`{{#if @last}}`, helpers `inc`, `dec`, `elements`, nested `(inc @index)`.
"""
from __future__ import annotations

import re
import sys
from pathlib import Path

PRIMES = {
    # program_structure/src/utils/constants.rs:3-13
    "bn128": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
    "bls12381": 52435875175126190479447740508185965837690552500527637822603658699938581184513,
    "goldilocks": 18446744069414584321,
    "grumpkin": 21888242871839275222246405745257275088696311157297823662689037894645226208583,
    "pallas": 28948022309329048855892746252171976963363056481941560715954676764349967630337,
    "vesta": 28948022309329048855892746252171976963363056481941647379679742748393362948097,
    "secq256r1": 115792089210356248762697446949407573530086143415290314195533631308867097853951,
    "bls12377": 8444461749428370424248824938781546531375899335154063827935233455917409239041,
}

TOKEN = re.compile(r"\{\{(.*?)\}\}", re.S)


def _u64_list(v: int, n64: int):
    # c_code_generator.rs:1058-1074 — little-endian u64 words as 0x%x
    return ["0x%x" % ((v >> (64 * i)) & (2**64 - 1)) for i in range(n64)]


def params_for(p: int) -> dict:
    """Template parameters, formulas of c_code_generator.rs:1089-1128."""
    pbits = p.bit_length()
    n64 = (pbits + 63) // 64
    nbits = n64 * 64
    inv = pow(p, -1, 1 << 64)
    np_ = (1 << 64) - inv
    lbo = ((1 << 64) >> (nbits - pbits)) - 1
    return {
        "cannotOptimize": (p >> ((n64 - 1) * 64)) > (((1 << 64) - 1) >> 1) - 1,
        "list0n64": list(range(n64)),
        "list0n64_1": list(range(n64 - 1)),
        "list1n64": list(range(1, n64)),
        "n64": n64,
        "fr_n64": n64,
        "qbits": pbits,
        "lboMask": "0x%x" % lbo,
        "fr_np": "0x%x" % np_,
        "fr_q_list": _u64_list(p, n64),
        "fr_r2_list": _u64_list(pow(2, 2 * nbits, p), n64),
        "fr_r3_list": _u64_list(pow(2, 3 * nbits, p), n64),
        "half_list": _u64_list(p // 2, n64),
    }


def _parse(src: str):
    """Token stream -> nested node list."""
    pos = 0
    root: list = []
    stack = [root]
    frames = []  # (kind, node)
    for m in TOKEN.finditer(src):
        if m.start() > pos:
            stack[-1].append(("text", src[pos:m.start()]))
        pos = m.end()
        tag = m.group(1).strip()
        if tag.startswith("#if"):
            node = ["if", tag[3:].strip(), [], []]
            stack[-1].append(node)
            frames.append(node)
            stack.append(node[2])
        elif tag == "else":
            node = frames[-1]
            assert node[0] == "if", "else outside if"
            stack.pop()
            stack.append(node[3])
        elif tag == "/if":
            node = frames.pop()
            assert node[0] == "if"
            stack.pop()
        elif tag.startswith("#each"):
            node = ["each", tag[5:].strip(), []]
            stack[-1].append(node)
            frames.append(node)
            stack.append(node[2])
        elif tag == "/each":
            node = frames.pop()
            assert node[0] == "each"
            stack.pop()
        else:
            stack[-1].append(("expr", tag))
    if pos < len(src):
        stack[-1].append(("text", src[pos:]))
    assert not frames, "unterminated block"
    return root


def _eval(expr: str, ctx: list):
    expr = expr.strip()
    while expr.startswith("(") and expr.endswith(")"):
        expr = expr[1:-1].strip()
    parts = expr.split(None, 1)
    if len(parts) == 2 and parts[0] in ("inc", "dec", "elements"):
        v = _eval(parts[1], ctx)
        if parts[0] == "inc":
            return int(v) + 1
        if parts[0] == "dec":
            return int(v) - 1
        return ",".join(str(x) for x in v)
    assert len(parts) == 1, "unsupported expression: %r" % expr
    name = parts[0]
    for frame in reversed(ctx):
        if name in frame:
            return frame[name]
    raise KeyError(name)


def _render(nodes, ctx, out):
    for n in nodes:
        kind = n[0]
        if kind == "text":
            out.append(n[1])
        elif kind == "expr":
            out.append(str(_eval(n[1], ctx)))
        elif kind == "if":
            branch = n[2] if _eval(n[1], ctx) else n[3]
            _render(branch, ctx, out)
        elif kind == "each":
            lst = _eval(n[1], ctx)
            for i, item in enumerate(lst):
                frame = {"@index": i, "this": item, "@last": i == len(lst) - 1}
                _render(n[2], ctx + [frame], out)


def render(template: str, params: dict) -> str:
    out: list = []
    _render(_parse(template), [params], out)
    return "".join(out)


def render_prime(prime: str, ref_root: Path, out_dir: Path) -> None:
    p = PRIMES[prime]
    params = params_for(p)
    gen = ref_root / "code_producers/src/c_elements/generic"
    out_dir.mkdir(parents=True, exist_ok=True)
    for name in ("fr.hpp", "fr.cpp"):
        text = (gen / name).read_text()
        (out_dir / name).write_text(render(text, params))


if __name__ == "__main__":
    prime, ref_root, out_dir = sys.argv[1:4]
    render_prime(prime, Path(ref_root), Path(out_dir))

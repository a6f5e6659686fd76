"""ctypes binding to oracle/_ref/<prime>/libfr_shim.so (the compiled *reference* field library).

TEST INFRASTRUCTURE ONLY.  See oracle/fr_shim.cpp for the C side.
"""
from __future__ import annotations

import ctypes
from pathlib import Path

BINOPS = ("add", "sub", "mul", "div", "idiv", "mod", "pow", "shl", "shr", "band", "bor", "bxor",
          "eq", "neq", "lt", "gt", "leq", "geq", "land", "lor")
UNOPS = ("neg", "bnot", "lnot", "inv", "square")
REP_AUTO, REP_LONG, REP_MONT, REP_SHORT = 0, 1, 2, 3


class RefFr:
    def __init__(self, so_path: Path):
        self.lib = ctypes.CDLL(str(so_path))
        self.lib.ofr_binop.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int,
                                       ctypes.c_char_p, ctypes.c_int]
        self.lib.ofr_unop.argtypes = [ctypes.c_int, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
        self.lib.ofr_is_true.argtypes = [ctypes.c_char_p, ctypes.c_int]
        self.lib.ofr_to_int.argtypes = [ctypes.c_char_p, ctypes.c_int]
        self.lib.ofr_str2element.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint]
        self.lib.ofr_q.argtypes = [ctypes.c_char_p]
        self.lib.ofr_raw_mmul.argtypes = [ctypes.c_char_p] * 3
        self.lib.ofr_mul_chain.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_uint64]
        buf = ctypes.create_string_buffer(32)
        self.lib.ofr_q(buf)
        self.q = int.from_bytes(buf.raw, "little")

    @staticmethod
    def _b(x: int) -> bytes:
        return x.to_bytes(32, "little")

    def binop(self, name: str, a: int, b: int, repa=REP_AUTO, repb=REP_AUTO) -> int:
        out = ctypes.create_string_buffer(32)
        rc = self.lib.ofr_binop(BINOPS.index(name), out, self._b(a), repa, self._b(b), repb)
        if rc:
            raise ValueError("operand not representable (rc=%d)" % rc)
        return int.from_bytes(out.raw, "little")

    def unop(self, name: str, a: int, repa=REP_AUTO) -> int:
        out = ctypes.create_string_buffer(32)
        rc = self.lib.ofr_unop(UNOPS.index(name), out, self._b(a), repa)
        if rc:
            raise ValueError("operand not representable (rc=%d)" % rc)
        return int.from_bytes(out.raw, "little")

    def is_true(self, a: int, repa=REP_AUTO) -> bool:
        return bool(self.lib.ofr_is_true(self._b(a), repa))

    def to_int(self, a: int, repa=REP_AUTO) -> int:
        return self.lib.ofr_to_int(self._b(a), repa)

    def str2element(self, s: str, base: int) -> int:
        out = ctypes.create_string_buffer(32)
        self.lib.ofr_str2element(out, s.encode(), base)
        return int.from_bytes(out.raw, "little")

    def raw_mmul(self, a: int, b: int) -> int:
        out = ctypes.create_string_buffer(32)
        self.lib.ofr_raw_mmul(out, self._b(a), self._b(b))
        return int.from_bytes(out.raw, "little")

    def mul_chain(self, a: int, b: int, n: int) -> int:
        out = ctypes.create_string_buffer(32)
        self.lib.ofr_mul_chain(out, self._b(a), self._b(b), n)
        return int.from_bytes(out.raw, "little")

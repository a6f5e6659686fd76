"""CPU execution of the emitted bit-plane program's IR (circom_amd/hip_elements/bitjit.py) — TEST INFRASTRUCTURE ONLY.

Executes exactly what the assembly does, on `width` instances at once (a register = one Python int whose bit i is the value
in instance i; all lanes of the GPU run the same instruction stream on their own 32 instances, so one wide "lane" stands for
them), with the failure modes of the hardware made loud:
  * a register is POISON until written; reading poison raises,
  * vector memory is an in-order queue: a loaded register holds its value only after an `s_waitcnt vmcnt(n)` that covers
    the load; reading it earlier, or overwriting it while the load is in flight (the data would land on the new value),
    raises; a load of a row whose store by this wave has not completed raises (the allocator must have waited),
  * every row is written at most once and never before it is read as an input.
"""
from __future__ import annotations


class JitHazard(Exception):
    pass


def run_ir(jp, rows: dict, width: int):
    """rows: slot -> mask for the rows present before the kernel (constants 0 / ones and the main inputs).
    Returns (rows after the kernel, fallback mask, r1cs flag mask)."""
    full = (1 << width) - 1
    V = [None] * jp.n_vgpr
    A = [None] * jp.n_agpr
    pend = {}                       # vgpr -> (issue index, value)
    V[1] = 0
    V[2] = 0
    mem = dict(rows)
    written = {}                    # slot -> issue index of the store
    issued = 0
    done = -1

    def rd(x, i):
        if x == -1:
            return 0
        if x == -2:
            return full
        if x in pend:
            raise JitHazard("instruction %d reads v%d while its load is in flight" % (i, x))
        v = V[x]
        if v is None:
            raise JitHazard("instruction %d reads v%d before it is written" % (i, x))
        return v

    def wr(x, val, i):
        if x in pend:
            raise JitHazard("instruction %d overwrites v%d while a load into it is in flight" % (i, x))
        V[x] = val

    # a loop (bitjit.lower_jit(loop=True)): the body runs once per iteration with its rows relative to the iteration's base and
    # the values from outside the iteration read through the iteration's table; NOTHING may live in a register across an
    # iteration boundary (the allocator promises it: every register is poisoned there) and no memory operation may be in flight
    prog = []
    i = 0
    ir = jp.ir
    while i < len(ir):
        if ir[i][0] == "loop":
            j = i + 1
            while ir[j][0] != "endloop":
                j += 1
            _, K, R, base, tab = ir[i]
            for it in range(K):
                prog.append((i, ("iter", it)))
                for q in range(i + 1, j):
                    x = ir[q]
                    if x[0] in ("stL", "staL", "ldL"):
                        x = (x[0][:-1], x[1], base + it * R + x[2])
                    elif x[0] == "ldx":
                        x = ("ld", x[1], int(tab[it][x[2]]))
                    prog.append((q, x))
            prog.append((j, ("iter", -1)))
            i = j + 1
        else:
            prog.append((i, ir[i]))
            i += 1

    for i, ins in prog:
        k = ins[0]
        if k == "iter":
            if pend or done != issued - 1:
                raise JitHazard("memory operations in flight across a loop boundary (instruction %d)" % i)
            for r in range(3, len(V)):
                V[r] = None
            for r in range(len(A)):
                A[r] = None
            continue
        if k == "g":
            s0, s1, s2 = rd(ins[2], i), rd(ins[3], i), rd(ins[4], i)
            tt = ins[5]
            r = 0
            for m in range(8):
                if (tt >> m) & 1:
                    t = full
                    t &= s0 if m & 4 else ~s0
                    t &= s1 if m & 2 else ~s1
                    t &= s2 if m & 1 else ~s2
                    r |= t & full
            wr(ins[1], r, i)
        elif k == "st" or k == "sta":
            val = rd(ins[1], i) if k == "st" else A[ins[1]]
            if val is None:
                raise JitHazard("instruction %d stores an unwritten register" % i)
            slot = ins[2]
            if slot in mem:
                raise JitHazard("instruction %d writes row %d a second time (or over an input)" % (i, slot))
            if not 0 <= slot < jp.n_slots:
                raise JitHazard("row %d outside the chunk" % slot)
            mem[slot] = val
            written[slot] = issued
            issued += 1
        elif k == "ld":
            slot = ins[2]
            if slot not in mem:
                raise JitHazard("instruction %d loads row %d which holds nothing" % (i, slot))
            if slot in written and written[slot] > done:
                raise JitHazard("instruction %d loads row %d before its store completed" % (i, slot))
            if ins[1] in pend:
                raise JitHazard("instruction %d loads into v%d while a load into it is in flight" % (i, ins[1]))
            V[ins[1]] = None
            pend[ins[1]] = (issued, mem[slot])
            issued += 1
        elif k == "w":
            done = max(done, issued - 1 - ins[1])
            for x in [x for x, (idx, _) in pend.items() if idx <= done]:
                V[x] = pend.pop(x)[1]
        elif k == "aw":
            A[ins[1]] = rd(ins[2], i)
        elif k == "ar":
            if A[ins[2]] is None:
                raise JitHazard("instruction %d reads a%d before it is written" % (i, ins[2]))
            wr(ins[1], A[ins[2]], i)
        elif k == "acc":
            V[ins[1]] |= rd(ins[2], i)
        elif k == "acc3":
            V[ins[1]] |= rd(ins[2], i) | rd(ins[3], i)
        elif k == "accc":
            V[ins[1]] = full
        else:
            raise ValueError(k)
    return mem, V[1], V[2]

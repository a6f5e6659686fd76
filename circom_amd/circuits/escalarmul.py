"""BabyJubjub scalar multiplication the way circomlib structures it (montgomery.circom, escalarmulany.circom,
escalarmulfix.circom, mux3.circom) - circomlib itself is absent from the reference tree, so the templates are re-authored
from the constructions, and every one of them is pinned against plain-integer Edwards arithmetic in tests/test_escalarmul.py:

  * points travel in MONTGOMERY form (u, v) = ((1 + y) / (1 - y), u / x) inside a segment: an addition or a doubling is one
    division hint (`lamda`) + three constraints instead of BabyAdd's two hints + six,
  * Montgomery addition is incomplete (it fails for equal u-coordinates), so every chain starts from an offset and subtracts it
    at the end with ONE complete Edwards addition: EscalarMulAny accumulates P + sum_{i >= 1} e_i 2^i P and subtracts P when
    e_0 = 0; EscalarMulFix adds (w_i + 1) 8^i B per 3-bit window w_i (an 8-entry table 1B..8B behind a MultiMux3) on top of
    2 * 8^nW B and subtracts sum_i 8^i B + 2 * 8^nW B,
  * long scalars are cut into segments (148 bits / 82 windows) whose partial results are added in Edwards form.

This is the relation BASELINE config 4 names (EdDSA as circomlib has it) next to the bit-serial ladder of babyjub.py, which
stays as the first relation (`semaphore20`, `semaphore20p`); `SemaphoreStyle(levels, mode="window")` = `semaphore20w`."""
from ..frontend.dsl import template
from .babyjub import A as ED_A, D as ED_D, BASE8, BabyAdd
from .basic import IsZero

# Montgomery form of BabyJubjub: B v^2 = u^3 + A u^2 + u with A = 2 (a + d) / (a - d), B = 4 / (a - d)
MONT_A = 168698
MONT_B = 1
assert 2 * (ED_A + ED_D) == MONT_A * (ED_A - ED_D) and 4 == MONT_B * (ED_A - ED_D)


@template
def Edwards2Montgomery(c):
    inp = c.input("in", 2)
    out = c.output("out", 2)
    c.hint(out[0], (1 + inp[1]) / (1 - inp[1]))
    c.hint(out[1], out[0] / inp[0])
    c.enforce(out[0] * (1 - inp[1]), 1 + inp[1])
    c.enforce(out[1] * inp[0], out[0])


@template
def Montgomery2Edwards(c):
    inp = c.input("in", 2)
    out = c.output("out", 2)
    c.hint(out[0], inp[0] / inp[1])
    c.hint(out[1], (inp[0] - 1) / (inp[0] + 1))
    c.enforce(out[0] * inp[1], inp[0])
    c.enforce(out[1] * (inp[0] + 1), inp[0] - 1)


@template
def MontgomeryAdd(c):
    in1 = c.input("in1", 2)
    in2 = c.input("in2", 2)
    out = c.output("out", 2)
    lamda = c.signal("lamda")
    c.hint(lamda, (in2[1] - in1[1]) / (in2[0] - in1[0]))
    c.enforce(lamda * (in2[0] - in1[0]), in2[1] - in1[1])
    c.set(out[0], MONT_B * lamda * lamda - MONT_A - in1[0] - in2[0])
    c.set(out[1], lamda * (in1[0] - out[0]) - in1[1])


@template
def MontgomeryDouble(c):
    inp = c.input("in", 2)
    out = c.output("out", 2)
    lamda = c.signal("lamda")
    x1_2 = c.signal("x1_2")
    c.set(x1_2, inp[0] * inp[0])
    c.hint(lamda, (3 * x1_2 + 2 * MONT_A * inp[0] + 1) / (2 * MONT_B * inp[1]))
    c.enforce(lamda * (2 * MONT_B * inp[1]), 3 * x1_2 + 2 * MONT_A * inp[0] + 1)
    c.set(out[0], MONT_B * lamda * lamda - MONT_A - 2 * inp[0])
    c.set(out[1], lamda * (inp[0] - out[0]) - inp[1])


SEG_ANY = 148


# ---- any point: bit-serial Montgomery ladder per segment (escalarmulany.circom) ---------------------------------------------
@template
def Multiplexor2(c):
    sel = c.input("sel")
    inp = c.input("in", 2, 2)
    out = c.output("out", 2)
    c.set(out[0], (inp[1][0] - inp[0][0]) * sel + inp[0][0])
    c.set(out[1], (inp[1][1] - inp[0][1]) * sel + inp[0][1])


@template
def BitElementMulAny(c):
    sel = c.input("sel")
    dbl_in = c.input("dblIn", 2)
    add_in = c.input("addIn", 2)
    dbl_out = c.output("dblOut", 2)
    add_out = c.output("addOut", 2)
    doubler = c.component("doubler", MontgomeryDouble())
    adder = c.component("adder", MontgomeryAdd())
    selector = c.component("selector", Multiplexor2())
    c.set(selector["sel"], sel)
    for k in range(2):
        c.set(doubler["in"][k], dbl_in[k])
    for k in range(2):
        c.set(adder["in1"][k], doubler["out"][k])
        c.set(adder["in2"][k], add_in[k])
        c.set(selector["in"][0][k], add_in[k])
    for k in range(2):
        c.set(selector["in"][1][k], adder["out"][k])
    for k in range(2):                                     # (outputs are read once the sub-components have all their inputs)
        c.set(dbl_out[k], doubler["out"][k])
        c.set(add_out[k], selector["out"][k])


@template
def SegmentMulAny(c, n):
    """out = (sum e_i 2^i) p for n <= 148 bits; dbl = 2^(n-1) p in Montgomery form (the next segment doubles it once more)"""
    assert 2 <= n <= SEG_ANY
    e = c.input("e", n)
    p = c.input("p", 2)
    out = c.output("out", 2)
    dbl = c.output("dbl", 2)
    e2m = c.component("e2m", Edwards2Montgomery())
    c.set(e2m["in"][0], p[0]); c.set(e2m["in"][1], p[1])
    bits = []
    for i in range(n - 1):
        b = c.component("bits", BitElementMulAny(), i)
        src_d = e2m["out"] if i == 0 else bits[i - 1]["dblOut"]
        src_a = e2m["out"] if i == 0 else bits[i - 1]["addOut"]
        for k in range(2):
            c.set(b["dblIn"][k], src_d[k])
            c.set(b["addIn"][k], src_a[k])
        c.set(b["sel"], e[i + 1])
        bits.append(b)
    c.set(dbl[0], bits[n - 2]["dblOut"][0]); c.set(dbl[1], bits[n - 2]["dblOut"][1])
    m2e = c.component("m2e", Montgomery2Edwards())
    c.set(m2e["in"][0], bits[n - 2]["addOut"][0]); c.set(m2e["in"][1], bits[n - 2]["addOut"][1])
    # the chain carries p + sum_{i >= 1} e_i 2^i p: take p away again when e_0 = 0 (one complete Edwards addition)
    eadder = c.component("eadder", BabyAdd())
    c.set(eadder["x1"], m2e["out"][0]); c.set(eadder["y1"], m2e["out"][1])
    c.set(eadder["x2"], -p[0]); c.set(eadder["y2"], p[1])
    last = c.component("lastSel", Multiplexor2())
    c.set(last["sel"], e[0])
    c.set(last["in"][0][0], eadder["xout"]); c.set(last["in"][0][1], eadder["yout"])
    c.set(last["in"][1][0], m2e["out"][0]); c.set(last["in"][1][1], m2e["out"][1])
    c.set(out[0], last["out"][0]); c.set(out[1], last["out"][1])


@template
def EscalarMulAny(c, n):
    """out = (sum e_i 2^i) p for any point p of the curve (the identity gives the identity: the ladder then runs on BASE8 and
    its result is masked)"""
    e = c.input("e", n)
    p = c.input("p", 2)
    out = c.output("out", 2)
    nseg = (n - 1) // SEG_ANY + 1
    nlast = n - (nseg - 1) * SEG_ANY
    zero = c.component("zeropoint", IsZero())
    c.set(zero["in"], p[0])
    segs, adders = [], []
    for s in range(nseg):
        ns = SEG_ANY if s < nseg - 1 else nlast
        seg = c.component("segments", SegmentMulAny(ns), s)
        for i in range(ns):
            c.set(seg["e"][i], e[s * SEG_ANY + i])
        if s == 0:
            c.set(seg["p"][0], p[0] + (BASE8[0] - p[0]) * zero["out"])
            c.set(seg["p"][1], p[1] + (BASE8[1] - p[1]) * zero["out"])
        else:
            dblr = c.component("doublers", MontgomeryDouble(), s - 1)
            m2e = c.component("m2e", Montgomery2Edwards(), s - 1)
            add = c.component("adders", BabyAdd(), s - 1)
            for k in range(2):
                c.set(dblr["in"][k], segs[s - 1]["dbl"][k])
            for k in range(2):
                c.set(m2e["in"][k], dblr["out"][k])
            c.set(seg["p"][0], m2e["out"][0]); c.set(seg["p"][1], m2e["out"][1])
            prev = (segs[0]["out"][0], segs[0]["out"][1]) if s == 1 else (adders[s - 2]["xout"], adders[s - 2]["yout"])
            c.set(add["x1"], prev[0]); c.set(add["y1"], prev[1])
            c.set(add["x2"], seg["out"][0]); c.set(add["y2"], seg["out"][1])
            adders.append(add)
        segs.append(seg)
    rx, ry = (segs[0]["out"][0], segs[0]["out"][1]) if nseg == 1 else (adders[nseg - 2]["xout"], adders[nseg - 2]["yout"])
    c.set(out[0], rx * (1 - zero["out"]))
    c.set(out[1], ry + (1 - ry) * zero["out"])


# ---- a fixed base: 3-bit windows behind an 8-entry table (escalarmulfix.circom, mux3.circom) -------------------------------
@template
def MultiMux3(c, n):
    cc = c.input("c", n, 8)
    s = c.input("s", 3)
    out = c.output("out", n)
    s10 = c.signal("s10")
    c.set(s10, s[1] * s[0])
    a210 = c.signal("a210", n); a21 = c.signal("a21", n); a20 = c.signal("a20", n); a2 = c.signal("a2", n)
    a10 = c.signal("a10", n); a1 = c.signal("a1", n); a0 = c.signal("a0", n); a = c.signal("a", n)
    for i in range(n):
        k = cc[i]
        c.set(a210[i], (k[7] - k[6] - k[5] + k[4] - k[3] + k[2] + k[1] - k[0]) * s10)
        c.set(a21[i], (k[6] - k[4] - k[2] + k[0]) * s[1])
        c.set(a20[i], (k[5] - k[4] - k[1] + k[0]) * s[0])
        c.set(a2[i], k[4] - k[0])
        c.set(a10[i], (k[3] - k[2] - k[1] + k[0]) * s10)
        c.set(a1[i], (k[2] - k[0]) * s[1])
        c.set(a0[i], (k[1] - k[0]) * s[0])
        c.set(a[i], k[0])
        c.set(out[i], (a210[i] + a21[i] + a20[i] + a2[i]) * s[2] + (a10[i] + a1[i] + a0[i] + a[i]))


@template
def WindowMulFix(c):
    """out = (in + 1) * base for the 3-bit number `in`, out8 = 8 * base (Montgomery form)"""
    inp = c.input("in", 3)
    base = c.input("base", 2)
    out = c.output("out", 2)
    out8 = c.output("out8", 2)
    mux = c.component("mux", MultiMux3(2))
    for j in range(3):
        c.set(mux["s"][j], inp[j])
    dbl2 = c.component("dbl2", MontgomeryDouble())
    c.set(dbl2["in"][0], base[0]); c.set(dbl2["in"][1], base[1])
    c.set(mux["c"][0][0], base[0]); c.set(mux["c"][1][0], base[1])
    c.set(mux["c"][0][1], dbl2["out"][0]); c.set(mux["c"][1][1], dbl2["out"][1])
    prev = dbl2["out"]
    for k in range(3, 9):                                  # k * base = base + (k - 1) * base
        adr = c.component("adr%d" % k, MontgomeryAdd())
        c.set(adr["in1"][0], base[0]); c.set(adr["in1"][1], base[1])
        c.set(adr["in2"][0], prev[0]); c.set(adr["in2"][1], prev[1])
        c.set(mux["c"][0][k - 1], adr["out"][0]); c.set(mux["c"][1][k - 1], adr["out"][1])
        prev = adr["out"]
    c.set(out8[0], prev[0]); c.set(out8[1], prev[1])
    c.set(out[0], mux["out"][0]); c.set(out[1], mux["out"][1])


@template
def SegmentMulFix(c, n_windows):
    """out = (sum e_i 2^i) base for 3 * n_windows bits (Edwards form), dbl = 8^n_windows * base (Montgomery form)"""
    e = c.input("e", n_windows * 3)
    base = c.input("base", 2)
    out = c.output("out", 2)
    dbl = c.output("dbl", 2)
    e2m = c.component("e2m", Edwards2Montgomery())
    c.set(e2m["in"][0], base[0]); c.set(e2m["in"][1], base[1])
    windows = []
    for i in range(n_windows):
        w = c.component("windows", WindowMulFix(), i)
        src = e2m["out"] if i == 0 else windows[i - 1]["out8"]
        c.set(w["base"][0], src[0]); c.set(w["base"][1], src[1])
        for j in range(3):
            c.set(w["in"][j], e[3 * i + j])
        windows.append(w)
    # the offset both chains start from / end with: 2 * 8^n_windows * base (no partial sum ever meets its summand)
    dbl_last = c.component("dblLast", MontgomeryDouble())
    c.set(dbl_last["in"][0], windows[-1]["out8"][0]); c.set(dbl_last["in"][1], windows[-1]["out8"][1])
    # cadders: sum_i 8^i base + offset (what the window encoding (w + 1) adds on top of the scalar)
    cadders = []
    for i in range(n_windows):
        ca = c.component("cadders", MontgomeryAdd(), i)
        a1 = e2m["out"] if i == 0 else cadders[i - 1]["out"]
        a2 = windows[i]["out8"] if i < n_windows - 1 else dbl_last["out"]
        c.set(ca["in1"][0], a1[0]); c.set(ca["in1"][1], a1[1])
        c.set(ca["in2"][0], a2[0]); c.set(ca["in2"][1], a2[1])
        cadders.append(ca)
    adders = []
    for i in range(n_windows):
        ad = c.component("adders", MontgomeryAdd(), i)
        a1 = dbl_last["out"] if i == 0 else adders[i - 1]["out"]
        c.set(ad["in1"][0], a1[0]); c.set(ad["in1"][1], a1[1])
        c.set(ad["in2"][0], windows[i]["out"][0]); c.set(ad["in2"][1], windows[i]["out"][1])
        adders.append(ad)
    m2e = c.component("m2e", Montgomery2Edwards())
    cm2e = c.component("cm2e", Montgomery2Edwards())
    c.set(m2e["in"][0], adders[-1]["out"][0]); c.set(m2e["in"][1], adders[-1]["out"][1])
    c.set(cm2e["in"][0], cadders[-1]["out"][0]); c.set(cm2e["in"][1], cadders[-1]["out"][1])
    cadd = c.component("cAdd", BabyAdd())
    c.set(cadd["x1"], m2e["out"][0]); c.set(cadd["y1"], m2e["out"][1])
    c.set(cadd["x2"], -cm2e["out"][0]); c.set(cadd["y2"], cm2e["out"][1])
    c.set(out[0], cadd["xout"]); c.set(out[1], cadd["yout"])
    c.set(dbl[0], windows[-1]["out8"][0]); c.set(dbl[1], windows[-1]["out8"][1])


WIN_SEG = 82            # windows per segment: 246 bits


@template
def EscalarMulFix(c, n, base):
    """out = (sum e_i 2^i) * base for a compile-time point `base` of the prime-order subgroup"""
    e = c.input("e", n)
    out = c.output("out", 2)
    nseg = (n - 1) // (3 * WIN_SEG) + 1
    nlast = n - (nseg - 1) * 3 * WIN_SEG
    segs, adders = [], []
    for s in range(nseg):
        nbits = 3 * WIN_SEG if s < nseg - 1 else nlast
        nwin = (nbits - 1) // 3 + 1
        seg = c.component("segments", SegmentMulFix(nwin), s)
        for i in range(nwin * 3):
            c.set(seg["e"][i], e[s * 3 * WIN_SEG + i] if i < nbits else 0)
        if s == 0:
            c.set(seg["base"][0], base[0]); c.set(seg["base"][1], base[1])
        else:
            m2e = c.component("m2e", Montgomery2Edwards(), s - 1)
            add = c.component("adders", BabyAdd(), s - 1)
            c.set(m2e["in"][0], segs[s - 1]["dbl"][0]); c.set(m2e["in"][1], segs[s - 1]["dbl"][1])
            c.set(seg["base"][0], m2e["out"][0]); c.set(seg["base"][1], m2e["out"][1])
            prev = (segs[0]["out"][0], segs[0]["out"][1]) if s == 1 else (adders[s - 2]["xout"], adders[s - 2]["yout"])
            c.set(add["x1"], prev[0]); c.set(add["y1"], prev[1])
            c.set(add["x2"], seg["out"][0]); c.set(add["y2"], seg["out"][1])
            adders.append(add)
        segs.append(seg)
    if nseg == 1:
        c.set(out[0], segs[0]["out"][0]); c.set(out[1], segs[0]["out"][1])
    else:
        c.set(out[0], adders[nseg - 2]["xout"]); c.set(out[1], adders[nseg - 2]["yout"])

"""SHA-256 circuit in the structure of circomlib's `circuits/sha256/` directory
(constants / xor3 / rotate / shift / sigma / ch / maj / t1 / t2 / sigmaplus / binsum /
sha256compression + its witness-side `sha256compression` function / sha256).

circomlib is not available in this container, so the templates are re-authored
from the algorithm with circomlib's decomposition: every 32-bit word is 32 bit-signals, modular
additions are `BinSum`s whose output bits are produced with `<--` (`(lin >> k) & 1`) and then
constrained, Ch/Maj/Xor3 are the usual degree-2 bit formulas.  The digest is pinned against
`hashlib.sha256` in tests/test_sha256.py.
"""
from ..frontend.dsl import template

H_INIT = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]
K_TABLE = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2,
]


def nbits(a):
    n, r = 1, 0
    while n - 1 < a:
        r += 1
        n *= 2
    return r


@template
def H(c, x):
    out = c.output("out", 32)
    for i in range(32):
        c.set(out[i], (H_INIT[x] >> i) & 1)


@template
def K(c, x):
    out = c.output("out", 32)
    for i in range(32):
        c.set(out[i], (K_TABLE[x] >> i) & 1)


@template
def Xor3(c, n):
    a = c.input("a", n)
    b = c.input("b", n)
    cc = c.input("c", n)
    out = c.output("out", n)
    mid = c.signal("mid", n)
    for k in range(n):
        c.set(mid[k], b[k] * cc[k])
        c.set(out[k], a[k] * (1 - 2 * b[k] - 2 * cc[k] + 4 * mid[k]) + b[k] + cc[k] - 2 * mid[k])


@template
def RotR(c, n, r):
    inp = c.input("in", n)
    out = c.output("out", n)
    for i in range(n):
        c.set(out[i], inp[(i + r) % n])


@template
def ShR(c, n, r):
    inp = c.input("in", n)
    out = c.output("out", n)
    for i in range(n):
        if i + r >= n:
            c.set(out[i], 0)
        else:
            c.set(out[i], inp[i + r])


@template
def SmallSigma(c, ra, rb, rc):
    inp = c.input("in", 32)
    out = c.output("out", 32)
    rota = c.component("rota", RotR(32, ra))
    rotb = c.component("rotb", RotR(32, rb))
    shrc = c.component("shrc", ShR(32, rc))
    for k in range(32):
        c.set(rota["in"][k], inp[k])
        c.set(rotb["in"][k], inp[k])
        c.set(shrc["in"][k], inp[k])
    xor3 = c.component("xor3", Xor3(32))
    for k in range(32):
        c.set(xor3["a"][k], rota["out"][k])
        c.set(xor3["b"][k], rotb["out"][k])
        c.set(xor3["c"][k], shrc["out"][k])
    for k in range(32):
        c.set(out[k], xor3["out"][k])


@template
def BigSigma(c, ra, rb, rc):
    inp = c.input("in", 32)
    out = c.output("out", 32)
    rota = c.component("rota", RotR(32, ra))
    rotb = c.component("rotb", RotR(32, rb))
    rotc = c.component("rotc", RotR(32, rc))
    for k in range(32):
        c.set(rota["in"][k], inp[k])
        c.set(rotb["in"][k], inp[k])
        c.set(rotc["in"][k], inp[k])
    xor3 = c.component("xor3", Xor3(32))
    for k in range(32):
        c.set(xor3["a"][k], rota["out"][k])
        c.set(xor3["b"][k], rotb["out"][k])
        c.set(xor3["c"][k], rotc["out"][k])
    for k in range(32):
        c.set(out[k], xor3["out"][k])


@template
def Ch_t(c, n):
    a = c.input("a", n)
    b = c.input("b", n)
    cc = c.input("c", n)
    out = c.output("out", n)
    for k in range(n):
        c.set(out[k], a[k] * (b[k] - cc[k]) + cc[k])


@template
def Maj_t(c, n):
    a = c.input("a", n)
    b = c.input("b", n)
    cc = c.input("c", n)
    out = c.output("out", n)
    mid = c.signal("mid", n)
    for k in range(n):
        c.set(mid[k], b[k] * cc[k])
        c.set(out[k], a[k] * (b[k] + cc[k] - 2 * mid[k]) + mid[k])


@template
def BinSum(c, n, ops):
    nout = nbits((2 ** n - 1) * ops)
    inp = c.input("in", ops, n)
    out = c.output("out", nout)
    lin = c.const(0)
    lout = c.const(0)
    e2 = 1
    for k in range(n):
        for j in range(ops):
            lin = lin + inp[j][k] * e2
        e2 = e2 + e2
    e2 = 1
    for k in range(nout):
        c.hint(out[k], (lin >> k) & 1)
        c.enforce(out[k] * (out[k] - 1), 0)      # ensure out is binary
        lout = lout + out[k] * e2
        e2 = e2 + e2
    c.enforce(lin, lout)                          # ensure the sum


@template
def T1(c):
    h = c.input("h", 32)
    e = c.input("e", 32)
    f = c.input("f", 32)
    g = c.input("g", 32)
    k = c.input("k", 32)
    w = c.input("w", 32)
    out = c.output("out", 32)
    ch = c.component("ch", Ch_t(32))
    bigsigma1 = c.component("bigsigma1", BigSigma(6, 11, 25))
    for ki in range(32):
        c.set(bigsigma1["in"][ki], e[ki])
        c.set(ch["a"][ki], e[ki])
        c.set(ch["b"][ki], f[ki])
        c.set(ch["c"][ki], g[ki])
    sum_ = c.component("sum", BinSum(32, 5))
    for ki in range(32):
        c.set(sum_["in"][0][ki], h[ki])
        c.set(sum_["in"][1][ki], bigsigma1["out"][ki])
        c.set(sum_["in"][2][ki], ch["out"][ki])
        c.set(sum_["in"][3][ki], k[ki])
        c.set(sum_["in"][4][ki], w[ki])
    for ki in range(32):
        c.set(out[ki], sum_["out"][ki])


@template
def T2(c):
    a = c.input("a", 32)
    b = c.input("b", 32)
    cc = c.input("c", 32)
    out = c.output("out", 32)
    bigsigma0 = c.component("bigsigma0", BigSigma(2, 13, 22))
    maj = c.component("maj", Maj_t(32))
    for k in range(32):
        c.set(bigsigma0["in"][k], a[k])
        c.set(maj["a"][k], a[k])
        c.set(maj["b"][k], b[k])
        c.set(maj["c"][k], cc[k])
    sum_ = c.component("sum", BinSum(32, 2))
    for k in range(32):
        c.set(sum_["in"][0][k], bigsigma0["out"][k])
        c.set(sum_["in"][1][k], maj["out"][k])
    for k in range(32):
        c.set(out[k], sum_["out"][k])


@template
def SigmaPlus(c):
    in2 = c.input("in2", 32)
    in7 = c.input("in7", 32)
    in15 = c.input("in15", 32)
    in16 = c.input("in16", 32)
    out = c.output("out", 32)
    sigma1 = c.component("sigma1", SmallSigma(17, 19, 10))
    sigma0 = c.component("sigma0", SmallSigma(7, 18, 3))
    for k in range(32):
        c.set(sigma1["in"][k], in2[k])
        c.set(sigma0["in"][k], in15[k])
    sum_ = c.component("sum", BinSum(32, 4))
    for k in range(32):
        c.set(sum_["in"][0][k], sigma1["out"][k])
        c.set(sum_["in"][1][k], in7[k])
        c.set(sum_["in"][2][k], sigma0["out"][k])
        c.set(sum_["in"][3][k], in16[k])
    for k in range(32):
        c.set(out[k], sum_["out"][k])


# ---- witness-side helper: circomlib's `sha256compression` *function* (native 32-bit word arithmetic on
# run-time values; only used to compute `out <-- ...`, then constrained against the adders) -------------
M32 = 0xFFFFFFFF


def _rrot(x, n):
    return ((x >> n) | (x << (32 - n))) & M32


def _bsigma0(x):
    return _rrot(x, 2) ^ _rrot(x, 13) ^ _rrot(x, 22)


def _bsigma1(x):
    return _rrot(x, 6) ^ _rrot(x, 11) ^ _rrot(x, 25)


def _ssigma0(x):
    return _rrot(x, 7) ^ _rrot(x, 18) ^ (x >> 3)


def _ssigma1(x):
    return _rrot(x, 17) ^ _rrot(x, 19) ^ (x >> 10)


def _maj(x, y, z):
    return (x & y) ^ (x & z) ^ (y & z)


def _ch(x, y, z):
    return (x & y) ^ ((M32 ^ x) & z)


def sha256compression_fn(c, hin, inp):
    Hs = []
    for i in range(8):
        acc = c.const(0)
        for j in range(32):
            acc = acc + (hin[i * 32 + j] << j)
        Hs.append(acc)
    a, b, cc, d, e, f, g, h = Hs
    w = [None] * 64
    for i in range(64):
        if i < 16:
            acc = c.const(0)
            for j in range(32):
                acc = acc + (inp[i * 32 + 31 - j] << j)
            w[i] = acc
        else:
            w[i] = (_ssigma1(w[i - 2]) + w[i - 7] + _ssigma0(w[i - 15]) + w[i - 16]) & M32
        t1 = (h + _bsigma1(e) + _ch(e, f, g) + K_TABLE[i] + w[i]) & M32
        t2 = (_bsigma0(a) + _maj(a, b, cc)) & M32
        h = g
        g = f
        f = e
        e = (d + t1) & M32
        d = cc
        cc = b
        b = a
        a = (t1 + t2) & M32
    Hs = [(x + y) & M32 for x, y in zip(Hs, (a, b, cc, d, e, f, g, h))]
    out = [None] * 256
    for i in range(8):
        for j in range(32):
            out[i * 32 + 31 - j] = (Hs[i] >> j) & 1
    return out


@template
def Sha256compression(c):
    hin = c.input("hin", 256)
    inp = c.input("inp", 512)
    out = c.output("out", 256)
    regs = [c.signal(nm, 65, 32) for nm in "abcdefgh"]
    a, b, cc, d, e, f, g, h = regs
    w = c.signal("w", 64, 32)

    outCalc = sha256compression_fn(c, hin, inp)
    for i in range(256):
        c.hint(out[i], outCalc[i])

    sigmaPlus = [c.component("sigmaPlus", SigmaPlus(), i) for i in range(48)]
    ct_k = [c.component("ct_k", K(i), i) for i in range(64)]
    t1 = [c.component("t1", T1(), i) for i in range(64)]
    t2 = [c.component("t2", T2(), i) for i in range(64)]
    suma = [c.component("suma", BinSum(32, 2), i) for i in range(64)]
    sume = [c.component("sume", BinSum(32, 2), i) for i in range(64)]
    fsum = [c.component("fsum", BinSum(32, 2), i) for i in range(8)]

    for t in range(64):
        if t < 16:
            for k in range(32):
                c.set(w[t][k], inp[t * 32 + 31 - k])
        else:
            sp = sigmaPlus[t - 16]
            for k in range(32):
                c.set(sp["in2"][k], w[t - 2][k])
                c.set(sp["in7"][k], w[t - 7][k])
                c.set(sp["in15"][k], w[t - 15][k])
                c.set(sp["in16"][k], w[t - 16][k])
            for k in range(32):
                c.set(w[t][k], sp["out"][k])

    for k in range(32):
        for r, reg in enumerate(regs):
            c.set(reg[0][k], hin[32 * r + k])

    for t in range(64):
        for k in range(32):
            c.set(t1[t]["h"][k], h[t][k])
            c.set(t1[t]["e"][k], e[t][k])
            c.set(t1[t]["f"][k], f[t][k])
            c.set(t1[t]["g"][k], g[t][k])
            c.set(t1[t]["k"][k], ct_k[t]["out"][k])
            c.set(t1[t]["w"][k], w[t][k])
            c.set(t2[t]["a"][k], a[t][k])
            c.set(t2[t]["b"][k], b[t][k])
            c.set(t2[t]["c"][k], cc[t][k])
        for k in range(32):
            c.set(sume[t]["in"][0][k], d[t][k])
            c.set(sume[t]["in"][1][k], t1[t]["out"][k])
            c.set(suma[t]["in"][0][k], t1[t]["out"][k])
            c.set(suma[t]["in"][1][k], t2[t]["out"][k])
        for k in range(32):
            c.set(h[t + 1][k], g[t][k])
            c.set(g[t + 1][k], f[t][k])
            c.set(f[t + 1][k], e[t][k])
            c.set(e[t + 1][k], sume[t]["out"][k])
            c.set(d[t + 1][k], cc[t][k])
            c.set(cc[t + 1][k], b[t][k])
            c.set(b[t + 1][k], a[t][k])
            c.set(a[t + 1][k], suma[t]["out"][k])

    for k in range(32):
        for r, reg in enumerate(regs):
            c.set(fsum[r]["in"][0][k], hin[32 * r + k])
            c.set(fsum[r]["in"][1][k], reg[64][k])
    for k in range(32):
        for r in range(8):
            c.enforce(out[32 * r + 31 - k], fsum[r]["out"][k])


@template
def Sha256(c, nBits):
    inp = c.input("in", nBits)
    out = c.output("out", 256)
    nBlocks = ((nBits + 64) // 512) + 1
    paddedIn = c.signal("paddedIn", nBlocks * 512)
    for k in range(nBits):
        c.set(paddedIn[k], inp[k])
    c.set(paddedIn[nBits], 1)
    for k in range(nBits + 1, nBlocks * 512 - 64):
        c.set(paddedIn[k], 0)
    for k in range(64):
        c.set(paddedIn[nBlocks * 512 - k - 1], (nBits >> k) & 1)

    hc = [c.component("h%s0" % "abcdefgh"[i], H(i)) for i in range(8)]
    comp = []
    for i in range(nBlocks):
        s = c.component("sha256compression", Sha256compression(), i)
        comp.append(s)
        if i == 0:
            for k in range(32):
                for r in range(8):
                    c.set(s["hin"][r * 32 + k], hc[r]["out"][k])
        else:
            for k in range(32):
                for r in range(8):
                    c.set(s["hin"][32 * r + k], comp[i - 1]["out"][32 * r + 31 - k])
        for k in range(512):
            c.set(s["inp"][k], paddedIn[i * 512 + k])
    for k in range(256):
        c.set(out[k], comp[nBlocks - 1]["out"][k])

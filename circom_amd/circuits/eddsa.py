"""EdDSA-Poseidon signature verification over BabyJubjub and a Semaphore-style membership circuit
(BASELINE config 4: "Poseidon Merkle depth 20 + EdDSA"), in the structure of circomlib's
`eddsaposeidon.circom` / `compconstant.circom` / `aliascheck.circom` / `comparators.circom`.

circomlib is absent from the reference tree and from this container, so the templates are re-authored from
the algorithms; the two scalar multiplications use the bit-serial `ScalarMulBits` of babyjub.py instead of
circomlib's windowed Montgomery-form ladders (same inputs, same outputs, more `<--` divisions: 4 per scalar
bit, which is what makes this circuit the stress case for the DIV/INV operators of the schedule).
Expected outputs are pinned against plain-integer arithmetic (eddsa_host.py) in tests/test_eddsa.py.
"""
from ..frontend.dsl import template
from .basic import IsZero, Num2Bits
from .babyjub import BabyAdd, BabyDbl, ScalarMulBits, ScalarMulBitsProj, BASE8
from .escalarmul import EscalarMulAny, EscalarMulFix
from .merkle import MerkleTreeInclusionProof
from .poseidon import Poseidon

SUBGROUP_ORDER = 2736030358979909402780800718157159386076813972158567259200215660948447373041


@template
def CompConstant(c, ct):
    """out = 1 iff the 254-bit number in[] (little endian) is greater than the constant ct."""
    inp = c.input("in", 254)
    out = c.output("out")
    parts = c.signal("parts", 127)
    sout = c.signal("sout")
    total = c.const(0)
    b = (1 << 128) - 1
    a = 1
    e = 1
    for i in range(127):
        clsb = (ct >> (2 * i)) & 1
        cmsb = (ct >> (2 * i + 1)) & 1
        slsb, smsb = inp[2 * i], inp[2 * i + 1]
        if cmsb == 0 and clsb == 0:
            c.set(parts[i], -b * smsb * slsb + b * smsb + b * slsb)
        elif cmsb == 0 and clsb == 1:
            c.set(parts[i], a * smsb * slsb - a * slsb + b * smsb - a * smsb + a)
        elif cmsb == 1 and clsb == 0:
            c.set(parts[i], b * smsb * slsb - a * smsb + a)
        else:
            c.set(parts[i], -a * smsb * slsb + a)
        total = total + parts[i]
        b -= e
        a += e
        e *= 2
    c.set(sout, total)
    n2b = c.component("num2bits", Num2Bits(135))
    c.set(n2b["in"], sout)
    c.set(out, n2b["out"][127])


@template
def AliasCheck(c):
    inp = c.input("in", 254)
    cc = c.component("compConstant", CompConstant(c.fp.q - 1))
    for i in range(254):
        c.set(cc["in"][i], inp[i])
    c.enforce(cc["out"], 0)


@template
def Num2BitsStrict(c):
    inp = c.input("in")
    out = c.output("out", 254)
    n2b = c.component("n2b", Num2Bits(254))
    c.set(n2b["in"], inp)
    ac = c.component("aliasCheck", AliasCheck())
    for i in range(254):
        c.set(out[i], n2b["out"][i])
        c.set(ac["in"][i], n2b["out"][i])


@template
def ForceEqualIfEnabled(c):
    enabled = c.input("enabled")
    inp = c.input("in", 2)
    isz = c.component("isz", IsZero())
    c.set(isz["in"], inp[1] - inp[0])
    c.enforce((1 - isz["out"]) * enabled, 0)


@template
def EdDSAPoseidonVerifier(c, proj=False):
    """proj = False: bit-serial affine ladders (babyjub.ScalarMulBits); True: the same relation with the witnesses walked on a
    projective ladder (ScalarMulBitsProj); "window": circomlib's structure - EscalarMulAny (Montgomery-form ladder in
    segments) for h * 8A and EscalarMulFix (3-bit windows) for S * B8 (circuits/escalarmul.py)"""
    window = proj == "window"
    Mul = ScalarMulBitsProj if proj else ScalarMulBits
    enabled = c.input("enabled")
    Ax = c.input("Ax"); Ay = c.input("Ay")
    S = c.input("S")
    R8x = c.input("R8x"); R8y = c.input("R8y")
    M = c.input("M")

    # S < subgroup order
    snum = c.component("snum2bits", Num2Bits(253))
    c.set(snum["in"], S)
    cc = c.component("compConstant", CompConstant(SUBGROUP_ORDER - 1))
    for i in range(253):
        c.set(cc["in"][i], snum["out"][i])
    c.set(cc["in"][253], 0)
    c.enforce(cc["out"] * enabled, 0)

    # h = H(R8, A, M)
    h = c.component("hash", Poseidon(5))
    for k, v in enumerate((R8x, R8y, Ax, Ay, M)):
        c.set(h["inputs"][k], v)
    h2b = c.component("h2bits", Num2BitsStrict())
    c.set(h2b["in"], h["out"])

    # right2 = h * 8 * A
    d1 = c.component("dbl1", BabyDbl()); c.set(d1["x"], Ax); c.set(d1["y"], Ay)
    d2 = c.component("dbl2", BabyDbl()); c.set(d2["x"], d1["xout"]); c.set(d2["y"], d1["yout"])
    d3 = c.component("dbl3", BabyDbl()); c.set(d3["x"], d2["xout"]); c.set(d3["y"], d2["yout"])
    az = c.component("isZero", IsZero())
    c.set(az["in"], d3["xout"])
    c.enforce(az["out"] * enabled, 0)                     # A is not in the small subgroup

    mul_any = c.component("mulAny", EscalarMulAny(254) if window else Mul(254))
    for i in range(254):
        c.set(mul_any["e"][i], h2b["out"][i])
    if window:
        c.set(mul_any["p"][0], d3["xout"]); c.set(mul_any["p"][1], d3["yout"])
        any_out = (mul_any["out"][0], mul_any["out"][1])
    else:
        c.set(mul_any["px"], d3["xout"]); c.set(mul_any["py"], d3["yout"])
        any_out = (mul_any["outx"], mul_any["outy"])

    # right = R8 + right2
    add1 = c.component("add1", BabyAdd())
    c.set(add1["x1"], R8x); c.set(add1["y1"], R8y)
    c.set(add1["x2"], any_out[0]); c.set(add1["y2"], any_out[1])

    # left = S * B8
    mul_fix = c.component("mulFix", EscalarMulFix(253, BASE8) if window else Mul(253))
    for i in range(253):
        c.set(mul_fix["e"][i], snum["out"][i])
    if not window:
        c.set(mul_fix["px"], BASE8[0]); c.set(mul_fix["py"], BASE8[1])
    fix_out = (mul_fix["out"][0], mul_fix["out"][1]) if window else (mul_fix["outx"], mul_fix["outy"])

    ex = c.component("eqCheckX", ForceEqualIfEnabled())
    c.set(ex["enabled"], enabled); c.set(ex["in"][0], fix_out[0]); c.set(ex["in"][1], add1["xout"])
    ey = c.component("eqCheckY", ForceEqualIfEnabled())
    c.set(ey["enabled"], enabled); c.set(ey["in"][0], fix_out[1]); c.set(ey["in"][1], add1["yout"])


@template
def SemaphoreStyle(c, nLevels, proj=False):
    """Membership + signed signal: the leaf Poseidon(Ax, Ay) is in the tree with the given root, the key signed
    the message (external nullifier), and nullifierHash = Poseidon(Ax, Ay, M) identifies the (key, message) pair."""
    Ax = c.input("Ax"); Ay = c.input("Ay")
    S = c.input("S")
    R8x = c.input("R8x"); R8y = c.input("R8y")
    M = c.input("externalNullifier")
    pathIndices = c.input("treePathIndices", nLevels)
    siblings = c.input("treeSiblings", nLevels)
    root = c.output("root")
    nullifier = c.output("nullifierHash")

    leaf = c.component("leaf", Poseidon(2))
    c.set(leaf["inputs"][0], Ax); c.set(leaf["inputs"][1], Ay)
    tree = c.component("tree", MerkleTreeInclusionProof(nLevels))
    c.set(tree["leaf"], leaf["out"])
    for i in range(nLevels):
        c.set(tree["pathIndices"][i], pathIndices[i])
        c.set(tree["siblings"][i], siblings[i])
    c.set(root, tree["root"])

    ver = c.component("verifier", EdDSAPoseidonVerifier(proj))
    c.set(ver["enabled"], 1)
    for name, v in (("Ax", Ax), ("Ay", Ay), ("S", S), ("R8x", R8x), ("R8y", R8y), ("M", M)):
        c.set(ver[name], v)

    nh = c.component("nullifier", Poseidon(3))
    c.set(nh["inputs"][0], Ax); c.set(nh["inputs"][1], Ay); c.set(nh["inputs"][2], M)
    c.set(nullifier, nh["out"])

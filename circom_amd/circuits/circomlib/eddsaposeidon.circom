/*
    EdDSA-Poseidon signature verification over BabyJubjub and a Semaphore-style membership circuit (BASELINE config 4:
    "Poseidon Merkle depth 20 + EdDSA"), in the shape of circomlib's eddsaposeidon.circom / compconstant.circom /
    aliascheck.circom; the two scalar multiplications are the bit-serial ladders of babyjub.circom (proj = 0: affine hints,
    1: projective hints, same constraints) or circomlib's own structure (proj = 2: EscalarMulAny in Montgomery form for
    h * 8A, windowed EscalarMulFix for S * B8).  The circom text of circom_amd/circuits/eddsa.py.
*/
pragma circom 2.0.0;

include "bitify.circom";
include "comparators.circom";
include "babyjub.circom";
include "escalarmulany.circom";
include "escalarmulfix.circom";
include "merkle.circom";
include "poseidon.circom";

// out = 1 iff the 254-bit number in[] (little endian) is greater than the constant ct
template CompConstant(ct) {
    signal input in[254];
    signal output out;
    signal parts[127];
    signal sout;
    var clsb;
    var cmsb;
    var slsb;
    var smsb;
    var sum = 0;
    var b = (1 << 128) - 1;
    var a = 1;
    var e = 1;
    var i;
    for (i = 0; i < 127; i++) {
        clsb = (ct >> (i * 2)) & 1;
        cmsb = (ct >> (i * 2 + 1)) & 1;
        slsb = in[i * 2];
        smsb = in[i * 2 + 1];
        if ((cmsb == 0) && (clsb == 0)) {
            parts[i] <== -b * smsb * slsb + b * smsb + b * slsb;
        } else if ((cmsb == 0) && (clsb == 1)) {
            parts[i] <== a * smsb * slsb - a * slsb + b * smsb - a * smsb + a;
        } else if ((cmsb == 1) && (clsb == 0)) {
            parts[i] <== b * smsb * slsb - a * smsb + a;
        } else {
            parts[i] <== -a * smsb * slsb + a;
        }
        sum = sum + parts[i];
        b = b - e;
        a = a + e;
        e = e * 2;
    }
    sout <== sum;
    component num2bits = Num2Bits(135);
    num2bits.in <== sout;
    out <== num2bits.out[127];
}

template AliasCheck() {
    signal input in[254];
    component compConstant = CompConstant(-1);
    for (var i = 0; i < 254; i++) {
        in[i] ==> compConstant.in[i];
    }
    compConstant.out === 0;
}

template Num2Bits_strict() {
    signal input in;
    signal output out[254];
    component n2b = Num2Bits(254);
    in ==> n2b.in;
    component aliasCheck = AliasCheck();
    for (var i = 0; i < 254; i++) {
        n2b.out[i] ==> out[i];
        n2b.out[i] ==> aliasCheck.in[i];
    }
}

template ForceEqualIfEnabled() {
    signal input enabled;
    signal input in[2];
    component isz = IsZero();
    in[1] - in[0] ==> isz.in;
    (1 - isz.out) * enabled === 0;
}

template EdDSAPoseidonVerifier(proj) {
    signal input enabled;
    signal input Ax;
    signal input Ay;
    signal input S;
    signal input R8x;
    signal input R8y;
    signal input M;
    var i;

    // S < subgroup order
    component snum2bits = Num2Bits(253);
    snum2bits.in <== S;
    component compConstant = CompConstant(2736030358979909402780800718157159386076813972158567259200215660948447373040);
    for (i = 0; i < 253; i++) {
        snum2bits.out[i] ==> compConstant.in[i];
    }
    compConstant.in[253] <== 0;
    compConstant.out * enabled === 0;

    // h = H(R8, A, M)
    component hash = Poseidon(5);
    hash.inputs[0] <== R8x;
    hash.inputs[1] <== R8y;
    hash.inputs[2] <== Ax;
    hash.inputs[3] <== Ay;
    hash.inputs[4] <== M;
    component h2bits = Num2Bits_strict();
    h2bits.in <== hash.out;

    // right2 = h * 8 * A
    component dbl1 = BabyDbl();
    dbl1.x <== Ax;
    dbl1.y <== Ay;
    component dbl2 = BabyDbl();
    dbl2.x <== dbl1.xout;
    dbl2.y <== dbl1.yout;
    component dbl3 = BabyDbl();
    dbl3.x <== dbl2.xout;
    dbl3.y <== dbl2.yout;
    component isZero = IsZero();
    isZero.in <== dbl3.xout;
    isZero.out * enabled === 0;               // A is not in the small subgroup

    component mulAny;
    if (proj == 2) {
        mulAny = EscalarMulAny(254);
    } else if (proj == 1) {
        mulAny = ScalarMulBitsProj(254);
    } else {
        mulAny = ScalarMulBits(254);
    }
    for (i = 0; i < 254; i++) {
        mulAny.e[i] <== h2bits.out[i];
    }
    var anyOut[2];
    if (proj == 2) {
        mulAny.p[0] <== dbl3.xout;
        mulAny.p[1] <== dbl3.yout;
        anyOut = [mulAny.out[0], mulAny.out[1]];
    } else {
        mulAny.px <== dbl3.xout;
        mulAny.py <== dbl3.yout;
        anyOut = [mulAny.outx, mulAny.outy];
    }

    // right = R8 + right2
    component add1 = BabyAdd();
    add1.x1 <== R8x;
    add1.y1 <== R8y;
    add1.x2 <== anyOut[0];
    add1.y2 <== anyOut[1];

    // left = S * B8
    var BASE8[2] = [
        5299619240641551281634865583518297030282874472190772894086521144482721001553,
        16950150798460657717958625567821834550301663161624707787222815936182638968203
    ];
    component mulFix;
    if (proj == 2) {
        mulFix = EscalarMulFix(253, BASE8);
    } else if (proj == 1) {
        mulFix = ScalarMulBitsProj(253);
    } else {
        mulFix = ScalarMulBits(253);
    }
    for (i = 0; i < 253; i++) {
        mulFix.e[i] <== snum2bits.out[i];
    }
    var fixOut[2];
    if (proj == 2) {
        fixOut = [mulFix.out[0], mulFix.out[1]];
    } else {
        mulFix.px <== BASE8[0];
        mulFix.py <== BASE8[1];
        fixOut = [mulFix.outx, mulFix.outy];
    }

    component eqCheckX = ForceEqualIfEnabled();
    eqCheckX.enabled <== enabled;
    eqCheckX.in[0] <== fixOut[0];
    eqCheckX.in[1] <== add1.xout;
    component eqCheckY = ForceEqualIfEnabled();
    eqCheckY.enabled <== enabled;
    eqCheckY.in[0] <== fixOut[1];
    eqCheckY.in[1] <== add1.yout;
}

// membership + signed signal: the leaf Poseidon(Ax, Ay) is in the tree with the given root, the key signed the message
// (external nullifier), and nullifierHash = Poseidon(Ax, Ay, M) identifies the (key, message) pair
template SemaphoreStyle(nLevels, proj) {
    signal input Ax;
    signal input Ay;
    signal input S;
    signal input R8x;
    signal input R8y;
    signal input externalNullifier;
    signal input treePathIndices[nLevels];
    signal input treeSiblings[nLevels];
    signal output root;
    signal output nullifierHash;

    component leaf = Poseidon(2);
    leaf.inputs[0] <== Ax;
    leaf.inputs[1] <== Ay;
    component tree = MerkleTreeInclusionProof(nLevels);
    tree.leaf <== leaf.out;
    for (var i = 0; i < nLevels; i++) {
        tree.pathIndices[i] <== treePathIndices[i];
        tree.siblings[i] <== treeSiblings[i];
    }
    root <== tree.root;

    component verifier = EdDSAPoseidonVerifier(proj);
    verifier.enabled <== 1;
    verifier.Ax <== Ax;
    verifier.Ay <== Ay;
    verifier.S <== S;
    verifier.R8x <== R8x;
    verifier.R8y <== R8y;
    verifier.M <== externalNullifier;

    component nullifier = Poseidon(3);
    nullifier.inputs[0] <== Ax;
    nullifier.inputs[1] <== Ay;
    nullifier.inputs[2] <== externalNullifier;
    nullifierHash <== nullifier.out;
}

/*
    Scalar multiplication of ANY BabyJubjub point, the way circomlib's escalarmulany.circom structures it: a bit-serial
    Montgomery-form ladder in segments of 148 bits; the chain carries p + sum_{i >= 1} e_i 2^i p (Montgomery addition is
    incomplete) and takes p away again when e_0 = 0 with one complete Edwards addition; the identity is handled by running
    on BASE8 and masking.  The circom text of circom_amd/circuits/escalarmul.py (second part).
*/
pragma circom 2.0.0;

include "montgomery.circom";
include "babyjub.circom";
include "comparators.circom";

template Multiplexor2() {
    signal input sel;
    signal input in[2][2];
    signal output out[2];
    out[0] <== (in[1][0] - in[0][0]) * sel + in[0][0];
    out[1] <== (in[1][1] - in[0][1]) * sel + in[0][1];
}

template BitElementMulAny() {
    signal input sel;
    signal input dblIn[2];
    signal input addIn[2];
    signal output dblOut[2];
    signal output addOut[2];
    component doubler = MontgomeryDouble();
    component adder = MontgomeryAdd();
    component selector = Multiplexor2();
    var k;
    sel ==> selector.sel;
    for (k = 0; k < 2; k++) {
        dblIn[k] ==> doubler.in[k];
    }
    for (k = 0; k < 2; k++) {
        doubler.out[k] ==> adder.in1[k];
        addIn[k] ==> adder.in2[k];
        addIn[k] ==> selector.in[0][k];
    }
    for (k = 0; k < 2; k++) {
        adder.out[k] ==> selector.in[1][k];
    }
    for (k = 0; k < 2; k++) {
        doubler.out[k] ==> dblOut[k];
        selector.out[k] ==> addOut[k];
    }
}

// out = (sum e_i 2^i) p for 2 <= n <= 148 bits; dbl = 2^(n-1) p in Montgomery form
template SegmentMulAny(n) {
    assert(n >= 2);
    assert(n <= 148);
    signal input e[n];
    signal input p[2];
    signal output out[2];
    signal output dbl[2];
    component e2m = Edwards2Montgomery();
    p[0] ==> e2m.in[0];
    p[1] ==> e2m.in[1];
    component bits[n - 1];
    var i;
    var k;
    for (i = 0; i < n - 1; i++) {
        bits[i] = BitElementMulAny();
        if (i == 0) {
            for (k = 0; k < 2; k++) {
                e2m.out[k] ==> bits[i].dblIn[k];
                e2m.out[k] ==> bits[i].addIn[k];
            }
        } else {
            for (k = 0; k < 2; k++) {
                bits[i - 1].dblOut[k] ==> bits[i].dblIn[k];
                bits[i - 1].addOut[k] ==> bits[i].addIn[k];
            }
        }
        e[i + 1] ==> bits[i].sel;
    }
    bits[n - 2].dblOut[0] ==> dbl[0];
    bits[n - 2].dblOut[1] ==> dbl[1];
    component m2e = Montgomery2Edwards();
    bits[n - 2].addOut[0] ==> m2e.in[0];
    bits[n - 2].addOut[1] ==> m2e.in[1];
    component eadder = BabyAdd();
    m2e.out[0] ==> eadder.x1;
    m2e.out[1] ==> eadder.y1;
    -p[0] ==> eadder.x2;
    p[1] ==> eadder.y2;
    component lastSel = Multiplexor2();
    e[0] ==> lastSel.sel;
    eadder.xout ==> lastSel.in[0][0];
    eadder.yout ==> lastSel.in[0][1];
    m2e.out[0] ==> lastSel.in[1][0];
    m2e.out[1] ==> lastSel.in[1][1];
    lastSel.out[0] ==> out[0];
    lastSel.out[1] ==> out[1];
}

template EscalarMulAny(n) {
    signal input e[n];
    signal input p[2];
    signal output out[2];
    var BASE8[2] = [
        5299619240641551281634865583518297030282874472190772894086521144482721001553,
        16950150798460657717958625567821834550301663161624707787222815936182638968203
    ];
    var nseg = (n - 1) \ 148 + 1;
    var nlast = n - (nseg - 1) * 148;
    component zeropoint = IsZero();
    zeropoint.in <== p[0];
    component segments[nseg];
    component doublers[nseg - 1];
    component m2e[nseg - 1];
    component adders[nseg - 1];
    var s;
    var i;
    var k;
    var ns;
    for (s = 0; s < nseg; s++) {
        ns = (s < nseg - 1) ? 148 : nlast;
        segments[s] = SegmentMulAny(ns);
        for (i = 0; i < ns; i++) {
            e[s * 148 + i] ==> segments[s].e[i];
        }
        if (s == 0) {
            segments[s].p[0] <== p[0] + (BASE8[0] - p[0]) * zeropoint.out;
            segments[s].p[1] <== p[1] + (BASE8[1] - p[1]) * zeropoint.out;
        } else {
            doublers[s - 1] = MontgomeryDouble();
            m2e[s - 1] = Montgomery2Edwards();
            adders[s - 1] = BabyAdd();
            for (k = 0; k < 2; k++) {
                segments[s - 1].dbl[k] ==> doublers[s - 1].in[k];
            }
            for (k = 0; k < 2; k++) {
                doublers[s - 1].out[k] ==> m2e[s - 1].in[k];
            }
            m2e[s - 1].out[0] ==> segments[s].p[0];
            m2e[s - 1].out[1] ==> segments[s].p[1];
            if (s == 1) {
                segments[0].out[0] ==> adders[s - 1].x1;
                segments[0].out[1] ==> adders[s - 1].y1;
            } else {
                adders[s - 2].xout ==> adders[s - 1].x1;
                adders[s - 2].yout ==> adders[s - 1].y1;
            }
            segments[s].out[0] ==> adders[s - 1].x2;
            segments[s].out[1] ==> adders[s - 1].y2;
        }
    }
    if (nseg == 1) {
        out[0] <== segments[0].out[0] * (1 - zeropoint.out);
        out[1] <== segments[0].out[1] + (1 - segments[0].out[1]) * zeropoint.out;
    } else {
        out[0] <== adders[nseg - 2].xout * (1 - zeropoint.out);
        out[1] <== adders[nseg - 2].yout + (1 - adders[nseg - 2].yout) * zeropoint.out;
    }
}

/*
    Scalar multiplication of a FIXED BabyJubjub point, the way circomlib's escalarmulfix.circom structures it: 3-bit windows
    behind an 8-entry table (1B..8B in Montgomery form, MultiMux3), every window adds (w_i + 1) 8^i B on top of an offset
    2 * 8^nW B; a second chain accumulates sum_i 8^i B + offset and ONE complete Edwards addition takes it away; segments of
    82 windows (246 bits).  The circom text of circom_amd/circuits/escalarmul.py (third part).
*/
pragma circom 2.0.0;

include "montgomery.circom";
include "babyjub.circom";
include "mux3.circom";

// out = (in + 1) * base for the 3-bit number `in`, out8 = 8 * base (Montgomery form)
template WindowMulFix() {
    signal input in[3];
    signal input base[2];
    signal output out[2];
    signal output out8[2];
    component mux = MultiMux3(2);
    for (var j = 0; j < 3; j++) {
        mux.s[j] <== in[j];
    }
    component dbl2 = MontgomeryDouble();
    dbl2.in[0] <== base[0];
    dbl2.in[1] <== base[1];
    mux.c[0][0] <== base[0];
    mux.c[1][0] <== base[1];
    mux.c[0][1] <== dbl2.out[0];
    mux.c[1][1] <== dbl2.out[1];
    // k * base = base + (k - 1) * base
    component adr3 = MontgomeryAdd();
    adr3.in1[0] <== base[0];
    adr3.in1[1] <== base[1];
    adr3.in2[0] <== dbl2.out[0];
    adr3.in2[1] <== dbl2.out[1];
    mux.c[0][2] <== adr3.out[0];
    mux.c[1][2] <== adr3.out[1];
    component adr4 = MontgomeryAdd();
    adr4.in1[0] <== base[0];
    adr4.in1[1] <== base[1];
    adr4.in2[0] <== adr3.out[0];
    adr4.in2[1] <== adr3.out[1];
    mux.c[0][3] <== adr4.out[0];
    mux.c[1][3] <== adr4.out[1];
    component adr5 = MontgomeryAdd();
    adr5.in1[0] <== base[0];
    adr5.in1[1] <== base[1];
    adr5.in2[0] <== adr4.out[0];
    adr5.in2[1] <== adr4.out[1];
    mux.c[0][4] <== adr5.out[0];
    mux.c[1][4] <== adr5.out[1];
    component adr6 = MontgomeryAdd();
    adr6.in1[0] <== base[0];
    adr6.in1[1] <== base[1];
    adr6.in2[0] <== adr5.out[0];
    adr6.in2[1] <== adr5.out[1];
    mux.c[0][5] <== adr6.out[0];
    mux.c[1][5] <== adr6.out[1];
    component adr7 = MontgomeryAdd();
    adr7.in1[0] <== base[0];
    adr7.in1[1] <== base[1];
    adr7.in2[0] <== adr6.out[0];
    adr7.in2[1] <== adr6.out[1];
    mux.c[0][6] <== adr7.out[0];
    mux.c[1][6] <== adr7.out[1];
    component adr8 = MontgomeryAdd();
    adr8.in1[0] <== base[0];
    adr8.in1[1] <== base[1];
    adr8.in2[0] <== adr7.out[0];
    adr8.in2[1] <== adr7.out[1];
    mux.c[0][7] <== adr8.out[0];
    mux.c[1][7] <== adr8.out[1];
    out8[0] <== adr8.out[0];
    out8[1] <== adr8.out[1];
    out[0] <== mux.out[0];
    out[1] <== mux.out[1];
}

// out = (sum e_i 2^i) base for 3 * nWindows bits (Edwards form), dbl = 8^nWindows * base (Montgomery form)
template SegmentMulFix(nWindows) {
    signal input e[nWindows * 3];
    signal input base[2];
    signal output out[2];
    signal output dbl[2];
    var i;
    var j;
    component e2m = Edwards2Montgomery();
    e2m.in[0] <== base[0];
    e2m.in[1] <== base[1];
    component windows[nWindows];
    for (i = 0; i < nWindows; i++) {
        windows[i] = WindowMulFix();
        if (i == 0) {
            windows[i].base[0] <== e2m.out[0];
            windows[i].base[1] <== e2m.out[1];
        } else {
            windows[i].base[0] <== windows[i - 1].out8[0];
            windows[i].base[1] <== windows[i - 1].out8[1];
        }
        for (j = 0; j < 3; j++) {
            windows[i].in[j] <== e[3 * i + j];
        }
    }
    // the offset both chains start from / end with: 2 * 8^nWindows * base
    component dblLast = MontgomeryDouble();
    dblLast.in[0] <== windows[nWindows - 1].out8[0];
    dblLast.in[1] <== windows[nWindows - 1].out8[1];
    // cadders: sum_i 8^i base + offset
    component cadders[nWindows];
    for (i = 0; i < nWindows; i++) {
        cadders[i] = MontgomeryAdd();
        if (i == 0) {
            cadders[i].in1[0] <== e2m.out[0];
            cadders[i].in1[1] <== e2m.out[1];
        } else {
            cadders[i].in1[0] <== cadders[i - 1].out[0];
            cadders[i].in1[1] <== cadders[i - 1].out[1];
        }
        if (i < nWindows - 1) {
            cadders[i].in2[0] <== windows[i].out8[0];
            cadders[i].in2[1] <== windows[i].out8[1];
        } else {
            cadders[i].in2[0] <== dblLast.out[0];
            cadders[i].in2[1] <== dblLast.out[1];
        }
    }
    component adders[nWindows];
    for (i = 0; i < nWindows; i++) {
        adders[i] = MontgomeryAdd();
        if (i == 0) {
            adders[i].in1[0] <== dblLast.out[0];
            adders[i].in1[1] <== dblLast.out[1];
        } else {
            adders[i].in1[0] <== adders[i - 1].out[0];
            adders[i].in1[1] <== adders[i - 1].out[1];
        }
        adders[i].in2[0] <== windows[i].out[0];
        adders[i].in2[1] <== windows[i].out[1];
    }
    component m2e = Montgomery2Edwards();
    component cm2e = Montgomery2Edwards();
    m2e.in[0] <== adders[nWindows - 1].out[0];
    m2e.in[1] <== adders[nWindows - 1].out[1];
    cm2e.in[0] <== cadders[nWindows - 1].out[0];
    cm2e.in[1] <== cadders[nWindows - 1].out[1];
    component cAdd = BabyAdd();
    cAdd.x1 <== m2e.out[0];
    cAdd.y1 <== m2e.out[1];
    cAdd.x2 <== -cm2e.out[0];
    cAdd.y2 <== cm2e.out[1];
    out[0] <== cAdd.xout;
    out[1] <== cAdd.yout;
    dbl[0] <== windows[nWindows - 1].out8[0];
    dbl[1] <== windows[nWindows - 1].out8[1];
}

// out = (sum e_i 2^i) * BASE for a compile-time point BASE of the prime-order subgroup
template EscalarMulFix(n, BASE) {
    signal input e[n];
    signal output out[2];
    var nseg = (n - 1) \ 246 + 1;
    var nlast = n - (nseg - 1) * 246;
    component segments[nseg];
    component m2e[nseg - 1];
    component adders[nseg - 1];
    var s;
    var i;
    var nbits;
    var nwin;
    for (s = 0; s < nseg; s++) {
        nbits = (s < nseg - 1) ? 246 : nlast;
        nwin = (nbits - 1) \ 3 + 1;
        segments[s] = SegmentMulFix(nwin);
        for (i = 0; i < nwin * 3; i++) {
            if (i < nbits) {
                segments[s].e[i] <== e[s * 246 + i];
            } else {
                segments[s].e[i] <== 0;
            }
        }
        if (s == 0) {
            segments[s].base[0] <== BASE[0];
            segments[s].base[1] <== BASE[1];
        } else {
            m2e[s - 1] = Montgomery2Edwards();
            adders[s - 1] = BabyAdd();
            m2e[s - 1].in[0] <== segments[s - 1].dbl[0];
            m2e[s - 1].in[1] <== segments[s - 1].dbl[1];
            segments[s].base[0] <== m2e[s - 1].out[0];
            segments[s].base[1] <== m2e[s - 1].out[1];
            if (s == 1) {
                adders[s - 1].x1 <== segments[0].out[0];
                adders[s - 1].y1 <== segments[0].out[1];
            } else {
                adders[s - 1].x1 <== adders[s - 2].xout;
                adders[s - 1].y1 <== adders[s - 2].yout;
            }
            adders[s - 1].x2 <== segments[s].out[0];
            adders[s - 1].y2 <== segments[s].out[1];
        }
    }
    if (nseg == 1) {
        out[0] <== segments[0].out[0];
        out[1] <== segments[0].out[1];
    } else {
        out[0] <== adders[nseg - 2].xout;
        out[1] <== adders[nseg - 2].yout;
    }
}

/*
    Bit-level pieces of SHA-256 on words of 32 bit signals (the shapes of circomlib's sha256/xor3.circom, rotate.circom,
    shift.circom, sigma.circom, ch.circom, maj.circom): degree-2 formulas per bit, rotations and shifts as wiring.
*/
pragma circom 2.0.0;

template Xor3(n) {
    signal input a[n];
    signal input b[n];
    signal input c[n];
    signal output out[n];
    signal mid[n];
    for (var k = 0; k < n; k++) {
        mid[k] <== b[k] * c[k];
        out[k] <== a[k] * (1 - 2 * b[k] - 2 * c[k] + 4 * mid[k]) + b[k] + c[k] - 2 * mid[k];
    }
}

template RotR(n, r) {
    signal input in[n];
    signal output out[n];
    for (var i = 0; i < n; i++) {
        out[i] <== in[(i + r) % n];
    }
}

template ShR(n, r) {
    signal input in[n];
    signal output out[n];
    for (var i = 0; i < n; i++) {
        if (i + r >= n) {
            out[i] <== 0;
        } else {
            out[i] <== in[i + r];
        }
    }
}

template SmallSigma(ra, rb, rc) {
    signal input in[32];
    signal output out[32];
    var k;
    component rota = RotR(32, ra);
    component rotb = RotR(32, rb);
    component shrc = ShR(32, rc);
    for (k = 0; k < 32; k++) {
        rota.in[k] <== in[k];
        rotb.in[k] <== in[k];
        shrc.in[k] <== in[k];
    }
    component xor3 = Xor3(32);
    for (k = 0; k < 32; k++) {
        xor3.a[k] <== rota.out[k];
        xor3.b[k] <== rotb.out[k];
        xor3.c[k] <== shrc.out[k];
    }
    for (k = 0; k < 32; k++) {
        out[k] <== xor3.out[k];
    }
}

template BigSigma(ra, rb, rc) {
    signal input in[32];
    signal output out[32];
    var k;
    component rota = RotR(32, ra);
    component rotb = RotR(32, rb);
    component rotc = RotR(32, rc);
    for (k = 0; k < 32; k++) {
        rota.in[k] <== in[k];
        rotb.in[k] <== in[k];
        rotc.in[k] <== in[k];
    }
    component xor3 = Xor3(32);
    for (k = 0; k < 32; k++) {
        xor3.a[k] <== rota.out[k];
        xor3.b[k] <== rotb.out[k];
        xor3.c[k] <== rotc.out[k];
    }
    for (k = 0; k < 32; k++) {
        out[k] <== xor3.out[k];
    }
}

template Ch_t(n) {
    signal input a[n];
    signal input b[n];
    signal input c[n];
    signal output out[n];
    for (var k = 0; k < n; k++) {
        out[k] <== a[k] * (b[k] - c[k]) + c[k];
    }
}

template Maj_t(n) {
    signal input a[n];
    signal input b[n];
    signal input c[n];
    signal output out[n];
    signal mid[n];
    for (var k = 0; k < n; k++) {
        mid[k] <== b[k] * c[k];
        out[k] <== a[k] * (b[k] + c[k] - 2 * mid[k]) + mid[k];
    }
}

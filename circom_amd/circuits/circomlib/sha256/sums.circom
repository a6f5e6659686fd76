/*
    The word additions of a SHA-256 round (the shapes of circomlib's sha256/t1.circom, t2.circom, sigmaplus.circom):
    T1 = h + Sigma1(e) + Ch(e, f, g) + k + w, T2 = Sigma0(a) + Maj(a, b, c), SigmaPlus = the message-schedule word.
*/
pragma circom 2.0.0;

include "../binsum.circom";
include "logic.circom";

template T1() {
    signal input h[32];
    signal input e[32];
    signal input f[32];
    signal input g[32];
    signal input k[32];
    signal input w[32];
    signal output out[32];
    var ki;
    component ch = Ch_t(32);
    component bigsigma1 = BigSigma(6, 11, 25);
    for (ki = 0; ki < 32; ki++) {
        bigsigma1.in[ki] <== e[ki];
        ch.a[ki] <== e[ki];
        ch.b[ki] <== f[ki];
        ch.c[ki] <== g[ki];
    }
    component sum = BinSum(32, 5);
    for (ki = 0; ki < 32; ki++) {
        sum.in[0][ki] <== h[ki];
        sum.in[1][ki] <== bigsigma1.out[ki];
        sum.in[2][ki] <== ch.out[ki];
        sum.in[3][ki] <== k[ki];
        sum.in[4][ki] <== w[ki];
    }
    for (ki = 0; ki < 32; ki++) {
        out[ki] <== sum.out[ki];
    }
}

template T2() {
    signal input a[32];
    signal input b[32];
    signal input c[32];
    signal output out[32];
    var k;
    component bigsigma0 = BigSigma(2, 13, 22);
    component maj = Maj_t(32);
    for (k = 0; k < 32; k++) {
        bigsigma0.in[k] <== a[k];
        maj.a[k] <== a[k];
        maj.b[k] <== b[k];
        maj.c[k] <== c[k];
    }
    component sum = BinSum(32, 2);
    for (k = 0; k < 32; k++) {
        sum.in[0][k] <== bigsigma0.out[k];
        sum.in[1][k] <== maj.out[k];
    }
    for (k = 0; k < 32; k++) {
        out[k] <== sum.out[k];
    }
}

template SigmaPlus() {
    signal input in2[32];
    signal input in7[32];
    signal input in15[32];
    signal input in16[32];
    signal output out[32];
    var k;
    component sigma1 = SmallSigma(17, 19, 10);
    component sigma0 = SmallSigma(7, 18, 3);
    for (k = 0; k < 32; k++) {
        sigma1.in[k] <== in2[k];
        sigma0.in[k] <== in15[k];
    }
    component sum = BinSum(32, 4);
    for (k = 0; k < 32; k++) {
        sum.in[0][k] <== sigma1.out[k];
        sum.in[1][k] <== in7[k];
        sum.in[2][k] <== sigma0.out[k];
        sum.in[3][k] <== in16[k];
    }
    for (k = 0; k < 32; k++) {
        out[k] <== sum.out[k];
    }
}

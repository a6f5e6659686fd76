/*
    The compression function on 32-bit words, as witness-side code (the shape of circomlib's
    sha256/sha256compression_function.circom): the component calls it with its input SIGNALS, so every operation below runs
    on run-time values; loop bounds and array indices are compile-time constants, so the call unrolls into the rows of
    the calling component.
*/
pragma circom 2.0.0;

include "constants.circom";

function rrot(x, n) {
    return ((x >> n) | (x << (32 - n))) & 0xFFFFFFFF;
}

function bsigma0(x) {
    return rrot(x, 2) ^ rrot(x, 13) ^ rrot(x, 22);
}

function bsigma1(x) {
    return rrot(x, 6) ^ rrot(x, 11) ^ rrot(x, 25);
}

function ssigma0(x) {
    return rrot(x, 7) ^ rrot(x, 18) ^ (x >> 3);
}

function ssigma1(x) {
    return rrot(x, 17) ^ rrot(x, 19) ^ (x >> 10);
}

function Maj(x, y, z) {
    return (x & y) ^ (x & z) ^ (y & z);
}

function Ch(x, y, z) {
    return (x & y) ^ ((0xFFFFFFFF ^ x) & z);
}

function sha256compression(hin, inp) {
    var H[8];
    var a;
    var b;
    var c;
    var d;
    var e;
    var f;
    var g;
    var h;
    var out[256];
    for (var i = 0; i < 8; i++) {
        H[i] = 0;
        for (var j = 0; j < 32; j++) {
            H[i] += hin[i * 32 + j] << j;
        }
    }
    a = H[0];
    b = H[1];
    c = H[2];
    d = H[3];
    e = H[4];
    f = H[5];
    g = H[6];
    h = H[7];
    var w[64];
    var T1;
    var T2;
    for (var i = 0; i < 64; i++) {
        if (i < 16) {
            w[i] = 0;
            for (var j = 0; j < 32; j++) {
                w[i] += inp[i * 32 + 31 - j] << j;
            }
        } else {
            w[i] = (ssigma1(w[i - 2]) + w[i - 7] + ssigma0(w[i - 15]) + w[i - 16]) & 0xFFFFFFFF;
        }
        T1 = (h + bsigma1(e) + Ch(e, f, g) + sha256K(i) + w[i]) & 0xFFFFFFFF;
        T2 = (bsigma0(a) + Maj(a, b, c)) & 0xFFFFFFFF;
        h = g;
        g = f;
        f = e;
        e = (d + T1) & 0xFFFFFFFF;
        d = c;
        c = b;
        b = a;
        a = (T1 + T2) & 0xFFFFFFFF;
    }
    H[0] = (H[0] + a) & 0xFFFFFFFF;
    H[1] = (H[1] + b) & 0xFFFFFFFF;
    H[2] = (H[2] + c) & 0xFFFFFFFF;
    H[3] = (H[3] + d) & 0xFFFFFFFF;
    H[4] = (H[4] + e) & 0xFFFFFFFF;
    H[5] = (H[5] + f) & 0xFFFFFFFF;
    H[6] = (H[6] + g) & 0xFFFFFFFF;
    H[7] = (H[7] + h) & 0xFFFFFFFF;
    for (var i = 0; i < 8; i++) {
        for (var j = 0; j < 32; j++) {
            out[i * 32 + 31 - j] = (H[i] >> j) & 1;
        }
    }
    return out;
}

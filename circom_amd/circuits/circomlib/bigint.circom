/*
    Non-native big-integer templates in the shape of 0xPARC circom-ecdsa's bigint.circom, compact constraint side (the
    whole numbers fit one field element: n * 2k <= 250): the circom text of circom_amd/circuits/bigint.py.
*/
pragma circom 2.0.0;

include "bitify.circom";
include "comparators.circom";
include "bigint_func.circom";

// a[2k] = div[k+1] * b[k] + mod[k], mod < b
template BigMod(n, k) {
    assert(n * 2 * k <= 250);
    signal input a[2 * k];
    signal input b[k];
    signal output div[k + 1];
    signal output mod[k];

    var longdiv[2][k + 1] = long_div(n, k, k, a, b);
    for (var i = 0; i < k + 1; i++) {
        div[i] <-- longdiv[0][i];
    }
    for (var i = 0; i < k; i++) {
        mod[i] <-- longdiv[1][i];
    }
    component div_range[k + 1];
    for (var i = 0; i < k + 1; i++) {
        div_range[i] = Num2Bits(n);
        div_range[i].in <== div[i];
    }
    component mod_range[k];
    for (var i = 0; i < k; i++) {
        mod_range[i] = Num2Bits(n);
        mod_range[i].in <== mod[i];
    }
    var A = a[0] + 0;
    for (var i = 1; i < 2 * k; i++) {
        A += a[i] * (1 << (n * i));
    }
    var B = b[0] + 0;
    for (var i = 1; i < k; i++) {
        B += b[i] * (1 << (n * i));
    }
    var D = div[0] + 0;
    for (var i = 1; i < k + 1; i++) {
        D += div[i] * (1 << (n * i));
    }
    var M = mod[0] + 0;
    for (var i = 1; i < k; i++) {
        M += mod[i] * (1 << (n * i));
    }
    D * B + M === A;
    component lt = LessThan(n * k);
    lt.in[0] <== M;
    lt.in[1] <== B;
    lt.out === 1;
}

// out = a * b mod p on k-limb numbers
template BigMultModP(n, k) {
    assert(n * 2 * k <= 250);
    signal input a[k];
    signal input b[k];
    signal input p[k];
    signal output out[k];
    var A = a[0] + 0;
    for (var i = 1; i < k; i++) {
        A += a[i] * (1 << (n * i));
    }
    var B = b[0] + 0;
    for (var i = 1; i < k; i++) {
        B += b[i] * (1 << (n * i));
    }
    signal prod;
    prod <== A * B;
    component prod_bits = Num2Bits(2 * n * k);
    prod_bits.in <== prod;
    component big_mod = BigMod(n, k);
    for (var i = 0; i < 2 * k; i++) {
        var limb = prod_bits.out[n * i] + 0;
        for (var j = 1; j < n; j++) {
            limb += prod_bits.out[n * i + j] * (1 << j);
        }
        big_mod.a[i] <== limb;
    }
    for (var i = 0; i < k; i++) {
        big_mod.b[i] <== p[i];
    }
    for (var i = 0; i < k; i++) {
        out[i] <== big_mod.mod[i];
    }
}

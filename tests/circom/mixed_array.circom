pragma circom 2.0.0;

// out[j][k] = in[j]^(k+1): a template whose output size depends on BOTH parameters
template PowerSums(n, m) {
    signal input in[n];
    signal output out[n][m];
    for (var j = 0; j < n; j++) {
        out[j][0] <== in[j] + 0;
        for (var k = 1; k < m; k++) {
            out[j][k] <== out[j][k - 1] * in[j];
        }
    }
}

// one template NAME, different parameters per array element: a Mixed cluster
// (compiler/src/intermediate_representation/translate.rs:1017-1045); the circom text of circuits/basic.py MixedArray
template MixedArray(nw, widths) {
    var n_in = 0;
    for (var i = 0; i < nw; i++) {
        n_in += widths[i][0];
    }
    signal input x[n_in];
    signal output s;
    component ps[nw];
    for (var i = 0; i < nw; i++) {
        ps[i] = PowerSums(widths[i][0], widths[i][1]);
    }
    var k0 = 0;
    for (var i = 0; i < nw; i++) {
        for (var j = 0; j < widths[i][0]; j++) {
            ps[i].in[j] <== x[k0 + j];
        }
        k0 += widths[i][0];
    }
    var acc = 0;
    for (var i = 0; i < nw; i++) {
        for (var j = 0; j < widths[i][0]; j++) {
            acc += ps[i].out[j][widths[i][1] - 1] * (i + 2);
        }
    }
    s <== acc;
}

component main = MixedArray(4, [[2, 3], [1, 5], [3, 2], [2, 3]]);

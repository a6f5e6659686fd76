/* Two-way multiplexer (the shape of circomlib's mux1.circom Mux1). */
pragma circom 2.0.0;

template Mux1() {
    signal input c[2];
    signal input s;
    signal output out;
    out <== (c[1] - c[0]) * s + c[0];
}

template MultiMux1(n) {
    signal input c[n][2];
    signal input s;
    signal output out[n];
    for (var i = 0; i < n; i++) {
        out[i] <== (c[i][1] - c[i][0]) * s + c[i][0];
    }
}

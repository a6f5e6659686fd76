/* Eight-way multiplexer on n values at once (the shape of circomlib's mux3.circom MultiMux3). */
pragma circom 2.0.0;

template MultiMux3(n) {
    signal input c[n][8];
    signal input s[3];
    signal output out[n];
    signal s10;
    s10 <== s[1] * s[0];
    signal a210[n];
    signal a21[n];
    signal a20[n];
    signal a2[n];
    signal a10[n];
    signal a1[n];
    signal a0[n];
    signal a[n];
    for (var i = 0; i < n; i++) {
        a210[i] <== (c[i][7] - c[i][6] - c[i][5] + c[i][4] - c[i][3] + c[i][2] + c[i][1] - c[i][0]) * s10;
        a21[i] <== (c[i][6] - c[i][4] - c[i][2] + c[i][0]) * s[1];
        a20[i] <== (c[i][5] - c[i][4] - c[i][1] + c[i][0]) * s[0];
        a2[i] <== c[i][4] - c[i][0];
        a10[i] <== (c[i][3] - c[i][2] - c[i][1] + c[i][0]) * s10;
        a1[i] <== (c[i][2] - c[i][0]) * s[1];
        a0[i] <== (c[i][1] - c[i][0]) * s[0];
        a[i] <== c[i][0];
        out[i] <== (a210[i] + a21[i] + a20[i] + a2[i]) * s[2] + (a10[i] + a1[i] + a0[i] + a[i]);
    }
}

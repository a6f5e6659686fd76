/*
    BabyJubjub in Montgomery form, B v^2 = u^3 + A u^2 + u with A = 168698, B = 1 (the shape of circomlib's
    montgomery.circom): conversions, incomplete addition, doubling - one division hint + three constraints each.
    The circom text of circom_amd/circuits/escalarmul.py (first part).
*/
pragma circom 2.0.0;

template Edwards2Montgomery() {
    signal input in[2];
    signal output out[2];
    out[0] <-- (1 + in[1]) / (1 - in[1]);
    out[1] <-- out[0] / in[0];
    out[0] * (1 - in[1]) === (1 + in[1]);
    out[1] * in[0] === out[0];
}

template Montgomery2Edwards() {
    signal input in[2];
    signal output out[2];
    out[0] <-- in[0] / in[1];
    out[1] <-- (in[0] - 1) / (in[0] + 1);
    out[0] * in[1] === in[0];
    out[1] * (in[0] + 1) === in[0] - 1;
}

template MontgomeryAdd() {
    signal input in1[2];
    signal input in2[2];
    signal output out[2];
    var A = 168698;
    var B = 1;
    signal lamda;
    lamda <-- (in2[1] - in1[1]) / (in2[0] - in1[0]);
    lamda * (in2[0] - in1[0]) === (in2[1] - in1[1]);
    out[0] <== B * lamda * lamda - A - in1[0] - in2[0];
    out[1] <== lamda * (in1[0] - out[0]) - in1[1];
}

template MontgomeryDouble() {
    signal input in[2];
    signal output out[2];
    var A = 168698;
    var B = 1;
    signal lamda;
    signal x1_2;
    x1_2 <== in[0] * in[0];
    lamda <-- (3 * x1_2 + 2 * A * in[0] + 1) / (2 * B * in[1]);
    lamda * (2 * B * in[1]) === (3 * x1_2 + 2 * A * in[0] + 1);
    out[0] <== B * lamda * lamda - A - 2 * in[0];
    out[1] <== lamda * (in[0] - out[0]) - in[1];
}

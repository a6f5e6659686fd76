/* Boolean gates as degree-2 polynomials (the shape of circomlib's gates.circom). */
pragma circom 2.0.0;

template XOR() {
    signal input a;
    signal input b;
    signal output out;
    out <== a + b - 2 * a * b;
}

template AND() {
    signal input a;
    signal input b;
    signal output out;
    out <== a * b;
}

template OR() {
    signal input a;
    signal input b;
    signal output out;
    out <== a + b - a * b;
}

template NOT() {
    signal input in;
    signal output out;
    out <== 1 + in - 2 * in;
}

// the AND of n bits by halving: a template that instantiates ITSELF with smaller parameters (the shape of circomlib's MultiAND);
// ands[0] and ands[1] may differ in their parameter: a Mixed cluster
template MultiAND(n) {
    signal input in[n];
    signal output out;
    component and1;
    component and2;
    component ands[2];
    if (n == 1) {
        out <== in[0];
    } else if (n == 2) {
        and1 = AND();
        and1.a <== in[0];
        and1.b <== in[1];
        out <== and1.out;
    } else {
        and2 = AND();
        var n1 = n \ 2;
        var n2 = n - n \ 2;
        ands[0] = MultiAND(n1);
        ands[1] = MultiAND(n2);
        var i;
        for (i = 0; i < n1; i++) ands[0].in[i] <== in[i];
        for (i = 0; i < n2; i++) ands[1].in[i] <== in[n1 + i];
        and2.a <== ands[0].out;
        and2.b <== ands[1].out;
        out <== and2.out;
    }
}

pragma circom 2.0.0;

// every operator of the witness-code expression language on two run-time values
// (the circom text of circom_amd/circuits/opzoo.py; goldens: tests/golden/reference_wtns.json "opzoo")
template OperatorZoo() {
    signal input a;
    signal input b;
    signal output out[25];
    var nz = b == 0 ? 1 : b;       // \ and % by zero abort the reference (GMP division by zero)
    out[0] <-- a + b;
    out[1] <-- a - b;
    out[2] <-- a * b;
    out[3] <-- a / b;
    out[4] <-- a \ nz;
    out[5] <-- a % nz;
    out[6] <-- a ** (b & 255);
    out[7] <-- a << b;
    out[8] <-- a >> b;
    out[9] <-- a & b;
    out[10] <-- a | b;
    out[11] <-- a ^ b;
    out[12] <-- a < b;
    out[13] <-- a > b;
    out[14] <-- a <= b;
    out[15] <-- a >= b;
    out[16] <-- a == b;
    out[17] <-- a != b;
    out[18] <-- a && b;
    out[19] <-- a || b;
    out[20] <-- -a;
    out[21] <-- ~a;
    out[22] <-- !a;
    out[23] <-- a < b ? a * 3 : b - 1;
    out[24] <-- ((a >> 3) & 0xFFFF) * (b % 1000 + 1) + (a \ 7 % 11) - (~b & 15);
}

component main = OperatorZoo();

pragma circom 2.0.0;

include "comparators.circom";
include "switcher.circom";
include "binsum.circom";
include "bitify.circom";

// order two n-bit numbers, report equality and the binary sum of their bit decompositions
// (the circom text of circom_amd/circuits/stdlib.py SortPair)
template SortPair(n) {
    signal input in[2];
    signal output lo;
    signal output hi;
    signal output eq;
    signal output sum;
    component gt = GreaterThan(n);
    gt.in[0] <== in[0];
    gt.in[1] <== in[1];
    component sw = Switcher();
    sw.sel <== gt.out;
    sw.L <== in[0];
    sw.R <== in[1];
    lo <== sw.outL;
    hi <== sw.outR;
    component ie = IsEqual();
    ie.in[0] <== in[0];
    ie.in[1] <== in[1];
    eq <== ie.out;
    component ba = Num2Bits(n);
    ba.in <== in[0];
    component bb = Num2Bits(n);
    bb.in <== in[1];
    component bs = BinSum(n, 2);
    for (var k = 0; k < n; k++) {
        bs.in[0][k] <== ba.out[k];
        bs.in[1][k] <== bb.out[k];
    }
    component b2n = Bits2Num(n + 1);
    for (var k = 0; k < n + 1; k++) {
        b2n.in[k] <== bs.out[k];
    }
    sum <== b2n.out;
}

component main = SortPair(16);

pragma circom 2.0.0;
include "bitify.circom";
component main = Num2Bits(16);

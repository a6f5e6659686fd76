pragma circom 2.0.0;
include "gates.circom";
component main = MultiAND(5);

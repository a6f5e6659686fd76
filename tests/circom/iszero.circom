pragma circom 2.0.0;
include "comparators.circom";
component main = IsZero();

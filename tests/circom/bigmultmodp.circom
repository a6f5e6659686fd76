pragma circom 2.0.0;
include "bigint.circom";
component main = BigMultModP(32, 3);

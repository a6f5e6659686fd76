pragma circom 2.0.0;
include "bigint_ecdsa.circom";
component main = BigMultModPStyle(28, 3);

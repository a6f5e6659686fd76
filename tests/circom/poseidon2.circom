pragma circom 2.0.0;
include "poseidon.circom";
component main = Poseidon(2);

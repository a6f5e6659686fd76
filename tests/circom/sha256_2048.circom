pragma circom 2.0.0;
include "sha256/sha256.circom";
component main = Sha256(2048);

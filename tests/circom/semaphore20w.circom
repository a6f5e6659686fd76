pragma circom 2.0.0;
include "eddsaposeidon.circom";
component main = SemaphoreStyle(20, 2);

pragma circom 2.0.0;
include "bigint_func.circom";

// the hint side of circom-ecdsa's BigModInv: out = in^-1 mod p by Fermat (a run-time loop over the bits of p - 2)
template ModInv(n, k) {
    signal input a[k];
    signal input p[k];
    signal output out[k];
    var r[k] = mod_inv(n, k, a, p);
    for (var i = 0; i < k; i++) {
        out[i] <-- r[i];
    }
}

component main = ModInv(16, 2);

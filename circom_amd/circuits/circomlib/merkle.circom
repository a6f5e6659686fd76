/*
    Poseidon Merkle-tree inclusion proof (the core of Semaphore / Tornado-style circuits): MultiMux1 orders (node, sibling)
    by a path-index bit, Poseidon(2) hashes the pair, level by level.  The circom text of circom_amd/circuits/merkle.py.
*/
pragma circom 2.0.0;

include "mux1.circom";
include "poseidon.circom";

template MerkleTreeInclusionProof(nLevels) {
    signal input leaf;
    signal input pathIndices[nLevels];
    signal input siblings[nLevels];
    signal output root;
    signal hashes[nLevels + 1];
    component mux[nLevels];
    component poseidons[nLevels];
    hashes[0] <== leaf;
    for (var i = 0; i < nLevels; i++) {
        pathIndices[i] * (1 - pathIndices[i]) === 0;
        mux[i] = MultiMux1(2);
        mux[i].c[0][0] <== hashes[i];
        mux[i].c[0][1] <== siblings[i];
        mux[i].c[1][0] <== siblings[i];
        mux[i].c[1][1] <== hashes[i];
        mux[i].s <== pathIndices[i];
        poseidons[i] = Poseidon(2);
        poseidons[i].inputs[0] <== mux[i].out[0];
        poseidons[i].inputs[1] <== mux[i].out[1];
        hashes[i + 1] <== poseidons[i].out;
    }
    root <== hashes[nLevels];
}

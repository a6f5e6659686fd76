pragma circom 2.0.0;

// a sub-component that logs while it runs: its line appears where the component FIRES (after its last input)
template LogSquare() {
    signal input in;
    signal output out;
    out <== in * in;
    log("square of", in, "is", out);
}

// log(...) in the shapes LogBucket knows (log_bucket.rs:105-162); the circom text of circuits/basic.py LogDemo
template LogDemo() {
    signal input a;
    signal input b;
    signal output out;
    log("inputs:", a, b);
    component sq = LogSquare();
    sq.in <== a + b;
    log(a * b + 7);
    log();
    log("constant", 42);
    assert(a != 13);
    out <== sq.out + a;
    log("out =", out, "(after the check)");
    log("100%% of", 2, "checks passed");
}

component main = LogDemo();

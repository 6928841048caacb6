pragma circom 2.0.0;

// Written for this repo's fixtures (NOT a reference file): the reference's Mux3 / MultiMux3 templates
// (tests/circuits/machine-learning/circomlib/mux3.circom, resolved at fixture-generation time) fed by EXPRESSIONS: the table
// entries are affine in x, the selector bits come from comparisons — all evaluated after the multiplexers' bodies have been
// appended (SURVEY D.3), so the DFS order of topological_sort.rs is not the list order.
include "circomlib/mux3.circom";

template TwoMux() {
    signal input x;
    signal input y;
    signal input t[3];
    signal output out;
    signal output other;

    component m = Mux3();
    component w = MultiMux3(2);
    for (var i = 0; i < 8; i++) {
        m.c[i] <== x * i + y;
        w.c[0][i] <== y * i + x;
        w.c[1][i] <== x + i;
    }
    for (var j = 0; j < 3; j++) {
        m.s[j] <== t[j] > 10;
        w.s[j] <== t[j] < 7;
    }
    out <== m.out + w.out[0];
    other <== w.out[1] * m.out;
}

component main = TwoMux();

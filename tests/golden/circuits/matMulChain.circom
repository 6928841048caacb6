pragma circom 2.0.0;

// Written for this repo's fixtures (NOT a reference file): two of the reference's matMul templates
// (tests/circuits/machine-learning/circomlib-matrix/matMul.circom and what it includes, resolved at fixture-generation time)
// in a chain, (A + B) x B and then that x (A - B), with the operand matrices computed by expressions that are evaluated
// after both multipliers' bodies have been appended (SURVEY D.3): a 100+-gate circuit whose DFS order is not the list order.
include "circomlib-matrix/matMul.circom";

template Chain(n) {
    signal input a[n][n];
    signal input b[n][n];
    signal output out[n][n];
    signal output tr;

    component p = matMul(n, n, n);
    component q = matMul(n, n, n);
    for (var i = 0; i < n; i++) {
        for (var j = 0; j < n; j++) {
            q.a[i][j] <== p.out[i][j];
            q.b[i][j] <== a[i][j] - b[i][j];
        }
    }
    for (var i = 0; i < n; i++) {
        for (var j = 0; j < n; j++) {
            p.a[i][j] <== a[i][j] + b[i][j];
            p.b[i][j] <== b[i][j];
        }
    }
    for (var i = 0; i < n; i++) {
        for (var j = 0; j < n; j++) {
            out[i][j] <== q.out[i][j];
        }
    }
    tr <== q.out[0][0] + q.out[n - 1][n - 1];
}

component main = Chain(3);

pragma circom 2.1.0;

// Written for this repo's fixtures (not a reference file): the component's body is instantiated — its gate appended —
// BEFORE the expression that feeds its input is evaluated, so the gate list is NOT in dependency order and the
// reference's DFS (topological_sort.rs) returns a non-identity permutation (SURVEY.md D.3).
template Square() {
    signal input a;
    signal output b;
    b <== a * a;
}

template Chain() {
    signal input x;
    signal input z;
    signal output y;
    signal output w;

    component s = Square();
    component t = Square();
    t.a <== s.b - z;
    s.a <== x + 1;
    y <== t.b;
    w <== s.b * z;
}

component main = Chain();

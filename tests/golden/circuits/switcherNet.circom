pragma circom 2.0.0;

// Written for this repo's fixtures (NOT a reference file): a 4-input compare-exchange network over the reference's own
// Switcher template (tests/circuits/machine-learning/circomlib/switcher.circom, resolved at fixture-generation time).  Every
// Switcher is instantiated — its three gates appended — BEFORE the comparison that drives its selector and the wiring of its
// inputs are evaluated (process.rs:218-236, :367; SURVEY D.3), so the flat gate list is not in dependency order and
// topological_sort.rs returns a non-identity order.
include "circomlib/switcher.circom";

template SortFour() {
    signal input in[4];
    signal output out[4];
    signal output lo;

    component sw[5];
    for (var k = 0; k < 5; k++) {
        sw[k] = Switcher();
    }
    // layer 1: (0,1) (2,3)
    sw[0].sel <== in[0] > in[1];
    sw[0].L <== in[0];
    sw[0].R <== in[1];
    sw[1].sel <== in[2] > in[3];
    sw[1].L <== in[2];
    sw[1].R <== in[3];
    // layer 2: (0,2) (1,3)
    sw[2].sel <== sw[0].outL > sw[1].outL;
    sw[2].L <== sw[0].outL;
    sw[2].R <== sw[1].outL;
    sw[3].sel <== sw[0].outR > sw[1].outR;
    sw[3].L <== sw[0].outR;
    sw[3].R <== sw[1].outR;
    // layer 3: (1,2)
    sw[4].sel <== sw[2].outR > sw[3].outL;
    sw[4].L <== sw[2].outR;
    sw[4].R <== sw[3].outL;

    out[0] <== sw[2].outL;
    out[1] <== sw[4].outL;
    out[2] <== sw[4].outR;
    out[3] <== sw[3].outR;
    lo <== sw[2].outL + 0;
}

component main = SortFour();

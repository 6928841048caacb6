pragma circom 2.0.0;

// Written for this repo (NOT a reference file): a POSEIDON-SHAPED permutation over Z/2^32 — BASELINE.json configs[1] ("circomlib
// Poseidon(nInputs=2), ~300 gates") as a circuit the reference's front-end can take: the real Poseidon's round constants do not
// fit the reference's u32 literals (process.rs:300-303, SURVEY D.4), so the field is Z/2^32 (what --boolify-width 32 computes in),
// with Poseidon's structure: state of t = 3 words (capacity word 0, the two inputs), 8 full rounds (add round constants, x^5 on
// every word, MDS) with 4 partial rounds (x^5 on word 0 only) in the middle, x^5 as x2 = x x, x4 = x2 x2, x5 = x4 x — AMul chains,
// the 2 824-gate multiplier template of the bit-blast — and a circulant 3 x 3 matrix {{5,7,3},{3,5,7},{7,3,5}} as AMul by constants
// + AAdd.  Round constants: x -> 1664525 x + 1013904223 mod 2^32 from seed 20241008 (tests/test_hash_circuits.py recomputes them).
// 301 arithmetic gates.  NOT a hash — the S-box is no permutation of Z/2^32 —: a known answer (numpy) for a Poseidon-shaped circuit.

template Pow5() {
    signal input in;
    signal output out;
    signal x2;
    signal x4;
    x2 <== in * in;
    x4 <== x2 * x2;
    out <== x4 * in;
}

template Mix() {
    signal input in[3];
    signal output out[3];
    out[0] <== 5 * in[0] + 7 * in[1] + 3 * in[2];
    out[1] <== 3 * in[0] + 5 * in[1] + 7 * in[2];
    out[2] <== 7 * in[0] + 3 * in[1] + 5 * in[2];
}

template PoseidonLike() {
    signal input in[2];
    signal output out;
    var c[36];
    c[0] = 2954275599;
    c[1] = 3929316642;
    c[2] = 1879603737;
    c[3] = 972299428;
    c[4] = 1027719091;
    c[5] = 1134690678;
    c[6] = 1556351581;
    c[7] = 295274520;
    c[8] = 2546756759;
    c[9] = 4292059402;
    c[10] = 1179786465;
    c[11] = 1477775564;
    c[12] = 3189642683;
    c[13] = 4292972510;
    c[14] = 652457381;
    c[15] = 2910579392;
    c[16] = 1476550431;
    c[17] = 4034601458;
    c[18] = 3537377449;
    c[19] = 1941735924;
    c[20] = 3033345219;
    c[21] = 2310728518;
    c[22] = 2927675885;
    c[23] = 563257960;
    c[24] = 968794791;
    c[25] = 37404634;
    c[26] = 2116390257;
    c[27] = 1790650396;
    c[28] = 614933707;
    c[29] = 1231532974;
    c[30] = 3561514805;
    c[31] = 4050145040;
    c[32] = 3335214895;
    c[33] = 3709213378;
    c[34] = 3794377529;
    c[35] = 1962249028;
    signal st[13][3];
    signal ark[12][3];
    signal sb[12][3];
    component s5[28];
    component mix[12];
    // (the components first, their inputs afterwards: the gate list is NOT in dependency order)
    for (var i = 0; i < 28; i++) {
        s5[i] = Pow5();
    }
    for (var r = 0; r < 12; r++) {
        mix[r] = Mix();
    }
    st[0][0] <== in[0] ^ in[0];
    st[0][1] <== in[0];
    st[0][2] <== in[1];
    var k = 0;
    for (var r = 0; r < 12; r++) {
        for (var j = 0; j < 3; j++) {
            ark[r][j] <== st[r][j] + c[3 * r + j];
        }
        if (r < 4 || r >= 8) {
            for (var j = 0; j < 3; j++) {
                s5[k].in <== ark[r][j];
                sb[r][j] <== s5[k].out;
                k++;
            }
        } else {
            s5[k].in <== ark[r][0];
            sb[r][0] <== s5[k].out;
            k++;
            sb[r][1] <== ark[r][1];
            sb[r][2] <== ark[r][2];
        }
        for (var j = 0; j < 3; j++) {
            mix[r].in[j] <== sb[r][j];
        }
        for (var j = 0; j < 3; j++) {
            st[r + 1][j] <== mix[r].out[j];
        }
    }
    out <== st[12][0];
}

component main = PoseidonLike();

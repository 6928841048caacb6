pragma circom 2.0.0;

// Written for this repo (NOT a reference file; nothing like it ships with the reference): the Keccak-f[1600] permutation, lane
// level — the 25 lanes of 64 bits as the signals of an arithmetic circuit that is meant to be compiled with --boolify-width 64
// (BASELINE.json configs[3]: "Keccak-256 full sponge, --boolify-width 64"; a bit-level circomlib-style original is outside what
// the reference's front-end supports, SURVEY D.4).  Operators of the reference's subset only (README.md:14-40): ^ & | >> <<.
// The reference's variables and literals are u32 (process.rs:558-579: `value: u32`), so what is 64 bits wide comes in as
// INPUTS: the 24 round constants rc[] and the all-ones word `ones` (~x = x ^ ones).  Lane (x, y) is in[x + 5 y].
// out = Keccak-f[1600](in): with in = a padded message block (rate 1088 bits, pad10*1 with the 0x06 domain bits of SHA-3)
// out[0..4] is the SHA3-256 digest of a message of at most 135 bytes — an external known answer (hashlib.sha3_256).

template Rot(r) {
    signal input in;
    signal output out;
    out <== (in << r) | (in >> (64 - r));
}

template Round() {
    signal input in[25];
    signal input rc;
    signal input ones;
    signal output out[25];

    // theta
    signal c[5];
    signal d[5];
    signal t[25];
    component r1[5];
    for (var x = 0; x < 5; x++) {
        r1[x] = Rot(1);
    }
    for (var x = 0; x < 5; x++) {
        c[x] <== in[x] ^ in[x + 5] ^ in[x + 10] ^ in[x + 15] ^ in[x + 20];
    }
    for (var x = 0; x < 5; x++) {
        r1[x].in <== c[(x + 1) % 5];
        d[x] <== c[(x + 4) % 5] ^ r1[x].out;
    }
    for (var i = 0; i < 25; i++) {
        t[i] <== in[i] ^ d[i % 5];
    }

    // rho and pi
    signal b[25];
    component rr[24];
    b[0] <== t[0];
    var x = 1;
    var y = 0;
    for (var k = 0; k < 24; k++) {
        rr[k] = Rot(((k + 1) * (k + 2) / 2) % 64);
        rr[k].in <== t[x + 5 * y];
        var ny = (2 * x + 3 * y) % 5;
        b[y + 5 * ny] <== rr[k].out;
        x = y;
        y = ny;
    }

    // chi and iota
    signal e[25];
    for (var j = 0; j < 5; j++) {
        for (var i = 0; i < 5; i++) {
            e[i + 5 * j] <== b[i + 5 * j] ^ ((b[(i + 1) % 5 + 5 * j] ^ ones) & b[(i + 2) % 5 + 5 * j]);
        }
    }
    out[0] <== e[0] ^ rc;
    for (var i = 1; i < 25; i++) {
        out[i] <== e[i];
    }
}

template KeccakF1600() {
    signal input in[25];
    signal input rc[24];
    signal input ones;
    signal output out[25];
    component rnd[24];
    // (the rounds first, their inputs afterwards: a component's body is appended where it is instantiated, so the gate list
    // is NOT in dependency order — topological_sort.rs has work to do, SURVEY D.3)
    for (var r = 0; r < 24; r++) {
        rnd[r] = Round();
    }
    for (var r = 0; r < 24; r++) {
        rnd[r].rc <== rc[r];
        rnd[r].ones <== ones;
    }
    for (var i = 0; i < 25; i++) {
        rnd[0].in[i] <== in[i];
    }
    for (var r = 1; r < 24; r++) {
        for (var i = 0; i < 25; i++) {
            rnd[r].in[i] <== rnd[r - 1].out[i];
        }
    }
    for (var i = 0; i < 25; i++) {
        out[i] <== rnd[23].out[i];
    }
}

component main = KeccakF1600();

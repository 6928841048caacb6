pragma circom 2.0.0;

// Written for this repo (NOT a reference file; nothing like it ships with the reference): the SHA3-256 SPONGE over nBlocks rate
// blocks (1088 bits = 17 lanes each), lane level — keccakF1600.circom's permutation, absorbed block by block: BASELINE.json
// configs[3] ("Keccak-256 full sponge, --boolify-width 64, ~150K arith gates") at its stated size as a REAL circuit — nBlocks =
// 29 unrolls to ~148 000 arithmetic gates.  Operators of the reference's subset only (README.md:14-40): ^ & | >> <<.  The
// reference's variables and literals are u32 (process.rs:558-579), so what is 64 bits wide comes in as INPUTS: rc[24] and `ones`.
// in[17 b + i] = lane i (little-endian) of padded block b (pad10*1 with SHA-3's 0x06 domain bits); out[0..4] = the digest:
// hashlib.sha3_256 of a message whose padding fills exactly nBlocks blocks (3 808 .. 3 943 bytes for nBlocks = 29).

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

template Sha3_256(nBlocks) {
    signal input in[17 * nBlocks];
    signal input rc[24];
    signal input ones;
    signal output out[4];
    component f[nBlocks];
    // (the permutations first, their inputs afterwards: the gate list is NOT in dependency order)
    for (var b = 0; b < nBlocks; b++) {
        f[b] = KeccakF1600();
    }
    for (var b = 0; b < nBlocks; b++) {
        for (var r = 0; r < 24; r++) {
            f[b].rc[r] <== rc[r];
        }
        f[b].ones <== ones;
    }
    // block 0 is absorbed into the all-zero state: the rate lanes are the block, the capacity lanes zero (x ^ x)
    for (var i = 0; i < 17; i++) {
        f[0].in[i] <== in[i];
    }
    for (var i = 17; i < 25; i++) {
        f[0].in[i] <== ones ^ ones;
    }
    for (var b = 1; b < nBlocks; b++) {
        for (var i = 0; i < 17; i++) {
            f[b].in[i] <== f[b - 1].out[i] ^ in[17 * b + i];
        }
        for (var i = 17; i < 25; i++) {
            f[b].in[i] <== f[b - 1].out[i];
        }
    }
    for (var i = 0; i < 4; i++) {
        out[i] <== f[nBlocks - 1].out[i];
    }
}

component main = Sha3_256(29);

pragma circom 2.0.0;

// Written for this repo (NOT a reference file; nothing like it ships with the reference): SHA-256 over nBlocks 512-bit blocks, word
// level — the templates of sha256Block.circom with the hash state as a signal, chained: BASELINE.json configs[2] ("circomlib Sha256
// ..., --boolify-width 32, ~30K arith gates") at its stated size as a REAL circuit — nBlocks = 9 unrolls to ~31 000 arithmetic
// gates (the bit-level circomlib original is outside what the reference's front-end supports, SURVEY D.4).  Operators of the
// reference's subset only (README.md:14-40): + ^ & | >> << and the prefix ~.  in[16 b + i] = word i (big-endian) of padded block
// b; out[0..8] = the digest: for a message whose padding fills exactly nBlocks blocks (512 .. 567 bytes for nBlocks = 9) that is
// hashlib.sha256 of it — an external known answer.

template RotR(r) {
    signal input in;
    signal output out;
    out <== (in >> r) | (in << (32 - r));
}

template SmallSigma(ra, rb, sh) {
    signal input in;
    signal output out;
    component a = RotR(ra);
    component b = RotR(rb);
    a.in <== in;
    b.in <== in;
    out <== a.out ^ b.out ^ (in >> sh);
}

template BigSigma(ra, rb, rc) {
    signal input in;
    signal output out;
    component a = RotR(ra);
    component b = RotR(rb);
    component c = RotR(rc);
    a.in <== in;
    b.in <== in;
    c.in <== in;
    out <== a.out ^ b.out ^ c.out;
}

template Ch() {
    signal input e;
    signal input f;
    signal input g;
    signal output out;
    out <== (e & f) ^ ((~e) & g);
}

template Maj() {
    signal input a;
    signal input b;
    signal input c;
    signal output out;
    out <== (a & b) ^ (a & c) ^ (b & c);
}

template Sha256Compress() {
    signal input hin[8];
    signal input in[16];
    signal output out[8];
    var k[64];
    k[0] = 1116352408;
    k[1] = 1899447441;
    k[2] = 3049323471;
    k[3] = 3921009573;
    k[4] = 961987163;
    k[5] = 1508970993;
    k[6] = 2453635748;
    k[7] = 2870763221;
    k[8] = 3624381080;
    k[9] = 310598401;
    k[10] = 607225278;
    k[11] = 1426881987;
    k[12] = 1925078388;
    k[13] = 2162078206;
    k[14] = 2614888103;
    k[15] = 3248222580;
    k[16] = 3835390401;
    k[17] = 4022224774;
    k[18] = 264347078;
    k[19] = 604807628;
    k[20] = 770255983;
    k[21] = 1249150122;
    k[22] = 1555081692;
    k[23] = 1996064986;
    k[24] = 2554220882;
    k[25] = 2821834349;
    k[26] = 2952996808;
    k[27] = 3210313671;
    k[28] = 3336571891;
    k[29] = 3584528711;
    k[30] = 113926993;
    k[31] = 338241895;
    k[32] = 666307205;
    k[33] = 773529912;
    k[34] = 1294757372;
    k[35] = 1396182291;
    k[36] = 1695183700;
    k[37] = 1986661051;
    k[38] = 2177026350;
    k[39] = 2456956037;
    k[40] = 2730485921;
    k[41] = 2820302411;
    k[42] = 3259730800;
    k[43] = 3345764771;
    k[44] = 3516065817;
    k[45] = 3600352804;
    k[46] = 4094571909;
    k[47] = 275423344;
    k[48] = 430227734;
    k[49] = 506948616;
    k[50] = 659060556;
    k[51] = 883997877;
    k[52] = 958139571;
    k[53] = 1322822218;
    k[54] = 1537002063;
    k[55] = 1747873779;
    k[56] = 1955562222;
    k[57] = 2024104815;
    k[58] = 2227730452;
    k[59] = 2361852424;
    k[60] = 2428436474;
    k[61] = 2756734187;
    k[62] = 3204031479;
    k[63] = 3329325298;

    // message schedule
    signal w[64];
    component s0[48];
    component s1[48];
    for (var i = 0; i < 16; i++) {
        w[i] <== in[i];
    }
    // (the components first, their inputs afterwards: a component's body is appended where it is instantiated, so the gate
    // list is NOT in dependency order — topological_sort.rs has work to do, SURVEY D.3)
    for (var i = 16; i < 64; i++) {
        s0[i - 16] = SmallSigma(7, 18, 3);
        s1[i - 16] = SmallSigma(17, 19, 10);
    }
    for (var i = 16; i < 64; i++) {
        s0[i - 16].in <== w[i - 15];
        s1[i - 16].in <== w[i - 2];
        w[i] <== s1[i - 16].out + w[i - 7] + s0[i - 16].out + w[i - 16];
    }

    // 64 rounds over the working variables st[round][a..h]
    signal st[65][8];
    signal t1[64];
    signal t2[64];
    component bs1[64];
    component bs0[64];
    component ch[64];
    component mj[64];
    for (var j = 0; j < 8; j++) {
        st[0][j] <== hin[j];
    }
    for (var i = 0; i < 64; i++) {
        bs1[i] = BigSigma(6, 11, 25);
        bs0[i] = BigSigma(2, 13, 22);
        ch[i] = Ch();
        mj[i] = Maj();
    }
    for (var i = 0; i < 64; i++) {
        bs1[i].in <== st[i][4];
        ch[i].e <== st[i][4];
        ch[i].f <== st[i][5];
        ch[i].g <== st[i][6];
        bs0[i].in <== st[i][0];
        mj[i].a <== st[i][0];
        mj[i].b <== st[i][1];
        mj[i].c <== st[i][2];
        t1[i] <== st[i][7] + bs1[i].out + ch[i].out + k[i] + w[i];
        t2[i] <== bs0[i].out + mj[i].out;
        st[i + 1][0] <== t1[i] + t2[i];
        st[i + 1][1] <== st[i][0];
        st[i + 1][2] <== st[i][1];
        st[i + 1][3] <== st[i][2];
        st[i + 1][4] <== st[i][3] + t1[i];
        st[i + 1][5] <== st[i][4];
        st[i + 1][6] <== st[i][5];
        st[i + 1][7] <== st[i][6];
    }
    for (var j = 0; j < 8; j++) {
        out[j] <== st[64][j] + hin[j];
    }
}

template Sha256(nBlocks) {
    signal input in[16 * nBlocks];
    signal output out[8];
    var h0[8];
    h0[0] = 1779033703;
    h0[1] = 3144134277;
    h0[2] = 1013904242;
    h0[3] = 2773480762;
    h0[4] = 1359893119;
    h0[5] = 2600822924;
    h0[6] = 528734635;
    h0[7] = 1541459225;


    component blk[nBlocks];
    // (the blocks first, their inputs afterwards — like the components inside a block: the gate list is NOT in dependency order)
    for (var b = 0; b < nBlocks; b++) {
        blk[b] = Sha256Compress();
    }
    for (var j = 0; j < 8; j++) {
        blk[0].hin[j] <== h0[j];
    }
    for (var b = 0; b < nBlocks; b++) {
        for (var i = 0; i < 16; i++) {
            blk[b].in[i] <== in[16 * b + i];
        }
    }
    for (var b = 1; b < nBlocks; b++) {
        for (var j = 0; j < 8; j++) {
            blk[b].hin[j] <== blk[b - 1].out[j];
        }
    }
    for (var j = 0; j < 8; j++) {
        out[j] <== blk[nBlocks - 1].out[j];
    }
}

component main = Sha256(9);

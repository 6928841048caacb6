// c2a_peel2.h — the DECOUPLED peel: the exact DFS tree of topological_sort (/root/reference/src/topological_sort.rs:3-50,
// deps of /root/reference/src/compiler.rs:408-421) as ONE launch with two kinds of waves.
//
// What c2a_peel.h does in one step of one wave — take the tickets of a gate's producers, learn which of them this gate
// completed, go on with one, hand the other over, AND run the gate's tournament and store its 512-byte record — is split:
//
//   CLAIM waves (a few per XCD): one LANE per chain, nothing but tickets.  A lane holds a claimed gate g and its two static
//   records; per turn it takes a ticket on each producer (4-byte returning atomics) while the static records of both
//   producers are already on their way; a producer whose last ticket this was is claimed.  The lane goes on with the first
//   claimed producer, a second one goes to the wave's LDS pool, where lanes without work pick gates up (a global queue
//   takes what a pool cannot hold, and feeds waves that have run dry).  One turn = one memory round trip (~0.7 us), and a
//   wave runs 64 chains side by side: the reverse Kahn front of a 2 000-wide graph fits a few dozen waves.
//   Every gate a lane takes on is written to the lane's MAILBOX, a ring of 16-byte entries {gate, edge label, "my previous
//   entry is one of this gate's consumers", where its consumer list is}.
//
//   TOURNAMENT waves, one per claim lane: each reads its mailbox in order and builds the node records exactly as
//   c2a_peel.h describes (path strings in self-validating words, one round trip per comparison).  Consecutive entries of
//   a chain are compared out of registers (the gate just finished is the first candidate of the next one); every other
//   candidate is read from memory and read again if it is not there yet.  No tickets, no hand-off, no termination
//   protocol on this side: a mailbox ends with a sentinel.
//
// The claim front does not wait for tournaments, so it runs at the speed of a ticket round trip instead of a whole step
// (wait + issue + tournament + stores, ~1.9 us in the one-wave launch); the tournaments trail it by a few entries and find
// their static data shipped with the entry.
//
// No deadlock.  Call tau(x) the moment gate x was claimed.  A mailbox is written in tau order and read in order; the
// consumers of x all have smaller tau.  The unfinished gate with the smallest tau therefore has all its candidates
// finished, and its tournament wave has nothing older in its mailbox: it is working on it.  Back-pressure (a claim lane
// waits while its ring is full) cannot block that gate either: a full ring only holds entries older than anything the
// waiting lane has in hand, and those are finished, i.e. consumed.  Tournament waves must be resident (the grid is clamped
// by the occupancy query): each owns one mailbox.
// Termination of the claim side: monotone counts, T = gates that came into being (B per shard + pushes P), F = gates dealt
// with (E per shard + pops C); per gate every T event precedes its F events (a gate handed over through the global queue:
// B when claimed, P then E when pushed, B then C when popped, E when its tickets are taken), so "F read first == T read afterwards" means equal at every moment in between (the argument of c2a_peel.h).
#pragma once
#include "c2a_peel.h"

namespace c2a {

// control block (u32 words; hot words on 128-byte lines of their own)
enum P2Ctl { P2_PROCESSED = 0, P2_MAXLEVEL = 1, P2_ABORT = CTL_ABORT /* 2: shared with the out-of-line helpers */, P2_REREADS = 3,
             P2_CLAIMED = 4, P2_PUSHED = 5, P2_POPPED = 6, P2_STALLS = 7, P2_TURNS = 8,
             P2_DONE = 32, P2_SEEDNEXT = 64, P2_HQPC = 96 /* u64: pushes (low) | pops (high) */,
             P2_ACCT = 128 /* kP2Shards u64 words, 32 u32 apart: B (low) | E (high) */, P2_WORDS = 128 + 8 * 32 };
constexpr u32 kP2Shards = 8;
constexpr u32 kPoolWords = 12;              // {gate | label << 31, consumer list offset, count, -, static record a (4), b (4)}
constexpr u32 kHqWords2 = 16;               // a global hand-off entry: 12 self-validating words (payload | run << 32) of one 128-byte line
constexpr u32 kMbSentinel = 0x7FFFFFFFu;
#ifdef C2A_EMULATE
constexpr u32 kP2LdsBlocks = 8;             // (the emulation keeps __shared__ in one static copy: one slice per claim block)
constexpr u32 kPoolCap = 16;                // (... and a pool that overflows into the global queue all the time)
constexpr u32 kPoolRoom = 8;                // (claim waves of at most 8 lanes there)
constexpr u32 kP2PollLimit = 1u << 16;
#else
constexpr u32 kP2LdsBlocks = 1;
constexpr u32 kPoolCap = 128;               // entries of a claim wave's LDS pool (a power of two)
constexpr u32 kPoolRoom = 64;               // room the pool must have at the top of a turn (every lane may add an entry); also the batch to and from the global queue
constexpr u32 kP2PollLimit = 1u << 21;
#endif

struct Peel2Args {
    u32 epoch;                 // tag (0/1) of this run's node words
    u32 n;
    const uint4* gstat;        // [2n] static records (c2a_peel.h)
    const u32* clist;          // consumer lists
    u64* node;                 // [n][64] node records
    u32* fill;                 // [n] tickets (zeroed per run)
    uint4* meta;               // [n]
    u32* child;                // [2n]
    u32 n_claim;               // blocks 0 .. n_claim-1 are claim waves, the rest tournament waves (one per claim lane)
    u32 n_lanes;               // lanes of a claim wave that take part (64; the emulation runs fewer)
    u64* mb;                   // [n_claim * n_lanes][mb_cap][2] mailboxes
    u32 mb_cap, mb_lg;         // entries per ring (a power of two), its log2
    const u64* mb_seq_in;      // [mailboxes] absolute number of each mailbox's next entry (never reset: the tag of an entry is its lap)
    u64* mb_seq_out;           // ... after this run
    u64* mb_rd;                // [mailboxes * 16] entries consumed, published by the tournament wave (a 128-byte line each)
    u64* hq;                   // [hq_cap][16] global hand-off entries
    u32 hq_cap, run;
    const u32* seeds;          // [n_regions][region_cap] producers claimed by the sinks pass
    const u32* seed_cnt;
    u32 n_regions, region_cap;
    u32* ctl;                  // [P2_WORDS]
    u32 dbg;                   // measurements only: bit 0 = tournament waves drain their mailboxes without working (times the claim side alone)
};

// ------------------------------------------------------------------------------------------------
// tournament side
// ------------------------------------------------------------------------------------------------
// An entry on its way through a tournament wave.  Three of these rotate: the entry in hand (its candidate records are on
// their way or there), the next one (its consumer list is on its way, then its records) and the one after (just decoded).
// Nothing in flight is ever copied (a register copy of a pending load is a use: its wait would drain everything issued
// since), so the three change ROLES, not places.
struct TourSlot {
    u32 st;                    // 0 empty; 1 decoded, consumer list issued; 2 candidate records issued; 3 the sentinel
    u32 g, label, cont, off, cnt;
    u32 blk;                   // lanes 0..63: the first 64 entries of the consumer list (consumer | edge label << 31)
    u32 e0, e1, e2, e3, k;     // the (up to four) candidates whose records are in flight ...
    u64 w0, w1, w2, w3;        // ... this lane's word of each
    u32 more;                  // the list holds candidates beyond those: the tournament reads the list itself (cold)
};

__device__ __forceinline__ void tour_role(const Peel2Args& A, u32 mail) {
    const u32 lane = threadIdx.x;
    const u32 epoch = A.epoch;
    const u64* ring = A.mb + (u64)mail * A.mb_cap * 2;
    u64 rseq = A.mb_seq_in[mail];      // entries finished
    u64 dseq = rseq;                   // entries decoded
    u32 processed = 0, max_level = 0, rereads = 0;
    // the gate just finished (a consumer of the next one when the entry says so): its record and what a candidate needs of it
    u32 own = C2A_NONE, own_root = 0, own_depth = 0, own_pos = 0, own_level = 0;
    u64 own_w = 0, own_str = 0;
    // the words of the next entry to decode (lanes 0, 1), loaded one step ahead
    u64 ew = 0;
    auto load_entry = [&]() {
        const u32 slot = (u32)dseq & (A.mb_cap - 1u);
        if (lane < 2) ew = ld_nw(&ring[(u64)slot * 2 + lane]);
        wave_join();
    };
    // decode the entry in ew into S if it is there; its consumer list is issued at once
    auto try_decode = [&](TourSlot& S) -> bool {
        const u32 tag = ((u32)(dseq >> A.mb_lg) + 1u) & 0xFFFFu;      // (lap + 1: a slot never written holds 0)
        const u64 wa = rdlane64(ew, 0), wb = rdlane64(ew, 1);
        if ((u32)(wa >> 48) != tag || (u32)(wb >> 48) != tag) return false;
        S.g = (u32)wa & kIdMask; S.label = ((u32)wa >> 31) & 1u; S.cont = (u32)(wa >> 32) & 1u;
        S.off = (u32)wb; S.cnt = (u32)(wb >> 32) & 0xFFFFu;
        ++dseq;
        if (S.g == kMbSentinel) { S.st = 3; return true; }
        if (S.cnt == 0xFFFFu) S.cnt = uniform(A.gstat[2 * (u64)S.g].w);      // (a list too long for the entry's 16 bits)
        S.blk = A.clist[S.off + lane];                                  // (clist is padded by 64 entries)
        S.st = 1;
        return true;
    };
    // pick the candidates of S (its consumer list is in) and issue their records; prev = the gate of the entry before S
    auto issue_records = [&](TourSlot& S, u32 prev) {
        u64 todo = __ballot(lane < S.cnt && !(S.cont && (S.blk & kIdMask) == prev && (S.blk >> 31) == S.label));
        S.k = 0; S.more = S.cnt > 64u ? 1u : 0u;
        if (todo) { S.e0 = rdlane(S.blk, ctz64(todo)); todo &= todo - 1; S.k = 1; S.w0 = ld_nw(&A.node[(u64)(S.e0 & kIdMask) * kNodeWords + lane]); }
        if (todo) { S.e1 = rdlane(S.blk, ctz64(todo)); todo &= todo - 1; S.k = 2; S.w1 = ld_nw(&A.node[(u64)(S.e1 & kIdMask) * kNodeWords + lane]); }
        if (todo) { S.e2 = rdlane(S.blk, ctz64(todo)); todo &= todo - 1; S.k = 3; S.w2 = ld_nw(&A.node[(u64)(S.e2 & kIdMask) * kNodeWords + lane]); }
        if (todo) { S.e3 = rdlane(S.blk, ctz64(todo)); todo &= todo - 1; S.k = 4; S.w3 = ld_nw(&A.node[(u64)(S.e3 & kIdMask) * kNodeWords + lane]); }
        if (todo) S.more = 1u;
        S.st = 2;
    };
    bool failed = false;
    ull tt_entry = 0, tt_settle = 0, tt_start = c2a_now();
    // one step: `cur` is worked on, `n1` and `n2` are brought forward.  true = the mailbox has ended (or the launch was given up)
    auto step = [&](TourSlot& cur, TourSlot& n1, TourSlot& n2) -> bool {
        // ---- the entry in hand must have its records on their way (a wave that has caught up with its claim lane does the
        // stages one after the other, waiting for each)
        bool decoded = false;
        const ull te0 = (A.dbg & 4u) ? c2a_now() : 0;
        if (cur.st == 0) {
            u32 polls = 0;
            for (;;) {
                if (try_decode(cur)) break;
                ++polls;
                if ((polls & 15u) == 0) {
                    u32 ab = 0;
                    if (lane == 0) ab = ld_a32(&A.ctl[P2_ABORT]); wave_join();
                    if (rdlane(ab, 0)) { failed = true; break; }
                    if (polls > kP2PollLimit) { if (lane == 0) atomicAdd(&A.ctl[P2_ABORT], 1u); wave_join(); failed = true; break; }
                }
                peel_sleep(polls < 8 ? 2 : (polls < 64 ? 8 : 32));
                load_entry();
            }
            if (failed) return true;
            load_entry();
            decoded = true;
        }
        if (A.dbg & 4u) tt_entry += c2a_now() - te0;
        if (cur.st == 3) return true;
        if (cur.st == 1) issue_records(cur, own);
        // ---- bring the next two forward: records of the next one (its list was issued a step ago), then ONE new entry (the
        // words of the entry after that are loaded now and looked at in the next step: looking at once would wait for them)
        if (n1.st == 1) issue_records(n1, cur.g);
        else if (n1.st == 0 && !decoded) { if (try_decode(n1)) { load_entry(); decoded = true; } }
        if (!decoded && n1.st != 0 && n1.st != 3 && n2.st == 0) { if (try_decode(n2)) load_entry(); }
        // ---- the tournament.  The champion so far (wave-uniform); ch == NONE: the virtual-root candidate [g]
        const u32 g = cur.g;
        if (A.dbg & 1u) { cur.st = 0; ++processed; ++rseq; if (((u32)rseq & 15u) == 0) { if (lane == 0) st_nw(&A.mb_rd[(u64)mail * 16], rseq); wave_join(); } return false; }
        u32 ch = C2A_NONE, ch_el = 0, ch_root = g, ch_depth = 0, ch_pos = 0;
        u64 ch_w = 0, ch_x = 0;
        u32 level = 0;
        const bool use_own = cur.cont && own != C2A_NONE;
        if (use_own) {
            level = own_level + 1u;
            if (own_root < g) {              // the gate just finished is the champion to beat: its string with the edge label appended
                ch = own; ch_el = cur.label; ch_root = own_root; ch_depth = own_depth; ch_pos = own_pos; ch_w = own_w;
                ch_x = own_str;
                if (lane == kHdrWords + (own_pos >> 8)) ch_x |= (u64)cur.label << (own_pos & 255u);
            }
        }
        bool gave_up = false;
        auto candidate = [&](u64 w, u32 e) {
            const u32 c = e & kIdMask, el = e >> 31;
            const u64 h0 = rdlane64(w, 0);
            const u32 croot = hdr_hi(h0), cdepth = (u32)h0;
            if (cdepth == 0u && lane >= kHdrWords) w = (u64)epoch << 63;          // (a depth-0 record has header words only)
            const u32 clevel = (rdlane((u32)(w >> 32), 1) & kIdMask) + 1u;
            const u32 cpos = rdlane((u32)w, 2);
            level = clevel > level ? clevel : level;
            u64 x = w & kPayload;
            if (lane == kHdrWords + (cpos >> 8)) x |= (u64)el << (cpos & 255u);
            u32 less;
            if (croot != ch_root) {
                less = croot < ch_root ? 1u : 0u;
            } else if ((cdepth > ch_depth ? cdepth : ch_depth) < kChunkBits) {
                const u64 d = x ^ ch_x;
                const u64 bal = __ballot(d != 0) & ~7ull;
                const u32 L = ctz64(bal);
                less = (u32)(~(rdlane64(x, L) >> ctz64(rdlane64(d, L)))) & 1u;
            } else if (c == ch) {
                less = el < ch_el ? 1u : 0u;
            } else {
                less = uniform(deep_less(A.node, epoch, A.ctl, c, el, cdepth, w, ch, ch_el, ch_depth, ch_w, lane) ? 1u : 0u);
            }
            if (less) { ch = c; ch_el = el; ch_root = croot; ch_depth = cdepth; ch_pos = cpos; ch_w = w; ch_x = x; }
        };
        // a record is all there when every word carries this run's tag — or, for a depth-0 record (a sink: header words
        // only), when its three header words do; what is not there yet is read again (its gate is being worked on right now)
        auto settle = [&](u64& w, u32 e) {
            u32 polls = 0;
            for (;;) {
                const u64 badm = __ballot((u32)(w >> 63) != epoch);
                if (badm == 0 || ((badm & 7ull) == 0 && (u32)rdlane64(w, 0) == 0u)) break;
                // not there yet: watch its FIRST 64 bytes only (eight lanes, one sector — polling whole records was 2 TB/s of
                // traffic), then read all of it again
                for (;;) {
                    if (++polls > kP2PollLimit) { gave_up = true; break; }
                    peel_sleep(polls < 4 ? 1 : (polls < 32 ? 4 : 16));
                    u64 h = 0;
                    if (lane < 8) h = ld_nw(&A.node[(u64)(e & kIdMask) * kNodeWords + lane]);
                    if (((u32)__ballot(lane < 8 && (u32)(h >> 63) == epoch) & 7u) == 7u) break;
                }
                if (gave_up) break;
                w = ld_nw(&A.node[(u64)(e & kIdMask) * kNodeWords + lane]);
            }
            rereads += polls;
        };
        const ull ts0 = (A.dbg & 4u) ? c2a_now() : 0;
        if ((A.dbg & 4u) && cur.k >= 1) { settle(cur.w0, cur.e0); if (cur.k >= 2 && !gave_up) settle(cur.w1, cur.e1); if (cur.k >= 3 && !gave_up) settle(cur.w2, cur.e2); if (cur.k >= 4 && !gave_up) settle(cur.w3, cur.e3); tt_settle += c2a_now() - ts0; }
        if (cur.k >= 1) { settle(cur.w0, cur.e0); if (!gave_up) candidate(cur.w0, cur.e0); }
        if (cur.k >= 2 && !gave_up) { settle(cur.w1, cur.e1); if (!gave_up) candidate(cur.w1, cur.e1); }
        if (cur.k >= 3 && !gave_up) { settle(cur.w2, cur.e2); if (!gave_up) candidate(cur.w2, cur.e2); }
        if (cur.k >= 4 && !gave_up) { settle(cur.w3, cur.e3); if (!gave_up) candidate(cur.w3, cur.e3); }
        if (cur.more && !gave_up) {
            // cold: the consumer list itself, one record at a time, for what was not loaded ahead
            for (u32 eb = 0; eb < cur.cnt && !gave_up; eb += 64) {
                u32 blk = A.clist[cur.off + eb + lane];
                C2A_PIN(blk);
                u64 todo = __ballot(eb + lane < cur.cnt && !(use_own && (blk & kIdMask) == own && (blk >> 31) == cur.label) &&
                                    !(cur.k >= 1 && blk == cur.e0) && !(cur.k >= 2 && blk == cur.e1) && !(cur.k >= 3 && blk == cur.e2) && !(cur.k >= 4 && blk == cur.e3));
                while (todo && !gave_up) {
                    const u32 e = rdlane(blk, ctz64(todo));
                    todo &= todo - 1;
                    u64 w = ld_nw(&A.node[(u64)(e & kIdMask) * kNodeWords + lane]);
                    C2A_PIN(w);
                    settle(w, e);
                    if (!gave_up) candidate(w, e);
                }
            }
        }
        if (gave_up) { if (lane == 0) atomicAdd(&A.ctl[P2_ABORT], 1u); wave_join(); failed = true; return true; }
        // ---- the node: its string is the champion's string with the label appended
        u32 depth = 0, my_label = 0, cprev = C2A_NONE, my_pos = 0;
        u64 str = 0;
        if (ch != C2A_NONE) {
            depth = ch_depth + 1; my_label = ch_el;
            u32 wi = ch_pos >> 8, bp = ch_pos & 255u;
            if (wi >= kStrWords) {           // the parent filled its chunk: a fresh one, the parent is its anchor
                cprev = ch; wi = 0; bp = 0;
                str = lane == kHdrWords ? (u64)my_label : 0ull;
            } else {
                cprev = rdlane((u32)ch_w, 1);
                str = ch_x;
            }
            ++bp;
            if (bp == kWordBits) { bp = 0; ++wi; }
            my_pos = (wi << 8) | bp;
        }
        max_level = level > max_level ? level : max_level;
        const u32 tag_hi = epoch << 31;
        u32 w_lo = (u32)str, w_hi = (u32)(str >> 32) | tag_hi;
        w_lo = wrlane_c<0>(depth, w_lo);  w_hi = wrlane_c<0>((ch_root & kIdMask) | tag_hi, w_hi);
        w_lo = wrlane_c<1>(cprev, w_lo);  w_hi = wrlane_c<1>((level & kIdMask) | tag_hi, w_hi);
        w_lo = wrlane_c<2>(my_pos, w_lo); w_hi = wrlane_c<2>(tag_hi, w_hi);
        const u64 my_w = (u64)w_lo | ((u64)w_hi << 32);
        st_nw(&A.node[(u64)g * kNodeWords + lane], my_w);
        if (lane == 0) {
            A.meta[g] = make_uint4(ch, depth, ch_root, my_label | (level << 1));
            if (ch != C2A_NONE) A.child[2 * (u64)ch + my_label] = g;
        }
        wave_join();
        own = g; own_root = ch_root; own_depth = depth; own_pos = my_pos; own_level = level; own_w = my_w; own_str = str;
        cur.st = 0;
        ++processed;
        ++rseq;
        if (((u32)rseq & 15u) == 0) { if (lane == 0) st_nw(&A.mb_rd[(u64)mail * 16], rseq); wave_join(); }
        return false;
    };
    TourSlot S0, S1, S2;
    S0.st = S1.st = S2.st = 0;
    S0.g = S1.g = S2.g = 0;
    load_entry();
    for (;;) {
        if (step(S0, S1, S2)) break;
        if (step(S1, S2, S0)) break;
        if (step(S2, S0, S1)) break;
    }
    if (lane == 0) {
        st_nw(&A.mb_rd[(u64)mail * 16], rseq);
        if (processed) atomicAdd(&A.ctl[P2_PROCESSED], processed);
        if (max_level) atomicMax(&A.ctl[P2_MAXLEVEL], max_level);
        if (rereads) atomicAdd(&A.ctl[P2_REREADS], rereads);
        if (A.dbg & 4u) { atomicAdd(reinterpret_cast<ull*>(A.ctl + 16), tt_entry); atomicAdd(reinterpret_cast<ull*>(A.ctl + 18), tt_settle); atomicAdd(reinterpret_cast<ull*>(A.ctl + 20), c2a_now() - tt_start); }
    }
}

// ------------------------------------------------------------------------------------------------
// claim side
// ------------------------------------------------------------------------------------------------
// LDS written by some lanes of THE wave is read by others: one wave runs in lock step, so all it takes is that the writes
// have been performed (lgkmcnt) — __syncthreads() would also wait for every global store in flight (the mailbox entries:
// ~0.7 us each time)
__device__ __forceinline__ void wave_lds_sync() {
#ifdef C2A_EMULATE
    __syncthreads();
#else
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#endif
}
__device__ __forceinline__ u32 lanes_below(u64 mask, u32 lane) { return (u32)__popcll(mask & ((1ull << lane) - 1ull)); }

// back-pressure: a lane's ring is full — the wave waits until its tournament wave has caught up (returns what that wave
// has consumed by now; ~0: the launch was given up)
__device__ __attribute__((noinline)) u64 mb_wait_room(const u64* rd_word, u32* ctl, bool put, u64 wseq, u64 rd_seen, u32 cap, u32 lane) {
    u32 polls = 0;
    for (;;) {
        const bool full = put && wseq - rd_seen >= (u64)(cap - 2u);
        if (__ballot(full) == 0ull) break;
        if (full) rd_seen = ld_nw(rd_word);
        if ((++polls & 63u) == 0) {
            u32 ab = 0;
            if (lane == 0) ab = ld_a32(&ctl[P2_ABORT]); wave_join();
            if (rdlane(ab, 0) || polls > kP2PollLimit) return ~0ull;
        }
        peel_sleep(16);
    }
    return rd_seen;
}

__device__ __forceinline__ void claim_role(const Peel2Args& A) {
    __shared__ u32 s_pool[kP2LdsBlocks][kPoolCap * kPoolWords];
    u32* pool = s_pool[kP2LdsBlocks > 1 ? blockIdx.x % kP2LdsBlocks : 0];
    const u32 lane = threadIdx.x, wv = blockIdx.x;
    const bool on = lane < A.n_lanes;
    const u32 mail = wv * A.n_lanes + (on ? lane : 0u);
    ull* acct = reinterpret_cast<ull*>(A.ctl + P2_ACCT + (wv & (kP2Shards - 1u)) * 32);
    ull* hq_pc = reinterpret_cast<ull*>(A.ctl + P2_HQPC);
    u64* ring = A.mb + (u64)mail * A.mb_cap * 2;
    // lane state: the gate in hand and its two static records 
    bool have = false;
    u32 g = 0;
    uint4 ga = make_uint4(0, 0, 0, 0), gb = make_uint4(0, 0, 0, 0);
    u64 wseq = on ? A.mb_seq_in[mail] : 0ull, rd_seen = wseq;
    // wave state
    u32 pool_head = 0, pool_cnt = 0;            // LDS pool (a ring): entries [head, head + cnt); never more than kPoolCap - kPoolRoom at the top of a turn
    bool seeds_left = true;
    u32 region = 0, ridx = 0, rcnt = 0;
    u32 st_claimed = 0, st_pushed = 0, st_popped = 0, st_stalls = 0, st_turns = 0;
    ull st_wait = 0, st_rest = 0, st_lanes = 0, st_t = c2a_now();
    u32 st_active = 0, idle_turns = 0;
    bool aborted = false;

    // an entry for the gate this lane takes on now.  (The ring has room practically always; waiting for room is out of line
    // so that the hot path holds no loop — a loop that may load makes the compiler wait for EVERYTHING in flight at its
    // head, i.e. for the acknowledgement of the entries stored a moment ago, on every call)
    auto mb_put = [&](bool put, u32 gate, u32 label, u32 cont, u32 off, u32 cnt) {
        if (__ballot(put && wseq - rd_seen >= (u64)(A.mb_cap - 2u)) != 0ull) {
            const u64 r = mb_wait_room(A.mb_rd + (u64)mail * 16, A.ctl, put, wseq, rd_seen, A.mb_cap, lane);
            if (r == ~0ull) aborted = true; else rd_seen = r;
            ++st_stalls;
        }
        if (put && !aborted) {
            const u32 slot = (u32)wseq & (A.mb_cap - 1u);
            const u64 tag = (u64)(((u32)(wseq >> A.mb_lg) + 1u) & 0xFFFFu) << 48;
            const u32 c16 = cnt < 0xFFFFu ? cnt : 0xFFFFu;
            st_nw(&ring[(u64)slot * 2], tag | ((u64)(cont & 1u) << 32) | (u64)(gate | (label << 31)));
            st_nw(&ring[(u64)slot * 2 + 1], tag | ((u64)c16 << 32) | (u64)off);
            ++wseq;
        }
    };
    // lanes without a gate take the oldest pool entries
    auto take_from_pool = [&]() {
        const u64 idle = __ballot(on && !have);
        if (idle == 0ull || pool_cnt == 0) return;
        const u32 n_id = (u32)__popcll(idle);
        const u32 take = n_id < pool_cnt ? n_id : pool_cnt;
        const u32 r = lanes_below(idle, lane);
        const bool mine = on && !have && r < take;
        u32 e0 = 0, off = 0, cnt = 0;
        if (mine) {
            const u32* p = pool + ((pool_head + r) & (kPoolCap - 1u)) * kPoolWords;
            e0 = p[0]; off = p[1]; cnt = p[2];
            ga = make_uint4(p[4], p[5], p[6], p[7]); gb = make_uint4(p[8], p[9], p[10], p[11]);
            g = e0 & kIdMask; have = true;
        }
        pool_head = (pool_head + take) & (kPoolCap - 1u); pool_cnt -= take;
        mb_put(mine, e0 & kIdMask, e0 >> 31, 0u, off, cnt);
    };

    for (;;) {
        ++st_turns;
        // ---- 1. tickets on both producers; their static records are loaded alongside (whichever is claimed is next).
        // Everything is ISSUED before anything is looked at: one round trip per turn.  (The loads are branch-free — a lane
        // without a producer reads record 0 and ignores it —, and the tickets' results are pinned below the last issue:
        // left alone, the compiler moves "ticket + 1" up into the branch of the atomic and waits for it right there.)
        const u32 d0 = ga.x, d1 = ga.y;
        const u32 cnt0 = have ? gb.y : 0u, cnt1 = have ? gb.w : 0u;           // (0: no such producer)
        u32 t0 = 0, t1 = 0;
        if (cnt0 > 1u) t0 = atomicAdd(&A.fill[d0], 1u);                       // (a single-consumer producer needs no ticket)
        if (cnt1 > 1u) t1 = atomicAdd(&A.fill[d1], 1u);
        const u64 i0 = cnt0 ? d0 : 0u, i1 = cnt1 ? d1 : 0u;
        const uint4 p0a = A.gstat[2 * i0], p0b = A.gstat[2 * i0 + 1];
        const uint4 p1a = A.gstat[2 * i1], p1b = A.gstat[2 * i1 + 1];
        // (three lanes also look at the global queue and the flags: the answers are there when the tickets are)
        u64 pc = 0;
        u32 flags = 0;
        if (lane == 0) pc = ld_nw(reinterpret_cast<const u64*>(hq_pc));
        if (lane - 1u < 2u) flags = ld_a32(&A.ctl[lane == 1 ? P2_ABORT : P2_DONE]);
        const ull tw0 = (A.dbg & 2u) ? c2a_now() : 0;
        C2A_PIN(t0); C2A_PIN(t1);
        if (A.dbg & 2u) { u32 x = p0a.x ^ p0b.x ^ p1a.x ^ p1b.x ^ (u32)pc ^ flags; C2A_PIN(x); const ull tw1 = c2a_now(); st_wait += tw1 - tw0; st_rest += tw0 - st_t; st_t = tw1; }
        // ---- 2. what this gate completed
        const bool c0 = cnt0 && t0 + 1u == cnt0, c1 = cnt1 && t1 + 1u == cnt1;
        const bool extra = c0 && c1;
        const u32 n_claimed = (u32)__popcll(__ballot(c0)) + (u32)__popcll(__ballot(c1));
        const u32 n_done = (u32)__popcll(__ballot(have));
        st_lanes += n_done; st_active += n_done ? 1u : 0u;
        pc = rdlane64(pc, 0);
        if (rdlane(flags, 1) || aborted) break;
        const bool done_flag = rdlane(flags, 2) != 0;
        st_claimed += n_claimed;
        u32 b_add = n_claimed, e_add = n_done;
        // ---- 3. second producers into the pool (it has room for 64: see 6.)
        {
            const u64 exm = __ballot(extra);
            if (extra) {
                u32* p = pool + ((pool_head + pool_cnt + lanes_below(exm, lane)) & (kPoolCap - 1u)) * kPoolWords;
                p[0] = d1 | 0x80000000u; p[1] = gb.z; p[2] = gb.w; p[3] = 0;
                p[4] = p1a.x; p[5] = p1a.y; p[6] = p1a.z; p[7] = p1a.w; p[8] = p1b.x; p[9] = p1b.y; p[10] = p1b.z; p[11] = p1b.w;
            }
            pool_cnt += (u32)__popcll(exm);
        }
        // ---- 4. this lane's next gate: the first producer it claimed
        {
            const bool go_on = c0 || c1;
            const bool first0 = c0;
            const u32 ng = first0 ? d0 : d1;
            const u32 n_off = first0 ? gb.x : gb.z, n_cnt = first0 ? gb.y : gb.w;
            mb_put(go_on, ng, first0 ? 0u : 1u, 1u, n_off, n_cnt);
            if (go_on) { g = ng; ga = first0 ? p0a : p1a; gb = first0 ? p0b : p1b; }
            have = go_on;
        }
        if (aborted) break;
        wave_lds_sync();                                               // (the pool entries written above are read below)
        // ---- 5. lanes without work: the pool
        take_from_pool();
        if (aborted) break;
        wave_lds_sync();                                               // (the pool slots read above may be written below / next turn)
        // ---- 6. the pool against the global queue and the seeds, 64 entries at a time
        if (pool_cnt > kPoolCap - kPoolRoom) {
            // too full to be sure of room for the next turn: the oldest entries go to the global queue, a batch at a time.  The pushes are
            // counted (P) before this turn's gates are counted as dealt with (E): a returning atomic, waited for
            u64 base = 0;
            if (lane == 0) base = atomicAdd(hq_pc, (ull)kPoolRoom); wave_join();
            base = rdlane64(base, 0) & 0xFFFFFFFFull;
            if (lane < kPoolRoom) {
                const u32* p = pool + ((pool_head + lane) & (kPoolCap - 1u)) * kPoolWords;
                u64* q = A.hq + ((base + lane) % A.hq_cap) * kHqWords2;
                const u64 rt = (u64)A.run << 32;
#pragma unroll
                for (int k = 0; k < 12; ++k) st_nw(q + k, rt | p[k]);
            }
            pool_head = (pool_head + kPoolRoom) & (kPoolCap - 1u); pool_cnt -= kPoolRoom;
            e_add += kPoolRoom;                                    // (handed over: dealt with as far as this wave goes; P counts them from here)
            st_pushed += kPoolRoom;
            wave_lds_sync();
        } else if (pool_cnt == 0 && __ballot(on && !have) != 0ull) {
            if (seeds_left) {
                while (ridx >= rcnt) {
                    u32 r = 0;
                    if (lane == 0) r = atomicAdd(&A.ctl[P2_SEEDNEXT], 1u); wave_join();
                    r = rdlane(r, 0);
                    if (r >= A.n_regions) { seeds_left = false; break; }
                    region = r; ridx = 0; rcnt = uniform(A.seed_cnt[r]);
                }
                if (seeds_left) {
                    // (a seed's static records are waited for right here: a load still pending when the loop starts over
                    // would make the compiler wait for everything in flight at the top of every turn)
                    const u64 idle = __ballot(on && !have);
                    const u32 n_id = (u32)__popcll(idle);
                    const u32 avail = rcnt - ridx;
                    const u32 take = n_id < avail ? n_id : avail;
                    const u32 r = lanes_below(idle, lane);
                    const bool mine = on && !have && r < take;
                    u32 sd = 0;
                    uint4 sa = make_uint4(0, 0, 0, 0), sb = sa;
                    if (mine) {
                        sd = A.seeds[(u64)region * A.region_cap + ridx + r];
                        sa = A.gstat[2 * (u64)sd]; sb = A.gstat[2 * (u64)sd + 1];
                    }
                    C2A_PIN(sa.x); C2A_PIN(sa.y); C2A_PIN(sa.z); C2A_PIN(sa.w); C2A_PIN(sb.x); C2A_PIN(sb.y); C2A_PIN(sb.z); C2A_PIN(sb.w);
                    if (mine) { g = sd; ga = sa; gb = sb; have = true; }
                    ridx += take;
                    mb_put(mine, sd, 0u, 0u, sa.z, sa.w);
                    if (aborted) break;
                }
            } else {
                const u32 prod = (u32)pc, cons = (u32)(pc >> 32);
                if (prod != cons) {
                    // up to 64 entries that are pushed already, into the pool (compare-and-swap on the pair of counts); B counts
                    // them BEFORE the pop (a returning atomic, waited for)
                    const u32 avail = prod - cons;
                    const u32 want = avail < kPoolRoom ? avail : kPoolRoom;
                    u64 old = 0;
                    if (lane == 0) old = atomicAdd(acct, (ull)want); wave_join();
                    C2A_PIN(old);
                    u64 seen = 0;
                    if (lane == 0) seen = atomicCAS(hq_pc, (ull)pc, (ull)(pc + ((u64)want << 32))); wave_join();
                    seen = rdlane64(seen, 0);
                    if (seen == pc) {
                        const bool mine = lane < want;
                        const u64* q = A.hq + (((u64)cons + lane) % A.hq_cap) * kHqWords2;
                        u64 v[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                        u32 polls = 0;
                        for (;;) {
                            bool ok = true;
                            if (mine) {
#pragma unroll
                                for (int k = 0; k < 12; ++k) { v[k] = ld_nw(q + k); ok = ok && (u32)(v[k] >> 32) == A.run; }
                            }
                            if (__ballot(!ok) == 0ull) break;
                            if (++polls > kP2PollLimit) { aborted = true; break; }
                            peel_sleep(4);
                        }
                        if (aborted) break;
                        if (mine) {
                            u32* p = pool + ((pool_head + lane) & (kPoolCap - 1u)) * kPoolWords;
#pragma unroll
                            for (int k = 0; k < 12; ++k) p[k] = (u32)v[k];
                        }
                        pool_cnt += want;
                        st_popped += want;
                        wave_lds_sync();
                        take_from_pool();
                        if (aborted) break;
                        wave_lds_sync();
                    } else {
                        e_add += want;                                 // (somebody else took them: give the count back)
                    }
                }
            }
        }
        // ---- 7. the turn's counts (after the pushes, before anything of the next turn)
        if (b_add | e_add) { if (lane == 0) atomicAdd(acct, (ull)b_add | ((ull)e_add << 32)); wave_join(); }
        // ---- 8. a wave with nothing at all: is everything over?
        if (__ballot(have) == 0ull && pool_cnt == 0 && !seeds_left) {
            if (done_flag) break;
            // (a wave without work looks at the counts once in a while only: the lines they live on take the working waves' atomics)
            if ((++idle_turns & 7u) != 0) { peel_sleep(64); continue; }
            // every F part first (E of all shards, the pops), then every T part (B, the pushes): each read waits for the one before
            u64 a1 = 0;
            if (lane < kP2Shards) a1 = ld_nw(reinterpret_cast<const u64*>(A.ctl + P2_ACCT + lane * 32));
            if (lane == kP2Shards) a1 = ld_nw(reinterpret_cast<const u64*>(hq_pc));
            C2A_PIN(a1);
            u32 f = (u32)(a1 >> 32);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) f += __shfl_xor(f, off, 64);
            f = uniform(f);
            u64 a2 = 0;
            if (lane < kP2Shards) a2 = ld_nw(reinterpret_cast<const u64*>(A.ctl + P2_ACCT + lane * 32));
            if (lane == kP2Shards) a2 = ld_nw(reinterpret_cast<const u64*>(hq_pc));
            C2A_PIN(a2);
            u32 t = (u32)a2;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off, 64);
            t = uniform(t);
            if (f == t) { if (lane == 0) st_a32(&A.ctl[P2_DONE], 1u); wave_join(); break; }
            peel_sleep(32);
        }
    }
    // ---- the mailboxes end here
    if (!aborted) mb_put(on, kMbSentinel, 0u, 0u, 0u, 0u);
    if (on) A.mb_seq_out[mail] = wseq;
    if (lane == 0) {
        if (st_claimed) atomicAdd(&A.ctl[P2_CLAIMED], st_claimed);
        if (st_pushed) atomicAdd(&A.ctl[P2_PUSHED], st_pushed);
        if (st_popped) atomicAdd(&A.ctl[P2_POPPED], st_popped);
        if (st_stalls) atomicAdd(&A.ctl[P2_STALLS], st_stalls);
        atomicAdd(&A.ctl[P2_TURNS], st_turns);
        if (A.dbg & 2u) {
            atomicAdd(reinterpret_cast<ull*>(A.ctl + 10), st_wait); atomicAdd(reinterpret_cast<ull*>(A.ctl + 12), st_rest);
            atomicAdd(reinterpret_cast<ull*>(A.ctl + 14), st_lanes); atomicAdd(&A.ctl[9], st_active);
        }
    }
}

// the gates the sinks pass claimed are the first that come into being: B of shard 0 starts at their number
__global__ void k_seed_total(u32 n_regions, const u32* __restrict__ seed_cnt, u32* acct0) {
    __shared__ u32 s_sum[256];
    u32 t = 0;
    for (u32 i = threadIdx.x; i < n_regions; i += blockDim.x) t += seed_cnt[i];
    s_sum[threadIdx.x] = t;
    __syncthreads();
    for (u32 off = blockDim.x / 2; off; off >>= 1) {
        if (threadIdx.x < off) s_sum[threadIdx.x] += s_sum[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) { acct0[0] = s_sum[0]; acct0[1] = 0; }
}

__global__ void __launch_bounds__(64) k_peel2(Peel2Args A) {
    if (blockIdx.x < A.n_claim) claim_role(A);
    else tour_role(A, blockIdx.x - A.n_claim);
}

}  // namespace c2a

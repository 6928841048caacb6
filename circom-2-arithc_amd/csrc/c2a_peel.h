// c2a_peel.h — the exact DFS tree of topological_sort (/root/reference/src/topological_sort.rs:3-50) without a DFS,
// as ONE dataflow launch on gfx950.
//
// Claim (re-proved by the tests): the DFS tree parent of gate g is the consumer that reaches g by the
// lexicographically smallest path from a virtual root whose children are the gates 0..n-1 in id order
// (topological_sort.rs:11-13) and whose other edges carry label 0 (lh producer) / 1 (rh producer)
// (compiler.rs:408-421, topological_sort.rs:42-44); the DFS post-order is the post-order of that tree.  A gate can pick
// its parent once every consumer has picked its own (reverse Kahn order from the sinks).
// GATE IDS IN THIS FILE ARE RANKS (c2a_kernels.h RELABELLING: the position of a gate's out node among the produced nodes, i.e.
// the creation order) — records, tickets, consumer lists and tree entries of gates that are claimed together lie together —,
// except where the DFS roots are compared: word 2 of a gate's first static record is its ORIGINAL id, the "root" fields of
// records and tree entries hold original ids, and a gate is its own root when no consumer's root has a smaller original id.
//
// NODE RECORDS.  Every gate owns one 512-byte record of 64 self-validating 8-byte words: the top two bits of a word are the
// tag of the run that wrote it (a word is written by ONE agent-scope store, so it is never torn; a reader that sees a stale
// tag simply reads again — no flag, no fence, no acknowledgement wait anywhere on the path).  THREE states: 1 and 2 take
// turns from run to run (a word of the run before is STALE), 0 is a word that no run has ever written — the buffer is
// zeroed once per loaded graph —, i.e. the zero padding behind a SHORT record (a sink writes three header words, a gate of
// the whole-level passes sixteen words): valid in every run but the first after the clear, where "not written yet" and
// "never written" cannot be told apart and every lane must carry this run's tag (the reader's out-of-line path then
// decides by the record's depth which lanes count).  Round 4 had ONE tag bit: the padding of short records looked valid in
// every second run only — k_peel alternated between 6.95 and 7.20 ms with the run tag.
//   word 0      root gate id << 32 | depth in the DFS tree
//   word 1      reverse Kahn level << 32 | cprev (ancestor at the start of the node's current chunk; deep trees only)
//   word 2      high half: where a child's edge label goes in the RECORD: lane (= header words + string word) << 8 | shift (no
//               division, no offset on the hot path);
//               low half: the ROOT KEY, the original id of the DFS root — a candidate's compare word of lane 2, so that ONE
//               compare decides root and path: the lowest differing lane, and whoever holds the SMALLER word there wins
//   words 3..63 the node's path string, ZERO-PADDED, MOST SIGNIFICANT FIRST: 62 payload bits per word, the label of the edge
//               entering depth j + 1 at bit 61 - j % 62 of word j / 62 — the lexicographic order of two paths is the numeric order
//               of their first differing words (round 4 stored the first label at bit 0: a compare was four lane reads and a
//               chain of scalar bit tricks; this is two vector compares and a bit test)
// A string is held in chunks of kChunkBits = 61 x 62 = 3782 bits; a node keeps its CURRENT chunk only.  Comparing two
// candidates P(a).la and P(b).lb that are shallower than a chunk (the 10 M-gate headline graph: depth 3 471): append
// each label to its string, XOR, ballot, count trailing zeros — neither path can be a prefix of the other (that would be
// a cycle), so the first differing bit decides and the zero padding needs no length masks.  A new node's string is its
// parent's string with the label appended: exactly the register the comparison already built.  Deeper trees add one
// cprev hop per chunk level (out of line).
//
// DATAFLOW.  The consumer list of every gate is static (clist, built by k_gstat from the deps closure), so nothing but
// the node records is exchanged between waves.  A gate g is CLAIMED by the wave that takes the last ticket on fill[g];
// tickets are taken when a consumer is claimed, NOT when it is finished, so the ticket round trip overlaps the
// consumer's own tournament and the claiming wave goes on with g right after finishing the consumer: its own record is
// still in registers, the other consumers' records are loaded in one round trip (and read again if a word is not there
// yet).  A single-consumer producer needs no ticket at all.  The chain loop is SOFTWARE-PIPELINED: the loads and tickets
// of the next step are issued as soon as this step's tickets say where the chain goes on, before this step's own
// tournament and stores.
// A step is ~300 instructions of one wave, most of them scalar and dependent on the one before; the memory round trip of
// the tickets and loads issued at its top hides behind them.  What the step is written in — scalar-memory atomics for the
// tickets and scalar stores for the tree entries, with results in registers reserved from the compiler; branch weights
// that keep the hot path falling through; lane reads / writes and DPP instead of shuffles — is in c2a_wave.h, each
// primitive next to its twin for the host emulation.  Up to eight candidate records are loaded one step ahead.
// The first few levels behind the sinks are far wider than the body of a circuit and are done a whole level at once by
// k_peel_shallow (a lane per gate) before the launch; what the last of those passes claims seeds it.
// HAND-OFF.  A second producer completed by the same gate goes to one of F first-in-first-out arrays with tickets on
// BOTH sides: a wave without work takes a consumer ticket c on one of them and watches slot c alone; the pusher takes a
// producer ticket p (the round trip overlaps its tournament) and stores the entry into slot p — there is no claim step
// and no search on the receiving side.  F arrays, not one: a fifth of all gates is handed off, and one counter takes
// agent-scope atomics from eight XCDs at some ten nanoseconds apiece.  An entry is one 128-byte line of sixteen
// self-validating words (payload | number of the run << 32, so the slots are never cleared): the pushed gate's two gstat
// records, its first seven consumers and its id — all of it in the pusher's registers anyway (prefetched as static data
// of the producers of ITS gate) — so the receiver issues the gate's tickets and record loads straight from the entry.
// A wave is committed to the slot its ticket names; an entry pushed to an array where nobody waits any more would sit
// there for good, so a waiting wave that sees such a backlog takes a ticket that is ALREADY served (compare-and-swap on
// the array's pair of counts) and works on that entry; its own ticket stays good until it is served.
// No deadlock: a ticket on fill[] is only ever taken by a wave for a step that it then runs to the end, so when a gate is
// claimed every one of its candidates is in a step that is running (or done) on some resident wave; a wave only ever
// waits for such a record, the graph is acyclic, so every wait ends.
// Termination: two counters that only grow (kept in 64 parts).  BEGIN counts units of work when they come into being: a
// wave that starts (one unit until it first runs out of work), and every pushed entry — which is counted by the ticket its
// pusher takes on the array anyway (BEGIN = wave starts + the producer sides of the arrays' ticket words): a returning atomic
// the pusher waits for BEFORE the entry is visible.  END counts a unit when the wave that worked on it runs out of work.
// END never passes BEGIN, so "END (read first) == BEGIN (read afterwards)" means they were equal at every moment in
// between: nothing active, nothing in flight, nobody can push any more.  How entries travel (which array, whose ticket)
// plays no part in it, and co-residency of the grid is not required: a wave that starts late adds its unit, finds the
// seed pool empty, ends it, and sees the counters equal.  (One waiting wave in 64 looks at the counters every 32 polls
// and raises a DONE flag for the others; every wave looks for itself once in 1024 polls.)
// HUBS AND RELAYS.  A step loads eight candidate records ahead and takes the rest of a consumer list eight at a time, on ONE
// wave: fine for the fan-outs a circuit mostly has, not for a produced node that a thousand — or a million — gates read (a
// broadcast selector, a scale factor: compiler.rs:408-421 allows any fan-out).  k_gstat therefore gives every gate with more than
// kHubMin consumers a TREE OF RELAYS: virtual gates n, n + 1, ... (behind the real ones in every per-gate array), each the only
// "producer" of at most kRelayFan consumers — a segment of the hub's consumer list as it lies; the next level's relays consume
// the relays; the hub's own consumers are the (at most kHubMin) relays of the top level.  A relay is an ordinary gate to the
// launch — tickets, a record, a chain step — with the edge to its parent labelled 0, a root key no gate can have (it is never
// its own DFS root) and a level that does not count (the reverse Kahn levels stay exact).  Its tournament picks the smallest
// path among its segment, the parent's among the relays: the hub ends up with the smallest of all, with one more 0 bit per relay
// level behind the real label.  Those extra bits change no comparison: two paths that both run through the hub carry them at
// the same place, and any other pair differs before them (neither is a prefix of the other).  The work of a hub's tournament is
// thereby spread over as many waves as it has segments, and over TIME: a relay runs when the last of its sixteen consumers is
// claimed, not when the hub is.  Where the tree is read afterwards (k_euler_next: tree_child / tree_parent) the relays are
// skipped again: the hub's parent is the consumer its bottom relay chose, with that edge's label.
// Watchdog: lack of GLOBAL progress (a heartbeat the working waves bump) — a long critical path followed by one wave is
// not an error.
#pragma once
#include "c2a_platform.h"
#include "c2a_wave.h"

#ifndef C2A_POLL_CAP
// longest back-off of a wave that waits for a hand-off entry, in units of 64 clocks.  Round 5 backed off to 64 units (~1.8 us): on a wide graph
// the launch does not wait for hand-offs, but a DEEP AND NARROW circuit — a chain of hash blocks, a few gates per level, four gates in ten handed
// off — waits for every one of them: SHA-256 over 8 blocks 6.10 -> 5.06 ms with 4 units, a Merkle tree of 290 blocks 6.5 -> 5.85, the headline
// 5.88 -> 5.86 (its tail: the end of the launch is seen sooner); 1 unit: no better, 16: half the gain (tools/ab_configs.sh)
#define C2A_POLL_CAP 4
#endif
#ifndef C2A_PARK_MONITORS
#define C2A_PARK_MONITORS 512u      /* one parked wave in this many watches the hand-off arrays' backlog (a power of two) */
#endif
#ifndef C2A_HB_MASK
// a wave tells the watchdog that the launch is alive once in (mask + 1) steps: an atomic on the ONE line every waiting wave looks at (a waiting
// wave gives up after 3 s without one; a step is ~1 us).  Once in 64 steps -> once in 512: SHA-256 Merkle tree of 290 blocks 5.73 -> 5.55 ms, of
// 2 900 blocks 12.7 -> 12.5, the headline's k_peel 5.89 -> 5.84 (tools/ab_family.sh, ab_configs.sh).  (The emulation counts looks, not time.)
#ifdef C2A_EMULATE
#define C2A_HB_MASK 63u
#else
#define C2A_HB_MASK 511u
#endif
#endif
#ifndef C2A_SCAN_MASK
// one waiting wave in 64 reads the launch's 192 counter lines (is it over?  is an entry stranded?) every (mask + 1) looks at its slot = ~14 us;
// every wave does once in 16 384 looks (~1.8 ms: ending never depends on a particular wave being resident).  Per 32 / 128 / 512 looks: SHA-256 x 8
// blocks 4.94 / 4.95 / 4.95 ms, the headline's k_peel 5.80 / 5.80 / 5.90 (a longer tail); every wave once in 1 024 looks (with the 256-clock look
// that is every 0.11 ms, 450 GB/s of reads on the lines the tickets live on when most waves wait): 5.06-5.12 ms for the SHA-256 chain
#define C2A_SCAN_MASK 127u
#endif
#ifndef C2A_PRIO_TOUR
#define C2A_PRIO_TOUR 1      /* wave priority from the issue of the next step to the end of the step (3 in front of it) */
#endif

namespace c2a {

constexpr u32 kIdMask = 0x7FFFFFFFu;
constexpr u32 kNodeWords = 64;
constexpr u32 kHdrWords = 3;
constexpr u32 kStrWords = kNodeWords - kHdrWords;                     // 61
constexpr u32 kWordBits = 62;
constexpr u32 kChunkBits = kStrWords * kWordBits;                     // 3782
constexpr u32 kTagShift = 62;
constexpr u64 kPayload = (1ull << kTagShift) - 1ull;
constexpr u32 kHdrMask = 0x3FFFFFFFu;                                 // the high half of a header word below the tag (gate ids are below 2^29)
// A word is BAD (stale, or not there yet) by ONE compare of its high half: hi ^ (stale tag << 30) puts the stale tag at 0, a
// never-written word between (1 or 2) and this run's tag at 3 — bad below 3 << 30 where this run's tag is required (the
// header lanes always, every lane in the first run after the clear), bad below 1 << 30 where never-written counts as
// zero padding.
constexpr u32 kThrExact = 0xC0000000u, kThrPad = 0x40000000u;
struct TagCheck { u32 flip; u32 thr; };                               // per lane: thr = lane < kHdrWords ? kThrExact : pad threshold of the run
__device__ __forceinline__ TagCheck tag_check(u32 epoch, u32 pad_thr, u32 lane) { return TagCheck{(3u - epoch) << 30, lane < kHdrWords ? kThrExact : pad_thr}; }
__device__ __forceinline__ bool tag_bad(const TagCheck& T, u64 w) { return ((u32)(w >> 32) ^ T.flip) < T.thr; }
__device__ __forceinline__ bool tag_stale_or_never(u32 epoch, u64 w) { return (u32)(w >> kTagShift) != epoch; }      // (cold paths: this run's tag exactly)

#ifdef C2A_EMULATE
constexpr u32 kPollLimit = 1u << 16;        // (the emulation runs every wave of the launch side by side and switches at the back-offs: a poll is one turn of all the others)
#else
constexpr u32 kPollLimit = 1u << 21;        // ~1 s of polling for a record: give up (reported as an error) instead of hanging
#endif
// The watchdog of a waiting wave: no GLOBAL progress (the heartbeat) for ~3 s.  By the clock on the device — a waiting wave looks at its
// slot every 256 clocks (C2A_POLL_CAP) and at the launch's state every 32 looks —, by the number of looks under the emulation (no clock there).
constexpr u32 kWatchdogChecks = 1u << 15;
constexpr ull kWatchdogTicks = 300000000ull;      // of the constant 100 MHz clock
// control block (u32 words; every hot word on its own 128-byte line)
enum PeelCtl { CTL_PROCESSED = 0, CTL_MAXLEVEL = 1, CTL_ABORT = 2, CTL_REREADS = 3, CTL_DONE = 4, CTL_NEEDDEEP = 5, CTL_HEARTBEAT = 32, CTL_SEEDNEXT = 64,
               CTL_BEGIN = 128, CTL_END = CTL_BEGIN + 64 * 32, CTL_PARK = CTL_END + 64 * 32,
               // gates done (word 0) and the highest level seen (word 1), in kAcctShards parts like BEGIN / END: thousands of
               // workgroups (sinks pass, level-1 pass) and every wave of the launch report here as they leave, and atomics on
               // ONE word go one at a time, ~10 ns each (measured: 4 096 workgroups x 3 such atomics were 120 us of the level-1 pass)
               CTL_PROC = CTL_PARK + 64 * 32, CTL_WORDS = CTL_PROC + 64 * 32 };
constexpr u32 kPcStride = 16;               // u64 words between two hand-off arrays' ticket words (128 bytes)
constexpr u32 kSlotWords = 16;              // a hand-off entry: words 0..7 gstat[2g], gstat[2g + 1]; 8..14 first consumers; 15 gate id
constexpr u32 kSlotCons = 7;
constexpr u32 kAcctShards = 64;             // BEGIN / END are kept in this many parts, kAcctStride words (128 bytes) apart
constexpr u32 kAcctStride = 32;
constexpr u32 kFillDummyStride = 32;        // u32 words between two waves' dummy ticket words behind fill[n]
constexpr u32 kFillDummyWaves = 8192;       // ... of at most this many waves (256 CUs x 32)

// ---- hubs and relays (HUBS AND RELAYS above) ----
#ifndef C2A_RELAY_FAN
#define C2A_RELAY_FAN 8
#endif
#ifndef C2A_HUB_MIN
#define C2A_HUB_MIN 8
#endif
constexpr u32 kRelayFan = C2A_RELAY_FAN;    // consumers of one relay: eight records loaded ahead + one batch of the list loop
constexpr u32 kHubMin = C2A_HUB_MIN;        // a gate with MORE consumers than this gets a relay tree
constexpr u32 kShallowSkip = C2A_NONE - 1u; // (k_peel_shallow: a lane whose gate turned out to be a relay)
static_assert((kHdrMask >> 29) & 1u, "inc_unless_bit<29> tells a relay by its original id");
constexpr u32 kRelayOrig = kHdrMask;        // a relay's "original id" (word 2 of its first static record): above every gate id, so it is never its own DFS root — and how a step knows a relay
static_assert(kRelayFan >= 8 && kRelayFan <= 32 && kHubMin >= kRelayFan && kHubMin <= 32, "a relay's consumers and a hub's top-level relays fit the prefetched half list (32 lanes)");
__host__ __device__ constexpr u32 relay_ceil(u32 a) { return (a + kRelayFan - 1u) / kRelayFan; }
// relays of a hub with N > kHubMin consumers (all levels), and where its top level starts / how many relays that level has
__host__ __device__ constexpr u32 relay_count(u32 N) { u32 c = relay_ceil(N), t = c; while (c > kHubMin) { c = relay_ceil(c); t += c; } return t; }
struct RelayTop { u32 off, cnt; };
__host__ __device__ constexpr RelayTop relay_top(u32 N) { u32 c = relay_ceil(N), o = 0; while (c > kHubMin) { o += c; c = relay_ceil(c); } return RelayTop{o, c}; }
// The relays of ALL hubs together: at most kRelayNum / kRelayDen per edge (worst case: a hub just above kHubMin), and a graph
// of n gates has at most 2 n edges — the per-gate arrays of the launch hold n + relay_cap(n) entries.
constexpr u32 kRelayNum = relay_count(kHubMin + 1u), kRelayDen = kHubMin + 1u;
constexpr bool relay_bound_holds() { for (u32 N = kHubMin + 1u; N < 5000u; ++N) if ((u64)relay_count(N) * kRelayDen > (u64)N * kRelayNum) return false; return true; }
static_assert(relay_bound_holds(), "relay_count(N) <= N * kRelayNum / kRelayDen");
__host__ __device__ constexpr u64 relay_cap(u64 n) { return (2 * n * kRelayNum + kRelayDen - 1) / kRelayDen + 64; }
// WHICH relays a hub gets follows from where its consumer list lies: relays floor(off * kRelayNum / kRelayDen) ... of a hub whose
// list starts at edge `off` — the lists of two hubs are disjoint runs of edges, a hub of N consumers needs at most
// floor(N * kRelayNum / kRelayDen) relays, and floor(a + b) >= floor(a) + floor(b): the runs of relays are disjoint too.  No
// counter, no table, no pass over the gates to hand them out (a counter took one same-address atomic per hub: 0.5 ms for 50 000).
__host__ __device__ constexpr u32 relay_base(u32 off) { return (u32)((u64)off * kRelayNum / kRelayDen); }
// where a gate's OWN consumer list starts in clist (chain starts and the list loop of a step only; a chain step gets its producers'
// lists with their static records): a relay keeps it in word 2 of its second static record (a relay has no second producer), a hub's
// list are its top-level relays, any other gate's is where the scan of the counts put it
__device__ __forceinline__ u32 own_list_off(u32 n, const u32* cons_off, const uint4* gstat, u32 g) {
    if (g >= n) return gstat[2 * (u64)g + 1].z;
    const u32 o = cons_off[g], N = cons_off[(u64)g + 1] - o;
    return N > kHubMin ? 2u * n + 64u + relay_base(o) + relay_top(N).off : o;
}

// what only the edges of the launch touch (kept out of the kernel's scalar registers)
struct PeelCold {
    const u32* cons_off;       // [n + 1] the scan of the consumer counts (own_list_off: chain starts and the list loop only)
    u32 n;                     // real gates (ids from n on are relays)
    const u32* seeds;          // [*seed_total] the gates the launch starts chains from (claimed by the last k_peel_shallow pass), one flat list
    const u32* seed_total;     // how many (device side: the host never learns it)
    u32 seed_chunk;            // a wave takes this many at a time
    ull* stats;                // optional diagnostics (32 words), nullptr normally
    ull* q_time;               // with stats: when the push of every hand-off entry was decided ...
    ull* p_time;               // ... and when the receiver had it ready to issue
    ull* t_trace;              // with stats + C2A_PEEL_TRACE: three words per gate (step start << 2 | how it came; record stored; phases: tools/peel_trace.py)
};

struct PeelArgs {
    u32 epoch;                 // tag (1 / 2, taking turns) of this run's node words
    u32 pad_thr;               // kThrExact in the first run after the node records were cleared, else kThrPad (NODE RECORDS above)
    u32 n;                     // real gates (the sinks pass walks these; relays are never sinks)
    u32 n_all;                 // n + relay_cap(n): entries of every per-gate array below (the waves' dummy ticket words lie behind fill[n_all])
    const uint4* gstat;        // [2 n_all] {dep0, dep1, original gate id, cons_cnt} {cons_off[dep0], cons_cnt[dep0], cons_off[dep1], cons_cnt[dep1]}
    const u32* clist;          // [2n + 64 | relay_cap + 64] consumer | edge label << 31, grouped by producer; behind the real edges: entry i = relay i (the consumer lists of relays' parents)
    u64* node;                 // [n][64] node records
    u32* fill;                 // [n] claim tickets taken so far (zeroed per run)
    uint4* meta;               // [n] {parent | NONE, depth, root, label | level << 1}: read by later launches only
    u32* child;                // [2n] tree children by label (0xFF-filled per run)
    u32 n_fifos, q_cap;        // hand-off arrays (a power of two) and slots of each
    u64* q_pc;                 // [n_fifos * kPcStride] tickets taken: producer side (low word) | consumer side (high word)
    u64* fifo;                 // [n_fifos][q_cap][kSlotWords] slots (used once per run, no wrap-around: a wave spreads its pushes
                               // round robin and holds at most one unserved consumer ticket; never cleared: the tag says which run)
    u32 run;                   // tag of this run's hand-off entries (never zero)
    u32 n_primary;             // waves beyond this many are a RESERVE: parked — out of the hand-off lines — until the launch is short of waves,
    u32 reserve_min;           // i.e. until this many pushed entries are waiting with nobody in line for them (THE RESERVE, below)
    u32* ctl;                  // [CTL_WORDS]
    const PeelCold* cold;
    // the sinks pass only
    u32* seeds_w; u32* seed_cnt_w; u32 region_cap;
    u32 proc_word, proc_mask;  // sinks pass: its count of gates done goes to ctl[proc_word + (workgroup & proc_mask) * kAcctStride]
};

__device__ __forceinline__ u32 chunk_of(u32 depth) { return depth ? (depth - 1) / kChunkBits : 0u; }
__device__ __forceinline__ u32 chunk_len(u32 depth) { return depth - chunk_of(depth) * kChunkBits; }
__device__ __forceinline__ u32 ctz64(u64 x) { return (u32)__builtin_ctzll(x); }      // (callers never pass 0)
// index of the lowest set bit of a wave-uniform mask, which it loses (s_ff1 + s_bitset0: `m &= m - 1` is three instructions)
__device__ __forceinline__ u32 pop_lowest(u64& m) {
    const u32 i = ctz64(m);
#ifdef C2A_EMULATE
    m &= m - 1;
#else
    asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(i));
#endif
    return i;
}

__device__ __forceinline__ u64 ld_nw(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_nw(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 ld_a32(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_a32(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ ull ld_word_time(const ull* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 hdr0_word(u32 root, u32 depth) { return ((u64)(root & kHdrMask) << 32) | depth; }
__device__ __forceinline__ u64 hdr1_word(u32 level, u32 cprev) { return ((u64)(level & kHdrMask) << 32) | cprev; }
__device__ __forceinline__ u32 hdr_hi(u64 w) { return (u32)(w >> 32) & kHdrMask; }

// ---- out-of-line pieces of the deep-tree comparison (cold; plain by-value arguments so that the kernel's argument
// block is never dragged into memory) ----
__device__ __forceinline__ u64 ld_word_wait(const u64* p, u32 epoch, u32* ctl) {
    u64 v = ld_nw(p);
    u32 spins = 0;
    while (tag_stale_or_never(epoch, v)) {
        if (++spins > (1u << 22)) { atomicAdd(&ctl[CTL_ABORT], 1u); break; }
        peel_sleep(4);
        v = ld_nw(p);
    }
    return v;
}
__device__ __forceinline__ u64 ld_rec_wait(const u64* node_base, u32 epoch, u32* ctl, u32 node, u32 lane) {
    const u64* p = node_base + (u64)node * kNodeWords + lane;
    u64 v = ld_nw(p);
    u32 spins = 0;
    while (__ballot(tag_stale_or_never(epoch, v)) != 0ull) {
        if (++spins > (1u << 22)) { if (lane == 0) atomicAdd(&ctl[CTL_ABORT], 1u); break; }
        peel_sleep(4);
        v = ld_nw(p);
    }
    return v;
}
// bit j of the string chunk held one word per lane
__device__ __forceinline__ u32 str_bit(u64 w, u32 j) { return (u32)(rdlane64(w, kHdrWords + j / kWordBits) >> (kWordBits - 1u - j % kWordBits)) & 1u; }
// P(a).la < P(b).lb by lengths (general form: chunks of any fill)
__device__ __forceinline__ bool str_less_len(u64 wa, u32 lena, u32 la, u64 wb, u32 lenb, u32 lb, u32 lane) {
    const u32 minlen = lena < lenb ? lena : lenb;
    const u32 lo = (lane - kHdrWords) * kWordBits;
    // (the first k bits of a word are its TOP k payload bits)
    u64 m = kPayload;
    if (lane < kHdrWords || lo >= minlen) m = 0;
    else if (minlen - lo < kWordBits) m = kPayload & ~((1ull << (kWordBits - (minlen - lo))) - 1ull);
    const u64 a = wa & m, b = wb & m;
    const u64 bal = __ballot(a != b);
    if (bal) return ((__ballot(a < b) >> ctz64(bal)) & 1ull) != 0;
    if (lena == lenb) return la < lb;
    if (lena < lenb) return la < str_bit(wb, lena);
    return str_bit(wa, lenb) < lb;
}
// The lanes of a record that must be there for a node of this depth (in its current chunk): the header words, the string
// words that hold its bits, and the word its child's label goes to.  Everything beyond is zero padding whether it has been
// written or not — a sink's record is its three header words, a record of the shallow passes its first sixteen.
__device__ __forceinline__ u64 needed_lanes(u32 depth) {
    if (depth == 0u) return 7ull;
    const u32 top = kHdrWords + chunk_len(depth) / kWordBits;      // (the label of a child goes to bit chunk_len of the chunk)
    return top >= 63u ? ~0ull : (2ull << top) - 1ull;
}
// A candidate's record was not (all) there when its load arrived: read it again until it is (a sink's record has header
// words only).  Gives up after kPollLimit polls: the caller sees the stale tag.
__device__ __attribute__((noinline)) u64 peel_reread(const u64* node_base, u32 epoch, u32* ctl, u32 c, u64 w, u32 lane) {
    const u64* p = node_base + (u64)c * kNodeWords + lane;
    u32 polls = 0;
    for (;;) {
        u64 badm = __ballot(tag_stale_or_never(epoch, w));
        if ((badm & 1ull) == 0) badm &= needed_lanes((u32)rdlane64(w, 0));          // (word 0 is there: its depth says what else must be)
        if (badm == 0 || ++polls > kPollLimit) break;
        // (the look at ABORT travels WITH the next read of the record, not in front of it: one round trip, not two)
        u32 ab = 0;
        if ((polls & 63u) == 1u) ab = ld_a32(&ctl[CTL_ABORT]);
        peel_sleep(polls < 8 ? 4 : 16);
        w = ld_nw(p);
        if (__ballot(ab != 0u) != 0ull) break;       // the launch is being given up: nobody waits any more
    }
    // (counted in the workgroup's own part of the accounts, word 2: ONE word for all of them sat on the line ABORT and DONE live on,
    // which every waiting wave reads — in a graph where the waves wait for one another's records (strictly layered: STRICT LAYERS in
    // DESIGN.md) that was an atomic per re-read on the line everybody polls)
    if (polls && lane == 0) atomicAdd(&ctl[CTL_PROC + (blockIdx.x & (kAcctShards - 1u)) * kAcctStride + 2], polls);
    return w;
}
// the plain launch has met a node that fills its chunk: tell the host (which runs the DEEP build) and end the launch
__device__ __attribute__((noinline)) void peel_need_deep(u32* ctl, u32 lane) {
    if (lane == 0) { atomicAdd(&ctl[CTL_NEEDDEEP], 1u); atomicAdd(&ctl[CTL_ABORT], 1u); }
}
// Trees deeper than one chunk: bring the two nodes to the first chunk in which their paths can differ (cprev hops),
// then compare that chunk.  wa_in / wb_in: the two nodes' own records (already in registers).
__device__ __attribute__((noinline)) bool deep_less(const u64* node_base, u32 epoch, u32* ctl, u32 a, u32 la, u32 da, u64 wa_in,
                                                    u32 b, u32 lb, u32 db, u64 wb_in, u32 lane) {
#define C2A_CPREV(x) ((u32)ld_word_wait(node_base + (u64)(x) * kNodeWords + 1, epoch, ctl))
#define C2A_BIT0(x) ((u32)(ld_word_wait(node_base + (u64)(x) * kNodeWords + kHdrWords, epoch, ctl) >> (kWordBits - 1u)) & 1u)
    u32 ia = chunk_of(da), ib = chunk_of(db);
    u32 lena = chunk_len(da), lenb = chunk_len(db);
    u32 below_a = C2A_NONE, below_b = C2A_NONE;
    bool a_own = true, b_own = true;
    u32 pa = (u32)rdlane64(wa_in, 1), pb = (u32)rdlane64(wb_in, 1);      // cprev of the node currently held in a / b
    // (a hop is a dependent memory round trip, and a tree 10^7 deep has 2 600 chunk levels: a long climb is work too — the watchdog goes by this)
    u32 hops = 0;
#define C2A_HOP() do { if (C2A_UNLIKELY((++hops & 255u) == 0)) { if (lane == 0) atomicAdd(&ctl[CTL_HEARTBEAT], 1u); wave_join(); } } while (0)
    while (ia > ib) { below_a = a; a = pa; --ia; lena = kChunkBits; a_own = false; if (ia) pa = C2A_CPREV(a); C2A_HOP(); }
    while (ib > ia) { below_b = b; b = pb; --ib; lenb = kChunkBits; b_own = false; if (ib) pb = C2A_CPREV(b); C2A_HOP(); }
    while (ia > 0 && a != b) {
        if (pa == pb) break;
        below_a = a; below_b = b;
        a = pa; b = pb; --ia;
        lena = lenb = kChunkBits; a_own = b_own = false;
        if (ia) { pa = C2A_CPREV(a); pb = C2A_CPREV(b); }
        C2A_HOP();
    }
#undef C2A_HOP
    if (a == b) {      // one node is the chunk-boundary ancestor of the other: the other's next label decides
        if (below_a != C2A_NONE) return C2A_BIT0(below_a) < lb;
        return la < C2A_BIT0(below_b);
    }
    const u64 wa = a_own ? wa_in : ld_rec_wait(node_base, epoch, ctl, a, lane);
    const u64 wb = b_own ? wb_in : ld_rec_wait(node_base, epoch, ctl, b, lane);
    return str_less_len(wa, lena, la, wb, lenb, lb, lane);
#undef C2A_CPREV
#undef C2A_BIT0
}

// ------------------------------------------------------------------------------------------------
// static per-gate data of the dataflow launch (after the consumer counts have been scanned):
// gstat and the consumer lists (eslot[2g + l] = index of edge (g, l) in its producer's list, from k_deps)
// ------------------------------------------------------------------------------------------------
// The relays above relay `idx` of level `lvl_off` (HUBS AND RELAYS): written by ONE thread — the one that holds the first edge
// of the bottom relay's segment goes on upwards while the relay it just wrote is the first of ITS parent's segment.
// Hub h: N consumers at clist[off_h ...), relays n + b ...; xbase: where entry i = relay i starts in clist.
__device__ __attribute__((noinline)) void write_relays(u32 n, u32 h, u32 N, u32 off_h, u32 j, uint4* gstat, u32* fill, uint2* child) {
    const u32 xbase = 2u * n + 64u, b = relay_base(off_h);
    u32 cnt_lvl = relay_ceil(N), lvl_off = 0, idx = j;
    u32 list_off = off_h + j * kRelayFan, list_cnt = N - j * kRelayFan < kRelayFan ? N - j * kRelayFan : kRelayFan;
    for (;;) {
        const u64 r = (u64)n + b + lvl_off + idx;
        const bool top = cnt_lvl <= kHubMin;
        const u32 pj = idx / kRelayFan;
        const u32 parent = top ? h : n + b + lvl_off + cnt_lvl + pj;
        const u32 p_off = xbase + b + lvl_off + (top ? 0u : pj * kRelayFan);
        const u32 p_cnt = top ? cnt_lvl : (cnt_lvl - pj * kRelayFan < kRelayFan ? cnt_lvl - pj * kRelayFan : kRelayFan);
        gstat[2 * r] = make_uint4(parent, C2A_NONE, kRelayOrig, list_cnt);
        gstat[2 * r + 1] = make_uint4(p_off, p_cnt, list_off, 0u);      // (word 2: the relay's OWN list — it has no second producer, and any offset into clist will do there)
        fill[r] = 0u;
        child[r] = make_uint2(C2A_NONE, C2A_NONE);
        if (top || idx % kRelayFan != 0u) break;
        list_off = p_off; list_cnt = p_cnt; lvl_off += cnt_lvl; idx = pj; cnt_lvl = relay_ceil(cnt_lvl);
    }
}
// the edge (consumer, slot `slot` of the list at off, cnt) of a HUB goes to the relay that owns the slot: {relay id, where that relay's list starts, its length}
struct RelayEdge { u32 id, off, cnt; };
__device__ __forceinline__ RelayEdge relay_edge(u32 n, u32 off, u32 cnt, u32 slot) {
    const u32 j = slot / kRelayFan;
    return RelayEdge{n + relay_base(off) + j, off + j * kRelayFan, cnt - j * kRelayFan < kRelayFan ? cnt - j * kRelayFan : kRelayFan};
}
__global__ void __launch_bounds__(256) k_gstat(u32 n, const u32* __restrict__ dep0, const u32* __restrict__ dep1, const u32* __restrict__ cons_off,
                        const uint2* __restrict__ eslot, const u32* __restrict__ orig, uint4* gstat, u32* clist, u32* fill, uint2* child, u32* relay_total) {
    // (the consumer count of a gate is the difference of two neighbouring offsets — cons_off has n + 1 entries —: ONE
    // 8-byte access per producer instead of two 4-byte ones in two arrays.  Word 2 of a gate's first record is its ORIGINAL
    // id — what the DFS roots are compared by, topological_sort.rs:11-13; the launch works in rank space, c2a_kernels.h
    // RELABELLING —; the offset of its own consumer list is only wanted off the hot path: own_list_off.
    // A producer with more than kHubMin consumers is a HUB: the edge goes to the relay that owns its slot of the hub's list, the
    // hub's own list are its top-level relays (relay_base: which relays a hub gets follows from where its list lies); the
    // thread that holds the first edge of a relay's segment writes the relay's static records.  Everything of a gate is
    // LOADED before any of it is looked at — one memory round trip per gate, as without hubs —, the hubs are ONE unlikely branch.
    // *relay_total: relays of all hubs (the launch has run n + that many steps when nothing is left behind): summed per
    // workgroup — one atomic per hub on that word would be 0.6 ms of same-address atomics for 50 000 hubs)
    __shared__ u32 s_relays;
    if (threadIdx.x == 0) s_relays = 0u;
    __syncthreads();
    u32 my_relays = 0;
    const XcdSweep R = xcd_sweep(n);
    for (u64 g = R.i; g < R.end; g += R.step) {
        u32 d0 = dep0[g], d1 = dep1[g];
        const uint2 es = eslot[g];
        const u32 o = cons_off[g];
        u32 own_cnt = cons_off[g + 1] - o;
        uint4 g2 = make_uint4(0, 0, 0, 0);
        if (d0 != C2A_NONE) { g2.x = cons_off[d0]; g2.y = cons_off[(u64)d0 + 1] - g2.x; }
        if (d1 != C2A_NONE) { g2.z = cons_off[d1]; g2.w = cons_off[(u64)d1 + 1] - g2.z; }
        if (d0 != C2A_NONE) clist[g2.x + es.x] = (u32)g;
        if (d1 != C2A_NONE) clist[g2.z + es.y] = (u32)g | 0x80000000u;
        if (C2A_UNLIKELY(own_cnt > kHubMin || g2.y > kHubMin || g2.w > kHubMin)) {
            if (own_cnt > kHubMin) { my_relays += relay_count(own_cnt); own_cnt = relay_top(own_cnt).cnt; }
            if (g2.y > kHubMin) {
                if (es.x % kRelayFan == 0u) write_relays(n, d0, g2.y, g2.x, es.x / kRelayFan, gstat, fill, child);
                const RelayEdge E = relay_edge(n, g2.x, g2.y, es.x);
                d0 = E.id; g2.x = E.off; g2.y = E.cnt;
            }
            if (g2.w > kHubMin) {
                if (es.y % kRelayFan == 0u) write_relays(n, d1, g2.w, g2.z, es.y / kRelayFan, gstat, fill, child);
                const RelayEdge E = relay_edge(n, g2.z, g2.w, es.y);
                d1 = E.id; g2.z = E.off; g2.w = E.cnt;
            }
        }
        gstat[2 * g] = make_uint4(d0, d1, orig[g], own_cnt);
        gstat[2 * g + 1] = g2;
    }
    if (my_relays) atomicAdd(&s_relays, my_relays);
    __syncthreads();
    if (threadIdx.x == 0 && s_relays) atomicAdd(relay_total, s_relays);
}
// entry i of the relay part of clist = relay i, edge label 0 (the consumer lists of relays' parents are runs of it): once per loaded graph
__global__ void k_relay_list(u32 n, u32 cap, u32* clist) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < cap; i += (u64)gridDim.x * blockDim.x) clist[2ull * n + 64ull + i] = n + (u32)i;
}
// The relays leave the DFS tree where it is READ (k_euler_next, behind the launch): what hangs below (c, l) is the hub at the top
// of the relay chain that chose that edge — if every relay above chose the one below it —, and a hub's parent is the consumer its
// bottom relay chose, over that relay's edge label.
__device__ __forceinline__ u32 tree_child(u32 n, const u32* __restrict__ child, u32 c) {
    while (C2A_UNLIKELY(c != C2A_NONE && c >= n)) c = child[2 * (u64)c];      // (a relay's one child hangs at label 0)
    return c;
}
struct TreeParent { u32 p, label; };
__device__ __forceinline__ TreeParent tree_parent(u32 n, const uint4* __restrict__ meta, uint4 m) {
    while (C2A_UNLIKELY(m.x != C2A_NONE && m.x >= n)) m = meta[m.x];
    return TreeParent{m.x, m.w & 1u};
}

// ------------------------------------------------------------------------------------------------
// sinks (gates nobody consumes: DFS roots of depth 0, no candidates) — a plain grid-stride pass; the producers they
// claim seed the dataflow launch.  No shared counter: workgroup b appends to its own region under its own counter.
// A sink writes the three header words of its record only (its string is empty; readers treat depth 0 so).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_peel_sinks(PeelArgs A, PeelCold* cold_dst, PeelCold cold) {
    __shared__ u32 s_done[4];
    // (what only the edges of the dataflow launch touch travels as one small block in HBM — it keeps that kernel's scalar registers free —,
    // written here, two launches ahead of its first reader, from this launch's argument block: a launch of its own in round 5)
    if (blockIdx.x == 0 && threadIdx.x == 0) *cold_dst = cold;
    const u32 lane = threadIdx.x & 63u;
    u32* out = A.seeds_w + (u64)blockIdx.x * A.region_cap;
    u32* counter = &A.seed_cnt_w[blockIdx.x];
    const u64 lt_mask = (1ull << lane) - 1ull;
    const u64 tag = (u64)A.epoch << kTagShift;
    u32 done = 0;
    for (u64 base = (u64)blockIdx.x * 256; base < A.n; base += (u64)gridDim.x * 256) {
        const u64 g = base + threadIdx.x;
        u32 rdy[2] = {C2A_NONE, C2A_NONE};
        if (g < A.n) {
            const uint4 gi = A.gstat[2 * g];
            if (gi.w == 0) {
                ++done;
                A.meta[g] = make_uint4(C2A_NONE, 0u, gi.z, 0u);
                A.node[g * kNodeWords] = tag | hdr0_word(gi.z, 0u);
                A.node[g * kNodeWords + 1] = tag | hdr1_word(0u, C2A_NONE);
                A.node[g * kNodeWords + 2] = tag | ((u64)((kHdrWords << 8) | (kWordBits - 1u)) << 32) | gi.z;
                const uint4 g2 = A.gstat[2 * g + 1];
                const u32 deps[2] = {gi.x, gi.y}, cnts[2] = {g2.y, g2.w};
#pragma unroll
                for (u32 l = 0; l < 2; ++l) {
                    const u32 d = deps[l];
                    if (d == C2A_NONE) continue;
                    if (cnts[l] == 1u || atomicAdd(&A.fill[d], 1u) + 1u == cnts[l]) rdy[l] = d;
                }
            }
        }
#pragma unroll
        for (u32 l = 0; l < 2; ++l) {
            const u64 mask = __ballot(rdy[l] != C2A_NONE);
            if (mask) {
                u32 b = 0;
                if (lane == (u32)ctz64(mask)) b = atomicAdd(counter, (u32)__popcll(mask));      // this workgroup's own counter
                b = __shfl(b, (int)ctz64(mask), 64);
                if (rdy[l] != C2A_NONE) out[b + (u32)__popcll(mask & lt_mask)] = rdy[l];
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) done += __shfl_xor(done, off, 64);
    if (lane == 0) s_done[threadIdx.x >> 6] = done;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 t = s_done[0] + s_done[1] + s_done[2] + s_done[3];
        if (t) atomicAdd(&A.ctl[A.proc_word + (blockIdx.x & A.proc_mask) * kAcctStride], t);
    }
}

// ---- the first few levels behind the sinks, a whole level at once.  Level 1 (every consumer is a sink: 450 000 gates of the
// headline graph), 2 (175 000), 3 (73 000), 4 (33 000) ... are far WIDER than the body of the graph (2 000 per level): in the
// dataflow launch each of their gates would start a chain — three dependent round trips before its first step — and those
// starts kept all 2 048 waves busy for the launch's first 0.5 ms while the hand-off entries of that time, the critical path
// among them, waited for its end (tools/peel_trace.py: the path began 0.54 ms into the launch).  Down here the paths are
// short — a gate of level L is at most L deep, its string fits ONE word — so a LANE decides a gate: its key is (DFS root,
// string with the edge label appended), the smallest key among its consumers wins (stored most significant first, the first bit of
// a string is the most significant: integer order = lexicographic order; neither path is a prefix of the other, so zero
// padding decides nothing), if that root is smaller than the gate's own id (else the gate is a DFS root itself).  A wave
// per gate then writes the 64-word record.  Workgroup b takes region b of the pass before, collects the producers it claims
// in ITS region of `out` (no shared counter; a region that is full sends the rest straight to the flat list: any claimed
// gate may start a chain at any time), and the LAST pass moves its regions to one flat list (a single reservation per
// workgroup): the seeds of the dataflow launch, which its waves take a few at a time.
__global__ void __launch_bounds__(256) k_peel_shallow(PeelArgs A, const u32* __restrict__ cons_off, u32 lvl, u32 last, const u32* __restrict__ in, const u32* __restrict__ in_cnt, u32 in_cap,
                                                      u32* out, u32* out_cnt, u32 out_cap, u32* flat, u32* flat_total) {
    // what a lane decided about its gate, for the wave that writes the records
    __shared__ u32 s_g[256], s_root[256], s_depth[256];       // gate (C2A_NONE: none), DFS root, depth
    __shared__ u64 s_str[256];                                // its string (one word: depth <= lvl)
    __shared__ u32 s_base, s_cnt;
    const u32 lane = threadIdx.x & 63u;
    const u32 cnt = in_cnt[blockIdx.x] < in_cap ? in_cnt[blockIdx.x] : in_cap;      // (what did not fit went to the flat list)
    const u32* src = in + (u64)blockIdx.x * in_cap;
    u32* dst = out + (u64)blockIdx.x * out_cap;
    u32* counter = &out_cnt[blockIdx.x];
    const u64 tag = (u64)A.epoch << kTagShift;
    const u64 lt_mask = (1ull << lane) - 1ull;
    for (u32 base = 0; base < cnt; base += 256) {
        // ---- a LANE per gate: the smallest key, the tree entry, the tickets of its producers (every load of a gate waits
        // for the one before: 256 gates wait together)
        const u32 i = base + threadIdx.x;
        u32 g = C2A_NONE, rdy[2] = {C2A_NONE, C2A_NONE};
        uint4 gi = make_uint4(0u, 0u, 0u, 0u);
        if (i < cnt) {
            g = src[i];
            gi = A.gstat[2 * (u64)g];
            // a RELAY (c2a_peel.h HUBS AND RELAYS) is not for these passes — its level does not count and its string need not fit
            // a word —: it starts a chain of the dataflow launch (and is counted there)
        }
        {
            // (one reservation per wave — a hub of a million sinks has tens of thousands of relays claimed by the sinks pass —;
            // the workgroup reports its whole region as done below: minus these)
            const bool relay = i < cnt && gi.z == kRelayOrig;
            const u64 rm = __ballot(relay);
            if (rm) {
                u32 b = 0;
                if (lane == (u32)ctz64(rm)) {
                    b = atomicAdd(flat_total, (u32)__popcll(rm));
                    atomicAdd(&A.ctl[CTL_PROC + (blockIdx.x & (kAcctShards - 1u)) * kAcctStride], 0u - (u32)__popcll(rm));
                }
                b = __shfl(b, (int)ctz64(rm), 64);
                if (relay) { flat[b + (u32)__popcll(rm & lt_mask)] = g; g = kShallowSkip; }
            }
        }
        if (i < cnt && g != kShallowSkip) {
            const uint4 gi2 = A.gstat[2 * (u64)g + 1];
            const u32 g_off = cons_off[g], g_orig = gi.z;
            u32 b_root = C2A_NONE, b_c = C2A_NONE, b_el = 0, b_depth = 0;
            u64 b_x = 0;
            // (records of the passes before: plain loads; a sink's record has header words only — its string is empty.  The
            // loads of up to eight consumers go out TOGETHER, list entries, then header words, then strings: three round
            // trips per gate instead of three per consumer)
            auto meet = [&](u32 e, u64 h0w, u64 sw) {
                const u32 c = e & kIdMask, el = e >> 31;
                const u64 h0 = h0w & kPayload;
                const u32 croot = hdr_hi(h0), cdepth = (u32)h0;
                const u64 x = (cdepth ? (sw & kPayload) : 0ull) | ((u64)el << (kWordBits - 1u - cdepth));
                if (croot < b_root || (croot == b_root && x < b_x)) { b_root = croot; b_c = c; b_el = el; b_depth = cdepth; b_x = x; }
            };
            constexpr u32 kTogether = 8;
            u32 ee[kTogether]; u64 hh[kTogether], ss[kTogether];
#pragma unroll
            for (u32 k = 0; k < kTogether; ++k) ee[k] = k < gi.w ? A.clist[g_off + k] : 0u;
#pragma unroll
            for (u32 k = 0; k < kTogether; ++k) hh[k] = k < gi.w ? A.node[(u64)(ee[k] & kIdMask) * kNodeWords] : 0ull;
#pragma unroll
            for (u32 k = 0; k < kTogether; ++k) ss[k] = k < gi.w ? A.node[(u64)(ee[k] & kIdMask) * kNodeWords + kHdrWords] : 0ull;
#pragma unroll
            for (u32 k = 0; k < kTogether; ++k) if (k < gi.w) meet(ee[k], hh[k], ss[k]);
            for (u32 e_i = kTogether; e_i < gi.w; ++e_i) {
                const u32 e = A.clist[g_off + e_i];
                meet(e, A.node[(u64)(e & kIdMask) * kNodeWords], A.node[(u64)(e & kIdMask) * kNodeWords + kHdrWords]);
            }
            const bool has = b_root < g_orig;        // (else every consumer belongs to a later DFS root: [g] itself)
            const u32 ch = has ? b_c : C2A_NONE, depth = has ? b_depth + 1u : 0u, root = has ? b_root : g_orig, label = has ? b_el : 0u;
            A.meta[g] = make_uint4(ch, depth, root, label | (lvl << 1));
            if (has) A.child[2 * (u64)ch + label] = g;
            s_root[threadIdx.x] = root; s_depth[threadIdx.x] = depth; s_str[threadIdx.x] = has ? b_x : 0ull;
            const u32 deps[2] = {gi.x, gi.y}, cnts[2] = {gi2.y, gi2.w};
#pragma unroll
            for (u32 l = 0; l < 2; ++l) {
                const u32 d = deps[l];
                if (d != C2A_NONE && (cnts[l] == 1u || atomicAdd(&A.fill[d], 1u) + 1u == cnts[l])) rdy[l] = d;
            }
        }
        s_g[threadIdx.x] = g;
#pragma unroll
        for (u32 l = 0; l < 2; ++l) {
            const u64 mask = __ballot(rdy[l] != C2A_NONE);
            if (mask) {
                u32 b = 0;
                if (lane == (u32)ctz64(mask)) b = atomicAdd(counter, (u32)__popcll(mask));      // this workgroup's own counter
                b = __shfl(b, (int)ctz64(mask), 64);
                if (rdy[l] != C2A_NONE) {
                    const u32 at = b + (u32)__popcll(mask & lt_mask);
                    if (at < out_cap) dst[at] = rdy[l];
                    else flat[atomicAdd(flat_total, 1u)] = rdy[l];          // (the region is full: a seed of the launch right away)
                }
            }
        }
        __syncthreads();
        // ---- sixteen lanes per record: its first LINE (header + 13 string words; readers take the rest for zero padding:
        // needed_lanes) — a quarter of the bytes of a whole record, and these passes are bound by what they write
        for (u32 j = threadIdx.x >> 4; j < 256; j += 16) {
            const u32 gj = s_g[j];
            if (gj == C2A_NONE) break;               // (the gates of a batch are its first lanes)
            if (gj == kShallowSkip) continue;        // (a relay: passed on to the launch)
            const u32 root = s_root[j], depth = s_depth[j], l16 = threadIdx.x & 15u;
            u64 w = tag;
            if (l16 == 0) w |= hdr0_word(root, depth);
            else if (l16 == 1) w |= hdr1_word(lvl, C2A_NONE);
            else if (l16 == 2) w |= ((u64)((kHdrWords << 8) | (kWordBits - 1u - depth)) << 32) | root;      // (where a child's label goes: string word 0, `depth` bits down — depth < 62 here —; the root key)
            else if (l16 == kHdrWords) w |= s_str[j];
            A.node[(u64)gj * kNodeWords + l16] = w;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && cnt) {
        u32* acct = &A.ctl[CTL_PROC + (blockIdx.x & (kAcctShards - 1u)) * kAcctStride];
        atomicAdd(acct, cnt); atomicMax(acct + 1, lvl);
    }
    if (!last) return;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 c0 = atomicAdd(counter, 0u);   // (the waves counted by atomics: read it the same way)
        s_cnt = c0 < out_cap ? c0 : out_cap;
        s_base = s_cnt ? atomicAdd(flat_total, s_cnt) : 0u;
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < s_cnt; i += 256) flat[s_base + i] = __hip_atomic_load(&dst[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}


// everything issued for one gate at the top of its step; two of these swap roles (nothing is ever copied: a register
// copy of a value still in flight is a use, and its wait would drain the step that was just issued)
struct StepIO {
    u32 cnt0, cnt1;            // consumers of producer 0 / 1 (0: no such producer); the tickets taken on them are in the
                               // fixed scalar registers of this set (SCALAR TICKETS above)
    u32 gw;                    // lane 8 l + j (l < 2, j < 8): word j of the two gstat records of producer l
    u32 clp;                   // lanes 0..31: producer 0's consumers, 32..63: producer 1's
    u32 e0, e1, e2, e3, e4, e5, e6, e7, take;  // consumer | label << 31 of the (up to six) records in flight
    u32 more;                  // the consumer list holds candidates beyond those (the step reads the list itself: cold)
    u64 w0, w1, w2, w3, w4, w5, w6, w7;        // this lane's word of those records
};

// the dataflow launch: 64-thread workgroups (one wave each)
template <int SET> struct StepSet { static constexpr int value = SET; };
// DEEP: the DFS tree may be deeper than one chunk of path string (kChunkBits levels).  The launch WITHOUT it leaves the
// chunk-boundary code out of the hot path (a compare and a branch per candidate, one more per step) and raises CTL_NEEDDEEP +
// ABORT when a node fills its chunk: the host then runs the DEEP launch on clean buffers and remembers it for the loaded graph.
template <bool STATS, bool DEEP>
__global__ void __launch_bounds__(64) C2A_PEEL_KERNEL_ATTR k_peel(PeelArgs A_in) {
    SRegs sr;
    PeelArgs A = A_in;
    A.gstat = own_sgprs(A_in.gstat); A.clist = own_sgprs(A_in.clist); A.node = own_sgprs(A_in.node); A.fill = own_sgprs(A_in.fill);
    A.meta = own_sgprs(A_in.meta); A.child = own_sgprs(A_in.child); A.fifo = own_sgprs(A_in.fifo); A.q_pc = own_sgprs(A_in.q_pc);
    A.ctl = own_sgprs(A_in.ctl); A.cold = own_sgprs(A_in.cold);
    A.epoch = own_sgpr(A_in.epoch); A.pad_thr = own_sgpr(A_in.pad_thr); A.n_fifos = own_sgpr(A_in.n_fifos); A.q_cap = own_sgpr(A_in.q_cap); A.run = own_sgpr(A_in.run); A.n_primary = own_sgpr(A_in.n_primary); A.reserve_min = own_sgpr(A_in.reserve_min);
    const u32 lane = threadIdx.x;
    const u32 me = blockIdx.x;
    // this wave is one unit of work from now until it first runs out of work; BEGIN counts it before it moves
    {
        u32 r = 0;
        if (lane == 0) r = atomicAdd(&A.ctl[CTL_BEGIN + (me & (kAcctShards - 1u)) * kAcctStride], 1u);
        C2A_PIN(r);
    }
    const u32 epoch = A.epoch;
    const TagCheck tagc = tag_check(epoch, A.pad_thr, lane);
    const u32 cmp_hi_mask = lane == 2u ? 0u : kHdrMask;      // a record word -> its compare word: the payload; lane 2: the root key alone
    // byte offset of the word of a hand-off entry that this lane writes: lanes 8..15 words 0..7 (the pushed gate's static
    // records), lanes 32..39 words 8..15 (its first consumers, its id)
    const u32 ent_off = (lane >= 8u && lane < 16u) ? (lane - 8u) * 8u : ((lane >= 32u && lane < 40u) ? (lane - 24u) * 8u : C2A_NONE);
    bool parked = me >= A.n_primary;         // THE RESERVE (below): this wave stays out — of the seeds too — until the launch is short of waves
    const u32 dummy_idx = ((A.n_all + 3u) & ~3u) + me * kFillDummyStride; (void)dummy_idx;      // this wave's dummy ticket word (a line of its own behind fill[n_all], from a 16-byte boundary on)
    bool seeds_left = !parked;
    u32 region = 0, idx = 0, region_cnt = 0;
    u32 push_rr = me, pop_rr = me * 7u;      // round-robin cursors over the hand-off arrays
    u32 held = 0;                            // this wave holds a consumer ticket that has not been served yet ...
    u64 held_slot = 0;                       // ... for this slot
    u32 processed = 0, max_level = 0, iters = 0;
    u32 st_pops = 0, st_polls = 0, st_push = 0, st_seeds = 0;
    ull st_busy = 0, st_idle = 0, st_t0 = STATS ? c2a_now() : 0;
    ull ph_w1 = 0, ph_w2 = 0, ph_w3 = 0;
    // STATS: the diagnostic arrays, read once (a load through A.cold inside a step would put its round trip into the step)
    ull* const dq_time = STATS ? uniform_ptr(A.cold->q_time) : nullptr;
    ull* const dp_time = STATS ? uniform_ptr(A.cold->p_time) : nullptr;
    ull* const dt_trace = STATS ? uniform_ptr(A.cold->t_trace) : nullptr;
    ull ph_a = 0, ph_b = 0, ph_c = 0, ph_d = 0, ph_steps = 0, ph_noload = 0, ph_start = 0, ph_push = 0, ph_cold = 0, ph_pwait = 0;      // STATS: phase times of the chain step
    for (;;) {
        // ---- next piece of work: the seed pool, then the hand-off slots
        u32 g = C2A_NONE;
        u32 came = 0;                        // STATS: 1 popped from a hand-off array, 2 a seed (0: chain step)
        uint4 gi, gi2;
        u32 cl0 = 0, cl0_base = 0, cl0_cap = 64;       // g's consumers: one per lane from lane cl0_base on, cl0_cap lanes
        ull pop_slot = ~0ull;                // STATS: slot of the popped entry
        if (C2A_UNLIKELY(seeds_left)) {
            // (pointers read from memory are generic pointers, and a load through one counts as divergent: every value
            // read through A.cold is declared wave-uniform by hand)
            const PeelCold* C = A.cold;
            const u32 chunk = uniform(C->seed_chunk);
            while (idx >= region_cnt) {
                u32 r = 0;
                if (lane == 0) r = atomicAdd(&A.ctl[CTL_SEEDNEXT], 1u); wave_join();
                r = rdlane(r, 0);
                const u32 total = uniform(*C->seed_total);
                if ((u64)r * chunk >= total) { seeds_left = false; break; }
                region = r; idx = 0; region_cnt = total - r * chunk < chunk ? total - r * chunk : chunk;
            }
            if (seeds_left) { g = uniform(C->seeds[(u64)region * chunk + idx]); ++idx; if (STATS) { ++st_seeds; came = 2; } }
        }
        const ull ph_s0 = STATS ? c2a_now() : 0;
        if (C2A_UNLIKELY(g != C2A_NONE)) {
            // static data of a seed (a chain step gets all of this prefetched by the step before; a popped gate brings it
            // along); consumed HERE, in scalar registers where wave-uniform: a load still pending at the loop header would
            // cost every chain step a wait
            const u32 g_off = uniform(own_list_off(uniform(A.cold->n), A.cold->cons_off, A.gstat, g));
            gi = uniform4(A.gstat[2 * (u64)g]);
            gi2 = uniform4(A.gstat[2 * (u64)g + 1]);
            cl0 = A.clist[g_off + lane];                                     // clist is padded by 64 entries
            C2A_PIN(cl0);
        } else {
            wave_priority(0);
            if (STATS) { const ull t = c2a_now(); st_busy += t - st_t0; st_t0 = t; }
            // ---- this unit of work is over: END counts it (every entry this wave pushed meanwhile is in BEGIN already:
            // those were returning atomics, waited for)
            if (lane == 0) atomicAdd(&A.ctl[CTL_END + (me & (kAcctShards - 1u)) * kAcctStride], 1u); wave_join();
            // ---- a consumer ticket — kept until it is served, whatever else this wave does meanwhile — and the slot it names
            if (C2A_UNLIKELY(parked)) {
                // THE RESERVE.  Eight waves per CU are what a graph a few thousand gates wide can use: every further wave in the lines
                // makes the others' looks and the ticket lines slower (headline + 1.6 %, strict layers + 9 % with sixteen).  A graph
                // tens of thousands of gates wide — a matrix product's reduction chains, a butterfly, the leaves of a Merkle tree — is
                // short of WAVES, not of latency (2 048 waves x a gate per microsecond): there sixteen per CU are worth 20-35 %.  Which
                // of the two a launch is shows in the hand-off arrays: entries pushed and nobody in line for them.  A parked wave looks
                // at a flag every ~7 us (one of 64 copies: 2 048 waves on ONE line cost the headline 2 %); one in C2A_PARK_MONITORS adds
                // up the arrays' backlog every ~7 us — 64 lines that every push and pop hits with an atomic: with one monitor in 32 the
                // headline's launch took 8.5 ms instead of 5.9 — and raises the flag at reserve_min entries; a wave that has seen the
                // flag is an ordinary wave from then on.  Ending, and picking up a stranded entry, never depends
                // on a parked wave: those are the waiting waves' scans.
                bool leave = false;
                const bool monitor = ((me - A.n_primary) & (C2A_PARK_MONITORS - 1u)) == 0;
                for (u32 sb = 0;; ++sb) {
                    // (the word this wave looks at is one of kAcctShards copies, each on a line of its own: 1 = come in, 2 = the launch
                    // is over; whoever raises either writes all copies with one store.  ABORT — an atomic — is looked at now and then)
                    u32 c3 = 0;
                    if (lane == 0 || (lane < 3 && (sb & 7u) == 0)) c3 = ld_a32(&A.ctl[lane == 0 ? CTL_PARK + (me & (kAcctShards - 1u)) * kAcctStride : (lane == 1 ? CTL_ABORT : CTL_DONE)]);
                    wave_join();
                    if (rdlane(c3, 0) == 2u || rdlane(c3, 1) | rdlane(c3, 2)) { leave = true; break; }
                    if (rdlane(c3, 0)) break;
                    if (monitor) {
                        const u64 pcw = lane < A.n_fifos ? ld_nw(&A.q_pc[(u64)lane * kPcStride]) : 0ull;
                        u32 back = (u32)pcw - (u32)(pcw >> 32);                  // pushed - taken (tickets of waves in line count as taken)
                        back = (back & 0x80000000u) ? 0u : back;
#pragma unroll
                        for (int off = 32; off >= 1; off >>= 1) back += __shfl_xor(back, off, 64);
                        if (uniform(back) >= A.reserve_min) { st_a32(&A.ctl[CTL_PARK + lane * kAcctStride], 1u); wave_join(); break; }
                    }
                    peel_sleep(127); peel_sleep(127);
                }
                if (leave) break;
                // (into a line first: an entry it receives is a unit of work that BEGIN has counted; the seed pool after that, as part of it)
                parked = false; seeds_left = true;
            }
            if (C2A_LIKELY(!held)) {
                const u32 f = (pop_rr++) & (A.n_fifos - 1u);
                // (a scalar ticket here too — nothing else is in flight in those registers while a wave has no work)
                sreg_add64<kSregPush>(sr, &A.q_pc[(u64)f * kPcStride], 1ull << 32);
                sreg_wait();
                held_slot = (u64)f * A.q_cap + sreg_get<kSregPush + 1>(sr);
                held = 1;
            }
            u64 slot_i = held_slot;          // the slot this wave watches: its own, or (see below) one that is served already
            u32 polls = 0, hb_seen = 0, hb_checks = 0;
            ull hb_t0 = c2a_now(); (void)hb_t0;
            u64 v = 0;
            bool got = false;
            for (;;) {
                v = ld_nw(A.fifo + slot_i * kSlotWords + (lane & (kSlotWords - 1u)));
                if (((u32)__ballot((u32)(v >> 32) == A.run) & 0xFFFFu) == 0xFFFFu) { got = true; break; }
                ++polls;
                if (C2A_UNLIKELY((polls & 31u) == 0)) {
                    // cheap and frequent: has somebody seen the end (or given up)?  is the launch still making progress?
                    u32 c3 = 0;
                    if (lane < 3) c3 = ld_a32(&A.ctl[lane == 0 ? CTL_ABORT : (lane == 1 ? CTL_DONE : CTL_HEARTBEAT)]); wave_join();
                    const u32 aborted = rdlane(c3, 0), done = rdlane(c3, 1), hb = rdlane(c3, 2);
                    if (aborted || done) break;
                    bool starved;
#ifdef C2A_EMULATE
                    starved = hb == hb_seen && ++hb_checks > kWatchdogChecks;
#else
                    const ull t_now = c2a_now();
                    starved = hb == hb_seen && t_now - hb_t0 > kWatchdogTicks;
                    if (hb != hb_seen) hb_t0 = t_now;
#endif
                    if (hb != hb_seen) { hb_seen = hb; hb_checks = 0; }
                    else if (starved) { if (lane == 0) atomicAdd(&A.ctl[CTL_ABORT], 1u); wave_join(); break; }
                    // The full look at the counters (192 lines that every pusher writes) is for one wave in 64 at this rate —
                    // with every waiting wave doing it, those reads alone were 0.4-1.5 TB/s on the lines the tickets live
                    // on — and for every wave once in 1024 polls, so that ending (and picking up a stranded entry) never
                    // depends on a particular wave being resident
                    if (((me & 63u) == 0 && (polls & C2A_SCAN_MASK) == 0) || (polls & 16383u) == 0) {
                        // every END, then every BEGIN: each read waits for the one before
                        u32 n_end = lane < kAcctShards ? ld_a32(&A.ctl[CTL_END + lane * kAcctStride]) : 0u;
#pragma unroll
                        for (int off = 32; off >= 1; off >>= 1) n_end += __shfl_xor(n_end, off, 64);
                        n_end = uniform(n_end);      // (a shuffle result counts as divergent: the branches below must not)
                        // (BEGIN = the waves that started + every entry ever pushed; the latter IS the producer side of the hand-off
                        // arrays' ticket words — a pusher's ticket is a returning atomic it waits for before its entry can be seen —,
                        // so a push counts its unit with the atomic it needs anyway)
                        u32 n_begin = (lane < kAcctShards ? ld_a32(&A.ctl[CTL_BEGIN + lane * kAcctStride]) : 0u) +
                                      (lane < A.n_fifos ? (u32)ld_nw(&A.q_pc[(u64)lane * kPcStride]) : 0u);
#pragma unroll
                        for (int off = 32; off >= 1; off >>= 1) n_begin += __shfl_xor(n_begin, off, 64);
                        n_begin = uniform(n_begin);
                        if (n_end == n_begin) {      // nothing active, nothing in flight: the peel is over — tell everybody
                            if (lane == 0) st_a32(&A.ctl[CTL_DONE], 1u);
                            st_a32(&A.ctl[CTL_PARK + lane * kAcctStride], 2u);      // (the parked waves' copies: THE RESERVE)
                            wave_join();
                            break;
                        }
                        // An entry in an array where nobody waits (every wave is committed to a slot elsewhere) would sit
                        // there for good: a wave that sees such a backlog takes a ticket that is served ALREADY (compare-and-
                        // swap on the pair of counts) and works on that entry; its own ticket stays good for later
                        if (slot_i == held_slot) {
                            const u64 pcw = lane < A.n_fifos ? ld_nw(&A.q_pc[(u64)lane * kPcStride]) : 0ull;
                            const u64 bm = __ballot((u32)pcw > (u32)(pcw >> 32));
                            if (bm) {
                                const u32 r = me & 63u;
                                const u64 from_r = bm & (~0ull << r);
                                const u32 j = ctz64(from_r ? from_r : bm);
                                const u64 expect = rdlane64(pcw, j);
                                u64 seen = 0;
                                if (lane == 0) seen = atomicCAS(reinterpret_cast<ull*>(&A.q_pc[(u64)j * kPcStride]), (ull)expect, (ull)(expect + (1ull << 32))); wave_join();
                                if (rdlane64(seen, 0) == expect) slot_i = (u64)j * A.q_cap + (u32)(expect >> 32);
                            }
                        }
                    }
                }
                // back off: the longer nothing turns up, the less often this wave asks (64 clocks per unit)
                peel_sleep(polls < 16 ? 4 : (polls < 64 || C2A_POLL_CAP <= 16 ? (C2A_POLL_CAP < 16 ? C2A_POLL_CAP : 16) : C2A_POLL_CAP));
            }
            if (STATS) st_polls += polls;
            if (STATS) { const ull tt = c2a_now(); st_idle += tt - st_t0; st_t0 = tt; }
            if (C2A_UNLIKELY(!got)) break;
            if (slot_i == held_slot) held = 0;
            if (STATS) { pop_slot = slot_i; ++st_pops; came = 1; }
            const u32 pv = (u32)v;
            gi = make_uint4(rdlane(pv, 0), rdlane(pv, 1), rdlane(pv, 2), rdlane(pv, 3));
            gi2 = make_uint4(rdlane(pv, 4), rdlane(pv, 5), rdlane(pv, 6), rdlane(pv, 7));
            g = rdlane(pv, 15);
            // its consumers came along (lanes 8..14) unless it has more than seven — a relay with all its eight, a hub's top level:
            // then the list itself is loaded HERE, one round trip before the step is issued (a relay's entry even carries where its
            // list starts: word 2 of its second static record), instead of leaving all of it to the list loop of the step
            cl0 = pv; cl0_base = 8; cl0_cap = kSlotCons;
            if (C2A_UNLIKELY(gi.w > kSlotCons)) {
                const u32 l_off = gi.z == kRelayOrig ? gi2.z : uniform(own_list_off(uniform(A.cold->n), A.cold->cons_off, A.gstat, g));
                cl0 = A.clist[l_off + lane];                                 // (clist is padded by 64 entries)
                C2A_PIN(cl0);
                cl0_base = 0; cl0_cap = 64;
            }
        }
        // (wave-uniform by construction — say so: one value the compiler takes for divergent here, and every value of the
        // chain loop that depends on the gate id moves to vector registers and is handled as divergent code)
        g = uniform(g);
        if (STATS) {
            const ull tn = c2a_now();
            ph_start += tn - ph_s0;
            if (pop_slot != ~0ull && lane == 0) dp_time[pop_slot] = tn;
        }

        // issue everything the step of a gate needs from memory.  scl: the gate's consumer list, one entry per lane from
        // lane cbase on, when it fits ccap lanes (else nothing is loaded ahead and the step reads the list itself)
        auto issue = [&](auto set, StepIO& S, u32 dep0, u32 dep1, u32 n_cons, u32 off0, u32 cnt0, u32 off1, u32 cnt1, u32 scl, u32 cbase, u32 ccap,
                         bool have_own, u32 own_id) {
            // (ONE pair of ticket registers: the step reads the pair at its top, before it issues the next step into it)
            constexpr int R0 = kSregFill0, R1 = kSregFill1; (void)set;
            S.cnt0 = cnt0; S.cnt1 = cnt1;                                // (0 where there is no producer: gstat holds 0 then)
            // a ticket is needed where there is a producer with other consumers (else the register says 0: with cnt 1 that
            // reads as "the last ticket", with cnt 0 as "nothing claimed")
            sfill_take<R0>(sr, A.fill, dep0, cnt0, dummy_idx);
            sfill_take<R1>(sr, A.fill, dep1, cnt1, dummy_idx);
            // static data of both producers, one word per lane, BRANCH-FREE (clamped index, result discarded where there
            // is nothing to load)
            {
                const u32 dw = (lane & 8u) ? dep1 : dep0;
                S.gw = reinterpret_cast<const u32*>(A.gstat)[8 * (u64)(dw != C2A_NONE ? dw : 0u) + (lane & 7u)];
            }
            {
                const u32 half = lane >> 5, i = lane & 31u;
                const u32 poff = half ? off1 : off0;
                S.clp = A.clist[poff + i];                               // (clist is padded by 64 entries: lanes beyond the list read
                                                                         // somebody else's entries, which nobody looks at)
            }
            // the records of the first two consumers that are not the gate in hand (slots that are not loaded stay
            // UNDEFINED on purpose: merging a loaded value with a constant is a register copy, a copy is a use, and its wait
            // would land right behind the load)
            // (a ballot per compare, the masks combined in scalar registers: the ballot of a combined condition is compiled as
            // select + compare on top of the compares)
            const bool in_lanes = n_cons <= ccap;
            u64 smask = __ballot(lane - cbase < n_cons) & (in_lanes ? ~0ull : 0ull);
            if (have_own) smask &= __ballot((scl & kIdMask) != own_id);
            S.take = 0;                  // (e0 / e1 stay undefined like w0 / w1: they are only looked at under take)
            if (C2A_LIKELY(smask != 0)) {
                S.e0 = rdlane(scl, pop_lowest(smask)); S.take = 1;
                S.w0 = ld_nw(&A.node[(u64)(S.e0 & kIdMask) * kNodeWords + lane]);
                if (smask) {
                    S.e1 = rdlane(scl, pop_lowest(smask)); S.take = 2;
                    S.w1 = ld_nw(&A.node[(u64)(S.e1 & kIdMask) * kNodeWords + lane]);
                    // (a third and a fourth: one gate in six has more than two other consumers, and the cold loop below costs it
                    // two dependent round trips per candidate)
                    if (C2A_UNLIKELY(smask != 0)) {
                        S.e2 = rdlane(scl, pop_lowest(smask)); S.take = 3;
                        S.w2 = ld_nw(&A.node[(u64)(S.e2 & kIdMask) * kNodeWords + lane]);
                        if (C2A_UNLIKELY(smask != 0)) {
                            S.e3 = rdlane(scl, pop_lowest(smask)); S.take = 4;
                            S.w3 = ld_nw(&A.node[(u64)(S.e3 & kIdMask) * kNodeWords + lane]);
                            // (a fifth and a sixth: one gate in 45 — but one step in 15 of the critical path — has more than four)
                            if (C2A_UNLIKELY(smask != 0)) {
                                S.e4 = rdlane(scl, pop_lowest(smask)); S.take = 5;
                                S.w4 = ld_nw(&A.node[(u64)(S.e4 & kIdMask) * kNodeWords + lane]);
                                if (smask) {
                                    S.e5 = rdlane(scl, pop_lowest(smask)); S.take = 6;
                                    S.w5 = ld_nw(&A.node[(u64)(S.e5 & kIdMask) * kNodeWords + lane]);
                                    if (smask) {
                                        S.e6 = rdlane(scl, pop_lowest(smask)); S.take = 7;
                                        S.w6 = ld_nw(&A.node[(u64)(S.e6 & kIdMask) * kNodeWords + lane]);
                                        if (smask) {
                                            S.e7 = rdlane(scl, pop_lowest(smask)); S.take = 8;
                                            S.w7 = ld_nw(&A.node[(u64)(S.e7 & kIdMask) * kNodeWords + lane]);
                                        }
                                    }
                                }
                            }
                        }
                    }
                }
            }
            S.more = (!in_lanes || smask != 0) ? 1u : 0u;
        };
        StepIO S0, S1;
        issue(StepSet<0>(), S0, gi.x, gi.y, gi.w, gi2.x, gi2.y, gi2.z, gi2.w, cl0, cl0_base, cl0_cap, false, 0u);
        // The champion of a gate's tournament so far (wave-uniform); ch == NONE: the virtual-root candidate [g].  The gate
        // just finished stays in these registers as the first candidate of the next one: it is the champion to beat unless
        // its DFS root is not smaller than the next gate's own id (then the step starts from [g])
        // The champion is kept as what a take has to copy and no more: its list entry (consumer | label << 31), where its child's
        // label goes, its record and its compare word; depth, DFS root and level are read out of the record ONCE, when the
        // tournament is over (per candidate that was two lane reads and three scalar instructions for values only the winner needs).
        u32 ch_e = C2A_NONE, ch_root = C2A_NONE, ch_pos = 0;      // (ch_pos: RAW, the high half of the record's word 2 — tag bits and all)      // (ch_root: the root of the gate just finished / of [g] itself)
        u64 ch_w = 0, ch_x = 0;              // the champion's record / its string with the edge label appended
        // the gate just finished (a consumer of the gate in hand); at the start of a chain there is none: no gate has id NONE,
        // level NONE + 1 is 0, and a champion root of NONE (above) is smaller than no gate id — no flag to test
        u32 own_node = C2A_NONE, own_lraw = 0;      // (own_lraw: its level as its record holds it — tag | level, the high half of word 1)

        // one step: `cur` is in hand (issued one step ago), `nx` receives the next one.  true = the chain ends (or abort)
        auto step = [&](auto cur_set, auto nx_set, StepIO& cur, StepIO& nx) -> bool {
            constexpr int RC0 = kSregFill0, RC1 = kSregFill1; (void)cur_set;
            const ull ph0 = STATS ? c2a_now() : 0;
            // (two waves share a SIMD's issue slots: from here to the issue of the next step — the claim path — this one goes first)
            wave_priority(3);
            // ---- everything of THIS step (issued one step ago) is needed now, and is pinned HERE: a register of `cur` that
            // the compiler still counts as in flight further down would put its wait behind the issue of the next step
            const u32 rmask = sfill_claims<RC0, RC1>(sr, cur.cnt0, cur.cnt1);
            const ull ph0a = STATS ? c2a_now() : 0;
            C2A_PIN(cur.gw);
            C2A_PIN(cur.clp);
            const ull ph0b = STATS ? c2a_now() : 0;
            C2A_PIN(cur.w0); C2A_PIN(cur.w1); C2A_PIN(cur.w2); C2A_PIN(cur.w3); C2A_PIN(cur.w4); C2A_PIN(cur.w5); C2A_PIN(cur.w6); C2A_PIN(cur.w7);
            if (STATS) { const ull ph0c = c2a_now(); ph_w1 += ph0a - ph0; ph_w2 += ph0b - ph0a; ph_w3 += ph0c - ph0b; }
            // (the next gate's static records are written over gi / gi2 below: what the rest of this step needs of its own)
            const u32 gc = g, g_dep0 = gi.x, g_dep1 = gi.y, g_orig = gi.z, g_cnt = gi.w;
            // producer l is claimed when its ticket was the last of cnt (no ticket was taken for a single-consumer producer:
            // 0 of 1; no producer: cnt 0 — the one comparison covers all three)
            const ull ph1 = STATS ? c2a_now() : 0;
            // ---- go on with the first claimed producer: issue its step now; a second one goes to whoever has no work: its
            // producer ticket is taken here and its entry stored after the tournament (the ticket is back by then)
            u32 nxt = C2A_NONE, nxt_label = 0;
            u32 push_t = 0, push_c = 0, push_f = 0;
            if (C2A_LIKELY(rmask != 0)) {
                const u32 j0 = (rmask & 1u) ^ 1u;
                nxt = j0 ? g_dep1 : g_dep0; nxt_label = j0;
                // lanes 0..7: the static records of nxt — the row shift always, then ONE select by a scalar mask (as a branch
                // around the shift this was eleven instructions of flags)
                const u32 gsel = select_uniform(j0, row_shl8(cur.gw), cur.gw);
                gi = make_uint4(rdlane(gsel, 0), rdlane(gsel, 1), rdlane(gsel, 2), rdlane(gsel, 3));
                // (a LOCAL: nothing of it is wanted behind the issue, and a value carried around the loop by reference is kept alive there)
                const uint4 ngi2 = make_uint4(rdlane(gsel, 4), rdlane(gsel, 5), rdlane(gsel, 6), rdlane(gsel, 7));
                if (C2A_LIKELY(rmask == 3u)) {      // (two steps in three of the critical path push: that path falls through)
                    if (STATS) ++st_push;
                    // one of the hand-off arrays: the ticket now, the entry after the tournament; BEGIN counts the entry
                    // before anybody can see it
                    push_f = (push_rr++) & (A.n_fifos - 1u);
                    sreg_inc64<kSregPush>(sr, &A.q_pc[(u64)push_f * kPcStride]);
                }
                // the consumers of nxt are already here (lanes 32 j0 ... of the prefetched lists) unless it has more than 32
                issue(nx_set, nx, gi.x, gi.y, gi.w, ngi2.x, ngi2.y, ngi2.z, ngi2.w, cur.clp, 32u * j0, 32u, true, gc);
            }
            wave_priority(C2A_PRIO_TOUR);
            const ull ph2 = STATS ? c2a_now() : 0;
            // ---- tournament of THIS gate
            // the reverse Kahn level = 1 + the highest level among the consumers: the maximum is taken over the RAW high halves of
            // their records' word 1 (tag | level: every record of a run carries the same tag), masked and bumped once at the end
            u32 lraw = own_lraw;
            if (!(ch_root < g_orig)) { ch_e = C2A_NONE; ch_root = g_orig; ch_pos = 0; ch_w = 0; ch_x = (u64)wrlane_c<2>(g_orig, 0u); }
            // one candidate: its record must be all there (else read it again: out of line), then it meets the champion
            auto candidate = [&](u64& w, u32 e) {             // (w by reference: the cold path mends it in place, no copy)
                const u32 el = e >> 31;
                u64 badm = __ballot(tag_bad(tagc, w));
                if (C2A_UNLIKELY(badm != 0)) {
                    badm = __ballot(tag_stale_or_never(epoch, w));
                    // not all there: a short record (a sink: header words only; a gate of the shallow passes: one line — what
                    // lies beyond the lanes its depth needs is zero padding whatever those words hold) or a record that is still
                    // on its way
                    if ((badm & 1ull) != 0 || (badm & needed_lanes((u32)rdlane64(w, 0))) != 0) {
                        w = peel_reread(A.node, epoch, A.ctl, e & kIdMask, w, lane);
                        badm = __ballot(tag_stale_or_never(epoch, w));
                        // It never arrived (or the launch is being given up already): fail loudly — ABORT tells the host, which
                        // discards the run — and leave the candidate out.  The chain goes on (no flag to carry through the hot
                        // path); once ABORT is up no re-read waits any more, chains run out and waiting waves leave.
                        // (left out = replaced by a record that loses to everything: DFS root 2^30 - 1 and its root key, no gate has that id.  No early
                        // exit from the candidate: what merges behind this cold block is the record alone, no flag — 8.95 -> 8.58 ms)
                        if ((badm & 1ull) != 0 || (badm & needed_lanes((u32)rdlane64(w, 0))) != 0) {
                            if (lane == 0) atomicAdd(&A.ctl[CTL_ABORT], 1u); wave_join();
                            w = ((u64)epoch << kTagShift) | (lane == 0 ? (u64)kHdrMask << 32 : (lane == 2u ? (u64)kHdrMask : 0ull));
                        }
                    }
                    if (tag_stale_or_never(epoch, w)) w = 0ull;       // (the zero padding of a short record, whatever those words hold)
                }
                const u32 clraw = rdlane((u32)(w >> 32), 1);
                const u32 cpos = rdlane((u32)(w >> 32), 2);      // (raw: lane << 8 | bit under the tag bits)
                lraw = clraw > lraw ? clraw : lraw;
                // the candidate's string with its edge label appended (meaningful while the chunk has room: cpos < kStrWords << 8)
                u64 x = (u64)(u32)w | ((u64)((u32)(w >> 32) & cmp_hi_mask) << 32);
                // (the label as a VECTOR value — one lane holds it — shifted into place: a 64-bit scalar shift, two moves, two selects and
                // two ORs otherwise)
                x |= (u64)(lane == ((cpos >> 8) & 0x3FFFFFu) ? el : 0u) << (cpos & 63u);
                C2A_OPAQUE(x);       // (built HERE, in front of the branches: behind them it is shared code, and the way back into it costs a flag)
                // ONE decision, then ONE place where the champion changes.  The hot case — same DFS root as the champion (the first
                // few roots own nearly everything below them), both paths inside one chunk — is one straight line; everything else
                // (another root, deep trees) one block out of line that ends in the same decision.  (A take per branch made the
                // compiler carry two flags through the branches: ten instructions per candidate of nothing but that.)
                u32 win;
                bool deep = false;
                u32 cdepth = 0, ch_depth = 0;
                if (DEEP) {      // (the plain build never looks at depths here: a node that fills its chunk ends that launch)
                    cdepth = rdlane((u32)w, 0); ch_depth = ch_e != C2A_NONE ? rdlane((u32)ch_w, 0) : 0u;
                    deep = !((cdepth > ch_depth ? cdepth : ch_depth) < kChunkBits);
                }
                if (C2A_LIKELY(!deep)) {
                    // lane 2 holds the root keys: another DFS root decides there (the smaller original id wins; [g] itself is a
                    // champion whose key is the gate's own id).  Same root: neither path is a prefix of the other (that would be a
                    // cycle), and the same node with the other label differs in the appended bit — the first differing bit decides
                    // (strings are stored most significant first: the lowest differing lane, and there the smaller word)
                    const u64 ne = __ballot(x != ch_x) & ~3ull, lt = __ballot(x < ch_x);
                    win = (u32)(lt >> ctz64(ne)) & 1u;
                } else {
                    const u32 croot = hdr_hi(rdlane64(w, 0)), c = e & kIdMask, ch = ch_e & kIdMask;
                    if (croot != ch_root) win = croot < ch_root ? 1u : 0u;      // a larger DFS root loses at once (also to [g] itself)
                    else if (c == ch) win = el < (ch_e >> 31) ? 1u : 0u;
                    // (the result of an out-of-line call counts as divergent; left like that, every value that depends on the
                    // champion would move to vector registers and the whole tournament would be compiled as divergent code)
                    else win = uniform(deep_less(A.node, epoch, A.ctl, c, el, cdepth, w, ch, ch_e >> 31, ch_depth, ch_w, lane) ? 1u : 0u);
                }
                // (as selects — always — this measured slower than the branch: 7.33 against 7.25 ms for the stage)
                if (win) {
                    ch_e = e; ch_pos = cpos; ch_w = w; ch_x = x;
                    if (DEEP) ch_root = hdr_hi(rdlane64(w, 0));       // (the deep compare of a later candidate goes by it; the plain build reads it at the end)
                }
            };
            // the (up to two) records loaded ahead ...
            if (C2A_LIKELY(cur.take >= 1)) {
                candidate(cur.w0, cur.e0);
                if (cur.take >= 2) {
                    candidate(cur.w1, cur.e1);
                    if (cur.take >= 3) {
                        candidate(cur.w2, cur.e2);
                        if (C2A_UNLIKELY(cur.take >= 4)) {
                            candidate(cur.w3, cur.e3);
                            if (cur.take >= 5) {
                                candidate(cur.w4, cur.e4);
                                if (cur.take >= 6) {
                                    candidate(cur.w5, cur.e5);
                                    if (cur.take >= 7) {
                                        candidate(cur.w6, cur.e6);
                                        if (cur.take >= 8) candidate(cur.w7, cur.e7);
                                    }
                                }
                            }
                        }
                    }
                }
            }
            // ... then — cold — the consumer list itself when it holds more than that: EIGHT records in flight at a time, like the
            // hot path (round 5 loaded one per memory round trip: ~1 us per consumer, on one wave, with everything upstream of the
            // gate waiting).  With the relay trees no list is longer than max(kHubMin, kRelayFan) entries: one batch.
            if (C2A_UNLIKELY(cur.more != 0)) {
                const u32 g_off = uniform(own_list_off(uniform(A.cold->n), A.cold->cons_off, A.gstat, gc));
                for (u32 eb = 0; eb < g_cnt; eb += 64) {
                    u32 blk = A.clist[g_off + eb + lane];
                    C2A_PIN(blk);                                // (consumed here, like the records below)
                    u64 smask = __ballot(eb + lane < g_cnt && !((blk & kIdMask) == own_node) &&
                                         !(cur.take >= 1 && blk == cur.e0) && !(cur.take >= 2 && blk == cur.e1) &&
                                         !(cur.take >= 3 && blk == cur.e2) && !(cur.take >= 4 && blk == cur.e3) &&
                                         !(cur.take >= 5 && blk == cur.e4) && !(cur.take >= 6 && blk == cur.e5) &&
                                         !(cur.take >= 7 && blk == cur.e6) && !(cur.take >= 8 && blk == cur.e7));
                    while (smask) {
                        // (lanes beyond the batch load the batch's first record again: no branch around a load, and a value that
                        // is never looked at)
                        u32 c_e0, c_e1, c_e2, c_e3, c_e4, c_e5, c_e6, c_e7, c_n = 0;
                        c_e0 = rdlane(blk, pop_lowest(smask)); ++c_n;
                        c_e1 = c_e2 = c_e3 = c_e4 = c_e5 = c_e6 = c_e7 = c_e0;
                        if (smask) { c_e1 = rdlane(blk, pop_lowest(smask)); ++c_n; }
                        if (smask) { c_e2 = rdlane(blk, pop_lowest(smask)); ++c_n; }
                        if (smask) { c_e3 = rdlane(blk, pop_lowest(smask)); ++c_n; }
                        if (smask) { c_e4 = rdlane(blk, pop_lowest(smask)); ++c_n; }
                        if (smask) { c_e5 = rdlane(blk, pop_lowest(smask)); ++c_n; }
                        if (smask) { c_e6 = rdlane(blk, pop_lowest(smask)); ++c_n; }
                        if (smask) { c_e7 = rdlane(blk, pop_lowest(smask)); ++c_n; }
                        u64 c_w0 = ld_nw(&A.node[(u64)(c_e0 & kIdMask) * kNodeWords + lane]);
                        u64 c_w1 = ld_nw(&A.node[(u64)(c_e1 & kIdMask) * kNodeWords + lane]);
                        u64 c_w2 = ld_nw(&A.node[(u64)(c_e2 & kIdMask) * kNodeWords + lane]);
                        u64 c_w3 = ld_nw(&A.node[(u64)(c_e3 & kIdMask) * kNodeWords + lane]);
                        u64 c_w4 = ld_nw(&A.node[(u64)(c_e4 & kIdMask) * kNodeWords + lane]);
                        u64 c_w5 = ld_nw(&A.node[(u64)(c_e5 & kIdMask) * kNodeWords + lane]);
                        u64 c_w6 = ld_nw(&A.node[(u64)(c_e6 & kIdMask) * kNodeWords + lane]);
                        u64 c_w7 = ld_nw(&A.node[(u64)(c_e7 & kIdMask) * kNodeWords + lane]);
                        // (consumed here: pending at the join they would cost the hot path a wait)
                        C2A_PIN(c_w0); C2A_PIN(c_w1); C2A_PIN(c_w2); C2A_PIN(c_w3); C2A_PIN(c_w4); C2A_PIN(c_w5); C2A_PIN(c_w6); C2A_PIN(c_w7);
                        // ONE instance of the tournament code for the batch: the records move up a register per round
#pragma unroll 1
                        for (u32 k = 0; k < c_n; ++k) {
                            candidate(c_w0, c_e0);
                            c_w0 = c_w1; c_w1 = c_w2; c_w2 = c_w3; c_w3 = c_w4; c_w4 = c_w5; c_w5 = c_w6; c_w6 = c_w7;
                            c_e0 = c_e1; c_e1 = c_e2; c_e2 = c_e3; c_e3 = c_e4; c_e4 = c_e5; c_e5 = c_e6; c_e6 = c_e7;
                        }
                        // (a long list is work too: the watchdog goes by this)
                        if (lane == 0) atomicAdd(&A.ctl[CTL_HEARTBEAT], 1u);
                        wave_join();
                    }
                }
            }
            const ull ph2a = STATS ? c2a_now() : 0;
            if (rmask == 3u) {
                // the entry: lanes 8..15 hold the pushed gate's static records, lanes 32..38 its first consumers
                sreg_wait();                                        // (both atomics are back: the unit is in BEGIN before the entry can be seen)
                push_t = sreg_get<kSregPush>(sr); push_c = sreg_get<kSregPush + 1>(sr);
                if (STATS) ph_pwait += c2a_now() - ph2a;
                const u32 t = push_f * A.q_cap + push_t;            // (all slots together stay below 2^32: the host checks)
                if (STATS && lane == 0) dq_time[t] = ph0;
                // ONE masked store: a lane knows which word of the entry is its own (ent_off, set up once per wave)
                // (the two lane masks are made HERE, a compare each: as loop invariants they would live in scalar register pairs,
                // i.e. be spilled and fetched back, two lane reads and an exec shuffle apiece)
                u32 ln = lane, eo = ent_off;
                C2A_OPAQUE(ln); C2A_OPAQUE(eo);
                const u32 c39 = wrlane_c<39>(g_dep1, cur.clp);
                const u32 lo = ln < 32u ? cur.gw : c39;
                char* slot = reinterpret_cast<char*>(A.fifo + (u64)t * kSlotWords);
                if (eo != C2A_NONE) st_nw(reinterpret_cast<u64*>(slot + eo), ((u64)A.run << 32) | lo);
                wave_join();
            }
            const ull ph3 = STATS ? c2a_now() : 0;
            // ---- the node: its string is the champion's string with the label appended — the register built above
            u32 depth = 0, my_label = 0, cprev = C2A_NONE, my_pos = (kHdrWords << 8) | (kWordBits - 1u);
            u64 str = ch_x;      // (a DFS root of its own: the empty string, lane 2 = its root key — the reset above)
            const u32 ch = ch_e != C2A_NONE ? (ch_e & kIdMask) : C2A_NONE;
            if (C2A_LIKELY(ch_e != C2A_NONE)) {
                // (what only the winner is asked: depth and DFS root, out of its record)
                const u64 h0 = rdlane64(ch_w, 0);
                depth = (u32)h0 + 1; my_label = ch_e >> 31; ch_root = hdr_hi(h0);
                cprev = rdlane((u32)ch_w, 1);
                const u32 ppos = ch_pos & kHdrMask;
                u32 np = ppos - 1u;                            // where MY child's label goes: one bit down ...
                if (C2A_UNLIKELY(ppos >= (kNodeWords << 8))) {       // the parent filled its chunk: a fresh one, the parent is its anchor
                    if (DEEP) {
                        cprev = ch; np = (kHdrWords << 8) | (kWordBits - 2u);
                        str = lane == kHdrWords ? (u64)my_label << (kWordBits - 1u) : (lane == 2u ? (u64)ch_root : 0ull);
                    } else peel_need_deep(A.ctl, lane);        // (this launch is over: what it writes from here on is never read)
                }
                if (C2A_UNLIKELY((ppos & 255u) == 0u)) np = ppos + 256u + kWordBits - 1u;      // ... or the top bit of the next word
                my_pos = np;
            }
            // the tree entry and the child link: wave-uniform data, read by later launches only — SCALAR stores (no exec
            // shuffle, no moves into vector registers; written back at the end of the wave: sstore_flush)
            // (a relay's level does not count — the reverse Kahn levels of the real gates stay exact —: HUBS AND RELAYS)
            // (ids — original ones too — are below 2^29; a relay's "original id" has bit 29 set)
            const u32 level = inc_unless_bit<29>(lraw & kHdrMask, g_orig);
            sstore_x4(&A.meta[gc], ch, depth, ch_root, my_label | (level << 1));
            if (C2A_LIKELY(ch_e != C2A_NONE)) sstore_x1_at(A.child, (2u * ch + my_label) * 4u, gc);      // (gate ids are below 2^29: the byte offset fits 32 bits)
            // (the three header words go into lanes 0..2 with v_writelane: a lane == k ladder is masked code)
            const u32 tag_hi = epoch << 30;
            u32 w_lo = (u32)str, w_hi = (u32)(str >> 32) | tag_hi;
            w_lo = wrlane_c<0>(depth, w_lo);  w_hi = wrlane_c<0>((ch_root & kHdrMask) | tag_hi, w_hi);
            w_lo = wrlane_c<1>(cprev, w_lo);  w_hi = wrlane_c<1>((level & kHdrMask) | tag_hi, w_hi);
            w_hi = wrlane_c<2>(my_pos | tag_hi, w_hi);      // (lane 2's low half is the root key already: str carries it)
            const u64 my_w = (u64)w_lo | ((u64)w_hi << 32);
            st_nw(&A.node[(u64)gc * kNodeWords + lane], my_w);
            ++processed;
            if (C2A_UNLIKELY((processed & C2A_HB_MASK) == 0) && lane == 0) atomicAdd(&A.ctl[CTL_HEARTBEAT], 1u);
            if (STATS) {
                const ull ph4 = c2a_now();
                if (dt_trace && lane == 0) {
                    dt_trace[3 * (u64)gc] = (ph0 << 2) | came; dt_trace[3 * (u64)gc + 1] = ph4;
                    // what the step was made of: top wait, issue, tournament, stores (ticks, 12 bits each) | pushed << 48 | records loaded ahead << 49 | cold << 52
                    const ull c12 = 0xFFFull;
                    dt_trace[3 * (u64)gc + 2] = ((ph1 - ph0) & c12) | (((ph2 - ph1) & c12) << 12) | (((ph3 - ph2) & c12) << 24) | (((ph4 - ph3) & c12) << 36) |
                                                       ((ull)(rmask == 3u) << 48) | ((ull)(cur.take < 7 ? cur.take : 7) << 49) | ((ull)(cur.more != 0) << 52);
                }
                came = 0;
                ++ph_steps;
                if (cur.take == 0) ++ph_noload;
                ph_a += ph1 - ph0; ph_b += ph2 - ph1; ph_c += ph3 - ph2; ph_d += ph4 - ph3; ph_push += ph3 - ph2a; ph_cold += cur.more ? 1 : 0;
            }
            // (levels grow along a chain: its last step has the highest)
            if (C2A_UNLIKELY(nxt == C2A_NONE)) { max_level = level > max_level ? level : max_level; return true; }      // the chain ends here
            // what the next step reuses: this gate as the first candidate of nxt (same DFS root: ch_root stays)
            own_node = gc; own_lraw = level | tag_hi;
            ch_x = str;
            ch_x |= (u64)(lane == (my_pos >> 8) ? nxt_label : 0u) << (my_pos & 63u);
            ch_e = gc | (nxt_label << 31); ch_pos = my_pos; ch_w = my_w;
            g = nxt;
            return false;
        };
        for (;;) {
            if (step(StepSet<0>(), StepSet<1>(), S0, S1)) break;
            if (step(StepSet<1>(), StepSet<0>(), S1, S0)) break;
        }
        if (C2A_UNLIKELY((++iters & 63u) == 0)) {                   // somebody gave up (watchdog): leave, the host reports it
            u32 ab = 0;
            if (lane == 0) ab = ld_a32(&A.ctl[CTL_ABORT]); wave_join();
            if (rdlane(ab, 0)) break;
        }
    }
    sstore_flush();
    if (lane == 0) {
        u32* acct = &A.ctl[CTL_PROC + (me & (kAcctShards - 1u)) * kAcctStride];
        if (processed) atomicAdd(acct, processed);
        if (max_level) atomicMax(acct + 1, max_level);
        if (STATS) {
            ull* stats = A.cold->stats;
            if (stats) {
                const ull t = c2a_now(); st_busy += t - st_t0;
                atomicAdd(&stats[0], (ull)st_pops); atomicAdd(&stats[1], (ull)st_polls); atomicAdd(&stats[2], (ull)st_push);
                atomicAdd(&stats[3], st_busy); atomicAdd(&stats[4], st_idle); atomicAdd(&stats[5], (ull)st_seeds); atomicAdd(&stats[6], (ull)processed);
                atomicAdd(&stats[7], ph_w1 | (ph_w2 << 32)); atomicAdd(&stats[16], ph_w3);
                atomicAdd(&stats[9], ph_a); atomicAdd(&stats[10], ph_b); atomicAdd(&stats[11], ph_c); atomicAdd(&stats[12], ph_d);
                atomicAdd(&stats[13], ph_steps); atomicAdd(&stats[14], ph_noload); atomicAdd(&stats[15], ph_start);
                atomicAdd(&stats[17], ph_push); atomicAdd(&stats[18], ph_cold); atomicAdd(&stats[19], ph_pwait);
            }
        }
    }
}

}  // namespace c2a

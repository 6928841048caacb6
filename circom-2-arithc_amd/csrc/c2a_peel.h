// c2a_peel.h — the exact DFS tree of topological_sort (/root/reference/src/topological_sort.rs:3-50) without a DFS,
// as ONE dataflow launch on gfx950.
//
// Claim (re-proved by the tests): the DFS tree parent of gate g is the consumer that reaches g by the
// lexicographically smallest path from a virtual root whose children are the gates 0..n-1 in id order
// (topological_sort.rs:11-13) and whose other edges carry label 0 (lh producer) / 1 (rh producer)
// (compiler.rs:408-421, topological_sort.rs:42-44); the DFS post-order is the post-order of that tree.  A gate can pick
// its parent once every consumer has picked its own (reverse Kahn order from the sinks).
//
// NODE RECORDS.  Every gate owns one 512-byte record of 64 self-validating 8-byte words: bit 63 of a word is the tag of
// the run that wrote it (a word is written by ONE agent-scope store, so it is never torn; a reader that sees the wrong
// tag simply reads again — no flag, no fence, no acknowledgement wait anywhere on the path):
//   word 0      root gate id << 32 | depth in the DFS tree
//   word 1      reverse Kahn level << 32 | cprev (ancestor at the start of the node's current chunk; deep trees only)
//   words 2..63 the node's path string: 63 payload bits per word, bit j = label of the edge entering depth j+1
// A string is held in chunks of kChunkBits = 62 x 63 = 3906 bits; a node keeps its CURRENT chunk only, so comparing two
// candidates shallower than a chunk (the 10 M-gate headline graph: depth 3 471) is ONE coalesced load per candidate,
// XOR, ballot, count trailing zeros.  Deeper trees add one cprev hop per chunk level.
//
// DATAFLOW.  The consumer list of every gate is static (clist, built by k_ginfo from the deps closure), so nothing but
// the node records is exchanged between waves.  A gate g is CLAIMED by the wave that takes the last ticket on fill[g];
// tickets are taken when a consumer is claimed, NOT when it is finished, so the ticket round trip overlaps the
// consumer's own tournament and the claiming wave goes on with g right after finishing the consumer: its own record is
// still in registers, the other consumers' records are loaded in one round trip (and read again if a word is not there
// yet).  A single-consumer producer needs no ticket at all.  A second producer completed by the same gate is handed to a
// ticket queue; idle waves pop.  No deadlock: a ticket is only ever taken by a wave at the START of a step that it then
// runs to the end, so when a gate is claimed every one of its candidates is in a step that is running (or done) on some
// resident wave; a wave only ever waits for such a record, the graph is acyclic, so every wait ends.
// Termination: a wave counts as idle from its first empty-handed poll until the moment BEFORE it tries to claim an
// entry; when the idle count reaches the number of waves that have STARTED, nobody can push any more and the peel is
// over once every queue is seen empty — co-residency of the whole grid is not required (a wave that starts late finds
// the seed pool empty and leaves).  Watchdog: lack of GLOBAL progress (a heartbeat the working waves bump) — a long
// critical path followed by one wave is not an error.
#pragma once
#include "c2a_platform.h"

namespace c2a {

constexpr u32 kIdMask = 0x7FFFFFFFu;
constexpr u32 kNodeWords = 64;
constexpr u32 kHdrWords = 2;
constexpr u32 kWordBits = 63;
constexpr u32 kChunkBits = (kNodeWords - kHdrWords) * kWordBits;      // 3906
constexpr u64 kTagBit = 1ull << 63;
constexpr u64 kPayload = kTagBit - 1ull;

constexpr u32 kIdleCounters = 64;
constexpr u32 kQStride = 16;                // u64 words between two queues' head/tail words (one queue per 128-byte line)
constexpr int kStrMax = 4;                  // candidate records held in registers per round
#ifdef C2A_EMULATE
constexpr u32 kPollLimit = 1;               // steps are atomic there: a missing record is a bug, fail at once
#else
constexpr u32 kPollLimit = 1u << 21;        // ~1 s of polling for a record: give up (reported as an error) instead of hanging
#endif
constexpr u32 kWatchdogChecks = 1u << 15;   // idle-side checks (one per 32 polls, ~100 us apart) without global progress
// control block (u32 words; every hot word on its own 64-byte line)
enum PeelCtl { CTL_PROCESSED = 0, CTL_MAXLEVEL = 1, CTL_ABORT = 2, CTL_REREADS = 3, CTL_HEARTBEAT = 16, CTL_STARTED = 32, CTL_SEEDNEXT = 48, CTL_WORDS = 64 };

struct PeelArgs {
    u32 n;
    u32 epoch;                 // tag (0/1) of this run's node words
    const uint4* ginfo;        // [n] {dep0, dep1, cons_off, cons_cnt}
    const uint4* ginfo2;       // [n] {cons_off[dep0], cons_cnt[dep0], cons_off[dep1], cons_cnt[dep1]}
    const u32* clist;          // [edges + 64] consumer | edge label << 31, grouped by producer
    u64* node;                 // [n][64] node records
    u32* fill;                 // [n] claim tickets taken so far (zeroed per run)
    uint4* meta;               // [n] {parent | NONE, depth, root, label | level << 1}: read by later launches only
    u32* child;                // [2n] tree children by label (0xFF-filled per run)
    u32* seeds;                // [n_regions][region_cap] producers claimed by the sinks pass
    u32* seed_cnt;             // [n_regions]
    u32 n_regions, region_cap;
    u32 n_queues, q_cap;
    u64* q_ht;                 // [n_queues * kQStride] head (low word) | tail (high word)
    u64* q_items;              // [n_queues][q_cap] (gate + 1) | cons_off << 32; 0 = not written yet
    u32* link;                 // [n] wave-private overflow stacks (queue full — never in practice)
    u32* idle;                 // [kIdleCounters * 16]
    u32* ctl;                  // [CTL_WORDS]
    ull* stats;                // optional diagnostics (16 words), nullptr normally
};

__device__ __forceinline__ u32 chunk_of(u32 depth) { return depth ? (depth - 1) / kChunkBits : 0u; }
__device__ __forceinline__ u32 chunk_len(u32 depth) { return depth - chunk_of(depth) * kChunkBits; }
__device__ __forceinline__ u32 str_words(u32 len) { return (len + kWordBits - 1) / kWordBits; }
__device__ __forceinline__ u32 ctz64(u64 x) { return (u32)__ffsll((long long)x) - 1u; }

__device__ __forceinline__ u64 ld_nw(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_nw(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u32 ld_a32(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_a32(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// wave-uniform pick of one lane's value (v_readlane: a few cycles; ds_bpermute through __shfl costs ~100)
__device__ __forceinline__ u32 rdlane(u32 v, u32 j) {
#ifdef C2A_EMULATE
    return __shfl(v, (int)j, 64);
#else
    return (u32)__builtin_amdgcn_readlane((int)v, (int)j);
#endif
}
__device__ __forceinline__ u32 uniform(u32 v) {
#ifdef C2A_EMULATE
    return v;
#else
    return (u32)__builtin_amdgcn_readfirstlane((int)v);
#endif
}
__device__ __forceinline__ uint4 uniform4(const uint4& v) { return make_uint4(uniform(v.x), uniform(v.y), uniform(v.z), uniform(v.w)); }
__device__ __forceinline__ u64 rdlane64(u64 v, u32 j) { return (u64)rdlane((u32)v, j) | ((u64)rdlane((u32)(v >> 32), j) << 32); }

__device__ __forceinline__ ull c2a_now() {
#ifdef C2A_EMULATE
    return 0;
#else
    return wall_clock64();      // constant 100 MHz
#endif
}
__device__ __forceinline__ void peel_sleep(int units) {
#ifndef C2A_EMULATE
    if (units <= 4) __builtin_amdgcn_s_sleep(4); else if (units <= 16) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(64);
#else
    (void)units;
#endif
}

__device__ __forceinline__ u64 hdr0_word(u32 root, u32 depth) { return ((u64)(root & kIdMask) << 32) | depth; }
__device__ __forceinline__ u64 hdr1_word(u32 level, u32 cprev) { return ((u64)(level & kIdMask) << 32) | cprev; }
__device__ __forceinline__ u32 hdr_hi(u64 w) { return (u32)(w >> 32) & kIdMask; }

// Is this record (one word per lane) completely there?  Header first, then as many string words as its depth needs.
__device__ __forceinline__ bool rec_valid(u64 w, u32 lane, u32 epoch) {
    const u64 h0 = rdlane64(w, 0), h1 = rdlane64(w, 1);
    if ((u32)(h0 >> 63) != epoch || (u32)(h1 >> 63) != epoch) return false;
    const u32 nw = str_words(chunk_len((u32)h0));
    return __ballot(lane >= kHdrWords && lane < kHdrWords + nw && (u32)(w >> 63) != epoch) == 0ull;
}

// bit j of the string chunk held one word per lane
__device__ __forceinline__ u32 str_bit(u64 w, u32 j) { return (u32)(rdlane64(w, kHdrWords + j / kWordBits) >> (j % kWordBits)) & 1u; }

// P(a).la < P(b).lb ?  wa / wb: this lane's word of the records holding the chunk in which the two paths can first
// differ, lena / lenb: bits of that chunk.  Wave-uniform result.
__device__ __forceinline__ bool str_less_wave(u64 wa, u32 lena, u32 la, u64 wb, u32 lenb, u32 lb, u32 lane) {
    const u32 minlen = lena < lenb ? lena : lenb;
    const u32 lo = (lane - kHdrWords) * kWordBits;
    u64 x = (wa ^ wb) & kPayload;
    if (lane < kHdrWords || lo >= minlen) x = 0;
    else if (minlen - lo < kWordBits) x &= (1ull << (minlen - lo)) - 1ull;
    const u64 bal = __ballot(x != 0);
    if (bal) {
        const u32 L = ctz64(bal);
        const u64 xl = rdlane64(x, L);
        const u64 al = rdlane64(wa, L);
        return ((al >> ctz64(xl)) & 1ull) == 0;
    }
    if (lena == lenb) return la < lb;
    if (lena < lenb) return la < str_bit(wb, lena);
    return str_bit(wa, lenb) < lb;
}

// a word another wave wrote some time ago (an ancestor's): read until its tag is this run's
__device__ __forceinline__ u64 ld_word_wait(const u64* p, u32 epoch, u32* ctl) {
    u64 v = ld_nw(p);
    u32 spins = 0;
    while ((u32)(v >> 63) != epoch) {
        if (++spins > (1u << 22)) { atomicAdd(&ctl[CTL_ABORT], 1u); break; }
        peel_sleep(4);
        v = ld_nw(p);
    }
    return v;
}
// whole record of a finished node, one word per lane (lanes past its string read word 0 again)
__device__ __forceinline__ u64 ld_rec_wait(const u64* node_base, u32 epoch, u32* ctl, u32 node, u32 len, u32 lane) {
    const u32 nw = kHdrWords + str_words(len);
    const u64* p = node_base + (u64)node * kNodeWords + (lane < nw ? lane : 0u);
    u64 v = ld_nw(p);
    u32 spins = 0;
    while (__ballot((u32)(v >> 63) != epoch) != 0ull) {
        if (++spins > (1u << 22)) { if (lane == 0) atomicAdd(&ctl[CTL_ABORT], 1u); break; }
        peel_sleep(4);
        v = ld_nw(p);
    }
    return v;
}

// Trees deeper than one chunk: bring the two nodes to the first chunk in which their paths can differ (cprev hops),
// then compare that chunk.  wa_in / wb_in: the two nodes' own records (already in registers).  Out of line and with
// plain by-value arguments: it must not drag the kernel's argument block into memory.
__device__ __attribute__((noinline)) bool deep_less(const u64* node_base, u32 epoch, u32* ctl, u32 a, u32 la, u32 da, u64 wa_in,
                                                    u32 b, u32 lb, u32 db, u64 wb_in, u32 lane) {
#define C2A_CPREV(x) ((u32)ld_word_wait(node_base + (u64)(x) * kNodeWords + 1, epoch, ctl))
#define C2A_BIT0(x) ((u32)ld_word_wait(node_base + (u64)(x) * kNodeWords + kHdrWords, epoch, ctl) & 1u)
    u32 ia = chunk_of(da), ib = chunk_of(db);
    u32 lena = chunk_len(da), lenb = chunk_len(db);
    u32 below_a = C2A_NONE, below_b = C2A_NONE;
    bool a_own = true, b_own = true;
    u32 pa = (u32)rdlane64(wa_in, 1), pb = (u32)rdlane64(wb_in, 1);      // cprev of the node currently held in a / b
    while (ia > ib) { below_a = a; a = pa; --ia; lena = kChunkBits; a_own = false; if (ia) pa = C2A_CPREV(a); }
    while (ib > ia) { below_b = b; b = pb; --ib; lenb = kChunkBits; b_own = false; if (ib) pb = C2A_CPREV(b); }
    while (ia > 0 && a != b) {
        if (pa == pb) break;
        below_a = a; below_b = b;
        a = pa; b = pb; --ia;
        lena = lenb = kChunkBits; a_own = b_own = false;
        if (ia) { pa = C2A_CPREV(a); pb = C2A_CPREV(b); }
    }
    if (a == b) {      // one node is the chunk-boundary ancestor of the other: the other's next label decides
        if (below_a != C2A_NONE) return C2A_BIT0(below_a) < lb;
        return la < C2A_BIT0(below_b);
    }
    const u64 wa = a_own ? wa_in : ld_rec_wait(node_base, epoch, ctl, a, lena, lane);
    const u64 wb = b_own ? wb_in : ld_rec_wait(node_base, epoch, ctl, b, lenb, lane);
    return str_less_wave(wa, lena, la, wb, lenb, lb, lane);
#undef C2A_CPREV
#undef C2A_BIT0
}

// ------------------------------------------------------------------------------------------------
// static per-gate data of the dataflow launch (after the consumer counts have been scanned):
// ginfo, ginfo2 and the consumer lists (eslot[2g + l] = index of edge (g, l) in its producer's list, from k_deps)
// ------------------------------------------------------------------------------------------------
__global__ void k_ginfo(u32 n, const u32* __restrict__ dep0, const u32* __restrict__ dep1, const u32* __restrict__ cons_off,
                        const u32* __restrict__ cons_cnt, const u32* __restrict__ eslot, uint4* ginfo, uint4* ginfo2, u32* clist) {
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (u64)gridDim.x * blockDim.x) {
        const u32 d0 = dep0[g], d1 = dep1[g];
        ginfo[g] = make_uint4(d0, d1, cons_off[g], cons_cnt[g]);
        uint4 g2 = make_uint4(0, 0, 0, 0);
        if (d0 != C2A_NONE) { g2.x = cons_off[d0]; g2.y = cons_cnt[d0]; clist[g2.x + eslot[2 * g]] = (u32)g; }
        if (d1 != C2A_NONE) { g2.z = cons_off[d1]; g2.w = cons_cnt[d1]; clist[g2.z + eslot[2 * g + 1]] = (u32)g | 0x80000000u; }
        ginfo2[g] = g2;
    }
}

// ------------------------------------------------------------------------------------------------
// sinks (gates nobody consumes: DFS roots of depth 0, no candidates) — a plain grid-stride pass; the producers they
// claim seed the dataflow launch.  No shared counter: workgroup b appends to its own region under its own counter.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_peel_sinks(PeelArgs A) {
    __shared__ u32 s_done[4];
    const u32 lane = threadIdx.x & 63u;
    u32* out = A.seeds + (u64)blockIdx.x * A.region_cap;
    u32* counter = &A.seed_cnt[blockIdx.x];
    const u64 lt_mask = (1ull << lane) - 1ull;
    const u64 tag = A.epoch ? kTagBit : 0ull;
    u32 done = 0;
    for (u64 base = (u64)blockIdx.x * 256; base < A.n; base += (u64)gridDim.x * 256) {
        const u64 g = base + threadIdx.x;
        u32 rdy[2] = {C2A_NONE, C2A_NONE};
        if (g < A.n) {
            const uint4 gi = A.ginfo[g];
            if (gi.w == 0) {
                ++done;
                A.meta[g] = make_uint4(C2A_NONE, 0u, (u32)g, 0u);
                A.node[g * kNodeWords] = tag | hdr0_word((u32)g, 0u);
                A.node[g * kNodeWords + 1] = tag | hdr1_word(0u, C2A_NONE);
                const uint4 g2 = A.ginfo2[g];
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
        if (t) atomicAdd(&A.ctl[CTL_PROCESSED], t);
    }
}

// hand a claimed gate to whoever is idle: ticket on a queue (round robin over all queues), then the entry.
// false = the queue is full (cannot happen with the host's q_cap unless requeues pile up)
__device__ __forceinline__ bool peel_push(const PeelArgs& A, u32& push_rr, u32 lane, u32 gate, u32 cons_off) {
    u32* head_w = reinterpret_cast<u32*>(A.q_ht);
    const u32 q = (push_rr++) % A.n_queues;
    u32 t = 0;
    if (lane == 0) t = atomicAdd(&head_w[2 * (u64)q * kQStride + 1], 1u);
    t = rdlane(t, 0);
    if (t >= A.q_cap) return false;
    if (lane == 0) st_nw(&A.q_items[(u64)q * A.q_cap + t], (u64)(gate + 1u) | ((u64)cons_off << 32));
    return true;
}

// the dataflow launch: 64-thread workgroups (one wave each)
template <bool STATS>
__global__ void __launch_bounds__(64) k_peel(PeelArgs A) {
    const u32 lane = threadIdx.x;
    const u32 me = blockIdx.x;
    const u64 tag = A.epoch ? kTagBit : 0ull;
    u32* head_w = reinterpret_cast<u32*>(A.q_ht);            // head of queue q = word 2 q kQStride, tail = the next word
    const u32 home_q = me % A.n_queues;
    u32 push_rr = me;                        // round-robin cursor of this wave's pushes
    u32 roam = me * 0x9E3779B1u;             // pseudo-random walk over the other queues
    bool registered = false;                 // counted in idle[]
    bool seeds_left = true;
    u32 region = 0, idx = 0, region_cnt = 0;
    u32 head = C2A_NONE;                     // wave-private stack (overflow of the hand-off queues)
    u32 processed = 0, max_level = 0, iters = 0;
    u32 st_pops = 0, st_polls = 0, st_push = 0, st_seeds = 0, st_rpolls = 0;
    ull st_busy = 0, st_idle = 0, st_t0 = STATS ? c2a_now() : 0;
    ull ph_a = 0, ph_b = 0, ph_c = 0, ph_d = 0, ph_steps = 0, ph_noload = 0, ph_start = 0;      // STATS: phase times of the chain step
    if (lane == 0) atomicAdd(&A.ctl[CTL_STARTED], 1u);
    for (;;) {
        // ---- next piece of work: own stack, the seed pool, then the hand-off queues
        u32 g = C2A_NONE;
        u32 coff = C2A_NONE;                 // cons_off of g when the entry carried it
        if (head != C2A_NONE) {
            g = head;
            head = ld_a32(&A.link[g]);
        } else {
            if (seeds_left) {
                while (idx >= region_cnt) {
                    u32 r = 0;
                    if (lane == 0) r = atomicAdd(&A.ctl[CTL_SEEDNEXT], 1u);
                    r = rdlane(r, 0);
                    if (r >= A.n_regions) { seeds_left = false; break; }
                    region = r; idx = 0; region_cnt = A.seed_cnt[r];
                }
                if (seeds_left) { g = A.seeds[(u64)region * A.region_cap + idx]; ++idx; ++st_seeds; }
            }
            if (g == C2A_NONE) {
                if (STATS) { const ull t = c2a_now(); st_busy += t - st_t0; st_t0 = t; }
                // poll: home queue, then a roaming one
                u32 polls = 0, hint = C2A_NONE, hb_seen = 0, hb_checks = 0;
                for (;;) {
                    u32 q = (polls & 1u) ? (roam = roam * 1664525u + 1013904223u, (roam >> 8) % A.n_queues) : home_q;
#ifdef C2A_EMULATE
                    q = (home_q + polls) % A.n_queues;           // deterministic sweep
#endif
                    if (hint != C2A_NONE) { q = hint; hint = C2A_NONE; }
                    u64 ht = 0;
                    if (lane == 0) ht = ld_nw(&A.q_ht[(u64)q * kQStride]);
                    ht = rdlane64(ht, 0);
                    const u32 qh = (u32)ht, qt_raw = (u32)(ht >> 32);
                    const u32 qt = qt_raw < A.q_cap ? qt_raw : A.q_cap;
                    if (qh < qt) {
                        if (registered) { if (lane == 0) atomicAdd(&A.idle[(me % kIdleCounters) * 16], 0xFFFFFFFFu); registered = false; }
                        u32 old = 0;
                        if (lane == 0) old = atomicCAS(&head_w[2 * (u64)q * kQStride], qh, qh + 1);
                        old = rdlane(old, 0);
                        if (old == qh) {
                            u64 v = 0;
                            u32 spins = 0;
                            do {
                                if (lane == 0) v = ld_nw(&A.q_items[(u64)q * A.q_cap + qh]);
                                v = rdlane64(v, 0);
                            } while (v == 0 && ++spins < (1u << 22));
                            if (v) {
                                g = (u32)v - 1u;
                                coff = (u32)(v >> 32);
                                ++st_pops;
                            } else if (lane == 0) atomicAdd(&A.ctl[CTL_ABORT], 1u);
                            break;
                        }
                        hint = q;                                    // lost the race: look at the same queue again at once
                        continue;
                    }
                    if (!registered) { if (lane == 0) atomicAdd(&A.idle[(me % kIdleCounters) * 16], 1u); registered = true; }
                    ++polls;
                    if ((polls & 31u) == 0) {
                        u32 cnt = lane < kIdleCounters ? ld_a32(&A.idle[lane * 16]) : 0u;
#pragma unroll
                        for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
                        // the number of started waves is read AFTER the idle counters (a wave registers as started first)
                        u32 c3 = 0;
                        if (cnt != 0xFFFFFFFFu && lane < 3) c3 = ld_a32(&A.ctl[lane == 0 ? CTL_STARTED : (lane == 1 ? CTL_ABORT : CTL_HEARTBEAT)]);
                        const u32 started = rdlane(c3, 0), aborted = rdlane(c3, 1), hb = rdlane(c3, 2);
                        if (aborted) break;
                        if (cnt >= started) {
                            // every started wave is idle or gone, so nobody pushes any more: finished unless an entry is still queued
                            u32 found = C2A_NONE;
                            for (u32 qq = lane; qq < A.n_queues; qq += 64) {
                                const u64 x = ld_nw(&A.q_ht[(u64)qq * kQStride]);
                                const u32 xt = (u32)(x >> 32) < A.q_cap ? (u32)(x >> 32) : A.q_cap;
                                if ((u32)x < xt) found = qq;
                            }
                            const u64 fm = __ballot(found != C2A_NONE);
                            if (fm == 0) break;
                            hint = rdlane(found, ctz64(fm));
                            continue;
                        }
                        if (hb != hb_seen) { hb_seen = hb; hb_checks = 0; }
                        else if (++hb_checks > kWatchdogChecks) { if (lane == 0) atomicAdd(&A.ctl[CTL_ABORT], 1u); break; }
                    }
                    // back off: the longer nothing turns up, the less often this wave asks (64 clocks per unit, <= ~3 us)
                    peel_sleep(polls < 8 ? 4 : (polls < 64 ? 16 : 64));
                }
                st_polls += polls;
                if (STATS) { const ull t = c2a_now(); st_idle += t - st_t0; st_t0 = t; }
                if (g == C2A_NONE) break;
            }
        }
        const ull ph_s0 = STATS ? c2a_now() : 0;
        // ---- static data of g (a chain step gets all of this prefetched by the step before)
        // (all of it consumed HERE, in scalar registers where wave-uniform: a load still pending at the loop header would
        // make every chain step wait for the previous step's stores)
        uint4 gi = uniform4(A.ginfo[g]);
        uint4 gi2 = uniform4(A.ginfo2[g]);
        if (coff == C2A_NONE) coff = gi.z;
        u32 cl = A.clist[coff + lane];                                   // clist is padded by 64 entries
#ifndef C2A_EMULATE
        asm volatile("" : "+v"(cl) :: "memory");
#endif
        // ---- follow the chain from g.  When the chain goes on from gate p to its producer g, p's own record is still in
        // registers: if p is g's ONLY consumer the whole step needs no load and no ticket at all.
        if (STATS) ph_start += c2a_now() - ph_s0;
        bool own_valid = false;
        u32 own_node = 0, own_label = 0, own_depth = 0, own_root = 0, own_level = 0;
        u64 own_w = 0;
        for (;;) {
            const ull ph0 = STATS ? c2a_now() : 0;
            ull ph1 = 0;
            const u32 cnt = gi.w;
            const u32 dl = lane == 0 ? gi.x : (lane == 1 ? gi.y : C2A_NONE);
            const u32 dcnt = lane == 0 ? gi2.y : (lane == 1 ? gi2.w : 0u);
            // claim tickets on the producers (nothing of this gate's tournament is needed for them)
            u32 kfill = 0;
            if (dl != C2A_NONE && dcnt > 1u) kfill = atomicAdd(&A.fill[dl], 1u);
            // static data of both producers, BRANCH-FREE (clamped index, result discarded by lanes with nothing to load)
            const u32 dl_c = dl != C2A_NONE ? dl : 0u;
            const uint4 gd_raw = A.ginfo[dl_c];
            const uint4 gd2_raw = A.ginfo2[dl_c];
            u32 clp;                                                     // lanes 0..31: dep0's consumers, 32..63: dep1's
            {
                const u32 half = lane >> 5, i = lane & 31u;
                const u32 poff = half ? gi2.z : gi2.x, pcnt = half ? gi2.w : gi2.y;
                clp = A.clist[poff + (i < pcnt ? i : 0u)];
            }
            // ---- tournament.  Champion so far (wave-uniform); NONE = the virtual-root candidate [g]
            u32 ch = C2A_NONE, ch_el = 0, ch_root = g, ch_depth = 0, level = 0;
            u64 ch_w = 0;
            if (own_valid) {
                level = own_level + 1;
                if (own_root < g) { ch = own_node; ch_el = own_label; ch_root = own_root; ch_depth = own_depth; ch_w = own_w; }
            }
            bool gave_up = false;
            // one block of <= 64 consumers (one per lane)
            auto tournament_block = [&](u32 ce, u32 n_here) {
                const bool valid = lane < n_here;
                const u32 c = ce & kIdMask, l = ce >> 31;
                u64 smask = __ballot(valid && !(own_valid && c == own_node));
                while (smask) {
                    u32 cc[kStrMax], cel[kStrMax];
                    u32 take = 0;
#pragma unroll
                    for (int t = 0; t < kStrMax; ++t) {
                        if (smask) {
                            const u32 j = ctz64(smask);
                            smask &= smask - 1;
                            cc[t] = rdlane(c, j); cel[t] = rdlane(l, j);
                            take = (u32)t + 1;
                        } else { cc[t] = cc[0]; cel[t] = 0; }
                    }
                    // the records back to back, BRANCH-FREE (unused slots read the first candidate again)
                    u64 sw[kStrMax];
                    if (take == 1) {
                        sw[0] = ld_nw(&A.node[(u64)cc[0] * kNodeWords + lane]);
#pragma unroll
                        for (int t = 1; t < kStrMax; ++t) sw[t] = 0;
                    } else {
#pragma unroll
                        for (int t = 0; t < kStrMax; ++t) sw[t] = ld_nw(&A.node[(u64)cc[t] * kNodeWords + lane]);
                    }
                    // all there?  A record still being written is read again
                    u32 polls = 0;
                    for (;;) {
                        u32 bad = 0;
#pragma unroll
                        for (int t = 0; t < kStrMax; ++t)
                            if ((u32)t < take && !rec_valid(sw[t], lane, A.epoch)) bad |= 1u << t;
                        if (STATS && ph1 == 0) ph1 = c2a_now();
                        if (!bad) break;
                        if (++polls > kPollLimit) { gave_up = true; break; }
                        peel_sleep(polls < 8 ? 4 : 16);
#pragma unroll
                        for (int t = 0; t < kStrMax; ++t)
                            if (bad & (1u << t)) sw[t] = ld_nw(&A.node[(u64)cc[t] * kNodeWords + lane]);
                    }
                    st_rpolls += polls;
                    if (gave_up) return;
#pragma unroll
                    for (int t = 0; t < kStrMax; ++t) {
                        if ((u32)t < take) {
                            const u64 h0 = rdlane64(sw[t], 0), h1 = rdlane64(sw[t], 1);
                            const u32 croot = hdr_hi(h0), cdepth = (u32)h0, clevel = hdr_hi(h1) + 1u;
                            level = clevel > level ? clevel : level;
                            bool less;
                            if (croot != ch_root) less = croot < ch_root;          // a larger DFS root loses at once (also to [g] itself)
                            else if (ch == C2A_NONE) less = false;                 // (root == g: impossible in a DAG)
                            else if (cc[t] == ch) less = cel[t] < ch_el;
                            else if (cdepth <= kChunkBits && ch_depth <= kChunkBits)
                                less = str_less_wave(sw[t], cdepth, cel[t], ch_w, ch_depth, ch_el, lane);
                            else less = deep_less(A.node, A.epoch, A.ctl, cc[t], cel[t], cdepth, sw[t], ch, ch_el, ch_depth, ch_w, lane);
                            if (less) { ch = cc[t]; ch_el = cel[t]; ch_root = croot; ch_depth = cdepth; ch_w = sw[t]; }
                        }
                    }
                }
            };
            // the first block's consumers are already in registers: NO load (and so no wait for the tickets, the static
            // prefetches and the previous step's stores) stands between the top of the step and the record loads
            tournament_block(cl, cnt < 64u ? cnt : 64u);
            for (u32 eb = 64; eb < cnt && !gave_up; eb += 64)
                tournament_block(A.clist[gi.z + eb + lane], cnt - eb < 64u ? cnt - eb : 64u);
            const ull ph2 = STATS ? c2a_now() : 0;
            // who claimed the producers (the tickets came back with the records: vmcnt returns in order)
            const uint4 gd = dl != C2A_NONE ? gd_raw : make_uint4(0, 0, 0, 0);
            const uint4 gd2 = dl != C2A_NONE ? gd2_raw : make_uint4(0, 0, 0, 0);
            // (the empty asm is the ticket's first use as far as the compiler can tell: it keeps the wait for the returning
            // atomic — which would otherwise be pulled up to the top of the step — down here)
#ifndef C2A_EMULATE
            asm volatile("" : "+v"(kfill) :: "memory");
#endif
            const bool last = dl != C2A_NONE && (dcnt == 1u || kfill + 1u == dcnt);
            const u64 rmask = __ballot(last);
            if (gave_up) { if (lane == 0) atomicAdd(&A.ctl[CTL_ABORT], 1u); break; }      // a record never arrived: fail loudly
            const ull ph3 = STATS ? c2a_now() : 0;
            // ---- the node
            const u32 depth = ch == C2A_NONE ? 0u : ch_depth + 1;
            const u32 my_label = ch == C2A_NONE ? 0u : ch_el;
            max_level = level > max_level ? level : max_level;
            if (lane == 0) {
                A.meta[g] = make_uint4(ch, depth, ch_root, my_label | (level << 1));
                if (ch != C2A_NONE) A.child[2 * (u64)ch + my_label] = g;
            }
            u64 my_w;
            {
                const bool fresh = ch == C2A_NONE || ch_depth == 0 || chunk_of(depth) != chunk_of(ch_depth);
                const u32 bit = depth ? (depth - 1) - chunk_of(depth) * kChunkBits : 0u;
                const u32 cprev = chunk_of(depth) == 0 ? C2A_NONE : (fresh ? ch : (u32)rdlane64(ch_w, 1));
                const u32 wi = bit / kWordBits;
                // the parent's words of this chunk (bits past its end are zero by construction, lanes past them garbage)
                u64 w = (!fresh && lane - kHdrWords < str_words(bit)) ? (ch_w & kPayload) : 0ull;
                if (lane == 0) w = hdr0_word(ch_root, depth);
                else if (lane == 1) w = hdr1_word(level, cprev);
                else if (depth && lane - kHdrWords == wi) w |= (u64)my_label << (bit % kWordBits);
                my_w = w | tag;
                const u32 nw = kHdrWords + (depth ? wi + 1 : 0u);
                if (lane < nw) st_nw(&A.node[(u64)g * kNodeWords + lane], my_w);         // words past the end are never read
            }
            ++processed;
            if ((processed & 63u) == 0 && lane == 0) atomicAdd(&A.ctl[CTL_HEARTBEAT], 1u);
            if (STATS) {
                const ull ph4 = c2a_now();
                ++ph_steps;
                if (ph1 == 0) { ++ph_noload; ph1 = ph0; }
                ph_a += ph1 - ph0; ph_b += ph2 - ph1; ph_c += ph3 - ph2; ph_d += ph4 - ph3;
            }
            if (rmask == 0) break;                                  // the chain ends here
            // continue with the first claimed producer; a second one goes to whoever is idle
            const u32 j0 = ctz64(rmask);
            const u32 nxt = rdlane(dl, j0);
            if (rmask & (rmask - 1)) {
                const u32 other = rdlane(dl, 1);
                ++st_push;
                if (!peel_push(A, push_rr, lane, other, rdlane(gd.z, 1))) {
                    if (lane == 0) st_a32(&A.link[other], head);
                    head = other;
                }
            }
            // what the next step reuses: this gate as a candidate of nxt, and nxt's static data
            own_valid = true; own_node = g; own_label = j0; own_depth = depth; own_root = ch_root; own_level = level; own_w = my_w;
            g = nxt;
            gi = make_uint4(rdlane(gd.x, j0), rdlane(gd.y, j0), rdlane(gd.z, j0), rdlane(gd.w, j0));
            gi2 = make_uint4(rdlane(gd2.x, j0), rdlane(gd2.y, j0), rdlane(gd2.z, j0), rdlane(gd2.w, j0));
            if (gi.w <= 32u) cl = __shfl(clp, (int)((lane + 32u * j0) & 63u), 64);
            else cl = A.clist[gi.z + lane];
        }
        if ((++iters & 63u) == 0) {                                 // somebody gave up (watchdog): leave, the host reports it
            u32 ab = 0;
            if (lane == 0) ab = ld_a32(&A.ctl[CTL_ABORT]);
            if (rdlane(ab, 0)) break;
        }
    }
    if (lane == 0) {
        if (!registered) atomicAdd(&A.idle[(me % kIdleCounters) * 16], 1u);      // a wave that has left counts as idle for good
        if (processed) atomicAdd(&A.ctl[CTL_PROCESSED], processed);
        if (max_level) atomicMax(&A.ctl[CTL_MAXLEVEL], max_level);
        if (st_rpolls) atomicAdd(&A.ctl[CTL_REREADS], st_rpolls);
        if (STATS && A.stats) {
            if (STATS) { const ull t = c2a_now(); st_busy += t - st_t0; }
            atomicAdd(&A.stats[0], (ull)st_pops); atomicAdd(&A.stats[1], (ull)st_polls); atomicAdd(&A.stats[2], (ull)st_push);
            atomicAdd(&A.stats[3], st_busy); atomicAdd(&A.stats[4], st_idle); atomicAdd(&A.stats[5], (ull)st_seeds); atomicAdd(&A.stats[6], (ull)processed);
            atomicMax(&A.stats[7], st_busy);
            atomicAdd(&A.stats[8], (ull)st_rpolls);
            atomicAdd(&A.stats[9], ph_a); atomicAdd(&A.stats[10], ph_b); atomicAdd(&A.stats[11], ph_c); atomicAdd(&A.stats[12], ph_d);
            atomicAdd(&A.stats[13], ph_steps); atomicAdd(&A.stats[14], ph_noload); atomicAdd(&A.stats[15], ph_start);
        }
    }
}

}  // namespace c2a

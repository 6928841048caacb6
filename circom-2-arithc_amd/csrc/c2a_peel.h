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
//   word 2      where a child's edge label goes in the string: word index << 8 | bit (no division on the hot path)
//   words 3..63 the node's path string, ZERO-PADDED: 63 payload bits per word, bit j = label of the edge entering depth j+1
// A string is held in chunks of kChunkBits = 61 x 63 = 3843 bits; a node keeps its CURRENT chunk only.  Comparing two
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
// yet).  A single-consumer producer needs no ticket at all.  A second producer completed by the same gate is handed to a
// ticket queue; idle waves pop.  The chain loop is SOFTWARE-PIPELINED: the loads and tickets of the next step are issued
// as soon as this step's tickets say where the chain goes on, before this step's own tournament and stores.
// No deadlock: a ticket is only ever taken by a wave for a step that it then runs to the end, so when a gate is claimed
// every one of its candidates is in a step that is running (or done) on some resident wave; a wave only ever waits for
// such a record, the graph is acyclic, so every wait ends.
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
constexpr u32 kHdrWords = 3;
constexpr u32 kStrWords = kNodeWords - kHdrWords;                     // 61
constexpr u32 kWordBits = 63;
constexpr u32 kChunkBits = kStrWords * kWordBits;                     // 3843
constexpr u64 kTagBit = 1ull << 63;
constexpr u64 kPayload = kTagBit - 1ull;

constexpr u32 kIdleCounters = 64;
constexpr u32 kQStride = 16;                // u64 words between two queues' head/tail words (one queue per 128-byte line)
#ifdef C2A_EMULATE
constexpr u32 kPollLimit = 1;               // steps are atomic there: a missing record is a bug, fail at once
#else
constexpr u32 kPollLimit = 1u << 21;        // ~1 s of polling for a record: give up (reported as an error) instead of hanging
#endif
constexpr u32 kWatchdogChecks = 1u << 15;   // idle-side checks (one per 32 polls, ~100 us apart) without global progress
// control block (u32 words; every hot word on its own 64-byte line)
enum PeelCtl { CTL_PROCESSED = 0, CTL_MAXLEVEL = 1, CTL_ABORT = 2, CTL_REREADS = 3, CTL_HEARTBEAT = 16, CTL_STARTED = 32, CTL_SEEDNEXT = 48, CTL_WORDS = 64 };

// what only the edges of the launch touch (kept out of the kernel's scalar registers)
struct PeelCold {
    const u32* seeds;          // [n_regions][region_cap] producers claimed by the sinks pass
    const u32* seed_cnt;       // [n_regions]
    u32 n_regions, region_cap;
    ull* stats;                // optional diagnostics (32 words), nullptr normally
    ull* q_time;               // with stats: when the push of every queue entry was decided
};

struct PeelArgs {
    u32 epoch;                 // tag (0/1) of this run's node words
    u32 n_queues, q_cap;
    u32 n;
    const uint4* gstat;        // [2n] {dep0, dep1, cons_off, cons_cnt} {cons_off[dep0], cons_cnt[dep0], cons_off[dep1], cons_cnt[dep1]}
    const u32* clist;          // [edges + 64] consumer | edge label << 31, grouped by producer
    u64* node;                 // [n][64] node records
    u32* fill;                 // [n] claim tickets taken so far (zeroed per run)
    uint4* meta;               // [n] {parent | NONE, depth, root, label | level << 1}: read by later launches only
    u32* child;                // [2n] tree children by label (0xFF-filled per run)
    u64* q_ht;                 // [n_queues * kQStride] head (low word) | tail (high word)
    u64* q_items;              // [n_queues][q_cap] (gate + 1) | cons_off << 32; 0 = not written yet
    u32* idle;                 // [kIdleCounters * 16]
    u32* ctl;                  // [CTL_WORDS]
    u32* link;                 // [n] wave-private overflow stacks (queue full — never in practice)
    const PeelCold* cold;
    // the sinks pass only
    u32* seeds_w; u32* seed_cnt_w; u32 region_cap;
};

__device__ __forceinline__ u32 chunk_of(u32 depth) { return depth ? (depth - 1) / kChunkBits : 0u; }
__device__ __forceinline__ u32 chunk_len(u32 depth) { return depth - chunk_of(depth) * kChunkBits; }
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
// Where one lane does something and the wave then LEAVES a loop (break / return), the lanes must be seen to meet again
// first: otherwise the compiler threads the jump into both sides of the `if (lane == 0)`, the loop exit becomes a join of
// a divergent branch, and every value carried around that loop is handled as divergent (vector registers, masked code).
__device__ __forceinline__ void wave_join() {
#ifndef C2A_EMULATE
    __builtin_amdgcn_wave_barrier();
#endif
}
// A kernel argument in scalar registers OF ITS OWN.  The argument block arrives as one 16-register load; under register
// pressure the allocator spills and reloads that block as a unit — sixteen v_readlane for every pointer the chain step
// touches (measured: 315 -> 180 vector instructions per gate).  A copy through an opaque s_mov makes each pointer its own
// two-register value.  (The copy is made on a GLOBAL-address-space pointer and cast back: a generic pointer of unknown
// origin would turn every access through it into a flat_ instruction.)
template <class T> __device__ __forceinline__ T* own_sgprs(T* p) {
#ifdef C2A_EMULATE
    return p;
#else
    typedef __attribute__((address_space(1))) T* G;
    G q;
    asm volatile("s_mov_b64 %0, %1" : "=s"(q) : "s"((G)p));
    return (T*)q;
#endif
}
__device__ __forceinline__ u32 own_sgpr(u32 v) {
#ifdef C2A_EMULATE
    return v;
#else
    u32 q;
    asm volatile("s_mov_b32 %0, %1" : "=s"(q) : "s"(v));
    return q;
#endif
}
// the compiler must treat v as used (and redefined) here: pins the wait for a pending load to this point
#ifdef C2A_EMULATE
#define C2A_PIN(v) ((void)0)
#else
#define C2A_PIN(v) asm volatile("" : "+v"(v) :: "memory")
#endif

__device__ __forceinline__ ull ld_word_time(const ull* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 hdr0_word(u32 root, u32 depth) { return ((u64)(root & kIdMask) << 32) | depth; }
__device__ __forceinline__ u64 hdr1_word(u32 level, u32 cprev) { return ((u64)(level & kIdMask) << 32) | cprev; }
__device__ __forceinline__ u32 hdr_hi(u64 w) { return (u32)(w >> 32) & kIdMask; }

// ---- out-of-line pieces of the deep-tree comparison (cold; plain by-value arguments so that the kernel's argument
// block is never dragged into memory) ----
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
__device__ __forceinline__ u64 ld_rec_wait(const u64* node_base, u32 epoch, u32* ctl, u32 node, u32 lane) {
    const u64* p = node_base + (u64)node * kNodeWords + lane;
    u64 v = ld_nw(p);
    u32 spins = 0;
    while (__ballot((u32)(v >> 63) != epoch) != 0ull) {
        if (++spins > (1u << 22)) { if (lane == 0) atomicAdd(&ctl[CTL_ABORT], 1u); break; }
        peel_sleep(4);
        v = ld_nw(p);
    }
    return v;
}
// bit j of the string chunk held one word per lane
__device__ __forceinline__ u32 str_bit(u64 w, u32 j) { return (u32)(rdlane64(w, kHdrWords + j / kWordBits) >> (j % kWordBits)) & 1u; }
// P(a).la < P(b).lb by lengths (general form: chunks of any fill)
__device__ __forceinline__ bool str_less_len(u64 wa, u32 lena, u32 la, u64 wb, u32 lenb, u32 lb, u32 lane) {
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
// Trees deeper than one chunk: bring the two nodes to the first chunk in which their paths can differ (cprev hops),
// then compare that chunk.  wa_in / wb_in: the two nodes' own records (already in registers).
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
__global__ void k_gstat(u32 n, const u32* __restrict__ dep0, const u32* __restrict__ dep1, const u32* __restrict__ cons_off,
                        const u32* __restrict__ cons_cnt, const u32* __restrict__ eslot, uint4* gstat, u32* clist) {
    for (u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (u64)gridDim.x * blockDim.x) {
        const u32 d0 = dep0[g], d1 = dep1[g];
        gstat[2 * g] = make_uint4(d0, d1, cons_off[g], cons_cnt[g]);
        uint4 g2 = make_uint4(0, 0, 0, 0);
        if (d0 != C2A_NONE) { g2.x = cons_off[d0]; g2.y = cons_cnt[d0]; clist[g2.x + eslot[2 * g]] = (u32)g; }
        if (d1 != C2A_NONE) { g2.z = cons_off[d1]; g2.w = cons_cnt[d1]; clist[g2.z + eslot[2 * g + 1]] = (u32)g | 0x80000000u; }
        gstat[2 * g + 1] = g2;
    }
}

// ------------------------------------------------------------------------------------------------
// sinks (gates nobody consumes: DFS roots of depth 0, no candidates) — a plain grid-stride pass; the producers they
// claim seed the dataflow launch.  No shared counter: workgroup b appends to its own region under its own counter.
// A sink writes the three header words of its record only (its string is empty; readers treat depth 0 so).
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_peel_sinks(PeelArgs A) {
    __shared__ u32 s_done[4];
    const u32 lane = threadIdx.x & 63u;
    u32* out = A.seeds_w + (u64)blockIdx.x * A.region_cap;
    u32* counter = &A.seed_cnt_w[blockIdx.x];
    const u64 lt_mask = (1ull << lane) - 1ull;
    const u64 tag = A.epoch ? kTagBit : 0ull;
    u32 done = 0;
    for (u64 base = (u64)blockIdx.x * 256; base < A.n; base += (u64)gridDim.x * 256) {
        const u64 g = base + threadIdx.x;
        u32 rdy[2] = {C2A_NONE, C2A_NONE};
        if (g < A.n) {
            const uint4 gi = A.gstat[2 * g];
            if (gi.w == 0) {
                ++done;
                A.meta[g] = make_uint4(C2A_NONE, 0u, (u32)g, 0u);
                A.node[g * kNodeWords] = tag | hdr0_word((u32)g, 0u);
                A.node[g * kNodeWords + 1] = tag | hdr1_word(0u, C2A_NONE);
                A.node[g * kNodeWords + 2] = tag;
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
        if (t) atomicAdd(&A.ctl[CTL_PROCESSED], t);
    }
}

// everything issued for one gate at the top of its step; two of these swap roles (nothing is ever copied: a register
// copy of a value still in flight is a use, and its wait would drain the step that was just issued)
struct StepIO {
    u32 kfill;                 // lane l < 2: ticket taken on producer l
    uint4 ga, gb;              // lane l < 2: the two gstat records of producer l
    u32 clp;                   // lanes 0..31: producer 0's consumers, 32..63: producer 1's
    u32 cl;                    // this gate's consumers (its first block), one per lane from lane cbase on
    u32 cbase, ccap;           // first lane / lanes of that block
    u64 rest;                  // lanes of cl whose records are not loaded yet
    u32 e0, e1, take;          // consumer | label << 31 of the (up to two) records in flight
    u64 w0, w1;                // this lane's word of those records
};

// the dataflow launch: 64-thread workgroups (one wave each)
template <bool STATS>
__global__ void __launch_bounds__(64) k_peel(PeelArgs A_in) {
    PeelArgs A = A_in;
    A.gstat = own_sgprs(A_in.gstat); A.clist = own_sgprs(A_in.clist); A.node = own_sgprs(A_in.node); A.fill = own_sgprs(A_in.fill);
    A.meta = own_sgprs(A_in.meta); A.child = own_sgprs(A_in.child); A.q_ht = own_sgprs(A_in.q_ht); A.q_items = own_sgprs(A_in.q_items);
    A.idle = own_sgprs(A_in.idle); A.ctl = own_sgprs(A_in.ctl); A.link = own_sgprs(A_in.link); A.cold = own_sgprs(A_in.cold);
    A.epoch = own_sgpr(A_in.epoch); A.n_queues = own_sgpr(A_in.n_queues); A.q_cap = own_sgpr(A_in.q_cap);
    const u32 lane = threadIdx.x;
    const u32 me = blockIdx.x;
    const u32 epoch = A.epoch;
    const u64 tag = epoch ? kTagBit : 0ull;
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
    ull ph_w1 = 0, ph_w2 = 0, ph_w3 = 0, ho_sum = 0, ho_cnt = 0, ho_hist[5] = {0, 0, 0, 0, 0};
    ull ph_a = 0, ph_b = 0, ph_c = 0, ph_d = 0, ph_steps = 0, ph_noload = 0, ph_start = 0;      // STATS: phase times of the chain step
    if (lane == 0) atomicAdd(&A.ctl[CTL_STARTED], 1u); wave_join();
    for (;;) {
        // ---- next piece of work: own stack, the seed pool, then the hand-off queues
        u32 g = C2A_NONE;
        u32 coff = C2A_NONE;                 // cons_off of g when the entry carried it
        ull pop_when = 0;                    // STATS: push time of the popped entry
        if (head != C2A_NONE) {
            g = head;
            head = uniform(ld_a32(&A.link[g]));
        } else {
            if (seeds_left) {
                // (pointers read from memory are generic pointers, and a load through one counts as divergent: every value
                // read through A.cold is declared wave-uniform by hand)
                const PeelCold* C = A.cold;
                while (idx >= region_cnt) {
                    u32 r = 0;
                    if (lane == 0) r = atomicAdd(&A.ctl[CTL_SEEDNEXT], 1u); wave_join();
                    r = rdlane(r, 0);
                    if (r >= uniform(C->n_regions)) { seeds_left = false; break; }
                    region = r; idx = 0; region_cnt = uniform(C->seed_cnt[r]);
                }
                if (seeds_left) { g = uniform(C->seeds[(u64)region * uniform(C->region_cap) + idx]); ++idx; if (STATS) ++st_seeds; }
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
                    if (lane == 0) ht = ld_nw(&A.q_ht[(u64)q * kQStride]); wave_join();
                    ht = rdlane64(ht, 0);
                    const u32 qh = (u32)ht, qt_raw = (u32)(ht >> 32);
                    const u32 qt = qt_raw < A.q_cap ? qt_raw : A.q_cap;
                    if (qh < qt) {
                        if (registered) { if (lane == 0) atomicAdd(&A.idle[(me % kIdleCounters) * 16], 0xFFFFFFFFu); wave_join(); registered = false; }
                        // claim the head entry and read it in the same round trip (lane 1 reads what lane 0 claims)
                        u32 old = 0;
                        u64 v = 0;
                        if (lane == 0) old = atomicCAS(&head_w[2 * (u64)q * kQStride], qh, qh + 1); wave_join();
                        if (lane == 1) v = ld_nw(&A.q_items[(u64)q * A.q_cap + qh]); wave_join();
                        old = rdlane(old, 0);
                        v = rdlane64(v, 1);
                        if (old == qh) {
                            u32 spins = 0;
                            while (v == 0 && ++spins < (1u << 22)) {
                                if (lane == 0) v = ld_nw(&A.q_items[(u64)q * A.q_cap + qh]); wave_join();
                                v = rdlane64(v, 0);
                            }
                            if (v) {
                                g = (u32)v - 1u;
                                coff = (u32)(v >> 32);
                                if (STATS) pop_when = ld_word_time(A.cold->q_time + (u64)q * A.q_cap + qh);
                                if (STATS) ++st_pops;
                            } else { if (lane == 0) atomicAdd(&A.ctl[CTL_ABORT], 1u); wave_join(); }
                            break;
                        }
                        hint = q;                                    // lost the race: look at the same queue again at once
                        continue;
                    }
                    if (!registered) { if (lane == 0) atomicAdd(&A.idle[(me % kIdleCounters) * 16], 1u); wave_join(); registered = true; }
                    ++polls;
                    if ((polls & 31u) == 0) {
                        u32 cnt = lane < kIdleCounters ? ld_a32(&A.idle[lane * 16]) : 0u;
#pragma unroll
                        for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
                        cnt = uniform(cnt);          // (a shuffle result counts as divergent: the branches below must not)
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
                        else if (++hb_checks > kWatchdogChecks) { if (lane == 0) atomicAdd(&A.ctl[CTL_ABORT], 1u); wave_join(); break; }
                    }
                    // back off: the longer nothing turns up, the less often this wave asks (64 clocks per unit, <= ~3 us)
                    peel_sleep(polls < 16 ? 4 : 16);
                }
                if (STATS) st_polls += polls;
                if (STATS) { const ull t = c2a_now(); st_idle += t - st_t0; st_t0 = t; }
                if (g == C2A_NONE) break;
            }
        }
        // (wave-uniform by construction — say so: one value the compiler takes for divergent here, and every value of the
        // chain loop that depends on the gate id moves to vector registers and is handled as divergent code)
        g = uniform(g); coff = uniform(coff);
        const ull ph_s0 = STATS ? c2a_now() : 0;
        // ---- static data of g (a chain step gets all of this prefetched by the step before); consumed HERE, in scalar
        // registers where wave-uniform: a load still pending at the loop header would cost every chain step a wait
        uint4 gi = uniform4(A.gstat[2 * (u64)g]);
        uint4 gi2 = uniform4(A.gstat[2 * (u64)g + 1]);
        if (coff == C2A_NONE) coff = gi.z;
        u32 cl0 = A.clist[coff + lane];                                  // clist is padded by 64 entries
        C2A_PIN(cl0);
        if (STATS) {
            const ull tn = c2a_now();
            ph_start += tn - ph_s0;
            if (pop_when) { const ull dt = tn - pop_when; ho_sum += dt; ++ho_cnt; ho_hist[dt < 200 ? 0 : dt < 400 ? 1 : dt < 800 ? 2 : dt < 1600 ? 3 : 4] += 1; }
        }

        // issue everything the step of a gate needs from memory
        auto issue = [&](StepIO& S, u32 dep0, u32 dep1, u32 n_cons, u32 off0, u32 cnt0, u32 off1, u32 cnt1, u32 scl, u32 cbase, u32 ccap,
                         bool have_own, u32 own_id) {
            const u32 dl = lane == 0 ? dep0 : (lane == 1 ? dep1 : C2A_NONE);
            const u32 dcnt = lane == 0 ? cnt0 : (lane == 1 ? cnt1 : 0u);
            S.kfill = 0;
            if (dl != C2A_NONE && dcnt > 1u) S.kfill = atomicAdd(&A.fill[dl], 1u);
            // static data of both producers, BRANCH-FREE (clamped index, result discarded by lanes with nothing to load)
            const u32 dl_c = dl != C2A_NONE ? dl : 0u;
            S.ga = A.gstat[2 * (u64)dl_c];
            S.gb = A.gstat[2 * (u64)dl_c + 1];
            {
                const u32 half = lane >> 5, i = lane & 31u;
                const u32 poff = half ? off1 : off0, pcnt = half ? cnt1 : cnt0;
                S.clp = A.clist[poff + (i < pcnt ? i : 0u)];
            }
            const u32 n_here = n_cons < ccap ? n_cons : ccap;
            S.cl = scl; S.cbase = cbase; S.ccap = ccap;
            u64 smask = __ballot(lane - cbase < n_here && !(have_own && (scl & kIdMask) == own_id));
            S.take = 0; S.e0 = 0; S.e1 = 0;
            if (smask) {
                S.e0 = rdlane(scl, ctz64(smask)); smask &= smask - 1; S.take = 1;
                if (smask) { S.e1 = rdlane(scl, ctz64(smask)); smask &= smask - 1; S.take = 2; }
            }
            S.rest = smask;
            // the records (slots that are not loaded stay UNDEFINED on purpose: merging a loaded value with a constant is
            // a register copy, a copy is a use, and its wait would land right behind the load)
            if (S.take >= 1) S.w0 = ld_nw(&A.node[(u64)(S.e0 & kIdMask) * kNodeWords + lane]);
            if (S.take >= 2) S.w1 = ld_nw(&A.node[(u64)(S.e1 & kIdMask) * kNodeWords + lane]);
        };
        StepIO S0, S1;
        issue(S0, gi.x, gi.y, gi.w, gi2.x, gi2.y, gi2.z, gi2.w, cl0, 0u, 64u, false, 0u);
        // the gate just finished stays in registers as a candidate of the next one
        bool own_valid = false;
        u32 own_node = 0, own_label = 0, own_depth = 0, own_root = 0, own_level = 0, own_pos = 0;
        u64 own_w = 0, own_x = 0;            // its record, and its string with own_label appended
        // a second claimed producer's queue ticket is taken at the top of a step and its entry written one step later
        // (the ticket is back by then: nothing in between may wait for memory)
        bool push_pending = false;
        u32 push_q = 0, push_t = 0, push_gate = 0, push_off = 0;
        ull push_when = 0;                   // STATS: when the pending push was decided
        auto write_entry = [&](u32 q, u32 t, u32 gate, u32 off) {
            if (STATS && t < A.q_cap && lane == 0) A.cold->q_time[(u64)q * A.q_cap + t] = push_when;
            if (t < A.q_cap) { if (lane == 0) st_nw(&A.q_items[(u64)q * A.q_cap + t], (u64)(gate + 1u) | ((u64)off << 32)); wave_join(); }
            else { if (lane == 0) st_a32(&A.link[gate], head); wave_join(); head = gate; }      // queue full: cannot happen with the host's q_cap
        };

        // one step: `cur` is in hand (issued one step ago), `nx` receives the next one.  true = the chain ends (or abort)
        auto step = [&](StepIO& cur, StepIO& nx) -> bool {
            const ull ph0 = STATS ? c2a_now() : 0;
            // ---- everything of THIS step (issued one step ago) is needed now, and is pinned HERE: a register of `cur` that
            // the compiler still counts as in flight further down would put its wait behind the issue of the next step
            C2A_PIN(cur.kfill);
            const ull ph0a = STATS ? c2a_now() : 0;
            C2A_PIN(cur.ga.x); C2A_PIN(cur.ga.y); C2A_PIN(cur.ga.z); C2A_PIN(cur.ga.w);
            C2A_PIN(cur.gb.x); C2A_PIN(cur.gb.y); C2A_PIN(cur.gb.z); C2A_PIN(cur.gb.w);
            C2A_PIN(cur.clp);
            const ull ph0b = STATS ? c2a_now() : 0;
            C2A_PIN(cur.w0); C2A_PIN(cur.w1);
            if (STATS) { const ull ph0c = c2a_now(); ph_w1 += ph0a - ph0; ph_w2 += ph0b - ph0a; ph_w3 += ph0c - ph0b; }
            const bool old_pending = push_pending;
            const u32 old_q = push_q, old_t = old_pending ? rdlane(push_t, 0) : 0u, old_gate = push_gate, old_off = push_off;
            push_pending = false;
            const u32 g_dep0 = gi.x, g_dep1 = gi.y, g_off = gi.z, g_cnt = gi.w, g_dcnt0 = gi2.y, g_dcnt1 = gi2.w;
            const u32 dl = lane == 0 ? g_dep0 : (lane == 1 ? g_dep1 : C2A_NONE);
            const u32 dcnt = lane == 0 ? g_dcnt0 : (lane == 1 ? g_dcnt1 : 0u);
            const bool last = dl != C2A_NONE && (dcnt == 1u || cur.kfill + 1u == dcnt);
            const u32 rmask = (u32)__ballot(last) & 3u;
            const ull ph1 = STATS ? c2a_now() : 0;
            // ---- go on with the first claimed producer: issue its step now; a second one goes to whoever is idle
            u32 nxt = C2A_NONE, nxt_label = 0;
            uint4 ngi = make_uint4(0, 0, 0, 0), ngi2 = make_uint4(0, 0, 0, 0);
            if (rmask) {
                const u32 j0 = (rmask & 1u) ? 0u : 1u;
                nxt = j0 ? g_dep1 : g_dep0; nxt_label = j0;
                ngi = make_uint4(rdlane(cur.ga.x, j0), rdlane(cur.ga.y, j0), rdlane(cur.ga.z, j0), rdlane(cur.ga.w, j0));
                ngi2 = make_uint4(rdlane(cur.gb.x, j0), rdlane(cur.gb.y, j0), rdlane(cur.gb.z, j0), rdlane(cur.gb.w, j0));
                if (rmask == 3u) {
                    push_pending = true;
                    if (STATS) push_when = ph0;
                    push_gate = g_dep1; push_off = rdlane(cur.ga.z, 1);
                    push_q = (push_rr++) % A.n_queues;
                    if (STATS) ++st_push;
                    push_t = 0;
                    if (lane == 0) push_t = atomicAdd(&head_w[2 * (u64)push_q * kQStride + 1], 1u); wave_join();
                }
                // the consumers of nxt are already here (lanes 32 j0 ... of the prefetched lists) unless it has more than 32
                // of them: then none is loaded ahead and the cold loop below reads the list itself
                issue(nx, ngi.x, ngi.y, ngi.w, ngi2.x, ngi2.y, ngi2.z, ngi2.w, cur.clp, 32u * j0, ngi.w <= 32u ? 32u : 0u, true, g);
            }
            if (old_pending) write_entry(old_q, old_t, old_gate, old_off);
            const ull ph2 = STATS ? c2a_now() : 0;
            // ---- tournament of THIS gate.  Champion so far (wave-uniform); NONE = the virtual-root candidate [g]
            u32 ch = C2A_NONE, ch_el = 0, ch_root = g, ch_depth = 0, ch_pos = 0, level = 0;
            u64 ch_w = 0, ch_x = 0;              // the champion's record / its string with the edge label appended
            if (own_valid) {
                level = own_level + 1;
                if (own_root < g) { ch = own_node; ch_el = own_label; ch_root = own_root; ch_depth = own_depth; ch_pos = own_pos; ch_w = own_w; ch_x = own_x; }
            }
            bool gave_up = false;
            // one candidate: its record must be all there (else read it again: cold), then it meets the champion
            auto candidate = [&](u64 w, u32 e) {
                const u32 c = e & kIdMask, el = e >> 31;
                u64 badm = __ballot((u32)(w >> 63) != epoch);
                if (badm) {
                    // a sink's record has header words only; anything else is still being written
                    u32 polls = 0;
                    for (;;) {
                        if ((badm & 7ull) == 0 && (u32)rdlane64(w, 0) == 0u) { if (lane >= kHdrWords) w = tag; break; }
                        if (++polls > kPollLimit) { gave_up = true; return; }
                        peel_sleep(polls < 8 ? 4 : 16);
                        w = ld_nw(&A.node[(u64)c * kNodeWords + lane]);
                        badm = __ballot((u32)(w >> 63) != epoch);
                        if (!badm) break;
                    }
                    st_rpolls += polls;
                }
                const u64 h0 = rdlane64(w, 0);
                const u32 croot = hdr_hi(h0), cdepth = (u32)h0;
                const u32 clevel = (rdlane((u32)(w >> 32), 1) & kIdMask) + 1u;
                const u32 cpos = rdlane((u32)w, 2);
                level = clevel > level ? clevel : level;
                // the candidate's string with its edge label appended (meaningful while the chunk has room: cpos < kStrWords << 8)
                u64 x = w & kPayload;
                if (lane == kHdrWords + (cpos >> 8)) x |= (u64)el << (cpos & 255u);
                bool less;
                if (croot != ch_root) less = croot < ch_root;              // a larger DFS root loses at once (also to [g] itself)
                else if (ch == C2A_NONE) less = false;                     // (root == g: impossible in a DAG)
                else if (c == ch) less = el < ch_el;
                else if (cdepth < kChunkBits && ch_depth < kChunkBits) {
                    // neither path is a prefix of the other (that would be a cycle): the first differing bit decides
                    const u64 d = x ^ ch_x;
                    const u64 bal = __ballot(d != 0) & ~7ull;
                    const u32 L = ctz64(bal);
                    less = ((rdlane64(x, L) >> ctz64(rdlane64(d, L))) & 1ull) == 0;
                } else {
                    // (the result of an out-of-line call counts as divergent; left like that, every value that depends on the
                    // champion would move to vector registers and the whole tournament would be compiled as divergent code)
                    less = uniform(deep_less(A.node, epoch, A.ctl, c, el, cdepth, w, ch, ch_el, ch_depth, ch_w, lane) ? 1u : 0u) != 0u;
                }
                if (less) { ch = c; ch_el = el; ch_root = croot; ch_depth = cdepth; ch_pos = cpos; ch_w = w; ch_x = x; }
            };
            // the (up to two) records loaded ahead, then — cold — whatever else the consumer list holds, one at a time:
            // more than two other consumers (cur.rest), or a list that was not prefetched (cur.ccap < 64: blocks from eb on)
            {
                u32 k = 0, blk = cur.cl, eb = cur.ccap == 64u ? 64u : 0u;
                u64 smask = cur.rest;
                const bool more_blocks = g_cnt > cur.ccap;
                for (;;) {
                    u64 w;
                    u32 e;
                    if (k < cur.take) {
                        w = k ? cur.w1 : cur.w0; e = k ? cur.e1 : cur.e0;
                    } else {
                        if (smask == 0) {
                            if (!more_blocks || eb >= g_cnt) break;
                            blk = A.clist[g_off + eb + lane];
                            C2A_PIN(blk);                        // (consumed here, like w below)
                            smask = __ballot(eb + lane < g_cnt && !(own_valid && (blk & kIdMask) == own_node));
                            eb += 64;
                            if (smask == 0) continue;
                        }
                        e = rdlane(blk, ctz64(smask));
                        smask &= smask - 1;
                        w = ld_nw(&A.node[(u64)(e & kIdMask) * kNodeWords + lane]);
                        C2A_PIN(w);                              // (consumed here: pending at the join it would cost the hot path a wait)
                    }
                    candidate(w, e);
                    if (gave_up) break;
                    ++k;
                }
            }
            if (gave_up) { if (lane == 0) atomicAdd(&A.ctl[CTL_ABORT], 1u); wave_join(); return true; }      // a record never arrived: fail loudly
            const ull ph3 = STATS ? c2a_now() : 0;
            // ---- the node: its string is the champion's string with the label appended — the register built above
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
            if (lane == 0) {
                A.meta[g] = make_uint4(ch, depth, ch_root, my_label | (level << 1));
                if (ch != C2A_NONE) A.child[2 * (u64)ch + my_label] = g;
            }
            wave_join();
            u64 my_w = str;
            if (lane == 0) my_w = hdr0_word(ch_root, depth);
            else if (lane == 1) my_w = hdr1_word(level, cprev);
            else if (lane == 2) my_w = my_pos;
            my_w |= tag;
            st_nw(&A.node[(u64)g * kNodeWords + lane], my_w);
            ++processed;
            if ((processed & 63u) == 0 && lane == 0) atomicAdd(&A.ctl[CTL_HEARTBEAT], 1u);
            if (STATS) {
                const ull ph4 = c2a_now();
                ++ph_steps;
                if (cur.take == 0) ++ph_noload;
                ph_a += ph1 - ph0; ph_b += ph2 - ph1; ph_c += ph3 - ph2; ph_d += ph4 - ph3;
            }
            if (nxt == C2A_NONE) return true;                       // the chain ends here
            // what the next step reuses: this gate as a candidate of nxt
            own_valid = true; own_node = g; own_label = nxt_label; own_depth = depth; own_root = ch_root; own_level = level; own_pos = my_pos; own_w = my_w;
            own_x = str;
            if (lane == kHdrWords + (my_pos >> 8)) own_x |= (u64)nxt_label << (my_pos & 255u);
            g = nxt; gi = ngi; gi2 = ngi2;
            return false;
        };
        for (;;) {
            if (step(S0, S1)) break;
            if (step(S1, S0)) break;
        }
        if (push_pending) write_entry(push_q, rdlane(push_t, 0), push_gate, push_off);
        if ((++iters & 63u) == 0) {                                 // somebody gave up (watchdog): leave, the host reports it
            u32 ab = 0;
            if (lane == 0) ab = ld_a32(&A.ctl[CTL_ABORT]); wave_join();
            if (rdlane(ab, 0)) break;
        }
    }
    if (lane == 0) {
        if (!registered) atomicAdd(&A.idle[(me % kIdleCounters) * 16], 1u);      // a wave that has left counts as idle for good
        if (processed) atomicAdd(&A.ctl[CTL_PROCESSED], processed);
        if (max_level) atomicMax(&A.ctl[CTL_MAXLEVEL], max_level);
        if (st_rpolls) atomicAdd(&A.ctl[CTL_REREADS], st_rpolls);
        if (STATS) {
            ull* stats = A.cold->stats;
            if (stats) {
                const ull t = c2a_now(); st_busy += t - st_t0;
                atomicAdd(&stats[0], (ull)st_pops); atomicAdd(&stats[1], (ull)st_polls); atomicAdd(&stats[2], (ull)st_push);
                atomicAdd(&stats[3], st_busy); atomicAdd(&stats[4], st_idle); atomicAdd(&stats[5], (ull)st_seeds); atomicAdd(&stats[6], (ull)processed);
                atomicAdd(&stats[7], ph_w1 | (ph_w2 << 32)); atomicAdd(&stats[16], ph_w3);
                atomicAdd(&stats[8], (ull)st_rpolls);
                atomicAdd(&stats[9], ph_a); atomicAdd(&stats[10], ph_b); atomicAdd(&stats[11], ph_c); atomicAdd(&stats[12], ph_d);
                atomicAdd(&stats[17], ho_sum); atomicAdd(&stats[18], ho_cnt);
                for (int k = 0; k < 5; ++k) atomicAdd(&stats[19 + k], ho_hist[k]);
                atomicAdd(&stats[13], ph_steps); atomicAdd(&stats[14], ph_noload); atomicAdd(&stats[15], ph_start);
            }
        }
    }
}

}  // namespace c2a

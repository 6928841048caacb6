// c2a_kernels.h — device code of the flat-gate-graph stage, written for gfx950 (wave64, HBM-bound
// integer/indexing work: no MFMA anywhere on this path).
//
// What each group replaces in the reference (all paths relative to /root/reference):
//   graph prep      : the producer map + deps closure           src/compiler.rs:401-421
//   peel + paths    : (with order kernels) topological_sort      src/topological_sort.rs:3-50
//   order           : DFS post-order numbering                   src/topological_sort.rs:46 (sorted.push)
//   wires + emit    : first-seen wire numbering, gate emission   src/compiler.rs:423-464
//   boolify         : boolify(&circuit, width)                   src/main.rs:30-32 (crate absent: frozen spec)
//
// The exact DFS post-order is reproduced without a DFS: the DFS tree parent of a gate is the
// consumer that reaches it by the lexicographically smallest path from a virtual root (children = gates in id
// order; edge labels 0 = lh producer, 1 = rh producer).  A gate picks its parent once all its consumers have picked
// theirs (reverse Kahn order from the sinks) by comparing the consumers' paths, held as 512-byte bit strings (one
// memory round trip per comparison).  By default the whole peel is ONE dataflow launch (k_peel_async: a wave goes on
// with the producer its ticket completed, ticket queues hand over the rest, no level barrier); a launch-per-level
// variant is kept for A/B.  The post-order index of every gate then comes from an Euler tour + list ranking.
#pragma once
#include "c2a_platform.h"

namespace c2a {

constexpr int kThreads = 256;

__device__ __forceinline__ u64 gtid() { return (u64)blockIdx.x * blockDim.x + threadIdx.x; }
__device__ __forceinline__ u64 gstride() { return (u64)gridDim.x * blockDim.x; }

// ------------------------------------------------------------------------------------------------
// exclusive scan (u32 in -> TOut out), tile = 256 threads x 8 items, hierarchical over tile partials
// ------------------------------------------------------------------------------------------------
constexpr int kScanItems = 8;
constexpr int kScanTile = kThreads * kScanItems;

template <typename TIn, typename TOut>
__global__ void __launch_bounds__(kThreads) k_scan_tile(const TIn* in, TOut* out, TOut* partials, u64 n) {
    __shared__ TOut sh[kThreads];
    const u32 tid = threadIdx.x;
    const u64 base = (u64)blockIdx.x * kScanTile + (u64)tid * kScanItems;
    TOut v[kScanItems];
    TOut sum = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        const u64 idx = base + i;
        const TOut x = idx < n ? (TOut)in[idx] : (TOut)0;
        v[i] = sum;
        sum += x;
    }
    sh[tid] = sum;
    __syncthreads();
    for (u32 off = 1; off < (u32)kThreads; off <<= 1) {
        const TOut t = tid >= off ? sh[tid - off] : (TOut)0;
        __syncthreads();
        sh[tid] += t;
        __syncthreads();
    }
    const TOut excl = sh[tid] - sum;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        const u64 idx = base + i;
        if (idx < n) out[idx] = excl + v[i];
    }
    if (tid == kThreads - 1) partials[blockIdx.x] = sh[tid];
}

template <typename TOut>
__global__ void k_scan_add(TOut* out, const TOut* partials, u64 n) {
    for (u64 i = gtid(); i < n; i += gstride()) out[i] += partials[i / kScanTile];
}

// out[n] = grand total (the scanned array has n+1 entries)
template <typename TOut>
__global__ void k_scan_total(TOut* out_n, const TOut* total) {
    if (gtid() == 0) *out_n = *total;
}

// ------------------------------------------------------------------------------------------------
// graph prep
// ------------------------------------------------------------------------------------------------
// producer[node] = last gate writing it (compiler.rs:403-406: later insert overwrites) -> max gate id.
// prod1 holds gate id + 1 (0 = no producer); must be zeroed.
// *dup is raised when two gates write one node (the reference keeps the last writer, compiler.rs:403-406): the wire
// numbering then takes its general path (first-seen by atomicMin over every reference)
__global__ void k_producer(u32 n, const u32* __restrict__ out, u32* prod1, u32* dup) {
    for (u64 g = gtid(); g < n; g += gstride())
        if (atomicMax(&prod1[out[g]], (u32)g + 1) != 0) *dup = 1u;
}

// deps closure (compiler.rs:408-421) + consumer counts.  dep1 is dropped when equal to dep0: a second
// visit of the same gate is a no-op in the DFS (topological_sort.rs:30-32).
__global__ void k_deps(u32 n, const u32* __restrict__ lh, const u32* __restrict__ rh, const u32* __restrict__ prod1,
                       u32* dep0, u32* dep1, u32* cons_cnt, u32* eslot) {
    for (u64 g = gtid(); g < n; g += gstride()) {
        const u32 p0 = prod1[lh[g]], p1 = prod1[rh[g]];
        const u32 d0 = p0 ? p0 - 1 : C2A_NONE;
        u32 d1 = p1 ? p1 - 1 : C2A_NONE;
        if (d1 == d0) d1 = C2A_NONE;
        dep0[g] = d0;
        dep1[g] = d1;
        // eslot[2g + l] = index of the edge (g, l) in its producer's consumer list: a static home for the candidate record
        eslot[2 * g] = d0 != C2A_NONE ? atomicAdd(&cons_cnt[d0], 1u) : 0u;
        eslot[2 * g + 1] = d1 != C2A_NONE ? atomicAdd(&cons_cnt[d1], 1u) : 0u;
    }
}

// One atomic per workgroup instead of one per wave on a hot append counter (a single address takes
// ~12 ns per atomic: MI355X_MICROARCH.md price list, row "fanin").  Must be called by every thread of a
// 256-thread workgroup the same number of times; returns the slot of the threads that `want` one.
__device__ __forceinline__ u32 block_append_slot(bool want, u32* counter) {
    __shared__ u32 s_cnt[kThreads / 64];
    __shared__ u32 s_base;
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const u64 mask = __ballot(want);
    if (lane == 0) s_cnt[wv] = (u32)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 run = 0;
        for (int w = 0; w < kThreads / 64; ++w) { const u32 t = s_cnt[w]; s_cnt[w] = run; run += t; }
        s_base = run ? atomicAdd(counter, run) : 0u;
    }
    __syncthreads();
    const u32 slot = s_base + s_cnt[wv] + (u32)__popcll(mask & ((1ull << lane) - 1ull));
    __syncthreads();
    return slot;
}

// per-gate static record {dep0, dep1, cons_off, cons_cnt}: one 16-byte load instead of four
__global__ void k_ginfo(u32 n, const u32* __restrict__ dep0, const u32* __restrict__ dep1,
                        const u32* __restrict__ cons_off, const u32* __restrict__ cons_cnt, uint4* ginfo) {
    for (u64 g = gtid(); g < n; g += gstride()) ginfo[g] = make_uint4(dep0[g], dep1[g], cons_off[g], cons_cnt[g]);
}

// ------------------------------------------------------------------------------------------------
// The frontier of the reverse Kahn peel.
// Frontier slot: 32 bytes {gate, dep0, dep1, cons_off} {cons_cnt, -, -, -}.  Two slot arrays alternate by level
// parity, and a slot is at an address that does not depend on the level's size, so a level's first hop (the slot)
// does not wait for the counts: both are loaded together.
// The frontier is SEGMENTED: kSeg independent append counters and slot regions.  One counter for the whole level was
// the peel's bottleneck at ~2 000 gates per level: ~290 workgroup tickets queue on one word at ~12 ns each (memory-side
// atomics), ~3 us of an 8 us level.  With 16 segments a counter sees ~18 tickets.  A level's gates are the
// concatenation of its segments (tree position = fbase[level] + segment prefix + index), every kernel recomputes the
// 16-entry prefix from the counts.  Who appends where is chosen so that no segment can overflow n/8 + 1024 slots
// whatever the graph: a workgroup class (1/16 of a level's gates, +- one workgroup) feeds one segment and a gate
// appends at most two producers.  Counters live in a ring of kRing levels: level L reads row L % kRing, adds to
// row (L+1) % kRing and clears row (L+2) % kRing; level sizes for later stages come from fbase[].
// ------------------------------------------------------------------------------------------------
struct FrontierSlot { uint4 a, b; };
constexpr u32 kSeg = 16;
constexpr u32 kRing = 4;
constexpr u32 kSlotPad = 4096;             // slots past a segment's end that a speculative slot load may touch
__host__ __device__ inline u32 seg_capacity(u32 n) { return n / 8 + 1024 + kSlotPad; }

// level 0: gates nobody consumes.  Ring row 0 must be zero.  gridDim.x % kSeg == 0.
__global__ void __launch_bounds__(kThreads) k_init_frontier(u32 n, u32 seg_cap, const uint4* __restrict__ ginfo, FrontierSlot* slots,
                                                            u32* fring) {
    const u32 seg = blockIdx.x % kSeg;
    FrontierSlot* out = slots + (u64)seg * seg_cap;
    for (u64 base = (u64)blockIdx.x * kThreads; base < n; base += (u64)gridDim.x * kThreads) {
        const u64 g = base + threadIdx.x;
        uint4 gi = make_uint4(0, 0, 0, 1);
        if (g < n) gi = ginfo[g];
        const bool sink = g < n && gi.w == 0;
        const u32 p = block_append_slot(sink, &fring[seg]);
        if (sink) { out[p].a = make_uint4((u32)g, gi.x, gi.y, gi.z); out[p].b = make_uint4(gi.w, 0, 0, 0); }
    }
}

// totals after the peel: {gates peeled, non-empty levels} from the level boundaries
__global__ void k_peel_totals(const u32* __restrict__ fbase, u32 n_levels, u32* out2) {
    __shared__ u32 s_lv[kThreads];
    u32 lv = 0;
    for (u32 i = threadIdx.x; i < n_levels; i += kThreads) lv += fbase[i + 1] != fbase[i] ? 1u : 0u;
    s_lv[threadIdx.x] = lv;
    __syncthreads();
    for (u32 off = kThreads / 2; off; off >>= 1) {
        if (threadIdx.x < off) s_lv[threadIdx.x] += s_lv[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) { out2[0] = fbase[n_levels]; out2[1] = s_lv[0]; }
}

// ------------------------------------------------------------------------------------------------
// peel one level + pick DFS-tree parents
// ------------------------------------------------------------------------------------------------
// Tree node == peel position: nodes of recent levels are contiguous in every per-node table.
// meta[pos] = {parent pos | NONE, depth, root gate id, label of the edge parent->node | reverse Kahn level << 1}.
// Candidate lists are filled as consumers are peeled, AFTER their tournament, with everything a comparison
// starts from: cand[cons_off[d] + k] = {consumer pos | edge label << 31, consumer depth, consumer root, consumer's
// own edge label}; the push that completes d's list (k + 1 == cons_cnt[d]) appends d to the next frontier.
constexpr u32 kIdMask = 0x7FFFFFFFu;

struct PeelArgs {
    u32 n;
    u32 seg_cap;               // slots per frontier segment
    const uint4* ginfo;        // [n] {dep0, dep1, cons_off, cons_cnt}
    FrontierSlot* slots[2];    // [kSeg][seg_cap] by level parity
    uint4* cand;               // [edges]
    u32* fill;                 // pushes so far per gate (zeroed)
    uint4* meta;               // [n] by position
    u64* pstr;                 // [n][64] path strings by position
    u32* cprev;                // [n] ancestor at the start of the node's current chunk (only beyond depth 4096)
    u32* order;                // position -> gate
    u32* child;                // [2n] tree children by label: child[2*p + l] (written as each gate picks its parent)
    u32* posof;                // gate -> position
    u32* fbase;                // [levels+2] first position of each level
    u32* fring;                // [kRing][kSeg] frontier segment sizes
    ull* prof;                 // optional phase timestamps (diagnostics; nullptr normally)
    u32 prof_level0;           // first level recorded
};

// Two access flavours.  SC1 = false: plain loads/stores.  SC1 = true: relaxed agent-scope atomics
// (global_load/store ... sc1), for words that other workgroups of the same launch may be writing.
template <bool SC1> __device__ __forceinline__ u32 ld_u32(const u32* p) {
    if (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool SC1> __device__ __forceinline__ void st_u32(u32* p, u32 v) {
    if (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}

// stores of the per-level kernels (experiment switch C2A_ST: 0 plain, 1 nontemporal, 2 agent-scope write-through)
#ifndef C2A_ST
#define C2A_ST 0
#endif
typedef unsigned v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_g(u32* p, u32 v) {
#if defined(C2A_EMULATE) || C2A_ST == 0
    *p = v;
#elif C2A_ST == 1
    __builtin_nontemporal_store(v, p);
#else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void st_g(u64* p, u64 v) {
#if defined(C2A_EMULATE) || C2A_ST == 0
    *p = v;
#elif C2A_ST == 1
    __builtin_nontemporal_store(v, p);
#else
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}
__device__ __forceinline__ void st_g(uint4* p, const uint4& v) {
#if defined(C2A_EMULATE) || C2A_ST == 0
    *p = v;
#elif C2A_ST == 1
    v4u t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<v4u*>(p));
#else
    u64* q = reinterpret_cast<u64*>(p);
    __hip_atomic_store(q, (u64)v.x | ((u64)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, (u64)v.z | ((u64)v.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

__device__ __forceinline__ u32 wave_min_u32(u32 v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const u32 o = __shfl_xor(v, off, 64);
        v = o < v ? o : v;
    }
    return v;
}

// lanes of one wave exchange data through LDS: the hardware runs a wave's DS ops in order, the barrier
// only has to stop the compiler (and gives the host emulation its rendezvous point)
__device__ __forceinline__ void wave_lds_sync() {
#ifdef C2A_EMULATE
    (void)__ballot(1);
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

__device__ __forceinline__ ull c2a_now() {
#ifdef C2A_EMULATE
    return 0;
#else
    return wall_clock64();      // constant 100 MHz
#endif
}
// diagnostics (PROF instantiation only): every wave of kProfLevels levels stores its phase timestamps
constexpr u32 kProfLevels = 32, kProfWaves = 32768;
#define C2A_PROF(slot, value)                                                                                   \
    do {                                                                                                        \
        if (PROF && A.prof && lane == 0 && level >= A.prof_level0 && level < A.prof_level0 + kProfLevels) {     \
            const u32 wg_ = blockIdx.x * WPB + wv;                                                              \
            if (wg_ < kProfWaves) A.prof[((u64)(level - A.prof_level0) * kProfWaves + wg_) * 8 + (slot)] = (value); \
        }                                                                                                       \
    } while (0)

// This level's segment sizes -> exclusive prefix (lanes 0..kSeg-1 hold count and prefix of their segment).
// Whole-wave call.  `total` is wave-uniform.
__device__ __forceinline__ void level_segments(const u32* __restrict__ fring, u32 level, u32 lane, u32& my_cnt, u32& my_pre, u32& total) {
    my_cnt = lane < kSeg ? fring[(level % kRing) * kSeg + lane] : 0u;
    u32 inc = my_cnt;
#pragma unroll
    for (int off = 1; off < (int)kSeg; off <<= 1) {
        const u32 o = __shfl_up(inc, off, 64);
        if (lane >= (u32)off) inc += o;
    }
    my_pre = inc - my_cnt;
    total = __shfl(inc, (int)kSeg - 1, 64);
}

// ================================================================================================
// PATH STRINGS — the representation of "the path from the DFS root to a tree node".
// Every tree node stores the edge labels of its path as a bit string: bit j = label of the edge entering depth
// j+1.  A string is held in chunks of kChunkBits = 4096 bits = 512 B = one coalesced 8-byte load per lane of a wave;
// a node keeps only its CURRENT chunk (bits [ci*K, depth), words past the end unwritten and never read) plus
// cprev = its ancestor at depth ci*K, whose own string is the complete previous chunk — so storage is 512 B per node
// whatever the depth, and for trees shallower than 4096 (the 10 M-gate headline config: 3 471) comparing two
// candidates is: load both strings (one round trip, coalesced), XOR, ballot, count trailing zeros.  Deeper trees
// add one cprev hop per chunk level.  A new node's string = parent's string + one bit: ONE hop to build.
// (Round-1 history: base-16 / base-64 ancestor rows with lift + diverge needed <= 6 dependent hops per comparison
// and 3 per new node — peel 66 ms vs 44 ms at 10 M gates; a persistent single-XCD launch with L2-level hand-off
// was correct but issue-bound on 32 CUs: 10.3 us per level vs 8.2.  Both are in the git history)
// ================================================================================================
constexpr u32 kChunkBits = 4096;
constexpr u32 kChunkWords = kChunkBits / 64;

__device__ __forceinline__ u32 chunk_of(u32 depth) { return depth ? (depth - 1) / kChunkBits : 0u; }
__device__ __forceinline__ u32 chunk_len(u32 depth) { return depth - chunk_of(depth) * kChunkBits; }
__device__ __forceinline__ u32 ctz64(u64 x) { return (u32)__ffsll((long long)x) - 1u; }

// Bring two distinct tree nodes under one root to the first chunk in which their paths can differ.
// a/b: positions (in/out), lena/lenb: bits of that chunk (out); below_a/below_b: when a (b) had to climb, the
// node of its chain one chunk below the returned one (its bit 0 is the label right after the returned chunk).
template <bool SC1 = false>
__device__ __forceinline__ void resolve_chunks(const u32* cprev, u32& a, u32& lena, u32& below_a, u32 da, u32& b,
                                               u32& lenb, u32& below_b, u32 db) {
    u32 ia = chunk_of(da), ib = chunk_of(db);
    lena = chunk_len(da); lenb = chunk_len(db);
    below_a = C2A_NONE; below_b = C2A_NONE;
    if ((ia | ib) == 0) return;
    while (ia > ib) { below_a = a; a = ld_u32<SC1>(&cprev[a]); --ia; lena = kChunkBits; }
    while (ib > ia) { below_b = b; b = ld_u32<SC1>(&cprev[b]); --ib; lenb = kChunkBits; }
    while (ia > 0 && a != b) {
        const u32 pa = ld_u32<SC1>(&cprev[a]), pb = ld_u32<SC1>(&cprev[b]);
        if (pa == pb) break;
        below_a = a; below_b = b;
        a = pa; b = pb; --ia;
        lena = lenb = kChunkBits;
    }
}

// lane-sequential comparison (one lane owns the whole comparison): is P(a).la < P(b).lb ?
__device__ __forceinline__ bool str_less_lane(const u64* __restrict__ pstr, const u32* __restrict__ cprev, u32 a, u32 la, u32 da,
                                              u32 b, u32 lb, u32 db) {
    u32 lena, lenb, ba, bb;
    resolve_chunks(cprev, a, lena, ba, da, b, lenb, bb, db);
    if (a == b) {      // one node is the chunk-boundary ancestor of the other: the other's next label decides
        if (ba != C2A_NONE) return (pstr[(u64)ba * kChunkWords] & 1ull) < lb;
        return la < (pstr[(u64)bb * kChunkWords] & 1ull);
    }
    const u64* sa = pstr + (u64)a * kChunkWords;
    const u64* sb = pstr + (u64)b * kChunkWords;
    const u32 minlen = lena < lenb ? lena : lenb;
    for (u32 w = 0; w * 64 < minlen; ++w) {
        const u64 wa = sa[w];
        u64 x = wa ^ sb[w];
        const u32 rem = minlen - w * 64;
        if (rem < 64) x &= (1ull << rem) - 1ull;
        if (x) return ((wa >> ctz64(x)) & 1ull) == 0;
    }
    if (lena == lenb) return la < lb;
    if (lena < lenb) return la < ((sb[lena >> 6] >> (lena & 63u)) & 1ull);
    return ((sa[lenb >> 6] >> (lenb & 63u)) & 1ull) < lb;
}

// wave-uniform pick of one lane's value (v_readlane: a few cycles; ds_bpermute through __shfl costs ~100)
__device__ __forceinline__ u32 rdlane(u32 v, u32 j) {
#ifdef C2A_EMULATE
    return __shfl(v, (int)j, 64);
#else
    return (u32)__builtin_amdgcn_readlane((int)v, (int)j);
#endif
}
__device__ __forceinline__ u64 rdlane64(u64 v, u32 j) { return (u64)rdlane((u32)v, j) | ((u64)rdlane((u32)(v >> 32), j) << 32); }

// wave-cooperative comparison of two strings already in registers (this lane's word of each): wave-uniform result
__device__ __forceinline__ bool str_less_wave(u64 wa, u32 lena, u32 la, u64 wb, u32 lenb, u32 lb, u32 lane) {
    const u32 minlen = lena < lenb ? lena : lenb;
    u64 x = wa ^ wb;
    const u32 lo = lane * 64;
    if (lo >= minlen) x = 0;
    else if (minlen - lo < 64) x &= (1ull << (minlen - lo)) - 1ull;
    const u64 bal = __ballot(x != 0);
    if (bal) {
        const u32 L = ctz64(bal);
        const u64 xl = rdlane64(x, L);
        const u64 al = rdlane64(wa, L);
        return ((al >> ctz64(xl)) & 1ull) == 0;
    }
    if (lena == lenb) return la < lb;
    if (lena < lenb) return la < ((rdlane64(wb, lena >> 6) >> (lena & 63u)) & 1ull);
    return ((rdlane64(wa, lenb >> 6) >> (lenb & 63u)) & 1ull) < lb;
}

// the new node's string: parent's current chunk + one bit, or a fresh chunk when the parent filled its own
__device__ __forceinline__ u64 child_word(u64 parent_word, u32 parent_depth, u32 label, u32 word_index, bool& fresh) {
    const u32 d = parent_depth + 1;
    fresh = parent_depth == 0 || chunk_of(d) != chunk_of(parent_depth);
    const u32 bit = (d - 1) - chunk_of(d) * kChunkBits;
    u64 w = fresh ? 0ull : parent_word;
    if ((bit >> 6) == word_index) w |= (u64)label << (bit & 63u);
    return w;
}

// ---- one lane per gate (wide frontiers: throughput).  256 threads, gridDim.x % kSeg == 0.
// A workgroup's appends go to segment blockIdx.x % kSeg, one ticket per wave and label.
__global__ void __launch_bounds__(kThreads) k_peel_level_str(PeelArgs A, u32 level) {
    __shared__ u32 s_pre[kSeg + 1];
    const u32 lane = threadIdx.x & 63u;
    FrontierSlot* cur = (level & 1u) ? A.slots[1] : A.slots[0];
    FrontierSlot* nxt = (level & 1u) ? A.slots[0] : A.slots[1];
    u32 my_cnt, my_pre, n_front;
    level_segments(A.fring, level, lane, my_cnt, my_pre, n_front);
    const u32 lo = A.fbase[level];
    if (threadIdx.x < kSeg) s_pre[threadIdx.x] = my_pre;
    if (threadIdx.x == 0) s_pre[kSeg] = n_front;
    if (gtid() == 0) A.fbase[level + 1] = lo + n_front;
    if (gtid() < kSeg) A.fring[((level + 2) % kRing) * kSeg + (u32)gtid()] = 0u;
    __syncthreads();
    const u32 seg_out = blockIdx.x % kSeg;
    u32* counter = &A.fring[((level + 1) % kRing) * kSeg + seg_out];
    FrontierSlot* out = nxt + (u64)seg_out * A.seg_cap;
    const u64 lt_mask = (1ull << lane) - 1ull;
    for (u64 base = (u64)blockIdx.x * kThreads; base < n_front; base += gstride()) {
        const u64 i = base + threadIdx.x;
        const bool active = i < n_front;
        u32 rdy[2] = {C2A_NONE, C2A_NONE};
        uint4 rec[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
        if (active) {
            u32 seg = 0;                                        // segment of flat index i: largest s with pre[s] <= i
#pragma unroll
            for (u32 step = kSeg / 2; step; step >>= 1) if (s_pre[seg + step] <= (u32)i) seg += step;
            const FrontierSlot* sl = cur + (u64)seg * A.seg_cap + ((u32)i - s_pre[seg]);
            const uint4 sa = sl->a;
            const u32 cnt = sl->b.x;
            const u32 g = sa.x;
            const u32 pos = lo + (u32)i;
            u32 best = C2A_NONE, best_el = 0, best_root = g, best_depth = 0;
            const u32 e0 = sa.w, e1 = e0 + cnt;
            for (u32 e = e0; e < e1; ++e) {
                const uint4 cr = A.cand[e];
                const u32 pc = cr.x & kIdMask, el = cr.x >> 31;
                bool take;
                if (best == C2A_NONE) take = cr.z < g;
                else if (cr.z != best_root) take = cr.z < best_root;
                else if (pc == best) take = el < best_el;
                else take = str_less_lane(A.pstr, A.cprev, pc, el, cr.y, best, best_el, best_depth);
                if (take) { best = pc; best_el = el; best_root = cr.z; best_depth = cr.y; }
            }
            const u32 depth = best == C2A_NONE ? 0u : best_depth + 1;
            const u32 my_label = best == C2A_NONE ? 0u : best_el;
            A.meta[pos] = make_uint4(best, depth, best_root, my_label | (level << 1));
            A.order[pos] = g;
            A.posof[g] = pos;
            if (best != C2A_NONE) {
                A.child[2 * (u64)best + my_label] = pos;
                u64* dst = A.pstr + (u64)pos * kChunkWords;
                const u64* src = A.pstr + (u64)best * kChunkWords;
                const u32 bit = (depth - 1) - chunk_of(depth) * kChunkBits;
                const bool fresh_chunk = best_depth == 0 || chunk_of(depth) != chunk_of(best_depth);
                // words past a string's end are never written and never read: the parent has `bit` bits, the child bit + 1
                for (u32 w = 0; w * 64 <= bit; ++w) {
                    u64 v = (!fresh_chunk && w * 64 < bit) ? src[w] : 0ull;
                    if (w == (bit >> 6)) v |= (u64)my_label << (bit & 63u);
                    dst[w] = v;
                }
                if (chunk_of(depth)) A.cprev[pos] = fresh_chunk ? best : A.cprev[best];
            }
            const u32 deps[2] = {sa.y, sa.z};
#pragma unroll
            for (u32 l = 0; l < 2; ++l) {
                const u32 d = deps[l];
                if (d == C2A_NONE) continue;
                const uint4 gd = A.ginfo[d];
                const u32 k = atomicAdd(&A.fill[d], 1u);
                A.cand[gd.z + k] = make_uint4(pos | (l << 31), depth, best_root, my_label);
                if (k + 1 == gd.w) { rdy[l] = d; rec[l] = gd; }
            }
        }
        // the wave's appends: one ticket per label (the loop bound is workgroup-uniform, so the wave is converged here)
#pragma unroll
        for (u32 l = 0; l < 2; ++l) {
            const u64 mask = __ballot(rdy[l] != C2A_NONE);
            if (mask) {
                u32 b = 0;
                if (lane == (u32)ctz64(mask)) b = atomicAdd(counter, (u32)__popcll(mask));
                b = __shfl(b, (int)ctz64(mask), 64);
                if (rdy[l] != C2A_NONE) {
                    const u32 p = b + (u32)__popcll(mask & lt_mask);
                    out[p].b = make_uint4(rec[l].w, 0, 0, 0);
                    out[p].a = make_uint4(rdy[l], rec[l].x, rec[l].y, rec[l].z);
                }
            }
        }
    }
}

// ---- one wave per gate (narrow frontiers: latency): survivor strings live in registers, the whole tournament costs
// ONE round trip to memory when every survivor is shallower than a chunk.
// Workgroup = WPB gate waves + ONE append wave; gridDim.x % kSeg == 0; workgroup b serves segment b % kSeg, chunks
// b / kSeg, + gridDim.x / kSeg, ... of WPB gates; the appends of chunk j of segment s go to segment (s + j) % kSeg.
// Dependent memory round trips per level:
//   1. frontier slot (fixed address, loaded before the counts are known) || this level's 16 segment counts
//   2. candidate records  ||  ginfo of the two producers  ||  fill[] tickets of the two pushes
//   3. survivor strings   ||  (append wave) next-level segment ticket -> next-level slots
//   then only stores (meta, order, child, the new string, the two candidate records).
// The pushes' tickets need nothing from the tournament, so "am I the last consumer of this producer" — and with it the
// whole next-level frontier — is known after round trip 2, and the append overlaps the tournament.
constexpr int kStrMax = 4;                // survivor strings held in registers per round (rounds repeat for more)

template <int WPB, bool PROF>
__global__ void __launch_bounds__((WPB + 1) * 64) k_peel_level_wave_str(PeelArgs A, u32 level) {
    const ull t_begin = PROF ? c2a_now() : 0;
    __shared__ u32 s_ready[2 * WPB];
    __shared__ uint4 s_rec[2 * WPB];
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const bool gate_wave = wv < (u32)WPB;
    // both pointers come with the one kernarg fetch; indexing the array by level would be a second, dependent scalar load
    FrontierSlot* cur = (level & 1u) ? A.slots[1] : A.slots[0];
    FrontierSlot* nxt = (level & 1u) ? A.slots[0] : A.slots[1];
    const u32 seg = blockIdx.x % kSeg, j0 = blockIdx.x / kSeg, jstep = gridDim.x / kSeg;
    const FrontierSlot* my_seg = cur + (u64)seg * A.seg_cap;
    const u64 lt_mask = (1ull << lane) - 1ull;
    uint4 sa0 = make_uint4(0, 0, 0, 0);
    u32 cnt0 = 0;
    if (gate_wave) {      // speculative, before the level's counts are known: the segments are padded (kSlotPad)
        const FrontierSlot* sl = my_seg + (j0 * WPB + wv);
        sa0 = sl->a; cnt0 = sl->b.x;
    }
    u32 my_cnt, my_pre, n_front;
    level_segments(A.fring, level, lane, my_cnt, my_pre, n_front);
    const u32 seg_cnt = __shfl(my_cnt, (int)seg, 64), seg_pre = __shfl(my_pre, (int)seg, 64);
    const u32 lo = A.fbase[level];
    if (gtid() == 0) A.fbase[level + 1] = lo + n_front;
    if (gtid() < kSeg) A.fring[((level + 2) % kRing) * kSeg + (u32)gtid()] = 0u;
    u32* next_row = &A.fring[((level + 1) % kRing) * kSeg];
    C2A_PROF(6, t_begin); C2A_PROF(0, c2a_now() - t_begin);
    for (u32 j = j0; j * WPB < seg_cnt; j += jstep) {
        const u32 idx = j * WPB + wv;
        const bool active = gate_wave && idx < seg_cnt;
        uint4 sa = sa0;
        u32 cnt = cnt0;
        if (j != j0 && active) { sa = my_seg[idx].a; cnt = my_seg[idx].b.x; }
        // ---- round trip 2: producers' ginfo, fill tickets, first block of candidate records
        const u32 g = sa.x;
        const u32 e0 = sa.w, e1 = e0 + cnt;
        const u32 dl = active ? (lane == 0 ? sa.y : (lane == 1 ? sa.z : C2A_NONE)) : C2A_NONE;
        // the ticket first (it is the long one), then the two loads BRANCH-FREE (clamped index, result discarded by the
        // lanes that have nothing to load): a predicated load is a branch, and the wait-count pass drains vmcnt at its join
        u32 kfill = 0;
        if (dl != C2A_NONE) kfill = atomicAdd(&A.fill[dl], 1u);
        const bool has_cand = active && e0 + lane < e1;
        const uint4 cr_raw = A.cand[has_cand ? e0 + lane : 0u];
        const uint4 gd_raw = A.ginfo[dl != C2A_NONE ? dl : 0u];
        const uint4 cr_first = has_cand ? cr_raw : make_uint4(0, 0, 0xFFFFFFFFu, 0);
        const uint4 gd = dl != C2A_NONE ? gd_raw : make_uint4(0, 0, 0, 0);
        const bool last_push = dl != C2A_NONE && kfill + 1 == gd.w;
        if (gate_wave && lane < 2) { s_ready[2 * wv + lane] = last_push ? dl : C2A_NONE; s_rec[2 * wv + lane] = gd; }
        if (active) { C2A_PROF(7, 1000ull + cnt); C2A_PROF(1, c2a_now() - t_begin); }
        __syncthreads();
        if (active) C2A_PROF(2, c2a_now() - t_begin);
        if (!gate_wave) {
            // ---- append wave: one ticket on the target segment's counter for the whole workgroup
            const u32 d = lane < 2 * WPB ? s_ready[lane] : C2A_NONE;
            const u64 mask = __ballot(d != C2A_NONE);
            if (mask) {
                const u32 seg_out = (seg + j) % kSeg;
                u32 base = 0;
                if (lane == 0) base = atomicAdd(&next_row[seg_out], (u32)__popcll(mask));
                base = __shfl(base, 0, 64);
                if (d != C2A_NONE) {
                    FrontierSlot* out = nxt + (u64)seg_out * A.seg_cap + (base + (u32)__popcll(mask & lt_mask));
                    const uint4 r = s_rec[lane];
                    st_g(&out->b, make_uint4(r.w, 0, 0, 0));
                    st_g(&out->a, make_uint4(d, r.x, r.y, r.z));
                }
            }
        } else if (active) {
            // champion so far (wave-uniform); NONE = the virtual-root candidate [g].  champ_w = this lane's word of
            // the champion's string when champ_loaded.  Everything wave-uniform lives in scalar registers: candidates
            // are picked out of their lanes with v_readlane (a few cycles), never through LDS or ds_bpermute (~100).
            u32 ch = C2A_NONE, ch_el = 0, ch_root = g, ch_depth = 0;
            u64 champ_w = 0;
            bool champ_loaded = false;
            for (u32 eb = e0; eb < e1; eb += 64) {
                const u32 e = eb + lane;
                const bool valid = e < e1;
                uint4 cr = cr_first;
                if (eb != e0) { cr = make_uint4(0, 0, 0xFFFFFFFFu, 0); if (valid) cr = A.cand[e]; }
                const u32 c = cr.x & kIdMask, l = cr.x >> 31, cdepth = cr.y;
                const u32 croot = valid ? cr.z : 0xFFFFFFFFu;
                const u64 vmask = __ballot(valid);
                u32 rmin;
                if (__popcll(vmask) <= 8) {                     // the usual case: a scalar loop over the few valid lanes
                    rmin = 0xFFFFFFFFu;
                    for (u64 mm = vmask; mm; mm &= mm - 1) { const u32 r = rdlane(croot, ctz64(mm)); rmin = r < rmin ? r : rmin; }
                } else {
                    rmin = wave_min_u32(croot);
                }
                if (rmin > ch_root) continue;
                if (!((ch != C2A_NONE) && (ch_root == rmin))) {   // new smallest root: the first survivor starts as champion
                    champ_loaded = false;
                    ch = C2A_NONE;
                }
                ch_root = rmin;
                u64 smask = __ballot(valid && croot == rmin);
                if (ch == C2A_NONE) {
                    const u32 j = ctz64(smask);
                    smask &= smask - 1;
                    ch = rdlane(c, j); ch_el = rdlane(l, j); ch_depth = rdlane(cdepth, j);
                }
                // sequential tournament, kStrMax survivors per round with their strings in registers
                while (smask) {
                    u32 cc[kStrMax], cl[kStrMax], cd[kStrMax];
                    u32 take = 0;
#pragma unroll
                    for (int t = 0; t < kStrMax; ++t) {
                        if (smask) {
                            const u32 j = ctz64(smask);
                            smask &= smask - 1;
                            cc[t] = rdlane(c, j); cl[t] = rdlane(l, j); cd[t] = rdlane(cdepth, j);
                            take = (u32)t + 1;
                        } else {
                            cc[t] = ch; cl[t] = 0; cd[t] = 0;          // unused slot: length 0, loads word 0 of the champion
                        }
                    }
                    // all of this round in chunk 0?  then one coalesced load per string
                    bool shallow = ch_depth <= kChunkBits;
#pragma unroll
                    for (int t = 0; t < kStrMax; ++t) shallow = shallow && cd[t] <= kChunkBits;
#ifdef C2A_PROF_PRELOAD
                    if (PROF) C2A_PROF(5, c2a_now() - t_begin);
#endif
                    if (shallow) {
                        // the loads back to back, BRANCH-FREE: a load under a lane predicate is a branch, and the
                        // compiler's wait-count pass then drains vmcnt before the next one (measured: +0.65 us per candidate).
                        // Lanes past a string's end (and unused slots, length 0) read word 0 of the string and discard it.
                        u64 sw[kStrMax];
#pragma unroll
                        for (int t = 0; t < kStrMax; ++t) {
                            const bool on = lane * 64 < cd[t];
                            const u64 v = A.pstr[(u64)cc[t] * kChunkWords + (on ? lane : 0u)];
                            sw[t] = on ? v : 0ull;
                        }
                        if (!champ_loaded) {
                            const bool on = lane * 64 < ch_depth;
                            const u64 v = A.pstr[(u64)ch * kChunkWords + (on ? lane : 0u)];
                            champ_w = on ? v : 0ull;
                            champ_loaded = true;
                        }
#ifdef C2A_PROF_STRINGS
                        if (PROF) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); C2A_PROF(5, c2a_now() - t_begin); }
#endif
#pragma unroll
                        for (int t = 0; t < kStrMax; ++t) {
                            if ((u32)t < take) {
                                bool less;
                                if (cc[t] == ch) less = cl[t] < ch_el;
                                else less = str_less_wave(sw[t], cd[t], cl[t], champ_w, ch_depth, ch_el, lane);
                                if (less) { ch = cc[t]; ch_el = cl[t]; ch_depth = cd[t]; champ_w = sw[t]; }
                            }
                        }
                    } else {
                        // deep trees: chunk resolution (cprev hops) per comparison, strings loaded per comparison
                        for (u32 t = 0; t < take; ++t) {
                            const u32 cct = cc[t], clt = cl[t], cdt = cd[t];
                            bool less;
                            if (cct == ch) less = clt < ch_el;
                            else {
                                u32 ra = cct, rb = ch, lena, lenb, ba, bb;
                                resolve_chunks(A.cprev, ra, lena, ba, cdt, rb, lenb, bb, ch_depth);
                                if (ra == rb) {
                                    if (ba != C2A_NONE) less = (A.pstr[(u64)ba * kChunkWords] & 1ull) < ch_el;
                                    else less = clt < (A.pstr[(u64)bb * kChunkWords] & 1ull);
                                } else {
                                    const u64 wa = lane * 64 < lena ? A.pstr[(u64)ra * kChunkWords + lane] : 0ull;
                                    const u64 wb = lane * 64 < lenb ? A.pstr[(u64)rb * kChunkWords + lane] : 0ull;
                                    less = str_less_wave(wa, lena, clt, wb, lenb, ch_el, lane);
                                }
                            }
                            if (less) { ch = cct; ch_el = clt; ch_depth = cdt; champ_loaded = false; }
                        }
                    }
                }
            }
            C2A_PROF(3, c2a_now() - t_begin);
            const u32 depth = ch == C2A_NONE ? 0u : ch_depth + 1;
            const u32 my_label = ch == C2A_NONE ? 0u : ch_el;
            const u32 pos = lo + seg_pre + idx;
            if (lane == 0) {
                st_g(&A.meta[pos], make_uint4(ch, depth, ch_root, my_label | (level << 1)));
                st_g(&A.order[pos], g);
                st_g(&A.posof[g], pos);
                if (ch != C2A_NONE) st_g(&A.child[2 * (u64)ch + my_label], pos);
            }
            if (dl != C2A_NONE) st_g(&A.cand[gd.z + kfill], make_uint4(pos | (lane << 31), depth, ch_root, my_label));   // lane == edge label
            if (ch != C2A_NONE) {
                const bool need_parent = ch_depth != 0 && chunk_of(depth) == chunk_of(ch_depth);
                if (need_parent && !champ_loaded) champ_w = lane * 64 < chunk_len(ch_depth) ? A.pstr[(u64)ch * kChunkWords + lane] : 0ull;
                bool fresh;
                const u64 nw = child_word(need_parent ? champ_w : 0ull, ch_depth, my_label, lane, fresh);
                if (lane * 64 < chunk_len(depth)) st_g(&A.pstr[(u64)pos * kChunkWords + lane], nw);      // words past the end are never read
                if (lane == 0 && chunk_of(depth)) st_g(&A.cprev[pos], fresh ? ch : A.cprev[ch]);
            }
            C2A_PROF(4, c2a_now() - t_begin);
        }
        __syncthreads();
#if !defined(C2A_PROF_STRINGS) && !defined(C2A_PROF_PRELOAD)
        if (active) C2A_PROF(5, c2a_now() - t_begin);
#endif
    }
}

// ================================================================================================
// ASYNCHRONOUS PEEL — the same tournament per gate, but no level barrier at all.
// A launch per level costs ~7 us per level at ~2 000 gates per level whatever is done inside (dispatch + three dependent
// round trips + the slowest of ~2 000 waves), and the graph is ~5 000 levels deep.  Here ONE launch runs the whole peel
// as a dataflow: the wave whose push completes a producer's consumer list goes on with that producer at once, so the
// critical path is a chain of (candidate records -> survivor strings -> stores acknowledged -> commit ticket) steps,
// ~2 us each, with no launch, no frontier and no counters in between.  No wave ever waits for another one (nothing
// spins, nothing can deadlock; the host emulation runs the workgroups one after the other): a gate is processed by
// exactly the wave that completed it, work found beyond the one gate a wave can continue with goes on a wave-private
// intrusive stack (link[]), and a wave that runs out of work exits.
// Data exchanged between waves inside the launch: candidate records, path strings, cprev — written with agent-scope
// write-through stores, drained (vmcnt 0) BEFORE the commit ticket, read with agent-scope loads AFTER the ticket that
// completes the list (verified hand-off: tools/ubench/xcd2.hip, sc1 store -> atomic -> sc1 load, 0 stale reads).
// A candidate record's home is static (cand[cons_off[d] + eslot[2g+l]], the edge's index from k_deps), so a push
// needs ONE ticket and that ticket is the commit.  Tree node identity = gate id (order/posof are the identity).
// The sinks (gates nobody consumes: 14 % of the headline graph) are peeled by a plain grid-stride kernel first; the
// producers they complete seed the dataflow launch.
// ================================================================================================
struct AsyncArgs {
    u32 n;
    u32 seg_cap;
    const uint4* ginfo;        // [n] {dep0, dep1, cons_off, cons_cnt}
    const u32* eslot;          // [2n] index of edge (g, l) in its producer's candidate list
    FrontierSlot* seeds;       // [n_regions][region_cap] gates completed by the sinks, one region per workgroup of k_async_sinks
    u32* seed_cnt;             // [n_regions]
    u32 n_regions, region_cap;
    uint4* cand;
    u32* fill;
    uint4* meta;               // by gate id
    u64* pstr;
    u32* cprev;
    u32* child;
    u32* link;                 // [n] wave-private stacks (used when a hand-off queue is full — never in practice)
    u32* totals;               // [0] gates processed by the dataflow launch + sinks, [1] max level, [2] watchdog trips
    // hand-off of completed producers a wave cannot continue with: n_queues ticket queues of q_cap entries each
    u32 n_queues, q_cap;
    u64* q_ht;                 // [n_queues * kQStride] head (low word) | tail (high word), one queue per 128-byte line
    u32* q_items;              // [n_queues][q_cap] gate + 1, 0 = not written yet (zeroed per run, every entry used once)
    u32* idle;                 // [kIdleCounters * 16] waves with nothing to do (one counter per 64-byte line)
    ull* stats;                // optional diagnostics (16 words), nullptr normally
    ull* q_time;               // with stats: push time of every queue entry
};
constexpr u32 kIdleCounters = 64;
constexpr u32 kQStride = 16;               // u64 words between two queues' head/tail words
constexpr u32 kWatchdogPolls = 1u << 22;   // ~2 s of polling: give up instead of hanging the GPU (reported as an error)

template <bool SC1> __device__ __forceinline__ uint4 ld_rec(const uint4* p) {
    if (SC1) {
        const u64* q = reinterpret_cast<const u64*>(p);
        const u64 a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_uint4((u32)a, (u32)(a >> 32), (u32)b, (u32)(b >> 32));
    }
    return *p;
}
__device__ __forceinline__ void st_rec_sc1(uint4* p, const uint4& v) {
    u64* q = reinterpret_cast<u64*>(p);
    __hip_atomic_store(q, (u64)v.x | ((u64)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, (u64)v.z | ((u64)v.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 ld_str(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_str(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// sinks: roots of depth 0 without candidates; their pushes are published by the kernel boundary.  256 threads,
// gridDim.x % kSeg == 0; the producers completed here are appended to seed segment blockIdx.x % kSeg.
// sinks: roots of depth 0 without candidates; their pushes are published by the kernel boundary.  256 threads.
// The producers completed here seed the dataflow launch.  No shared counter anywhere: workgroup b appends to its own
// region seeds[b * region_cap ...] under its own counter (a block handles at most region_cap / 2 gates, a gate completes
// at most two producers); the dataflow waves share the regions out statically.
__global__ void __launch_bounds__(kThreads) k_async_sinks(AsyncArgs A) {
    __shared__ u32 s_done[kThreads / 64];
    const u32 lane = threadIdx.x & 63u;
    FrontierSlot* out = A.seeds + (u64)blockIdx.x * A.region_cap;
    u32* counter = &A.seed_cnt[blockIdx.x];
    const u64 lt_mask = (1ull << lane) - 1ull;
    u32 done = 0;
    for (u64 base = (u64)blockIdx.x * kThreads; base < A.n; base += gstride()) {
        const u64 g = base + threadIdx.x;
        u32 rdy[2] = {C2A_NONE, C2A_NONE};
        uint4 rec[2] = {make_uint4(0, 0, 0, 0), make_uint4(0, 0, 0, 0)};
        if (g < A.n) {
            const uint4 gi = A.ginfo[g];
            if (gi.w == 0) {
                ++done;
                A.meta[g] = make_uint4(C2A_NONE, 0u, (u32)g, 0u);
                const u32 deps[2] = {gi.x, gi.y};
#pragma unroll
                for (u32 l = 0; l < 2; ++l) {
                    const u32 d = deps[l];
                    if (d == C2A_NONE) continue;
                    const uint4 gd = A.ginfo[d];
                    A.cand[gd.z + A.eslot[2 * g + l]] = make_uint4((u32)g | (l << 31), 0u, (u32)g, 0u);
                    const u32 k = atomicAdd(&A.fill[d], 1u);
                    if (k + 1 == gd.w) { rdy[l] = d; rec[l] = gd; }
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
                if (rdy[l] != C2A_NONE) {
                    const u32 p = b + (u32)__popcll(mask & lt_mask);
                    out[p].b = make_uint4(rec[l].w, 0, 0, 0);
                    out[p].a = make_uint4(rdy[l], rec[l].x, rec[l].y, rec[l].z);
                }
            }
        }
    }
    // one update of the global count per workgroup
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) done += __shfl_xor(done, off, 64);
    if (lane == 0) s_done[threadIdx.x >> 6] = done;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 t = 0;
        for (int w = 0; w < kThreads / 64; ++w) t += s_done[w];
        if (t) atomicAdd(&A.totals[0], t);
    }
}

// advance (region, idx) to this wave's next seed; false when its share is exhausted
__device__ __forceinline__ bool next_seed(const AsyncArgs& A, u32& region, u32& idx, u32& region_cnt) {
    while (region < A.n_regions) {
        if (idx < region_cnt) return true;
        region += gridDim.x;
        idx = 0;
        region_cnt = region < A.n_regions ? A.seed_cnt[region] : 0u;
    }
    return false;
}

// the dataflow launch: 64-thread workgroups (one wave each)
template <bool STATS>
__global__ void __launch_bounds__(64) k_peel_async(AsyncArgs A) {
    const u32 lane = threadIdx.x;
    u32 region = blockIdx.x, idx = 0;               // this wave's share of the seeds: regions me, me + waves, ...
    u32 region_cnt = region < A.n_regions ? A.seed_cnt[region] : 0u;
    u32 head = C2A_NONE;                     // wave-private stack (overflow of the hand-off queues)
    u32 processed = 0, max_level = 0;
    const u32 W = gridDim.x, me = blockIdx.x;
    const u32 home_q = me % A.n_queues;
    u32* head_w = reinterpret_cast<u32*>(A.q_ht);            // head of queue q = word 2 q kQStride, tail = the next word
    u32 push_rr = me;                        // round-robin cursor of this wave's pushes
    u32 roam = me * 0x9E3779B1u;             // pseudo-random walk over the other queues
    bool registered = false;                 // counted in idle[]
    u32 st_pops = 0, st_polls = 0, st_push = 0, st_seeds = 0;      // diagnostics (C2A_ASYNC_STATS)
    ull st_busy = 0, st_idle = 0, st_t0 = STATS ? c2a_now() : 0, st_hand = 0, st_final = 0;
    for (;;) {
        // ---- next piece of work: own stack, own share of the seeds, then the hand-off queues
        u32 g = C2A_NONE;
        uint4 gi = make_uint4(0, 0, 0, 0);
        if (head != C2A_NONE) {
            g = head;
            head = ld_u32<true>(&A.link[g]);
            gi = A.ginfo[g];
        } else if (next_seed(A, region, idx, region_cnt)) {
            const FrontierSlot* sl = A.seeds + (u64)region * A.region_cap + idx;
            ++idx;
            ++st_seeds;
            const uint4 a = sl->a;
            g = a.x;
            gi = make_uint4(a.y, a.z, a.w, sl->b.x);
        } else {
            if (STATS) { const ull t = c2a_now(); st_busy += t - st_t0; st_t0 = t; }
            // poll: home queue, then a roaming one; a wave counts as idle from its first empty-handed poll until the
            // moment BEFORE it tries to claim an entry, so "all waves idle" implies that nothing is queued or in flight
            u32 polls = 0;
            u32 hint = C2A_NONE;                                 // a queue seen non-empty by the termination check
            for (;;) {
#ifdef C2A_EMULATE
                const u32 q = (home_q + polls) % A.n_queues;     // workgroups run one after the other: scan every queue once
#else
                u32 q = (polls & 1u) ? (roam = roam * 1664525u + 1013904223u, (roam >> 8) % A.n_queues) : home_q;
                if (hint != C2A_NONE) { q = hint; hint = C2A_NONE; }
#endif
                // one lane looks, everybody acts on what it saw (wave-uniform by construction, also for the host emulation
                // where the lanes of a wave run one after the other)
                u64 ht = 0;
                if (lane == 0) ht = __hip_atomic_load(&A.q_ht[(u64)q * kQStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ht = rdlane64(ht, 0);
                const u32 qh = (u32)ht, qt = (u32)(ht >> 32);
                if (qh < qt) {
                    if (registered) { if (lane == 0) atomicAdd(&A.idle[(me % kIdleCounters) * 16], 0xFFFFFFFFu); registered = false; }
                    u32 old = 0;
                    if (lane == 0) old = atomicCAS(&head_w[2 * (u64)q * kQStride], qh, qh + 1);
                    old = rdlane(old, 0);
                    if (old == qh) {
                        u32 v = 0, spins = 0;
                        do {
                            if (lane == 0) v = ld_u32<true>(&A.q_items[(u64)q * A.q_cap + qh]);
                            v = rdlane(v, 0);
                        } while (v == 0 && ++spins < kWatchdogPolls);
                        if (v) { g = v - 1; gi = A.ginfo[g]; ++st_pops; if (STATS) { const ull tp = __hip_atomic_load(&A.q_time[(u64)q * A.q_cap + qh], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); if (tp) { const ull dt = c2a_now() - tp; st_hand += dt; const u32 b = dt < 100 ? 0 : dt < 200 ? 1 : dt < 400 ? 2 : dt < 800 ? 3 : dt < 1600 ? 4 : 5; if (lane == 0) atomicAdd(&A.stats[10 + b], 1ull); } } }
                        else if (lane == 0) atomicAdd(&A.totals[2], 1u);
                        break;
                    }
                    hint = q;                                    // lost the race: look at the same queue again at once
                    continue;
                }
#ifdef C2A_EMULATE
                if (++polls >= A.n_queues) break;                // nothing left for this workgroup
#else
                if (!registered) { if (lane == 0) atomicAdd(&A.idle[(me % kIdleCounters) * 16], 1u); registered = true; }
                ++polls;
                if ((polls & 31u) == 0) {
                    u32 cnt = lane < kIdleCounters ? ld_u32<true>(&A.idle[lane * 16]) : 0u;
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
                    if (cnt >= W) {
                        // every wave is idle or gone, so nobody pushes any more: finished unless an entry is still queued
                        u32 found = C2A_NONE;
                        for (u32 qq = lane; qq < A.n_queues; qq += 64) {
                            const u64 x = __hip_atomic_load(&A.q_ht[(u64)qq * kQStride], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if ((u32)x < (u32)(x >> 32)) found = qq;
                        }
                        const u64 fm = __ballot(found != C2A_NONE);
                        if (fm == 0) break;
                        hint = rdlane(found, ctz64(fm));
                        continue;
                    }
                }
                if (polls >= kWatchdogPolls) { if (lane == 0) atomicAdd(&A.totals[2], 1u); break; }
                // back off: the longer nothing turns up, the less often this wave asks (64 clocks per unit, <= ~3 us)
                if (polls < 8) __builtin_amdgcn_s_sleep(4); else if (polls < 64) __builtin_amdgcn_s_sleep(16); else __builtin_amdgcn_s_sleep(64);
#endif
            }
            st_polls += polls;
            if (STATS) { const ull t = c2a_now(); st_idle += t - st_t0; if (g == C2A_NONE) st_final = t - st_t0; st_t0 = t; }
            if (g == C2A_NONE) break;
        }
        // ---- follow the chain from g.  When the chain goes on from gate p to its producer g, p's own record and string
        // are still in registers: if p is g's ONLY consumer the whole tournament needs no load at all.
        bool own_valid = false;
        u32 own_node = 0, own_label = 0, own_depth = 0, own_root = 0, own_level = 0;
        u64 own_str = 0;
        for (;;) {
            const u32 e0 = gi.z, e1 = e0 + gi.w;
            const u32 dl = lane == 0 ? gi.x : (lane == 1 ? gi.y : C2A_NONE);
            // static data of the pushes, fetched while the tournament runs
            const uint4 gd_raw = A.ginfo[dl != C2A_NONE ? dl : 0u];
            const u32 es = A.eslot[2 * (u64)g + (lane & 1u)];
            u32 ch = C2A_NONE, ch_el = 0, ch_root = g, ch_depth = 0, level = 0;
            u64 champ_w = 0;
            bool champ_loaded = false;
            const bool only_me = own_valid && gi.w == 1;
            if (only_me) {
                // the single candidate is the gate this wave just finished: same rules as below (a larger DFS root loses to
                // [g] itself), no round trip
                level = own_level + 1;
                if (own_root < g) { ch = own_node; ch_el = own_label; ch_depth = own_depth; ch_root = own_root; champ_w = own_str; champ_loaded = true; }
            }
            for (u32 eb = e0; eb < (only_me ? e0 : e1); eb += 64) {
                const u32 e = eb + lane;
                const bool valid = e < e1;
                const uint4 cr_raw = ld_rec<true>(&A.cand[valid ? e : e0]);
                const u32 c = cr_raw.x & kIdMask, l = cr_raw.x >> 31, cdepth = cr_raw.y;
                const u32 croot = valid ? cr_raw.z : 0xFFFFFFFFu;
                const u32 clevel = valid ? (cr_raw.w >> 1) + 1u : 0u;
                const u64 vmask = __ballot(valid);
                u32 rmin = 0xFFFFFFFFu;
                if (__popcll(vmask) <= 8) {
                    for (u64 mm = vmask; mm; mm &= mm - 1) {
                        const u32 j = ctz64(mm);
                        const u32 r = rdlane(croot, j), lv = rdlane(clevel, j);
                        rmin = r < rmin ? r : rmin;
                        level = lv > level ? lv : level;
                    }
                } else {
                    rmin = wave_min_u32(croot);
                    u32 lv = clevel;
#pragma unroll
                    for (int off = 32; off >= 1; off >>= 1) { const u32 o = __shfl_xor(lv, off, 64); lv = o > lv ? o : lv; }
                    level = lv > level ? lv : level;
                }
                if (rmin > ch_root) continue;
                if (!((ch != C2A_NONE) && (ch_root == rmin))) { champ_loaded = false; ch = C2A_NONE; }
                ch_root = rmin;
                u64 smask = __ballot(valid && croot == rmin);
                if (ch == C2A_NONE) {
                    const u32 j = ctz64(smask);
                    smask &= smask - 1;
                    ch = rdlane(c, j); ch_el = rdlane(l, j); ch_depth = rdlane(cdepth, j);
                }
                while (smask) {
                    u32 cc[kStrMax], cl[kStrMax], cd[kStrMax];
                    u32 take = 0;
#pragma unroll
                    for (int t = 0; t < kStrMax; ++t) {
                        if (smask) {
                            const u32 j = ctz64(smask);
                            smask &= smask - 1;
                            cc[t] = rdlane(c, j); cl[t] = rdlane(l, j); cd[t] = rdlane(cdepth, j);
                            take = (u32)t + 1;
                        } else {
                            cc[t] = ch; cl[t] = 0; cd[t] = 0;
                        }
                    }
                    bool shallow = ch_depth <= kChunkBits;
#pragma unroll
                    for (int t = 0; t < kStrMax; ++t) shallow = shallow && cd[t] <= kChunkBits;
                    if (shallow && take == 1) {
                        // the usual case, two survivors: exactly two loads (every request counts — the launch is bound by
                        // the memory side's request rate, not by bytes)
                        const bool on0 = lane * 64 < cd[0];
                        const u64 v0 = ld_str(&A.pstr[(u64)cc[0] * kChunkWords + (on0 ? lane : 0u)]);
                        if (!champ_loaded) {
                            const bool on = lane * 64 < ch_depth;
                            const u64 v = ld_str(&A.pstr[(u64)ch * kChunkWords + (on ? lane : 0u)]);
                            champ_w = on ? v : 0ull;
                            champ_loaded = true;
                        }
                        const u64 s0 = on0 ? v0 : 0ull;
                        bool less;
                        if (cc[0] == ch) less = cl[0] < ch_el;
                        else less = str_less_wave(s0, cd[0], cl[0], champ_w, ch_depth, ch_el, lane);
                        if (less) { ch = cc[0]; ch_el = cl[0]; ch_depth = cd[0]; champ_w = s0; }
                    } else if (shallow) {
                        u64 sw[kStrMax];
#pragma unroll
                        for (int t = 0; t < kStrMax; ++t) {
                            const bool on = lane * 64 < cd[t];
                            const u64 v = ld_str(&A.pstr[(u64)cc[t] * kChunkWords + (on ? lane : 0u)]);
                            sw[t] = on ? v : 0ull;
                        }
                        if (!champ_loaded) {
                            const bool on = lane * 64 < ch_depth;
                            const u64 v = ld_str(&A.pstr[(u64)ch * kChunkWords + (on ? lane : 0u)]);
                            champ_w = on ? v : 0ull;
                            champ_loaded = true;
                        }
#pragma unroll
                        for (int t = 0; t < kStrMax; ++t) {
                            if ((u32)t < take) {
                                bool less;
                                if (cc[t] == ch) less = cl[t] < ch_el;
                                else less = str_less_wave(sw[t], cd[t], cl[t], champ_w, ch_depth, ch_el, lane);
                                if (less) { ch = cc[t]; ch_el = cl[t]; ch_depth = cd[t]; champ_w = sw[t]; }
                            }
                        }
                    } else {
                        for (u32 t = 0; t < take; ++t) {
                            const u32 cct = cc[t], clt = cl[t], cdt = cd[t];
                            bool less;
                            if (cct == ch) less = clt < ch_el;
                            else {
                                u32 ra = cct, rb = ch, lena, lenb, ba, bb;
                                resolve_chunks<true>(A.cprev, ra, lena, ba, cdt, rb, lenb, bb, ch_depth);
                                if (ra == rb) {
                                    if (ba != C2A_NONE) less = (u32)(ld_str(&A.pstr[(u64)ba * kChunkWords]) & 1ull) < ch_el;
                                    else less = clt < (u32)(ld_str(&A.pstr[(u64)bb * kChunkWords]) & 1ull);
                                } else {
                                    const u64 wa = lane * 64 < lena ? ld_str(&A.pstr[(u64)ra * kChunkWords + lane]) : 0ull;
                                    const u64 wb = lane * 64 < lenb ? ld_str(&A.pstr[(u64)rb * kChunkWords + lane]) : 0ull;
                                    less = str_less_wave(wa, lena, clt, wb, lenb, ch_el, lane);
                                }
                            }
                            if (less) { ch = cct; ch_el = clt; ch_depth = cdt; champ_loaded = false; }
                        }
                    }
                }
            }
            // ---- the node: meta / child are read after the launch only (plain), the string and cprev by other waves
            const u32 depth = ch == C2A_NONE ? 0u : ch_depth + 1;
            const u32 my_label = ch == C2A_NONE ? 0u : ch_el;
            const u32 tag = my_label | (level << 1);
            max_level = level > max_level ? level : max_level;
            ++processed;
            if (lane == 0) {
                A.meta[g] = make_uint4(ch, depth, ch_root, tag);
                if (ch != C2A_NONE) A.child[2 * (u64)ch + my_label] = g;
            }
            u64 own_word = 0;
            if (ch != C2A_NONE) {
                const bool need_parent = ch_depth != 0 && chunk_of(depth) == chunk_of(ch_depth);
                if (need_parent && !champ_loaded) {
                    const bool on = lane * 64 < chunk_len(ch_depth);
                    const u64 v = ld_str(&A.pstr[(u64)ch * kChunkWords + (on ? lane : 0u)]);
                    champ_w = on ? v : 0ull;
                }
                bool fresh;
                const u64 nw = child_word(need_parent ? champ_w : 0ull, ch_depth, my_label, lane, fresh);
                if (lane * 64 < chunk_len(depth)) { st_str(&A.pstr[(u64)g * kChunkWords + lane], nw); own_word = nw; }
                if (lane == 0 && chunk_of(depth)) st_u32<true>(&A.cprev[g], fresh ? ch : ld_u32<true>(&A.cprev[ch]));
            }
            // ---- pushes: record to its static home, everything acknowledged, then the commit ticket
            const uint4 gd = dl != C2A_NONE ? gd_raw : make_uint4(0, 0, 0, 0);
            if (dl != C2A_NONE) st_rec_sc1(&A.cand[gd.z + es], make_uint4(g | (lane << 31), depth, ch_root, tag));     // lane == edge label
#ifndef C2A_EMULATE
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            u32 kfill = 0;
            if (dl != C2A_NONE) kfill = atomicAdd(&A.fill[dl], 1u);
            const bool last_push = dl != C2A_NONE && kfill + 1 == gd.w;
            const u64 rmask = __ballot(last_push);
            if (rmask == 0) break;                                  // the chain ends here
            // continue with the first completed producer; a second one goes on the stack
            const u32 j0 = ctz64(rmask);
            const u32 nxt = rdlane(dl, j0);
            const uint4 ngi = make_uint4(rdlane(gd.x, j0), rdlane(gd.y, j0), rdlane(gd.z, j0), rdlane(gd.w, j0));
            if (rmask & (rmask - 1)) {
                // hand the second one to whoever is idle: ticket on a queue (round robin over all queues), then the entry
                const u32 other = rdlane(dl, 1);
                const u32 q = (push_rr++) % A.n_queues;
                ++st_push;
                u32 t = 0;
                if (lane == 0) t = atomicAdd(&head_w[2 * (u64)q * kQStride + 1], 1u);
                t = rdlane(t, 0);
                if (t < A.q_cap) {
                    if (STATS && lane == 0) __hip_atomic_store(&A.q_time[(u64)q * A.q_cap + t], c2a_now(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (lane == 0) st_u32<true>(&A.q_items[(u64)q * A.q_cap + t], other + 1u);
                } else {                                           // cannot happen with q_cap >= n / n_queues + waves (see host)
                    if (lane == 0) st_u32<true>(&A.link[other], head);
                    head = other;
                }
            }
            // what the next step may reuse: this gate as a candidate of nxt
            own_valid = true; own_node = g; own_label = j0; own_depth = depth; own_root = ch_root; own_level = level;
            own_str = own_word;
            g = nxt; gi = ngi;
        }
    }
    if (lane == 0) {
        if (!registered) atomicAdd(&A.idle[(me % kIdleCounters) * 16], 1u);      // a wave that has left counts as idle for good
        if (processed) atomicAdd(&A.totals[0], processed);
        if (max_level) atomicMax(&A.totals[1], max_level);
        if (STATS) {
            atomicAdd(&A.stats[0], (ull)st_pops); atomicAdd(&A.stats[1], (ull)st_polls); atomicAdd(&A.stats[2], (ull)st_push);
            atomicAdd(&A.stats[3], st_busy); atomicAdd(&A.stats[4], st_idle); atomicAdd(&A.stats[5], (ull)st_seeds); atomicAdd(&A.stats[6], (ull)processed);
            atomicMax(&A.stats[7], st_busy);
            atomicAdd(&A.stats[8], st_hand); atomicAdd(&A.stats[9], st_final);
        }
    }
}

// tree node == gate id in the asynchronous peel
__global__ void k_identity(u32 n, u32* a, u32* b) {
    for (u64 i = gtid(); i < n; i += gstride()) { a[i] = (u32)i; b[i] = (u32)i; }
}

// ------------------------------------------------------------------------------------------------
// post-order numbering: Euler tour of the DFS tree + list ranking (random splitters)
// ------------------------------------------------------------------------------------------------
// also collects the DFS-tree depth (stat) with one atomic per workgroup — never one per gate on a single word
__global__ void __launch_bounds__(kThreads) k_rootflag(u32 n, const uint4* __restrict__ meta,
                                                       const u32* __restrict__ posof, u32* rflag, u32* maxdepth) {
    __shared__ u32 s_max[kThreads];
    u32 md = 0;
    for (u64 g = gtid(); g < n; g += gstride()) {
        const uint4 m = meta[posof[g]];
        rflag[g] = m.x == C2A_NONE ? 1u : 0u;
        md = m.y > md ? m.y : md;
    }
    s_max[threadIdx.x] = md;
    __syncthreads();
    for (u32 off = kThreads / 2; off; off >>= 1) {
        if (threadIdx.x < off) { const u32 o = s_max[threadIdx.x + off]; if (o > s_max[threadIdx.x]) s_max[threadIdx.x] = o; }
        __syncthreads();
    }
    if (threadIdx.x == 0 && s_max[0]) atomicMax(maxdepth, s_max[0]);
}

// DFS roots in ascending gate id (topological_sort.rs:11-13), as tree positions
__global__ void k_rootlist(u32 n, const u32* __restrict__ rflag, const u32* __restrict__ ridx,
                           const u32* __restrict__ posof, u32* rlist) {
    for (u64 g = gtid(); g < n; g += gstride())
        if (rflag[g]) rlist[ridx[g]] = posof[g];
}

// element 2x = enter(x), 2x+1 = exit(x); the tour visits label-0 child, label-1 child, then exits.
// child[2p + l] was written by the peel when the child picked (p, l) as its parent (NONE otherwise).
__global__ void k_euler_next(u32 n, const uint4* __restrict__ meta, const u32* __restrict__ order,
                             const u32* __restrict__ child, const u32* __restrict__ ridx, const u32* __restrict__ rlist,
                             u32 n_roots, u32* next) {
    for (u64 i = gtid(); i < n; i += gstride()) {
        const u32 x = (u32)i;
        const u32 c0 = child[2 * i], c1 = child[2 * i + 1];
        next[2 * i] = c0 != C2A_NONE ? 2 * c0 : (c1 != C2A_NONE ? 2 * c1 : 2 * x + 1);
        const uint4 m = meta[x];
        u32 nx;
        if (m.x == C2A_NONE) {
            const u32 k = ridx[order[x]];
            nx = k + 1 < n_roots ? 2 * rlist[k + 1] : C2A_NONE;
        } else {
            const u32 s1 = (m.w & 1u) == 0 ? child[2 * (u64)m.x + 1] : C2A_NONE;
            nx = s1 != C2A_NONE ? 2 * s1 : 2 * m.x + 1;
        }
        next[2 * i + 1] = nx;
    }
}

#ifndef C2A_SPLIT_SHIFT
#define C2A_SPLIT_SHIFT 26      // one splitter per 2^(32 - shift) tour elements
#endif
__device__ __forceinline__ bool is_splitter(u32 e, u32 head) { return e == head || ((e * 0x9E3779B1u) >> C2A_SPLIT_SHIFT) == 0u; }

// splitter compaction, 8 elements per lane: one atomic and three barriers per 2048 elements
__global__ void __launch_bounds__(kThreads) k_rank_mark(u32 m, const u32* __restrict__ rlist, u32* scount, u32* slist,
                                                        u32* owner) {
    __shared__ u32 s_w[kThreads / 64];
    __shared__ u32 s_base;
    const u32 head = 2 * rlist[0];
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    for (u64 base = (u64)blockIdx.x * (kThreads * 8); base < m; base += (u64)gridDim.x * (kThreads * 8)) {
        const u64 e0 = base + (u64)tid * 8;
        u32 bits = 0, cnt = 0;
#pragma unroll
        for (u32 k = 0; k < 8; ++k) {
            const u64 e = e0 + k;
            const u32 sp = (e < m && is_splitter((u32)e, head)) ? 1u : 0u;
            bits |= sp << k;
            cnt += sp;
        }
        u32 inc = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const u32 o = __shfl_up(inc, off, 64);
            if (lane >= (u32)off) inc += o;
        }
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        if (tid == 0) {
            u32 run = 0;
            for (int w = 0; w < kThreads / 64; ++w) { const u32 t = s_w[w]; s_w[w] = run; run += t; }
            s_base = run ? atomicAdd(scount, run) : 0u;
        }
        __syncthreads();
        u32 pos = s_base + s_w[wv] + inc - cnt;
#pragma unroll
        for (u32 k = 0; k < 8; ++k)
            if (bits & (1u << k)) { slist[pos] = (u32)(e0 + k); owner[e0 + k] = pos; ++pos; }
        __syncthreads();
    }
}

// one thread per splitter walks its sublist: local[e] = number of exits before e inside the sublist
__global__ void k_rank_walk(const u32* __restrict__ scount, const u32* __restrict__ rlist, const u32* __restrict__ slist,
                            const u32* __restrict__ next, u32* owner, u32* local, u32* snext, u32* ssum) {
    const u32 S = *scount;
    const u32 head = 2 * rlist[0];
    for (u64 k = gtid(); k < S; k += gstride()) {
        u32 e = slist[k], acc = 0;
        for (;;) {
            local[e] = acc;
            owner[e] = (u32)k;
            acc += e & 1u;
            const u32 e2 = next[e];
            if (e2 == C2A_NONE) { snext[k] = C2A_NONE; break; }
            if (is_splitter(e2, head)) { snext[k] = owner[e2]; break; }
            e = e2;
        }
        ssum[k] = acc;
    }
}

// pointer jumping over the splitter list: after ceil(log2 S) rounds val[k] = sum over k..end
__global__ void k_rank_jump(const u32* __restrict__ scount, const u32* __restrict__ nxt_in, const u32* __restrict__ val_in,
                            u32* nxt_out, u32* val_out) {
    const u32 S = *scount;
    for (u64 k = gtid(); k < S; k += gstride()) {
        const u32 nx = nxt_in[k];
        u32 v = val_in[k];
        u32 nn = C2A_NONE;
        if (nx != C2A_NONE) { v += val_in[nx]; nn = nxt_in[nx]; }
        val_out[k] = v;
        nxt_out[k] = nn;
    }
}

// sorted[post-order index of x] = gate(x)   (== sorted.push(i), topological_sort.rs:46)
__global__ void k_rank_final(u32 n, const u32* __restrict__ order, const u32* __restrict__ owner,
                             const u32* __restrict__ local, const u32* __restrict__ suffix, u32* sorted) {
    for (u64 x = gtid(); x < n; x += gstride()) {
        const u32 e = 2 * (u32)x + 1;
        const u32 post = (n - suffix[owner[e]]) + local[e];
        sorted[post] = order[x];
    }
}

// ------------------------------------------------------------------------------------------------
// serial DFS on one lane: literal topological_sort.rs with an explicit stack.  Only used to produce the
// reference's cycle diagnostic ("detected at i={}", topological_sort.rs:34-38) once the parallel peel
// has found leftover gates, and as an in-library cross-check (c2a_topo_sort_serial).
// state[g]: bit0 visiting, bit1 visited, bits 2..3 next dep index.  result = {status, cycle_at, count}
// ------------------------------------------------------------------------------------------------
__global__ void k_serial_dfs(u32 n, const u32* __restrict__ dep0, const u32* __restrict__ dep1, u8* state, u32* stack,
                             u32* sorted, u32* result) {
    if (gtid() != 0) return;
    u32 ns = 0;
    for (u32 root = 0; root < n; ++root) {
        if (state[root] & 2u) continue;
        u32 sp = 0;
        stack[sp++] = root;
        state[root] = 1u;
        while (sp) {
            const u32 i = stack[sp - 1];
            const u32 st = state[i];
            const u32 k = st >> 2;
            if (k < 2) {
                state[i] = (u8)((st & 3u) | ((k + 1) << 2));
                const u32 j = k == 0 ? dep0[i] : dep1[i];
                if (j == C2A_NONE) continue;
                const u32 sj = state[j];
                if (sj & 2u) continue;
                if (sj & 1u) { result[0] = 1; result[1] = j; result[2] = ns; return; }
                state[j] = 1u;
                stack[sp++] = j;
            } else {
                sorted[ns++] = i;
                state[i] = (u8)(st | 2u);
                --sp;
            }
        }
    }
    result[0] = 0; result[1] = 0; result[2] = ns;
}

// ------------------------------------------------------------------------------------------------
// wire numbering (compiler.rs:388-449) and gate emission (compiler.rs:451-464)
// node_wire1[node] = wire id + 1 (0 = none); nflag bit0 = input node, bit1 = output node.
// ------------------------------------------------------------------------------------------------
__global__ void k_mark_inputs(u32 n_in, const u32* __restrict__ in_nodes, u32* node_wire1, u8* nflag) {
    for (u64 i = gtid(); i < n_in; i += gstride()) {
        const u32 node = in_nodes[i];
        atomicMax(&node_wire1[node], (u32)i + 1);   // duplicate node: the later insert wins (:392-395)
        nflag[node] = 1;                            // (all writers store the same byte)
    }
}

// outputs are marked in a second launch so that "input and output" is seen whatever the order
__global__ void k_mark_outputs(u32 n_out, const u32* __restrict__ out_nodes, u8* nflag, u32* err) {
    for (u64 i = gtid(); i < n_out; i += gstride()) {
        const u32 node = out_nodes[i];
        if (nflag[node] & 1u) atomicOr(err, 1u);    // compiler.rs:363-383
        else nflag[node] = 2;
    }
}

// the payload as 16-byte records {lh, rh, out, op}: the kernels below visit gates in SORTED order, i.e. at random
// gate ids — one line per gate instead of four (built once per c2a_load_gates, outside any timed region)
__global__ void k_pack_gates(u32 n, const u32* __restrict__ lh, const u32* __restrict__ rh, const u32* __restrict__ out,
                             const u8* __restrict__ op, uint4* gate4) {
    for (u64 g = gtid(); g < n; g += gstride()) gate4[g] = make_uint4(lh[g], rh[g], out[g], op[g]);
}

// One lane per sorted position handles its three walk entries [lh, rh, out] (compiler.rs:427-430): walk index 3*pos+k.
// first[node] = first index in the walk
// first[node] = index of the node's first appearance in the walk `for gate in sorted: [lh, rh, out]` (compiler.rs:427-430).
// Every node has one writer (*dup == 0, the normal case): the sorted order is topological, so a produced node is first
// seen as its producer's `out` — nothing to compute — and only references to un-produced nodes (inputs, constants) need
// the atomicMin.  Otherwise: atomicMin over all 3n references.
__global__ void k_first_seen(u32 n, const u32* __restrict__ sorted, const uint4* __restrict__ gate4, const u32* __restrict__ prod1,
                             const u32* __restrict__ dup, u32* first) {
    const bool general = *dup != 0;
    for (u64 pos = gtid(); pos < n; pos += gstride()) {
        const uint4 g = gate4[sorted[pos]];
        const u32 i = 3u * (u32)pos;
        if (general) {
            atomicMin(&first[g.x], i);
            atomicMin(&first[g.y], i + 1);
            atomicMin(&first[g.z], i + 2);
        } else {
            if (prod1[g.x] == 0) atomicMin(&first[g.x], i);
            if (prod1[g.y] == 0) atomicMin(&first[g.y], i + 1);
        }
    }
}

__global__ void k_new_wire_flags(u32 n, const u32* __restrict__ sorted, const uint4* __restrict__ gate4,
                                 const u32* __restrict__ first, const u8* __restrict__ nflag, const u32* __restrict__ prod1,
                                 const u32* __restrict__ dup, u32* flag) {
    const bool general = *dup != 0;
    for (u64 pos = gtid(); pos < n; pos += gstride()) {
        const uint4 g = gate4[sorted[pos]];
        const u32 i = 3u * (u32)pos;
        if (general) {
            flag[i] = (first[g.x] == i && nflag[g.x] == 0) ? 1u : 0u;                 // :431-438
            flag[i + 1] = (first[g.y] == i + 1 && nflag[g.y] == 0) ? 1u : 0u;
            flag[i + 2] = (first[g.z] == i + 2 && nflag[g.z] == 0) ? 1u : 0u;
        } else {
            flag[i] = (prod1[g.x] == 0 && first[g.x] == i && nflag[g.x] == 0) ? 1u : 0u;
            flag[i + 1] = (prod1[g.y] == 0 && first[g.y] == i + 1 && nflag[g.y] == 0) ? 1u : 0u;
            flag[i + 2] = nflag[g.z] == 0 ? 1u : 0u;                                  // first seen here: its only writer
        }
    }
}

__global__ void k_assign_wires(u32 n, const u32* __restrict__ sorted, const uint4* __restrict__ gate4,
                               const u32* __restrict__ flag, const u32* __restrict__ idx, u32 n_in, u32* node_wire1) {
    for (u64 pos = gtid(); pos < n; pos += gstride()) {
        const u32 i = 3u * (u32)pos;
        const u32 f0 = flag[i], f1 = flag[i + 1], f2 = flag[i + 2];
        if (!(f0 | f1 | f2)) continue;
        const uint4 g = gate4[sorted[pos]];
        if (f0) node_wire1[g.x] = n_in + idx[i] + 1;                               // :440-441
        if (f1) node_wire1[g.y] = n_in + idx[i + 1] + 1;
        if (f2) node_wire1[g.z] = n_in + idx[i + 2] + 1;
    }
}

__global__ void k_assign_outputs(u32 n_out, const u32* __restrict__ out_nodes, u32 n_in, const u32* __restrict__ n_mid,
                                 u32* node_wire1) {
    const u32 base = n_in + *n_mid;
    for (u64 j = gtid(); j < n_out; j += gstride()) atomicMax(&node_wire1[out_nodes[j]], base + (u32)j + 1);   // :446-449
}

__global__ void k_emit(u32 n, const u32* __restrict__ sorted, const uint4* __restrict__ gate4,
                       const u32* __restrict__ node_wire1, u32* e_in0, u32* e_in1, u32* e_out, u8* e_op) {
    for (u64 pos = gtid(); pos < n; pos += gstride()) {
        const uint4 g = gate4[sorted[pos]];
        e_in0[pos] = node_wire1[g.x] - 1;
        e_in1[pos] = node_wire1[g.y] - 1;
        e_out[pos] = node_wire1[g.z] - 1;
        e_op[pos] = (u8)g.w;
    }
}

__global__ void k_unbias(u64 n, const u32* __restrict__ in1, u32* out) {
    for (u64 i = gtid(); i < n; i += gstride()) out[i] = in1[i] - 1;   // 0 -> 0xFFFFFFFF (no wire)
}

// ------------------------------------------------------------------------------------------------
// boolify: table-driven map.  One template per (op,width): entries {in0 ref, in1 ref, out ref, bool op}
// with refs = kind<<30 | index, kind 0 = A bit, 1 = B bit, 2 = O bit, 3 = aux wire of this gate.
// ------------------------------------------------------------------------------------------------
struct BoolTables {
    u32 toff[20];    // first entry of op's template
    u32 tsize[20];   // T(op,w)
    u32 taux[20];    // AUX(op,w)
};

__global__ void k_bool_sizes(u32 n, const u8* __restrict__ e_op, const BoolTables* __restrict__ T, u32* tsz, u32* asz) {
    for (u64 p = gtid(); p < n; p += gstride()) {
        const u32 o = e_op[p];
        tsz[p] = T->tsize[o];
        asz[p] = T->taux[o];
    }
}

struct BoolArgs {
    u32 n;            // arithmetic gates
    u32 width;
    u32 M;            // non-output arithmetic wires
    u64 aux_base;     // M*w
    u64 out_base;     // M*w + aux_total
    const u32* e_in0; const u32* e_in1; const u32* e_out; const u8* e_op;
    const u64* goff;  // [n+1] first boolean gate of each arithmetic gate
    const u64* aoff;  // [n+1] first aux wire (relative) of each arithmetic gate
    const uint4* tmpl;
    u32* b_in0; u32* b_in1; u32* b_out; u8* b_op;
    u32 p_first;      // first arithmetic gate (sorted position) of this launch
    u32 p_end;        // one past the last
    u64 q_bias;       // boolean gate q is stored at index q - q_bias (multiple of 4: keeps the 16-byte alignment)
};


__device__ __forceinline__ u32 bool_wire(u32 W, u32 bit, u32 width, u32 M, u64 out_base) {
    return W < M ? W * width + bit : (u32)(out_base + (u64)(W - M) * width + bit);
}

// largest i with s_goff[i] <= r
__device__ __forceinline__ u32 bool_owner(u32 r, const u32* s_goff, u32 cnt) {
    u32 lo = 0, hi = cnt;
    while (hi - lo > 1) {
        const u32 mid = (lo + hi) >> 1;
        if (s_goff[mid] <= r) lo = mid; else hi = mid;
    }
    return lo;
}

// One workgroup per CHUNK arithmetic gates.  The prologue turns each of them into four wire bases
// {A bits, B bits, O bits, aux} (so a symbolic ref resolves as base[kind] + index: one LDS read and one add),
// its template offset and the block-relative index of its first boolean gate.  Every lane then produces FOUR
// consecutive boolean gates per iteration — owner by binary search over the block's <= CHUNK offsets in LDS —
// and stores them as one 16-byte vector per SoA stream (4-byte-per-lane stores are issue-bound on gfx950, not
// bandwidth-bound); the <= 3 unaligned gates at each end of the workgroup's range go out as scalars.
// Measured alternatives that LOST in interleaved same-session A/B runs (kept out of the tree): a start-bit map
// + popcount instead of the search, a "four gates, one owner" fast path, software-pipelined template loads,
// persistent workgroups with LDS-staged packed templates, non-temporal stores (profiles/r01_boolify_ab.txt).
template <int CHUNK>
__global__ void __launch_bounds__(kThreads) k_boolify(BoolArgs A, const BoolTables* __restrict__ T) {
    __shared__ u32 s_goff[CHUNK + 1];       // first boolean gate of each arithmetic gate, relative to the block's first
    __shared__ uint4 s_base[CHUNK];         // wire bases per ref kind
    __shared__ u32 s_top[CHUNK];            // template offset
    const u32 tid = threadIdx.x;
    const u64 p0 = (u64)A.p_first + (u64)blockIdx.x * CHUNK;
    const u32 cnt = (u32)((A.p_end - p0) < (u64)CHUNK ? (A.p_end - p0) : (u64)CHUNK);
    const u64 q0 = A.goff[p0];
    for (u32 i = tid; i <= cnt; i += kThreads) s_goff[i] = (u32)(A.goff[p0 + i] - q0);
    for (u32 i = tid; i < cnt; i += kThreads) {
        const u32 wa = A.e_in0[p0 + i], wb = A.e_in1[p0 + i], wo = A.e_out[p0 + i];
        s_base[i] = make_uint4(bool_wire(wa, 0, A.width, A.M, A.out_base), bool_wire(wb, 0, A.width, A.M, A.out_base),
                               bool_wire(wo, 0, A.width, A.M, A.out_base), (u32)(A.aux_base + A.aoff[p0 + i]));
        s_top[i] = T->toff[A.e_op[p0 + i]];
    }
    __syncthreads();
    const u32 total = s_goff[cnt];                       // boolean gates of this block (< 2^32 by construction)
    const u32 head = (u32)((4 - (q0 & 3)) & 3);          // unaligned gates before the first 16-byte boundary
    const u32 r0 = head < total ? head : total;
    const u32 r1 = r0 + ((total - r0) & ~3u);
    const u32* base_words = reinterpret_cast<const u32*>(s_base);
    typedef u32 u32x4 __attribute__((vector_size(16)));
    // ---- aligned body: groups of 4 (r = block-relative index)
    for (u32 r = r0 + 4u * tid; r < r1; r += 4u * kThreads) {
        u32 idx = bool_owner(r, s_goff, cnt);
        u32 start = s_goff[idx], bound = s_goff[idx + 1], top = s_top[idx];
        u32 v0[4], v1[4], v2[4], vop = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            while (r + j >= bound) { ++idx; start = bound; bound = s_goff[idx + 1]; top = s_top[idx]; }
            const uint4 e = A.tmpl[top + (r + j - start)];
            v0[j] = base_words[idx * 4 + (e.x >> 30)] + (e.x & 0x3FFFFFFFu);
            v1[j] = base_words[idx * 4 + (e.y >> 30)] + (e.y & 0x3FFFFFFFu);
            v2[j] = base_words[idx * 4 + (e.z >> 30)] + (e.z & 0x3FFFFFFFu);
            vop |= (e.w & 0xFFu) << (8 * j);
        }
        const u64 q = q0 + r - A.q_bias;
        *reinterpret_cast<u32x4*>(A.b_in0 + q) = u32x4{v0[0], v0[1], v0[2], v0[3]};
        *reinterpret_cast<u32x4*>(A.b_in1 + q) = u32x4{v1[0], v1[1], v1[2], v1[3]};
        *reinterpret_cast<u32x4*>(A.b_out + q) = u32x4{v2[0], v2[1], v2[2], v2[3]};
        *reinterpret_cast<u32*>(A.b_op + q) = vop;
    }
    // ---- unaligned head [0,r0) and tail [r1,total): at most 3 + 3 gates
    {
        const u32 nh = r0, nt = total - r1;
        if (tid < nh + nt) {
            const u32 r = tid < nh ? tid : r1 + (tid - nh);
            const u32 lo = bool_owner(r, s_goff, cnt);
            const uint4 e = A.tmpl[s_top[lo] + (r - s_goff[lo])];
            const u64 q = q0 + r - A.q_bias;
            A.b_in0[q] = base_words[lo * 4 + (e.x >> 30)] + (e.x & 0x3FFFFFFFu);
            A.b_in1[q] = base_words[lo * 4 + (e.y >> 30)] + (e.y & 0x3FFFFFFFu);
            A.b_out[q] = base_words[lo * 4 + (e.z >> 30)] + (e.z & 0x3FFFFFFFu);
            A.b_op[q] = (u8)e.w;
        }
    }
}

// order-sensitive 64-bit checksum of a u32 stream: sum over i of mix(i, v[i]) (commutative combine of
// position-salted hashes => parallel, deterministic).  Used by the full-size parity tests.
__device__ __forceinline__ u64 mix64(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}
// ------------------------------------------------------------------------------------------------
// Functional check at scale — the reference's simulation harness (tests/integration.rs:191-237: run the Bristol
// circuit on inputs, compare outputs) as HIP kernels, for circuits far too large for a CPU simulator: the
// arithmetic circuit and its boolean image are both evaluated on 64 pseudo-random input vectors and EVERY
// arithmetic wire is compared with its w boolean wires.  Gates of one reverse-Kahn level are independent and
// producers always sit in a higher level than their consumers, so the peel's level lists, walked from the last
// level down to level 0, are a ready-made parallel schedule.
// Semantics: tests/integration.rs:94-115 taken mod 2^w; where the reference would panic the value is defined by
// DESIGN.md §5.2 (wrap-around, x/0 = 2^w-1, x%0 = x, shifts by >= w give 0).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u64 eval_arith_op(u32 op, u64 a, u64 b, u32 w, u64 mk) {
    u64 r = 0;
    switch (op) {
    case 0: r = a + b; break;                       // AAdd
    case 1: case 12: r = b ? a / b : mk; break;     // ADiv, AIntDiv
    case 2: r = a == b; break;                      // AEq
    case 3: r = a >= b; break;                      // AGEq
    case 4: r = a > b; break;                       // AGt
    case 5: r = a <= b; break;                      // ALEq
    case 6: r = a < b; break;                       // ALt
    case 7: r = a * b; break;                       // AMul
    case 8: r = a != b; break;                      // ANeq
    case 9: r = a - b; break;                       // ASub
    case 10: r = a ^ b; break;                      // AXor
    case 11: { u64 base = a, acc = 1, ex = b; while (ex) { if (ex & 1) acc *= base; base *= base; ex >>= 1; } r = acc; } break;   // APow
    case 13: r = b ? a % b : a; break;              // AMod
    case 14: r = b >= w ? 0 : a << b; break;        // AShiftL
    case 15: r = b >= w ? 0 : a >> b; break;        // AShiftR
    case 16: r = (a != 0) || (b != 0); break;       // ABoolOr
    case 17: r = (a != 0) && (b != 0); break;       // ABoolAnd
    case 18: r = a | b; break;                      // ABitOr
    case 19: r = a & b; break;                      // ABitAnd
    default: r = 0;
    }
    return r & mk;
}

__global__ void k_eval_mark_produced(u32 n, const u32* __restrict__ e_out, u8* produced) {
    for (u64 p = gtid(); p < n; p += gstride()) produced[e_out[p]] = 1;
}
__global__ void k_eval_inverse(u32 n, const u32* __restrict__ sorted, u32* spos) {
    for (u64 p = gtid(); p < n; p += gstride()) spos[sorted[p]] = (u32)p;
}

// free wires (inputs, constants: not produced by any gate) get pseudo-random w-bit values, 64 vectors per wire
// (vectors 0..3 are the corner values 0, 2^w-1, 1, 2^(w-1)); boolean image = the transposed bits
__global__ void k_eval_init(u32 wire_count, u32 width, u32 M, u64 out_base, u64 seed, const u8* __restrict__ produced,
                            u64* aval, u64* bval) {
    const u64 mk = width >= 64 ? ~0ull : ((1ull << width) - 1ull);
    for (u64 i = gtid(); i < (u64)wire_count * 64; i += gstride()) {
        const u32 W = (u32)(i >> 6), t = (u32)(i & 63);
        if (produced[W]) continue;
        u64 v = mix64(seed ^ ((u64)W << 8) ^ t) & mk;
        if (t == 0) v = 0; else if (t == 1) v = mk; else if (t == 2) v = 1 & mk; else if (t == 3) v = (1ull << (width - 1)) & mk;
        aval[i] = v;
    }
    for (u64 i = gtid(); i < (u64)wire_count * width; i += gstride()) {
        const u32 W = (u32)(i / width), bit = (u32)(i - (u64)W * width);
        if (produced[W]) continue;
        const u64 mk2 = mk;
        u64 word = 0;
        for (u32 t = 0; t < 64; ++t) {
            u64 v = mix64(seed ^ ((u64)W << 8) ^ t) & mk2;
            if (t == 0) v = 0; else if (t == 1) v = mk2; else if (t == 2) v = 1 & mk2; else if (t == 3) v = (1ull << (width - 1)) & mk2;
            word |= ((v >> bit) & 1ull) << t;
        }
        bval[bool_wire(W, bit, width, M, out_base)] = word;
    }
}

// one level: lane per (gate, vector) for the arithmetic side, lane per gate for its boolean template
// level lists for the level-parallel evaluation: counting sort of the tree nodes by reverse Kahn level (meta.w >> 1)
__global__ void k_level_hist(u32 n, const uint4* __restrict__ meta, u32* lcount) {
    for (u64 i = gtid(); i < n; i += gstride()) atomicAdd(&lcount[meta[i].w >> 1], 1u);
}
__global__ void k_level_scatter(u32 n, const uint4* __restrict__ meta, const u32* __restrict__ order, const u32* __restrict__ lbase,
                                u32* cursor, u32* lorder) {
    for (u64 i = gtid(); i < n; i += gstride()) {
        const u32 lv = meta[i].w >> 1;
        lorder[lbase[lv] + atomicAdd(&cursor[lv], 1u)] = order[i];
    }
}

__global__ void k_eval_level_arith(u32 lo, u32 cnt, u32 width, const u32* __restrict__ order, const u32* __restrict__ spos,
                                   const u32* __restrict__ e_in0, const u32* __restrict__ e_in1, const u32* __restrict__ e_out,
                                   const u8* __restrict__ e_op, u64* aval) {
    const u64 mk = width >= 64 ? ~0ull : ((1ull << width) - 1ull);
    for (u64 i = gtid(); i < (u64)cnt * 64; i += gstride()) {
        const u32 p = spos[order[lo + (u32)(i >> 6)]], t = (u32)(i & 63);
        aval[(u64)e_out[p] * 64 + t] = eval_arith_op(e_op[p], aval[(u64)e_in0[p] * 64 + t], aval[(u64)e_in1[p] * 64 + t], width, mk);
    }
}
__global__ void k_eval_level_bool(u32 lo, u32 cnt, const u32* __restrict__ order, const u32* __restrict__ spos,
                                  const u64* __restrict__ goff, const u32* __restrict__ b_in0, const u32* __restrict__ b_in1,
                                  const u32* __restrict__ b_out, const u8* __restrict__ b_op, u64* bval) {
    for (u64 i = gtid(); i < cnt; i += gstride()) {
        const u32 p = spos[order[lo + (u32)i]];
        for (u64 k = goff[p]; k < goff[p + 1]; ++k) {
            const u64 a = bval[b_in0[k]], b = bval[b_in1[k]];
            const u32 o = b_op[k];
            bval[b_out[k]] = o == 0 ? (a ^ b) : (o == 1 ? (a & b) : ~a);
        }
    }
}
// every (arithmetic wire, vector): value == the w boolean wires read back
__global__ void k_eval_compare(u32 wire_count, u32 width, u32 M, u64 out_base, const u64* __restrict__ aval,
                               const u64* __restrict__ bval, ull* mismatches) {
    u64 bad = 0;
    for (u64 i = gtid(); i < (u64)wire_count * 64; i += gstride()) {
        const u32 W = (u32)(i >> 6), t = (u32)(i & 63);
        u64 v = 0;
        for (u32 bit = 0; bit < width; ++bit) v |= ((bval[bool_wire(W, bit, width, M, out_base)] >> t) & 1ull) << bit;
        bad += v != aval[i];
    }
    if (bad) atomicAdd(mismatches, (ull)bad);
}

__global__ void k_checksum_u32(u64 n, const u32* __restrict__ v, ull* acc) {
    u64 local = 0;
    for (u64 i = gtid(); i < n; i += gstride()) local += mix64((i << 32) ^ (i >> 32) ^ ((u64)v[i] * 0x9E3779B97F4A7C15ULL));
    if (local) atomicAdd(acc, (ull)local);
}
__global__ void k_checksum_u8(u64 n, const u8* __restrict__ v, ull* acc) {
    u64 local = 0;
    for (u64 i = gtid(); i < n; i += gstride()) local += mix64((i << 32) ^ (i >> 32) ^ ((u64)v[i] * 0x9E3779B97F4A7C15ULL));
    if (local) atomicAdd(acc, (ull)local);
}

}  // namespace c2a

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
// theirs (reverse Kahn order from the sinks) by comparing the consumers' paths, held as bit strings in 512-byte
// self-validating node records (one memory round trip per tournament).  The whole peel is ONE dataflow launch
// (c2a_peel.h: a wave goes on with the producer it claimed, ticket queues hand over the rest, no level barrier).
// The post-order index of every gate then comes from an Euler tour + list ranking.
#pragma once
#include "c2a_platform.h"

namespace c2a {

constexpr int kThreads = 256;

__device__ __forceinline__ u64 gtid() { return (u64)blockIdx.x * blockDim.x + threadIdx.x; }
__device__ __forceinline__ u64 gstride() { return (u64)gridDim.x * blockDim.x; }

// XCD-AWARE grid-stride.  Workgroup b runs on XCD b % 8 (observed on gfx950, not promised: this is for speed only), and every
// XCD has an L2 of its own.  A kernel that walks the gates in rank order and looks a bounded distance back (a gate's producers,
// their consumer lists) has a moving window of nearby data; with the plain grid-stride all eight XCDs sweep the SAME span at
// once and each L2 has to hold the whole span's window (6 MB for a sweep of 4 096 workgroups: it does not fit).  Here every
// XCD takes one contiguous eighth of the range and sweeps it with its own workgroups: the window an L2 sees is an eighth as wide.
#ifndef C2A_XCD_AWARE
#define C2A_XCD_AWARE 1
#endif
struct XcdSweep { u64 i, end, step; };
__device__ __forceinline__ XcdSweep xcd_sweep(u64 n) {
#if C2A_XCD_AWARE
#ifdef C2A_EMULATE
    constexpr u64 kXcdMin = 1024;                  // (so that the CPU suite walks this mapping too)
#else
    constexpr u64 kXcdMin = 8ull * 4096ull;
#endif
    if ((gridDim.x & 7u) == 0u && n >= kXcdMin) {
        const u64 xcd = blockIdx.x & 7u, j = blockIdx.x >> 3, wgs = gridDim.x >> 3;
        const u64 share = (((n + 7) / 8) + blockDim.x - 1) / blockDim.x * blockDim.x;
        const u64 lo = xcd * share, hi = lo + share < n ? lo + share : n;
        return XcdSweep{lo + j * blockDim.x + threadIdx.x, hi, wgs * blockDim.x};
    }
#endif
    return XcdSweep{gtid(), n, gstride()};
}

}  // namespace c2a

#include "c2a_peel.h"       // (also the agent-scope access helpers the scan below uses)

namespace c2a {

constexpr u64 kScanAgg = 1ull << 62, kScanPre = 2ull << 62, kScanVal = (1ull << 62) - 1ull;
// (wave 0 of a workgroup: publish this tile's sum, find the sum of all tiles before it.  Descriptors: value | state << 62 —
// 1: the tile's own sum, 2: the sum up to and including the tile —, `stride` words apart; lane l looks at tile look - l, the
// window up to the first tile whose sum is inclusive must be all there, else it is looked at again)
__device__ __forceinline__ u64 tile_lookback(u64* desc, u32 stride, u32 tile, u64 tile_total, u32 lane) {
    u64 excl = 0;
    if (tile == 0) { if (lane == 0) st_nw(&desc[0], kScanPre | tile_total); return 0; }
    if (lane == 0) st_nw(&desc[(u64)tile * stride], kScanAgg | tile_total);
    i64 look = (i64)tile - 1;
    for (;;) {
        const i64 t = look - (i64)lane;
        const u64 d = t >= 0 ? ld_nw(&desc[(u64)t * stride]) : kScanPre;
        const u64 pre = __ballot((d >> 62) == 2u);
        const u64 notready = __ballot((d >> 62) == 0u);
        const u32 upto = pre ? (u32)__builtin_ctzll(pre) : 63u;              // lanes 0..upto are summed
        const u64 need = upto == 63u ? ~0ull : ((2ull << upto) - 1ull);
        if (notready & need) continue;                     // (looked at again at once: the tiles before this one are running)
        u64 x = lane <= upto ? (d & kScanVal) : 0ull;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off, 64);
        excl += x;
        if (pre) break;
        look -= 64;
    }
    if (lane == 0) st_nw(&desc[(u64)tile * stride], kScanPre | ((excl + tile_total) & kScanVal));
    return excl;
}

// ------------------------------------------------------------------------------------------------
// exclusive scan in ONE launch (decoupled look-back), STREAMING: a tile is 256 threads x 4 rounds x 4 consecutive elements,
// so that a lane's accesses are 16-byte vectors next to its neighbours' (the first version gave every thread 16 consecutive
// elements: 4-byte accesses 64 bytes apart, one cache line per lane per instruction — the texture path, not HBM, set its pace);
// a workgroup is four such tiles side by side — one ticket for the four (the tickets come from one word, ~11 ns apiece: a
// ticket per 2 048-element tile was 55 us of a 95 us scan), each quarter publishes its own sum and looks back on its own (tiles
// worked off one after the other by one workgroup would chain the look-backs: every tile would wait for the whole tile before).  NC sums are carried at once (the boolify plan
// scans the template sizes and the aux-wire counts of the same gates together); the element values come from a functor (a plain
// array, template sizes from op bytes, bit counts of a bitmap's words: the array a three-kernel scan would read is never
// written).  out[c][n] = the total.
// ------------------------------------------------------------------------------------------------
constexpr int kScanQuarter = 256, kScanRounds = 4, kScanVec = 4;
constexpr int kScanTile = kScanQuarter * kScanRounds * kScanVec;      // 4 096 elements
// tiles per ticket = 256-thread parts of a workgroup: four for one sum, two for two (the two-sum scan needs more than the 128
// registers a 1 024-thread workgroup leaves a thread: it spilled 300 bytes per lane)
template <int NC> struct ScanGeom { static constexpr int kGroup = NC == 1 ? 4 : 2; static constexpr int kThreads = kScanQuarter * kGroup; };

template <int NC, class F, typename TOut>
__global__ void __launch_bounds__(ScanGeom<NC>::kThreads) k_scan_stream(u64 n, F f, TOut* out0, TOut* out1, u64* desc, u32* counter) {
    constexpr int kScanGroup = ScanGeom<NC>::kGroup;
    __shared__ u32 s_group;
    __shared__ u64 s_part[kScanGroup][NC][kScanRounds][kScanQuarter / 64];
    __shared__ u64 s_excl[kScanGroup][NC];
    const u32 q = threadIdx.x / kScanQuarter, tid = threadIdx.x % kScanQuarter, lane = tid & 63u, wv = tid >> 6;
    if (threadIdx.x == 0) s_group = atomicAdd(counter, 1u);
    __syncthreads();
    const u64 n_tiles = (n + kScanTile - 1) / kScanTile;
    const u64 tile = (u64)s_group * kScanGroup + q;         // (a tile beyond the end: its lanes see no element and publish nothing)
    const bool live = tile < n_tiles;
    const u64 tbase = tile * kScanTile;
    u32 v[kScanRounds][kScanVec][NC];                      // (element values are 32-bit, sums 64-bit: with 64-bit values the two-sum scan spilled)
    u64 rs[kScanRounds][NC];
#pragma unroll
    for (int j = 0; j < kScanRounds; ++j) {
#pragma unroll
        for (int c = 0; c < NC; ++c) rs[j][c] = 0;
#pragma unroll
        for (int e = 0; e < kScanVec; ++e) {
            const u64 idx = tbase + (u64)j * (kScanQuarter * kScanVec) + (u64)tid * kScanVec + e;
            u32 x[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) x[c] = 0;
            if (live && idx < n) f(idx, x);
#pragma unroll
            for (int c = 0; c < NC; ++c) { v[j][e][c] = x[c]; rs[j][c] += x[c]; }
        }
    }
    // inclusive scan of the per-thread sums of every round inside the wave; the waves' totals meet in LDS
    u64 inc[kScanRounds][NC];
#pragma unroll
    for (int j = 0; j < kScanRounds; ++j)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            u64 t = rs[j][c];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const u64 o = __shfl_up(t, off, 64);
                if (lane >= (u32)off) t += o;
            }
            inc[j][c] = t;
            if (lane == 63) s_part[q][c][j][wv] = t;
        }
    __syncthreads();
    // element order inside a tile: round, then wave, then lane
    u64 before[kScanRounds][NC], tile_total[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        u64 run = 0;
#pragma unroll
        for (int j = 0; j < kScanRounds; ++j)
#pragma unroll
            for (int w = 0; w < kScanQuarter / 64; ++w) { if ((u32)w == wv) before[j][c] = run; run += s_part[q][c][j][w]; }
        tile_total[c] = run;
    }
    if (live && wv == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const u64 e = tile_lookback(desc + c, (u32)NC, (u32)tile, tile_total[c], lane);
            if (lane == 0) s_excl[q][c] = e;
        }
    }
    __syncthreads();
    if (!live) return;
#pragma unroll
    for (int j = 0; j < kScanRounds; ++j) {
        u64 run[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) run[c] = s_excl[q][c] + before[j][c] + inc[j][c] - rs[j][c];
#pragma unroll
        for (int e = 0; e < kScanVec; ++e) {
            const u64 idx = tbase + (u64)j * (kScanQuarter * kScanVec) + (u64)tid * kScanVec + e;
            if (idx < n) {
                out0[idx] = (TOut)run[0];
                if (NC > 1) out1[idx] = (TOut)run[NC - 1];
            }
#pragma unroll
            for (int c = 0; c < NC; ++c) run[c] += v[j][e][c];
        }
    }
    if (tile + 1 == n_tiles && tid == 0) {           // the last tile: the totals
        out0[n] = (TOut)(s_excl[q][0] + tile_total[0]);
        if (NC > 1) out1[n] = (TOut)(s_excl[q][NC - 1] + tile_total[NC - 1]);
    }
}

// ONE launch that zeroes everything a build wants zeroed (scalars, node flags, scan descriptors, the dataflow launch's control
// words and tickets, the root bitmap, the event bits of the numbering): thirteen hipMemsetAsync per build — a launch and a gap
// each — were ~70 us of a 13 ms step.  Regions are 16-byte aligned and padded; end[k] = 16-byte words up to and including region k.
constexpr int kClearMax = 12;
struct ClearList { uint4* p[kClearMax]; u64 end[kClearMax]; u32 cnt; };
__global__ void k_clear(ClearList L) {
    const u64 total = L.end[L.cnt - 1];
    for (u64 i = gtid(); i < total; i += gstride()) {
        u32 k = 0;
        while (i >= L.end[k]) ++k;
        L.p[k][i - (k ? L.end[k - 1] : 0ull)] = make_uint4(0u, 0u, 0u, 0u);
    }
}

// element functors
struct ScanFromU32 { const u32* in; __device__ __forceinline__ void operator()(u64 i, u32* x) const { x[0] = in[i]; } };

// ------------------------------------------------------------------------------------------------
// argument validation of c2a_load_gates on the device, behind the copy (the host loop over all gates it replaces was 5.8 of the
// 16 ms a 10 M-gate load took): the first gate whose node ids do not address the node table or whose op is no AGateType (the
// host formats the message from its own copy of that gate), and the gates per type (the totals of a boolify plan)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kThreads) k_validate(u32 n, const u32* __restrict__ lh, const u32* __restrict__ rh, const u32* __restrict__ out,
                                                       const u8* __restrict__ op, u32 n_nodes, u32 n_types, u32* first_bad, u32* hist) {
    __shared__ u32 s_h[32];
    if (threadIdx.x < 32) s_h[threadIdx.x] = 0u;
    __syncthreads();
    u32 bad = C2A_NONE;
    for (u64 g = gtid(); g < n; g += gstride()) {
        const u32 o = op[g];
        if (lh[g] >= n_nodes || rh[g] >= n_nodes || out[g] >= n_nodes || o >= n_types) bad = bad < (u32)g ? bad : (u32)g;
        else atomicAdd(&s_h[o], 1u);
    }
    if (bad != C2A_NONE) atomicMin(first_bad, bad);
    __syncthreads();
    if (threadIdx.x < 32 && s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_h[threadIdx.x]);
}

// ------------------------------------------------------------------------------------------------
// graph prep
// ------------------------------------------------------------------------------------------------
// producer[node] = last gate writing it (compiler.rs:401-406: a later insert overwrites an earlier one).
// NO ATOMICS on the way in: every gate stores its payload into its OUT NODE's record — one scattered 16-byte store, tagged with
// the number of this build (the records are never cleared between builds) — and k_relabel, which walks the node table anyway,
// counts the tagged records: as many as gates <=> every node has one writer, which is what the reference's front-end produces.
// Fewer <=> two gates wrote one node (*dup, raised by k_relabel): the two kernels behind it then redo the producer map with
// atomicMax (the last writer = the largest gate id), the relabelling is the identity and the wire numbering takes its general
// path.  (Round 3: 10 M scattered atomicMax, 0.40 ms, for a case that does not occur.)
// What the rest of the sort keeps per gate (consumer count, claim tickets, tree children) is reset on the way: three coalesced
// stores instead of three clears of their own.
__global__ void k_producer(u32 n, const u32* __restrict__ lh, const u32* __restrict__ rh, const u32* __restrict__ out, const u8* __restrict__ op,
                           uint4* nrec, u32 build, u32* cons_cnt, u32* fill, uint2* child) {
    for (u64 g = gtid(); g < n; g += gstride()) {
        cons_cnt[g] = 0u;
        if (fill) { fill[g] = 0u; child[g] = make_uint2(C2A_NONE, C2A_NONE); }
        nrec[out[g]] = make_uint4(lh[g], rh[g], (u32)g, (u32)op[g] | (build << 8));
    }
}
__global__ void k_dup_clear(u32 n_nodes, const u32* __restrict__ dup, u32* prod1) {
    if (*dup == 0u) return;
    for (u64 v = gtid(); v < n_nodes; v += gstride()) prod1[v] = 0u;
}
__global__ void k_dup_producer(u32 n, const u32* __restrict__ dup, const u32* __restrict__ out, u32* prod1) {
    if (*dup == 0u) return;
    for (u64 g = gtid(); g < n; g += gstride()) atomicMax(&prod1[out[g]], (u32)g + 1u);
}

// RELABELLING.  Gate ids arrive in no particular order (the headline input permutes them), but NODE ids are handed out by a
// counter as the circuit is built (compiler.rs:497-500): the position of a gate's out node among the produced nodes — its
// RANK — follows the creation order, producers sit a bounded distance before their consumers, and what the sort does per edge
// (deps, consumer lists, node records) becomes local in rank space.  One pass over the node table: rank = number of produced
// nodes before this one; orig[rank] = the gate, gate4[rank] = its payload {lh node, rh node, out node, op}, prod1[node] =
// rank + 1 (0: nobody produces it).  From here to k_rank_final every "gate id" is a rank; the DFS roots are compared by orig[]
// (topological_sort.rs:11-13 walks gate ids).  With duplicate writers nothing is moved: rank = gate id (k_deps copies the
// payload as it lies).
// One launch (decoupled look-back over tiles of 4 096 nodes, like k_scan_1pass) in a STRIPED arrangement — lane l of a round
// looks at node base + l — so that the flags are a ballot, a node's rank inside the wave a population count, and every access
// streams: the node table and the records are read as whole lines, orig[] / gate4[] are written as whole lines (consecutive
// produced nodes have consecutive ranks).
constexpr int kRelThreads = 256, kRelRounds = 16, kRelTile = kRelThreads * kRelRounds;
__global__ void __launch_bounds__(kRelThreads) k_relabel(u32 n_nodes, u32 n, u32 build, u32* prod1, const uint4* __restrict__ nrec, u32* dup,
                                                         u32* orig, uint4* gate4, u64* desc, u32* counter, u32* node_wire1, u32* first_seen) {
    __shared__ u32 s_tile, s_wave[kRelThreads / 64];
    __shared__ u64 s_excl;
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    if (tid == 0) s_tile = atomicAdd(counter, 1u);
    __syncthreads();
    const u32 tile = s_tile;
    const u64 base = (u64)tile * kRelTile + (u64)wv * (64 * kRelRounds) + lane;
    const u64 lt_mask = (1ull << lane) - 1ull;
    u32 pre[kRelRounds];
    uint4 rec[kRelRounds];
#pragma unroll
    for (int i = 0; i < kRelRounds; ++i) { const u64 v = base + (u64)i * 64; rec[i] = v < n_nodes ? nrec[v] : make_uint4(0u, 0u, 0u, ~build << 8); }
    u32 cnt = 0;
#pragma unroll
    for (int i = 0; i < kRelRounds; ++i) {
        const u64 bal = __ballot((rec[i].w >> 8) == build);      // a record of THIS build: the node is produced
        pre[i] = cnt + (u32)__popcll(bal & lt_mask);
        cnt += (u32)__popcll(bal);
    }
    if (lane == 0) s_wave[wv] = cnt;
    __syncthreads();
    u32 wave_base = 0, tile_total = 0;
#pragma unroll
    for (u32 w = 0; w < kRelThreads / 64; ++w) { const u32 v = s_wave[w]; wave_base += w < wv ? v : 0u; tile_total += v; }
    if (wv == 0) {
        const u64 e = tile_lookback(desc, 1u, tile, tile_total, lane);
        if (lane == 0) s_excl = e;
    }
    __syncthreads();
    const u32 first = (u32)s_excl + wave_base;
    // (the tile that ends the table knows how many nodes are produced: fewer than gates = a node with two writers)
    if (tid == 0 && (u64)(tile + 1) * kRelTile >= n_nodes && (u32)s_excl + tile_total != n) *dup = 1u;
#pragma unroll
    for (int i = 0; i < kRelRounds; ++i) {
        const u64 v = base + (u64)i * 64;
        if (v >= n_nodes) continue;
        // (what the wire numbering starts from, per node — no wire yet, not seen yet —: this pass streams over the node table anyway; a launch
        // of its own, k_node_init, only when c2a_assign_wires is called again on a sorted circuit)
        node_wire1[v] = 0u; first_seen[v] = 0xFFFFFFFFu;
        if ((rec[i].w >> 8) != build) { prod1[v] = 0u; continue; }
        const u32 rank = first + pre[i];
        if (rank < n) {                            // (always, unless two gates wrote one node: then all of this is redone)
            orig[rank] = rec[i].z;
            gate4[rank] = make_uint4(rec[i].x, rec[i].y, (u32)v, rec[i].w & 0xFFu);
        }
        prod1[v] = rank + 1u;
    }
}

// deps closure (compiler.rs:408-421) + consumer counts, in rank space.  dep1 is dropped when equal to dep0: a second
// visit of the same gate is a no-op in the DFS (topological_sort.rs:30-32).
// The payload record gets five flags {lh node un-produced << 8 | rh node un-produced << 9 | out node is an IO node << 10 |
// lh node is a CONSTANT-like node (un-produced and no IO node: it gets its wire where the walk first sees it) << 11 | rh
// likewise << 12}: what the numbering kernels want to know about a gate's nodes comes along instead of costing scattered reads
// (the flags alone once more as a byte per gate: the passes of the POSITIONAL NUMBERING below that only want them).
// Nothing is counted or listed here: the reference's front-end makes one named constant node per literal and template context
// (process.rs:558-579), i.e. 5-30 % of a circuit's gates read one — a list appended to through one counter would be a million
// same-address atomics.
constexpr u32 kGateLhUnprod = 0x100u, kGateRhUnprod = 0x200u, kGateOutIO = 0x400u, kGateLhConst = 0x800u, kGateRhConst = 0x1000u;
// HOT PRODUCERS.  The consumer tickets are returning atomics on cons_cnt[producer], and atomics on ONE word go one at a time,
// ~11 ns apiece: a produced node that a million gates read (a hub: compiler.rs:408-421 allows any fan-out) was 11 ms of
// same-address atomics in this kernel.  Matching the lanes' producers against each other in every wave costs every graph
// 0.1 ms of ballots (measured); instead the hot producers are found ON THE WAY: the lane that draws ticket kHotEvery - 1 —
// exactly one lane per producer with that many consumers — appends the producer to a short list (hot[0] = entries claimed,
// hot[1 + e] = rank + 1, 0 = not written yet), every wave looks at the list's length once per round (one word; zero on a
// circuit without such a hub), and the lanes that name a listed producer do NOT draw a ticket: the wave counts them (lane e of
// run_v = its pending tickets on entry e), draws ONE range per entry when it has seen all its gates and fills their slots in
// on a second walk over its own gates.  A fan-out of 10^6: 16 384 single tickets + one per wave = 0.4 ms instead of 11.
constexpr u32 kHotMax = 64, kHotEvery = 1u << 14, kHotPending = 0x80000000u;
__global__ void __launch_bounds__(kThreads) k_deps(u32 n, const u32* __restrict__ lh, const u32* __restrict__ rh, const u32* __restrict__ out, const u8* __restrict__ op,
                       const u32* __restrict__ dup, const u32* __restrict__ prod1, const u8* __restrict__ nflag, u32* orig, uint4* gate4,
                       u32* dep0, u32* dep1, u32* cons_cnt, u32* eslot, u8* gflag, u32* hot, u32 hot_every) {
    const bool ident = *dup != 0u;
    const XcdSweep R = xcd_sweep(n);
    const u32 lane = threadIdx.x & 63u;
    const u64 lt_mask = (1ull << lane) - 1ull;
    u32 run_v = 0;                                 // lane e: tickets this wave owes the hot producer of entry e
    bool any_pending = false;                      // (wave-uniform)
    // (every lane of a wave goes round the same number of times: the hot producers are handled wave by wave)
    for (u64 gb = R.i - lane; gb < R.end; gb += R.step) {
        const u64 g = gb + lane;
        const bool live = g < R.end;
        u32 d0 = C2A_NONE, d1 = C2A_NONE;
        if (live) {
            uint4 r;
            if (ident) { r = make_uint4(lh[g], rh[g], out[g], (u32)op[g]); orig[g] = (u32)g; }
            else r = gate4[g];
            const u32 p0 = prod1[r.x], p1 = prod1[r.y];
            u32 w = (r.w & 0xFFu) | (p0 ? 0u : kGateLhUnprod) | (p1 ? 0u : kGateRhUnprod) | ((nflag[r.z] & 3u) ? kGateOutIO : 0u);
            if (!p0 && !(nflag[r.x] & 3u)) w |= kGateLhConst;             // (an input or output node: its wire is fixed)
            if (!p1 && !(nflag[r.y] & 3u)) w |= kGateRhConst;
            gate4[g] = make_uint4(r.x, r.y, r.z, w);
            gflag[g] = (u8)(w >> 8);
            d0 = p0 ? p0 - 1 : C2A_NONE;
            d1 = p1 ? p1 - 1 : C2A_NONE;
            if (d1 == d0) d1 = C2A_NONE;
            dep0[g] = d0;
            dep1[g] = d1;
        }
        // eslot[2g + l] = index of the edge (g, l) in its producer's consumer list
        bool t0 = d0 != C2A_NONE, t1 = d1 != C2A_NONE;
        u32 s0 = 0, s1 = 0;
        // (ONE lane's view of the list for the whole wave — the list grows while the kernel runs)
        u32 hc = rdlane(ld_a32(&hot[0]), 0);
        if (hc) {
            hc = hc < kHotMax ? hc : kHotMax;
            for (u32 e = 0; e < hc; ++e) {
                const u32 hv = rdlane(ld_a32(&hot[1 + e]), 0);
                const bool h0 = t0 && d0 + 1u == hv, h1 = t1 && d1 + 1u == hv;      // (hv 0 — not written yet — matches nothing: d + 1 of a wanted ticket is never 0)
                const u64 m0 = __ballot(h0), m1 = __ballot(h1);
                if ((m0 | m1) == 0ull) continue;
                const u32 r = (u32)__shfl((int)run_v, (int)e, 64);
                if (h0) { s0 = kHotPending | (e << 24) | (r + (u32)__popcll(m0 & lt_mask)); t0 = false; }
                if (h1) { s1 = kHotPending | (e << 24) | (r + (u32)__popcll(m0) + (u32)__popcll(m1 & lt_mask)); t1 = false; }
                if (lane == e) run_v += (u32)__popcll(m0) + (u32)__popcll(m1);
                any_pending = true;
            }
        }
        if (t0) { s0 = atomicAdd(&cons_cnt[d0], 1u); if (s0 == hot_every - 1u) { const u32 i = atomicAdd(&hot[0], 1u); if (i < kHotMax) st_a32(&hot[1 + i], d0 + 1u); } }
        if (t1) { s1 = atomicAdd(&cons_cnt[d1], 1u); if (s1 == hot_every - 1u) { const u32 i = atomicAdd(&hot[0], 1u); if (i < kHotMax) st_a32(&hot[1 + i], d1 + 1u); } }
        if (live) { eslot[2 * g] = s0; eslot[2 * g + 1] = s1; }
    }
    if (!any_pending) return;
    // one range of tickets per hot entry for the whole wave, then the slots of the edges that waited for it
    u32 base_v = 0;
    if (run_v) base_v = atomicAdd(&cons_cnt[ld_a32(&hot[1 + lane]) - 1u], run_v);
    for (u64 gb = R.i - lane; gb < R.end; gb += R.step) {
        const u64 g = gb + lane;
        const bool live = g < R.end;
        const u32 s0 = live ? eslot[2 * g] : 0u, s1 = live ? eslot[2 * g + 1] : 0u;
        const u32 b0 = (u32)__shfl((int)base_v, (int)((s0 >> 24) & 63u), 64), b1 = (u32)__shfl((int)base_v, (int)((s1 >> 24) & 63u), 64);
        if (s0 & kHotPending) eslot[2 * g] = b0 + (s0 & 0xFFFFFFu);
        if (s1 & kHotPending) eslot[2 * g + 1] = b1 + (s1 & 0xFFFFFFu);
    }
}

// ------------------------------------------------------------------------------------------------
// post-order numbering: Euler tour of the DFS tree + list ranking (random splitters)
// ------------------------------------------------------------------------------------------------
// DFS roots = tree nodes without a parent, in ascending ORIGINAL gate id (topological_sort.rs:11-13) — the tree lives in
// rank space, so: a bit per original id (1.25 MB for 10 M gates: the scattered atomics stay on chip), a scan over the bitmap's
// words, and every root finds its index by a look-up in the two small arrays.
// (*ok: the dataflow launch in front ended cleanly and left no gate behind — k_root_bits works that out —; the order stage is queued BEHIND it
// without a host round trip, and does nothing when it did not: the tree entries it would read are not there)
// This launch also POSTS what the dataflow launch in front of it reported (block 0; a launch of its own, k_post_peel, in round 5):
// every block works *ok out for itself from the launch's counters — gates done, summed over their kAcctShards parts, against n + the
// relays of the hubs — so nothing has to sit between the two launches.
// rflag: the root bits BY RANK, one u64 per wave of 64 ranks (a ballot): what k_root_list and k_euler_next ask instead of reading the
// 16-byte tree entries again.  The depth of the DFS forest (a statistic) is taken here too: one atomic per workgroup.
__global__ void __launch_bounds__(kThreads) k_root_bits(u32 n, const u32* __restrict__ ctl, const u32* __restrict__ edges, const u32* __restrict__ dup,
                                                        const u32* __restrict__ relay_total, u32* post, u32* ok_out, const uint4* __restrict__ meta,
                                                        const u32* __restrict__ orig, u32* rbits, u64* rflag, u32* maxdepth) {
    __shared__ u32 s_ok, s_max[kThreads / 64];
    const u32 lane = threadIdx.x & 63u;
    if (threadIdx.x < 64) {
        const u32 t = threadIdx.x;
        u32 done = t < kAcctShards ? ctl[CTL_PROC + t * kAcctStride] : 0u, lvl = t < kAcctShards ? ctl[CTL_PROC + t * kAcctStride + 1] : 0u;
        u32 rereads = (blockIdx.x == 0 && t < kAcctShards) ? ctl[CTL_PROC + t * kAcctStride + 2] : 0u;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { done += __shfl_xor(done, off, 64); rereads += __shfl_xor(rereads, off, 64); const u32 o = __shfl_xor(lvl, off, 64); lvl = o > lvl ? o : lvl; }
        if (t == 0) {
            // (the relays of the hubs are steps of the launch too: n + *relay_total when nothing is left behind; reported: gates done = steps
            // minus relays — a relay that did not run has a gate above it that did not either)
            const u32 relays = *relay_total;
            const u32 ok = (ctl[CTL_ABORT] == 0u && done == n + relays) ? 1u : 0u;
            s_ok = ok;
            if (blockIdx.x == 0) {
                post[0] = done - relays; post[1] = lvl; post[2] = ctl[CTL_ABORT]; post[3] = rereads; post[4] = *edges; post[5] = *dup; post[6] = ctl[CTL_NEEDDEEP]; post[7] = relays;
                *ok_out = ok;                    // (what the rest of the order stage, queued right behind, goes by)
            }
        }
    }
    __syncthreads();
    if (!s_ok) return;
    u32 md = 0;
    for (u64 rb = gtid() - lane; rb < n; rb += gstride()) {       // (whole waves: the flags by rank are a ballot)
        const u64 r = rb + lane;
        bool root = false;
        if (r < n) {
            const uint4 m = meta[r];
            md = m.y > md ? m.y : md;
            root = m.x == C2A_NONE;
            if (root) { const u32 o = orig[r]; atomicOr(&rbits[o >> 5], 1u << (o & 31u)); }
        }
        const u64 bal = __ballot(root);
        if (lane == 0) rflag[rb >> 6] = bal;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { const u32 o = __shfl_xor(md, off, 64); md = o > md ? o : md; }
    if (lane == 0) s_max[threadIdx.x >> 6] = md;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 m = 0;
        for (int w = 0; w < kThreads / 64; ++w) m = s_max[w] > m ? s_max[w] : m;
        // (a look first: the maximum only grows, so a workgroup that sees a value >= its own has nothing to add — atomics on
        // one word go one at a time, and there are thousands of workgroups)
        if (m && m > __hip_atomic_load(maxdepth, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(maxdepth, m);
    }
}
struct ScanPopc { const u32* w; __device__ __forceinline__ void operator()(u64 i, u32* x) const { x[0] = (u32)__popc(w[i]); } };
__global__ void k_root_list(u32 n, const u32* __restrict__ ok, const u64* __restrict__ rflag, const u32* __restrict__ orig, const u32* __restrict__ rbits,
                            const u32* __restrict__ rpre, u32* ridx, u32* rlist) {
    if (!*ok) return;
    for (u64 r = gtid(); r < n; r += gstride()) {
        if (!((rflag[r >> 6] >> (r & 63u)) & 1ull)) continue;
        const u32 o = orig[r];
        const u32 k = rpre[o >> 5] + (u32)__popc(rbits[o >> 5] & ((1u << (o & 31u)) - 1u));
        ridx[r] = k;
        rlist[k] = (u32)r;
    }
}

// element 2x = enter(x), 2x+1 = exit(x); the tour visits label-0 child, label-1 child, then exits.
// child[2p + l] was written by the peel when the child picked (p, l) as its parent (NONE otherwise).
// A node writes its own enter — and the EXITS OF ITS CHILDREN: behind the label-0 child comes the label-1 child if there is one, else
// the node's own exit; behind the label-1 child the node's exit.  (Round 5 had every node look its sibling up at its parent's entry:
// one scattered READ per node, a 128-byte line each — profiles/r06_pmc_calibration.txt —, 1.3 GB for 10 M gates; a scattered 4-byte
// STORE is a 32-byte request.)  A root's exit leads to the next root (ascending original id: ridx / rlist); a root is told by the
// root bits by rank (k_root_bits) — the tree entries themselves are not read here.
// (tree_child: the relays of a hub — c2a_peel.h HUBS AND RELAYS — are skipped where the tree is read: what hangs below (p, l) is the
// hub at the top of the relay chain that chose that edge)
__global__ void __launch_bounds__(kThreads) k_euler_next(u32 n, const u32* __restrict__ ok, const u64* __restrict__ rflag,
                             const u32* __restrict__ child, const u32* __restrict__ ridx, const u32* __restrict__ rlist,
                             const u32* __restrict__ n_roots_p, u32* next) {
    if (!*ok) return;
    const u32 n_roots = *n_roots_p;
    for (u64 i = gtid(); i < n; i += gstride()) {
        const u32 x = (u32)i;
        const uint2 cc = reinterpret_cast<const uint2*>(child)[i];
        const u32 c0 = tree_child(n, child, cc.x), c1 = tree_child(n, child, cc.y);
        next[2 * i] = c0 != C2A_NONE ? 2 * c0 : (c1 != C2A_NONE ? 2 * c1 : 2 * x + 1);
        if (c0 != C2A_NONE) next[2 * (u64)c0 + 1] = c1 != C2A_NONE ? 2 * c1 : 2 * x + 1;
        if (c1 != C2A_NONE) next[2 * (u64)c1 + 1] = 2 * x + 1;
        if ((rflag[i >> 6] >> (i & 63u)) & 1ull) {
            const u32 k = ridx[x];
            next[2 * i + 1] = k + 1 < n_roots ? 2 * rlist[k + 1] : C2A_NONE;
        }
    }
}

#ifndef C2A_SPLIT_SHIFT
#define C2A_SPLIT_SHIFT 26      // one splitter per 2^(32 - shift) tour elements
#endif
__device__ __forceinline__ bool is_splitter(u32 e, u32 head) { return e == head || ((e * 0x9E3779B1u) >> C2A_SPLIT_SHIFT) == 0u; }

// splitter compaction, 8 elements per lane.  Which elements are splitters is a hash of the element number — nothing is read —
// so a workgroup first COUNTS the splitters of its whole share (a run of whole 2 048-element tiles), reserves their places
// with ONE atomic, and then hands the places out tile by tile.  (One reservation per tile was 9 766 returning atomics on one
// word for the 10 M-gate graph, and those go one at a time, ~10 ns each: 100 of the kernel's 118 us.)
__global__ void __launch_bounds__(kThreads) k_rank_mark(u32 m, const u32* __restrict__ ok, const u32* __restrict__ rlist, u32* scount, u32* slist,
                                                        u32* owner) {
    __shared__ u32 s_w[kThreads / 64];
    __shared__ u32 s_base;
    if (!*ok) return;                            // (no splitter is counted: the walk and the jumps behind this find nothing to do)
    const u32 head = 2 * rlist[0];
    const u32 tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    constexpr u64 T = (u64)kThreads * 8;
    const u64 tiles = ((u64)m + T - 1) / T, per = (tiles + gridDim.x - 1) / gridDim.x;
    const u64 t0 = (u64)blockIdx.x * per, t1 = t0 + per < tiles ? t0 + per : tiles;
    u32 mine = 0;
    for (u64 t = t0; t < t1; ++t) {
        const u64 e0 = t * T + (u64)tid * 8;
#pragma unroll
        for (u32 k = 0; k < 8; ++k) mine += (e0 + k < m && is_splitter((u32)(e0 + k), head)) ? 1u : 0u;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off, 64);
    if (lane == 0) s_w[wv] = mine;
    __syncthreads();
    if (tid == 0) {
        u32 run = 0;
        for (int w = 0; w < kThreads / 64; ++w) run += s_w[w];
        s_base = run ? atomicAdd(scount, run) : 0u;
    }
    __syncthreads();
    u32 place = s_base;                          // where this workgroup's next splitter goes
    for (u64 t = t0; t < t1; ++t) {
        const u64 e0 = t * T + (u64)tid * 8;
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
        __syncthreads();                         // (s_w of the tile before has been read by everybody)
        if (lane == 63) s_w[wv] = inc;
        __syncthreads();
        u32 before = 0, total = 0;
#pragma unroll
        for (u32 w = 0; w < kThreads / 64; ++w) { const u32 v = s_w[w]; before += w < wv ? v : 0u; total += v; }
        u32 pos = place + before + inc - cnt;
#pragma unroll
        for (u32 k = 0; k < 8; ++k)
            if (bits & (1u << k)) { slist[pos] = (u32)(e0 + k); owner[e0 + k] = pos; ++pos; }
        place += total;
    }
}

// one thread per splitter walks its sublist.  Only EXIT elements are ranked in the end (sorted.push happens at an exit), so
// only they get a record — ol[x] = sublist << 32 | exits before exit(x) inside the sublist: one random 8-byte write per node
// instead of four 4-byte ones per node (owner[] keeps the index of the splitter elements only, written by k_rank_mark)
__global__ void k_rank_walk(const u32* __restrict__ scount, const u32* __restrict__ rlist, const u32* __restrict__ slist,
                            const u32* __restrict__ next, const u32* __restrict__ owner, u64* ol, uint2* sjump) {
    const u32 S = *scount;
    const u32 head = 2 * rlist[0];
    for (u64 k = gtid(); k < S; k += gstride()) {
        u32 e = slist[k], acc = 0, nx;
        for (;;) {
            if (e & 1u) { ol[e >> 1] = ((u64)k << 32) | acc; ++acc; }
            const u32 e2 = next[e];
            if (e2 == C2A_NONE) { nx = C2A_NONE; break; }
            if (is_splitter(e2, head)) { nx = owner[e2]; break; }
            e = e2;
        }
        sjump[k] = make_uint2(nx, acc);                           // {next sublist, exits in this one}: what k_rank_jump sums up
    }
}

// pointer jumping over the splitter list, kJumpSpan elements per launch: an element takes over the sums of its next
// kJumpSpan - 1 successors and points behind them, so the reach grows by that factor per launch and after
// ceil(log_kJumpSpan S) launches val[k] = sum over k..end.  (Doubling — span 2 — was 19 launches of 7.4 µs for 312 000
// splitters, most of it launch overhead: the arrays sit in L2.)
constexpr u32 kJumpSpan = 8;
__global__ void k_rank_jump(const u32* __restrict__ scount, const uint2* __restrict__ in, uint2* out) {
    const u32 S = *scount;
    for (u64 k = gtid(); k < S; k += gstride()) {
        uint2 a = in[k];                                          // {next, sum}: ONE gather per hop
        for (u32 h = 1; h < kJumpSpan && a.x != C2A_NONE; ++h) { const uint2 b = in[a.x]; a = make_uint2(b.x, a.y + b.y); }
        out[k] = a;
    }
}

// sorted[post-order index of x] = gate(x)   (== sorted.push(i), topological_sort.rs:46): the rank for the numbering kernels
// behind this one, the original gate id for the caller
__global__ void k_rank_final(u32 n, const u64* __restrict__ ol, const uint2* __restrict__ suffix, const u32* __restrict__ orig,
                             uint2* sorted2) {
    for (u64 x = gtid(); x < n; x += gstride()) {
        const u64 r = ol[x];
        const u32 post = (n - suffix[(u32)(r >> 32)].y) + (u32)r;
        sorted2[post] = make_uint2((u32)x, orig[x]);      // (ONE scattered 8-byte store; k_sorted_split makes the two arrays of it, streaming)
    }
}
__global__ void k_sorted_split(u32 n, const uint2* __restrict__ sorted2, u32* sorted_r, u32* sorted) {
    for (u64 i = gtid(); i < n; i += gstride()) { const uint2 v = sorted2[i]; sorted_r[i] = v.x; sorted[i] = v.y; }
}

// ------------------------------------------------------------------------------------------------
// serial DFS on one lane: literal topological_sort.rs with an explicit stack, in ORIGINAL gate ids (the deps of a gate are
// looked up as the reference does, compiler.rs:408-421: producer of lh, producer of rh — prod1 holds ranks, orig[] leads back).
// Produces the reference's cycle diagnostic ("detected at i={}", topological_sort.rs:34-38) once the parallel peel has found
// leftover gates; it is also the sort that cannot fail (the fall-back when the dataflow launch gives up) and an in-library
// cross-check (c2a_topo_sort_serial).
// state[g]: bit0 visiting, bit1 visited, bits 2..3 next dep index.  result = {status, cycle_at, count}
// ------------------------------------------------------------------------------------------------
__global__ void k_serial_dfs(u32 n, const u32* __restrict__ lh, const u32* __restrict__ rh, const u32* __restrict__ prod1,
                             const u32* __restrict__ orig, u8* state, u32* stack, u32* sorted, u32* result) {
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
                const u32 p = prod1[k == 0 ? lh[i] : rh[i]];
                if (p == 0u) continue;
                const u32 j = orig[p - 1u];
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
// what the kernels behind a serial sort want besides sorted[]: the order in rank space ...
__global__ void k_invert(u32 n, const u32* __restrict__ orig, u32* inv) {
    for (u64 r = gtid(); r < n; r += gstride()) inv[orig[r]] = (u32)r;
}
__global__ void k_sorted_to_rank(u32 n, const u32* __restrict__ sorted, const u32* __restrict__ inv, u32* sorted_r) {
    for (u64 i = gtid(); i < n; i += gstride()) sorted_r[i] = inv[sorted[i]];
}
// ... and — for the level-parallel evaluators — the reverse Kahn level of every gate, which the dataflow launch would have
// left in meta[]: consumers come after their producers in sorted[], so one walk from the back sees every consumer of a gate
// before the gate.  One lane; this only ever runs behind a launch that gave up twice.
__global__ void k_serial_levels(u32 n, const u32* __restrict__ sorted_r, const u32* __restrict__ dep0, const u32* __restrict__ dep1,
                                u32* lvl, uint4* meta, u32* max_level) {
    if (gtid() != 0) return;
    u32 mx = 0;
    for (u32 i = n; i-- > 0;) {
        const u32 r = sorted_r[i], L = lvl[r];
        meta[r] = make_uint4(C2A_NONE, 0u, 0u, L << 1);
        mx = L > mx ? L : mx;
        const u32 d0 = dep0[r], d1 = dep1[r];
        if (d0 != C2A_NONE && lvl[d0] < L + 1u) lvl[d0] = L + 1u;
        if (d1 != C2A_NONE && lvl[d1] < L + 1u) lvl[d1] = L + 1u;
    }
    *max_level = mx;
}

// ------------------------------------------------------------------------------------------------
// wire numbering (compiler.rs:388-449) and gate emission (compiler.rs:451-464)
// node_wire1[node] = wire id + 1 (0 = none); nflag bit0 = input node, bit1 = output node (readers mask with 3).
// ------------------------------------------------------------------------------------------------
// The IO flags of the nodes are set up in front of the sort (do_prep: nflag cleared, inputs marked, then outputs): k_deps
// folds "the out node is an IO node" into the gate's payload record, where the out nodes lie in rank order, and the numbering
// behind the sort no longer pays a scattered read per gate for it.
__global__ void k_mark_inputs(u32 n_in, const u32* __restrict__ in_nodes, u8* nflag) {
    for (u64 i = gtid(); i < n_in; i += gstride()) nflag[in_nodes[i]] = 1;      // (all writers store the same byte)
}
// the per-node state of the wire numbering in one launch: no wire, not seen yet; then the input wires (compiler.rs:388-395)
__global__ void k_node_init(u32 n_nodes, u32* node_wire1, u32* first) {
    for (u64 v = gtid(); v < n_nodes; v += gstride()) { node_wire1[v] = 0u; first[v] = 0xFFFFFFFFu; }
}
__global__ void k_input_wires(u32 n_in, const u32* __restrict__ in_nodes, u32* node_wire1) {
    for (u64 i = gtid(); i < n_in; i += gstride()) atomicMax(&node_wire1[in_nodes[i]], (u32)i + 1);   // duplicate node: the later insert wins (:392-395)
}

// outputs are marked in a second launch so that "input and output" is seen whatever the order
__global__ void k_mark_outputs(u32 n_out, const u32* __restrict__ out_nodes, u8* nflag, u32* err) {
    for (u64 i = gtid(); i < n_out; i += gstride()) {
        const u32 node = out_nodes[i];
        if (nflag[node] & 1u) atomicOr(err, 1u);    // compiler.rs:363-383
        else nflag[node] = 2;
    }
}

// One lane per sorted position handles its three walk entries [lh, rh, out] (compiler.rs:427-430): walk index 3*pos+k.
// first[node] = first index in the walk
// first[node] = index of the node's first appearance in the walk `for gate in sorted: [lh, rh, out]` (compiler.rs:427-430).
// Every node has one writer (*dup == 0, the normal case): the sorted order is topological, so a produced node is first
// seen as its producer's `out` — nothing to compute — and only references to un-produced nodes (inputs, constants) need
// the atomicMin.  Otherwise: atomicMin over all 3n references.
// first[node] = min(first[node], i) — with a LOOK first: a named constant that a whole template context reads (the reference's
// unroller makes one node per literal and context, process.rs:558-579: `0`, `1`) is ONE word for all its readers, atomics on one
// word go one at a time (~11 ns), and nearly all of them would lose anyway.  A constant read by 10^6 of 10 M gates: the numbering
// stage 12.5 ms without the look, 0.1 with it.  (A stale look only costs an atomic that loses: the atomic stays the arbiter.)
__device__ __forceinline__ void first_min(u32* first, u32 node, u32 i) {
    if (__hip_atomic_load(&first[node], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > i) atomicMin(&first[node], i);
}
__global__ void k_first_seen(u32 n, const u32* __restrict__ sorted, const uint4* __restrict__ gate4, const u32* __restrict__ prod1,
                             const u32* __restrict__ dup, u32* first) {
    const bool general = *dup != 0;
    for (u64 pos = gtid(); pos < n; pos += gstride()) {
        const uint4 g = gate4[sorted[pos]];
        const u32 i = 3u * (u32)pos;
        if (general) {
            first_min(first, g.x, i);
            first_min(first, g.y, i + 1);
            first_min(first, g.z, i + 2);
        } else {
            if (prod1[g.x] == 0) first_min(first, g.x, i);
            if (prod1[g.y] == 0) first_min(first, g.y, i + 1);
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
            flag[i] = (first[g.x] == i && (nflag[g.x] & 3u) == 0) ? 1u : 0u;                 // :431-438
            flag[i + 1] = (first[g.y] == i + 1 && (nflag[g.y] & 3u) == 0) ? 1u : 0u;
            flag[i + 2] = (first[g.z] == i + 2 && (nflag[g.z] & 3u) == 0) ? 1u : 0u;
        } else {
            flag[i] = (prod1[g.x] == 0 && first[g.x] == i && (nflag[g.x] & 3u) == 0) ? 1u : 0u;
            flag[i + 1] = (prod1[g.y] == 0 && first[g.y] == i + 1 && (nflag[g.y] & 3u) == 0) ? 1u : 0u;
            flag[i + 2] = (nflag[g.z] & 3u) == 0 ? 1u : 0u;                                  // first seen here: its only writer
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

// (the wires handed out in the walk = add + *n_mid: the walk's own count, or n + the net shift of the positional numbering)
__global__ void k_assign_outputs(u32 n_out, const u32* __restrict__ out_nodes, u32 n_in, u32 add, const u32* __restrict__ n_mid,
                                 u32* node_wire1) {
    const u32 base = n_in + add + *n_mid;
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

// ------------------------------------------------------------------------------------------------
// POSITIONAL NUMBERING — wires and emission of a circuit whose nodes have one writer each (*dup == 0, what the reference's
// front-end builds), without a walk in sorted order.
// The walk `for gate in sorted: [lh, rh, out]` (compiler.rs:427-430) hands a node a new wire where it is first seen unless it
// is an IO node (:431-438).  The sorted order is topological, so a PRODUCED node is first seen as its producer's out: the gate
// at sorted position q gets wire n_in + q — but for two kinds of EVENT that shift everything behind them: a gate whose out
// node is an IO node hands out nothing (-1 from walk index 3q + 2 on), and a constant-like node (un-produced, no IO node)
// takes a wire at the walk index it is first seen at (+1 from there on: 3q if the gate at q reads it as lh, 3q + 1 as rh).
// A circuit from the reference's unroller has MANY of both — a named constant node per literal and template context
// (process.rs:558-579: 5-30 % of the gates read one), an output signal per template — so the events are kept as three BITS per
// sorted position (A: the out node is an IO node, B0 / B1: the lh / rh operand is a constant-like node first seen here), in
// 16-byte blocks of 32 positions {A, B0, B1, -} (5 MB for 10 M gates: the look-ups below are served by the L2s), with the net
// shift in front of every block from ONE streaming scan over the blocks' population counts:
//     D(q) = dpre[q / 32] + popc(B0 | below q) + popc(B1 | below q) - popc(A | below q)      (the events of positions < q)
//     wire of a constant-like node first seen as lh of the gate at q = n_in + q + D(q), as rh: ... + B0(q)
//     wire of the out node of the gate at q                           = n_in + q + D(q) + B0(q) + B1(q)
// A gate then needs nothing but its own position and its producers' wires — wire_r[] is indexed by rank, producers sit a bounded
// distance before their consumers in rank space, so these gathers stay on chip; the sorted order is only ever written, one
// scattered 16-byte record per gate {in0, in1, rank, original id; the op in their top bits}, which k_emit_split streams into the arrays of the
// ABI.  (The walk it replaces: one 16-byte gather + a scatter + two 4-byte gathers by node id per sorted position, in an
// order that is local in no space.)  Round 4 kept the events as a sorted list of at most 4 096 — which only the synthetic
// headline graph (2 064 events) ever took.
//   k_pos_first     pos_r[rank] = post-order position; first[node] = min walk index over the uses of a constant-like node
//   k_pos_bits      the three bits of every position that has an event (atomicOr; gates without a flag touch nothing)
//   (scan)          dpre[] = net shift in front of every block, epre[] = events in front of it (its total: a statistic)
//   (k_assign_outputs: n_mid = n + net shift of all events)
//   k_pos_rank      wire_r[rank] = wire of the gate's out node; the constant-like nodes' wires, by the gate that sees them first
//   k_emit_rank     in0 / in1 = wire_r[] of the producers (un-produced nodes: by look-up), node -> wire, the 16-byte records
//   k_emit_split    e_in0 / e_in1 / e_out (by formula from the position) / e_op / sorted_r / sorted
// ------------------------------------------------------------------------------------------------
// where the list ranking (ol / suffix as k_rank_final reads them) puts gate x — or, when the sorted order exists already
// (c2a_topo_sort handed it to the caller), its inverse (k_eval_inverse into pos_r[]): FROM_SORTED
struct PosSrc { u32 n; const u64* ol; const uint2* suffix; const u32* pos; };
template <bool FROM_SORTED>
__device__ __forceinline__ u32 pos_of(const PosSrc& S, u64 x) {
    if (FROM_SORTED) return S.pos[x];
    const u64 r = S.ol[x];
    return (S.n - S.suffix[(u32)(r >> 32)].y) + (u32)r;
}
constexpr u32 kFlagOutIO = kGateOutIO >> 8, kFlagLhConst = kGateLhConst >> 8, kFlagRhConst = kGateRhConst >> 8;      // (bits of gflag[])
constexpr u32 kFlagEvent = kFlagOutIO | kFlagLhConst | kFlagRhConst;
template <bool FROM_SORTED>
__global__ void k_pos_first(PosSrc S, const u8* __restrict__ gflag, const uint4* __restrict__ gate4, u32* pos_r, u32* first, u32 n_in,
                            const u32* __restrict__ in_nodes, u32* node_wire1) {
    // (the input wires, compiler.rs:388-395, ride along: a launch of n_in threads of their own otherwise.  Duplicate node: the later insert wins)
    for (u64 i = gtid(); i < n_in; i += gstride()) atomicMax(&node_wire1[in_nodes[i]], (u32)i + 1);
    for (u64 x = gtid(); x < S.n; x += gstride()) {
        const u32 p = pos_of<FROM_SORTED>(S, x);
        if (!FROM_SORTED) pos_r[x] = p;
        const u32 f = gflag[x];
        if (!(f & (kFlagLhConst | kFlagRhConst))) continue;
        const uint4 g = gate4[x];
        if (f & kFlagLhConst) first_min(first, g.x, 3u * p);
        if (f & kFlagRhConst) first_min(first, g.y, 3u * p + 1u);
    }
}
__global__ void k_pos_bits(u32 n, const u8* __restrict__ gflag, const uint4* __restrict__ gate4, const u32* __restrict__ pos_r,
                           const u32* __restrict__ first, u32* blk) {
    for (u64 x = gtid(); x < n; x += gstride()) {
        const u32 f = gflag[x];
        if (!(f & kFlagEvent)) continue;
        const u32 p = pos_r[x], bit = 1u << (p & 31u);
        u32* w = blk + 4 * (u64)(p >> 5);
        if (f & kFlagOutIO) atomicOr(w, bit);
        if (!(f & (kFlagLhConst | kFlagRhConst))) continue;
        const uint4 g = gate4[x];
        if ((f & kFlagLhConst) && first[g.x] == 3u * p) atomicOr(w + 1, bit);
        if ((f & kFlagRhConst) && first[g.y] == 3u * p + 1u) atomicOr(w + 2, bit);      // (lh == rh == the same constant: seen as lh)
    }
}
// per block: {net shift, events} (the net shift may be negative: two's complement in 32 bits — the scan's sums are exact mod 2^32)
struct ScanPosBits {
    const uint4* blk;
    __device__ __forceinline__ void operator()(u64 i, u32* x) const {
        const uint4 w = blk[i];
        const u32 a = (u32)__popc(w.x), b = (u32)__popc(w.y) + (u32)__popc(w.z);
        x[0] = b - a; x[1] = a + b;
    }
};
struct PosBits { const uint4* blk; const u32* dpre; };
// n_in-less wire of what position p hands out first: p + D(p); b0 / b1 = the position's own B bits
__device__ __forceinline__ u32 pos_base(const PosBits& T, u32 p, u32* b0, u32* b1) {
    const uint4 w = T.blk[p >> 5];
    const u32 bit = 1u << (p & 31u), lt = bit - 1u;
    *b0 = (w.y & bit) ? 1u : 0u;
    *b1 = (w.z & bit) ? 1u : 0u;
    return p + T.dpre[p >> 5] + (u32)__popc(w.y & lt) + (u32)__popc(w.z & lt) - (u32)__popc(w.x & lt);
}
// the wire of every gate's out node (the formula; an IO node's wire is the node's), in rank order — and the wires of the
// constant-like nodes, written by the gate that sees them first
__global__ void k_pos_rank(u32 n, u32 n_in, const u8* __restrict__ gflag, const uint4* __restrict__ gate4, const u32* __restrict__ pos_r, PosBits T,
                           u32* node_wire1, u32* wire_r) {
    for (u64 x = gtid(); x < n; x += gstride()) {
        const u32 f = gflag[x];
        u32 b0, b1;
        const u32 base = n_in + pos_base(T, pos_r[x], &b0, &b1);
        if (b0 | b1) {
            const uint4 g = gate4[x];
            if (b0) node_wire1[g.x] = base + 1u;
            if (b1) node_wire1[g.y] = base + b0 + 1u;
        }
        wire_r[x] = (f & kFlagOutIO) ? node_wire1[gate4[x].z] - 1u : base + b0 + b1;
    }
}
// The record a gate leaves at its sorted position: {in0, in1, rank, original id} — 16 bytes, ONE scattered store per gate
// (a 32-byte record is two store instructions, i.e. twice the scattered transactions).
// Gate ids are below 2^29 (include/c2a.h), so the op code (5 bits) rides in the top bits of the two ids and bit 31 of the
// second says "the out node is an IO node"; every other gate's out wire follows from its position (k_emit_split).
// No look-ups of the event bits here: a gate's operands are its producers' out wires, wire_r[] of a rank a bounded distance back.
typedef uint4 EmitRec;
__global__ void k_emit_rank(u32 n, const uint4* __restrict__ gate4, const u32* __restrict__ dep0, const u32* __restrict__ dep1,
                            const u32* __restrict__ orig, const u32* __restrict__ pos_r, const u32* __restrict__ wire_r, u32* node_wire1, EmitRec* erec) {
    const XcdSweep R = xcd_sweep(n);
    for (u64 x = R.i; x < R.end; x += R.step) {
        const uint4 g = gate4[x];
        const u32 d0 = dep0[x], d1 = dep1[x];                                       // (dep1 is dropped when it equals dep0: k_deps)
        // an un-produced node: an input, a constant-like node (k_pos_rank), an output nobody produces
        const u32 in0 = (g.w & kGateLhUnprod) ? node_wire1[g.x] - 1u : wire_r[d0];
        const u32 in1 = (g.w & kGateRhUnprod) ? node_wire1[g.y] - 1u : wire_r[d1 != C2A_NONE ? d1 : d0];
        const bool io = (g.w & kGateOutIO) != 0u;
        if (!io) node_wire1[g.z] = wire_r[x] + 1u;                                  // node -> wire of the out node (near-streaming: out nodes ascend with the rank)
        const u32 op = g.w & 0x1Fu;
        erec[pos_r[x]] = make_uint4(in0, in1, (u32)x | (op & 7u) << 29, orig[x] | (op >> 3) << 29 | (io ? 0x80000000u : 0u));
    }
}
template <bool WITH_SORTED>
__global__ void k_emit_split(u32 n, u32 n_in, const EmitRec* __restrict__ erec, const uint4* __restrict__ gate4, const u32* __restrict__ node_wire1, PosBits T,
                             u32* e_in0, u32* e_in1, u32* e_out, u8* e_op, u32* sorted_r, u32* sorted) {
    // (one position per lane: with four per lane — 16-byte stores into every stream — the record loads are 64 bytes apart
    // between neighbouring lanes: 89 instead of 67 µs)
    for (u64 i = gtid(); i < n; i += gstride()) {
        const EmitRec e = erec[i];
        const u32 rank = e.z & 0x1FFFFFFFu;
        e_in0[i] = e.x; e_in1[i] = e.y;
        u32 b0, b1;
        const u32 base = n_in + pos_base(T, (u32)i, &b0, &b1);
        e_out[i] = (e.w & 0x80000000u) ? node_wire1[gate4[rank].z] - 1u : base + b0 + b1;
        e_op[i] = (u8)((e.z >> 29) | ((e.w >> 29) & 3u) << 3);
        if (WITH_SORTED) { sorted_r[i] = rank; sorted[i] = e.w & 0x1FFFFFFFu; }
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

// T(op, w) and AUX(op, w) of the gate at sorted position p, for the scan that places its boolean gates and aux wires
struct ScanBoolSizes {
    const u8* e_op; const BoolTables* T;
    __device__ __forceinline__ void operator()(u64 p, u32* x) const { const u32 o = e_op[p]; x[0] = T->tsize[o]; x[1] = T->taux[o]; }
};

// cut[k] = first sorted position p with goff[p] >= G k / N (k = 0..N; cut[N] = n), qcut[k] = goff[cut[k]]: N ranges of
// sorted positions holding (nearly) equal numbers of boolean gates
__global__ void k_shard_cuts(u32 n, u32 N, const u64* __restrict__ goff, u32* cut, u64* qcut) {
    for (u32 k = threadIdx.x; k <= N; k += blockDim.x) {
        const u64 G = goff[n];
        const u64 target = G / N * k + G % N * k / N;            // floor(G k / N) without overflow
        u32 lo = 0, hi = n;
        while (lo < hi) {
            const u32 mid = lo + (hi - lo) / 2;
            if (goff[mid] < target) lo = mid + 1; else hi = mid;
        }
        if (k == N) lo = n;
        cut[k] = lo;
        qcut[k] = goff[lo];
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
    u32 slices;       // workgroups per CHUNK arithmetic gates (gridDim.y): each takes a share of the chunk's boolean gates
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
// SLICES (round 6): a chunk of 256 multipliers is 723 000 boolean gates, of 256 dividers 2.1 M — with one workgroup per chunk a circuit of a
// few hundred such gates keeps two or three CUs busy (the Poseidon-shaped config: 0.20 ms for 559 000 boolean gates), and one of 10^5
// dividers 391 workgroups on 256 CUs (two rounds for 1.5 rounds' work: 4.0 TB/s against 5.2 for the small templates).  When the
// chunks are fewer than four per CU the host asks for several workgroups per chunk (gridDim.y): each repeats the chunk's prologue
// and takes a share of the aligned body, in whole groups of 16 gates; the first also does the unaligned ends.
// Measured alternatives that LOST in interleaved same-session A/B runs (kept out of the tree): a start-bit map
// + popcount instead of the search, a "four gates, one owner" fast path, software-pipelined template loads,
// persistent workgroups with LDS-staged packed templates, non-temporal stores (profiles/r01_boolify_ab.txt).
template <int CHUNK, int THREADS = kThreads>
__global__ void __launch_bounds__(THREADS) k_boolify(BoolArgs A, const BoolTables* __restrict__ T) {
    __shared__ u32 s_goff[CHUNK + 1];       // first boolean gate of each arithmetic gate, relative to the block's first
    __shared__ uint4 s_base[CHUNK];         // wire bases per ref kind
    __shared__ u32 s_top[CHUNK];            // template offset
    const u32 tid = threadIdx.x;
    const u64 p0 = (u64)A.p_first + (u64)blockIdx.x * CHUNK;
    const u32 cnt = (u32)((A.p_end - p0) < (u64)CHUNK ? (A.p_end - p0) : (u64)CHUNK);
    const u64 q0 = A.goff[p0];
    for (u32 i = tid; i <= cnt; i += THREADS) s_goff[i] = (u32)(A.goff[p0 + i] - q0);
    for (u32 i = tid; i < cnt; i += THREADS) {
        const u32 wa = A.e_in0[p0 + i], wb = A.e_in1[p0 + i], wo = A.e_out[p0 + i];
        s_base[i] = make_uint4(bool_wire(wa, 0, A.width, A.M, A.out_base), bool_wire(wb, 0, A.width, A.M, A.out_base),
                               bool_wire(wo, 0, A.width, A.M, A.out_base), (u32)(A.aux_base + A.aoff[p0 + i]));
        s_top[i] = T->toff[A.e_op[p0 + i]];
    }
    __syncthreads();
    const u32 total = s_goff[cnt];                       // boolean gates of this block (< 2^32 by construction)
    // the body starts at a boolean gate whose index is a multiple of 16 and covers whole groups of 16: a lane produces 4
    // consecutive gates, the four lanes of a quad 16 — the u32 streams leave as one 16-byte store per lane, the op bytes as one
    // 16-byte store per QUAD (lane 0 of it collects the three other words by DPP: as a 4-byte store per lane the op stream cost
    // 12 % of the kernel for 8 % of its bytes — 4-byte-per-lane stores are issue-bound on gfx950)
    const u32 head = (u32)((16 - (q0 & 15)) & 15);       // gates before the first 16-gate boundary
    const u32 r0 = head < total ? head : total;
    const u32 r1 = r0 + ((total - r0) & ~15u);
    const u32* base_words = reinterpret_cast<const u32*>(s_base);
    typedef u32 u32x4 __attribute__((vector_size(16)));
    const uint4* __restrict__ tmpl = A.tmpl;
    // ---- aligned body: groups of 4 (r = block-relative index).  r1 - r0 is a multiple of 16: the four lanes of a quad are in
    // here together (the emulation wants the whole wave at the exchange: the loop runs per wave, lanes past the end idle)
    // (this workgroup's share of the body: groups of 16 gates [nb y / S, nb (y + 1) / S) — all of it when the launch has one slice)
    u32 ra = r0, rb = r1;
    if (A.slices != 1u) {
        const u32 nb = (r1 - r0) >> 4;
        ra = r0 + 16u * (u32)((u64)nb * blockIdx.y / A.slices); rb = r0 + 16u * (u32)((u64)nb * (blockIdx.y + 1u) / A.slices);
    }
    for (u32 rw = ra + 4u * (tid & ~63u); rw < rb; rw += 4u * THREADS) {
        const u32 r = rw + 4u * (tid & 63u);
        const bool act = r < rb;
        u32 vop = 0;
        const u64 q = q0 + r - A.q_bias;
        if (act) {
            u32 idx = bool_owner(r, s_goff, cnt);
            u32 start = s_goff[idx], bound = s_goff[idx + 1], top = s_top[idx];
            u32 v0[4], v1[4], v2[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                while (r + j >= bound) { ++idx; start = bound; bound = s_goff[idx + 1]; top = s_top[idx]; }
                const uint4 e = tmpl[top + (r + j - start)];
                v0[j] = base_words[idx * 4 + (e.x >> 30)] + (e.x & 0x3FFFFFFFu);
                v1[j] = base_words[idx * 4 + (e.y >> 30)] + (e.y & 0x3FFFFFFFu);
                v2[j] = base_words[idx * 4 + (e.z >> 30)] + (e.z & 0x3FFFFFFFu);
                vop |= (e.w & 0xFFu) << (8 * j);
            }
            *reinterpret_cast<u32x4*>(A.b_in0 + q) = u32x4{v0[0], v0[1], v0[2], v0[3]};
            *reinterpret_cast<u32x4*>(A.b_in1 + q) = u32x4{v1[0], v1[1], v1[2], v1[3]};
            *reinterpret_cast<u32x4*>(A.b_out + q) = u32x4{v2[0], v2[1], v2[2], v2[3]};
        }
        const u32 o1 = quad_bcast<1>(vop), o2 = quad_bcast<2>(vop), o3 = quad_bcast<3>(vop);
        if (act && (tid & 3u) == 0u) *reinterpret_cast<u32x4*>(A.b_op + q) = u32x4{vop, o1, o2, o3};
    }
    // ---- head [0,r0) and tail [r1,total): at most 15 + 15 gates
    if (blockIdx.y == 0) {
        const u32 nh = r0, nt = total - r1;
        if (tid < nh + nt) {
            const u32 r = tid < nh ? tid : r1 + (tid - nh);
            const u32 lo = bool_owner(r, s_goff, cnt);
            const uint4 e = tmpl[s_top[lo] + (r - s_goff[lo])];
            const u64 q = q0 + r - A.q_bias;
            A.b_in0[q] = base_words[lo * 4 + (e.x >> 30)] + (e.x & 0x3FFFFFFFu);
            A.b_in1[q] = base_words[lo * 4 + (e.y >> 30)] + (e.y & 0x3FFFFFFFu);
            A.b_out[q] = base_words[lo * 4 + (e.z >> 30)] + (e.z & 0x3FFFFFFFu);
            A.b_op[q] = (u8)e.w;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// circuit.txt on the GPU (BristolCircuit::write_bristol, src/main.rs:34-35; crate absent, format = SURVEY C.2):
// one line per gate, "2 1 <in0> <in1> <out> <OP>\n" or "1 1 <in0> <out> <OP>\n" for a one-input op.  Two passes:
// line lengths -> exclusive scan (byte offsets) -> every lane prints its own line.  The text of 742 M boolean gates
// is ~27 GB: it is produced chunk by chunk and streamed to the host (c2a_format_bristol).
// ------------------------------------------------------------------------------------------------
struct FmtTable {
    u8 len[32];          // op name length
    u8 unary[32];        // 1 = printed with one input
    char name[32][8];    // op names, not terminated
};
__device__ __forceinline__ u32 dec_digits(u32 v) {
    return v < 10u ? 1u : v < 100u ? 2u : v < 1000u ? 3u : v < 10000u ? 4u : v < 100000u ? 5u : v < 1000000u ? 6u : v < 10000000u ? 7u
         : v < 100000000u ? 8u : v < 1000000000u ? 9u : 10u;
}
__device__ __forceinline__ u32 fmt_line_len(u32 a, u32 b, u32 o, u32 op, const FmtTable* T) {
    const u32 un = T->unary[op];
    return 4u + dec_digits(a) + 1u + (un ? 0u : dec_digits(b) + 1u) + dec_digits(o) + 1u + T->len[op] + 1u;
}
__global__ void k_fmt_len(u64 n, const u32* __restrict__ in0, const u32* __restrict__ in1, const u32* __restrict__ out,
                          const u8* __restrict__ op, const FmtTable* __restrict__ T, u32* len) {
    for (u64 k = gtid(); k < n; k += gstride()) len[k] = fmt_line_len(in0[k], in1[k], out[k], op[k], T);
}
__device__ __forceinline__ char* put_dec(char* p, u32 v) {
    const u32 d = dec_digits(v);
    for (u32 i = d; i-- > 0;) { p[i] = (char)('0' + v % 10u); v /= 10u; }
    return p + d;
}
__global__ void k_fmt_write(u64 n, const u32* __restrict__ in0, const u32* __restrict__ in1, const u32* __restrict__ out,
                            const u8* __restrict__ op, const FmtTable* __restrict__ T, const u64* __restrict__ off, char* text) {
    for (u64 k = gtid(); k < n; k += gstride()) {
        const u32 o = op[k], un = T->unary[o];
        char* p = text + off[k];
        *p++ = un ? '1' : '2'; *p++ = ' '; *p++ = '1'; *p++ = ' ';
        p = put_dec(p, in0[k]); *p++ = ' ';
        if (!un) { p = put_dec(p, in1[k]); *p++ = ' '; }
        p = put_dec(p, out[k]); *p++ = ' ';
        const u32 L = T->len[o];
        for (u32 i = 0; i < L; ++i) *p++ = T->name[o][i];
        *p = '\n';
    }
}

// ---- what the host wants to know at the end of a stage, gathered into ONE place it can read (host memory mapped into the
// device's address space: after the stream has drained the words are simply there).  Every small device-to-host copy is a
// command of its own, ~20 us behind the one before; a step had ten of them.
__global__ void k_post_words(u32* dst, const u32* a, u32 na, const u32* b, u32 nb, const u32* c3, u32 nc) {
    for (u32 i = threadIdx.x; i < na; i += blockDim.x) dst[i] = a[i];
    for (u32 i = threadIdx.x; i < nb; i += blockDim.x) dst[na + i] = b[i];
    for (u32 i = threadIdx.x; i < nc; i += blockDim.x) dst[na + nb + i] = c3[i];
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
// ONE ticket per lane on cnt[key] — the lanes of a wave that name the SAME key take one range together: the gates of a level of a wide
// and shallow circuit (10 M gates in ten levels) are a million tickets on one word, and atomics on one word go one at a time (~11 ns).
// The lanes' keys are matched bit by bit (a ballot per bit, stopping as soon as every lane is alone); every group's first lane draws
// for its group, all of them in the same instruction.  All lanes of the wave call this together (valid: this lane wants a ticket).
// (Off the metric's path — the level lists of the evaluator, the verifier and the prune pass — where 300 instructions per wave and round
// do not matter; k_deps finds its hot producers another way: c2a_kernels.h HOT PRODUCERS.)
__device__ __forceinline__ u32 wave_ticket(u32* cnt, u32 key, bool valid, u32 lane, u32 nbits) {
    u64 peers = __ballot(valid);
    for (u32 b = 0; b < nbits; ++b) {
        const bool bit = (key >> b) & 1u;
        const u64 m = __ballot(valid && bit);
        peers &= bit ? m : ~m;
        if ((b & 3u) == 3u && __ballot(valid && __popcll(peers) > 1) == 0ull) break;
    }
    const u64 lt_mask = (1ull << lane) - 1ull;
    const u32 leader = valid ? (u32)__builtin_ctzll(peers) : lane;
    u32 base = 0;
    if (valid && leader == lane) base = atomicAdd(&cnt[key], (u32)__popcll(peers));
    base = (u32)__shfl((int)base, (int)leader, 64);
    return base + (u32)__popcll(peers & lt_mask);
}
__global__ void __launch_bounds__(kThreads) k_level_hist(u32 n, const uint4* __restrict__ meta, u32* lcount, u32 nbits) {
    const u32 lane = threadIdx.x & 63u;
    for (u64 ib = gtid() - lane; ib < n; ib += gstride()) {
        const u64 i = ib + lane;
        const bool live = i < n;
        (void)wave_ticket(lcount, live ? meta[i].w >> 1 : 0u, live, lane, nbits);
    }
}
__global__ void __launch_bounds__(kThreads) k_level_scatter(u32 n, const uint4* __restrict__ meta, const u32* __restrict__ lbase,
                                u32* cursor, u32* lorder, u32 nbits) {
    const u32 lane = threadIdx.x & 63u;
    for (u64 ib = gtid() - lane; ib < n; ib += gstride()) {
        const u64 i = ib + lane;
        const bool live = i < n;
        const u32 lv = live ? meta[i].w >> 1 : 0u;
        const u32 slot = wave_ticket(cursor, lv, live, lane, nbits);
        if (live) lorder[lbase[lv] + slot] = (u32)i;
    }
}

// ---- the whole evaluation as ONE launch: a grid-wide barrier between reverse Kahn levels instead of two launches per level
// (10 000 launches for the 10 M-gate graph).  The values of a level are read by other workgroups — other XCDs, whose L2s
// are not coherent with each other — in the next one, so values go through agent-scope accesses (the launch boundary used
// to do that) and the barrier is a fence + a counter every workgroup bumps once per level.
// The boolean circuit as the level-parallel passes see it: on a multi-device context every device keeps the boolean gates of its own
// range of SORTED POSITIONS, an arithmetic gate's template lies in one piece, and the passes — which run on the primary
// device — read a piece where it lies (peer access over xGMI) instead of gathering 9.6 GB on the primary first.  Segment s holds the
// gates of sorted positions [p_hi[s - 1], p_hi[s]); the pointers are biased so that pointer[q] is boolean gate q of the WHOLE circuit.
constexpr u32 kMaxSegs = 64;
struct BoolSegs { u32 n_seg, pad; u32 p_hi[kMaxSegs]; const u32* in0[kMaxSegs]; const u32* in1[kMaxSegs]; const u32* out[kMaxSegs]; const u8* op[kMaxSegs]; };
struct BoolPtrs { const u32* in0; const u32* in1; const u32* out; const u8* op; };
__device__ __forceinline__ BoolPtrs seg_of(const BoolSegs* S, u32 p) {
    u32 s = 0;
    const u32 ns = S->n_seg;
    while (s + 1 < ns && p >= S->p_hi[s]) ++s;
    return BoolPtrs{S->in0[s], S->in1[s], S->out[s], S->op[s]};
}
struct EvalRun {
    u32 levels, width, mode;           // mode bit 0: the arithmetic circuit, bit 1: its boolean image
    const u32* lbase; const u32* order; const u32* spos;
    const u32* e_in0; const u32* e_in1; const u32* e_out; const u8* e_op;
    const u64* goff; const BoolSegs* segs;        // the boolean image (k_eval_run)
    const u32* b_in0; const u32* b_in1; const u32* b_out; const u8* b_op;      // k_eval_pruned: the pruned image (one piece, on the primary)
    u64* aval; u64* bval;
    u32* bar;                          // barrier counter (zeroed before the launch)
};
__device__ __forceinline__ u64 ev_ld(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ev_st(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// ---- a boolean template on ONE WAVE, 64 of its gates at a time (a lane per gate).  The gates of a template are in
// dependency order and read each other's wires: a lane that reads the out wire of an EARLIER lane of its chunk (dep: which) takes
// the value from that lane's register once that lane is done — rounds of "every lane whose producers are done" until the chunk
// is; a ripple-carry chain makes that as many rounds as the chunk is deep, a bitwise template one — and everything else comes
// from memory, all lanes at once.  (One lane walking its template gate by gate — the first version — paid two dependent
// round trips per gate: 560 us per level of the 10 M-gate graph, 2.8 s for a pass over its 742 M boolean gates.)
// Every store this wave has issued is performed (and nothing is moved across by the compiler).  The values of these passes
// travel by agent-scope loads and stores, which no cache holds back; a full agent-scope fence (__threadfence) would also write
// back and invalidate the XCD's L2 — once per wave — and that, not the work, was what a level cost (measured: 230 us per level
// of 2 000 gates with __threadfence between the chunks of a template).
__device__ __forceinline__ void store_fence() {
#ifdef C2A_EMULATE
    __threadfence();
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_s_waitcnt(0);
#endif
}
// between two chunks of a template: what the lanes stored is there for what the lanes load next (under emulation the lanes of
// a wave are fibers that only meet at wave-level intrinsics: they have to meet here)
__device__ __forceinline__ void chunk_fence() {
    store_fence();
#ifdef C2A_EMULATE
    (void)__ballot(1);
#endif
}

// ---- the grid-wide barrier of the level-parallel passes: every workgroup has finished a level before any starts the next.
// bar[0] counts arrivals (it only grows: the k-th barrier is over at k x gridDim.x), bar[32] is raised by a workgroup that
// has waited kBarrierPolls polls (~2 s) — some workgroup of the grid is not resident (CU masking, another process holding CUs:
// the launch sizes the grid by the occupancy query, which is a promise about THIS kernel only) — and then every workgroup
// leaves; the host reports it as an error instead of hanging the stream.  One lane per workgroup makes the arrival a release
// and the departure an acquire at agent scope on top of the agent-scope accesses the values travel by (a fence per workgroup
// and level is ~3 us; per WAVE it was what a level cost, see store_fence).  false: the launch is being given up.
#ifdef C2A_EMULATE
constexpr u32 kBarrierPolls = 1u << 22;
#else
constexpr u32 kBarrierPolls = 1u << 23;
#endif
constexpr u32 kBarAbort = 32;
__device__ __forceinline__ bool grid_barrier(u32* bar, u32& target) {
    __shared__ u32 s_ok;
    store_fence();
    __syncthreads();
    target += gridDim.x;
    if (threadIdx.x == 0) {
        u32 ok = 1;
#ifndef C2A_EMULATE
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
#endif
        atomicAdd(bar, 1u);
        u32 polls = 0;
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if ((++polls & 255u) == 0 && __hip_atomic_load(bar + kBarAbort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
            if (polls > kBarrierPolls) { atomicAdd(bar + kBarAbort, 1u); ok = 0; break; }
            peel_sleep(2);
        }
#ifndef C2A_EMULATE
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        s_ok = ok;
    }
    __syncthreads();
    store_fence();
    return s_ok != 0;
}

struct ChunkDeps { int d0, d1; };
// which earlier lane of the chunk (nv lanes hold gates) drives this lane's inputs; -1: nobody here (out wires are unique)
__device__ __forceinline__ ChunkDeps chunk_deps(u32 in0, u32 in1, u32 out, u32 nv, u32 lane) {
    ChunkDeps d{-1, -1};
    for (u32 i = 0; i + 1 < nv; ++i) {
        const u32 oi = rdlane(out, i);
        if (i < lane) { if (in0 == oi) d.d0 = (int)i; if (in1 == oi) d.d1 = (int)i; }
    }
    return d;
}
// gates [k0, k1) on wire values (64 vectors per wire, one bit each)
struct WireIdentity { __device__ __forceinline__ u32 operator()(u32 w) const { return w; } };
template <class Map = WireIdentity>
__device__ __forceinline__ void eval_template_wave(u64 k0, u64 k1, const u32* __restrict__ g_in0, const u32* __restrict__ g_in1,
                                                   const u32* __restrict__ g_out, const u8* __restrict__ g_op, u64* bval, u32 lane, Map map = Map()) {
    for (u64 base = k0; base < k1; base += 64) {
        const u32 nv = k1 - base < 64 ? (u32)(k1 - base) : 64u;
        const bool valid = lane < nv;
        u32 i0 = 0xFFFFFFFFu, i1 = 0xFFFFFFFFu, o = 0xFFFFFFFEu, op = 2;
        if (valid) { i0 = map(g_in0[base + lane]); i1 = map(g_in1[base + lane]); o = map(g_out[base + lane]); op = g_op[base + lane]; }
        ChunkDeps d = chunk_deps(i0, i1, o, nv, lane);
        if (op == 2u) d.d1 = -1;
        const u64 a_mem = valid && d.d0 < 0 ? ev_ld(&bval[i0]) : 0ull;
        const u64 b_mem = valid && op != 2u && d.d1 < 0 ? ev_ld(&bval[i1]) : 0ull;
        bool done = !valid;
        u64 v = 0;
        for (;;) {
            const u64 dm = __ballot(done);
            if (dm == ~0ull) break;
            const u64 av = __shfl(v, d.d0 < 0 ? (int)lane : d.d0, 64), bv = __shfl(v, d.d1 < 0 ? (int)lane : d.d1, 64);
            if (!done && (d.d0 < 0 || ((dm >> d.d0) & 1ull)) && (d.d1 < 0 || ((dm >> d.d1) & 1ull))) {
                const u64 a = d.d0 < 0 ? a_mem : av, b = d.d1 < 0 ? b_mem : bv;
                v = op == 0u ? (a ^ b) : (op == 1u ? (a & b) : ~a);
                done = true;
            }
        }
        if (valid) ev_st(&bval[o], v);
        if (base + 64 < k1) chunk_fence();       // (the next chunk reads these wires from memory)
    }
}
__global__ void __launch_bounds__(kThreads) k_eval_run(EvalRun R) {
    // (one piece — every single-device context —: its pointers are read once per wave, not once per gate)
    const bool one_seg = !R.segs || R.segs->n_seg == 1u;
    const BoolPtrs B_all = R.segs ? seg_of(R.segs, 0u) : BoolPtrs{nullptr, nullptr, nullptr, nullptr};
    const u64 mk = R.width >= 64 ? ~0ull : ((1ull << R.width) - 1ull);
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = gtid() >> 6, n_waves = gstride() >> 6;
    u32 target = 0;
    for (u32 lv = R.levels; lv-- > 0;) {
        const u32 lo = R.lbase[lv], cnt = R.lbase[lv + 1] - lo;
        if (R.mode & 1u)
            for (u64 i = gtid(); i < (u64)cnt * 64; i += gstride()) {
                const u32 p = R.spos[R.order[lo + (u32)(i >> 6)]], t = (u32)(i & 63);
                ev_st(&R.aval[(u64)R.e_out[p] * 64 + t],
                      eval_arith_op(R.e_op[p], ev_ld(&R.aval[(u64)R.e_in0[p] * 64 + t]), ev_ld(&R.aval[(u64)R.e_in1[p] * 64 + t]), R.width, mk));
            }
        if (R.mode & 2u)
            for (u64 i = wave; i < cnt; i += n_waves) {
                const u32 p = R.spos[R.order[lo + (u32)i]];
                const BoolPtrs B = one_seg ? B_all : seg_of(R.segs, p);
                eval_template_wave(R.goff[p], R.goff[p + 1], B.in0, B.in1, B.out, B.op, R.bval, lane);
            }
        if (lv == 0) break;
        if (!grid_barrier(R.bar, target)) return;
    }
}

// ---- the verifier of a MULTI-DEVICE context: every device checks the boolean gates it holds, where they lie.  The whole-circuit
// simulation above needs every boolean gate in one place (round 3 gathered 9.6 GB on the primary device for it); what a
// bit-blast can get wrong, though, is local to an arithmetic gate — its template, and which wires the template's gates name —
// so: a wave per arithmetic gate of the device's range draws 64 input vectors (a function of the input WIRE, so that a gate
// whose two inputs are one wire sees equal values), bit-slices them onto the boolean wires the gate's inputs MUST occupy
// (DESIGN.md 5.1), runs the gate's boolean gates out of a private scratch (every wire they name must be one of the gate's own:
// an A, B or O bit or one of its aux wires — anything else is a mismatch), and compares the O bits with the arithmetic op.
struct VerifyLocal {
    u32 p_lo, p_hi, p_base;            // arithmetic gates (sorted positions) [p_lo, p_hi); e_* / goff / aoff are indexed by p - p_base
    u32 width, M;
    u64 aux_base, out_base, q_bias, seed;
    const u32* e_in0; const u32* e_in1; const u32* e_out; const u8* e_op; const u64* goff; const u64* aoff;
    const u32* b_in0; const u32* b_in1; const u32* b_out; const u8* b_op;      // boolean gate q at index q - q_bias
    u64* scratch; u32 scratch_words;   // per wave: A bits, B bits, O bits, aux wires, one word nobody reads
    ull* bad;
};
struct LocalWires {
    u32 A, B, O, X, w, n_aux, junk; u32* stray;
    __device__ __forceinline__ u32 operator()(u32 wire) const {
        if (wire - A < w) return wire - A;
        if (wire - B < w) return w + (wire - B);
        if (wire - O < w) return 2u * w + (wire - O);
        if (wire - X < n_aux) return 3u * w + (wire - X);
        *stray = 1u;                                   // (a wire of some other gate: the template is wired wrongly)
        return junk;
    }
};
__device__ __forceinline__ u64 verify_value(u64 seed, u32 W, u32 t, u32 width, u64 mk) {
    u64 v = mix64(seed ^ ((u64)W << 8) ^ t) & mk;
    if (t == 0) v = 0; else if (t == 1) v = mk; else if (t == 2) v = 1 & mk; else if (t == 3) v = (1ull << (width - 1)) & mk;
    return v;
}
__global__ void __launch_bounds__(kThreads) k_verify_local(VerifyLocal V) {
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = gtid() >> 6, n_waves = gstride() >> 6;
    const u32 w = V.width;
    const u64 mk = w >= 64 ? ~0ull : ((1ull << w) - 1ull);
    u64* sc = V.scratch + wave * V.scratch_words;
    ull bad = 0;
    for (u64 p = V.p_lo + wave; p < V.p_hi; p += n_waves) {
        const u64 i = p - V.p_base;
        const u32 W0 = V.e_in0[i], W1 = V.e_in1[i], Wo = V.e_out[i], op = V.e_op[i];
        const u64 a = verify_value(V.seed, W0, lane, w, mk), b = verify_value(V.seed, W1, lane, w, mk);
        // bit-slice: word k of the A bits = bit k of the 64 vectors
        u64 wa = 0, wb = 0;
        for (u32 k = 0; k < w; ++k) {
            const u64 ma = __ballot((a >> k) & 1ull), mb = __ballot((b >> k) & 1ull);
            if (lane == k) { wa = ma; wb = mb; }
        }
        if (lane < w) { ev_st(&sc[lane], wa); ev_st(&sc[w + lane], wb); }
        chunk_fence();
        u32 stray = 0;
        const u64 x0 = V.aoff[i], x1 = V.aoff[i + 1];
        LocalWires L{bool_wire(W0, 0, w, V.M, V.out_base), bool_wire(W1, 0, w, V.M, V.out_base), bool_wire(Wo, 0, w, V.M, V.out_base),
                     (u32)(V.aux_base + x0), w, (u32)(x1 - x0), V.scratch_words - 1u, &stray};
        eval_template_wave(V.goff[i] - V.q_bias, V.goff[i + 1] - V.q_bias, V.b_in0, V.b_in1, V.b_out, V.b_op, sc, lane, L);
        chunk_fence();
        // lane t reads vector t of the O bits back and compares it with the arithmetic result
        u64 got = 0;
        for (u32 k = 0; k < w; ++k) got |= ((ev_ld(&sc[2u * w + k]) >> lane) & 1ull) << k;
        const u64 want = eval_arith_op(op, a, b, w, mk);
        const u64 wrong = __ballot(got != want) | (__ballot(stray != 0u) ? ~0ull : 0ull);
        bad += (ull)__popcll(wrong);
        chunk_fence();                                 // (the scratch is free for the next gate)
    }
    if (lane == 0 && bad) atomicAdd(V.bad, bad);
}

// ------------------------------------------------------------------------------------------------
// optional PRUNE pass over the bit-blasted circuit (what the absent `boolify` crate is believed to do after its per-gate
// blast, SURVEY C.2: constant folding + removal of logic no output reaches).  Not the metric's path: the frozen per-gate map
//  stays what c2a_boolify returns; this rewrites it into a smaller, functionally equal circuit.
//   rep[w]   what boolean wire w is known to be: 0 / 1 = the constant, x + 2 = the same as wire x (itself: a real wire)
//   fold     forward over the reverse Kahn levels (producers first), a lane per arithmetic gate walks its template in order:
//            XOR(a,a) = 0, XOR(a,0) = a, XOR(a,1) = INV a, AND(a,a) = a, AND(a,0) = 0, AND(a,1) = a, INV of a constant;
//            a gate that drives a circuit OUTPUT wire always stays (constants reach it through the two constant wires)
//   live     backward (consumers first), templates in reverse: a kept gate is live if its out wire is a circuit output or
//            is read by a live gate
//   compact  live gates of every arithmetic gate, in order, behind the two gates that make the constant wires
// ------------------------------------------------------------------------------------------------
struct PruneRun {
    u32 levels, width, M, n_out_wires;
    u64 out_base;                      // first boolean wire of the circuit outputs (n_out arithmetic wires x width)
    u32 zero_wire, one_wire;           // the two constant wires (new: wire_count, wire_count + 1)
    const u32* lbase; const u32* order; const u32* spos;
    const u64* goff; const BoolSegs* segs;
    u32* rep;                          // [wires + 2]
    u32* need;                         // [wires + 2]
    u32* t_in0; u32* t_in1; u8* t_op;  // [G] rewritten gates (t_op 0xFF: folded away; bit 7 set after `live`: dead)
    u32* live_cnt;                     // [n] live gates per arithmetic gate (sorted position)
    u32* bar;
    ull* counts;                       // [0] folded, [1] dead
};
__device__ __forceinline__ u32 ev_ld32(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void ev_st32(u32* p, u32 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// (both passes: a wave per arithmetic gate, its template 64 gates at a time like eval_template_wave)
__global__ void __launch_bounds__(kThreads) k_prune_fold(PruneRun R) {
    const bool one_seg = R.segs->n_seg == 1u;      // (one piece: its pointers are read once per wave, not once per gate)
    const BoolPtrs B_all = seg_of(R.segs, 0u);
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = gtid() >> 6, n_waves = gstride() >> 6;
    u32 target = 0;
    u64 folded = 0;
    const u64 out_end = R.out_base + (u64)R.n_out_wires;
    for (u32 lv = R.levels; lv-- > 0;) {
        const u32 lo = R.lbase[lv], cnt = R.lbase[lv + 1] - lo;
        for (u64 i = wave; i < cnt; i += n_waves) {
            const u32 p = R.spos[R.order[lo + (u32)i]];
            const u64 k0 = R.goff[p], k1 = R.goff[p + 1];
            const BoolPtrs B = one_seg ? B_all : seg_of(R.segs, p);
            for (u64 base = k0; base < k1; base += 64) {
                const u32 nv = k1 - base < 64 ? (u32)(k1 - base) : 64u;
                const bool valid = lane < nv;
                const u64 k = base + lane;
                u32 i0 = 0xFFFFFFFFu, i1 = 0xFFFFFFFFu, out = 0xFFFFFFFEu, op = 2;
                if (valid) { i0 = B.in0[k]; i1 = B.in1[k]; out = B.out[k]; op = B.op[k]; }
                ChunkDeps d = chunk_deps(i0, i1, out, nv, lane);
                if (op == 2u) d.d1 = -1;
                const u32 a_mem = valid && d.d0 < 0 ? ev_ld32(&R.rep[i0]) : 0u;
                const u32 b_mem = valid && op != 2u && d.d1 < 0 ? ev_ld32(&R.rep[i1]) : 0u;
                const bool is_output = (u64)out >= R.out_base && (u64)out < out_end;
                bool done = !valid;
                u32 v = 0;                                   // what the out wire is known to be (a rep)
                for (;;) {
                    const u64 dm = __ballot(done);
                    if (dm == ~0ull) break;
                    const u32 av = __shfl(v, d.d0 < 0 ? (int)lane : d.d0, 64), bv = __shfl(v, d.d1 < 0 ? (int)lane : d.d1, 64);
                    if (!done && (d.d0 < 0 || ((dm >> d.d0) & 1ull)) && (d.d1 < 0 || ((dm >> d.d1) & 1ull))) {
                        const u32 a = d.d0 < 0 ? a_mem : av;
                        const u32 b = op == 2u ? a : (d.d1 < 0 ? b_mem : bv);
                        u32 r = 0xFFFFFFFFu, nop = op, na = a, nb = b;                 // r: the folded value (a rep), or "keep"
                        if (op == 0u) {            // XOR
                            if (a == b) r = 0u; else if (a == 0u) r = b; else if (b == 0u) r = a;
                            else if (a == 1u) { nop = 2u; na = b; nb = b; } else if (b == 1u) { nop = 2u; nb = a; }
                        } else if (op == 1u) {     // AND
                            if (a == b) r = a; else if (a == 0u || b == 0u) r = 0u; else if (a == 1u) r = b; else if (b == 1u) r = a;
                        } else {                   // INV
                            if (a <= 1u) r = 1u - a;
                        }
                        if (r != 0xFFFFFFFFu && !is_output) {
                            v = r;
                            R.t_op[k] = 0xFFu;
                            ++folded;
                        } else {
                            // stays a gate (an output wire must be driven): its inputs by what they are known to be
                            if (r != 0xFFFFFFFFu) { nop = op; na = a; nb = b; }
                            const u32 wa = na == 0u ? R.zero_wire : (na == 1u ? R.one_wire : na - 2u);
                            const u32 wb = nb == 0u ? R.zero_wire : (nb == 1u ? R.one_wire : nb - 2u);
                            R.t_in0[k] = wa; R.t_in1[k] = nop == 2u ? wa : wb; R.t_op[k] = (u8)nop;
                            v = out + 2u;
                        }
                        done = true;
                    }
                }
                if (valid) ev_st32(&R.rep[out], v);
                if (base + 64 < k1) chunk_fence();
            }
        }
        if (lv == 0) break;
        if (!grid_barrier(R.bar, target)) return;
    }
    if (folded) atomicAdd(&R.counts[0], (ull)folded);
}
__global__ void __launch_bounds__(kThreads) k_prune_live(PruneRun R) {
    const bool one_seg = R.segs->n_seg == 1u;      // (one piece: its pointers are read once per wave, not once per gate)
    const BoolPtrs B_all = seg_of(R.segs, 0u);
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = gtid() >> 6, n_waves = gstride() >> 6;
    u32 target = 0;
    u64 dead = 0;
    const u64 out_end = R.out_base + (u64)R.n_out_wires;
    for (u32 lv = 0; lv < R.levels; ++lv) {
        const u32 lo = R.lbase[lv], cnt = R.lbase[lv + 1] - lo;
        for (u64 i = wave; i < cnt; i += n_waves) {
            const u32 p = R.spos[R.order[lo + (u32)i]];
            const u64 k0 = R.goff[p], k1 = R.goff[p + 1];
            const BoolPtrs B = one_seg ? B_all : seg_of(R.segs, p);
            u32 live_total = 0;
            // chunks from the last to the first; inside a chunk a gate is needed by the live gates behind it that read its wire
            for (u64 base = k0 + ((k1 - k0 - 1) & ~63ull); k1 > k0; base -= 64) {
                const u32 nv = k1 - base < 64 ? (u32)(k1 - base) : 64u;
                const u64 k = base + lane;
                u32 op = 0xFFu, out = 0xFFFFFFFEu, t0 = 0xFFFFFFFFu, t1 = 0xFFFFFFFFu;
                if (lane < nv) { op = R.t_op[k]; out = B.out[k]; }
                const bool kept = op != 0xFFu;
                if (kept) { t0 = R.t_in0[k]; t1 = op == 2u ? t0 : R.t_in1[k]; }
                u64 readers = 0;
                for (u32 j = 1; j < nv; ++j) {
                    const u32 r0 = rdlane(t0, j), r1 = rdlane(t1, j);
                    if (j > lane && (out == r0 || out == r1)) readers |= 1ull << j;      // (lanes that hold no kept gate read wire 0xFFFFFFFF)
                }
                const bool is_output = (u64)out >= R.out_base && (u64)out < out_end;
                bool live = kept && (is_output || ev_ld32(&R.need[out]) != 0u);
                for (;;) {
                    const u64 lm = __ballot(live);
                    const bool wake = kept && !live && (readers & lm) != 0ull;
                    if (!__ballot(wake)) break;
                    live = live || wake;
                }
                if (live) {
                    ev_st32(&R.need[t0], 1u);
                    if (op != 2u) ev_st32(&R.need[t1], 1u);
                } else if (kept) {
                    R.t_op[k] = (u8)(op | 0x80u);
                    ++dead;
                }
                live_total += (u32)__popcll(__ballot(live));
                if (base == k0) break;
                chunk_fence();
            }
            if (lane == 0) R.live_cnt[p] = live_total;
        }
        if (lv + 1 == R.levels) break;
        if (!grid_barrier(R.bar, target)) return;
    }
    if (dead) atomicAdd(&R.counts[1], (ull)dead);
}
__global__ void k_prune_init(u64 n_wires, u32* rep, u32* need) {
    for (u64 w = gtid(); w < n_wires; w += gstride()) { rep[w] = (u32)w + 2u; need[w] = 0u; }
}
// the live gates of every arithmetic gate, in order, from position 2 on (0, 1: the gates that make the constant wires)
__global__ void k_prune_compact(u32 n, const u64* __restrict__ goff, const u32* __restrict__ pgoff, const u32* __restrict__ t_in0,
                                const u32* __restrict__ t_in1, const BoolSegs* __restrict__ segs, const u8* __restrict__ t_op,
                                u32* p_in0, u32* p_in1, u32* p_out, u8* p_op) {
    for (u64 p = gtid(); p < n; p += gstride()) {
        u64 q = 2ull + pgoff[p];
        const u32* b_out = seg_of(segs, (u32)p).out;
        for (u64 k = goff[p]; k < goff[p + 1]; ++k) {
            const u32 op = t_op[k];
            if (op & 0x80u) continue;                  // folded (0xFF) or dead (bit 7)
            p_in0[q] = t_in0[k]; p_in1[q] = t_in1[k]; p_out[q] = b_out[k]; p_op[q] = (u8)op;
            ++q;
        }
    }
}
__global__ void k_prune_consts(u32 zero_wire, u32 one_wire, u32* p_in0, u32* p_in1, u32* p_out, u8* p_op) {
    if (gtid() == 0) {
        p_in0[0] = 0; p_in1[0] = 0; p_out[0] = zero_wire; p_op[0] = 0;          // ZERO = XOR(wire 0, wire 0)
        p_in0[1] = zero_wire; p_in1[1] = zero_wire; p_out[1] = one_wire; p_op[1] = 2;   // ONE = INV(ZERO)
    }
}
// the pruned circuit on values: the per-gate ranges are pgoff (+ 2), the two constant wires are set by the caller
__global__ void __launch_bounds__(kThreads) k_eval_pruned(EvalRun R, const u32* __restrict__ pgoff) {
    const u32 lane = threadIdx.x & 63u;
    const u64 wave = gtid() >> 6, n_waves = gstride() >> 6;
    u32 target = 0;
    for (u32 lv = R.levels; lv-- > 0;) {
        const u32 lo = R.lbase[lv], cnt = R.lbase[lv + 1] - lo;
        for (u64 i = wave; i < cnt; i += n_waves) {
            const u32 p = R.spos[R.order[lo + (u32)i]];
            eval_template_wave(2ull + pgoff[p], 2ull + pgoff[p + 1], R.b_in0, R.b_in1, R.b_out, R.b_op, R.bval, lane);
        }
        if (lv == 0) break;
        if (!grid_barrier(R.bar, target)) return;
    }
}

// caller-supplied values: inputs[i][t] for input wire i (wires 0 .. n_in-1, compiler.rs:388-395), vector t < n_vectors
__global__ void k_eval_set_arith(u32 n_wires, u32 n_vectors, u32 width, const u32* __restrict__ wires, const u64* __restrict__ vals, u64* aval) {
    const u64 mk = width >= 64 ? ~0ull : ((1ull << width) - 1ull);
    for (u64 i = gtid(); i < (u64)n_wires * n_vectors; i += gstride()) {
        const u32 k = (u32)(i / n_vectors), t = (u32)(i - (u64)k * n_vectors);
        const u32 W = wires ? wires[k] : k;
        aval[(u64)W * 64 + t] = vals[i] & mk;
    }
}
// ... and their boolean image: bit b of the value of arithmetic wire W in vector t is bit t of boolean wire (W, b)
__global__ void k_eval_set_bool(u32 n_wires, u32 n_vectors, u32 width, u32 M, u64 out_base, const u32* __restrict__ wires,
                                const u64* __restrict__ vals, u64* bval) {
    for (u64 i = gtid(); i < (u64)n_wires * width; i += gstride()) {
        const u32 k = (u32)(i / width), bit = (u32)(i - (u64)k * width);
        const u32 W = wires ? wires[k] : k;
        u64 word = 0;
        for (u32 t = 0; t < n_vectors; ++t) word |= ((vals[(u64)k * n_vectors + t] >> bit) & 1ull) << t;
        bval[bool_wire(W, bit, width, M, out_base)] = word;
    }
}
// the named outputs: the last n_out arithmetic wires (compiler.rs:445-449), as values
__global__ void k_eval_get(u32 n_out, u32 n_vectors, u32 width, u32 first_wire, u32 M, u64 out_base, int boolean,
                           const u64* __restrict__ aval, const u64* __restrict__ bval, u64* vals) {
    for (u64 i = gtid(); i < (u64)n_out * n_vectors; i += gstride()) {
        const u32 j = (u32)(i / n_vectors), t = (u32)(i - (u64)j * n_vectors);
        const u32 W = first_wire + j;
        u64 v = 0;
        if (boolean) for (u32 bit = 0; bit < width; ++bit) v |= ((bval[bool_wire(W, bit, width, M, out_base)] >> t) & 1ull) << bit;
        else v = aval[(u64)W * 64 + t];
        vals[i] = v;
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

// (v[k] is element base + k of the whole stream: a stream held in parts is summed part by part)
__global__ void k_checksum_u32(u64 n, u64 base, const u32* __restrict__ v, ull* acc) {
    u64 local = 0;
    for (u64 k = gtid(); k < n; k += gstride()) { const u64 i = base + k; local += mix64((i << 32) ^ (i >> 32) ^ ((u64)v[k] * 0x9E3779B97F4A7C15ULL)); }
    if (local) atomicAdd(acc, (ull)local);
}
__global__ void k_checksum_u8(u64 n, u64 base, const u8* __restrict__ v, ull* acc) {
    u64 local = 0;
    for (u64 k = gtid(); k < n; k += gstride()) { const u64 i = base + k; local += mix64((i << 32) ^ (i >> 32) ^ ((u64)v[k] * 0x9E3779B97F4A7C15ULL)); }
    if (local) atomicAdd(acc, (ull)local);
}

}  // namespace c2a

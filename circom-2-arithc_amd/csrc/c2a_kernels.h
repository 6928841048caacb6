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
// The exact DFS post-order is reproduced level-synchronously: the DFS tree parent of a
// gate is the consumer that reaches it by the lexicographically smallest path from a virtual root
// (children = gates in id order; edge labels 0 = lh producer, 1 = rh producer).  Gates are peeled in
// reverse Kahn levels from the sinks; each peeled gate picks its parent by comparing consumer paths with
// base-16 ancestor tables (64-byte rows: one cache line per hop), then the post-order index of every
// gate comes from an Euler tour + list ranking — no per-level sweep for the numbering.
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
__global__ void k_producer(u32 n, const u32* __restrict__ out, u32* prod1) {
    for (u64 g = gtid(); g < n; g += gstride()) atomicMax(&prod1[out[g]], (u32)g + 1);
}

// deps closure (compiler.rs:408-421) + consumer counts.  dep1 is dropped when equal to dep0: a second
// visit of the same gate is a no-op in the DFS (topological_sort.rs:30-32).
__global__ void k_deps(u32 n, const u32* __restrict__ lh, const u32* __restrict__ rh, const u32* __restrict__ prod1,
                       u32* dep0, u32* dep1, u32* cons_cnt) {
    for (u64 g = gtid(); g < n; g += gstride()) {
        const u32 p0 = prod1[lh[g]], p1 = prod1[rh[g]];
        const u32 d0 = p0 ? p0 - 1 : C2A_NONE;
        u32 d1 = p1 ? p1 - 1 : C2A_NONE;
        if (d1 == d0) d1 = C2A_NONE;
        dep0[g] = d0;
        dep1[g] = d1;
        if (d0 != C2A_NONE) atomicAdd(&cons_cnt[d0], 1u);
        if (d1 != C2A_NONE) atomicAdd(&cons_cnt[d1], 1u);
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

// Frontier slot: 32 bytes {gate, dep0, dep1, cons_off} {cons_cnt, -, -, -}.  Two slot arrays alternate by level
// parity and slot i of level L is at a fixed address, so a level's first hop (the slot) does not wait for the
// frontier size: fcount[L] is loaded alongside it.
struct FrontierSlot { uint4 a, b; };

// level 0 of the reverse Kahn peel: gates nobody consumes.  fcount[0] must be zero.
__global__ void __launch_bounds__(kThreads) k_init_frontier(u32 n, const uint4* __restrict__ ginfo, FrontierSlot* slots,
                                                            u32* fcount) {
    for (u64 base = (u64)blockIdx.x * kThreads; base < n; base += (u64)gridDim.x * kThreads) {
        const u64 g = base + threadIdx.x;
        uint4 gi = make_uint4(0, 0, 0, 1);
        if (g < n) gi = ginfo[g];
        const bool sink = g < n && gi.w == 0;
        const u32 p = block_append_slot(sink, &fcount[0]);
        if (sink) { slots[p].a = make_uint4((u32)g, gi.x, gi.y, gi.z); slots[p].b = make_uint4(gi.w, 0, 0, 0); }
    }
}

// totals after the peel: {gates peeled, non-empty levels}
__global__ void k_peel_totals(const u32* __restrict__ fcount, u32 n_levels, u32* out2) {
    __shared__ u32 s_sum[kThreads], s_lv[kThreads];
    u32 sum = 0, lv = 0;
    for (u32 i = threadIdx.x; i < n_levels; i += kThreads) { const u32 c = fcount[i]; sum += c; lv += c ? 1u : 0u; }
    s_sum[threadIdx.x] = sum; s_lv[threadIdx.x] = lv;
    __syncthreads();
    for (u32 off = kThreads / 2; off; off >>= 1) {
        if (threadIdx.x < off) { s_sum[threadIdx.x] += s_sum[threadIdx.x + off]; s_lv[threadIdx.x] += s_lv[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out2[0] = s_sum[0]; out2[1] = s_lv[0]; }
}

// ------------------------------------------------------------------------------------------------
// peel one level + pick DFS-tree parents
// ------------------------------------------------------------------------------------------------
// Tree node == peel position (fbase[level] + slot index): nodes of recent levels are contiguous, which keeps
// the short hops of a path comparison inside a few MB (at 10 M gates the tables span GBs and a scattered hop
// costs ~2x a local one).  meta[pos] = {parent pos | NONE, depth, root gate id, label of the edge parent->node}.
// anc plane j, row pos: 16 entries = ancestor at distance (d+1)*16^j, valid while <= depth, each entry
// carrying that ancestor's own edge label in bit 31 (so a path comparison never has to fetch meta[]).
// Candidate lists are filled as consumers are peeled, AFTER their tournament, with everything a comparison
// starts from: cand[cons_off[d] + k] = {consumer | edge label << 31, consumer depth, consumer root, consumer's
// own edge label}; the push that completes d's list (k + 1 == cons_cnt[d]) writes d's frontier slot.
// Dependent memory hops on a level's critical path: slot -> candidate records -> <= 3 (lift) + <= 3 (diverge)
// ancestor rows -> max(<= 3 row copies, fill atomic + slot atomic).
constexpr u32 kIdMask = 0x7FFFFFFFu;

struct PeelArgs {
    u32 n;
    const uint4* ginfo;        // [n] {dep0, dep1, cons_off, cons_cnt}
    FrontierSlot* slots[2];    // by level parity
    uint4* cand;               // [edges]
    u32* fill;                 // pushes so far per gate (zeroed)
    uint4* meta;               // [n] by position
    u32* anc;                  // [planes][n][16] by position (ancestor-row representation)
    u64* pstr;                 // [n][64] path strings by position (path-string representation)
    u32* cprev;                // [n] ancestor at the start of the node's current chunk (only beyond depth 4096)
    u32* order;                // position -> gate
    u32* child;                // [2n] tree children by label: child[2*p + l] (written as each gate picks its parent)
    u32* posof;                // gate -> position
    u32* fbase;                // [levels+2] first position of each level
    u32* fcount;               // [levels+2] frontier sizes (slot allocation + host monitoring)
    ull* prof;                 // optional phase timestamps (diagnostics; nullptr normally)
    u32 prof_level0;           // first level recorded
};

// Two access flavours.  SC1 = false: plain loads/stores — data produced by EARLIER launches (the kernel
// boundary publishes it).  SC1 = true: relaxed agent-scope atomics (global_load/store ... sc1: write-through
// stores, L1-bypassing loads) — data exchanged between workgroups INSIDE one persistent launch; the only form
// that handed off without stale reads in tools/ubench/xcd.hip (plain stores + nt or sc1 loads did not).
template <bool SC1> __device__ __forceinline__ u32 ld_u32(const u32* p) {
    if (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool SC1> __device__ __forceinline__ void st_u32(u32* p, u32 v) {
    if (SC1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *p = v;
}
template <bool SC1> __device__ __forceinline__ u64 ld_u64(const u64* p) {
    if (SC1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}
template <bool SC1> __device__ __forceinline__ uint4 ld_u128(const uint4* p) {
    if (SC1) {
        const u64* q = reinterpret_cast<const u64*>(p);
        const u64 a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const u64 b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_uint4((u32)a, (u32)(a >> 32), (u32)b, (u32)(b >> 32));
    }
    return *p;
}
template <bool SC1> __device__ __forceinline__ void st_u128(uint4* p, const uint4& v) {
    if (SC1) {
        u64* q = reinterpret_cast<u64*>(p);
        __hip_atomic_store(q, (u64)v.x | ((u64)v.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 1, (u64)v.z | ((u64)v.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        *p = v;
    }
}

// B = log2(entries per ancestor row): 4 -> base-16 rows of 64 B, 6 -> base-64 rows of 256 B.  A wider row costs
// bytes (not the budget here) and removes dependent hops (the budget): at depth < 4096 a lift, a divergence search
// and a row copy take <= 2 hops each with B = 6 instead of <= 3 with B = 4.
template <bool SC1, int B>
__device__ __forceinline__ u32 anc_entry(const u32* anc, u64 plane, int j, u32 x, u32 d) {
    return ld_u32<SC1>(anc + (u64)j * plane + ((u64)x << B) + d);
}

template <bool SC1, int B>
__device__ __forceinline__ void load_row(const u32* anc, u64 plane, int j, u32 x, u32 (&r)[1 << B]) {
    const uint4* p = reinterpret_cast<const uint4*>(anc + (u64)j * plane + ((u64)x << B));
#pragma unroll
    for (int q = 0; q < (1 << B) / 4; ++q) {
        const uint4 v = ld_u128<SC1>(p + q);
        r[4 * q + 0] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
    }
}

// Two lifts of the same node in one pass (their loads are independent and overlap): xe = ancestor at `dist`,
// ye = ancestor at `dist - 1`, both as labelled entries.  dist >= 1; `self` = the node's own labelled entry.
template <bool SC1, int B>
__device__ __forceinline__ void lift2(const u32* anc, u64 plane, u32 self, u32 dist, u32& xe, u32& ye) {
    u32 x = self, y = self;
    u32 dx = dist, dy = dist - 1;
    int j = 0;
    while (dx | dy) {
        const u32 ex = dx & ((1u << B) - 1u), ey = dy & ((1u << B) - 1u);
        u32 nx = x, ny = y;
        if (ex) nx = anc_entry<SC1, B>(anc, plane, j, x & kIdMask, ex - 1);
        if (ey) ny = anc_entry<SC1, B>(anc, plane, j, y & kIdMask, ey - 1);
        x = nx; y = ny;
        dx >>= B; dy >>= B;
        ++j;
    }
    xe = x; ye = y;
}

// ae != be (ids) at equal depth D >= 1 under one root: walk both up to the children of their lowest common
// ancestor, one 64-byte row per node per base-16 digit; entries keep their label bits.
template <bool SC1, int B>
__device__ __forceinline__ void diverge(const u32* anc, u64 plane, u32& ae, u32& be, u32 D) {
    if (D == 0) return;
    constexpr u32 R = 1u << B;
    int j = (31 - __clz(D)) / B;
    for (; j >= 0; --j) {
        u32 m = D >> (B * j);
        if (m == 0) continue;
        if (m > R) m = R;
        u32 ra[R], rb[R];
        load_row<SC1, B>(anc, plane, j, ae & kIdMask, ra);
        load_row<SC1, B>(anc, plane, j, be & kIdMask, rb);
        u32 pa = ae, pb = be, pd = 0;
#pragma unroll
        for (u32 d = 0; d < R; ++d) {
            if (d < m && ((ra[d] ^ rb[d]) & kIdMask)) { pa = ra[d]; pb = rb[d]; pd = d + 1; }
        }
        ae = pa; be = pb;
        D -= pd << (B * j);
    }
}

// A candidate = "the path to consumer c, then edge `el`".  ce = c | own edge label of c << 31.
// is P(a).ela < P(b).elb ?  ids differ, same root.
template <bool SC1, int B>
__device__ __forceinline__ bool path_less(const u32* anc, u64 plane, u32 ae, u32 ela, u32 da, u32 be, u32 elb, u32 db) {
    if (da > db) {
        u32 up, below;
        lift2<SC1, B>(anc, plane, ae, da - db, up, below);
        if (((up ^ be) & kIdMask) == 0) return (below >> 31) < elb;      // b is an ancestor of a
        ae = up;
        diverge<SC1, B>(anc, plane, ae, be, db);
    } else if (db > da) {
        u32 up, below;
        lift2<SC1, B>(anc, plane, be, db - da, up, below);
        if (((up ^ ae) & kIdMask) == 0) return ela < (below >> 31);      // a is an ancestor of b
        be = up;
        diverge<SC1, B>(anc, plane, ae, be, da);
    } else {
        diverge<SC1, B>(anc, plane, ae, be, da);
    }
    return (ae >> 31) < (be >> 31);
}

// Variant 1: one lane per frontier gate, candidates compared one after the other.  Used while the frontier
// is wide (the first levels); its latency per level is (largest fan-out) x (one path comparison).
template <int B>
__global__ void __launch_bounds__(kThreads) k_peel_level(PeelArgs A, u32 level) {
    FrontierSlot* cur = A.slots[level & 1u];
    FrontierSlot* nxt = A.slots[(level + 1) & 1u];
    constexpr u32 R = 1u << B;
    const u64 plane = (u64)A.n << B;
    const u32 n_front = A.fcount[level];
    const u32 lo = A.fbase[level];
    if (gtid() == 0) A.fbase[level + 1] = lo + n_front;
    for (u64 i = gtid(); i < n_front; i += gstride()) {
        const uint4 sa = cur[i].a;
        const u32 cnt = cur[i].b.x;
        const u32 g = sa.x;
        const u32 pos = lo + (u32)i;
        // ---- tournament over the candidate paths: [g] (child of the virtual root) and P(c).l per consumer
        u32 best = C2A_NONE, best_el = 0, best_root = g, best_depth = 0;     // best = labelled entry of the consumer
        const u32 e0 = sa.w, e1 = e0 + cnt;
        for (u32 e = e0; e < e1; ++e) {
            const uint4 cr = A.cand[e];
            const u32 ce = (cr.x & kIdMask) | (cr.w << 31), el = cr.x >> 31;
            bool take;
            if (best == C2A_NONE) take = cr.z < g;
            else if (cr.z != best_root) take = cr.z < best_root;
            else if (((ce ^ best) & kIdMask) == 0) take = el < best_el;
            else take = path_less<false, B>(A.anc, plane, ce, el, cr.y, best, best_el, best_depth);
            if (take) { best = ce; best_el = el; best_root = cr.z; best_depth = cr.y; }
        }
        const u32 depth = best == C2A_NONE ? 0u : best_depth + 1;
        const u32 my_label = best == C2A_NONE ? 0u : best_el;
        A.meta[pos] = make_uint4(best == C2A_NONE ? C2A_NONE : (best & kIdMask), depth, best_root, my_label);
        A.order[pos] = g;
        A.posof[g] = pos;
        if (best != C2A_NONE) A.child[2 * (u64)(best & kIdMask) + my_label] = pos;
        // ---- ancestor rows: row j = [q_j, row_j(q_j)[0..R-2]], q_0 = parent, q_{j+1} = my ancestor at R^(j+1)
        if (depth) {
            u32 q = best;                                   // labelled entry
            u32 need = 1;                                   // R^j
            for (int j = 0; need <= depth; ++j) {
                u32 r[R];
                load_row<false, B>(A.anc, plane, j, q & kIdMask, r);
                uint4* dst = reinterpret_cast<uint4*>(A.anc + (u64)j * plane + ((u64)pos << B));
                dst[0] = make_uint4(q, r[0], r[1], r[2]);
#pragma unroll
                for (u32 t = 1; t < R / 4; ++t) dst[t] = make_uint4(r[4 * t - 1], r[4 * t], r[4 * t + 1], r[4 * t + 2]);
                q = r[R - 2];
                if (need > (0xFFFFFFFFu >> B)) break;
                need <<= B;
            }
        }
        // ---- tell the producers; a producer joins the next frontier when its last consumer has been peeled
        const u32 deps[2] = {sa.y, sa.z};
#pragma unroll
        for (u32 l = 0; l < 2; ++l) {
            const u32 d = deps[l];
            if (d == C2A_NONE) continue;
            const uint4 gd = A.ginfo[d];
            const u32 k = atomicAdd(&A.fill[d], 1u);
            A.cand[gd.z + k] = make_uint4(pos | (l << 31), depth, best_root, my_label);
            if (k + 1 == gd.w) {
                const u32 p = atomicAdd(&A.fcount[level + 1], 1u);
                nxt[p].b = make_uint4(gd.w, 0, 0, 0);
                nxt[p].a = make_uint4(d, gd.x, gd.y, gd.z);
            }
        }
    }
}

// Variant 2: one WAVE per frontier gate.  Lanes load the candidate records in parallel, candidates with a
// larger DFS root are dropped by a wave-wide min, and the survivors play a one-round all-pairs tournament (one
// path comparison per lane, <= 11 candidates = 55 pairs per round) — so the latency per level is about ONE path
// comparison whatever the fan-out.  Ancestor rows are copied 16 lanes wide (one 64-byte line per plane) while
// the producers' fill atomics and the workgroup's single frontier append are in flight.
constexpr int kGroup = 11;                                // 11*10/2 = 55 pairs <= 64 lanes

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
// diagnostics: every wave of levels [256, 288) stores its phase timestamps (slot 6 = start, 7 = candidates)
constexpr u32 kProfLevels = 32, kProfWaves = 32768;
#define C2A_PROF_IF(on, slot, value) do { if (on) C2A_PROF(slot, value); } while (0)
#define C2A_PROF(slot, value)                                                                                   \
    do {                                                                                                        \
        if (A.prof && lane == 0 && level >= A.prof_level0 && level < A.prof_level0 + kProfLevels) {             \
            const u32 wg_ = blockIdx.x * WPB + wv;                                                              \
            if (wg_ < kProfWaves) A.prof[((u64)(level - A.prof_level0) * kProfWaves + wg_) * 8 + (slot)] = (value); \
        }                                                                                                       \
    } while (0)

// WPB = waves (= gates per pass) per workgroup: 16 -> fewest appends on the frontier counter, 4/8 -> shorter
// wait for the slowest wave of the group
template <int WPB, int B>
__global__ void __launch_bounds__(WPB * 64) k_peel_level_wave(PeelArgs A, u32 level) {
    const ull t_begin = A.prof ? c2a_now() : 0;
    __shared__ u32 s_c[WPB][72], s_l[WPB][72], s_d[WPB][72];
    __shared__ u32 s_ready[2 * WPB];
    __shared__ uint4 s_rec[2 * WPB];
    __shared__ u32 s_base;
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    FrontierSlot* cur = A.slots[level & 1u];
    FrontierSlot* nxt = A.slots[(level + 1) & 1u];
    constexpr u32 R = 1u << B;
    const u64 plane = (u64)A.n << B;
    const u64 lt_mask = (1ull << lane) - 1ull;
    // slot i sits at a fixed address: issue its load before the frontier size is known
    uint4 sa0 = make_uint4(0, 0, 0, 0);
    u32 cnt0 = 0;
    {
        const u64 i0 = (u64)blockIdx.x * WPB + wv;
        if (i0 < A.n) { sa0 = cur[i0].a; cnt0 = cur[i0].b.x; }
    }
    const u32 n_front = A.fcount[level];
    const u32 lo = A.fbase[level];                       // only needed after the tournament
    if (gtid() == 0) A.fbase[level + 1] = lo + n_front;
    for (u32 chunk = blockIdx.x; (u64)chunk * WPB < n_front; chunk += gridDim.x) {
        const u64 i = (u64)chunk * WPB + wv;
        u32 rdy = C2A_NONE;                              // producer completed by this wave's gate (lanes 0 / 1)
        uint4 rdy_rec = make_uint4(0, 0, 0, 0);
        uint4 sa = sa0;
        u32 cnt = cnt0;
        if (chunk != blockIdx.x && i < n_front) { sa = cur[i].a; cnt = cur[i].b.x; }
        if (i < n_front) {
            const u32 g = sa.x;
            const u32 e0 = sa.w, e1 = e0 + cnt;
            // lanes 0/1 own the two producers: fetch their records now, off the critical path
            const u32 dl = lane == 0 ? sa.y : (lane == 1 ? sa.z : C2A_NONE);
            uint4 gd = make_uint4(0, 0, 0, 0);
            if (dl != C2A_NONE) gd = A.ginfo[dl];
            C2A_PROF(0, c2a_now() - t_begin);
            C2A_PROF(6, t_begin);
            C2A_PROF(7, (ull)cnt + 1000);
            // champion so far (wave-uniform): labelled consumer entry; NONE = the virtual-root candidate [g]
            u32 ch = C2A_NONE, ch_el = 0, ch_root = g, ch_depth = 0;
            for (u32 eb = e0; eb < e1; eb += 64) {
                const u32 e = eb + lane;
                const bool valid = e < e1;
                u32 c = 0, l = 0, cdepth = 0, croot = 0xFFFFFFFFu;
                if (valid) {
                    const uint4 cr = A.cand[e];
                    c = (cr.x & kIdMask) | (cr.w << 31); l = cr.x >> 31;
                    cdepth = cr.y; croot = cr.z;
                }
                const u32 rmin = wave_min_u32(croot);
                C2A_PROF(1, c2a_now() - t_begin);
                if (rmin > ch_root) continue;                       // the whole chunk starts from a later DFS root
                const bool keep_ch = (ch != C2A_NONE) && (ch_root == rmin);
                const bool surv = valid && croot == rmin;
                const u64 smask = __ballot(surv);
                u32 m = (u32)__popcll(smask);
                if (surv) {
                    const u32 k = (u32)__popcll(smask & lt_mask);
                    s_c[wv][k] = c; s_l[wv][k] = l; s_d[wv][k] = cdepth;
                }
                if (keep_ch && lane == 0) { s_c[wv][m] = ch; s_l[wv][m] = ch_el; s_d[wv][m] = ch_depth; }
                m += keep_ch ? 1u : 0u;
                wave_lds_sync();
                // all-pairs rounds over groups of <= kGroup candidates: [winner so far] + next candidates
                u32 win = 0;        // index (in s_*) of the current winner
                u32 next = 1;       // next unplayed candidate
                while (next < m) {
                    const u32 take = (m - next) < (u32)(kGroup - 1) ? (m - next) : (u32)(kGroup - 1);
                    const u32 q = take + 1;                         // group: member 0 = win, member t = next+t-1
                    const u32 P = q * (q - 1) / 2;
                    u32 pi = 0, pj = 1;                             // lane -> pair (i<j), triangular enumeration
                    {
                        u32 rem = lane, row = 0, len = q - 1;
                        while (len && rem >= len) { rem -= len; ++row; --len; }
                        pi = row; pj = row + 1 + rem;
                    }
                    u32 loser = 0xFFFFFFFFu;
                    if (lane < P) {
                        const u32 xi = pi == 0 ? win : next + pi - 1, xj = next + pj - 1;
                        const u32 ci = s_c[wv][xi], cj = s_c[wv][xj];
                        const u32 li = s_l[wv][xi], lj = s_l[wv][xj];
                        bool less;
                        if (((ci ^ cj) & kIdMask) == 0) less = li < lj;
                        else less = path_less<false, B>(A.anc, plane, ci, li, s_d[wv][xi], cj, lj, s_d[wv][xj]);
                        loser = less ? pj : pi;
                    }
                    u32 w = 0;
                    for (u32 t = 0; t < q; ++t) {
                        const u64 lost = __ballot(loser == t);
                        if (lost == 0) w = t;
                    }
                    win = w == 0 ? win : next + w - 1;
                    next += take;
                }
                ch = s_c[wv][win]; ch_el = s_l[wv][win]; ch_depth = s_d[wv][win]; ch_root = rmin;
                wave_lds_sync();
            }
            C2A_PROF(2, c2a_now() - t_begin);
            const u32 depth = ch == C2A_NONE ? 0u : ch_depth + 1;
            const u32 my_label = ch == C2A_NONE ? 0u : ch_el;
            const u32 pos = lo + (u32)i;
            if (lane == 0) {
                A.meta[pos] = make_uint4(ch == C2A_NONE ? C2A_NONE : (ch & kIdMask), depth, ch_root, my_label);
                A.order[pos] = g;
                A.posof[g] = pos;
                if (ch != C2A_NONE) A.child[2 * (u64)(ch & kIdMask) + my_label] = pos;
            }
            // push myself to the producers (lanes 0/1): the atomic's round trip overlaps the row copies below
            u32 kfill = 0;
            if (dl != C2A_NONE) kfill = atomicAdd(&A.fill[dl], 1u);
            if (depth) {
                u32 q = ch;                                 // labelled entry
                u32 need = 1;
                for (int j = 0; need <= depth; ++j) {
                    u32 v = q;
                    if (lane >= 1 && lane < R) v = anc_entry<false, B>(A.anc, plane, j, q & kIdMask, lane - 1);
                    if (lane < R) A.anc[(u64)j * plane + ((u64)pos << B) + lane] = v;
                    q = __shfl(v, (int)R - 1, 64);
                    if (need > (0xFFFFFFFFu >> B)) break;
                    need <<= B;
                }
            }
            if (dl != C2A_NONE) {
                A.cand[gd.z + kfill] = make_uint4(pos | (lane << 31), depth, ch_root, my_label);   // lane == edge label
                if (kfill + 1 == gd.w) { rdy = dl; rdy_rec = gd; }
            }
            C2A_PROF(3, c2a_now() - t_begin);
        }
        // ---- one append per workgroup: a single counter takes ~12 ns per atomic, so per-gate appends would
        // cost more than the whole level (MI355X_MICROARCH.md price list, row "fanin")
        if (lane < 2) { s_ready[2 * wv + lane] = rdy; s_rec[2 * wv + lane] = rdy_rec; }
        __syncthreads();
        C2A_PROF(4, c2a_now() - t_begin);
        if (wv == 0) {
            const u32 d = lane < 2 * WPB ? s_ready[lane] : C2A_NONE;
            const u64 mask = __ballot(d != C2A_NONE);
            if (mask) {
                if (lane == 0) s_base = atomicAdd(&A.fcount[level + 1], (u32)__popcll(mask));
                wave_lds_sync();
                if (d != C2A_NONE) {
                    const u32 p = s_base + (u32)__popcll(mask & lt_mask);
                    const uint4 gd = s_rec[lane];
                    nxt[p].b = make_uint4(gd.w, 0, 0, 0);
                    nxt[p].a = make_uint4(d, gd.x, gd.y, gd.z);
                }
            }
        }
        __syncthreads();
        C2A_PROF(5, c2a_now() - t_begin);
    }
}

// ================================================================================================
// PATH STRINGS — the second representation of "the path from the DFS root to a tree node", built to cut the
// dependent memory hops of a path comparison from <= 6 (lift + diverge over ancestor rows) to ONE.
// Every tree node stores the edge labels of its path as a bit string: bit j = label of the edge entering depth
// j+1.  A string is held in chunks of kChunkBits = 4096 bits = 512 B = one coalesced 8-byte load per lane of a wave;
// a node keeps only its CURRENT chunk (bits [ci*K, depth), zero padded) plus cprev = its ancestor at depth ci*K,
// whose own string is the complete previous chunk — so storage is 512 B per node whatever the depth, and for trees
// shallower than 4096 (the 10 M-gate headline config: 3 471) comparing two candidates is: load both strings (one
// round trip, coalesced), XOR, ballot, count trailing zeros.  Deeper trees add one cprev hop per chunk level.
// A new node's string = parent's string + one bit: ONE hop to build (three for the ancestor rows).
// ================================================================================================
constexpr u32 kSlotPad = 1u << 16;        // slots past the frontier that a speculative prefetch may touch
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

// wave-cooperative comparison of two strings already in registers (this lane's word of each): wave-uniform result
__device__ __forceinline__ bool str_less_wave(u64 wa, u32 lena, u32 la, u64 wb, u32 lenb, u32 lb, u32 lane) {
    const u32 minlen = lena < lenb ? lena : lenb;
    u64 x = wa ^ wb;
    const u32 lo = lane * 64;
    if (lo >= minlen) x = 0;
    else if (minlen - lo < 64) x &= (1ull << (minlen - lo)) - 1ull;
    const u64 bal = __ballot(x != 0);
    if (bal) {
        const int L = (int)ctz64(bal);
        const u64 xl = __shfl(x, L, 64);
        const u64 al = __shfl(wa, L, 64);
        return ((al >> ctz64(xl)) & 1ull) == 0;
    }
    if (lena == lenb) return la < lb;
    if (lena < lenb) return la < ((__shfl(wb, (int)(lena >> 6), 64) >> (lena & 63u)) & 1ull);
    return ((__shfl(wa, (int)(lenb >> 6), 64) >> (lenb & 63u)) & 1ull) < lb;
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

// ---- one lane per gate (wide frontiers)
__global__ void __launch_bounds__(kThreads) k_peel_level_str(PeelArgs A, u32 level) {
    FrontierSlot* cur = A.slots[level & 1u];
    FrontierSlot* nxt = A.slots[(level + 1) & 1u];
    const u32 n_front = A.fcount[level];
    const u32 lo = A.fbase[level];
    if (gtid() == 0) A.fbase[level + 1] = lo + n_front;
    for (u64 i = gtid(); i < n_front; i += gstride()) {
        const uint4 sa = cur[i].a;
        const u32 cnt = cur[i].b.x;
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
        A.meta[pos] = make_uint4(best, depth, best_root, my_label);
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
            if (k + 1 == gd.w) {
                const u32 p = atomicAdd(&A.fcount[level + 1], 1u);
                nxt[p].b = make_uint4(gd.w, 0, 0, 0);
                nxt[p].a = make_uint4(d, gd.x, gd.y, gd.z);
            }
        }
    }
}

// ---- one wave per gate (narrow frontiers): survivor strings live in registers, the whole tournament costs ONE
// round trip to memory when every survivor is shallower than a chunk
constexpr int kStrMax = 12;               // survivor strings held in registers per round

// Workgroup = WPB gate waves + ONE append wave.  Dependent memory round trips per level:
//   1. frontier slot (fixed address, prefetched) + this level's count
//   2. candidate records  ||  ginfo of the two producers  ||  fill[] tickets of the two pushes
//   3. survivor strings   ||  (append wave) next-level counter ticket -> next-level slots
//   then only stores (meta, order, child, the new string, the two candidate records).
// The pushes' tickets need nothing from the tournament, so "am I the last consumer of this producer" — and with it the whole
// next-level frontier — is known after round trip 2, and the append overlaps the tournament.
#ifndef C2A_X
#define C2A_X 0      // timing experiments: 1 = no string loads, 2 = + no candidate loads, 3 = + no tournament/stores (results wrong)
#endif
template <int WPB, bool PROF>
__global__ void __launch_bounds__((WPB + 1) * 64) k_peel_level_wave_str(PeelArgs A, u32 level) {
    const ull t_begin = PROF ? c2a_now() : 0;
    __shared__ u32 s_c[WPB][72], s_l[WPB][72], s_d[WPB][72];
    __shared__ u32 s_ready[2 * WPB];
    __shared__ uint4 s_rec[2 * WPB];
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const bool gate_wave = wv < (u32)WPB;
    const u32 wvc = gate_wave ? wv : 0u;      // LDS row (the append wave never touches its row)
    // both pointers come with the one kernarg fetch; indexing the array by level would be a second, dependent scalar load
    FrontierSlot* cur = (level & 1u) ? A.slots[1] : A.slots[0];
    FrontierSlot* nxt = (level & 1u) ? A.slots[0] : A.slots[1];
    const u64 lt_mask = (1ull << lane) - 1ull;
    uint4 sa0 = make_uint4(0, 0, 0, 0);
    u32 cnt0 = 0;
    if (gate_wave) {
        const u64 i0 = (u64)blockIdx.x * WPB + wv;
        sa0 = cur[i0].a; cnt0 = cur[i0].b.x;      // speculative, before the level's count is known: the slot arrays are padded (kSlotPad)
    }
    const u32 n_front = A.fcount[level];
    const u32 lo = A.fbase[level];
    if (gtid() == 0) A.fbase[level + 1] = lo + n_front;
    if (gate_wave) { C2A_PROF_IF(PROF, 6, t_begin); C2A_PROF_IF(PROF, 0, c2a_now() - t_begin); }
    for (u32 chunk = blockIdx.x; (u64)chunk * WPB < n_front; chunk += gridDim.x) {
        const u64 i = (u64)chunk * WPB + wv;
        const bool active = gate_wave && i < n_front;
        uint4 sa = sa0;
        u32 cnt = cnt0;
        if (chunk != blockIdx.x && active) { sa = cur[i].a; cnt = cur[i].b.x; }
        // ---- round trip 2: producers' ginfo, fill tickets, first block of candidate records
        const u32 g = sa.x;
        const u32 e0 = sa.w, e1 = e0 + cnt;
        const u32 dl = active ? (lane == 0 ? sa.y : (lane == 1 ? sa.z : C2A_NONE)) : C2A_NONE;
        uint4 gd = make_uint4(0, 0, 0, 0);
        u32 kfill = 0;
        uint4 cr_first = make_uint4(0, 0, 0xFFFFFFFFu, 0);
#if C2A_X < 2
        if (active && e0 + lane < e1) cr_first = A.cand[e0 + lane];
#endif
        if (dl != C2A_NONE) { gd = A.ginfo[dl]; kfill = atomicAdd(&A.fill[dl], 1u); }
        const bool last_push = dl != C2A_NONE && kfill + 1 == gd.w;
        if (gate_wave && lane < 2) { s_ready[2 * wv + lane] = last_push ? dl : C2A_NONE; s_rec[2 * wv + lane] = gd; }
        if (active) { C2A_PROF_IF(PROF, 7, 1000ull + cnt); C2A_PROF_IF(PROF, 1, c2a_now() - t_begin); }
        __syncthreads();
        if (active) C2A_PROF_IF(PROF, 2, c2a_now() - t_begin);
        if (!gate_wave) {
            // ---- append wave: one ticket on the next level's counter for the whole workgroup
            const u32 d = lane < 2 * WPB ? s_ready[lane] : C2A_NONE;
            const u64 mask = __ballot(d != C2A_NONE);
            if (mask) {
                u32 base = 0;
                if (lane == 0) base = atomicAdd(&A.fcount[level + 1], (u32)__popcll(mask));
                base = __shfl(base, 0, 64);
                if (d != C2A_NONE) {
                    const u32 p = base + (u32)__popcll(mask & lt_mask);
                    const uint4 r = s_rec[lane];
                    nxt[p].b = make_uint4(r.w, 0, 0, 0);
                    nxt[p].a = make_uint4(d, r.x, r.y, r.z);
                }
            }
        } else if (active && C2A_X < 3) {
            // champion so far (wave-uniform); NONE = the virtual-root candidate [g].  champ_w = this lane's word of
            // the champion's string when champ_loaded
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
                const u32 rmin = wave_min_u32(croot);
                if (rmin > ch_root) continue;
                const bool keep_ch = (ch != C2A_NONE) && (ch_root == rmin);
                if (!keep_ch) champ_loaded = false;
                const bool surv = valid && croot == rmin;
                const u64 smask = __ballot(surv);
                const u32 m = (u32)__popcll(smask);
                if (surv) {
                    const u32 k = (u32)__popcll(smask & lt_mask);
                    s_c[wvc][k] = c; s_l[wvc][k] = l; s_d[wvc][k] = cdepth;
                }
                wave_lds_sync();
                // sequential tournament, kStrMax survivors per round with their strings in registers
                u32 next = 0;
                if (!keep_ch) {      // the first survivor becomes the champion without a comparison
                    ch = s_c[wvc][0]; ch_el = s_l[wvc][0]; ch_depth = s_d[wvc][0];
                    next = 1;
                }
                ch_root = rmin;
                while (next < m) {
                    const u32 take = (m - next) < (u32)kStrMax ? (m - next) : (u32)kStrMax;
                    // all of this round in chunk 0?  then one coalesced load per string, all issued back to back;
                    // lanes beyond a string's length skip their word (it is zero)
                    bool shallow = ch_depth <= kChunkBits;
                    for (u32 t = 0; t < take; ++t) shallow = shallow && s_d[wvc][next + t] <= kChunkBits;
                    if (shallow) {
                        u64 sw[kStrMax];
#pragma unroll
                        for (int t = 0; t < kStrMax; ++t)
                            sw[t] = (C2A_X < 1 && (u32)t < take && lane * 64 < s_d[wvc][next + t]) ? A.pstr[(u64)s_c[wvc][next + t] * kChunkWords + lane] : 0ull;
                        if (!champ_loaded) { champ_w = (C2A_X < 1 && lane * 64 < ch_depth) ? A.pstr[(u64)ch * kChunkWords + lane] : 0ull; champ_loaded = true; }
#pragma unroll
                        for (int t = 0; t < kStrMax; ++t) {
                            if ((u32)t < take) {
                                const u32 cc = s_c[wvc][next + t], cl = s_l[wvc][next + t], cd = s_d[wvc][next + t];
                                bool less;
                                if (cc == ch) less = cl < ch_el;
                                else less = str_less_wave(sw[t], cd, cl, champ_w, ch_depth, ch_el, lane);
                                if (less) { ch = cc; ch_el = cl; ch_depth = cd; champ_w = sw[t]; }
                            }
                        }
                    } else {
                        // deep trees: chunk resolution (cprev hops) per comparison, strings loaded per comparison
                        for (u32 t = 0; t < take; ++t) {
                            const u32 cc = s_c[wvc][next + t], cl = s_l[wvc][next + t], cd = s_d[wvc][next + t];
                            bool less;
                            if (cc == ch) less = cl < ch_el;
                            else {
                                u32 ra = cc, rb = ch, lena, lenb, ba, bb;
                                resolve_chunks(A.cprev, ra, lena, ba, cd, rb, lenb, bb, ch_depth);
                                if (ra == rb) {
                                    if (ba != C2A_NONE) less = (A.pstr[(u64)ba * kChunkWords] & 1ull) < ch_el;
                                    else less = cl < (A.pstr[(u64)bb * kChunkWords] & 1ull);
                                } else {
                                    const u64 wa = lena ? A.pstr[(u64)ra * kChunkWords + lane] : 0ull;
                                    const u64 wb = lenb ? A.pstr[(u64)rb * kChunkWords + lane] : 0ull;
                                    less = str_less_wave(wa, lena, cl, wb, lenb, ch_el, lane);
                                }
                            }
                            if (less) { ch = cc; ch_el = cl; ch_depth = cd; champ_loaded = false; }
                        }
                    }
                    next += take;
                }
                wave_lds_sync();
            }
            C2A_PROF_IF(PROF, 3, c2a_now() - t_begin);
            const u32 depth = ch == C2A_NONE ? 0u : ch_depth + 1;
            const u32 my_label = ch == C2A_NONE ? 0u : ch_el;
            const u32 pos = lo + (u32)i;
            if (lane == 0) {
                A.meta[pos] = make_uint4(ch, depth, ch_root, my_label);
                A.order[pos] = g;
                A.posof[g] = pos;
                if (ch != C2A_NONE) A.child[2 * (u64)ch + my_label] = pos;
            }
            if (dl != C2A_NONE) A.cand[gd.z + kfill] = make_uint4(pos | (lane << 31), depth, ch_root, my_label);   // lane == edge label
            if (ch != C2A_NONE) {
                const bool need_parent = ch_depth != 0 && chunk_of(depth) == chunk_of(ch_depth);
                if (need_parent && !champ_loaded) champ_w = (C2A_X < 1 && lane * 64 < chunk_len(ch_depth)) ? A.pstr[(u64)ch * kChunkWords + lane] : 0ull;
                bool fresh;
                const u64 nw = child_word(need_parent ? champ_w : 0ull, ch_depth, my_label, lane, fresh);
                if (lane * 64 < chunk_len(depth)) A.pstr[(u64)pos * kChunkWords + lane] = nw;      // words past the end are never read
                if (lane == 0 && chunk_of(depth)) A.cprev[pos] = fresh ? ch : A.cprev[ch];
            }
            C2A_PROF_IF(PROF, 4, c2a_now() - t_begin);
        }
        __syncthreads();
        if (active) C2A_PROF_IF(PROF, 5, c2a_now() - t_begin);
    }
}

// Variant 3: ONE persistent launch for all remaining (narrow) levels, confined to one XCD.
// A kernel boundary per level costs ~13 us here (cold caches after every boundary: ~0.6 us per dependent hop,
// plus launch/teardown).  Inside one launch the level step is a workgroup-aggregated append + a counter barrier:
// measured 1.7 us per produce/barrier/consume iteration for the 32 workgroups of one XCD (tools/ubench/xcd.hip).
// Rules (cdna_hip_programming.md §6 G16): every word exchanged between workgroups in the launch is accessed with
// agent-scope relaxed atomics (sc1), every storing wave drains vmcnt before the barrier, correctness never
// depends on placement — the XCD filter only decides WHO works (HW_REG_XCC_ID is ground truth, the first
// workgroup to arrive picks the XCD), the census tells the participants how many they are, and they are all
// resident by then (they have all reported).  One 16-lane group per gate, four gates per wave:
// candidates in chunks of 16, all-pairs rounds of <= 6 candidates (15 pairs), all cross-lane traffic by
// width-16 shuffles (group-uniform control flow).
struct PeelCtl {
    u32 chosen_xcd;     // 0xFFFFFFFF until the first workgroup arrives
    u32 joined, bystanders;
    u32 arrive, gen;    // barrier
    u32 last_level;     // first empty level (out)
    u32 pad[10];
};

constexpr int kPGroupsPerWg = 64;           // 1024 threads
constexpr int kPGroup = 6;                  // 6*5/2 = 15 pairs <= 16 lanes

__device__ __forceinline__ u32 xcc_id() {
#ifdef C2A_EMULATE
    return 0;
#else
    return __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 0xFu;      // HW_REG_XCC_ID[3:0]
#endif
}

// LDS hand-off between the lanes of one 16-lane group (group-uniform control flow): DS ops of a wave execute in
// order, so only the compiler has to be stopped; the emulation needs a real rendezvous of the group's fibers
__device__ __forceinline__ void group_lds_sync() {
#ifdef C2A_EMULATE
    (void)__shfl(0u, 0, 16);
#else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

__device__ __forceinline__ u32 group_or16(u32 v) {
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) v |= __shfl_xor(v, off, 16);
    return v;
}
__device__ __forceinline__ u32 group_min16(u32 v) {
#pragma unroll
    for (int off = 8; off >= 1; off >>= 1) { const u32 o = __shfl_xor(v, off, 16); v = o < v ? o : v; }
    return v;
}

// DS = true: bulk data (slots, candidate records, meta, ancestor rows) by sc1 accesses, no fences.
// DS = false: bulk data by plain accesses + one agent-scope release and acquire per workgroup per level (G16 recipe).
template <bool DS>
__global__ void __launch_bounds__(kPGroupsPerWg * 16) k_peel_persistent(PeelArgs A, u32 level0, PeelCtl* ctl) {
    __shared__ u32 s_c[kPGroupsPerWg][20], s_l[kPGroupsPerWg][20], s_d[kPGroupsPerWg][20];
    __shared__ u32 s_ready[2 * kPGroupsPerWg];
    __shared__ uint4 s_rec[2 * kPGroupsPerWg];
    __shared__ u32 s_cnt[2];
    __shared__ u32 s_base, s_P, s_rank, s_go;
    const u32 tid = threadIdx.x, lane = tid & 63u, l16 = tid & 15u, gq = tid >> 4;
    // ---- who works: the XCD of the first workgroup to arrive
    if (tid == 0) {
        const u32 x = xcc_id();
        const u32 prev = atomicCAS(&ctl->chosen_xcd, 0xFFFFFFFFu, x);
        const bool mine = prev == 0xFFFFFFFFu || prev == x;
        s_go = mine ? 1u : 0u;
        if (mine) s_rank = atomicAdd(&ctl->joined, 1u); else atomicAdd(&ctl->bystanders, 1u);
    }
    __syncthreads();
    if (!s_go) return;
    if (tid == 0) {
        while (ld_u32<true>(&ctl->joined) + ld_u32<true>(&ctl->bystanders) < gridDim.x) __builtin_amdgcn_s_sleep(2);
        s_P = ld_u32<true>(&ctl->joined);
    }
    __syncthreads();
    const u32 P = s_P, rank = s_rank;
    const u32 total_groups = P * kPGroupsPerWg;
    const u64 plane = (u64)A.n * 16;
    u32 epoch = 0;
    // diagnostics: per workgroup, per level (first 64 levels of the launch), wall-clock ticks (10 ns) at the phase ends
#define C2A_PPROF(slot)                                                                                        \
    do {                                                                                                       \
        if (A.prof && tid == 0 && level >= A.prof_level0 && level - A.prof_level0 < 64) A.prof[(((u64)(level - A.prof_level0)) * 64 + rank) * 8 + (slot)] = c2a_now(); \
    } while (0)
    for (u32 level = level0;; ++level) {
        C2A_PPROF(0);
        const u32 n_front = ld_u32<true>(&A.fcount[level]);
        if (A.prof && tid == 0 && rank == 0 && level >= A.prof_level0 && level - A.prof_level0 < 64) A.prof[(((u64)(level - A.prof_level0)) * 64 + 63) * 8 + 7] = n_front;
        if (n_front == 0) {                               // every participant reads the same value: uniform exit
            if (rank == 0 && tid == 0) st_u32<true>(&ctl->last_level, level);
            break;
        }
        const u32 lo = ld_u32<true>(&A.fbase[level]);
        if (rank == 0 && tid == 0) st_u32<true>(&A.fbase[level + 1], lo + n_front);
        FrontierSlot* cur = A.slots[level & 1u];
        FrontierSlot* nxt = A.slots[(level + 1) & 1u];
        for (u32 base = 0; base < n_front; base += total_groups) {
            const u32 i = base + rank * kPGroupsPerWg + gq;
            u32 rdy = C2A_NONE;
            uint4 rdy_rec = make_uint4(0, 0, 0, 0);
            if (i < n_front) {
                const uint4 sa = ld_u128<DS>(&cur[i].a);
                const u32 cnt = ld_u32<DS>(&cur[i].b.x);
                const u32 g = sa.x;
                const u32 e0 = sa.w, e1 = e0 + cnt;
                const u32 dl = l16 == 0 ? sa.y : (l16 == 1 ? sa.z : C2A_NONE);
                uint4 gd = make_uint4(0, 0, 0, 0);
                if (dl != C2A_NONE) gd = A.ginfo[dl];                 // static (written before this launch)
                u32 ch = C2A_NONE, ch_el = 0, ch_root = g, ch_depth = 0;
                for (u32 eb = e0; eb < e1; eb += 16) {
                    const u32 e = eb + l16;
                    const bool valid = e < e1;
                    u32 c = 0, l = 0, cdepth = 0, croot = 0xFFFFFFFFu;
                    if (valid) {
                        const uint4 cr = ld_u128<DS>(&A.cand[e]);
                        c = (cr.x & kIdMask) | (cr.w << 31); l = cr.x >> 31;
                        cdepth = cr.y; croot = cr.z;
                    }
                    const u32 rmin = group_min16(croot);
                    if (rmin > ch_root) continue;
                    const bool keep_ch = (ch != C2A_NONE) && (ch_root == rmin);
                    const bool surv = valid && croot == rmin;
                    const u32 smask = group_or16(surv ? (1u << l16) : 0u);
                    u32 m = (u32)__popc(smask);
                    if (surv) {
                        const u32 k = (u32)__popc(smask & ((1u << l16) - 1u));
                        s_c[gq][k] = c; s_l[gq][k] = l; s_d[gq][k] = cdepth;
                    }
                    if (keep_ch && l16 == 0) { s_c[gq][m] = ch; s_l[gq][m] = ch_el; s_d[gq][m] = ch_depth; }
                    m += keep_ch ? 1u : 0u;
                    group_lds_sync();
                    u32 win = 0, next = 1;
                    while (next < m) {
                        const u32 take = (m - next) < (u32)(kPGroup - 1) ? (m - next) : (u32)(kPGroup - 1);
                        const u32 q = take + 1;
                        const u32 NP = q * (q - 1) / 2;
                        u32 pi = 0, pj = 1;
                        {
                            u32 rem = l16, row = 0, len = q - 1;
                            while (len && rem >= len) { rem -= len; ++row; --len; }
                            pi = row; pj = row + 1 + rem;
                        }
                        u32 lost_bit = 0;
                        if (l16 < NP) {
                            const u32 xi = pi == 0 ? win : next + pi - 1, xj = next + pj - 1;
                            const u32 ci = s_c[gq][xi], cj = s_c[gq][xj];
                            const u32 li = s_l[gq][xi], lj = s_l[gq][xj];
                            bool less;
                            if (((ci ^ cj) & kIdMask) == 0) less = li < lj;
                            else less = path_less<DS, 4>(A.anc, plane, ci, li, s_d[gq][xi], cj, lj, s_d[gq][xj]);
                            lost_bit = 1u << (less ? pj : pi);
                        }
                        const u32 lost = group_or16(lost_bit);
                        const u32 w = (u32)__ffs((int)(~lost & ((1u << q) - 1u))) - 1u;    // the one member that never lost
                        win = w == 0 ? win : next + w - 1;
                        next += take;
                    }
                    ch = s_c[gq][win]; ch_el = s_l[gq][win]; ch_depth = s_d[gq][win]; ch_root = rmin;
                    group_lds_sync();
                }
                const u32 depth = ch == C2A_NONE ? 0u : ch_depth + 1;
                const u32 my_label = ch == C2A_NONE ? 0u : ch_el;
                const u32 pos = lo + i;
                if (l16 == 0) {
                    st_u128<DS>(&A.meta[pos], make_uint4(ch == C2A_NONE ? C2A_NONE : (ch & kIdMask), depth, ch_root, my_label));
                    st_u32<DS>(&A.order[pos], g);
                    st_u32<DS>(&A.posof[g], pos);
                    if (ch != C2A_NONE) st_u32<DS>(&A.child[2 * (u64)(ch & kIdMask) + my_label], pos);
                }
                u32 kfill = 0;
                if (dl != C2A_NONE) kfill = atomicAdd(&A.fill[dl], 1u);
                if (depth) {
                    u32 q = ch, need = 1;
                    for (int j = 0; need <= depth; ++j) {
                        u32 v = q;
                        if (l16 >= 1) v = anc_entry<DS, 4>(A.anc, plane, j, q & kIdMask, l16 - 1);
                        st_u32<DS>(&A.anc[(u64)j * plane + (u64)pos * 16 + l16], v);
                        q = __shfl(v, 15, 16);
                        if (need > (0xFFFFFFFFu >> 4)) break;
                        need <<= 4;
                    }
                }
                if (dl != C2A_NONE) {
                    st_u128<DS>(&A.cand[gd.z + kfill], make_uint4(pos | (l16 << 31), depth, ch_root, my_label));   // l16 == edge label
                    if (kfill + 1 == gd.w) { rdy = dl; rdy_rec = gd; }
                }
            }
            C2A_PPROF(1);
            // ---- one append per workgroup
            if (l16 < 2) { s_ready[2 * gq + l16] = rdy; s_rec[2 * gq + l16] = rdy_rec; }
            __syncthreads();
            C2A_PPROF(2);
            u32 d = C2A_NONE;
            u64 mask = 0;
            if (tid < 2 * kPGroupsPerWg) {                           // waves 0 and 1, whole waves
                d = s_ready[tid];
                mask = __ballot(d != C2A_NONE);
                if (lane == 0) s_cnt[tid >> 6] = (u32)__popcll(mask);
            }
            __syncthreads();
            if (tid == 0) { const u32 tot = s_cnt[0] + s_cnt[1]; s_base = tot ? atomicAdd(&A.fcount[level + 1], tot) : 0u; }
            __syncthreads();
            if (d != C2A_NONE) {
                const u32 p = s_base + (tid >= 64 ? s_cnt[0] : 0u) + (u32)__popcll(mask & ((1ull << lane) - 1ull));
                const uint4 gd = s_rec[tid];
                st_u32<DS>(&nxt[p].b.x, gd.w);
                st_u128<DS>(&nxt[p].a, make_uint4(d, gd.x, gd.y, gd.z));
            }
            __syncthreads();
            C2A_PPROF(3);
        }
        // ---- level barrier: drain every wave's write-through stores, then arrive / wait
#ifndef C2A_EMULATE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __syncthreads();
        C2A_PPROF(4);
        ++epoch;
        if (tid == 0) {
#ifndef C2A_EMULATE
            if (!DS) {      // publish this workgroup's plain stores (the asm wait restates the post-wbl2 wait: G16 pitfall 12)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
#endif
            const u32 a = atomicAdd(&ctl->arrive, 1u);
            if (a == P * epoch - 1) st_u32<true>(&ctl->gen, epoch);
            while (ld_u32<true>(&ctl->gen) < epoch) __builtin_amdgcn_s_sleep(1);
#ifndef C2A_EMULATE
            if (!DS) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");      // one lane's acquire + the barrier below covers the workgroup
#endif
        }
        __syncthreads();
        C2A_PPROF(5);
    }
}

// ------------------------------------------------------------------------------------------------
// Variant 4: the path-string peel as ONE persistent launch on ONE XCD (narrow frontiers).
// What a kernel boundary per level costs (tools/ubench/xcd2.hip, MI355X): ~2.3 us of launch gap, an L2 invalidate,
// and every first touch afterwards from HBM (220-370 ns per dependent hop instead of 85 ns in L2).  The 32 workgroups
// of one XCD share ONE L2, so inside one launch they can exchange data through it with no cache maintenance at all:
//   stores  : plain (the vector L1 is write-through; vmcnt(0) == the L2 has the data)
//   loads   : sc1 (agent-scope relaxed atomic loads: miss the L1 always, hit the shared L2)       [0 stale reads / 16 M]
//   atomics : L2 atomics (fill[], the frontier counter, the level barrier) — ~12 ns each on one word, 32 per level
//   level barrier among the 32 workgroups: 1.05-1.36 us; `buffer_inv sc1` (7 us) is never used
// The XCD filter only decides WHO works (HW_REG_XCC_ID is ground truth; the first workgroup to arrive picks the XCD;
// the census tells the participants how many they are); correctness never depends on placement.
// One 16-lane group per gate (64 gates per workgroup, 2048 per pass over the 32 CUs): lane l16 holds words
// {l16, 16+l16, 32+l16, 48+l16} of a string, so every 8-byte load instruction of a group covers 128 contiguous bytes
// and the k-th one is skipped for strings shorter than 1024k bits.
// The launch returns at the first level that is empty or wider than `cap` (the host goes on with a launch per level).
// ------------------------------------------------------------------------------------------------
constexpr int kSGroupsPerWg = 64;           // 1024 threads
constexpr int kSMax = 5;                    // survivor strings held in registers per tournament round
constexpr int kSW = 4;                      // u64 words of a string per lane

__device__ __forceinline__ u32 group_ballot16(bool p, u32 l16, u32 gshift) {
#ifdef C2A_EMULATE
    (void)gshift;
    return group_or16(p ? (1u << l16) : 0u);
#else
    (void)l16;
    return (u32)((__ballot(p) >> gshift) & 0xFFFFull);
#endif
}

template <bool SC1>
__device__ __forceinline__ void load_str16(const u64* pstr, u32 x, u32 len, u32 l16, u64 (&w)[kSW]) {
#pragma unroll
    for (int k = 0; k < kSW; ++k) {
        const u32 word = (u32)k * 16 + l16;
        w[k] = word * 64 < len ? ld_u64<SC1>(pstr + (u64)x * kChunkWords + word) : 0ull;
    }
}

// bit `pos` of a string held by a group (group-uniform pos)
__device__ __forceinline__ u32 str_bit16(const u64 (&w)[kSW], u32 pos) {
    const u32 word = pos >> 6, kk = word >> 4;
    u64 v = 0;
#pragma unroll
    for (int k = 0; k < kSW; ++k) {
        const u64 t = __shfl(w[k], (int)(word & 15u), 16);
        if ((u32)k == kk) v = t;
    }
    return (u32)((v >> (pos & 63u)) & 1ull);
}

// group-cooperative: is P(a).la < P(b).lb for two strings of one chunk held in registers?  (group-uniform result)
__device__ __forceinline__ bool str_less16(const u64 (&a)[kSW], u32 lena, u32 la, const u64 (&b)[kSW], u32 lenb, u32 lb, u32 l16,
                                           u32 gshift) {
    const u32 minlen = lena < lenb ? lena : lenb;
#pragma unroll
    for (int k = 0; k < kSW; ++k) {
        const u32 lo = ((u32)k * 16 + l16) * 64;
        u64 x = a[k] ^ b[k];
        if (lo >= minlen) x = 0;
        else if (minlen - lo < 64) x &= (1ull << (minlen - lo)) - 1ull;
        const u32 bal = group_ballot16(x != 0, l16, gshift);
        if (bal) {
            const int L = (int)__ffs((int)bal) - 1;
            const u64 xl = __shfl(x, L, 16);
            const u64 al = __shfl(a[k], L, 16);
            return ((al >> ctz64(xl)) & 1ull) == 0;
        }
    }
    if (lena == lenb) return la < lb;
    if (lena < lenb) return la < str_bit16(b, lena);
    return str_bit16(a, lenb) < lb;
}

__global__ void __launch_bounds__(kSGroupsPerWg * 16) k_peel_persistent_str(PeelArgs A, u32 level0, u32 cap, PeelCtl* ctl) {
    __shared__ u32 s_c[kSGroupsPerWg][20], s_l[kSGroupsPerWg][20], s_d[kSGroupsPerWg][20];
    __shared__ u32 s_ready[2 * kSGroupsPerWg];
    __shared__ uint4 s_rec[2 * kSGroupsPerWg];
    __shared__ u32 s_cnt[2];
    __shared__ u32 s_base, s_P, s_rank, s_go, s_nf;
    const u32 tid = threadIdx.x, lane = tid & 63u, l16 = tid & 15u, gq = tid >> 4, gshift = lane & 48u;
    // ---- who works: the XCD of the first workgroup to arrive
    if (tid == 0) {
        const u32 x = xcc_id();
        const u32 prev = atomicCAS(&ctl->chosen_xcd, 0xFFFFFFFFu, x);
        const bool mine = prev == 0xFFFFFFFFu || prev == x;
        s_go = mine ? 1u : 0u;
        if (mine) s_rank = atomicAdd(&ctl->joined, 1u); else atomicAdd(&ctl->bystanders, 1u);
    }
    __syncthreads();
    if (!s_go) return;
    if (tid == 0) {
        while (ld_u32<true>(&ctl->joined) + ld_u32<true>(&ctl->bystanders) < gridDim.x) __builtin_amdgcn_s_sleep(2);
        s_P = ld_u32<true>(&ctl->joined);
    }
    __syncthreads();
    const u32 P = s_P, rank = s_rank;
    const u32 total_groups = P * kSGroupsPerWg;
    u32 epoch = 0;
    u32 pf_acc = 0;                                        // sink of the look-ahead loads (never true, keeps them alive)
    // one lane per workgroup reads the level's frontier size (512 waves on one word is a 0.7 us pile-up on one L2 channel)
    if (tid == 0) s_nf = ld_u32<true>(&A.fcount[level0]);
    u32 lo = ld_u32<true>(&A.fbase[level0]);
    __syncthreads();
    for (u32 level = level0;; ++level) {
        C2A_PPROF(0);
        const u32 n_front = s_nf;
        if (n_front == 0 || n_front > cap) {               // every participant reads the same value: uniform exit
            if (rank == 0 && tid == 0) st_u32<true>(&ctl->last_level, level);
            if (pf_acc == 0x9E3779B9u && n_front == 0xFFFFFFFFu) ctl->pad[0] = pf_acc;
            break;
        }
        if (rank == 0 && tid == 0) A.fbase[level + 1] = lo + n_front;
        FrontierSlot* cur = A.slots[level & 1u];
        FrontierSlot* nxt = A.slots[(level + 1) & 1u];
        for (u32 base = 0; base < n_front; base += total_groups) {
            const u32 i = base + rank * kSGroupsPerWg + gq;
            const bool active = i < n_front;
            // ---- slot, then (in one round trip) candidate records, the producers' ginfo and the two push tickets
            u32 sword = 0;                                            // lanes 0-3: a.x..a.w, lane 4: b.x — one 4-byte load per lane
            if (active && l16 < 5) sword = ld_u32<true>(reinterpret_cast<const u32*>(&cur[i]) + l16);
            const uint4 sa = make_uint4(__shfl(sword, 0, 16), __shfl(sword, 1, 16), __shfl(sword, 2, 16), __shfl(sword, 3, 16));
            const u32 cnt = __shfl(sword, 4, 16);
            const u32 g = sa.x;
            const u32 e0 = sa.w, e1 = e0 + cnt;
            const u32 dl = active ? (l16 == 0 ? sa.y : (l16 == 1 ? sa.z : C2A_NONE)) : C2A_NONE;
            uint4 gd = make_uint4(0, 0, 0, 0);
            u32 kfill = 0;
            uint4 cr_first = make_uint4(0, 0, 0xFFFFFFFFu, 0);
            if (active && e0 + l16 < e1) cr_first = ld_u128<true>(&A.cand[e0 + l16]);
            if (dl != C2A_NONE) { gd = A.ginfo[dl]; kfill = atomicAdd(&A.fill[dl], 1u); }          // ginfo is static
            const bool last_push = dl != C2A_NONE && kfill + 1 == gd.w;
            if (l16 < 2) { s_ready[2 * gq + l16] = last_push ? dl : C2A_NONE; s_rec[2 * gq + l16] = gd; }
            C2A_PPROF(1);
            __syncthreads();
            // ---- one append per workgroup: the ticket is issued now, the slots are written after the tournament
            u32 d_app = C2A_NONE;
            u64 mask_app = 0;
            if (tid < 2 * kSGroupsPerWg) {                           // waves 0 and 1, whole waves
                d_app = s_ready[tid];
                mask_app = __ballot(d_app != C2A_NONE);
                if (lane == 0) s_cnt[tid >> 6] = (u32)__popcll(mask_app);
            }
            __syncthreads();
            C2A_PPROF(2);
            u32 ticket = 0;                                          // consumed after the tournament: the round trip overlaps it
            if (tid == 0) { const u32 tot = s_cnt[0] + s_cnt[1]; ticket = tot ? atomicAdd(&A.fcount[level + 1], tot) : 0u; }
            if (d_app != C2A_NONE) {
                // look-ahead for the next level: pull ginfo[] and fill[] of the appended gate's two producers into the L2
                // now (random lines, HBM), so that next level's tickets find them there
                const uint4 r = s_rec[tid];
                if (r.x != C2A_NONE) pf_acc ^= A.ginfo[r.x].w ^ ld_u32<true>(&A.fill[r.x]);
                if (r.y != C2A_NONE) pf_acc ^= A.ginfo[r.y].w ^ ld_u32<true>(&A.fill[r.y]);
            }
            if (active) {
                u32 ch = C2A_NONE, ch_el = 0, ch_root = g, ch_depth = 0;
                u64 champ[kSW] = {0, 0, 0, 0};
                bool champ_loaded = false;
                for (u32 eb = e0; eb < e1; eb += 16) {
                    const u32 e = eb + l16;
                    const bool valid = e < e1;
                    uint4 cr = cr_first;
                    if (eb != e0) { cr = make_uint4(0, 0, 0xFFFFFFFFu, 0); if (valid) cr = ld_u128<true>(&A.cand[e]); }
                    const u32 c = cr.x & kIdMask, l = cr.x >> 31, cdepth = cr.y;
                    const u32 croot = valid ? cr.z : 0xFFFFFFFFu;
                    const u32 rmin = group_min16(croot);
                    if (rmin > ch_root) continue;
                    const bool keep_ch = (ch != C2A_NONE) && (ch_root == rmin);
                    if (!keep_ch) champ_loaded = false;
                    const bool surv = valid && croot == rmin;
                    const u32 smask = group_ballot16(surv, l16, gshift);
                    const u32 m = (u32)__popc(smask);
                    if (surv) {
                        const u32 k = (u32)__popc(smask & ((1u << l16) - 1u));
                        s_c[gq][k] = c; s_l[gq][k] = l; s_d[gq][k] = cdepth;
                    }
                    group_lds_sync();
                    C2A_PPROF(3);
                    u32 next = 0;
                    if (!keep_ch) { ch = s_c[gq][0]; ch_el = s_l[gq][0]; ch_depth = s_d[gq][0]; next = 1; }
                    ch_root = rmin;
                    while (next < m) {
                        const u32 take = (m - next) < (u32)kSMax ? (m - next) : (u32)kSMax;
                        bool shallow = ch_depth <= kChunkBits;
                        for (u32 t = 0; t < take; ++t) shallow = shallow && s_d[gq][next + t] <= kChunkBits;
                        if (shallow) {
                            u64 sw[kSMax][kSW];
#pragma unroll
                            for (int t = 0; t < kSMax; ++t) {
                                const bool on = (u32)t < take;
                                load_str16<true>(A.pstr, on ? s_c[gq][next + t] : 0u, on ? s_d[gq][next + t] : 0u, l16, sw[t]);
                            }
                            if (!champ_loaded) { load_str16<true>(A.pstr, ch, ch_depth, l16, champ); champ_loaded = true; }
                            if (A.prof && (sw[0][0] ^ champ[0]) == 0x123456789ABCDEFull) return;
                            C2A_PPROF(4);
#pragma unroll
                            for (int t = 0; t < kSMax; ++t) {
                                if ((u32)t < take) {
                                    const u32 cc = s_c[gq][next + t], cl = s_l[gq][next + t], cd = s_d[gq][next + t];
                                    bool less;
                                    if (cc == ch) less = cl < ch_el;
                                    else less = str_less16(sw[t], cd, cl, champ, ch_depth, ch_el, l16, gshift);
                                    if (less) {
                                        ch = cc; ch_el = cl; ch_depth = cd;
#pragma unroll
                                        for (int k = 0; k < kSW; ++k) champ[k] = sw[t][k];
                                    }
                                }
                            }
                        } else {
                            // deep trees: chunk resolution (cprev hops) and string loads per comparison
                            for (u32 t = 0; t < take; ++t) {
                                const u32 cc = s_c[gq][next + t], cl = s_l[gq][next + t], cd = s_d[gq][next + t];
                                bool less;
                                if (cc == ch) less = cl < ch_el;
                                else {
                                    u32 ra = cc, rb = ch, lena, lenb, ba, bb;
                                    resolve_chunks<true>(A.cprev, ra, lena, ba, cd, rb, lenb, bb, ch_depth);
                                    if (ra == rb) {
                                        if (ba != C2A_NONE) less = (u32)(ld_u64<true>(A.pstr + (u64)ba * kChunkWords) & 1ull) < ch_el;
                                        else less = cl < (u32)(ld_u64<true>(A.pstr + (u64)bb * kChunkWords) & 1ull);
                                    } else {
                                        u64 wa[kSW], wb[kSW];
                                        load_str16<true>(A.pstr, ra, lena, l16, wa);
                                        load_str16<true>(A.pstr, rb, lenb, l16, wb);
                                        less = str_less16(wa, lena, cl, wb, lenb, ch_el, l16, gshift);
                                    }
                                }
                                if (less) { ch = cc; ch_el = cl; ch_depth = cd; champ_loaded = false; }
                            }
                        }
                        next += take;
                    }
                    group_lds_sync();
                }
                C2A_PPROF(5);
                const u32 depth = ch == C2A_NONE ? 0u : ch_depth + 1;
                const u32 my_label = ch == C2A_NONE ? 0u : ch_el;
                const u32 pos = lo + i;
                if (l16 == 0) {
                    A.meta[pos] = make_uint4(ch, depth, ch_root, my_label);
                    A.order[pos] = g;
                    A.posof[g] = pos;
                    if (ch != C2A_NONE) A.child[2 * (u64)ch + my_label] = pos;
                }
                if (dl != C2A_NONE) A.cand[gd.z + kfill] = make_uint4(pos | (l16 << 31), depth, ch_root, my_label);   // l16 == edge label
                if (ch != C2A_NONE) {
                    const bool need_parent = ch_depth != 0 && chunk_of(depth) == chunk_of(ch_depth);
                    if (need_parent && !champ_loaded) load_str16<true>(A.pstr, ch, chunk_len(ch_depth), l16, champ);
                    const u32 len = chunk_len(depth);
                    bool fresh = false;
#pragma unroll
                    for (int k = 0; k < kSW; ++k) {
                        const u32 word = (u32)k * 16 + l16;
                        const u64 v = child_word(need_parent ? champ[k] : 0ull, ch_depth, my_label, word, fresh);
                        if (word * 64 < len) A.pstr[(u64)pos * kChunkWords + word] = v;
                    }
                    if (l16 == 0 && chunk_of(depth)) A.cprev[pos] = fresh ? ch : ld_u32<true>(&A.cprev[ch]);
                }
            }
            C2A_PPROF(6);
            if (tid == 0) s_base = ticket;
            __syncthreads();
            if (d_app != C2A_NONE) {
                const u32 p = s_base + (tid >= 64 ? s_cnt[0] : 0u) + (u32)__popcll(mask_app & ((1ull << lane) - 1ull));
                const uint4 r = s_rec[tid];
                nxt[p].b = make_uint4(r.w, 0, 0, 0);
                nxt[p].a = make_uint4(d_app, r.x, r.y, r.z);
            }
            __syncthreads();
        }
        // ---- level barrier: every wave's (write-through) stores are in the L2, then arrive / wait
#ifndef C2A_EMULATE
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        __syncthreads();
        ++epoch;
        if (tid == 0) {
            const u32 a = atomicAdd(&ctl->arrive, 1u);
            if (a == P * epoch - 1) st_u32<true>(&ctl->gen, epoch);
            while (ld_u32<true>(&ctl->gen) < epoch) __builtin_amdgcn_s_sleep(1);
            s_nf = ld_u32<true>(&A.fcount[level + 1]);
        }
        lo += n_front;
        __syncthreads();
        C2A_PPROF(7);
    }
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
            const u32 s1 = m.w == 0 ? child[2 * (u64)m.x + 1] : C2A_NONE;
            nx = s1 != C2A_NONE ? 2 * s1 : 2 * m.x + 1;
        }
        next[2 * i + 1] = nx;
    }
}

__device__ __forceinline__ bool is_splitter(u32 e, u32 head) { return e == head || ((e * 0x9E3779B1u) >> 26) == 0u; }

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
__global__ void k_first_seen(u32 n, const u32* __restrict__ sorted, const uint4* __restrict__ gate4, u32* first) {
    for (u64 pos = gtid(); pos < n; pos += gstride()) {
        const uint4 g = gate4[sorted[pos]];
        const u32 i = 3u * (u32)pos;
        atomicMin(&first[g.x], i);
        atomicMin(&first[g.y], i + 1);
        atomicMin(&first[g.z], i + 2);
    }
}

__global__ void k_new_wire_flags(u32 n, const u32* __restrict__ sorted, const uint4* __restrict__ gate4,
                                 const u32* __restrict__ first, const u8* __restrict__ nflag, u32* flag) {
    for (u64 pos = gtid(); pos < n; pos += gstride()) {
        const uint4 g = gate4[sorted[pos]];
        const u32 i = 3u * (u32)pos;
        flag[i] = (first[g.x] == i && nflag[g.x] == 0) ? 1u : 0u;                 // :431-438
        flag[i + 1] = (first[g.y] == i + 1 && nflag[g.y] == 0) ? 1u : 0u;
        flag[i + 2] = (first[g.z] == i + 2 && nflag[g.z] == 0) ? 1u : 0u;
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

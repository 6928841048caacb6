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

// CSR of consumers: cons[cons_off[d] ..] = (consumer gate << 1) | label
__global__ void k_fill_csr(u32 n, const u32* __restrict__ dep0, const u32* __restrict__ dep1,
                           const u32* __restrict__ cons_off, u32* fill, u32* cons) {
    for (u64 g = gtid(); g < n; g += gstride()) {
        const u32 d0 = dep0[g], d1 = dep1[g];
        if (d0 != C2A_NONE) cons[cons_off[d0] + atomicAdd(&fill[d0], 1u)] = ((u32)g << 1);
        if (d1 != C2A_NONE) cons[cons_off[d1] + atomicAdd(&fill[d1], 1u)] = ((u32)g << 1) | 1u;
    }
}

// level 0 of the reverse Kahn peel: gates nobody consumes.  fcount[0] must be zero.
__global__ void k_init_frontier(u32 n, const u32* __restrict__ cons_cnt, u32* pending, u32* order, u32* posof,
                                u32* fcount) {
    for (u64 g = gtid(); g < n; g += gstride()) {
        const u32 c = cons_cnt[g];
        pending[g] = c;
        if (c == 0) {
            const u32 p = atomicAdd(&fcount[0], 1u);
            order[p] = (u32)g;
            posof[g] = p;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// peel one level + pick DFS-tree parents
// ------------------------------------------------------------------------------------------------
// tree node == peel position `pos` (index into order[]).  meta[pos] = {parent pos | NONE, depth, root
// gate id, label}.  anc plane j, row pos: 16 ancestors at distances (d+1)*16^j, valid while <= depth.
struct PeelArgs {
    u32 n;
    const u32* dep0;
    const u32* dep1;
    const u32* cons_off;
    const u32* cons;
    u32* pending;
    u32* order;
    u32* posof;
    uint4* meta;
    u32* anc;        // [planes][n][16]
    u32* fbase;      // [levels+2]
    u32* fcount;     // [levels+2]
    u32* maxdepth;   // running max tree depth (stat)
    u32* levels;     // number of non-empty levels (stat)
};

__device__ __forceinline__ u32 anc_entry(const u32* anc, u64 plane, int j, u32 x, u32 d) {
    return anc[(u64)j * plane + (u64)x * 16 + d];
}

__device__ __forceinline__ u32 level_anc(const u32* anc, u64 plane, u32 x, u32 dist) {
    int j = 0;
    while (dist) {
        const u32 d = dist & 15u;
        if (d) x = anc_entry(anc, plane, j, x, d - 1);
        dist >>= 4;
        ++j;
    }
    return x;
}

__device__ __forceinline__ void load_row(const u32* anc, u64 plane, int j, u32 x, u32 (&r)[16]) {
    const uint4* p = reinterpret_cast<const uint4*>(anc + (u64)j * plane + (u64)x * 16);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint4 v = p[q];
        r[4 * q + 0] = v.x; r[4 * q + 1] = v.y; r[4 * q + 2] = v.z; r[4 * q + 3] = v.w;
    }
}

// a != b at equal depth D >= 1 under one root: walk both up to the children of their lowest common
// ancestor (one 64-byte row per node per base-16 digit).
__device__ __forceinline__ void diverge(const u32* anc, u64 plane, u32& a, u32& b, u32 D) {
    if (D == 0) return;
    int j = (31 - __clz(D)) >> 2;
    for (; j >= 0; --j) {
        u32 m = D >> (4 * j);
        if (m == 0) continue;
        if (m > 16) m = 16;
        u32 ra[16], rb[16];
        load_row(anc, plane, j, a, ra);
        load_row(anc, plane, j, b, rb);
        u32 pa = a, pb = b, pd = 0;
#pragma unroll
        for (u32 d = 0; d < 16; ++d) {
            if (d < m && ra[d] != rb[d]) { pa = ra[d]; pb = rb[d]; pd = d + 1; }
        }
        a = pa; b = pb;
        D -= pd << (4 * j);
    }
}

// is P(a).la < P(b).lb ?  a != b, same root; da/db = depths.
__device__ __forceinline__ bool path_less(const u32* anc, u64 plane, const uint4* meta, u32 a, u32 la, u32 da, u32 b,
                                          u32 lb, u32 db) {
    if (da > db) {
        const u32 a1 = level_anc(anc, plane, a, da - db - 1);
        const uint4 m1 = meta[a1];
        if (m1.x == b) return m1.w < lb;   // b is an ancestor of a
        a = m1.x;
        diverge(anc, plane, a, b, db);
    } else if (db > da) {
        const u32 b1 = level_anc(anc, plane, b, db - da - 1);
        const uint4 m1 = meta[b1];
        if (m1.x == a) return la < m1.w;   // a is an ancestor of b
        b = m1.x;
        diverge(anc, plane, a, b, da);
    } else {
        diverge(anc, plane, a, b, da);
    }
    return meta[a].w < meta[b].w;
}

__global__ void __launch_bounds__(kThreads) k_peel_level(PeelArgs A, u32 level) {
    const u32 lo = A.fbase[level];
    const u32 cnt = A.fcount[level];
    const u32 next_base = lo + cnt;
    if (gtid() == 0) {
        A.fbase[level + 1] = next_base;
        if (cnt) atomicMax(A.levels, level + 1);
    }
    const u64 plane = (u64)A.n * 16;
    u32 local_maxdepth = 0;
    for (u64 i = gtid(); i < cnt; i += gstride()) {
        const u32 pos = lo + (u32)i;
        const u32 g = A.order[pos];
        // ---- tournament over the candidate paths: [g] (child of the virtual root) and P(c).l per consumer
        u32 best = C2A_NONE, best_label = 0, best_root = g, best_depth = 0;
        const u32 e0 = A.cons_off[g], e1 = A.cons_off[g + 1];
        for (u32 e = e0; e < e1; ++e) {
            const u32 ce = A.cons[e];
            const u32 c = ce >> 1, l = ce & 1u;
            const u32 pc = A.posof[c];
            const uint4 mc = A.meta[pc];
            bool take;
            if (best == C2A_NONE) take = mc.z < g;
            else if (mc.z != best_root) take = mc.z < best_root;
            else if (pc == best) take = l < best_label;
            else take = path_less(A.anc, plane, A.meta, pc, l, mc.y, best, best_label, best_depth);
            if (take) { best = pc; best_label = l; best_root = mc.z; best_depth = mc.y; }
        }
        const u32 depth = best == C2A_NONE ? 0u : best_depth + 1;
        A.meta[pos] = make_uint4(best, depth, best_root, best == C2A_NONE ? 0u : best_label);
        if (depth > local_maxdepth) local_maxdepth = depth;
        // ---- ancestor rows: row j = [q_j, row_j(q_j)[0..14]], q_0 = parent, q_{j+1} = my ancestor at 16^(j+1)
        if (depth) {
            u32 q = best;
            u32 need = 1;   // 16^j
            for (int j = 0; need <= depth; ++j) {
                u32 r[16];
                load_row(A.anc, plane, j, q, r);
                uint4* dst = reinterpret_cast<uint4*>(A.anc + (u64)j * plane + (u64)pos * 16);
                dst[0] = make_uint4(q, r[0], r[1], r[2]);
                dst[1] = make_uint4(r[3], r[4], r[5], r[6]);
                dst[2] = make_uint4(r[7], r[8], r[9], r[10]);
                dst[3] = make_uint4(r[11], r[12], r[13], r[14]);
                q = r[14];
                if (need > (0xFFFFFFFFu >> 4)) break;
                need <<= 4;
            }
        }
        // ---- release producers: a gate joins the next frontier when its last consumer is peeled
        const u32 d0 = A.dep0[g], d1 = A.dep1[g];
        if (d0 != C2A_NONE && atomicSub(&A.pending[d0], 1u) == 1u) {
            const u32 p = next_base + atomicAdd(&A.fcount[level + 1], 1u);
            A.order[p] = d0;
            A.posof[d0] = p;
        }
        if (d1 != C2A_NONE && atomicSub(&A.pending[d1], 1u) == 1u) {
            const u32 p = next_base + atomicAdd(&A.fcount[level + 1], 1u);
            A.order[p] = d1;
            A.posof[d1] = p;
        }
    }
    if (local_maxdepth) atomicMax(A.maxdepth, local_maxdepth);
}

// ------------------------------------------------------------------------------------------------
// post-order numbering: Euler tour of the DFS tree + list ranking (random splitters)
// ------------------------------------------------------------------------------------------------
__global__ void k_children(u32 n, const uint4* __restrict__ meta, u32* child0, u32* child1) {
    for (u64 x = gtid(); x < n; x += gstride()) {
        const uint4 m = meta[x];
        if (m.x != C2A_NONE) (m.w ? child1 : child0)[m.x] = (u32)x;
    }
}

__global__ void k_rootflag(u32 n, const uint4* __restrict__ meta, const u32* __restrict__ posof, u32* rflag) {
    for (u64 g = gtid(); g < n; g += gstride()) rflag[g] = meta[posof[g]].x == C2A_NONE ? 1u : 0u;
}

__global__ void k_rootlist(u32 n, const u32* __restrict__ rflag, const u32* __restrict__ ridx,
                           const u32* __restrict__ posof, u32* rlist) {
    for (u64 g = gtid(); g < n; g += gstride())
        if (rflag[g]) rlist[ridx[g]] = posof[g];
}

// element 2x = enter(x), 2x+1 = exit(x); the tour visits label-0 child, label-1 child, then exits.
__global__ void k_euler_next(u32 n, const uint4* __restrict__ meta, const u32* __restrict__ order,
                             const u32* __restrict__ child0, const u32* __restrict__ child1,
                             const u32* __restrict__ ridx, const u32* __restrict__ rlist, u32 n_roots, u32* next) {
    for (u64 x = gtid(); x < n; x += gstride()) {
        const u32 c0 = child0[x], c1 = child1[x];
        next[2 * x] = c0 != C2A_NONE ? 2 * c0 : (c1 != C2A_NONE ? 2 * c1 : (u32)(2 * x + 1));
        const uint4 m = meta[x];
        u32 nx;
        if (m.x == C2A_NONE) {
            const u32 k = ridx[order[x]];
            nx = k + 1 < n_roots ? 2 * rlist[k + 1] : C2A_NONE;
        } else {
            const u32 s1 = child1[m.x];
            nx = (m.w == 0 && s1 != C2A_NONE) ? 2 * s1 : 2 * m.x + 1;
        }
        next[2 * x + 1] = nx;
    }
}

__device__ __forceinline__ bool is_splitter(u32 e, u32 head) { return e == head || ((e * 0x9E3779B1u) >> 26) == 0u; }

__global__ void k_rank_mark(u32 m, const u32* __restrict__ rlist, u32* scount, u32* slist, u32* owner) {
    const u32 head = 2 * rlist[0];
    for (u64 e = gtid(); e < m; e += gstride()) {
        if (is_splitter((u32)e, head)) {
            const u32 k = atomicAdd(scount, 1u);
            slist[k] = (u32)e;
            owner[e] = k;
        }
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

__device__ __forceinline__ u32 seq_node(u64 i, const u32* sorted, const u32* lh, const u32* rh, const u32* out) {
    const u64 pos = i / 3;
    const u32 k = (u32)(i - pos * 3);
    const u32 g = sorted[pos];
    return k == 0 ? lh[g] : (k == 1 ? rh[g] : out[g]);
}

// first[node] = first index in the walk "for gate in sorted: [lh, rh, out]" (compiler.rs:427-430)
__global__ void k_first_seen(u64 m, const u32* __restrict__ sorted, const u32* __restrict__ lh,
                             const u32* __restrict__ rh, const u32* __restrict__ out, u32* first) {
    for (u64 i = gtid(); i < m; i += gstride()) atomicMin(&first[seq_node(i, sorted, lh, rh, out)], (u32)i);
}

__global__ void k_new_wire_flags(u64 m, const u32* __restrict__ sorted, const u32* __restrict__ lh,
                                 const u32* __restrict__ rh, const u32* __restrict__ out, const u32* __restrict__ first,
                                 const u8* __restrict__ nflag, u32* flag) {
    for (u64 i = gtid(); i < m; i += gstride()) {
        const u32 node = seq_node(i, sorted, lh, rh, out);
        flag[i] = (first[node] == (u32)i && nflag[node] == 0) ? 1u : 0u;   // :431-438
    }
}

__global__ void k_assign_wires(u64 m, const u32* __restrict__ sorted, const u32* __restrict__ lh,
                               const u32* __restrict__ rh, const u32* __restrict__ out, const u32* __restrict__ flag,
                               const u32* __restrict__ idx, u32 n_in, u32* node_wire1) {
    for (u64 i = gtid(); i < m; i += gstride())
        if (flag[i]) node_wire1[seq_node(i, sorted, lh, rh, out)] = n_in + idx[i] + 1;   // :440-441
}

__global__ void k_assign_outputs(u32 n_out, const u32* __restrict__ out_nodes, u32 n_in, const u32* __restrict__ n_mid,
                                 u32* node_wire1) {
    const u32 base = n_in + *n_mid;
    for (u64 j = gtid(); j < n_out; j += gstride()) atomicMax(&node_wire1[out_nodes[j]], base + (u32)j + 1);   // :446-449
}

__global__ void k_emit(u32 n, const u32* __restrict__ sorted, const u32* __restrict__ lh, const u32* __restrict__ rh,
                       const u32* __restrict__ out, const u8* __restrict__ op, const u32* __restrict__ node_wire1,
                       u32* e_in0, u32* e_in1, u32* e_out, u8* e_op) {
    for (u64 pos = gtid(); pos < n; pos += gstride()) {
        const u32 g = sorted[pos];
        e_in0[pos] = node_wire1[lh[g]] - 1;
        e_in1[pos] = node_wire1[rh[g]] - 1;
        e_out[pos] = node_wire1[out[g]] - 1;
        e_op[pos] = op[g];
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
};

constexpr int kBoolChunk = 128;   // arithmetic gates per workgroup

__device__ __forceinline__ u32 bool_wire(u32 W, u32 bit, u32 width, u32 M, u64 out_base) {
    return W < M ? W * width + bit : (u32)(out_base + (u64)(W - M) * width + bit);
}

__global__ void __launch_bounds__(kThreads) k_boolify(BoolArgs A, const BoolTables* __restrict__ T) {
    __shared__ u64 s_goff[kBoolChunk + 1];
    __shared__ u64 s_aoff[kBoolChunk];
    __shared__ u32 s_in0[kBoolChunk], s_in1[kBoolChunk], s_out[kBoolChunk], s_top[kBoolChunk];
    const u64 p0 = (u64)blockIdx.x * kBoolChunk;
    const u32 cnt = (u32)((A.n - p0) < (u64)kBoolChunk ? (A.n - p0) : (u64)kBoolChunk);
    for (u32 i = threadIdx.x; i <= cnt; i += blockDim.x) s_goff[i] = A.goff[p0 + i];
    for (u32 i = threadIdx.x; i < cnt; i += blockDim.x) {
        s_aoff[i] = A.aoff[p0 + i];
        s_in0[i] = A.e_in0[p0 + i];
        s_in1[i] = A.e_in1[p0 + i];
        s_out[i] = A.e_out[p0 + i];
        s_top[i] = T->toff[A.e_op[p0 + i]];
    }
    __syncthreads();
    const u64 q0 = s_goff[0], q1 = s_goff[cnt];
    for (u64 q = q0 + threadIdx.x; q < q1; q += blockDim.x) {
        // largest i with s_goff[i] <= q
        u32 lo = 0, hi = cnt;
        while (hi - lo > 1) {
            const u32 mid = (lo + hi) >> 1;
            if (s_goff[mid] <= q) lo = mid; else hi = mid;
        }
        const u32 k = (u32)(q - s_goff[lo]);
        const uint4 e = A.tmpl[s_top[lo] + k];
        const u32 refs[3] = {e.x, e.y, e.z};
        u32 w3[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const u32 kind = refs[r] >> 30, idx = refs[r] & 0x3FFFFFFFu;
            u32 v;
            if (kind == 3) v = (u32)(A.aux_base + s_aoff[lo] + idx);
            else v = bool_wire(kind == 0 ? s_in0[lo] : (kind == 1 ? s_in1[lo] : s_out[lo]), idx, A.width, A.M, A.out_base);
            w3[r] = v;
        }
        A.b_in0[q] = w3[0];
        A.b_in1[q] = w3[1];
        A.b_out[q] = w3[2];
        A.b_op[q] = (u8)e.w;
    }
}

// order-sensitive 64-bit checksum of a u32 stream: sum over i of mix(i, v[i]) (commutative combine of
// position-salted hashes => parallel, deterministic).  Used by the full-size parity tests.
__device__ __forceinline__ u64 mix64(u64 x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
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

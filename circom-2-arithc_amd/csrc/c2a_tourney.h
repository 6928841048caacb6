// c2a_tourney.h — the TOURNAMENT half of the decoupled peel (the other half, the claim engine, is in c2a_claim.h).
//
// The DFS tree of topological_sort (/root/reference/src/topological_sort.rs:3-50) is built from node records exactly as
// c2a_peel.h describes (path strings, self-validating 8-byte words, one round trip per comparison).  What differs is WHO
// runs a gate's tournament and WHEN:
//   * the claim engine walks the reverse Kahn order with nothing but tickets (4-byte atomics, one lane per chain) and
//     appends every gate it claims to an ORDER QUEUE — a gate is appended after all of its consumers;
//   * the waves of this kernel take queue positions in turn (wave v: positions v, v + W, v + 2 W, ... of its queue) and run
//     each gate's tournament as soon as the records of its consumers are there (self-validating: a reader that sees the
//     wrong tag reads again).  No tickets, no hand-off, no termination protocol on this side: a wave is done when the
//     claim engine has said how long its queue is and the wave has passed that.
// No deadlock: every gate's consumers sit at smaller queue times; the unfinished gate that was appended first has all its
// consumers finished, and the wave that owns its position has nothing older left — so it is working on it.
// The level-to-level latency of the tree is then: a record becomes visible -> the next gate's wave (already polling, its
// static data long since loaded) compares -> stores.  One memory round trip and ~150 instructions per level, instead of
// the whole ticket / issue / tournament / hand-off step of the one-kernel dataflow launch.
#pragma once
#include "c2a_peel.h"

namespace c2a {

struct TourArgs {
    u32 epoch;                 // tag (0/1) of this run's node words
    u32 n;
    const uint4* gstat;        // [2n] static records (c2a_peel.h)
    const u32* clist;          // consumer lists
    u64* node;                 // [n][64] node records
    uint4* meta;               // [n] {parent | NONE, depth, root, label | level << 1}
    u32* child;                // [2n] tree children by label
    const u64* oq;             // [n_queues][q_cap] order entries: gate | run << 32 (never cleared: the tag says which run)
    u32 n_queues, q_cap, run;
    const u32* oq_final;       // [n_queues] entries of each queue, valid once ctl[TCTL_CLAIM_DONE] is set
    u32* ctl;                  // [TCTL_WORDS]
};
enum TourCtl { TCTL_ABORT = CTL_ABORT /* shared with the out-of-line helpers of c2a_peel.h */, TCTL_CLAIM_DONE = 32, TCTL_PROCESSED = 64, TCTL_MAXLEVEL = 65, TCTL_REREADS = 66, TCTL_WORDS = 128 };

#ifdef C2A_EMULATE
constexpr u32 kTourPollLimit = 1u << 16;    // (the emulation interleaves the waves at every back-off: a poll is one turn of all the others)
#else
constexpr u32 kTourPollLimit = 1u << 21;
#endif
#ifndef C2A_TOUR_GROUP
#define C2A_TOUR_GROUP 4        // candidate records in flight per gate
#endif

// one gate's tournament.  All arguments wave-uniform.  Returns false when a record never arrived (abort).
__device__ __forceinline__ bool tour_gate(const TourArgs& A, u32 g, u32 lane, u32& max_level, u32& rereads) {
    const u32 epoch = A.epoch;
    const uint4 gi = A.gstat[2 * (u64)g];
    const u32 g_off = uniform(gi.z), g_cnt = uniform(gi.w);
    // the champion so far (wave-uniform); ch == NONE: the virtual-root candidate [g]
    u32 ch = C2A_NONE, ch_el = 0, ch_root = g, ch_depth = 0, ch_pos = 0;
    u64 ch_w = 0, ch_x = 0;
    u32 level = 0;
    bool gave_up = false;
    auto candidate = [&](u64 w, u32 e) {
        const u32 c = e & kIdMask, el = e >> 31;
        const u64 h0 = rdlane64(w, 0);
        const u32 croot = hdr_hi(h0), cdepth = (u32)h0;
        if (cdepth == 0u && lane >= kHdrWords) w = (u64)epoch << 63;          // (a depth-0 record has header words only: see valid())
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
    // a record is all there when every word carries this run's tag — or, for a depth-0 record (a sink: header words only),
    // when its three header words do
    auto valid = [&](u64 w) -> bool {
        const u64 badm = __ballot((u32)(w >> 63) != epoch);
        if (badm == 0) return true;
        return (badm & 7ull) == 0 && (u32)rdlane64(w, 0) == 0u;
    };
    for (u32 eb = 0; eb < g_cnt && !gave_up; eb += 64) {
        const u32 blk = A.clist[g_off + eb + lane];                  // (clist is padded by 64 entries)
        const u32 m = g_cnt - eb < 64u ? g_cnt - eb : 64u;
        for (u32 j = 0; j < m && !gave_up; j += C2A_TOUR_GROUP) {
            u32 e[C2A_TOUR_GROUP];
            u64 w[C2A_TOUR_GROUP];
            const u32 k = m - j < (u32)C2A_TOUR_GROUP ? m - j : (u32)C2A_TOUR_GROUP;
#pragma unroll
            for (u32 t = 0; t < C2A_TOUR_GROUP; ++t) {
                e[t] = rdlane(blk, j + t < m ? j + t : j);
                if (t < k) w[t] = ld_nw(&A.node[(u64)(e[t] & kIdMask) * kNodeWords + lane]);
            }
            // read again what is not there yet (the consumers are being worked on right now by other waves)
            u32 polls = 0;
            for (;;) {
                u32 bad = 0;
#pragma unroll
                for (u32 t = 0; t < C2A_TOUR_GROUP; ++t) if (t < k && !valid(w[t])) bad |= 1u << t;
                if (!bad) break;
                if (++polls > kTourPollLimit) { gave_up = true; break; }
                peel_sleep(polls < 4 ? 1 : (polls < 32 ? 4 : 16));
#pragma unroll
                for (u32 t = 0; t < C2A_TOUR_GROUP; ++t)
                    if (bad & (1u << t)) w[t] = ld_nw(&A.node[(u64)(e[t] & kIdMask) * kNodeWords + lane]);
            }
            rereads += polls;
            if (gave_up) break;
#pragma unroll
            for (u32 t = 0; t < C2A_TOUR_GROUP; ++t) if (t < k) candidate(w[t], e[t]);
        }
    }
    if (gave_up) return false;
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
    st_nw(&A.node[(u64)g * kNodeWords + lane], (u64)w_lo | ((u64)w_hi << 32));
    if (lane == 0) {
        A.meta[g] = make_uint4(ch, depth, ch_root, my_label | (level << 1));
        if (ch != C2A_NONE) A.child[2 * (u64)ch + my_label] = g;
    }
    wave_join();
    return true;
}

// 64-thread workgroups (one wave each); wave v works on queue v % n_queues, positions v / n_queues + k * (waves / n_queues)
__global__ void __launch_bounds__(64) k_tourney(TourArgs A) {
    const u32 lane = threadIdx.x;
    const u32 q = blockIdx.x % A.n_queues;
    const u32 stride = gridDim.x / A.n_queues;
    const u64* my_q = A.oq + (u64)q * A.q_cap;
    u32 processed = 0, max_level = 0, rereads = 0;
    for (u32 i = blockIdx.x / A.n_queues; i < A.q_cap; i += stride) {
        // ---- this wave's next entry (the claim engine may not have written it yet)
        u64 ent = 0;
        u32 polls = 0;
        bool stop = false;
        for (;;) {
            if (lane == 0) ent = ld_nw(&my_q[i]); wave_join();
            ent = rdlane64(ent, 0);
            if ((u32)(ent >> 32) == A.run) break;
            ++polls;
            if ((polls & 7u) == 0) {
                u32 c3 = 0;
                if (lane < 2) c3 = ld_a32(&A.ctl[lane == 0 ? TCTL_ABORT : TCTL_CLAIM_DONE]); wave_join();
                if (rdlane(c3, 0)) { stop = true; break; }
                if (rdlane(c3, 1)) {
                    // the claim engine is done: its queues have their final lengths (an entry below that length is written)
                    u32 fin = 0;
                    if (lane == 0) fin = ld_a32(&A.oq_final[q]); wave_join();
                    if (i >= rdlane(fin, 0)) { stop = true; break; }
                }
                if (polls > kTourPollLimit) { if (lane == 0) atomicAdd(&A.ctl[TCTL_ABORT], 1u); wave_join(); stop = true; break; }
            }
            peel_sleep(polls < 8 ? 4 : (polls < 64 ? 16 : 64));
        }
        if (stop) break;
        const u32 g = uniform((u32)ent & kIdMask);
        if (!tour_gate(A, g, lane, max_level, rereads)) { if (lane == 0) atomicAdd(&A.ctl[TCTL_ABORT], 1u); wave_join(); break; }
        ++processed;
    }
    if (lane == 0) {
        if (processed) atomicAdd(&A.ctl[TCTL_PROCESSED], processed);
        if (max_level) atomicMax(&A.ctl[TCTL_MAXLEVEL], max_level);
        if (rereads) atomicAdd(&A.ctl[TCTL_REREADS], rereads);
    }
}

// experiment / tests: order entries from a plain list of gate ids
__global__ void k_make_entries(u32 cnt, const u32* __restrict__ order, u32 run, u64* oq) {
    for (u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (u64)gridDim.x * blockDim.x) oq[i] = (u64)order[i] | ((u64)run << 32);
}

}  // namespace c2a

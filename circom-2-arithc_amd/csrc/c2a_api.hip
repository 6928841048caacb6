// c2a_api.hip — host runtime behind include/c2a.h: context, HBM workspace, kernel orchestration.
// Built by hipcc for gfx950 into libc2a_hip.so (the product) and, for the CPU test-suite only, by g++
// against tests/emul/hip_emul.h into tests/emul/libc2a_emul.so.
#include "../../include/c2a.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "c2a_kernels.h"
#include "c2a_templates.h"

using namespace c2a;

namespace {

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    template <class T> T* as() const { return reinterpret_cast<T*>(p); }
};

enum Stage { ST_EMPTY = 0, ST_LOADED, ST_SORTED, ST_WIRED, ST_EMITTED, ST_BOOLIFIED };

enum Ev { EV_PREP0, EV_PREP1, EV_PEEL1, EV_ORDER1, EV_WIRES0, EV_WIRES1, EV_EMIT0, EV_EMIT1, EV_BPREP0, EV_BPREP1,
          EV_BMAP1, EV_BUILD0, EV_BUILD1, EV_KPEEL0, EV_KPEEL1, EV_ORDER_RB /* the order stage's numbers have been posted */, EV_COUNT };

}  // namespace

// One extra device of a multi-device context: it bit-blasts its own sorted-position range of the emitted circuit.
struct PeerDev {
    int device = 0;
    hipStream_t stream{};
    u32 p_lo = 0, p_hi = 0;        // sorted positions [p_lo, p_hi)
    u64 q_lo = 0, q_hi = 0;        // boolean gates [q_lo, q_hi) of the whole circuit
    u64 q_bias = 0;                // boolean gate q is stored at index q - q_bias
    DevBuf e_in0, e_in1, e_out, e_op, goff, aoff, tmpl, tables, b_in0, b_in1, b_out, b_op, acc;
    DevBuf fmt_len, fmt_off, fmt_text, fmt_table, scan_tmp;      // c2a_format_bristol prints a device's own gates on that device
    DevBuf vscratch;                                             // c2a_verify_boolify checks a device's own gates on that device
    u32 tmpl_width = 0, tmpl_mask = 0;
};

struct c2a_ctx {
    int device = 0;                // primary device: the sort, the wire numbering, the emission, shard 0 of boolify
    std::vector<PeerDev> peers;    // devices 1..N-1 of c2a_create (boolify shards 1..N-1)
    u32 shard0_hi = 0;             // multi-device: the primary's own range is [0, shard0_hi)
    u64 shard0_qhi = 0;
    hipStream_t stream{};
    hipStream_t aux{};             // a second stream: the node-record clear of a new graph runs beside the payload's way over PCIe
    hipEvent_t ev[EV_COUNT]{};
    bool ev_valid[EV_COUNT]{};
    std::string err;
    Stage stage = ST_EMPTY;
    int n_cu = 256;
    u32 bool_threads = 1024;       // threads of a k_boolify workgroup (C2A_BOOL_THREADS: 256, 512, 1024).  What the kernel takes depends on where its 9.6 GB of
                                   // output landed (tools/bool_alloc.py: the same process, context after context, 1.7 - 2.3 ms with 256 threads; PMC: DRAM write
                                   // credit stalls x 3 in the slow ones, translation misses the same); 1 024 threads — a quarter of the workgroups in flight,
                                   // each done four times sooner: the window of memory being written is a quarter as wide — 1.7 - 1.95
    u32 bool_chunk = 256;          // arithmetic gates per k_boolify workgroup: 128, 256 (measured best) or 512
    u32 bool_slices = 0;           // workgroups per chunk (0: chosen per launch — bool_map; C2A_BOOL_SLICES forces it: tests)
    u32 peel_seed_chunk = 4;       // dataflow launch: seeds a wave takes at a time (1: the one counter they all hit cost 1.4 ms with 175 000 seeds; with the 33 000 the shallow passes leave: 2 / 4 / 8 / 16 = 7.72 / 7.70 / 7.78 / 7.85 ms)
    u32 peel_shallow = 4;          // levels behind the sinks done a whole level at once before the dataflow launch (>= 1, <= 48)
    u32 peel_sinks_blocks = 4096;  // grid cap of the sinks pass (latency-bound per thread: two dependent round trips per sink)
    u32 peel_waves = 8;            // dataflow launch: single-wave workgroups per CU (clamped by the occupancy query)
    u32 peel_fifos = 64;           // dataflow launch: hand-off arrays (a power of two <= 64)
    u32 peel_reserve = 8;          // dataflow launch: reserve waves per CU, parked until entries pile up in the hand-off arrays with nobody in line for them
    bool peel_waves_forced = false, peel_reserve_forced = false;      // (set by the environment: tests and tools/ sweeps get what they ask for at any size)
    u32 peel_release = 0;          // ... this many of them (0: twice the primary waves = 4 096.  10 M gates, k_peel ms without a reserve / with: a matrix
                                   // product's 29 000 reduction chains 7.35 / 4.96, a butterfly 20 x 2^19 7.66 / 5.52, ten layers of 10^6 4.28 / 3.54, a Merkle tree of
                                   // 2 900 SHA-256 blocks 12.5 / 11.0; the headline 5.85 / 5.88 (its backlog peaks at 2-4 000: at 2 048 the reserve comes in, 5.97),
                                   // strict layers 11.2 / 11.3, hubs 5.9 / 6.05 (comes in), SHA-256 x 8 4.95 / 4.98; at 6 144 the Merkle tree stays at 12.7)
    size_t peel_slots = 0; u32 peel_waves_used = 0; bool peel_want_stats = false;      // of the launch now queued (peel_launch -> peel_result)
    u32 build_no = 0;              // number of the last producer map on this context (tag of its node-table records: k_producer)
    u32 peel_run = 0;              // number of the last dataflow run on this context (tag of its hand-off entries)
    u32 peel_epoch = 0;            // tag of the node words written by the last run (1 / 2 take turns; 0 after a clear)
    bool circuit_only = false;     // c2a_load_circuit: the context holds an EMITTED circuit the host built itself (no gate graph: nothing to sort or number)
    bool wires_clear_valid = false; // the event bits (pblk) and the descriptors of their scan (scan_desc + R.bits) are as the build's ONE k_clear left them
    bool node_init_valid = false;  // node_wire1[] / first[] are as k_relabel left them (no wire, not seen): the wire numbering may start (else: k_node_init)
    bool io_clash = false;         // a node is both an input and an output (compiler.rs:363-383), found at load time
    bool peel_meta_valid = false;  // meta[] / stats.levels describe the circuit now loaded (c2a_verify_boolify schedules by them)
    bool node_clear = true;        // node records must be zeroed before the next run (new graph, or a run that failed)
    bool peel_deep = false;        // the DFS tree of the loaded graph is deeper than one chunk of path string: the DEEP build of the dataflow launch
    bool peel_gave_up = false;     // the last dataflow launch ended by its watchdog (do_peel retries once on clean buffers, then do_topo_sort sorts serially)
    bool serial_fallback = false;  // ... and that happened for the circuit now loaded (logged once per context)
    bool fallback_logged = false;
    bool numbering_walk = false;   // C2A_NUMBERING_WALK=1: never the positional numbering (tests and A/B runs: the walk in sorted order on any circuit)
    u32 debug_peel_abort = 0;      // tests only (c2a_debug_peel_abort): this many dataflow launches are treated as given up
    bool hot_every_forced = false; // (c2a_debug_hot_every was called: the tests' threshold, whatever the gate count)
    u32 hot_every = kHotEvery;     // k_deps: the consumer ticket that makes a producer "hot" (c2a_debug_hot_every lowers it for the tests)

    // problem
    u32 n = 0, n_nodes = 0, n_in = 0, n_out = 0;
    // results (host copies of scalars)
    u32 wire_count = 0, n_mid = 0;
    c2a_stats stats{};
    c2a_bool_info binfo{};
    u32 bool_width = 0;
    u32 bool_op_mask = 0;          // gate types whose templates are on the device (for bool_width)
    u32 bool_max_aux = 0;          // most aux wires any template of that width has (scratch of the local verifier)
    u64 op_hist[C2A_NUM_GATE_TYPES] = {};   // gates per type of the loaded circuit (c2a_load_gates looks at every op byte anyway): the totals of a
                                   // boolify plan are sums over it — the plan needs no read-back in the middle of c2a_boolify
    u32 host_tsize[20] = {}, host_taux[20] = {};      // T(op, w) / AUX(op, w) of the templates now on the device (bool_width)

    // device buffers
    DevBuf lh, rh, out, op, gate4, nrec, orig, in_nodes, out_nodes;
    DevBuf prod1, dep0, dep1, cons_cnt, cons_off, eslot, aq_items, aq_pc, aq_seeds, aq_seeds1, aq_seed_flat, aq_seed_cnt, fill, meta, node, child, gstat, clist, pctl, pcold;
    DevBuf rflag;                  // root bits by rank (k_root_bits), one u64 per 64 ranks
    DevBuf rbits, rpre, ridx, rlist, next, owner, local, slist, sjump, sjump2, sorted, sorted_r;
    DevBuf first, nflag, wflag, widx, node_wire1, node_wire, e_in0, e_in1, e_out, e_op;
    DevBuf pos_r, wire_r, erec, pblk, dpre, epre, gflag;      // positional numbering (c2a_kernels.h POSITIONAL NUMBERING)
    u32* hrb = nullptr;            // 256 words of host memory the device writes the end-of-stage numbers to (k_post_*) ...
    u32* hrb_dev = nullptr;        // ... as the device sees it.  Words 0-7: peel, 8-15: order, 16-23: wires, 24-31: boolify
    u32 rb_edges = 0, rb_dup = 0, rb_nmid = 0, rb_err = 0;  // read-back slots (edge count, duplicate-writer flag, wires handed out, in/out clash)
    u32 nmid_add = 0;              // what the posted wire count is relative to (positional numbering: n + the events' net shift)
    bool has_dup = false;          // two gates write one node (compiler.rs:403-406 keeps the last): the general numbering path
    bool positional = false;       // the circuit now sorted takes the positional numbering (one writer per node)
    bool sorted_ready = false;     // sorted[] / sorted_r[] are written (c2a_build_circuit leaves them to the emission's split pass)
    bool emitted_with_wires = false;   // the positional numbering has emitted the gates as well (do_emit has nothing left to do)
    u32 jump_rounds = 0; uint2* jump_a = nullptr; uint2* jump_b = nullptr;      // pointer jumping: launches queued, the ping-pong buffers as they stand
    const uint2* rank_suffix = nullptr;    // the splitter suffix sums the list ranking ended in (which of its ping-pong buffers)
    DevBuf scan_tmp, scan_desc, scalars, dfs_state, dfs_stack, peel_prof, peel_trace;
    DevBuf tsz, asz, goff, aoff, tmpl, tables, b_pool;
    DevBuf b_in0, b_in1, b_out, b_op;      // VIEWS into b_pool (ensure_bool_out): one hipMalloc for the four streams of the boolean circuit — four cost a one-shot caller 0.9 ms
    DevBuf fmt_len, fmt_off, fmt_text, fmt_table, shard_cut, shard_qcut;
    u64 fmt_chunk_first = 0, fmt_chunk_cnt = 0;      // boolean gates held by the chunk buffers (c2a_boolify_chunk)
    bool fmt_chunk_valid = false;                    // ... of the circuit and plan now current (reset wherever the plan is)
    DevBuf g_in0, g_in1, g_out, g_op, pr_rep, pr_need, pr_tin0, pr_tin1, pr_top, pr_live, pr_goff, pr_counts, p_in0, p_in1, p_out, p_op;
    DevBuf ev_produced, ev_spos, ev_aval, ev_bval, ev_lcount, ev_lbase, ev_lorder, ev_bar, ev_io, cb_in0, cb_in1, cb_out, cb_op;
    bool bool_planned = false;
    bool peer_access = false;      // multi-device: the primary can read every peer's memory (hipDeviceEnablePeerAccess at c2a_create): the level-parallel passes read the pieces of the boolean circuit where they lie
    DevBuf segs;                   // the BoolSegs of the last such pass
    bool gathered = false;         // multi-device WITHOUT peer access: g_* hold the whole boolean circuit of the last c2a_boolify on the primary device
    bool pruned = false;           // p_* hold the pruned image of the boolean circuit now in b_*
    c2a_prune_info pinfo{};
    std::vector<DevBuf*> all;

    c2a_ctx() {
        all = {&segs, &rflag, &lh, &rh, &out, &op, &gate4, &nrec, &orig, &in_nodes, &out_nodes, &prod1, &dep0, &dep1, &cons_cnt, &cons_off, &eslot, &aq_items, &aq_pc, &aq_seeds, &aq_seeds1, &aq_seed_flat, &aq_seed_cnt, &fill,
               &gstat, &clist, &pctl, &pcold, &meta, &node, &child, &rbits, &rpre, &ridx, &rlist, &next,
               &owner, &local, &slist, &sjump, &sjump2, &sorted, &sorted_r, &first, &nflag, &wflag, &widx, &node_wire1,
               &node_wire, &e_in0, &e_in1, &e_out, &e_op, &pos_r, &wire_r, &erec, &pblk, &dpre, &epre, &gflag, &scan_tmp, &scan_desc, &scalars, &dfs_state, &dfs_stack, &peel_prof, &peel_trace, &tsz, &asz, &goff,
               &aoff, &tmpl, &tables, &b_pool, &fmt_len, &fmt_off, &fmt_text, &fmt_table, &shard_cut, &shard_qcut, &ev_produced, &ev_spos, &ev_aval, &ev_bval, &ev_lcount, &ev_lbase, &ev_lorder, &ev_bar, &ev_io, &g_in0, &g_in1, &g_out, &g_op, &pr_rep, &pr_need, &pr_tin0, &pr_tin1, &pr_top, &pr_live, &pr_goff, &pr_counts, &p_in0, &p_in1, &p_out, &p_op, &cb_in0, &cb_in1, &cb_out, &cb_op};
    }
};

namespace {

// scalars block layout (u32 words unless noted)
enum Scalar { SC_MAXDEPTH = 0, SC_SCOUNT = 1, SC_ERR = 2, SC_NMID = 3, SC_NROOTS = 4, SC_LEVELS = 5, SC_PEELOK = 6 /* the dataflow launch ended cleanly and left no gate behind (k_root_bits) */,
              SC_RELAYS = 7 /* relays of all hubs of this build (k_gstat) */,
              SC_DFS = 8 /*3 words*/, SC_DUP = 52,
              SC_TOTAL64 = 16 /* u64 slots from here: 16..31 */, SC_HOT = 64 /* 1 + kHotMax words: the hot producers k_deps found (c2a_kernels.h HOT PRODUCERS) */,
              SC_WORDS = 64 + 1 + 64 + 3 };
static_assert(SC_WORDS >= SC_HOT + 1 + (int)kHotMax && SC_WORDS % 4 == 0, "the hot list lives behind the scalars (cleared with them by k_clear)");

// internal status of peel_result (never crosses the ABI): the plain build of the dataflow launch met a DFS tree deeper than one chunk of
// path string and ended itself — run the DEEP build
constexpr int kRerunDeep = -1000;

int fail(c2a_ctx* c, int code, const std::string& msg) {
    if (c) c->err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return fail(c, C2A_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));             \
    } while (0)

int ensure(c2a_ctx* c, DevBuf& b, size_t bytes);
// the four SoA streams of G boolean gates (+ 16 of slack each: k_boolify stores op bytes sixteen at a time) in ONE allocation
int ensure_bool_out(c2a_ctx* c, u64 G) {
    const size_t w4 = (((size_t)G + 16) * 4 + 4095) & ~(size_t)4095, w1 = ((size_t)G + 16 + 4095) & ~(size_t)4095;
    int r = ensure(c, c->b_pool, 3 * w4 + w1);
    if (r) {                                         // (the old pool is gone: its four views must not outlive it)
        for (DevBuf* v : {&c->b_in0, &c->b_in1, &c->b_out, &c->b_op}) { v->p = nullptr; v->cap = 0; }
        return r;
    }
    char* p = c->b_pool.as<char>();
    c->b_in0.p = p; c->b_in0.cap = w4; c->b_in1.p = p + w4; c->b_in1.cap = w4; c->b_out.p = p + 2 * w4; c->b_out.cap = w4; c->b_op.p = p + 3 * w4; c->b_op.cap = w1;
    return C2A_OK;
}
int ensure(c2a_ctx* c, DevBuf& b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return C2A_OK;
    if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e != hipSuccess) {
        b.p = nullptr;
        return fail(c, C2A_ERR_NOMEM, "hipMalloc(" + std::to_string(bytes) + " bytes): " + hipGetErrorString(e));
    }
    b.cap = bytes;
    return C2A_OK;
}
#define ENSURE(buf, bytes)                                   \
    do {                                                     \
        int _r = ensure(c, (buf), (size_t)(bytes));          \
        if (_r) return _r;                                   \
    } while (0)

inline u32 grid_for(u64 items, u32 cap_blocks) {
    u64 b = (items + kThreads - 1) / kThreads;
    if (b < 1) b = 1;
    if (b > 8) b = (b + 7) & ~7ull;              // (whole multiples of the eight XCDs: xcd_sweep gives each an eighth of the range)
    if (b > cap_blocks) b = cap_blocks;
    return (u32)b;
}

void rec(c2a_ctx* c, Ev e) {
    if (hipEventRecord(c->ev[e], c->stream) == hipSuccess) c->ev_valid[e] = true;
}

// one-launch exclusive scans (k_scan_stream): NC sums of the functor's element values into out0 (/ out1), n + 1 entries each
// (out[n] = the total).  Descriptors + the ticket counter live in scan_tmp.
inline size_t scan_desc_bytes(u64 n, int nc) { return (64 + (size_t)((n + kScanTile - 1) / kScanTile) * nc * 8 + 15) & ~(size_t)15; }
// (region: descriptors that are zero already — the build's own scans, cleared by ONE k_clear in front of the build — or
// nullptr: the context's scan_tmp, cleared here)
template <int NC, class F, typename TOut>
int scan_1pass(c2a_ctx* c, hipStream_t s, DevBuf& tmp, u64 n, F f, TOut* out0, TOut* out1, void* region = nullptr) {
    if (n == 0) {
        HIP_TRY(hipMemsetAsync(out0, 0, sizeof(TOut), s));
        if (NC > 1) HIP_TRY(hipMemsetAsync(out1, 0, sizeof(TOut), s));
        return C2A_OK;
    }
    const u64 tiles = (n + kScanTile - 1) / kScanTile, groups = (tiles + ScanGeom<NC>::kGroup - 1) / ScanGeom<NC>::kGroup;
    if (!region) {
        const size_t bytes = scan_desc_bytes(n, NC);
        ENSURE(tmp, bytes);
        HIP_TRY(hipMemsetAsync(tmp.p, 0, bytes, s));
        region = tmp.p;
    }
    C2A_LAUNCH((k_scan_stream<NC, F, TOut>), (u32)groups, ScanGeom<NC>::kThreads, s, n, f, out0, out1, reinterpret_cast<u64*>(region) + 8, reinterpret_cast<u32*>(region));
    return C2A_OK;
}

// exclusive scan of `in` (n entries, u32) into `out` (n+1 entries; out[n] = total).
template <typename TOut>
int scan_exclusive(c2a_ctx* c, const u32* in, TOut* out, u64 n, void* region = nullptr) {
    return scan_1pass<1>(c, c->stream, c->scan_tmp, n, ScanFromU32{in}, out, (TOut*)nullptr, region);
}

int read_scalars(c2a_ctx* c, u32* host, int first, int count) {
    HIP_TRY(hipMemcpyAsync(host, c->scalars.as<u32>() + first, sizeof(u32) * count, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return C2A_OK;
}

// the scalars block cleared and the IO flags of the nodes set (compiler.rs:363-395: a node that is both raises SC_ERR) — in
// front of k_deps, which folds "the out node is an IO node" into the payload records
// The build's zeroing in ONE launch (k_clear): the scalars block, the node flags, the descriptors of the build's four scans
// (one buffer, a region each: relabel, consumer counts, root bitmap, event bits), the dataflow launch's dummy ticket words /
// hand-off ticket words / control block / seed-region counts, the root bitmap, the event bits of the positional numbering.
struct BuildRegions { size_t relabel, cons, roots, bits, total; };
BuildRegions build_regions(const c2a_ctx* c) {
    BuildRegions R;
    R.relabel = 0;
    R.cons = R.relabel + ((64 + (size_t)(((u64)c->n_nodes + kRelTile - 1) / kRelTile) * 8 + 15) & ~(size_t)15);
    R.roots = R.cons + scan_desc_bytes(c->n, 1);
    R.bits = R.roots + scan_desc_bytes(((u64)c->n + 31) / 32, 1);
    R.total = R.bits + scan_desc_bytes(((u64)c->n + 31) / 32, 2);
    return R;
}
inline u32 n_all_of(u32 n) { return n + (u32)relay_cap(n); }                         // real gates + room for the relays of its hubs (c2a_peel.h)
inline size_t fill_dummy_off(u32 n) { return (((size_t)n + 3) & ~(size_t)3) * 4; }      // bytes: the waves' dummy ticket words start on a 16-byte boundary behind fill[n]
inline size_t seed_cnt_bytes(const c2a_ctx* c, u32 sink_blocks) { return (((size_t)(c->peel_shallow + 1) * sink_blocks + 16) * 4 + 15) & ~(size_t)15; }
inline u32 sink_blocks_of(const c2a_ctx* c);
int clear_for_build(c2a_ctx* c, bool peel_only = false) {
    ClearList L{};
    u64 run = 0;
    auto add = [&](void* p, size_t bytes) { if (!bytes) return; L.p[L.cnt] = reinterpret_cast<uint4*>(p); run += (bytes + 15) / 16; L.end[L.cnt] = run; ++L.cnt; };
    const u32 n = c->n;
    if (!peel_only) {
        add(c->scalars.p, SC_WORDS * 4);
        add(c->nflag.p, (size_t)c->n_nodes);
        if (n) {
            const BuildRegions R = build_regions(c);
            ENSURE(c->scan_desc, R.total);
            add(c->scan_desc.p, R.total);
            add(c->rbits.p, ((size_t)n + 31) / 32 * 4);
            add(c->pblk.p, ((size_t)n + 31) / 32 * 16);
            c->wires_clear_valid = true;
        }
    }
    if (n && peel_only) {                            // (a second attempt: what the first one's order stage has used)
        const BuildRegions R = build_regions(c);
        add(c->scan_desc.as<char>() + R.roots, R.bits - R.roots);
        add(c->rbits.p, ((size_t)n + 31) / 32 * 4);
    }
    if (n) {
        const u32 sb = sink_blocks_of(c);
        ENSURE(c->aq_pc, (size_t)c->peel_fifos * kPcStride * 8);
        ENSURE(c->pctl, ((size_t)CTL_WORDS * 4 + 15) & ~(size_t)15);
        ENSURE(c->aq_seed_cnt, seed_cnt_bytes(c, sb));
        add(reinterpret_cast<char*>(c->fill.p) + fill_dummy_off(n_all_of(n)), (size_t)kFillDummyStride * kFillDummyWaves * 4);
        add(c->aq_pc.p, (size_t)c->peel_fifos * kPcStride * 8);
        add(c->pctl.p, (size_t)CTL_WORDS * 4);
        add(c->aq_seed_cnt.p, seed_cnt_bytes(c, sb));
    }
    static_assert(kClearMax >= 9, "regions of a build");
    if (!L.cnt) return C2A_OK;
    C2A_LAUNCH(k_clear, grid_for(run, 2048), kThreads, c->stream, L);
    return C2A_OK;
}

int mark_io(c2a_ctx* c) {
    hipStream_t s = c->stream;
    if (c->n_in) C2A_LAUNCH_NOSYNC(k_mark_inputs, grid_for(c->n_in, 1024), kThreads, s, c->n_in, c->in_nodes.as<u32>(), c->nflag.as<u8>());
    if (c->n_out) C2A_LAUNCH_NOSYNC(k_mark_outputs, grid_for(c->n_out, 1024), kThreads, s, c->n_out, c->out_nodes.as<u32>(), c->nflag.as<u8>(), c->scalars.as<u32>() + SC_ERR);
    return C2A_OK;
}

// producer map (compiler.rs:401-406), relabelling by out-node order (c2a_kernels.h RELABELLING), deps closure
// (compiler.rs:408-421), consumer lists and the static records of the dataflow launch — everything behind k_relabel in rank space
int do_prep(c2a_ctx* c, bool for_peel = true) {
    const u32 n = c->n;
    hipStream_t s = c->stream;
    const u32 G = grid_for(n, 4096);
    int r0 = clear_for_build(c);
    if (r0) return r0;
    if ((r0 = mark_io(c))) return r0;
    const BuildRegions R = build_regions(c);
    char* desc = c->scan_desc.as<char>();
    u32* dup = c->scalars.as<u32>() + SC_DUP;
    // (the node records carry the number of the build that wrote them — 24 bits — and are only ever cleared when that wraps)
    if (++c->build_no >= (1u << 24)) { HIP_TRY(hipMemsetAsync(c->nrec.p, 0, (size_t)c->n_nodes * 16, s)); c->build_no = 1; }
    C2A_LAUNCH_NOSYNC(k_producer, G, kThreads, s, n, c->lh.as<u32>(), c->rh.as<u32>(), c->out.as<u32>(), c->op.as<u8>(), c->nrec.as<uint4>(), c->build_no,
                      c->cons_cnt.as<u32>(), for_peel ? c->fill.as<u32>() : (u32*)nullptr, for_peel ? c->child.as<uint2>() : (uint2*)nullptr);
    {
        const u64 tiles = ((u64)c->n_nodes + kRelTile - 1) / kRelTile;
        C2A_LAUNCH(k_relabel, (u32)tiles, kRelThreads, s, c->n_nodes, n, c->build_no, c->prod1.as<u32>(), (const uint4*)c->nrec.as<uint4>(), dup, c->orig.as<u32>(),
                   c->gate4.as<uint4>(), reinterpret_cast<u64*>(desc + R.relabel) + 8, reinterpret_cast<u32*>(desc + R.relabel), c->node_wire1.as<u32>(), c->first.as<u32>());
        c->node_init_valid = true;
    }
    // (two gates wrote one node — never, for a circuit the reference's front-end built: these two leave at once)
    C2A_LAUNCH_NOSYNC(k_dup_clear, 512, kThreads, s, c->n_nodes, (const u32*)dup, c->prod1.as<u32>());
    C2A_LAUNCH_NOSYNC(k_dup_producer, 512, kThreads, s, n, (const u32*)dup, (const u32*)c->out.as<u32>(), c->prod1.as<u32>());
    int r;
    // k_deps finds its HOT producers on the way (c2a_kernels.h): a wave must go round several times for what one round finds to serve the
    // next — at least eight rounds per thread (a 1 M-gate graph in one round of a million threads took every ticket one by one: a node
    // read by a tenth of its gates was 1.1 ms) —, and a producer counts as hot from a 64th of the gates on (at most from its 16 384th
    // consumer, at least from its 1 024th)
    u32 hot_every = c->hot_every;
    if (!c->hot_every_forced) { hot_every = 1024; while (hot_every < kHotEvery && (u64)hot_every * 128 <= n) hot_every *= 2; }
    const u32 G_deps = std::min<u32>(G, grid_for(((u64)n + 7) / 8, 4096));
    C2A_LAUNCH(k_deps, G_deps, kThreads, s, n, c->lh.as<u32>(), c->rh.as<u32>(), c->out.as<u32>(), c->op.as<u8>(), dup, c->prod1.as<u32>(), (const u8*)c->nflag.as<u8>(), c->orig.as<u32>(),
                      c->gate4.as<uint4>(), c->dep0.as<u32>(), c->dep1.as<u32>(), c->cons_cnt.as<u32>(), c->eslot.as<u32>(), c->gflag.as<u8>(), c->scalars.as<u32>() + SC_HOT, hot_every);
    if (!for_peel) return C2A_OK;
    r = scan_exclusive<u32>(c, c->cons_cnt.as<u32>(), c->cons_off.as<u32>(), n, desc + R.cons);
    if (r) return r;
    C2A_LAUNCH(k_gstat, G, kThreads, s, n, c->dep0.as<u32>(), c->dep1.as<u32>(), c->cons_off.as<u32>(),
               (const uint2*)c->eslot.as<uint2>(), c->orig.as<u32>(), c->gstat.as<uint4>(), c->clist.as<u32>(), c->fill.as<u32>(), c->child.as<uint2>(), c->scalars.as<u32>() + SC_RELAYS);
    return C2A_OK;
}

// The grid of the dataflow launch: `peel_waves` single-wave workgroups per CU that take part from the start, plus
// `peel_reserve` per CU that stay out of the hand-off lines until a backlog builds up in them (c2a_peel.h, THE RESERVE), clamped to what
// fits the device at once (the launch is CORRECT with any grid — termination counts units of work, not waves — but
// waves beyond residency only queue up behind it).
u32 peel_grid(c2a_ctx* c, bool stats, u32* n_primary) {
#ifdef C2A_EMULATE
    (void)stats; (void)c;
    *n_primary = 12;
    return 16;                                      // the emulation runs them one after the other
#else
    int per_cu = 0;
    hipError_t e = stats ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_peel<true, true>, 64, 0)
                         : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_peel<false, true>, 64, 0);
    if (e != hipSuccess || per_cu < 1) per_cu = 1;
    // (a small graph cannot be short of waves: below 2^20 gates no reserve is launched — 2 048 workgroups that would only start and leave —,
    // below 4 096 one wave per CU does: the launch of a 22-gate circuit is mostly its workgroups coming and going)
    const u32 n_all = n_all_of(c->n);
    const u32 w = std::min<u32>(n_all < 4096u && !c->peel_waves_forced ? 1u : c->peel_waves, (u32)per_cu);
    const u32 r = std::min<u32>(n_all < (1u << 20) && !c->peel_reserve_forced ? 0u : c->peel_reserve, (u32)per_cu - w);
    *n_primary = std::max<u32>(1u, (u32)c->n_cu * w);
    return *n_primary + (u32)c->n_cu * r;
#endif
}

inline u32 sink_blocks_of(const c2a_ctx* c) { return grid_for(c->n, c->peel_sinks_blocks); }

// The whole peel as one dataflow launch (+ one grid-stride launch for the sinks and one per whole-level pass): see c2a_peel.h.
// QUEUES it — the control words, tickets and counts it starts from were zeroed by clear_for_build — and the kernel that posts
// its numbers to the host and raises SC_PEELOK for the order stage queued right behind; peel_result() reads them once the host
// has waited for the event behind the order stage's own posting.
int peel_launch(c2a_ctx* c) {
    const u32 n = c->n;
    hipStream_t s = c->stream;
    c->peel_gave_up = false;                         // (only a watchdog abort observed in THIS launch may trigger the retry / the serial fall-back)
    const bool want_stats = std::getenv("C2A_PEEL_STATS") != nullptr;
    PeelArgs A;
    const u32 n_all = n_all_of(n);
    A.n = n; A.n_all = n_all; A.gstat = c->gstat.as<uint4>(); A.clist = c->clist.as<u32>();
    A.node = c->node.as<u64>(); A.fill = c->fill.as<u32>(); A.meta = c->meta.as<uint4>(); A.child = c->child.as<u32>();
    PeelCold cold;
    // node words carry the tag of the run that wrote them: zeroed memory first sees tag 1, then 2 and 1 take turns (a word
    // left over from two runs ago holds the same value: the peel of one loaded graph is deterministic)
    if (c->node_clear) {
        HIP_TRY(hipMemsetAsync(c->node.p, 0, (size_t)n_all * kNodeWords * 8, s));
        c->peel_epoch = 0;
    }
    // (1 and 2 take turns; in the first run after a clear every lane of a record must carry the run's tag — c2a_peel.h NODE RECORDS)
    A.pad_thr = c->peel_epoch == 0 ? kThrExact : kThrPad;
    c->peel_epoch = c->peel_epoch == 1u ? 2u : 1u;
    c->node_clear = true;                            // until this run has finished cleanly
    A.epoch = c->peel_epoch;
    u32 n_primary = 0;
    const u32 waves = peel_grid(c, want_stats, &n_primary);
    // the waves' dummy ticket words behind fill[n] (c2a_peel.h, SCALAR TICKETS): zero, and they stay zero (only 0 is ever
    // added); a ticket word is addressed by a 32-bit byte offset from fill
    if (waves > kFillDummyWaves || n_all >= (1u << 29)) { c->err = "peel: grid or gate count beyond the ticket words' addressing"; return C2A_ERR_ARG; }
    // hand-off arrays: every slot is used once per run (no wrap-around).  A wave spreads its pushes round robin, so an
    // array receives at most pushes / n_fifos + waves entries, and a wave holds at most one unserved consumer ticket
    A.n_fifos = c->peel_fifos;
    A.q_cap = n_all / A.n_fifos + 2 * waves + 64;
    const size_t slots = (size_t)A.n_fifos * A.q_cap;
    if (slots >= (1ull << 32)) { c->err = "peel: hand-off slots beyond 32-bit addressing"; return C2A_ERR_ARG; }
    // (slots are never cleared between runs: every word of an entry carries the number of the run that wrote it)
    if (c->aq_items.cap < slots * kSlotWords * 8 || c->peel_run == 0xFFFFFFFFu) {
        ENSURE(c->aq_items, slots * kSlotWords * 8);
        HIP_TRY(hipMemsetAsync(c->aq_items.p, 0, c->aq_items.cap, s));
        c->peel_run = 0;
    }
    A.run = ++c->peel_run;
    A.n_primary = n_primary;
    A.reserve_min = c->peel_release ? c->peel_release : 2u * n_primary;      // entries waiting with nobody in line for them that call the parked waves in (c2a_peel.h, THE RESERVE)
    A.fifo = c->aq_items.as<u64>(); A.q_pc = c->aq_pc.as<u64>(); A.ctl = c->pctl.as<u32>();
    cold.stats = nullptr; cold.q_time = nullptr; cold.p_time = nullptr; cold.t_trace = nullptr;
    const char* trace_dir = want_stats ? std::getenv("C2A_PEEL_TRACE") : nullptr;
    if (want_stats) {
        ENSURE(c->peel_prof, 256 + 2 * slots * 8);
        HIP_TRY(hipMemsetAsync(c->peel_prof.p, 0, 256 + 2 * slots * 8, s));
        cold.stats = c->peel_prof.as<ull>(); cold.q_time = c->peel_prof.as<ull>() + 32; cold.p_time = cold.q_time + slots;
        if (trace_dir) {
            ENSURE(c->peel_trace, (size_t)n_all * 24);
            HIP_TRY(hipMemsetAsync(c->peel_trace.p, 0, (size_t)n_all * 24, s));
            cold.t_trace = c->peel_trace.as<ull>();
        }
    }
    // seed regions: one per workgroup of the sinks pass; a workgroup sees at most gates_per_block gates, each claims <= 2 producers
    const u32 sink_blocks = sink_blocks_of(c);
    const u64 gates_per_block = ((u64)n + (u64)sink_blocks * kThreads - 1) / ((u64)sink_blocks * kThreads) * kThreads;
    const u32 sink_cap = (u32)(2 * gates_per_block);
    // (region counts of the sinks pass and of every shallow pass behind it, then the length of the flat seed list)
    const u32 shallow = c->peel_shallow;
    const u32 l1_cap = 2 * sink_cap;
    ENSURE(c->aq_seeds, (size_t)sink_blocks * l1_cap * 4);
    A.seeds_w = c->aq_seeds.as<u32>(); A.seed_cnt_w = c->aq_seed_cnt.as<u32>(); A.region_cap = sink_cap;
    A.proc_word = CTL_PROC; A.proc_mask = kAcctShards - 1u;
    // ... what those claim is done by k_peel_shallow, a whole level at once (level 1, 2, ... `peel_shallow`: far wider than the
    // body of the graph), and what the LAST of these passes claims starts the chains of the dataflow launch (collected per
    // workgroup first, then moved to ONE list the waves of the launch take seed_chunk at a time; its length stays on the
    // device: the word behind the region counts)
    ENSURE(c->aq_seeds1, (size_t)sink_blocks * l1_cap * 4); ENSURE(c->aq_seed_flat, ((size_t)n_all + 64) * 4);
    cold.cons_off = c->cons_off.as<u32>(); cold.n = n;
    cold.seeds = c->aq_seed_flat.as<u32>(); cold.seed_total = c->aq_seed_cnt.as<u32>() + (size_t)(shallow + 1) * sink_blocks; cold.seed_chunk = c->peel_seed_chunk;
    // what only the edges of the launch touch travels as one small block in HBM (keeps the kernel's scalar registers free)
    // (written by a one-thread launch that takes it by value: a copy from this stack object would need a host round trip)
    ENSURE(c->pcold, sizeof(PeelCold));
    A.cold = c->pcold.as<PeelCold>();
    C2A_LAUNCH(k_peel_sinks, sink_blocks, kThreads, s, A, c->pcold.as<PeelCold>(), cold);
    // (the passes take turns on two region buffers; the sinks pass wrote the first with regions of sink_cap words)
    for (u32 lvl = 1; lvl <= shallow; ++lvl) {
        u32* cnts = c->aq_seed_cnt.as<u32>();
        const u32* in = (lvl & 1u) ? c->aq_seeds.as<u32>() : c->aq_seeds1.as<u32>();
        u32* out = (lvl & 1u) ? c->aq_seeds1.as<u32>() : c->aq_seeds.as<u32>();
        C2A_LAUNCH(k_peel_shallow, sink_blocks, kThreads, s, A, (const u32*)c->cons_off.as<u32>(), lvl, lvl == shallow ? 1u : 0u, in, (const u32*)(cnts + (size_t)(lvl - 1) * sink_blocks), lvl == 1 ? sink_cap : l1_cap,
                   out, cnts + (size_t)lvl * sink_blocks, l1_cap, c->aq_seed_flat.as<u32>(), cnts + (size_t)(shallow + 1) * sink_blocks);
    }
    // (every wave of the launch is alive at once under emulation too, interleaved at the back-offs — in a shuffled order per
    // C2A_EMUL_SEED: the ticket / hand-off / termination protocol is exercised without a GPU)
    rec(c, EV_KPEEL0);
    // (trees deeper than one chunk of path string take the DEEP build of the launch: found out by the plain one, remembered per graph)
    if (want_stats) { if (c->peel_deep) C2A_LAUNCH_CONCURRENT((k_peel<true, true>), waves, 64, s, A); else C2A_LAUNCH_CONCURRENT((k_peel<true, false>), waves, 64, s, A); }
    else if (c->peel_deep) C2A_LAUNCH_CONCURRENT((k_peel<false, true>), waves, 64, s, A);
    else C2A_LAUNCH_CONCURRENT((k_peel<false, false>), waves, 64, s, A);
    rec(c, EV_KPEEL1);
    static_assert(CTL_PROCESSED == 0 && CTL_MAXLEVEL == 1 && CTL_ABORT == 2 && CTL_REREADS == 3, "the order k_root_bits posts them in");
    c->peel_slots = slots; c->peel_waves_used = waves; c->peel_want_stats = want_stats;
    return C2A_OK;
}

// what the launch reported (hrb[0..5], posted by k_root_bits; the host has waited for an event behind it)
int peel_result(c2a_ctx* c, u32* peeled_out) {
    const u32 n = c->n;
    const size_t slots = c->peel_slots;
    const u32 waves = c->peel_waves_used;
    const bool want_stats = c->peel_want_stats;
    const char* trace_dir = want_stats ? std::getenv("C2A_PEEL_TRACE") : nullptr;
    PeelCold cold;
    cold.q_time = want_stats ? c->peel_prof.as<ull>() + 32 : nullptr; cold.p_time = want_stats ? cold.q_time + slots : nullptr;
    struct { u32 n_fifos; } A{c->peel_fifos};
    u32 t4[4] = {c->hrb[0], c->hrb[1], c->hrb[2], c->hrb[3]};
    if (c->debug_peel_abort) { --c->debug_peel_abort; t4[CTL_ABORT] = 1; }    // (tests: a launch that "gave up", to exercise the retry and the serial fall-back — c2a_debug_peel_abort)
    c->rb_edges = c->hrb[4]; c->rb_dup = c->hrb[5];      // (ride along: one round trip)
    const bool need_deep = c->hrb[6] != 0 && !c->peel_deep;
    if (want_stats) {
        ull st[32];
        HIP_TRY(hipMemcpy(st, c->peel_prof.p, 256, hipMemcpyDeviceToHost));
        std::fprintf(stderr, "[c2a peel stats] waves %u, hand-off arrays %u | seeds %llu, received %llu, pushed %llu, processed %llu, idle polls %llu, record re-reads %llu | busy %.1f ms-waves, idle %.1f ms-waves\n",
                     waves, A.n_fifos, st[5], st[0], st[2], st[6], st[1], (ull)t4[CTL_REREADS], st[3] / 1e5, st[4] / 1e5);
        {
            // hand-off latency: the pusher noted when it decided to push (the top of its step), the receiver when it had the
            // gate ready to issue (100 MHz clock)
            std::vector<ull> tq(slots), tp(slots);
            HIP_TRY(hipMemcpy(tq.data(), cold.q_time, slots * 8, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(tp.data(), cold.p_time, slots * 8, hipMemcpyDeviceToHost));
            const double edges_us[] = {0.5, 1, 1.5, 2, 3, 4, 6, 8, 16, 1e30};
            ull hist[10] = {0}, cnt = 0, late = 0;
            double sum = 0;
            for (size_t i = 0; i < slots; ++i) {
                if (!tq[i] || !tp[i]) continue;
                if (tp[i] < tq[i]) { ++late; continue; }
                const double us = (double)(tp[i] - tq[i]) / 100.0;
                int k = 0;
                while (us >= edges_us[k]) ++k;
                ++hist[k]; ++cnt; sum += us;
            }
            std::fprintf(stderr, "[c2a peel stats] hand-off (push decided -> received gate ready to issue): mean %.2f us over %llu;", cnt ? sum / cnt : 0.0, cnt);
            const char* names[] = {"<0.5", "<1", "<1.5", "<2", "<3", "<4", "<6", "<8", "<16", ">=16"};
            for (int k = 0; k < 10; ++k) std::fprintf(stderr, " %s us: %llu", names[k], hist[k]);
            std::fprintf(stderr, " (receiver waiting before the push was decided: %llu)\n", late);
        }
        if (trace_dir) {
            // per gate: step start (<< 2 | how it came) and record stored (100 MHz clock), + the tree records, for tools/peel_trace.py
            std::vector<ull> tr((size_t)n * 3);
            std::vector<u32> mt((size_t)n * 4);
            HIP_TRY(hipMemcpy(tr.data(), c->peel_trace.p, (size_t)n * 24, hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(mt.data(), c->meta.p, (size_t)n * 16, hipMemcpyDeviceToHost));
            const std::string base(trace_dir);
            if (FILE* f = std::fopen((base + "/peel_trace.bin").c_str(), "wb")) { std::fwrite(tr.data(), 8, tr.size(), f); std::fclose(f); }
            if (FILE* f = std::fopen((base + "/peel_meta.bin").c_str(), "wb")) { std::fwrite(mt.data(), 4, mt.size(), f); std::fclose(f); }
        }
        if (st[13]) std::fprintf(stderr, "[c2a peel stats] wait at the top of a step (ns): ticket %.0f, then static data %.0f, then records %.0f\n",
                     (double)(st[7] & 0xFFFFFFFFull) * 10.0 / st[13], (double)(st[7] >> 32) * 10.0 / st[13], st[16] * 10.0 / st[13]);
        if (st[13]) std::fprintf(stderr, "[c2a peel stats] per chain step (ns): wait for tickets/static data/records %.0f, issue of the next step %.0f, tournament %.0f, record + stores %.0f | steps without a load %.1f %% | chain start (static loads) %.0f ns per chain\n",
                     st[9] * 10.0 / st[13], st[10] * 10.0 / st[13], st[11] * 10.0 / st[13], st[12] * 10.0 / st[13], 100.0 * st[14] / st[13], st[15] * 10.0 / (st[5] + st[0] + 1));
        if (st[13]) std::fprintf(stderr, "[c2a peel stats] of the tournament phase: %.0f ns per step go to writing hand-off entries (%.0f ns per entry, its wait for the tickets included); %.2f %% of the steps read the consumer list itself (cold)\n",
                     st[17] * 10.0 / st[13], st[17] * 10.0 / (st[2] + 1), 100.0 * st[18] / st[13]);
        if (st[2]) std::fprintf(stderr, "[c2a peel stats] per hand-off entry: %.0f ns waiting for its two tickets, %.0f ns writing it\n", st[19] * 10.0 / st[2], (st[17] - st[19]) * 10.0 / st[2]);
    }
    c->peel_gave_up = t4[CTL_ABORT] != 0;
    if (need_deep) { c->peel_deep = true; return kRerunDeep; }      // (not a failure: the caller runs the DEEP build of the launch on clean buffers)
    if (t4[CTL_ABORT]) return fail(c, C2A_ERR_HIP, "dataflow peel: watchdog tripped (" + std::to_string(t4[CTL_ABORT]) + " waves gave up waiting)");
    c->node_clear = false;
    *peeled_out = t4[CTL_PROCESSED];
    c->stats.levels = t4[CTL_PROCESSED] ? t4[CTL_MAXLEVEL] + 1 : 0;
    c->stats.level_launches = 2;
    c->stats.peel_waves = waves;
    c->stats.peel_rereads = t4[CTL_REREADS];
    c->stats.n_relays = c->hrb[7];
    return C2A_OK;
}

// jump launches that make a splitter list of S elements complete (the reach grows kJumpSpan-fold per launch)
inline u32 jump_rounds(u64 S) {
    u32 rounds = 0;
    for (u64 reach = 1; reach < S; reach *= kJumpSpan) ++rounds;
    return rounds;
}

// The order stage, QUEUED behind the peel without a host round trip: its first kernels go by SC_PEELOK (raised by the first of them, k_root_bits) and do
// nothing when the launch gave up or left gates behind; the walk and the jumps then find no splitter.  The number of jump
// launches comes from an upper bound on the splitter count (a hash of the element number picks one element in 64: the mean
// + 8 standard deviations; order_result checks the real count and queues what is missing — surplus launches only copy).
// The host waits for the event behind the posted words, i.e. while the walk and the jumps run.
int order_launch(c2a_ctx* c) {
    const u32 n = c->n;
    hipStream_t s = c->stream;
    const u32 G = grid_for(n, 4096);
    const BuildRegions R = build_regions(c);
    const u32* ok = c->scalars.as<u32>() + SC_PEELOK;
    // DFS roots (tree nodes without a parent) in ascending ORIGINAL gate id: a bit per id, a scan over the bitmap's words, the list
    const u32 W = (n + 31u) / 32u;
    // (the first launch of the stage also posts the dataflow launch's numbers to the host and raises *ok for the launches behind it)
    ENSURE(c->rflag, ((size_t)n / 64 + 2) * 8);
    C2A_LAUNCH(k_root_bits, G, kThreads, s, n, (const u32*)c->pctl.as<u32>(), (const u32*)(c->cons_off.as<u32>() + n), (const u32*)(c->scalars.as<u32>() + SC_DUP),
               (const u32*)(c->scalars.as<u32>() + SC_RELAYS), c->hrb_dev, c->scalars.as<u32>() + SC_PEELOK, (const uint4*)c->meta.as<uint4>(), (const u32*)c->orig.as<u32>(),
               c->rbits.as<u32>(), c->rflag.as<u64>(), c->scalars.as<u32>() + SC_MAXDEPTH);
    int r = scan_1pass<1>(c, s, c->scan_tmp, W, ScanPopc{c->rbits.as<u32>()}, c->rpre.as<u32>(), (u32*)nullptr, c->scan_desc.as<char>() + R.roots);
    if (r) return r;
    const u32* n_roots_p = c->rpre.as<u32>() + W;
    C2A_LAUNCH_NOSYNC(k_root_list, G, kThreads, s, n, ok, (const u64*)c->rflag.as<u64>(), (const u32*)c->orig.as<u32>(), (const u32*)c->rbits.as<u32>(),
                      (const u32*)c->rpre.as<u32>(), c->ridx.as<u32>(), c->rlist.as<u32>());
    C2A_LAUNCH_NOSYNC(k_euler_next, G, kThreads, s, n, ok, (const u64*)c->rflag.as<u64>(), (const u32*)c->child.as<u32>(),
                      (const u32*)c->ridx.as<u32>(), (const u32*)c->rlist.as<u32>(), n_roots_p, c->next.as<u32>());
    const u32 m = 2 * n;
    u32* scount = c->scalars.as<u32>() + SC_SCOUNT;
    C2A_LAUNCH(k_rank_mark, std::max<u32>(1u, std::min<u32>(2048u, (m + kThreads * 8 - 1) / (kThreads * 8))), kThreads, s, m, ok, c->rlist.as<u32>(), scount, c->slist.as<u32>(),
                      c->owner.as<u32>());
    static_assert(SC_SCOUNT == SC_MAXDEPTH + 1, "read as a pair");
    C2A_LAUNCH_NOSYNC(k_post_words, 1, 64, s, c->hrb_dev + 8, (const u32*)(c->scalars.as<u32>() + SC_MAXDEPTH), 2u, n_roots_p, 1u, (const u32*)nullptr, 0u);
    HIP_TRY(hipEventRecord(c->ev[EV_ORDER_RB], s));
    const u64 expect = (u64)m >> (32 - C2A_SPLIT_SHIFT);
    u64 sd = 1;
    while (sd * sd < expect) ++sd;
    const u64 bound = expect + 8 * sd + 64;
    C2A_LAUNCH_NOSYNC(k_rank_walk, grid_for(expect + 1024u, 8192), kThreads, s, (const u32*)scount, c->rlist.as<u32>(), c->slist.as<u32>(),
                      c->next.as<u32>(), (const u32*)c->owner.as<u32>(), c->local.as<u64>(), c->sjump.as<uint2>());
    // pointer jumping, ping-pong between the two {next, sum} arrays
    c->jump_rounds = jump_rounds(bound);
    c->jump_a = c->sjump.as<uint2>(); c->jump_b = c->sjump2.as<uint2>();
    for (u32 k = 0; k < c->jump_rounds; ++k) {
        C2A_LAUNCH_NOSYNC(k_rank_jump, grid_for(bound, 4096), kThreads, s, (const u32*)scount, (const uint2*)c->jump_a, c->jump_b);
        std::swap(c->jump_a, c->jump_b);
    }
    return C2A_OK;
}

// behind the host's wait for EV_ORDER_RB (and a peel that ended well): the stage's numbers, the jump launches an unlikely
// splitter count still needs, and — unless the emission will write it — the sorted order
int order_result(c2a_ctx* c, bool defer_sorted) {
    const u32 n = c->n;
    hipStream_t s = c->stream;
    const u32 G = grid_for(n, 4096);
    const u32 sc[3] = {c->hrb[8], c->hrb[9], c->hrb[10]};      // depth of the DFS forest, splitters, roots
    const u32 S = sc[1];
    c->stats.max_depth = sc[0];
    c->stats.n_splitters = S;
    c->stats.n_roots = sc[2];
    // one writer per node (what the reference's front-end builds) => the positional numbering, which needs no walk in sorted
    // order (c2a_kernels.h POSITIONAL NUMBERING)
    c->positional = !c->has_dup && !c->numbering_walk;
    c->stats.numbering_path = c->positional ? 1u : 0u;
    for (u32 need = jump_rounds(S); c->jump_rounds < need; ++c->jump_rounds) {
        C2A_LAUNCH_NOSYNC(k_rank_jump, grid_for(S, 4096), kThreads, s, (const u32*)(c->scalars.as<u32>() + SC_SCOUNT), (const uint2*)c->jump_a, c->jump_b);
        std::swap(c->jump_a, c->jump_b);
    }
    const uint2* vl_a = c->jump_a;
    c->rank_suffix = vl_a;
    c->sorted_ready = !(defer_sorted && c->positional);
    if (!c->sorted_ready) return C2A_OK;            // (c2a_build_circuit: the emission's records carry the sorted order along)
    // (the tour's next[] has been consumed by the walk: its buffer takes the packed {rank, original id} pairs)
    C2A_LAUNCH_NOSYNC(k_rank_final, G, kThreads, s, n, (const u64*)c->local.as<u64>(), vl_a, (const u32*)c->orig.as<u32>(), c->next.as<uint2>());
    C2A_LAUNCH_NOSYNC(k_sorted_split, G, kThreads, s, n, (const uint2*)c->next.as<uint2>(), c->sorted_r.as<u32>(), c->sorted.as<u32>());
    return C2A_OK;
}

// the reference's own DFS on one lane (original gate ids; needs prod1 / orig of do_prep): sorted[] or the cycle's gate
int run_serial_dfs(c2a_ctx* c, u32* status, u64* cycle_at) {
    const u32 n = c->n;
    hipStream_t s = c->stream;
    ENSURE(c->dfs_state, (size_t)n);
    ENSURE(c->dfs_stack, (size_t)n * 4);
    HIP_TRY(hipMemsetAsync(c->dfs_state.p, 0, (size_t)n, s));
    C2A_LAUNCH_NOSYNC(k_serial_dfs, 1, 64, s, n, c->lh.as<u32>(), c->rh.as<u32>(), (const u32*)c->prod1.as<u32>(), (const u32*)c->orig.as<u32>(), c->dfs_state.as<u8>(),
                      c->dfs_stack.as<u32>(), c->sorted.as<u32>(), c->scalars.as<u32>() + SC_DFS);
    u32 res[3] = {0, 0, 0};
    int r = read_scalars(c, res, SC_DFS, 3);
    if (r) return r;
    *status = res[0];
    *cycle_at = res[1];
    return C2A_OK;
}

// behind a serial sort: the order in rank space (what the numbering kernels walk) and — with_levels — the reverse Kahn levels
// the level-parallel evaluators schedule by (the dataflow launch leaves them in meta[]; here one more serial walk)
int after_serial_sort(c2a_ctx* c, bool with_levels) {
    const u32 n = c->n;
    hipStream_t s = c->stream;
    const u32 G = grid_for(n, 4096);
    u32* inv = c->next.as<u32>();                  // (scratch of the order stage, idle on this path)
    C2A_LAUNCH_NOSYNC(k_invert, G, kThreads, s, n, (const u32*)c->orig.as<u32>(), inv);
    C2A_LAUNCH_NOSYNC(k_sorted_to_rank, G, kThreads, s, n, (const u32*)c->sorted.as<u32>(), (const u32*)inv, c->sorted_r.as<u32>());
    if (!with_levels) return C2A_OK;
    u32* lvl = c->owner.as<u32>();
    HIP_TRY(hipMemsetAsync(lvl, 0, (size_t)n * 4, s));
    C2A_LAUNCH_NOSYNC(k_serial_levels, 1, 64, s, n, (const u32*)c->sorted_r.as<u32>(), (const u32*)c->dep0.as<u32>(), (const u32*)c->dep1.as<u32>(), lvl,
                      c->meta.as<uint4>(), c->scalars.as<u32>() + SC_LEVELS);
    u32 mx = 0;
    int r = read_scalars(c, &mx, SC_LEVELS, 1);
    if (r) return r;
    c->stats.levels = mx + 1;
    return C2A_OK;
}

int do_topo_sort(c2a_ctx* c, u64* cycle_at, bool defer_sorted = false) {
    if (c->stage < ST_LOADED || c->circuit_only) return fail(c, C2A_ERR_STATE, "c2a_topo_sort: no gates loaded");
    c->stage = ST_LOADED;
    c->bool_planned = false; c->fmt_chunk_valid = false; c->pruned = false; c->gathered = false;
    c->peel_meta_valid = false;
    c->positional = false; c->sorted_ready = true; c->emitted_with_wires = false;
    const u32 n = c->n;
    std::memset(c->ev_valid, 0, sizeof(c->ev_valid));
    c->stats = c2a_stats{};
    c->stats.n_gates = n;
    if (cycle_at) *cycle_at = 0;
    if (n == 0) {                                    // (no gates: no levels, and that is valid level data)
        int r0 = clear_for_build(c);
        if (r0 || (r0 = mark_io(c))) return r0;
        c->stage = ST_SORTED; c->peel_meta_valid = true;
        return C2A_OK;
    }
    rec(c, EV_PREP0);
    int r = do_prep(c);
    if (r) return r;
    rec(c, EV_PREP1);
    u32 peeled = 0;
    c->serial_fallback = false;
    for (int attempt = 0;; ++attempt) {
        if ((r = peel_launch(c))) return r;
        rec(c, EV_PEEL1);
        if ((r = order_launch(c))) return r;
        HIP_TRY(hipEventSynchronize(c->ev[EV_ORDER_RB]));       // (the walk and the jumps run meanwhile)
        r = peel_result(c, &peeled);
        const bool rerun_deep = r == kRerunDeep;
        if (!rerun_deep && !(r == C2A_ERR_HIP && c->peel_gave_up)) break;
        if (rerun_deep) --attempt;                              // (the plain launch found the tree deeper than one chunk: the DEEP one has not been tried yet)
        if (attempt <= 0) {
            // the launch's watchdog tripped (a wave waited too long for a record or for global progress) — or the plain build asked for
            // the DEEP one —: once more on clean buffers (node records re-zeroed, tickets and child pointers reset) before the serial DFS
            // takes over
            if (!rerun_deep) std::fprintf(stderr, "[c2a] the dataflow peel gave up (%s); retrying once on clean buffers\n", c->err.c_str());
            HIP_TRY(hipMemsetAsync(c->fill.p, 0, (size_t)n_all_of(c->n) * 4, c->stream));
            HIP_TRY(hipMemsetAsync(c->child.p, 0xFF, (size_t)n_all_of(c->n) * 8, c->stream));
            HIP_TRY(hipMemsetAsync(c->scalars.as<u32>() + SC_MAXDEPTH, 0, 8, c->stream));
            int rc = clear_for_build(c, true);
            if (rc) return rc;
            continue;
        }
        // The dataflow launch gave up twice (its watchdog: a wave waited too long).  The reference's sort cannot fail on an
        // acyclic graph (topological_sort.rs:3-21), and neither may this one: the same DFS on one lane (exact, always
        // terminates, slow), the order in rank space and the reverse Kahn levels for what comes behind.
        if (!c->fallback_logged) { std::fprintf(stderr, "[c2a] the dataflow peel gave up twice (%s): sorting with the serial DFS instead\n", c->err.c_str()); c->fallback_logged = true; }
        c->serial_fallback = true;
        break;
    }
    if (r == C2A_OK || c->serial_fallback) c->err.clear();      // (a retry that succeeded leaves no message behind)
    if (r && !c->serial_fallback) return r;
    c->stats.n_edges = c->rb_edges;                  // (read back with the launch's own counters)
    c->has_dup = c->rb_dup != 0;
    if (c->serial_fallback) {
        u32 status = 0;
        u64 at = 0;
        r = run_serial_dfs(c, &status, &at);
        if (r) return r;
        if (cycle_at) *cycle_at = at;
        if (status == 1) return fail(c, C2A_ERR_CYCLIC, "Cyclic dependency: detected at i=" + std::to_string(at));
        if ((r = after_serial_sort(c, true))) return r;
        rec(c, EV_ORDER1);
        c->stage = ST_SORTED;
        c->peel_meta_valid = true;
        return C2A_OK;
    }
    if (peeled != n) {
        // leftover gates sit on or above a dependency cycle: replay the reference's DFS for its message
        u32 status = 0;
        u64 at = 0;
        r = run_serial_dfs(c, &status, &at);
        if (r) return r;
        if (cycle_at) *cycle_at = at;
        if (status == 1) return fail(c, C2A_ERR_CYCLIC, "Cyclic dependency: detected at i=" + std::to_string(at));
        return fail(c, C2A_ERR_HIP, "internal: peel left gates behind but the serial DFS found no cycle");
    }
    r = order_result(c, defer_sorted);
    if (r) return r;
    rec(c, EV_ORDER1);
    c->stage = ST_SORTED;
    c->peel_meta_valid = true;
    return C2A_OK;
}

// the two words do_assign_wires reads back (the stream must have been synchronized)
int finish_wires(c2a_ctx* c) {
    c->rb_nmid = c->hrb[16] + c->nmid_add; c->rb_err = c->hrb[17];      // (k_post_words of do_assign_wires; the stream has drained since)
    if (c->positional) c->stats.numbering_events = c->hrb[18];
    if (c->rb_err) { c->stage = ST_SORTED; return fail(c, C2A_ERR_INCONSISTENCY, "Inconsistency: a node is used for both input and output"); }
    c->n_mid = c->rb_nmid;
    c->wire_count = c->n_in + c->rb_nmid + c->n_out;
    return C2A_OK;
}

int do_assign_wires(c2a_ctx* c, bool defer_readback = false) {
    if (c->stage < ST_SORTED || c->circuit_only) return fail(c, C2A_ERR_STATE, "c2a_assign_wires: call c2a_topo_sort first");
    const u32 n = c->n;
    hipStream_t s = c->stream;
    const u64 m = (u64)n * 3;
    rec(c, EV_WIRES0);
    c->emitted_with_wires = false;
    // (the IO flags of the nodes and the in / out clash word are do_prep's: they do not change between the sort and here)
    // (k_relabel leaves node -> wire / first-seen as the numbering wants them; a second numbering of the same sorted circuit resets them itself)
    if (!c->node_init_valid) C2A_LAUNCH_NOSYNC(k_node_init, grid_for(std::max<u32>(1u, c->n_nodes), 4096), kThreads, s, c->n_nodes, c->node_wire1.as<u32>(), c->first.as<u32>());
    c->node_init_valid = false;
    const bool inputs_ride = c->positional && c->n;      // (the positional numbering's first launch takes the input wires along)
    if (c->n_in && !inputs_ride) C2A_LAUNCH_NOSYNC(k_input_wires, grid_for(c->n_in, 1024), kThreads, s, c->n_in, c->in_nodes.as<u32>(), c->node_wire1.as<u32>());
    const u32 G = grid_for(n, 4096);
    // (the first-seen minimum looks before it draws: a thread must go round several times for the look to find something — eight rounds;
    // in ONE round every reader of a shared constant looks at the same time, sees nothing and draws: 123 000 readers in a 410 000-gate
    // circuit were 0.85 ms)
    const u32 G8 = std::min<u32>(G, grid_for(((u64)n + 7) / 8, 4096));
    const u32* n_mid_p;
    int r;
    const u32 PW = (n + 31u) / 32u;                 // 32-position blocks of the event bits
    const u32* n_events_p = nullptr;
    c->nmid_add = 0;
    if (c->positional && n) {
        // POSITIONAL NUMBERING (c2a_kernels.h): positions, the event bits that shift the numbering, then wires AND gates by formula
        const PosSrc S{n, c->local.as<u64>(), c->rank_suffix, c->pos_r.as<u32>()};
        // (zeroed by the build's k_clear — a numbering that is run again on the same sorted circuit zeroes them itself)
        const bool cleared = c->wires_clear_valid;
        c->wires_clear_valid = false;
        if (!cleared) HIP_TRY(hipMemsetAsync(c->pblk.p, 0, (size_t)PW * 16, s));
        if (c->sorted_ready) {                          // (the staged calls: positions = the inverse of the order the caller has been given)
            C2A_LAUNCH_NOSYNC(k_eval_inverse, G, kThreads, s, n, c->sorted_r.as<u32>(), c->pos_r.as<u32>());
            C2A_LAUNCH_NOSYNC(k_pos_first<true>, G8, kThreads, s, S, (const u8*)c->gflag.as<u8>(), (const uint4*)c->gate4.as<uint4>(), c->pos_r.as<u32>(), c->first.as<u32>(),
                              c->n_in, (const u32*)c->in_nodes.as<u32>(), c->node_wire1.as<u32>());
        } else
            C2A_LAUNCH_NOSYNC(k_pos_first<false>, G8, kThreads, s, S, (const u8*)c->gflag.as<u8>(), (const uint4*)c->gate4.as<uint4>(), c->pos_r.as<u32>(), c->first.as<u32>(),
                              c->n_in, (const u32*)c->in_nodes.as<u32>(), c->node_wire1.as<u32>());
        C2A_LAUNCH_NOSYNC(k_pos_bits, G, kThreads, s, n, (const u8*)c->gflag.as<u8>(), (const uint4*)c->gate4.as<uint4>(), (const u32*)c->pos_r.as<u32>(), (const u32*)c->first.as<u32>(),
                          c->pblk.as<u32>());
        r = scan_1pass<2>(c, s, c->scan_tmp, PW, ScanPosBits{c->pblk.as<uint4>()}, c->dpre.as<u32>(), c->epre.as<u32>(),
                          cleared ? (void*)(c->scan_desc.as<char>() + build_regions(c).bits) : nullptr);
        if (r) return r;
        n_mid_p = c->dpre.as<u32>() + PW;               // (the net shift of all events: the walk hands out n + that many wires, compiler.rs:440-441)
        n_events_p = c->epre.as<u32>() + PW;
        c->nmid_add = n;
    } else {
        if (n) {
            C2A_LAUNCH_NOSYNC(k_first_seen, G8, kThreads, s, n, c->sorted_r.as<u32>(), (const uint4*)c->gate4.as<uint4>(),
                              (const u32*)c->prod1.as<u32>(), (const u32*)(c->scalars.as<u32>() + SC_DUP), c->first.as<u32>());
            C2A_LAUNCH_NOSYNC(k_new_wire_flags, G, kThreads, s, n, c->sorted_r.as<u32>(), (const uint4*)c->gate4.as<uint4>(),
                              c->first.as<u32>(), c->nflag.as<u8>(), (const u32*)c->prod1.as<u32>(),
                              (const u32*)(c->scalars.as<u32>() + SC_DUP), c->wflag.as<u32>());
        }
        r = scan_exclusive<u32>(c, c->wflag.as<u32>(), c->widx.as<u32>(), m);
        if (r) return r;
        if (n) {
            C2A_LAUNCH_NOSYNC(k_assign_wires, G, kThreads, s, n, c->sorted_r.as<u32>(),
                              (const uint4*)c->gate4.as<uint4>(), c->wflag.as<u32>(), c->widx.as<u32>(), c->n_in,
                              c->node_wire1.as<u32>());
        }
        n_mid_p = c->widx.as<u32>() + m;
    }
    if (c->n_out)
        C2A_LAUNCH_NOSYNC(k_assign_outputs, grid_for(c->n_out, 1024), kThreads, s, c->n_out, c->out_nodes.as<u32>(), c->n_in, c->nmid_add,
                          n_mid_p, c->node_wire1.as<u32>());
    rec(c, EV_WIRES1);
    C2A_LAUNCH_NOSYNC(k_post_words, 1, 64, s, c->hrb_dev + 16, n_mid_p, 1u, (const u32*)(c->scalars.as<u32>() + SC_ERR), 1u, n_events_p, n_events_p ? 1u : 0u);
    if (c->positional && n) {
        // the gates' own out wires are part of node -> wire, and the formula that gives them gives in0 / in1 as well: the emission
        // belongs to this stage (do_emit finds it done)
        rec(c, EV_EMIT0);
        const PosBits T{c->pblk.as<uint4>(), c->dpre.as<u32>()};
        C2A_LAUNCH_NOSYNC(k_pos_rank, G, kThreads, s, n, c->n_in, (const u8*)c->gflag.as<u8>(), (const uint4*)c->gate4.as<uint4>(), (const u32*)c->pos_r.as<u32>(), T,
                          c->node_wire1.as<u32>(), c->wire_r.as<u32>());
        C2A_LAUNCH_NOSYNC(k_emit_rank, G, kThreads, s, n, (const uint4*)c->gate4.as<uint4>(), (const u32*)c->dep0.as<u32>(), (const u32*)c->dep1.as<u32>(),
                          (const u32*)c->orig.as<u32>(), (const u32*)c->pos_r.as<u32>(), (const u32*)c->wire_r.as<u32>(), c->node_wire1.as<u32>(), c->erec.as<EmitRec>());
        if (c->sorted_ready)
            C2A_LAUNCH_NOSYNC(k_emit_split<false>, G, kThreads, s, n, c->n_in, (const EmitRec*)c->erec.as<EmitRec>(), (const uint4*)c->gate4.as<uint4>(), (const u32*)c->node_wire1.as<u32>(), T, c->e_in0.as<u32>(), c->e_in1.as<u32>(), c->e_out.as<u32>(),
                              c->e_op.as<u8>(), c->sorted_r.as<u32>(), c->sorted.as<u32>());
        else
            C2A_LAUNCH_NOSYNC(k_emit_split<true>, G, kThreads, s, n, c->n_in, (const EmitRec*)c->erec.as<EmitRec>(), (const uint4*)c->gate4.as<uint4>(), (const u32*)c->node_wire1.as<u32>(), T, c->e_in0.as<u32>(), c->e_in1.as<u32>(), c->e_out.as<u32>(),
                              c->e_op.as<u8>(), c->sorted_r.as<u32>(), c->sorted.as<u32>());
        rec(c, EV_EMIT1);
        c->sorted_ready = true;
        c->emitted_with_wires = true;
    }
    c->stage = ST_WIRED;                             // (the emission may be queued behind this; finish_wires() makes it official)
    if (defer_readback) return C2A_OK;
    HIP_TRY(hipStreamSynchronize(s));
    return finish_wires(c);
}

int do_emit(c2a_ctx* c) {
    if (c->stage < ST_WIRED) return fail(c, C2A_ERR_STATE, "c2a_emit_gates: call c2a_assign_wires first");
    if (c->emitted_with_wires) { c->stage = ST_EMITTED; return C2A_OK; }      // (positional numbering: done with the wires)
    rec(c, EV_EMIT0);
    if (c->n)
        C2A_LAUNCH_NOSYNC(k_emit, grid_for(c->n, 4096), kThreads, c->stream, c->n, c->sorted_r.as<u32>(),
                          (const uint4*)c->gate4.as<uint4>(), c->node_wire1.as<u32>(), c->e_in0.as<u32>(), c->e_in1.as<u32>(),
                          c->e_out.as<u32>(), c->e_op.as<u8>());
    rec(c, EV_EMIT1);
    c->stage = ST_EMITTED;
    return C2A_OK;
}

float elapsed(c2a_ctx* c, Ev a, Ev b) {
    if (!c->ev_valid[a] || !c->ev_valid[b]) return 0.f;
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, c->ev[a], c->ev[b]) != hipSuccess) return 0.f;
    return ms;
}

int copy_out(c2a_ctx* c, void* host, const void* dev, size_t bytes) {
    if (!host || !bytes) return C2A_OK;
    HIP_TRY(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    return C2A_OK;
}

}  // namespace

#ifndef C2A_EMULATE
// The runtime loads the code object and resolves a kernel on its FIRST launch: for a one-shot caller (the reference calls
// build_circuit once per process, src/main.rs:28-32) that was 2.7 ms inside the first build's first stage.  Asking for the
// attributes of the build's kernels does the same work here, next to the 100 ms the HIP context takes to come up.
void warm_functions() {
    hipFuncAttributes a;
#define C2A_WARM(k) (void)hipFuncGetAttributes(&a, reinterpret_cast<const void*>(&k))
    C2A_WARM(k_clear); C2A_WARM(k_validate); C2A_WARM(k_mark_inputs); C2A_WARM(k_mark_outputs); C2A_WARM(k_producer); C2A_WARM(k_relabel); C2A_WARM(k_dup_clear);
    C2A_WARM(k_dup_producer); C2A_WARM(k_deps); C2A_WARM(k_gstat); C2A_WARM(k_peel_sinks); C2A_WARM(k_peel_shallow);
    C2A_WARM((k_peel<false, false>)); C2A_WARM(k_root_bits); C2A_WARM(k_root_list); C2A_WARM(k_euler_next); C2A_WARM(k_rank_mark);
    C2A_WARM(k_post_words); C2A_WARM(k_rank_walk); C2A_WARM(k_rank_jump); C2A_WARM(k_rank_final); C2A_WARM(k_sorted_split); C2A_WARM(k_node_init);
    C2A_WARM(k_input_wires); C2A_WARM(k_pos_first<false>); C2A_WARM(k_pos_first<true>); C2A_WARM(k_pos_bits); C2A_WARM(k_assign_outputs); C2A_WARM(k_pos_rank);
    C2A_WARM(k_emit_rank); C2A_WARM(k_emit_split<true>); C2A_WARM(k_emit_split<false>); C2A_WARM(k_unbias);
    C2A_WARM((k_scan_stream<1, ScanFromU32, u32>)); C2A_WARM((k_scan_stream<1, ScanPopc, u32>)); C2A_WARM((k_scan_stream<2, ScanPosBits, u32>));
    C2A_WARM((k_scan_stream<2, ScanBoolSizes, u64>)); C2A_WARM((k_boolify<256, 1024>));
#undef C2A_WARM
}
#endif

// ------------------------------------------------------------------------------------------------
extern "C" {

const char* c2a_version(void) {
#ifdef C2A_EMULATE
    return "c2a 0.3 (host emulation build — tests only)";
#else
    return "c2a 0.3 (hip gfx950)";
#endif
}

int c2a_abi_version(void) { return C2A_ABI_VERSION; }

int c2a_visible_devices(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count < 0) return 0;
    return count;
}

int c2a_create(int n_devices, const int* device_ids, c2a_ctx** out) {
    if (!out || n_devices < 1 || n_devices > 64 || !device_ids) return C2A_ERR_ARG;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) return C2A_ERR_HIP;
    for (int i = 0; i < n_devices; ++i)
        if (device_ids[i] < 0 || device_ids[i] >= count) return C2A_ERR_ARG;
    const int device_id = device_ids[0];
    if (hipSetDevice(device_id) != hipSuccess) return C2A_ERR_HIP;
    c2a_ctx* c = new (std::nothrow) c2a_ctx();
    if (!c) return C2A_ERR_NOMEM;
    c->device = device_id;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0)
        c->n_cu = prop.multiProcessorCount;
    if (const char* e = std::getenv("C2A_BOOL_SLICES")) { const u32 v = (u32)std::strtoul(e, nullptr, 10); if (v >= 1 && v <= 1024) c->bool_slices = v; }
    if (const char* e = std::getenv("C2A_BOOL_CHUNK")) { const u32 v = (u32)std::strtoul(e, nullptr, 10); if (v == 128 || v == 256 || v == 512) c->bool_chunk = v; }
    if (const char* e = std::getenv("C2A_PEEL_SEED_CHUNK")) { const u32 v = (u32)std::strtoul(e, nullptr, 10); if (v >= 1 && v <= 4096) c->peel_seed_chunk = v; }
    if (const char* e = std::getenv("C2A_PEEL_SHALLOW")) { const u32 v = (u32)std::strtoul(e, nullptr, 10); if (v >= 1 && v <= 48) c->peel_shallow = v; }
    if (const char* e = std::getenv("C2A_PEEL_WAVES")) { const u32 v = (u32)std::strtoul(e, nullptr, 10); if (v >= 1 && v <= 32) { c->peel_waves = v; c->peel_waves_forced = true; } }
    if (const char* e = std::getenv("C2A_PEEL_RESERVE")) { const u32 v = (u32)std::strtoul(e, nullptr, 10); if (v <= 32) { c->peel_reserve = v; c->peel_reserve_forced = true; } }
    if (const char* e = std::getenv("C2A_PEEL_RELEASE")) { const u32 v = (u32)std::strtoul(e, nullptr, 10); if (v >= 1) c->peel_release = v; }
    if (const char* e = std::getenv("C2A_NUMBERING_WALK")) c->numbering_walk = e[0] == '1';
    if (const char* e = std::getenv("C2A_PEEL_FIFOS")) { const u32 v = (u32)std::strtoul(e, nullptr, 10); if (v >= 1 && v <= 64 && (v & (v - 1)) == 0) c->peel_fifos = v; }
    if (hipStreamCreate(&c->stream) != hipSuccess) { delete c; return C2A_ERR_HIP; }
    if (hipStreamCreate(&c->aux) != hipSuccess) { c->aux = hipStream_t{}; c2a_destroy(c); return C2A_ERR_HIP; }
    {
        void* h = nullptr; void* d = nullptr;
        if (hipHostMalloc(&h, 256 * sizeof(u32), hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) { c2a_destroy(c); return C2A_ERR_HIP; }
        c->hrb = static_cast<u32*>(h);
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess) { c2a_destroy(c); return C2A_ERR_HIP; }
        c->hrb_dev = static_cast<u32*>(d);
    }
    for (int i = 0; i < EV_COUNT; ++i)
        if (hipEventCreate(&c->ev[i]) != hipSuccess) { c2a_destroy(c); return C2A_ERR_HIP; }
    // devices 1..N-1: one stream each (the same device may be listed twice: it then simply gets two shards)
    c->peers.resize((size_t)n_devices - 1);
    for (int i = 1; i < n_devices; ++i) {
        PeerDev& P = c->peers[(size_t)i - 1];
        P.device = device_ids[i];
        if (hipSetDevice(P.device) != hipSuccess || hipStreamCreate(&P.stream) != hipSuccess) { (void)hipSetDevice(device_id); c2a_destroy(c); return C2A_ERR_HIP; }
    }
    (void)hipSetDevice(device_id);
    // the primary reads the peers' pieces of the boolean circuit in place (c2a_eval, c2a_boolify_prune): map them if the devices allow it
    c->peer_access = !c->peers.empty();
    for (PeerDev& P : c->peers) {
        if (P.device == device_id) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, device_id, P.device) != hipSuccess || !can) { c->peer_access = false; break; }
        const hipError_t e = hipDeviceEnablePeerAccess(P.device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { c->peer_access = false; break; }
        (void)hipGetLastError();
    }
#ifndef C2A_EMULATE
    warm_functions();
#endif
    *out = c;
    return C2A_OK;
}

void c2a_destroy(c2a_ctx* c) {
    if (!c) return;
    for (PeerDev& P : c->peers) {
        (void)hipSetDevice(P.device);
        if (P.stream) { (void)hipStreamSynchronize(P.stream); }
        for (DevBuf* b : {&P.e_in0, &P.e_in1, &P.e_out, &P.e_op, &P.goff, &P.aoff, &P.tmpl, &P.tables, &P.b_in0, &P.b_in1, &P.b_out, &P.b_op, &P.acc,
                          &P.fmt_len, &P.fmt_off, &P.fmt_text, &P.fmt_table, &P.scan_tmp, &P.vscratch})
            if (b->p) (void)hipFree(b->p);
        if (P.stream) (void)hipStreamDestroy(P.stream);
    }
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    for (DevBuf* b : c->all) if (b->p) (void)hipFree(b->p);
    for (int i = 0; i < EV_COUNT; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    if (c->aux) { (void)hipStreamSynchronize(c->aux); (void)hipStreamDestroy(c->aux); }
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->hrb) (void)hipHostFree(c->hrb);
    delete c;
}

int c2a_device_count(const c2a_ctx* c) { return c ? 1 + (int)c->peers.size() : 0; }

const char* c2a_last_error(const c2a_ctx* c) { return c ? c->err.c_str() : "null context"; }

int c2a_load_gates(c2a_ctx* c, uint64_t n64, const uint32_t* lh, const uint32_t* rh, const uint32_t* out,
                   const uint8_t* op, uint32_t n_nodes, uint32_t n_in, const uint32_t* input_nodes, uint32_t n_out,
                   const uint32_t* output_nodes) {
    if (!c) return C2A_ERR_ARG;
    c->stage = ST_EMPTY;
    if (n64 >= (1ull << 31)) return fail(c, C2A_ERR_ARG, "c2a_load_gates: n must be < 2^31");
    if (n64 && (!lh || !rh || !out || !op)) return fail(c, C2A_ERR_ARG, "c2a_load_gates: null gate arrays");
    if ((n_in && !input_nodes) || (n_out && !output_nodes)) return fail(c, C2A_ERR_ARG, "c2a_load_gates: null IO lists");
    const u32 n = (u32)n64;
    // argument validation: the IO lists here, the gates on the device behind their copy (k_validate: ids must address the node
    // table; op must be an AGateType)
    for (u32 i = 0; i < n_in; ++i)
        if (input_nodes[i] >= n_nodes) return fail(c, C2A_ERR_ARG, "c2a_load_gates: input node id >= n_nodes");
    for (u32 i = 0; i < n_out; ++i)
        if (output_nodes[i] >= n_nodes) return fail(c, C2A_ERR_ARG, "c2a_load_gates: output node id >= n_nodes");
    HIP_TRY(hipSetDevice(c->device));
    c->n = n; c->n_nodes = n_nodes; c->n_in = n_in; c->n_out = n_out;
    c->bool_planned = false; c->fmt_chunk_valid = false; c->pruned = false; c->gathered = false; c->peel_meta_valid = false; c->stats = c2a_stats{}; c->binfo = c2a_bool_info{};
    c->io_clash = false;
    c->circuit_only = false;
    c->peel_deep = false;
    const size_t n4 = (size_t)n * 4, nn4 = (size_t)n_nodes * 4;
    ENSURE(c->lh, n4); ENSURE(c->rh, n4); ENSURE(c->out, n4); ENSURE(c->op, n); ENSURE(c->gate4, (size_t)n * 16);
    ENSURE(c->nrec, (size_t)n_nodes * 16); ENSURE(c->orig, n4);
    ENSURE(c->in_nodes, (size_t)n_in * 4); ENSURE(c->out_nodes, (size_t)n_out * 4);
    ENSURE(c->prod1, nn4); ENSURE(c->dep0, n4); ENSURE(c->dep1, n4); ENSURE(c->cons_cnt, n4);
    // (what the dataflow launch keeps per gate has room for the relays of the hubs behind the n gates: c2a_peel.h HUBS AND RELAYS)
    const size_t na = n_all_of(n), na4 = na * 4, rcap = (size_t)relay_cap(n);
    ENSURE(c->cons_off, n4 + 4); ENSURE(c->eslot, 2 * n4); ENSURE(c->fill, na4 + 16 + (size_t)kFillDummyStride * kFillDummyWaves * 4);
    ENSURE(c->meta, na * 16); ENSURE(c->gstat, na * 32); ENSURE(c->clist, 2 * n4 + 64 * 4 + (rcap + 64) * 4);
    ENSURE(c->node, na * kNodeWords * 8); ENSURE(c->child, 2 * na4);
    // a new graph needs clean node records (5 GB at 10 M gates, ~0.75 ms of HBM writes): cleared here, on a stream of its own,
    // beside the host-to-device copies below — a one-shot caller (the reference calls build_circuit once per process) never
    // waits for it, and a step on a loaded graph does not need it (the run tag alternates)
    c->node_clear = true;
    bool cleared = false;
    if (n && hipMemsetAsync(c->node.p, 0, na * kNodeWords * 8, c->aux) == hipSuccess) cleared = true;
    if (n) C2A_LAUNCH_NOSYNC(k_relay_list, grid_for(rcap, 1024), kThreads, c->aux, n, (u32)rcap, c->clist.as<u32>());      // (the relay part of the consumer lists: once per loaded graph)
    if (n_nodes) { HIP_TRY(hipMemsetAsync(c->nrec.p, 0, (size_t)n_nodes * 16, c->aux)); c->build_no = 0; }      // (no record of any build)
    ENSURE(c->rbits, n4 / 32 + 32); ENSURE(c->rpre, n4 / 32 + 16); ENSURE(c->ridx, n4); ENSURE(c->rlist, n4);
    ENSURE(c->next, 2 * n4); ENSURE(c->owner, 2 * n4); ENSURE(c->local, 2 * n4); ENSURE(c->slist, 2 * n4);
    ENSURE(c->sjump, 4 * n4); ENSURE(c->sjump2, 4 * n4);
    ENSURE(c->sorted, n4); ENSURE(c->sorted_r, n4);
    ENSURE(c->first, nn4); ENSURE(c->nflag, (size_t)n_nodes + 16); ENSURE(c->wflag, 3 * n4); ENSURE(c->widx, 3 * n4 + 4);
    ENSURE(c->node_wire1, nn4); ENSURE(c->node_wire, nn4);
    ENSURE(c->pos_r, n4); ENSURE(c->wire_r, n4); ENSURE(c->erec, (size_t)n * sizeof(EmitRec));
    ENSURE(c->pblk, ((size_t)n / 32 + 2) * 16); ENSURE(c->dpre, ((size_t)n / 32 + 4) * 4); ENSURE(c->epre, ((size_t)n / 32 + 4) * 4); ENSURE(c->gflag, (size_t)n + 16);
    ENSURE(c->e_in0, n4); ENSURE(c->e_in1, n4); ENSURE(c->e_out, n4); ENSURE(c->e_op, n);
    ENSURE(c->scalars, SC_WORDS * 4);
    hipStream_t s = c->stream;
    if (n) {
        HIP_TRY(hipMemcpyAsync(c->lh.p, lh, n4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->rh.p, rh, n4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->out.p, out, n4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->op.p, op, n, hipMemcpyHostToDevice, s));
    }
    if (n_in) HIP_TRY(hipMemcpyAsync(c->in_nodes.p, input_nodes, (size_t)n_in * 4, hipMemcpyHostToDevice, s));
    if (n_out) HIP_TRY(hipMemcpyAsync(c->out_nodes.p, output_nodes, (size_t)n_out * 4, hipMemcpyHostToDevice, s));
    // behind the copies: the gates validated and counted by type, and "a node is both an input and an output" — the reference
    // checks that BEFORE it sorts (compiler.rs:363-383 precede :408), so build_circuit must be able to report it first
    enum { LD_HIST = 0, LD_BAD = 32, LD_CLASH = 33, LD_WORDS = 34 };
    static_assert(LD_WORDS <= SC_WORDS && C2A_NUM_GATE_TYPES <= 32, "the scalars block holds the load's read-back");
    u32 ld[LD_WORDS];
    HIP_TRY(hipMemsetAsync(c->scalars.p, 0, SC_WORDS * 4, s));
    HIP_TRY(hipMemsetAsync(c->scalars.as<u32>() + LD_BAD, 0xFF, 4, s));
    if (n) C2A_LAUNCH(k_validate, grid_for(n, 2048), kThreads, s, n, (const u32*)c->lh.as<u32>(), (const u32*)c->rh.as<u32>(), (const u32*)c->out.as<u32>(), (const u8*)c->op.as<u8>(), n_nodes,
                      (u32)C2A_NUM_GATE_TYPES, c->scalars.as<u32>() + LD_BAD, c->scalars.as<u32>() + LD_HIST);
    if (n_in && n_out) {
        HIP_TRY(hipMemsetAsync(c->nflag.p, 0, (size_t)n_nodes, s));
        C2A_LAUNCH_NOSYNC(k_mark_inputs, grid_for(n_in, 1024), kThreads, s, n_in, c->in_nodes.as<u32>(), c->nflag.as<u8>());
        C2A_LAUNCH_NOSYNC(k_mark_outputs, grid_for(n_out, 1024), kThreads, s, n_out, c->out_nodes.as<u32>(), c->nflag.as<u8>(), c->scalars.as<u32>() + LD_CLASH);
    }
    HIP_TRY(hipMemcpyAsync(ld, c->scalars.p, sizeof(ld), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    HIP_TRY(hipStreamSynchronize(c->aux));
    if (ld[LD_BAD] != 0xFFFFFFFFu) {
        const u64 g = ld[LD_BAD];
        if (lh[g] >= n_nodes || rh[g] >= n_nodes || out[g] >= n_nodes)
            return fail(c, C2A_ERR_ARG, "c2a_load_gates: node id >= n_nodes at gate " + std::to_string(g));
        return fail(c, C2A_ERR_ARG, "c2a_load_gates: unknown gate type at gate " + std::to_string(g));
    }
    for (u32 t = 0; t < C2A_NUM_GATE_TYPES; ++t) c->op_hist[t] = ld[LD_HIST + t];
    c->io_clash = ld[LD_CLASH] != 0;
    if (cleared) { c->node_clear = false; c->peel_epoch = 0; }
    c->stage = ST_LOADED;
    return C2A_OK;
}

int c2a_load_circuit(c2a_ctx* c, uint64_t n64, const uint32_t* in0, const uint32_t* in1, const uint32_t* out, const uint8_t* op,
                     uint32_t wire_count, uint32_t n_in, uint32_t n_out) {
    if (!c) return C2A_ERR_ARG;
    c->stage = ST_EMPTY;
    if (n64 >= (1ull << 31)) return fail(c, C2A_ERR_ARG, "c2a_load_circuit: n must be < 2^31");
    if (n64 && (!in0 || !in1 || !out || !op)) return fail(c, C2A_ERR_ARG, "c2a_load_circuit: null gate arrays");
    if ((u64)n_in + n_out > wire_count) return fail(c, C2A_ERR_ARG, "c2a_load_circuit: more input + output wires than wires");
    const u32 n = (u32)n64;
    HIP_TRY(hipSetDevice(c->device));
    c->n = n; c->n_nodes = 0; c->n_in = n_in; c->n_out = n_out;
    c->bool_planned = false; c->fmt_chunk_valid = false; c->pruned = false; c->gathered = false; c->peel_meta_valid = false; c->stats = c2a_stats{}; c->binfo = c2a_bool_info{};
    c->stats.n_gates = n;
    c->io_clash = false; c->has_dup = false; c->positional = false; c->serial_fallback = false;
    std::memset(c->ev_valid, 0, sizeof(c->ev_valid));
    const size_t n4 = (size_t)n * 4;
    ENSURE(c->e_in0, n4); ENSURE(c->e_in1, n4); ENSURE(c->e_out, n4); ENSURE(c->e_op, n); ENSURE(c->scalars, SC_WORDS * 4);
    hipStream_t s = c->stream;
    if (n) {
        HIP_TRY(hipMemcpyAsync(c->e_in0.p, in0, n4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->e_in1.p, in1, n4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->e_out.p, out, n4, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->e_op.p, op, n, hipMemcpyHostToDevice, s));
    }
    // behind the copies: wire ids must address the wires, op must be an AGateType; the gates per type (the totals of a boolify plan)
    enum { LD_HIST = 0, LD_BAD = 32, LD_WORDS = 33 };
    u32 ld[LD_WORDS];
    HIP_TRY(hipMemsetAsync(c->scalars.p, 0, SC_WORDS * 4, s));
    HIP_TRY(hipMemsetAsync(c->scalars.as<u32>() + LD_BAD, 0xFF, 4, s));
    if (n) C2A_LAUNCH(k_validate, grid_for(n, 2048), kThreads, s, n, (const u32*)c->e_in0.as<u32>(), (const u32*)c->e_in1.as<u32>(), (const u32*)c->e_out.as<u32>(), (const u8*)c->e_op.as<u8>(), wire_count,
                      (u32)C2A_NUM_GATE_TYPES, c->scalars.as<u32>() + LD_BAD, c->scalars.as<u32>() + LD_HIST);
    HIP_TRY(hipMemcpyAsync(ld, c->scalars.p, sizeof(ld), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (ld[LD_BAD] != 0xFFFFFFFFu) {
        const u64 g = ld[LD_BAD];
        if (in0[g] >= wire_count || in1[g] >= wire_count || out[g] >= wire_count)
            return fail(c, C2A_ERR_ARG, "c2a_load_circuit: wire id >= wire_count at gate " + std::to_string(g));
        return fail(c, C2A_ERR_ARG, "c2a_load_circuit: unknown gate type at gate " + std::to_string(g));
    }
    for (u32 t = 0; t < C2A_NUM_GATE_TYPES; ++t) c->op_hist[t] = ld[LD_HIST + t];
    c->wire_count = wire_count; c->n_mid = wire_count - n_in - n_out;
    c->circuit_only = true; c->emitted_with_wires = true; c->sorted_ready = false;
    c->stage = ST_EMITTED;
    return C2A_OK;
}

int c2a_topo_sort(c2a_ctx* c, uint32_t* sorted, uint64_t* cycle_at) {
    if (!c) return C2A_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    int r = do_topo_sort(c, cycle_at);
    if (r) return r;
    r = copy_out(c, sorted, c->sorted.p, (size_t)c->n * 4);
    if (r) return r;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return C2A_OK;
}

int c2a_topo_sort_serial(c2a_ctx* c, uint32_t* sorted, uint64_t* cycle_at) {
    if (!c) return C2A_ERR_ARG;
    if (c->stage < ST_LOADED || c->circuit_only) return fail(c, C2A_ERR_STATE, "c2a_topo_sort_serial: no gates loaded");
    HIP_TRY(hipSetDevice(c->device));
    c->stage = ST_LOADED;
    c->bool_planned = false; c->fmt_chunk_valid = false; c->pruned = false; c->gathered = false; c->peel_meta_valid = false;
    c->positional = false; c->sorted_ready = true; c->emitted_with_wires = false;
    c->stats = c2a_stats{}; c->stats.n_gates = c->n;
    if (cycle_at) *cycle_at = 0;
    if (c->n == 0) { int r0 = clear_for_build(c); if (r0 || (r0 = mark_io(c))) return r0; c->stage = ST_SORTED; return C2A_OK; }
    // the producer map, the relabelling and the deps closure only (no peel)
    int r = do_prep(c, false);
    if (r) return r;
    u32 status = 0;
    u64 at = 0;
    r = run_serial_dfs(c, &status, &at);
    if (r) return r;
    {
        u32 dup = 0;
        HIP_TRY(hipMemcpyAsync(&dup, c->scalars.as<u32>() + SC_DUP, 4, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        c->has_dup = dup != 0;
    }
    if (status == 1) {
        if (cycle_at) *cycle_at = at;
        return fail(c, C2A_ERR_CYCLIC, "Cyclic dependency: detected at i=" + std::to_string(at));
    }
    if ((r = after_serial_sort(c, false))) return r;
    r = copy_out(c, sorted, c->sorted.p, (size_t)c->n * 4);
    if (r) return r;
    HIP_TRY(hipStreamSynchronize(c->stream));
    c->stage = ST_SORTED;
    return C2A_OK;
}

int c2a_assign_wires(c2a_ctx* c, uint32_t* node_to_wire, uint32_t* wire_count) {
    if (!c) return C2A_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    int r = do_assign_wires(c);
    if (r) return r;
    if (wire_count) *wire_count = c->wire_count;
    if (node_to_wire && c->n_nodes) {
        C2A_LAUNCH_NOSYNC(k_unbias, grid_for(c->n_nodes, 4096), kThreads, c->stream, (u64)c->n_nodes,
                          (const u32*)c->node_wire1.as<u32>(), c->node_wire.as<u32>());
        r = copy_out(c, node_to_wire, c->node_wire.p, (size_t)c->n_nodes * 4);
        if (r) return r;
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return C2A_OK;
}

int c2a_emit_gates(c2a_ctx* c, uint32_t* in0, uint32_t* in1, uint32_t* out, uint8_t* op) {
    if (!c) return C2A_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    int r = do_emit(c);
    if (r) return r;
    const size_t n4 = (size_t)c->n * 4;
    if ((r = copy_out(c, in0, c->e_in0.p, n4)) || (r = copy_out(c, in1, c->e_in1.p, n4)) ||
        (r = copy_out(c, out, c->e_out.p, n4)) || (r = copy_out(c, op, c->e_op.p, c->n)))
        return r;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return C2A_OK;
}

int c2a_build_circuit(c2a_ctx* c, uint64_t* cycle_at, uint32_t* wire_count) {
    if (!c) return C2A_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    if (c->stage < ST_LOADED || c->circuit_only) return fail(c, C2A_ERR_STATE, "c2a_build_circuit: no gates loaded");
    if (c->io_clash) return fail(c, C2A_ERR_INCONSISTENCY, "Inconsistency: a node is used for both input and output");
    hipEvent_t b0 = c->ev[EV_BUILD0];
    HIP_TRY(hipEventRecord(b0, c->stream));
    int r = do_topo_sort(c, cycle_at, true);
    if (r) return r;
    c->ev_valid[EV_BUILD0] = true;
    // (the numbering's read-back is picked up below — one host round trip for numbering + emission —: until it has been, the
    // stage must not stay at WIRED / EMITTED with the wire count of the graph before)
    struct StageGuard { c2a_ctx* c; bool armed; ~StageGuard() { if (armed && c->stage > ST_SORTED) c->stage = ST_SORTED; } } guard{c, true};
    if ((r = do_assign_wires(c, true))) return r;
    if ((r = do_emit(c))) return r;
    rec(c, EV_BUILD1);
    HIP_TRY(hipStreamSynchronize(c->stream));
    if ((r = finish_wires(c))) return r;
    guard.armed = false;
    if (wire_count) *wire_count = c->wire_count;
    return C2A_OK;
}

int c2a_template_size(uint32_t op, uint32_t width, uint64_t* n_gates, uint64_t* n_aux) {
    if (op >= C2A_NUM_GATE_TYPES || width == 0 || width > 64) return C2A_ERR_ARG;
    TemplateBuilder tb(width);
    tb.build(op);
    if (n_gates) *n_gates = tb.gates.size();
    if (n_aux) *n_aux = tb.aux;
    return C2A_OK;
}

}  // extern "C"

namespace {

// templates, per-gate sizes and the two offset scans; fills c->binfo (no output buffers)
int bool_plan(c2a_ctx* c, uint32_t width) {
    if (c->stage < ST_EMITTED) return fail(c, C2A_ERR_STATE, "c2a_boolify: call c2a_emit_gates / c2a_build_circuit first");
    if (width == 0 || width > 64) return fail(c, C2A_ERR_ARG, "c2a_boolify: width must be in 1..64");
    c->bool_planned = false; c->fmt_chunk_valid = false; c->pruned = false; c->gathered = false;      // (until this plan is complete; the chunk buffers belong to the plan before)
    hipStream_t s = c->stream;
    const u32 n = c->n;
    // templates for this width (host-generated once per width and set of gate types, cached in HBM) — of the types the loaded
    // circuit HAS only: the twenty of them are 0.2 M entries at width 32 (APow alone 178 100), 1.5 ms of a one-shot caller's
    // first c2a_boolify for seven types of a few hundred entries
    u32 need_mask = 0;
    for (u32 op = 0; op < C2A_NUM_GATE_TYPES; ++op) if (c->op_hist[op]) need_mask |= 1u << op;
    if (c->bool_width != width || (need_mask & ~c->bool_op_mask)) {
        if (c->bool_width == width) need_mask |= c->bool_op_mask;
        std::vector<TemplateEntry> all;
        BoolTables T{};
        u32 max_aux = 0;
        for (u32 op = 0; op < 20; ++op) {
            if (!(need_mask >> op & 1u)) continue;
            TemplateBuilder tb(width);
            tb.build(op);
            T.toff[op] = (u32)all.size();
            T.tsize[op] = (u32)tb.gates.size();
            T.taux[op] = tb.aux;
            max_aux = std::max<u32>(max_aux, tb.aux);
            all.insert(all.end(), tb.gates.begin(), tb.gates.end());
        }
        ENSURE(c->tmpl, all.size() * sizeof(TemplateEntry));
        ENSURE(c->tables, sizeof(BoolTables));
        HIP_TRY(hipMemcpyAsync(c->tmpl.p, all.data(), all.size() * sizeof(TemplateEntry), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(c->tables.p, &T, sizeof(T), hipMemcpyHostToDevice, s));
        HIP_TRY(hipStreamSynchronize(s));
        c->bool_width = width;
        c->bool_op_mask = need_mask;
        c->bool_max_aux = max_aux;
        std::memcpy(c->host_tsize, T.tsize, sizeof(T.tsize)); std::memcpy(c->host_taux, T.taux, sizeof(T.taux));
    }
    rec(c, EV_BPREP0);
    ENSURE(c->goff, ((size_t)n + 1) * 8); ENSURE(c->aoff, ((size_t)n + 1) * 8);
    // template sizes and aux-wire counts straight from the op bytes, both scanned in one launch
    int r = scan_1pass<2>(c, s, c->scan_tmp, n, ScanBoolSizes{c->e_op.as<u8>(), c->tables.as<BoolTables>()}, c->goff.as<u64>(), c->aoff.as<u64>());
    if (r) return r;
    // the totals (goff[n], aoff[n]) are sums over the gate types — known since c2a_load_gates: no read-back, the map kernel is
    // queued right behind the scan
    u64 G = 0, AUX = 0;
    for (u32 op = 0; op < C2A_NUM_GATE_TYPES; ++op) { G += c->op_hist[op] * c->host_tsize[op]; AUX += c->op_hist[op] * c->host_taux[op]; }
    const u64 wires = (u64)c->wire_count * width + AUX;
    if (wires >= 0xFFFFFFFFull) return fail(c, C2A_ERR_OVERFLOW, "c2a_boolify: boolean wire ids exceed u32");
    c->binfo.n_gates = G; c->binfo.wire_count = wires; c->binfo.aux_total = AUX; c->binfo.width = width;
    c->binfo.n_in = c->n_in; c->binfo.n_out = c->n_out; c->binfo.m_wires = c->wire_count - c->n_out;
    c->bool_planned = true;
    return C2A_OK;
}

// the map kernel over sorted positions [p_first, p_end) into buffers where boolean gate q sits at q - q_bias
// where the map kernel reads the emitted circuit from: the arrays are indexed by sorted position p - p_base
struct BoolSrc {
    const u32* e_in0; const u32* e_in1; const u32* e_out; const u8* e_op;
    const u64* goff; const u64* aoff; const uint4* tmpl; const BoolTables* tables;
    u32 p_base;
    hipStream_t stream;
};

BoolSrc primary_src(c2a_ctx* c) {
    return BoolSrc{c->e_in0.as<u32>(), c->e_in1.as<u32>(), c->e_out.as<u32>(), c->e_op.as<u8>(), c->goff.as<u64>(), c->aoff.as<u64>(),
                   c->tmpl.as<uint4>(), c->tables.as<BoolTables>(), 0u, c->stream};
}

// the map kernel over sorted positions [p_first, p_end) into buffers where boolean gate q sits at q - q_bias
int bool_map(c2a_ctx* c, const BoolSrc& S, u32 p_first, u32 p_end, u64 q_bias, u32* o_in0, u32* o_in1, u32* o_out, u8* o_op) {
    if (p_end <= p_first) return C2A_OK;
    const u32 width = c->binfo.width, M = c->binfo.m_wires;
    BoolArgs A;
    A.n = c->n; A.width = width; A.M = M; A.aux_base = (u64)M * width; A.out_base = (u64)M * width + c->binfo.aux_total;
    // (a shard holds its own slice of the emitted circuit: the kernel indexes by global sorted position)
    A.e_in0 = S.e_in0 - S.p_base; A.e_in1 = S.e_in1 - S.p_base; A.e_out = S.e_out - S.p_base; A.e_op = S.e_op - S.p_base;
    A.goff = S.goff - S.p_base; A.aoff = S.aoff - S.p_base; A.tmpl = S.tmpl;
    A.b_in0 = o_in0; A.b_in1 = o_in1; A.b_out = o_out; A.b_op = o_op;
    A.p_first = p_first; A.p_end = p_end; A.q_bias = q_bias;
    const u32 ch = c->bool_chunk;
    const u32 chunks = (p_end - p_first + ch - 1) / ch;
    const u32 th = c->bool_threads;
    // fewer chunks than four per CU: several workgroups per chunk (c2a_kernels.h, SLICES), as long as a workgroup keeps >= 8 192 boolean
    // gates — two rounds of its 1 024 lanes x 4 — (the boolean gates of this range are known from the plan: binfo, or the shard's share)
    u32 slices = 1;
    const u32 want = 4u * (u32)c->n_cu;
    if (chunks < want) {
        const u64 per_chunk = c->binfo.n_gates * (u64)(p_end - p_first) / std::max<u32>(1u, c->n) / chunks;      // (an average: shares are by arithmetic gate)
        const u64 by_work = per_chunk / 8192u;
        slices = (u32)std::min<u64>(std::min<u64>((want + chunks - 1) / chunks, std::max<u64>(1u, by_work)), 1024u);
    }
    if (c->bool_slices) slices = c->bool_slices;
    A.slices = slices;
    const dim3 blocks(chunks, slices);
#define C2A_BOOL_CASE(CH, TH) if (ch == CH && th == TH) C2A_LAUNCH((k_boolify<CH, TH>), blocks, TH, S.stream, A, S.tables); else
    C2A_BOOL_CASE(128, 256) C2A_BOOL_CASE(256, 256) C2A_BOOL_CASE(512, 256) C2A_BOOL_CASE(256, 512) C2A_BOOL_CASE(512, 1024)
        C2A_LAUNCH((k_boolify<256, 1024>), blocks, 1024, S.stream, A, S.tables);
#undef C2A_BOOL_CASE
    return C2A_OK;
}

// Multi-device boolify: the emitted circuit is cut by sorted-position range, one range per device of c2a_create; every
// extra device receives its slice (peer copies of e_*, goff, aoff + the templates), runs the same map kernel on its own
// stream and keeps its part of the boolean circuit.  No collective: the map is independent per arithmetic gate, the one
// thing a shard needs from the others — the index of its first boolean gate and aux wire — is in the scans of the plan.
// sorted-position cuts of the emitted circuit into N ranges of (nearly) equal BOOLEAN gate counts — the bytes a shard
// writes — by binary search in the scanned template sizes (a multiplier is 2 824 boolean gates, an XOR 32: equal
// arithmetic counts would not balance a circuit with clustered multipliers)
int shard_cuts(c2a_ctx* c, u32 N, std::vector<u32>& cut, std::vector<u64>& qcut) {
    cut.assign(N + 1, 0); qcut.assign(N + 1, 0);
    ENSURE(c->shard_cut, ((size_t)N + 1) * 4); ENSURE(c->shard_qcut, ((size_t)N + 1) * 8);
    C2A_LAUNCH_NOSYNC(k_shard_cuts, 1, 64, c->stream, c->n, N, (const u64*)c->goff.as<u64>(), c->shard_cut.as<u32>(), c->shard_qcut.as<u64>());
    HIP_TRY(hipMemcpyAsync(cut.data(), c->shard_cut.p, ((size_t)N + 1) * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(qcut.data(), c->shard_qcut.p, ((size_t)N + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return C2A_OK;
}

struct DeviceGuard {         // whatever path leaves the scope, the context's primary device is current again
    int device;
    explicit DeviceGuard(int d) : device(d) {}
    ~DeviceGuard() { (void)hipSetDevice(device); }
};

int bool_map_sharded(c2a_ctx* c) {
    const u32 N = 1 + (u32)c->peers.size();
    DeviceGuard guard(c->device);
    std::vector<u32> cut;
    std::vector<u64> qcut;
    { int r0 = shard_cuts(c, N, cut, qcut); if (r0) return r0; }
    c->shard0_hi = cut[1]; c->shard0_qhi = qcut[1];
    const u64 G0 = qcut[1];
    { int rb = ensure_bool_out(c, G0); if (rb) return rb; }
    rec(c, EV_BPREP1);
    // peers first (their copies and kernels overlap the primary's own shard)
    std::vector<TemplateEntry> tmpl_host;
    for (u32 k = 1; k < N; ++k) {
        PeerDev& P = c->peers[k - 1];
        P.p_lo = cut[k]; P.p_hi = cut[k + 1]; P.q_lo = qcut[k]; P.q_hi = qcut[k + 1]; P.q_bias = qcut[k] & ~15ull;      // (k_boolify stores op bytes 16 at a time: stored indices keep the gate index mod 16)
        const size_t np = P.p_hi - P.p_lo;
        const u64 cntq = P.q_hi - P.q_bias;
        HIP_TRY(hipSetDevice(P.device));
        auto ens = [&](DevBuf& b, size_t bytes) -> int {
            if (bytes == 0) bytes = 16;
            if (b.cap >= bytes) return C2A_OK;
            if (b.p) { (void)hipFree(b.p); b.p = nullptr; b.cap = 0; }
            if (hipMalloc(&b.p, bytes) != hipSuccess) { b.p = nullptr; return fail(c, C2A_ERR_NOMEM, "hipMalloc on device " + std::to_string(P.device)); }
            b.cap = bytes;
            return C2A_OK;
        };
        int r;
        if ((r = ens(P.e_in0, np * 4)) || (r = ens(P.e_in1, np * 4)) || (r = ens(P.e_out, np * 4)) || (r = ens(P.e_op, np)) ||
            (r = ens(P.goff, (np + 1) * 8)) || (r = ens(P.aoff, (np + 1) * 8)) || (r = ens(P.tmpl, c->tmpl.cap)) || (r = ens(P.tables, sizeof(BoolTables))) ||
            (r = ens(P.b_in0, (cntq + 16) * 4)) || (r = ens(P.b_in1, (cntq + 16) * 4)) || (r = ens(P.b_out, (cntq + 16) * 4)) || (r = ens(P.b_op, cntq + 16)) ||
            (r = ens(P.acc, 16))) { (void)hipSetDevice(c->device); return r; }
        HIP_TRY(hipMemcpyPeerAsync(P.e_in0.p, P.device, c->e_in0.as<u32>() + P.p_lo, c->device, np * 4, P.stream));
        HIP_TRY(hipMemcpyPeerAsync(P.e_in1.p, P.device, c->e_in1.as<u32>() + P.p_lo, c->device, np * 4, P.stream));
        HIP_TRY(hipMemcpyPeerAsync(P.e_out.p, P.device, c->e_out.as<u32>() + P.p_lo, c->device, np * 4, P.stream));
        HIP_TRY(hipMemcpyPeerAsync(P.e_op.p, P.device, c->e_op.as<u8>() + P.p_lo, c->device, np, P.stream));
        HIP_TRY(hipMemcpyPeerAsync(P.goff.p, P.device, c->goff.as<u64>() + P.p_lo, c->device, (np + 1) * 8, P.stream));
        HIP_TRY(hipMemcpyPeerAsync(P.aoff.p, P.device, c->aoff.as<u64>() + P.p_lo, c->device, (np + 1) * 8, P.stream));
        if (P.tmpl_width != c->bool_width || P.tmpl_mask != c->bool_op_mask) {
            HIP_TRY(hipMemcpyPeerAsync(P.tmpl.p, P.device, c->tmpl.p, c->device, c->tmpl.cap, P.stream));
            HIP_TRY(hipMemcpyPeerAsync(P.tables.p, P.device, c->tables.p, c->device, sizeof(BoolTables), P.stream));
            P.tmpl_width = c->bool_width; P.tmpl_mask = c->bool_op_mask;
        }
        const BoolSrc S{P.e_in0.as<u32>(), P.e_in1.as<u32>(), P.e_out.as<u32>(), P.e_op.as<u8>(), P.goff.as<u64>(), P.aoff.as<u64>(),
                        P.tmpl.as<uint4>(), P.tables.as<BoolTables>(), P.p_lo, P.stream};
        r = bool_map(c, S, P.p_lo, P.p_hi, P.q_bias, P.b_in0.as<u32>(), P.b_in1.as<u32>(), P.b_out.as<u32>(), P.b_op.as<u8>());
        if (r) { (void)hipSetDevice(c->device); return r; }
    }
    HIP_TRY(hipSetDevice(c->device));
    int r = bool_map(c, primary_src(c), 0, cut[1], 0, c->b_in0.as<u32>(), c->b_in1.as<u32>(), c->b_out.as<u32>(), c->b_op.as<u8>());
    if (r) return r;
    for (PeerDev& P : c->peers) { HIP_TRY(hipSetDevice(P.device)); HIP_TRY(hipStreamSynchronize(P.stream)); }
    HIP_TRY(hipSetDevice(c->device));
    rec(c, EV_BMAP1);
    return C2A_OK;
}

}  // namespace

extern "C" {

int c2a_boolify(c2a_ctx* c, uint32_t width, c2a_bool_info* info) {
    if (!c) return C2A_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    if (c->stage > ST_EMITTED) c->stage = ST_EMITTED;        // the boolean circuit of an earlier call is gone from here on, whatever happens below
    int r = bool_plan(c, width);
    if (r) return r;
    if (!c->peers.empty()) {
        r = bool_map_sharded(c);
        if (r) return r;
    } else {
        const u64 G = c->binfo.n_gates;
        if ((r = ensure_bool_out(c, G))) return r;
        rec(c, EV_BPREP1);
        r = bool_map(c, primary_src(c), 0, c->n, 0, c->b_in0.as<u32>(), c->b_in1.as<u32>(), c->b_out.as<u32>(), c->b_op.as<u8>());
        if (r) return r;
        rec(c, EV_BMAP1);
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (info) *info = c->binfo;
    c->stage = ST_BOOLIFIED;
    return C2A_OK;
}

int c2a_boolify_plan(c2a_ctx* c, uint32_t width, c2a_bool_info* info) {
    if (!c) return C2A_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    int r = bool_plan(c, width);
    if (r) return r;
    if (c->stage > ST_EMITTED) c->stage = ST_EMITTED;        // a full boolean circuit of another width is no longer valid
    if (info) *info = c->binfo;
    return C2A_OK;
}

int c2a_boolify_chunk(c2a_ctx* c, uint64_t first_gate, uint64_t n_gates, uint32_t* in0, uint32_t* in1, uint32_t* out,
                      uint8_t* op, uint64_t* first_bool_gate, uint64_t* n_bool_gates) {
    if (!c) return C2A_ERR_ARG;
    if (!c->bool_planned || c->stage < ST_EMITTED) return fail(c, C2A_ERR_STATE, "c2a_boolify_chunk: call c2a_boolify_plan first");
    if (first_gate > c->n || n_gates > c->n - first_gate) return fail(c, C2A_ERR_ARG, "c2a_boolify_chunk: range out of bounds");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    u64 q[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(&q[0], c->goff.as<u64>() + first_gate, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(&q[1], c->goff.as<u64>() + first_gate + n_gates, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    const u64 cntq = q[1] - q[0], bias = q[0] & ~15ull, lead = q[0] - bias;
    ENSURE(c->cb_in0, (cntq + 32) * 4); ENSURE(c->cb_in1, (cntq + 32) * 4); ENSURE(c->cb_out, (cntq + 32) * 4); ENSURE(c->cb_op, cntq + 32);
    int r = bool_map(c, primary_src(c), (u32)first_gate, (u32)(first_gate + n_gates), bias, c->cb_in0.as<u32>(), c->cb_in1.as<u32>(),
                     c->cb_out.as<u32>(), c->cb_op.as<u8>());
    if (r) return r;
    if ((r = copy_out(c, in0, c->cb_in0.as<u32>() + lead, cntq * 4)) || (r = copy_out(c, in1, c->cb_in1.as<u32>() + lead, cntq * 4)) ||
        (r = copy_out(c, out, c->cb_out.as<u32>() + lead, cntq * 4)) || (r = copy_out(c, op, c->cb_op.as<u8>() + lead, cntq)))
        return r;
    HIP_TRY(hipStreamSynchronize(s));
    c->fmt_chunk_first = q[0]; c->fmt_chunk_cnt = cntq; c->fmt_chunk_valid = true;
    if (first_bool_gate) *first_bool_gate = q[0];
    if (n_bool_gates) *n_bool_gates = cntq;
    return C2A_OK;
}

int c2a_boolify_shard_range(c2a_ctx* c, uint32_t k, uint32_t n_shards, uint64_t* first_gate, uint64_t* n_gates) {
    if (!c || !first_gate || !n_gates) return C2A_ERR_ARG;
    if (!c->bool_planned || c->stage < ST_EMITTED) return fail(c, C2A_ERR_STATE, "c2a_boolify_shard_range: call c2a_boolify_plan first");
    if (n_shards == 0 || n_shards > 4096 || k >= n_shards) return fail(c, C2A_ERR_ARG, "c2a_boolify_shard_range: shard index out of range");
    HIP_TRY(hipSetDevice(c->device));
    std::vector<u32> cut;
    std::vector<u64> qcut;
    int r = shard_cuts(c, n_shards, cut, qcut);
    if (r) return r;
    *first_gate = cut[k];
    *n_gates = cut[k + 1] - cut[k];
    return C2A_OK;
}

int c2a_bool_read(c2a_ctx* c, uint64_t first, uint64_t count, uint32_t* in0, uint32_t* in1, uint32_t* out, uint8_t* op) {
    if (!c) return C2A_ERR_ARG;
    if (c->stage < ST_BOOLIFIED) return fail(c, C2A_ERR_STATE, "c2a_bool_read: call c2a_boolify first");
    if (first > c->binfo.n_gates || count > c->binfo.n_gates - first) return fail(c, C2A_ERR_ARG, "c2a_bool_read: range out of bounds");
    // the boolean circuit may be spread over the devices of the context: copy each owner's part of [first, first + count)
    const u64 last = first + count;
    const u64 own_hi = c->peers.empty() ? c->binfo.n_gates : c->shard0_qhi;
    int r;
    HIP_TRY(hipSetDevice(c->device));
    if (first < own_hi && count) {
        const u64 hi = std::min(last, own_hi), cntq = hi - first;
        if ((r = copy_out(c, in0, c->b_in0.as<u32>() + first, cntq * 4)) || (r = copy_out(c, in1, c->b_in1.as<u32>() + first, cntq * 4)) ||
            (r = copy_out(c, out, c->b_out.as<u32>() + first, cntq * 4)) || (r = copy_out(c, op, c->b_op.as<u8>() + first, cntq)))
            return r;
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    for (PeerDev& P : c->peers) {
        const u64 lo = std::max(first, P.q_lo), hi = std::min(last, P.q_hi);
        if (lo >= hi) continue;
        const u64 cntq = hi - lo, src = lo - P.q_bias, dst = lo - first;
        HIP_TRY(hipSetDevice(P.device));
        if (in0) HIP_TRY(hipMemcpyAsync(in0 + dst, P.b_in0.as<u32>() + src, cntq * 4, hipMemcpyDeviceToHost, P.stream));
        if (in1) HIP_TRY(hipMemcpyAsync(in1 + dst, P.b_in1.as<u32>() + src, cntq * 4, hipMemcpyDeviceToHost, P.stream));
        if (out) HIP_TRY(hipMemcpyAsync(out + dst, P.b_out.as<u32>() + src, cntq * 4, hipMemcpyDeviceToHost, P.stream));
        if (op) HIP_TRY(hipMemcpyAsync(op + dst, P.b_op.as<u8>() + src, cntq, hipMemcpyDeviceToHost, P.stream));
        HIP_TRY(hipStreamSynchronize(P.stream));
    }
    HIP_TRY(hipSetDevice(c->device));
    return C2A_OK;
}

int c2a_checksum(c2a_ctx* c, int which, uint64_t* value) {
    if (!c || !value) return C2A_ERR_ARG;
    HIP_TRY(hipSetDevice(c->device));
    const void* p = nullptr;
    u64 cnt = 0;
    bool bytes = false;
    Stage need = ST_SORTED;
    const u64 own_bool = c->peers.empty() ? c->binfo.n_gates : c->shard0_qhi;     // boolean gates held by the primary device
    switch (which) {
    case 0: p = c->sorted.p; cnt = c->n; need = ST_SORTED; break;
    case 1: p = c->e_in0.p; cnt = c->n; need = ST_EMITTED; break;
    case 2: p = c->e_in1.p; cnt = c->n; need = ST_EMITTED; break;
    case 3: p = c->e_out.p; cnt = c->n; need = ST_EMITTED; break;
    case 4: p = c->e_op.p; cnt = c->n; need = ST_EMITTED; bytes = true; break;
    case 5: p = c->b_in0.p; cnt = own_bool; need = ST_BOOLIFIED; break;
    case 6: p = c->b_in1.p; cnt = own_bool; need = ST_BOOLIFIED; break;
    case 7: p = c->b_out.p; cnt = own_bool; need = ST_BOOLIFIED; break;
    case 8: p = c->b_op.p; cnt = own_bool; need = ST_BOOLIFIED; bytes = true; break;
    case 9: p = c->node_wire1.p; cnt = c->n_nodes; need = ST_WIRED; break;
    default: return fail(c, C2A_ERR_ARG, "c2a_checksum: unknown stream id");
    }
    if (c->stage < need) return fail(c, C2A_ERR_STATE, "c2a_checksum: result not computed yet");
    ull* acc = reinterpret_cast<ull*>(c->scalars.as<u32>() + SC_TOTAL64);
    HIP_TRY(hipMemsetAsync(acc, 0, 8, c->stream));
    if (cnt) {
        if (bytes) C2A_LAUNCH_NOSYNC(k_checksum_u8, grid_for(cnt, 2048), kThreads, c->stream, cnt, (u64)0, (const u8*)p, acc);
        else C2A_LAUNCH_NOSYNC(k_checksum_u32, grid_for(cnt, 2048), kThreads, c->stream, cnt, (u64)0, (const u32*)p, acc);
    }
    u64 v = 0;
    HIP_TRY(hipMemcpyAsync(&v, acc, 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    // a sharded boolean circuit: the checksum is a sum of position-salted terms, so the shards' parts simply add up
    if (which >= 5 && which <= 8) {
        for (PeerDev& P : c->peers) {
            const u64 cntq = P.q_hi - P.q_lo, skip = P.q_lo - P.q_bias;
            if (!cntq) continue;
            HIP_TRY(hipSetDevice(P.device));
            ull* pacc = P.acc.as<ull>();
            HIP_TRY(hipMemsetAsync(pacc, 0, 8, P.stream));
            const DevBuf& B = which == 5 ? P.b_in0 : which == 6 ? P.b_in1 : which == 7 ? P.b_out : P.b_op;
            if (bytes) C2A_LAUNCH_NOSYNC(k_checksum_u8, grid_for(cntq, 2048), kThreads, P.stream, cntq, P.q_lo, (const u8*)(B.as<u8>() + skip), pacc);
            else C2A_LAUNCH_NOSYNC(k_checksum_u32, grid_for(cntq, 2048), kThreads, P.stream, cntq, P.q_lo, (const u32*)(B.as<u32>() + skip), pacc);
            u64 pv = 0;
            HIP_TRY(hipMemcpyAsync(&pv, pacc, 8, hipMemcpyDeviceToHost, P.stream));
            HIP_TRY(hipStreamSynchronize(P.stream));
            v += pv;
        }
        HIP_TRY(hipSetDevice(c->device));
    }
    *value = v;
    return C2A_OK;
}

// The boolean circuit as the level-parallel passes see it (c2a_kernels.h BoolSegs): ONE piece on a single-device context; on a
// multi-device one every device keeps the gates of its own range of sorted positions and the primary READS them where they lie —
// peer access over xGMI, enabled at c2a_create — instead of gathering all of it (9.6 GB at the headline size) first.  Only where a
// peer cannot be mapped (no peer access between the two devices) the pieces are gathered on the primary as before (peer copies, made
// once per c2a_boolify).
static int bool_segs(c2a_ctx* c, const BoolSegs** out) {
    BoolSegs S{};
    const u64 G = c->binfo.n_gates;
    if (c->peers.empty()) {
        S.n_seg = 1; S.p_hi[0] = c->n;
        S.in0[0] = c->b_in0.as<u32>(); S.in1[0] = c->b_in1.as<u32>(); S.out[0] = c->b_out.as<u32>(); S.op[0] = c->b_op.as<u8>();
    } else if (c->peer_access) {
        S.n_seg = 1 + (u32)c->peers.size();
        S.p_hi[0] = c->shard0_hi;
        S.in0[0] = c->b_in0.as<u32>(); S.in1[0] = c->b_in1.as<u32>(); S.out[0] = c->b_out.as<u32>(); S.op[0] = c->b_op.as<u8>();
        for (size_t k = 0; k < c->peers.size(); ++k) {
            const PeerDev& P = c->peers[k];
            S.p_hi[k + 1] = P.p_hi;
            // (boolean gate q of the whole circuit is stored at index q - q_bias of the peer's arrays)
            S.in0[k + 1] = P.b_in0.as<u32>() - P.q_bias; S.in1[k + 1] = P.b_in1.as<u32>() - P.q_bias;
            S.out[k + 1] = P.b_out.as<u32>() - P.q_bias; S.op[k + 1] = P.b_op.as<u8>() - P.q_bias;
        }
    } else {
        if (!c->gathered) {
            ENSURE(c->g_in0, G * 4 + 16); ENSURE(c->g_in1, G * 4 + 16); ENSURE(c->g_out, G * 4 + 16); ENSURE(c->g_op, G + 16);
            const u64 g0 = c->shard0_qhi;
            HIP_TRY(hipMemcpyAsync(c->g_in0.p, c->b_in0.p, g0 * 4, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->g_in1.p, c->b_in1.p, g0 * 4, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->g_out.p, c->b_out.p, g0 * 4, hipMemcpyDeviceToDevice, c->stream));
            HIP_TRY(hipMemcpyAsync(c->g_op.p, c->b_op.p, g0, hipMemcpyDeviceToDevice, c->stream));
            for (PeerDev& P : c->peers) {
                const u64 cntq = P.q_hi - P.q_lo, skip = P.q_lo - P.q_bias;
                if (!cntq) continue;
                HIP_TRY(hipMemcpyPeerAsync(c->g_in0.as<u32>() + P.q_lo, c->device, P.b_in0.as<u32>() + skip, P.device, cntq * 4, c->stream));
                HIP_TRY(hipMemcpyPeerAsync(c->g_in1.as<u32>() + P.q_lo, c->device, P.b_in1.as<u32>() + skip, P.device, cntq * 4, c->stream));
                HIP_TRY(hipMemcpyPeerAsync(c->g_out.as<u32>() + P.q_lo, c->device, P.b_out.as<u32>() + skip, P.device, cntq * 4, c->stream));
                HIP_TRY(hipMemcpyPeerAsync(c->g_op.as<u8>() + P.q_lo, c->device, P.b_op.as<u8>() + skip, P.device, cntq, c->stream));
            }
            HIP_TRY(hipStreamSynchronize(c->stream));
            c->gathered = true;
        }
        S.n_seg = 1; S.p_hi[0] = c->n;
        S.in0[0] = c->g_in0.as<u32>(); S.in1[0] = c->g_in1.as<u32>(); S.out[0] = c->g_out.as<u32>(); S.op[0] = c->g_op.as<u8>();
    }
    // (the peers' kernels and copies must have landed before the primary reads their pieces)
    DeviceGuard guard(c->device);
    for (PeerDev& P : c->peers) { HIP_TRY(hipSetDevice(P.device)); HIP_TRY(hipStreamSynchronize(P.stream)); }
    HIP_TRY(hipSetDevice(c->device));
    ENSURE(c->segs, sizeof(BoolSegs));
    HIP_TRY(hipMemcpyAsync(c->segs.p, &S, sizeof(S), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));        // (S is a stack object)
    *out = c->segs.as<BoolSegs>();
    return C2A_OK;
}

// level lists of the loaded circuit (gates of one reverse Kahn level are independent; producers sit in higher levels than
// their consumers), the inverse of the sorted order, and the evaluation launch itself
static int eval_levels(c2a_ctx* c) {
    hipStream_t s = c->stream;
    const u32 n = c->n, L = c->stats.levels;
    ENSURE(c->ev_spos, (size_t)n * 4);
    ENSURE(c->ev_lcount, ((size_t)L + 1) * 4); ENSURE(c->ev_lbase, ((size_t)L + 2) * 4); ENSURE(c->ev_lorder, (size_t)n * 4);
    if (!n) return C2A_OK;
    C2A_LAUNCH_NOSYNC(k_eval_inverse, grid_for(n, 4096), kThreads, s, n, c->sorted_r.as<u32>(), c->ev_spos.as<u32>());
    HIP_TRY(hipMemsetAsync(c->ev_lcount.p, 0, ((size_t)L + 1) * 4, s));
    u32 lbits = 1;
    while (lbits < 32 && (1ull << lbits) <= (u64)L) ++lbits;      // (bits that tell two levels apart: the wave's lanes are matched by them)
    C2A_LAUNCH(k_level_hist, grid_for(n, 4096), kThreads, s, n, (const uint4*)c->meta.as<uint4>(), c->ev_lcount.as<u32>(), lbits);
    int r = scan_exclusive<u32>(c, c->ev_lcount.as<u32>(), c->ev_lbase.as<u32>(), L);
    if (r) return r;
    HIP_TRY(hipMemsetAsync(c->ev_lcount.p, 0, ((size_t)L + 1) * 4, s));
    C2A_LAUNCH(k_level_scatter, grid_for(n, 4096), kThreads, s, n, (const uint4*)c->meta.as<uint4>(),
               (const u32*)c->ev_lbase.as<u32>(), c->ev_lcount.as<u32>(), c->ev_lorder.as<u32>(), lbits);
    return C2A_OK;
}

// a level-parallel launch whose grid barrier gave up (kBarrierPolls: some workgroup was never resident) reports it here
static int barrier_gave_up(c2a_ctx* c, const char* who) {
    u32 ab = 0;
    HIP_TRY(hipMemcpy(&ab, c->ev_bar.as<u32>() + kBarAbort, 4, hipMemcpyDeviceToHost));
    if (ab) return fail(c, C2A_ERR_HIP, std::string(who) + ": the grid barrier of the level-parallel launch gave up (not every workgroup of the grid was resident)");
    return C2A_OK;
}

static int eval_run(c2a_ctx* c, u32 mode, u32 width) {
    hipStream_t s = c->stream;
    if (!c->n) return C2A_OK;
    EvalRun R;
    R.levels = c->stats.levels; R.width = width; R.mode = mode;
    R.lbase = c->ev_lbase.as<u32>(); R.order = c->ev_lorder.as<u32>(); R.spos = c->ev_spos.as<u32>();
    R.e_in0 = c->e_in0.as<u32>(); R.e_in1 = c->e_in1.as<u32>(); R.e_out = c->e_out.as<u32>(); R.e_op = c->e_op.as<u8>();
    R.segs = nullptr;
    if (mode & 2u) { int rv = bool_segs(c, &R.segs); if (rv) return rv; }
    R.goff = c->goff.as<u64>(); R.b_in0 = nullptr; R.b_in1 = nullptr; R.b_out = nullptr; R.b_op = nullptr;
    R.aval = c->ev_aval.as<u64>(); R.bval = c->ev_bval.as<u64>();
    ENSURE(c->ev_bar, 256);
    HIP_TRY(hipMemsetAsync(c->ev_bar.p, 0, 256, s));
    R.bar = c->ev_bar.as<u32>();
    // every workgroup waits for every other at the end of a level: the grid must be resident as a whole
    u32 grid = 8;
#ifndef C2A_EMULATE
    {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_eval_run, kThreads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        grid = (u32)c->n_cu * (u32)std::min(per_cu, 4);
    }
#endif
    C2A_LAUNCH_CONCURRENT(k_eval_run, grid, kThreads, s, R);
    return C2A_OK;
}

// the verifier of a multi-device context: every device checks the gates it holds (k_verify_local), nothing is gathered
static int verify_local(c2a_ctx* c, u64 seed, u64* n_checked, u64* n_mismatch) {
    DeviceGuard guard(c->device);
    const u32 width = c->binfo.width, M = c->binfo.m_wires;
    const u32 words = 3u * width + c->bool_max_aux + 1u;
    auto launch = [&](hipStream_t s, DevBuf& scratch, ull* bad, VerifyLocal V) -> int {
        if (V.p_hi <= V.p_lo) return C2A_OK;
        u32 blocks = std::min<u32>((u32)c->n_cu * 8u, (V.p_hi - V.p_lo + 3u) / 4u);
        while (blocks > 1 && (size_t)blocks * 4 * words * 8 > (256u << 20)) blocks /= 2;      // (a wave per gate, 4 waves per workgroup)
        ENSURE(scratch, (size_t)blocks * 4 * words * 8);
        HIP_TRY(hipMemsetAsync(bad, 0, 8, s));
        V.scratch = scratch.as<u64>(); V.scratch_words = words; V.bad = bad;
        V.width = width; V.M = M; V.aux_base = (u64)M * width; V.out_base = (u64)M * width + c->binfo.aux_total; V.seed = seed;
        C2A_LAUNCH(k_verify_local, blocks, kThreads, s, V);
        return C2A_OK;
    };
    HIP_TRY(hipSetDevice(c->device));
    ull* acc0 = reinterpret_cast<ull*>(c->scalars.as<u32>() + SC_TOTAL64);
    {
        VerifyLocal V{};
        V.p_lo = 0; V.p_hi = c->shard0_hi; V.p_base = 0; V.q_bias = 0;
        V.e_in0 = c->e_in0.as<u32>(); V.e_in1 = c->e_in1.as<u32>(); V.e_out = c->e_out.as<u32>(); V.e_op = c->e_op.as<u8>();
        V.goff = c->goff.as<u64>(); V.aoff = c->aoff.as<u64>();
        V.b_in0 = c->b_in0.as<u32>(); V.b_in1 = c->b_in1.as<u32>(); V.b_out = c->b_out.as<u32>(); V.b_op = c->b_op.as<u8>();
        int r = launch(c->stream, c->ev_bval, acc0, V);
        if (r) return r;
    }
    for (PeerDev& P : c->peers) {
        HIP_TRY(hipSetDevice(P.device));
        VerifyLocal V{};
        V.p_lo = P.p_lo; V.p_hi = P.p_hi; V.p_base = P.p_lo; V.q_bias = P.q_bias;
        V.e_in0 = P.e_in0.as<u32>(); V.e_in1 = P.e_in1.as<u32>(); V.e_out = P.e_out.as<u32>(); V.e_op = P.e_op.as<u8>();
        V.goff = P.goff.as<u64>(); V.aoff = P.aoff.as<u64>();
        V.b_in0 = P.b_in0.as<u32>(); V.b_in1 = P.b_in1.as<u32>(); V.b_out = P.b_out.as<u32>(); V.b_op = P.b_op.as<u8>();
        int r = launch(P.stream, P.vscratch, P.acc.as<ull>(), V);
        if (r) return r;
    }
    u64 bad = 0, part = 0;
    HIP_TRY(hipSetDevice(c->device));
    if (c->shard0_hi) { HIP_TRY(hipMemcpyAsync(&part, acc0, 8, hipMemcpyDeviceToHost, c->stream)); HIP_TRY(hipStreamSynchronize(c->stream)); bad += part; }
    for (PeerDev& P : c->peers) {
        if (P.p_hi <= P.p_lo) continue;
        HIP_TRY(hipSetDevice(P.device));
        HIP_TRY(hipMemcpyAsync(&part, P.acc.p, 8, hipMemcpyDeviceToHost, P.stream));
        HIP_TRY(hipStreamSynchronize(P.stream));
        bad += part;
    }
    if (n_checked) *n_checked = (u64)c->n * 64;
    if (n_mismatch) *n_mismatch = bad;
    c->stats.verifier = 2;                           // (every device checked its own gates locally: c2a.h c2a_stats)
    return C2A_OK;
}

int c2a_verify_boolify(c2a_ctx* c, uint64_t seed, uint64_t* n_checked, uint64_t* n_mismatch) {
    if (!c) return C2A_ERR_ARG;
    if (c->stage < ST_BOOLIFIED) return fail(c, C2A_ERR_STATE, "c2a_verify_boolify: call c2a_boolify first");
    if (!c->peers.empty()) return verify_local(c, seed, n_checked, n_mismatch);
    if (!c->peel_meta_valid) return fail(c, C2A_ERR_STATE, "c2a_verify_boolify: needs the level data of c2a_topo_sort (not of c2a_topo_sort_serial)");
    if (c->has_dup) return fail(c, C2A_ERR_STATE, "c2a_verify_boolify: two gates write one node (compiler.rs:403-406 keeps the last): the level-parallel passes do not order the writers of one wire — evaluate the emitted gate list sequentially instead");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const u32 n = c->n, wc = c->wire_count, width = c->binfo.width, M = c->binfo.m_wires;
    const u64 out_base = (u64)M * width + c->binfo.aux_total;
    ENSURE(c->ev_produced, wc);
    ENSURE(c->ev_aval, (size_t)wc * 64 * 8);
    ENSURE(c->ev_bval, (size_t)c->binfo.wire_count * 8);
    HIP_TRY(hipMemsetAsync(c->ev_produced.p, 0, wc, s));
    ull* acc = reinterpret_cast<ull*>(c->scalars.as<u32>() + SC_TOTAL64);
    HIP_TRY(hipMemsetAsync(acc, 0, 8, s));
    if (n) C2A_LAUNCH_NOSYNC(k_eval_mark_produced, grid_for(n, 4096), kThreads, s, n, c->e_out.as<u32>(), c->ev_produced.as<u8>());
    if (wc)
        C2A_LAUNCH_NOSYNC(k_eval_init, grid_for((u64)wc * 64, 8192), kThreads, s, wc, width, M, out_base, (u64)seed,
                          (const u8*)c->ev_produced.as<u8>(), c->ev_aval.as<u64>(), c->ev_bval.as<u64>());
    int r = eval_levels(c);
    if (r) return r;
    if ((r = eval_run(c, 3u, width))) return r;
    if (wc)
        C2A_LAUNCH_NOSYNC(k_eval_compare, grid_for((u64)wc * 64, 8192), kThreads, s, wc, width, M, out_base,
                          (const u64*)c->ev_aval.as<u64>(), (const u64*)c->ev_bval.as<u64>(), acc);
    u64 bad = 0;
    HIP_TRY(hipMemcpyAsync(&bad, acc, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (n && (r = barrier_gave_up(c, "c2a_verify_boolify"))) return r;
    if (n_checked) *n_checked = (u64)wc * 64;
    if (n_mismatch) *n_mismatch = bad;
    c->stats.verifier = 1;                           // (both circuits simulated wire by wire)
    return C2A_OK;
}

int c2a_boolify_prune(c2a_ctx* c, c2a_prune_info* info) {
    if (!c) return C2A_ERR_ARG;
    c->pruned = false;
    if (c->stage < ST_BOOLIFIED) return fail(c, C2A_ERR_STATE, "c2a_boolify_prune: call c2a_boolify first");
    if (!c->peel_meta_valid) return fail(c, C2A_ERR_STATE, "c2a_boolify_prune: needs the level data of c2a_topo_sort (not of c2a_topo_sort_serial)");
    if (c->has_dup) return fail(c, C2A_ERR_STATE, "c2a_boolify_prune: two gates write one node (compiler.rs:403-406 keeps the last): the level-parallel passes do not order the writers of one wire — evaluate the emitted gate list sequentially instead");
    if (c->binfo.wire_count + 2 >= 0xFFFFFFFFull) return fail(c, C2A_ERR_OVERFLOW, "c2a_boolify_prune: no room for the two constant wires in u32");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const u32 n = c->n, width = c->binfo.width;
    const u64 G = c->binfo.n_gates, wires = c->binfo.wire_count;
    ENSURE(c->pr_rep, (wires + 2) * 4); ENSURE(c->pr_need, (wires + 2) * 4);
    ENSURE(c->pr_tin0, G * 4 + 16); ENSURE(c->pr_tin1, G * 4 + 16); ENSURE(c->pr_top, G + 16);
    ENSURE(c->pr_live, (size_t)n * 4 + 16); ENSURE(c->pr_goff, ((size_t)n + 1) * 4); ENSURE(c->pr_counts, 64); ENSURE(c->ev_bar, 256);
    HIP_TRY(hipMemsetAsync(c->pr_counts.p, 0, 64, s));
    C2A_LAUNCH_NOSYNC(k_prune_init, grid_for(wires + 2, 8192), kThreads, s, wires + 2, c->pr_rep.as<u32>(), c->pr_need.as<u32>());
    int r = eval_levels(c);
    if (r) return r;
    PruneRun R;
    R.levels = c->stats.levels; R.width = width; R.M = c->binfo.m_wires; R.n_out_wires = c->n_out * width;
    R.out_base = (u64)c->binfo.m_wires * width + c->binfo.aux_total;
    R.zero_wire = (u32)wires; R.one_wire = (u32)wires + 1;
    R.lbase = c->ev_lbase.as<u32>(); R.order = c->ev_lorder.as<u32>(); R.spos = c->ev_spos.as<u32>();
    if ((r = bool_segs(c, &R.segs))) return r;
    R.goff = c->goff.as<u64>();
    R.rep = c->pr_rep.as<u32>(); R.need = c->pr_need.as<u32>();
    R.t_in0 = c->pr_tin0.as<u32>(); R.t_in1 = c->pr_tin1.as<u32>(); R.t_op = c->pr_top.as<u8>();
    R.live_cnt = c->pr_live.as<u32>(); R.bar = c->ev_bar.as<u32>(); R.counts = c->pr_counts.as<ull>();
    u32 grid = 8;
#ifndef C2A_EMULATE
    {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_prune_fold, kThreads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        grid = (u32)c->n_cu * (u32)std::min(per_cu, 4);
    }
#endif
    if (n) {
        HIP_TRY(hipMemsetAsync(c->ev_bar.p, 0, 256, s));
        C2A_LAUNCH_CONCURRENT(k_prune_fold, grid, kThreads, s, R);
        HIP_TRY(hipMemsetAsync(c->ev_bar.p, 0, 4, s));      // (the arrival count only: a give-up of the first pass must stay visible)
        C2A_LAUNCH_CONCURRENT(k_prune_live, grid, kThreads, s, R);
    }
    r = scan_exclusive<u32>(c, c->pr_live.as<u32>(), c->pr_goff.as<u32>(), n);
    if (r) return r;
    u32 kept = 0;
    ull cnts[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(&kept, c->pr_goff.as<u32>() + n, 4, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(cnts, c->pr_counts.p, 16, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    if (n && (r = barrier_gave_up(c, "c2a_boolify_prune"))) return r;
    const u64 PG = (u64)kept + 2;
    ENSURE(c->p_in0, PG * 4); ENSURE(c->p_in1, PG * 4); ENSURE(c->p_out, PG * 4); ENSURE(c->p_op, PG);
    C2A_LAUNCH_NOSYNC(k_prune_consts, 1, 64, s, R.zero_wire, R.one_wire, c->p_in0.as<u32>(), c->p_in1.as<u32>(), c->p_out.as<u32>(), c->p_op.as<u8>());
    if (n)
        C2A_LAUNCH_NOSYNC(k_prune_compact, grid_for(n, 4096), kThreads, s, n, (const u64*)c->goff.as<u64>(), (const u32*)c->pr_goff.as<u32>(),
                          (const u32*)c->pr_tin0.as<u32>(), (const u32*)c->pr_tin1.as<u32>(), R.segs, (const u8*)c->pr_top.as<u8>(),
                          c->p_in0.as<u32>(), c->p_in1.as<u32>(), c->p_out.as<u32>(), c->p_op.as<u8>());
    HIP_TRY(hipStreamSynchronize(s));
    c->pinfo.n_gates = PG; c->pinfo.n_gates_before = G; c->pinfo.n_folded = cnts[0]; c->pinfo.n_dead = cnts[1];
    c->pinfo.wire_count = wires + 2; c->pinfo.zero_wire = R.zero_wire; c->pinfo.one_wire = R.one_wire;
    c->pruned = true;
    if (info) *info = c->pinfo;
    return C2A_OK;
}

int c2a_pruned_read(c2a_ctx* c, uint64_t first, uint64_t count, uint32_t* in0, uint32_t* in1, uint32_t* out, uint8_t* op) {
    if (!c) return C2A_ERR_ARG;
    if (!c->pruned || c->stage < ST_BOOLIFIED) return fail(c, C2A_ERR_STATE, "c2a_pruned_read: call c2a_boolify_prune first");
    if (first > c->pinfo.n_gates || count > c->pinfo.n_gates - first) return fail(c, C2A_ERR_ARG, "c2a_pruned_read: range out of bounds");
    HIP_TRY(hipSetDevice(c->device));
    int r;
    if ((r = copy_out(c, in0, c->p_in0.as<u32>() + first, count * 4)) || (r = copy_out(c, in1, c->p_in1.as<u32>() + first, count * 4)) ||
        (r = copy_out(c, out, c->p_out.as<u32>() + first, count * 4)) || (r = copy_out(c, op, c->p_op.as<u8>() + first, count)))
        return r;
    HIP_TRY(hipStreamSynchronize(c->stream));
    return C2A_OK;
}

int c2a_eval(c2a_ctx* c, int which, uint32_t width, uint32_t n_vectors, const uint64_t* inputs, uint32_t n_const,
             const uint32_t* const_wires, const uint64_t* const_values, uint64_t* outputs) {
    if (!c) return C2A_ERR_ARG;
    if (which != 0 && which != 1 && which != 2) return fail(c, C2A_ERR_ARG, "c2a_eval: which must be 0 (arithmetic circuit), 1 (boolean circuit) or 2 (pruned boolean circuit)");
    if (which == 2 && !c->pruned) return fail(c, C2A_ERR_STATE, "c2a_eval: call c2a_boolify_prune first");
    if (n_vectors == 0 || n_vectors > 64) return fail(c, C2A_ERR_ARG, "c2a_eval: 1..64 vectors per call");
    if (c->stage < (which ? ST_BOOLIFIED : ST_EMITTED)) return fail(c, C2A_ERR_STATE, which ? "c2a_eval: call c2a_boolify first" : "c2a_eval: call c2a_emit_gates / c2a_build_circuit first");
    if (!c->peel_meta_valid) return fail(c, C2A_ERR_STATE, "c2a_eval: needs the level data of c2a_topo_sort (not of c2a_topo_sort_serial)");
    if (c->has_dup) return fail(c, C2A_ERR_STATE, "c2a_eval: two gates write one node (compiler.rs:403-406 keeps the last): the level-parallel passes do not order the writers of one wire — evaluate the emitted gate list sequentially instead");
    if (which) width = c->binfo.width;
    if (width == 0 || width > 64) return fail(c, C2A_ERR_ARG, "c2a_eval: width must be in 1..64");
    if ((c->n_in && !inputs) || (c->n_out && !outputs) || (n_const && (!const_wires || !const_values))) return fail(c, C2A_ERR_ARG, "c2a_eval: null value arrays");
    const u32 wc = c->wire_count;
    for (u32 i = 0; i < n_const; ++i)
        if (const_wires[i] >= wc) return fail(c, C2A_ERR_ARG, "c2a_eval: constant wire id out of range");
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const u32 M = wc - c->n_out;
    const u64 out_base = which ? (u64)c->binfo.m_wires * width + c->binfo.aux_total : 0;
    const size_t in_words = (size_t)c->n_in * n_vectors, out_words = (size_t)c->n_out * n_vectors, cst_words = (size_t)n_const * n_vectors;
    ENSURE(c->ev_io, (in_words + out_words + cst_words) * 8 + (size_t)n_const * 4 + 64);
    u64* d_in = c->ev_io.as<u64>(); u64* d_out = d_in + in_words; u64* d_cv = d_out + out_words; u32* d_cw = reinterpret_cast<u32*>(d_cv + cst_words);
    if (in_words) HIP_TRY(hipMemcpyAsync(d_in, inputs, in_words * 8, hipMemcpyHostToDevice, s));
    std::vector<u64> cv(cst_words);                      // a constant holds its value in every vector
    for (u32 i = 0; i < n_const; ++i) for (u32 t = 0; t < n_vectors; ++t) cv[(size_t)i * n_vectors + t] = const_values[i];
    if (n_const) {
        HIP_TRY(hipMemcpyAsync(d_cv, cv.data(), cst_words * 8, hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(d_cw, const_wires, (size_t)n_const * 4, hipMemcpyHostToDevice, s));
    }
    // every wire starts at 0 (a wire nothing drives stays 0), then the inputs and the constants
    if (which) {
        ENSURE(c->ev_bval, ((size_t)c->binfo.wire_count + 2) * 8);
        HIP_TRY(hipMemsetAsync(c->ev_bval.p, 0, (size_t)c->binfo.wire_count * 8 + 8, s));       // (incl. the pruned circuit's ZERO wire)
        HIP_TRY(hipMemsetAsync(c->ev_bval.as<u64>() + c->binfo.wire_count + 1, 0xFF, 8, s));    // (... and its ONE wire)
        if (c->n_in) C2A_LAUNCH_NOSYNC(k_eval_set_bool, grid_for((u64)c->n_in * width, 4096), kThreads, s, c->n_in, n_vectors, width, c->binfo.m_wires, out_base,
                                       (const u32*)nullptr, (const u64*)d_in, c->ev_bval.as<u64>());
        if (n_const) C2A_LAUNCH_NOSYNC(k_eval_set_bool, grid_for((u64)n_const * width, 4096), kThreads, s, n_const, n_vectors, width, c->binfo.m_wires, out_base,
                                       (const u32*)d_cw, (const u64*)d_cv, c->ev_bval.as<u64>());
    } else {
        ENSURE(c->ev_aval, (size_t)wc * 64 * 8);
        HIP_TRY(hipMemsetAsync(c->ev_aval.p, 0, (size_t)wc * 64 * 8, s));
        if (c->n_in) C2A_LAUNCH_NOSYNC(k_eval_set_arith, grid_for(in_words, 4096), kThreads, s, c->n_in, n_vectors, width, (const u32*)nullptr, (const u64*)d_in, c->ev_aval.as<u64>());
        if (n_const) C2A_LAUNCH_NOSYNC(k_eval_set_arith, grid_for(cst_words, 4096), kThreads, s, n_const, n_vectors, width, (const u32*)d_cw, (const u64*)d_cv, c->ev_aval.as<u64>());
    }
    int r = eval_levels(c);
    if (r) return r;
    if (which == 2) {
        if (c->n) {
            EvalRun R;
            R.levels = c->stats.levels; R.width = width; R.mode = 2;
            R.lbase = c->ev_lbase.as<u32>(); R.order = c->ev_lorder.as<u32>(); R.spos = c->ev_spos.as<u32>();
            R.e_in0 = c->e_in0.as<u32>(); R.e_in1 = c->e_in1.as<u32>(); R.e_out = c->e_out.as<u32>(); R.e_op = c->e_op.as<u8>();
            R.goff = c->goff.as<u64>(); R.segs = nullptr; R.b_in0 = c->p_in0.as<u32>(); R.b_in1 = c->p_in1.as<u32>(); R.b_out = c->p_out.as<u32>(); R.b_op = c->p_op.as<u8>();
            R.aval = c->ev_aval.as<u64>(); R.bval = c->ev_bval.as<u64>();
            ENSURE(c->ev_bar, 256);
            HIP_TRY(hipMemsetAsync(c->ev_bar.p, 0, 256, s));
            R.bar = c->ev_bar.as<u32>();
            u32 grid = 8;
#ifndef C2A_EMULATE
            {
                int per_cu = 0;
                if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_eval_pruned, kThreads, 0) != hipSuccess || per_cu < 1) per_cu = 1;
                grid = (u32)c->n_cu * (u32)std::min(per_cu, 4);
            }
#endif
            C2A_LAUNCH_CONCURRENT(k_eval_pruned, grid, kThreads, s, R, (const u32*)c->pr_goff.as<u32>());
        }
    } else if ((r = eval_run(c, which ? 2u : 1u, width))) return r;
    if (c->n_out) {
        C2A_LAUNCH_NOSYNC(k_eval_get, grid_for(out_words, 4096), kThreads, s, c->n_out, n_vectors, width, M, c->binfo.m_wires, out_base, which,
                          (const u64*)c->ev_aval.as<u64>(), (const u64*)c->ev_bval.as<u64>(), d_out);
        HIP_TRY(hipMemcpyAsync(outputs, d_out, out_words * 8, hipMemcpyDeviceToHost, s));
    }
    HIP_TRY(hipStreamSynchronize(s));
    if (c->n && (r = barrier_gave_up(c, "c2a_eval"))) return r;
    return C2A_OK;
}

// the gate lines of `count` gates (SoA pointers already at the first of them) on the CURRENT device: lengths, offsets, print,
// copy out.  text == nullptr: the size only.
struct FmtBufs { DevBuf* len; DevBuf* off; DevBuf* text; DevBuf* table; DevBuf* scan_tmp; };
static int fmt_range(c2a_ctx* c, hipStream_t s, const FmtBufs& B, const FmtTable& T, const u32* in0, const u32* in1, const u32* out, const u8* op,
                     u64 count, char* text, u64 capacity, u64* bytes_out) {
    *bytes_out = 0;
    if (count == 0) return C2A_OK;
    ENSURE(*B.table, sizeof(FmtTable)); ENSURE(*B.len, count * 4); ENSURE(*B.off, (count + 1) * 8);
    HIP_TRY(hipMemcpyAsync(B.table->p, &T, sizeof(T), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    const FmtTable* dT = B.table->as<FmtTable>();
    C2A_LAUNCH_NOSYNC(k_fmt_len, grid_for(count, 4096), kThreads, s, count, in0, in1, out, op, dT, B.len->as<u32>());
    int r = scan_1pass<1>(c, s, *B.scan_tmp, count, ScanFromU32{B.len->as<u32>()}, B.off->as<u64>(), (u64*)nullptr);
    if (r) return r;
    u64 bytes = 0;
    HIP_TRY(hipMemcpyAsync(&bytes, B.off->as<u64>() + count, 8, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    *bytes_out = bytes;
    if (!text) return C2A_OK;                        // size query
    if (bytes > capacity) return fail(c, C2A_ERR_ARG, "c2a_format_bristol: buffer too small (" + std::to_string(bytes) + " more bytes needed here)");
    ENSURE(*B.text, bytes);
    C2A_LAUNCH_NOSYNC(k_fmt_write, grid_for(count, 4096), kThreads, s, count, in0, in1, out, op, dT, (const u64*)B.off->as<u64>(), B.text->as<char>());
    HIP_TRY(hipMemcpyAsync(text, B.text->p, bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return C2A_OK;
}

int c2a_format_bristol(c2a_ctx* c, int which, uint64_t first, uint64_t count, char* text, uint64_t capacity, uint64_t* written) {
    if (!c || !written) return C2A_ERR_ARG;
    *written = 0;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const u32 *in0 = nullptr, *in1 = nullptr, *out = nullptr;
    const u8* op = nullptr;
    u64 total = 0;
    bool boolean = which != 0;
    switch (which) {
    case 0:
        if (c->stage < ST_EMITTED) return fail(c, C2A_ERR_STATE, "c2a_format_bristol: call c2a_emit_gates / c2a_build_circuit first");
        in0 = c->e_in0.as<u32>(); in1 = c->e_in1.as<u32>(); out = c->e_out.as<u32>(); op = c->e_op.as<u8>(); total = c->n; break;
    case 1:
        if (c->stage < ST_BOOLIFIED) return fail(c, C2A_ERR_STATE, "c2a_format_bristol: call c2a_boolify first");
        in0 = c->b_in0.as<u32>(); in1 = c->b_in1.as<u32>(); out = c->b_out.as<u32>(); op = c->b_op.as<u8>(); total = c->binfo.n_gates; break;
    case 2: {
        if (!c->bool_planned || !c->fmt_chunk_valid) return fail(c, C2A_ERR_STATE, "c2a_format_bristol: call c2a_boolify_chunk first");
        const u64 lead = c->fmt_chunk_first - (c->fmt_chunk_first & ~15ull);
        in0 = c->cb_in0.as<u32>() + lead; in1 = c->cb_in1.as<u32>() + lead; out = c->cb_out.as<u32>() + lead; op = c->cb_op.as<u8>() + lead;
        total = c->fmt_chunk_cnt; break;
    }
    default: return fail(c, C2A_ERR_ARG, "c2a_format_bristol: which must be 0 (arithmetic), 1 (boolean) or 2 (last boolean chunk)");
    }
    if (first > total || count > total - first) return fail(c, C2A_ERR_ARG, "c2a_format_bristol: range out of bounds");
    if (count == 0) return C2A_OK;
    if (count >= (1ull << 32)) return fail(c, C2A_ERR_ARG, "c2a_format_bristol: at most 2^32 - 1 gates per call");
    // op names: AGateType Display strings (a_gate_type.rs:6-28, compiler.rs:462) / XOR AND INV
    FmtTable T;
    std::memset(&T, 0, sizeof(T));
    if (boolean) {
        const char* nm[3] = {"XOR", "AND", "INV"};
        for (int i = 0; i < 3; ++i) { T.len[i] = 3; std::memcpy(T.name[i], nm[i], 3); }
        T.unary[C2A_INV] = 1;
    } else {
        const char* nm[20] = {"AAdd", "ADiv", "AEq", "AGEq", "AGt", "ALEq", "ALt", "AMul", "ANeq", "ASub", "AXor", "APow", "AIntDiv", "AMod",
                              "AShiftL", "AShiftR", "ABoolOr", "ABoolAnd", "ABitOr", "ABitAnd"};
        for (int i = 0; i < 20; ++i) { T.len[i] = (u8)std::strlen(nm[i]); std::memcpy(T.name[i], nm[i], T.len[i]); }
    }
    const FmtBufs own{&c->fmt_len, &c->fmt_off, &c->fmt_text, &c->fmt_table, &c->scan_tmp};
    if (which != 1 || c->peers.empty()) {
        u64 bytes = 0;
        int r = fmt_range(c, s, own, T, in0 + first, in1 + first, out + first, op + first, count, text, capacity, &bytes);
        *written = bytes;
        return r;
    }
    // the boolean circuit of a multi-device context: every device prints the gates it holds (no gather on the primary device),
    // the pieces land in the caller's buffer one behind the other
    DeviceGuard guard(c->device);
    const u64 last = first + count;
    u64 total_bytes = 0;
    {
        const u64 hi = std::min(last, c->shard0_qhi);
        if (first < hi) {
            u64 bytes = 0;
            int r = fmt_range(c, s, own, T, in0 + first, in1 + first, out + first, op + first, hi - first, text, capacity, &bytes);
            if (r) { *written = total_bytes + bytes; return r; }
            total_bytes += bytes;
        }
    }
    for (PeerDev& P : c->peers) {
        const u64 lo = std::max<u64>(first, P.q_lo), hi = std::min<u64>(last, P.q_hi);
        if (lo >= hi) continue;
        HIP_TRY(hipSetDevice(P.device));
        const u64 src = lo - P.q_bias;
        const FmtBufs pb{&P.fmt_len, &P.fmt_off, &P.fmt_text, &P.fmt_table, &P.scan_tmp};
        u64 bytes = 0;
        int r = fmt_range(c, P.stream, pb, T, P.b_in0.as<u32>() + src, P.b_in1.as<u32>() + src, P.b_out.as<u32>() + src, P.b_op.as<u8>() + src, hi - lo,
                          text ? text + total_bytes : nullptr, text ? capacity - std::min<u64>(capacity, total_bytes) : 0, &bytes);
        if (r) { *written = total_bytes + bytes; return r; }
        total_bytes += bytes;
    }
    *written = total_bytes;
    return C2A_OK;
}

int c2a_debug_peel_abort(c2a_ctx* c, uint32_t launches) {
    if (!c) return C2A_ERR_ARG;
    c->debug_peel_abort = launches;
    return C2A_OK;
}

int c2a_debug_set_build_no(c2a_ctx* c, uint32_t build_no) {
    if (!c) return C2A_ERR_ARG;
    if (c->stage < ST_LOADED || build_no >= (1u << 24)) return fail(c, C2A_ERR_ARG, "c2a_debug_set_build_no: load gates first; build numbers have 24 bits");
    HIP_TRY(hipSetDevice(c->device));
    // (records of builds up to this number may be in the table: start from a clean one, as after a load)
    HIP_TRY(hipMemsetAsync(c->nrec.p, 0, (size_t)c->n_nodes * 16, c->stream));
    c->build_no = build_no;
    return C2A_OK;
}

int c2a_debug_hot_every(c2a_ctx* c, uint32_t ticket) {
    if (!c || ticket < 2 || (ticket & (ticket - 1)) != 0) return C2A_ERR_ARG;
    c->hot_every = ticket; c->hot_every_forced = true;
    return C2A_OK;
}

int c2a_debug_patch_bool_op(c2a_ctx* c, uint64_t index, uint8_t new_op) {
    if (!c) return C2A_ERR_ARG;
    if (c->stage < ST_BOOLIFIED || index >= c->binfo.n_gates || new_op > C2A_INV) return fail(c, C2A_ERR_ARG, "c2a_debug_patch_bool_op: bad index / op");
    c->gathered = false;
    if (c->peers.empty() || index < c->shard0_qhi) {
        HIP_TRY(hipSetDevice(c->device));
        HIP_TRY(hipMemcpyAsync(c->b_op.as<u8>() + index, &new_op, 1, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return C2A_OK;
    }
    DeviceGuard guard(c->device);
    for (PeerDev& P : c->peers) {
        if (index < P.q_lo || index >= P.q_hi) continue;
        HIP_TRY(hipSetDevice(P.device));
        HIP_TRY(hipMemcpyAsync(P.b_op.as<u8>() + (index - P.q_bias), &new_op, 1, hipMemcpyHostToDevice, P.stream));
        HIP_TRY(hipStreamSynchronize(P.stream));
    }
    return C2A_OK;
}

int c2a_get_timings(c2a_ctx* c, c2a_timings* t) {
    if (!c || !t) return C2A_ERR_ARG;
    (void)hipStreamSynchronize(c->stream);
    t->prep = elapsed(c, EV_PREP0, EV_PREP1);
    t->peel = elapsed(c, EV_PREP1, EV_PEEL1);
    t->order = elapsed(c, EV_PEEL1, EV_ORDER1);
    t->wires = elapsed(c, EV_WIRES0, EV_WIRES1);
    t->emit = elapsed(c, EV_EMIT0, EV_EMIT1);
    t->bool_prep = elapsed(c, EV_BPREP0, EV_BPREP1);
    t->bool_map = elapsed(c, EV_BPREP1, EV_BMAP1);
    t->build_total = elapsed(c, EV_BUILD0, EV_BUILD1);
    t->boolify_total = elapsed(c, EV_BPREP0, EV_BMAP1);
    t->k_peel = c->serial_fallback ? 0.f : elapsed(c, EV_KPEEL0, EV_KPEEL1);
    return C2A_OK;
}

int c2a_get_stats(c2a_ctx* c, c2a_stats* s) {
    if (!c || !s) return C2A_ERR_ARG;
    *s = c->stats;
    s->path_chunks = c->stats.max_depth ? (c->stats.max_depth - 1) / kChunkBits + 1 : 1;
    return C2A_OK;
}

}  // extern "C"

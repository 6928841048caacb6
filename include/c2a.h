/*
 * c2a.h — C ABI of the MI355X-native back end for the flat-gate-graph stage of circom-2-arithc.
 *
 * The reference (Rust, /root/reference) has no FFI seam; this header IS the seam a maintainer binds
 * (INTEGRATION.md shows the Rust `extern "C"` block).  Every entry point names the reference code it
 * replaces.  Conventions:
 *   - plain pointers and sizes only; strings never cross the ABI (the host keeps the name tables of
 *     src/compiler.rs:323-361 and rebuilds CircuitInfo from node_to_wire);
 *   - caller-owned inputs are copied during c2a_load_gates and may be freed afterwards;
 *     library-owned device results live until the next c2a_load_gates / c2a_destroy;
 *   - every call returns an int status: 0 = OK, >0 = a reference-level error (CircuitError),
 *     <0 = argument / state / HIP failure; c2a_last_error() gives a message; no exceptions, no aborts;
 *   - one context per host thread; contexts are independent (one HIP stream each).
 *   - node ids are the reference's raw u32 node ids (sparse, src/compiler.rs:497-500); n_nodes is
 *     max node id + 1 (the library direct-addresses them: no remap needed).
 */
#ifndef C2A_H
#define C2A_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct c2a_ctx c2a_ctx;

enum c2a_status {
    C2A_OK = 0,
    C2A_ERR_CYCLIC = 1,          /* CircuitError::CyclicDependency (src/compiler.rs:570-571); *cycle_at = i of "detected at i={}" (src/topological_sort.rs:34-38) */
    C2A_ERR_INCONSISTENCY = 2,   /* CircuitError::Inconsistency: a node is both input and output (src/compiler.rs:363-383) */
    C2A_ERR_OVERFLOW = 3,        /* boolean wire ids would not fit u32 */
    C2A_ERR_ARG = -1,
    C2A_ERR_STATE = -2,          /* call order violated (load -> topo_sort -> assign_wires -> emit -> boolify) */
    C2A_ERR_HIP = -3,
    C2A_ERR_NOMEM = -4
};

/* AGateType discriminants == declaration order of src/a_gate_type.rs:8-27 */
enum c2a_gate_type {
    C2A_AAdd = 0, C2A_ADiv, C2A_AEq, C2A_AGEq, C2A_AGt, C2A_ALEq, C2A_ALt, C2A_AMul, C2A_ANeq, C2A_ASub,
    C2A_AXor, C2A_APow, C2A_AIntDiv, C2A_AMod, C2A_AShiftL, C2A_AShiftR, C2A_ABoolOr, C2A_ABoolAnd,
    C2A_ABitOr, C2A_ABitAnd, C2A_NUM_GATE_TYPES
};

/* boolean ops of the frozen bit-blast spec: Bristol-fashion XOR / AND / INV */
enum c2a_bool_op { C2A_XOR = 0, C2A_AND = 1, C2A_INV = 2 };

#define C2A_NO_WIRE 0xFFFFFFFFu

typedef struct c2a_bool_info {
    uint64_t n_gates;        /* boolean gates */
    uint64_t wire_count;     /* boolean wires = arithmetic wire_count*width + aux_total */
    uint64_t aux_total;      /* template-internal wires */
    uint32_t width;
    uint32_t n_in, n_out;    /* arithmetic IO counts; io_widths = ([width; n_in], [width; n_out]) */
    uint32_t m_wires;        /* M = arithmetic wire_count - n_out; bit b of wire W is W*width+b for W<M,
                                else M*width + aux_total + (W-M)*width + b */
} c2a_bool_info;

typedef struct c2a_timings {     /* milliseconds, HIP events on the context's stream, last run */
    float prep;                  /* producer map, relabelling by out-node order, deps, consumer lists */
    float peel;                  /* reverse Kahn peel + DFS-tree parent selection (dataflow launch, or all levels) */
    float order;                 /* Euler tour + list ranking -> sorted_gate_ids */
    float wires;                 /* first-seen wire numbering */
    float emit;                  /* gate emission */
    float bool_prep;             /* template sizes + offset scans */
    float bool_map;              /* the boolify map kernel alone (dominant kernel; roofline) */
    float build_total;           /* c2a_build_circuit wall (events) */
    float boolify_total;         /* c2a_boolify wall (events) */
    float k_peel;                /* the dataflow launch alone (an event pair around k_peel: `peel` also holds the sinks pass and the
                                    whole-level passes in front of it); 0 when the serial fall-back sorted */
} c2a_timings;

typedef struct c2a_stats {
    uint64_t n_gates;
    uint64_t n_edges;            /* dependency edges after dedupe */
    uint32_t levels;             /* reverse Kahn levels */
    uint32_t max_depth;          /* length of the longest path string of the DFS tree = its depth, plus one for every relay level under a
                                    hub on that path (n_relays below; equal to the depth of the DFS tree when the circuit has no hub) */
    uint32_t n_roots;            /* DFS roots (children of the virtual root) */
    uint32_t n_splitters;        /* list-ranking sublists */
    uint32_t level_launches;     /* peel kernel launches: 2 (the sinks pass + the one dataflow launch) */
    uint32_t peel_waves;         /* single-wave workgroups of the dataflow launch (CUs x min(knobs, occupancy query): eight per CU + eight parked in reserve) */
    uint32_t path_chunks;        /* path-string chunks (61 words of 62 bits: 3782 bits) the deepest DFS path spans (1 = every tournament is one round trip) */
    uint32_t peel_rereads;       /* times a wave read a candidate record again because a word of it had not arrived yet */
    uint32_t numbering_events;   /* what shifts the wire numbering away from "sorted position q gets wire n_in + q": gates whose out node
                                    is an IO node + constant-like nodes (un-produced, no IO node) — src/compiler.rs:431-438; counted by
                                    the positional numbering (numbering_path == 1; 0 otherwise), any number of them */
    uint32_t numbering_path;     /* 1: positional numbering (every node has one writer: wires and gates by formula from the sorted
                                    positions and three event bits per position); 0: the walk in sorted order (duplicate writers, a
                                    serial sort, or C2A_NUMBERING_WALK=1) */
    uint32_t n_relays;           /* virtual gates the sort added under HUBS — produced nodes with more than 16 consumers (a broadcast
                                    selector, a scale factor: src/compiler.rs:408-421 allows any fan-out): a hub's consumers are compared
                                    sixteen at a time by a tree of relays instead of one after the other (0 for a circuit without hubs) */
    uint32_t verifier;           /* which check the last c2a_verify_boolify ran: 1 = both circuits simulated wire by wire on 64 vectors
                                    (single-device context), 2 = every device checked its own gates locally (multi-device context:
                                    64 vectors through each gate's template, every wire it names must be its own); 0 = none yet */
} c2a_stats;

/*
 * Create a context on the HIP devices device_ids[0..n_devices).  device_ids[0] is the primary device: the
 * sort, the wire numbering and the emission run there (the sort is a chain of ~5 000 dependent steps and does not shard,
 * DESIGN.md §7); c2a_boolify cuts the emitted circuit by sorted-position range and bit-blasts one range per device
 * (peer copies of the slices, one stream per device, no collective).  n_devices == 1 is the single-GPU context.
 * Fails with C2A_ERR_HIP when there is no device / runtime, C2A_ERR_ARG for an id out of range.
 */
int c2a_create(int n_devices, const int* device_ids, c2a_ctx** ctx);
/* HIP devices this process can list in c2a_create (0 when there is no runtime / no GPU; never fails). */
int c2a_visible_devices(void);
int c2a_device_count(const c2a_ctx* ctx);
void c2a_destroy(c2a_ctx* ctx);
const char* c2a_last_error(const c2a_ctx* ctx);
const char* c2a_version(void);
/* Bumped whenever a signature or a struct layout of this header changes (round 2 changed c2a_create and c2a_stats without
 * a signal): a binding built against another header must refuse to go on.  c2a_abi_version() == C2A_ABI_VERSION. */
#define C2A_ABI_VERSION 7
int c2a_abi_version(void);

/*
 * Marshal `Compiler.gates` (src/compiler.rs:113, :85-90) as SoA plus the IO node lists that
 * build_circuit derives at src/compiler.rs:323-361 (in the canonical order of DESIGN.md §3; the
 * reference iterates a std HashMap there).  Host pointers; copied to HBM.  Also sizes all workspace.
 */
int c2a_load_gates(c2a_ctx* ctx, uint64_t n, const uint32_t* lh, const uint32_t* rh, const uint32_t* out,
                   const uint8_t* op, uint32_t n_nodes, uint32_t n_in, const uint32_t* input_nodes, uint32_t n_out,
                   const uint32_t* output_nodes);

/*
 * The SECOND half of the path alone: marshal an arithmetic BristolCircuit the host built itself (the reference's own
 * Compiler::build_circuit, src/compiler.rs:321-494: `gates` in their emitted order as SoA of wire ids + AGateType, `wire_count`,
 * and how many of the first / last wires are inputs / outputs) — what `boolify(&circuit, width)` takes (src/main.rs:30-32).
 * Behind it: c2a_boolify / c2a_boolify_plan + _chunk / c2a_bool_read / c2a_format_bristol / c2a_checksum; the calls that sort,
 * number or need the sort's level data (c2a_topo_sort, c2a_assign_wires, c2a_build_circuit, c2a_eval, c2a_verify_boolify on a
 * single device, c2a_boolify_prune) answer C2A_ERR_STATE.  For circuits that are DEEP AND NARROW — a chain of hash blocks: a few
 * gates per dependency level — the sort is bound by its ~1.3 us per level on the GPU and a CPU core's DFS is faster, while the
 * bit-blast is 10^2-10^4 x faster here: INTEGRATION.md "When to dispatch".
 * C2A_ERR_ARG: a wire id >= wire_count, an unknown gate type, n_in + n_out > wire_count.
 */
int c2a_load_circuit(c2a_ctx* ctx, uint64_t n, const uint32_t* in0, const uint32_t* in1, const uint32_t* out, const uint8_t* op,
                     uint32_t wire_count, uint32_t n_in, uint32_t n_out);

/*
 * == topological_sort(len, get_deps) (src/topological_sort.rs:3-21) with the deps closure of
 * src/compiler.rs:408-421.  sorted_gate_ids (host, n entries) may be NULL (result stays in HBM).
 * On C2A_ERR_CYCLIC *cycle_at is the gate index of the reference's message.
 * n + 4 n / 9 + 64 < 2^29, i.e. n < 371 M (C2A_ERR_ARG beyond: the sort's ticket words — one per gate and per relay of a hub, the
 * virtual gates a producer with more than 8 consumers gets — are addressed by 32-bit byte offsets; its workspace of
 * ~1.4 KB per gate runs out of a 288 GB device before that).
 */
int c2a_topo_sort(c2a_ctx* ctx, uint32_t* sorted_gate_ids, uint64_t* cycle_at);

/* Same contract, executed as the literal DFS on one GPU lane (diagnostics / cross-check; slow: measured on an MI355X 2.1-2.4 us per
 * gate — 2.1 s for 1 M gates, 24 s for 10 M, tools/serial_time.py; a host core does the same DFS in 41 ms / 1.7 s).  c2a_topo_sort
 * falls back to it by itself when its dataflow launch gives up twice (a watchdog, never seen on hardware outside the tests'
 * c2a_debug_peel_abort: 2.7 s / 31 s then, the serial walk for the levels included): like the reference's sort
 * (src/topological_sort.rs:3-21) it cannot fail on an acyclic graph. */
int c2a_topo_sort_serial(c2a_ctx* ctx, uint32_t* sorted_gate_ids, uint64_t* cycle_at);

/*
 * == wire numbering of src/compiler.rs:388-449: inputs first (list order), then first-seen along the
 * sorted gates, outputs last (list order).  node_to_wire (host, n_nodes entries, C2A_NO_WIRE = none)
 * may be NULL.  *wire_count == BristolCircuit.wire_count (src/compiler.rs:479).
 */
int c2a_assign_wires(c2a_ctx* ctx, uint32_t* node_to_wire, uint32_t* wire_count);

/* == gate emission of src/compiler.rs:451-464 in sorted order; any host pointer may be NULL. */
int c2a_emit_gates(c2a_ctx* ctx, uint32_t* in0, uint32_t* in1, uint32_t* out, uint8_t* op);

/* == Compiler::build_circuit numeric core (src/compiler.rs:385-464): the three calls above back to
 * back with everything left resident in HBM (this is the timed unit of bench.py). */
int c2a_build_circuit(c2a_ctx* ctx, uint64_t* cycle_at, uint32_t* wire_count);

/*
 * == boolify(&circuit, width) (src/main.rs:30-32), under the frozen bit-blast spec of DESIGN.md §5
 * (the crate's source is absent: parity with it is unpinned).  Result stays in HBM as SoA
 * in0[]/in1[]/out[]/op[] — on a multi-device context each device keeps the boolean gates of its own range; read ranges
 * back with c2a_bool_read (it gathers from the owners).  1 <= width <= 64.
 */
int c2a_boolify(c2a_ctx* ctx, uint32_t width, c2a_bool_info* info);
int c2a_bool_read(c2a_ctx* ctx, uint64_t first, uint64_t count, uint32_t* in0, uint32_t* in1, uint32_t* out,
                  uint8_t* op);

/*
 * Chunked emission, for boolean circuits larger than the HBM left over (or to stream them to disk / to shard them
 * over GPUs by sorted-position range): c2a_boolify_plan computes sizes, offsets and the wire layout only;
 * c2a_boolify_chunk then bit-blasts the arithmetic gates at sorted positions [first_gate, first_gate + n_gates) into a
 * chunk-sized buffer and copies them to the host arrays (any may be NULL).  The gates are identical to the
 * corresponding range [*first_bool_gate, +*n_bool_gates) of the full c2a_boolify result.
 */
int c2a_boolify_plan(c2a_ctx* ctx, uint32_t width, c2a_bool_info* info);
int c2a_boolify_chunk(c2a_ctx* ctx, uint64_t first_gate, uint64_t n_gates, uint32_t* in0, uint32_t* in1, uint32_t* out,
                      uint8_t* op, uint64_t* first_bool_gate, uint64_t* n_bool_gates);

/*
 * Where to cut: sorted positions [*first_gate, *first_gate + *n_gates) of shard `k` of `n_shards` ranges holding (nearly) equal
 * numbers of BOOLEAN gates — the bytes a shard writes; equal arithmetic counts would not balance a circuit with clustered
 * multipliers (T(AMul, 32) = 2 824 boolean gates, T(AXor, 32) = 32).  Needs c2a_boolify_plan / c2a_boolify; this is the cut a
 * multi-device context makes itself, exported for callers that shard over processes (bench.py --gpus N: one rank per GPU).
 */
int c2a_boolify_shard_range(c2a_ctx* ctx, uint32_t k, uint32_t n_shards, uint64_t* first_gate, uint64_t* n_gates);

/*
 * == the gate lines of BristolCircuit::write_bristol (src/main.rs:34-35; crate absent: Bristol-fashion text per SURVEY C.2),
 * printed on the GPU: "2 1 <in0> <in1> <out> <OP>\n", "1 1 <in0> <out> INV\n" for the one-input op.  Gates
 * [first, first + count) of  which = 0: the arithmetic circuit (c2a_emit_gates);  1: the boolean circuit (c2a_boolify; on a
 * multi-device context every device prints the gates it holds, nothing is gathered);  2: the chunk of the last c2a_boolify_chunk.  text == NULL only queries *written (bytes).
 * The header lines (gate / wire counts, io widths) are the host's: it knows the name tables.
 */
int c2a_format_bristol(c2a_ctx* ctx, int which, uint64_t first, uint64_t count, char* text, uint64_t capacity, uint64_t* written);

/* T(op,width) and AUX(op,width) of the frozen spec (host-side query; no GPU work). */
int c2a_template_size(uint32_t op, uint32_t width, uint64_t* n_gates, uint64_t* n_aux);

/* Position-salted 64-bit checksums computed on the GPU over resident results (full-size parity):
 * which: 0 sorted_gate_ids, 1 arith in0, 2 arith in1, 3 arith out, 4 arith op,
 *        5 bool in0, 6 bool in1, 7 bool out, 8 bool op, 9 node_to_wire */
int c2a_checksum(c2a_ctx* ctx, int which, uint64_t* value);

/*
 * Functional check at scale — the reference's simulation harness (tests/integration.rs:191-237) as HIP kernels:
 * evaluates the arithmetic circuit of c2a_build_circuit and the boolean circuit of c2a_boolify on 64 seeded
 * pseudo-random input vectors and compares EVERY arithmetic wire with its `width` boolean wires.
 * *n_checked = wire_count * 64 (wire, vector) pairs, *n_mismatch = how many differ (0 when the bit-blast is right).
 * Needs ~8 B per (arithmetic wire x 64) + 8 B per boolean wire of scratch HBM; requires c2a_boolify.
 * On a MULTI-DEVICE context nothing is gathered: every device checks the gates it holds, gate by gate — 64 vectors (a function
 * of the input wire) bit-sliced onto the boolean wires the gate's inputs must occupy, its boolean gates run out of a private
 * scratch (every wire they name must be the gate's own A / B / O bit or aux wire), the O bits compared with the arithmetic
 * op.  *n_checked = n_gates * 64 (gate, vector) pairs there.
 */
int c2a_verify_boolify(c2a_ctx* ctx, uint64_t seed, uint64_t* n_checked, uint64_t* n_mismatch);
/*
 * OPTIONAL prune pass over the circuit of c2a_boolify: what the `boolify` crate is believed to do
 * after its per-gate blast — constant folding (XOR(a,a) = 0, XOR(a,0) = a, XOR(a,1) = INV a, AND(a,a) = a,
 * AND(a,0) = 0, AND(a,1) = a, INV of a constant) and removal of gates no circuit output depends on.  The frozen per-gate map
 * stays what c2a_boolify returns (and what the metric measures); this produces a second, smaller, functionally equal circuit:
 * gates 0 and 1 make the two constant wires (zero_wire = XOR(w0, w0), one_wire = INV zero_wire), the live gates follow in
 * the order of the original; wires keep their ids (wire_count grows by the two constants).  c2a_eval(which = 2) runs it.
 */
typedef struct c2a_prune_info {
    uint64_t n_gates;          /* gates of the pruned circuit (incl. the two constant gates) */
    uint64_t n_gates_before;   /* gates of the circuit of c2a_boolify */
    uint64_t n_folded;         /* gates replaced by a constant or by another wire */
    uint64_t n_dead;           /* gates that no output depends on */
    uint64_t wire_count;       /* boolean wire_count + 2 */
    uint32_t zero_wire, one_wire;
} c2a_prune_info;
int c2a_boolify_prune(c2a_ctx* ctx, c2a_prune_info* info);
int c2a_pruned_read(c2a_ctx* ctx, uint64_t first, uint64_t count, uint32_t* in0, uint32_t* in1, uint32_t* out, uint8_t* op);

/*
 * == the reference's simulation harness with CALLER-SUPPLIED values (tests/integration.rs:191-237: named inputs in, named
 * outputs out), level-parallel on the GPU.  which = 0: the arithmetic circuit of c2a_build_circuit, evaluated mod 2^width
 * (tests/integration.rs:94-115 where that is defined, DESIGN.md §5.2 elsewhere);  1: the boolean circuit of c2a_boolify
 * (width is the boolify width; the values are bit-sliced onto the boolean wires and read back from them);  2: the pruned
 * circuit of c2a_boolify_prune, same convention.
 * inputs[i * n_vectors + t]: value of input wire i (wires 0 .. n_in-1 in the order of c2a_load_gates' input list) in vector
 * t;  n_const constants given as (ARITHMETIC wire id, value) — the host knows them from its name tables (ConstantInfo);
 * outputs[j * n_vectors + t]: value of output j (the last n_out wires).  1 <= n_vectors <= 64.  Wires nothing drives are 0.
 * C2A_ERR_STATE for a circuit in which two gates write one node (the reference keeps the last, src/compiler.rs:403-406): a
 * writer that is not the producer carries no dependency edge, the level-parallel passes (this one, c2a_verify_boolify,
 * c2a_boolify_prune) would race it against the wire's readers — evaluate the emitted gate list sequentially instead.
 * On a MULTI-DEVICE context which = 1 and c2a_boolify_prune run on the primary device and read every device's piece of the
 * boolean circuit where it lies (peer access, enabled at c2a_create; gathered on the primary only where two devices cannot map
 * each other's memory).
 */
int c2a_eval(c2a_ctx* ctx, int which, uint32_t width, uint32_t n_vectors, const uint64_t* inputs, uint32_t n_const,
             const uint32_t* const_wires, const uint64_t* const_values, uint64_t* outputs);
/* Fault injection for the tests of the verifier: overwrite the op of one boolean gate in HBM. */
int c2a_debug_patch_bool_op(c2a_ctx* ctx, uint64_t index, uint8_t new_op);
/* Fault injection for the tests of the sort that cannot fail (src/topological_sort.rs:3-21 always terminates on an acyclic graph):
 * the next `launches` dataflow launches are treated as if their watchdog had tripped — one: the retry on clean buffers; two: the
 * serial DFS takes over (c2a_topo_sort then costs what c2a_topo_sort_serial costs: one lane, 2.1-2.4 us per gate). */
int c2a_debug_peel_abort(c2a_ctx* ctx, uint32_t launches);
/* Tests: the number the next build's node-table records are tagged with follows `build_no` (24 bits; the table is cleared when the
 * numbers wrap).  After c2a_load_gates. */
int c2a_debug_set_build_no(c2a_ctx* ctx, uint32_t build_no);
/* Tests: a producer counts as HOT — its consumers take their list slots one range per wave instead of one same-address atomic each —
 * from its `ticket`-th consumer on (a power of two; default 16 384). */
int c2a_debug_hot_every(c2a_ctx* ctx, uint32_t ticket);

int c2a_get_timings(c2a_ctx* ctx, c2a_timings* t);
int c2a_get_stats(c2a_ctx* ctx, c2a_stats* s);

#ifdef __cplusplus
}
#endif
#endif /* C2A_H */

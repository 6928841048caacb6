/*
 * c2a_oracle.c — CPU restatement of the flat-gate-graph stage of circom-2-arithc.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (circom-2-arithc_amd/, include/)
 * may link, load or call this file.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it — as the checker / the timed CPU baseline, never as the thing shipped.
 *
 * Parity status
 *   - topological sort + build_circuit numeric core: follows the reference source literally
 *     (src/topological_sort.rs:3-50, src/compiler.rs:385-493).  The reference (Rust, 14 un-vendored
 *     git dependencies, no toolchain in this image) cannot be built here, so the restatement is
 *     pinned against the reference's own test expectations (tests/integration.rs:279-441) and the
 *     hand-traced flat gate lists of SURVEY.md Appendix A (tests/golden/).
 *   - boolify: the `boolify` crate (github.com/voltrevo/boolify rev 6376405, Cargo.toml:21) is
 *     absent from /root/reference and no reference test exercises it => PARITY UNPINNED for the
 *     gate-level boolean output.  This file implements the frozen bit-blast spec of DESIGN.md §5
 *     (procedural form); functional equivalence with the arithmetic semantics of
 *     tests/integration.rs:94-115 is what the tests prove.
 *
 * Two modes of the build_circuit core (same results, different cost model):
 *   mode 0 "faithful": hash maps for node->wire / node->producer / output set and a heap-allocated
 *                      dependency vector per DFS visit, like compiler.rs:385-449 + :408-421.
 *   mode 1 "flat":     dense arrays indexed by node id, no allocation in the loop.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_NONE 0xFFFFFFFFu

enum { ORC_OK = 0, ORC_CYCLIC = 1, ORC_INCONSISTENCY = 2, ORC_OVERFLOW = 3, ORC_ARG = 4, ORC_NOMEM = 5 };

/* AGateType discriminants, declaration order of src/a_gate_type.rs:8-27 */
enum {
    AAdd = 0, ADiv, AEq, AGEq, AGt, ALEq, ALt, AMul, ANeq, ASub, AXor, APow, AIntDiv, AMod,
    AShiftL, AShiftR, ABoolOr, ABoolAnd, ABitOr, ABitAnd, A_NUM_OPS
};

static const char* const k_op_names[A_NUM_OPS] = {
    "AAdd", "ADiv", "AEq", "AGEq", "AGt", "ALEq", "ALt", "AMul", "ANeq", "ASub", "AXor", "APow",
    "AIntDiv", "AMod", "AShiftL", "AShiftR", "ABoolOr", "ABoolAnd", "ABitOr", "ABitAnd"};

const char* orc_op_name(int op) { return (op >= 0 && op < A_NUM_OPS) ? k_op_names[op] : ""; }

/* boolean ops of the frozen bit-blast spec */
enum { B_XOR = 0, B_AND = 1, B_INV = 2 };

/* ------------------------------------------------------------------------------------------ */
/* u32 -> u32 hash map (open addressing), stands in for std::collections::HashMap in mode 0    */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    uint32_t* keys;
    uint32_t* vals;
    uint64_t mask;
} orc_map;

static uint64_t mix64(uint64_t x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
    return x;
}

static int map_init(orc_map* m, uint64_t expected) {
    uint64_t cap = 16;
    while (cap < expected * 2 + 2) cap <<= 1;
    m->keys = (uint32_t*)malloc(cap * sizeof(uint32_t));
    m->vals = (uint32_t*)malloc(cap * sizeof(uint32_t));
    if (!m->keys || !m->vals) return ORC_NOMEM;
    memset(m->keys, 0xFF, cap * sizeof(uint32_t)); /* key ORC_NONE == empty (node ids never reach it) */
    m->mask = cap - 1;
    return ORC_OK;
}
static void map_free(orc_map* m) { free(m->keys); free(m->vals); m->keys = m->vals = NULL; }
static void map_insert(orc_map* m, uint32_t k, uint32_t v) { /* insert or overwrite */
    uint64_t i = mix64(k) & m->mask;
    while (m->keys[i] != ORC_NONE && m->keys[i] != k) i = (i + 1) & m->mask;
    m->keys[i] = k; m->vals[i] = v;
}
static int map_get(const orc_map* m, uint32_t k, uint32_t* v) {
    uint64_t i = mix64(k) & m->mask;
    while (m->keys[i] != ORC_NONE) {
        if (m->keys[i] == k) { *v = m->vals[i]; return 1; }
        i = (i + 1) & m->mask;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------ */
/* topological_sort  (src/topological_sort.rs:3-50), recursion replaced by an explicit stack    */
/* with the identical visit order.                                                              */
/* get_deps is the closure of src/compiler.rs:408-421: [producer(lh) if any, producer(rh) if any]*/
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    const uint32_t *lh, *rh;
    int mode;
    const orc_map* prod_map;   /* mode 0 */
    const uint32_t* prod_flat; /* mode 1: producer[node] or ORC_NONE */
} deps_env;

/* returns number of deps (0..2) */
static int get_deps(const deps_env* e, uint64_t g, uint64_t d[2]) {
    int k = 0; uint32_t p;
    if (e->mode == 0) {
        if (map_get(e->prod_map, e->lh[g], &p)) d[k++] = p;   /* compiler.rs:412-414 */
        if (map_get(e->prod_map, e->rh[g], &p)) d[k++] = p;   /* compiler.rs:416-418 */
    } else {
        p = e->prod_flat[e->lh[g]]; if (p != ORC_NONE) d[k++] = p;
        p = e->prod_flat[e->rh[g]]; if (p != ORC_NONE) d[k++] = p;
    }
    return k;
}

typedef struct { uint64_t i; uint64_t* deps; uint8_t ndeps, next; uint64_t inl[2]; } dfs_frame;

static int topo_sort(uint64_t n, const deps_env* env, uint32_t* sorted, uint64_t* cycle_at) {
    uint8_t* visiting = (uint8_t*)calloc(n ? n : 1, 1);
    uint8_t* visited = (uint8_t*)calloc(n ? n : 1, 1);
    uint64_t cap = 1024, sp = 0, ns = 0;
    dfs_frame* st = (dfs_frame*)malloc(cap * sizeof(dfs_frame));
    int rc = ORC_OK;
    if (!visiting || !visited || !st) { rc = ORC_NOMEM; goto done; }
    for (uint64_t root = 0; root < n; ++root) {           /* topological_sort.rs:11-13 */
        uint64_t enter = root;
        int have = 1;
        while (have || sp) {
            if (have) {                                   /* topological_sort_visit(enter) */
                have = 0;
                if (visited[enter]) continue;             /* :30-32 */
                if (visiting[enter]) {                    /* :34-38 */
                    *cycle_at = enter; rc = ORC_CYCLIC; goto done;
                }
                visiting[enter] = 1;                      /* :40 */
                if (sp == cap) {
                    cap *= 2;
                    dfs_frame* t = (dfs_frame*)realloc(st, cap * sizeof(dfs_frame));
                    if (!t) { rc = ORC_NOMEM; goto done; }
                    st = t;
                }
                dfs_frame* f = &st[sp++];
                f->i = enter; f->next = 0;
                if (env->mode == 0) {                     /* Vec::<usize>::new() per call, compiler.rs:410 */
                    f->deps = (uint64_t*)malloc(2 * sizeof(uint64_t));
                    if (!f->deps) { --sp; rc = ORC_NOMEM; goto done; }
                    f->ndeps = (uint8_t)get_deps(env, enter, f->deps);
                } else {
                    f->deps = NULL;                       /* inl[] is used in place (frames move on realloc) */
                    f->ndeps = (uint8_t)get_deps(env, enter, f->inl);
                }
                continue;
            }
            dfs_frame* f = &st[sp - 1];
            if (f->next < f->ndeps) {                     /* :42-44 */
                enter = (env->mode == 0 ? f->deps : f->inl)[f->next++];
                have = 1;
            } else {
                sorted[ns++] = (uint32_t)f->i;            /* :46 */
                visited[f->i] = 1;                        /* :47 */
                if (env->mode == 0) free(f->deps);
                --sp;
            }
        }
    }
done:
    if (rc != ORC_OK && env->mode == 0) for (uint64_t k = 0; k < sp; ++k) free(st[k].deps);
    free(visiting); free(visited); free(st);
    return rc;
}

/* stand-alone entry: deps given explicitly (-1 = absent), used by unit tests of the DFS order */
int orc_topo_sort_deps(uint64_t n, const int64_t* dep0, const int64_t* dep1, uint32_t* sorted, uint64_t* cycle_at) {
    /* encode as a flat producer problem: node g+1 is produced by gate g; lh/rh name producers */
    uint32_t* lh = (uint32_t*)malloc((n ? n : 1) * 4), *rh = (uint32_t*)malloc((n ? n : 1) * 4);
    uint32_t* prod = (uint32_t*)malloc((n + 1) * 4);
    if (!lh || !rh || !prod) { free(lh); free(rh); free(prod); return ORC_NOMEM; }
    prod[0] = ORC_NONE;
    for (uint64_t g = 0; g < n; ++g) {
        prod[g + 1] = (uint32_t)g;
        lh[g] = dep0[g] < 0 ? 0 : (uint32_t)(dep0[g] + 1);
        rh[g] = dep1[g] < 0 ? 0 : (uint32_t)(dep1[g] + 1);
    }
    deps_env env = {lh, rh, 1, NULL, prod};
    int rc = topo_sort(n, &env, sorted, cycle_at);
    free(lh); free(rh); free(prod);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* build_circuit numeric core (src/compiler.rs:385-493)                                         */
/* Name handling (compiler.rs:323-383: IO/constant name maps, duplicate-name and in==out errors) */
/* is string work done by the caller; input_nodes / output_nodes arrive in the canonical order    */
/* of DESIGN.md §3 (the reference iterates a std HashMap there: order is not defined).           */
/* ------------------------------------------------------------------------------------------ */
typedef struct orc_circuit {
    uint64_t n;
    uint32_t n_nodes, n_in, n_out, wire_count;
    uint32_t* sorted;                 /* n  : sorted_gate_ids (compiler.rs:408)            */
    uint32_t *in0, *in1, *out;        /* n  : emitted gates in sorted order (compiler.rs:451-464) */
    uint8_t* op;                      /* n                                                  */
    uint32_t* node_wire;              /* n_nodes : node id -> wire id or ORC_NONE           */
} orc_circuit;

void orc_free_circuit(orc_circuit* c) {
    if (!c) return;
    free(c->sorted); free(c->in0); free(c->in1); free(c->out); free(c->op); free(c->node_wire); free(c);
}

int orc_build_circuit(uint64_t n, const uint32_t* lh, const uint32_t* rh, const uint32_t* out, const uint8_t* op,
                      uint32_t n_nodes, uint32_t n_in, const uint32_t* in_nodes, uint32_t n_out,
                      const uint32_t* out_nodes, int mode, orc_circuit** result, uint64_t* cycle_at) {
    *result = NULL;
    for (uint64_t g = 0; g < n; ++g)
        if (lh[g] >= n_nodes || rh[g] >= n_nodes || out[g] >= n_nodes || op[g] >= A_NUM_OPS) return ORC_ARG;
    for (uint32_t i = 0; i < n_in; ++i) if (in_nodes[i] >= n_nodes) return ORC_ARG;
    for (uint32_t i = 0; i < n_out; ++i) if (out_nodes[i] >= n_nodes) return ORC_ARG;

    orc_circuit* c = (orc_circuit*)calloc(1, sizeof(orc_circuit));
    if (!c) return ORC_NOMEM;
    uint64_t na = n ? n : 1;
    c->n = n; c->n_nodes = n_nodes; c->n_in = n_in; c->n_out = n_out;
    c->sorted = (uint32_t*)malloc(na * 4); c->in0 = (uint32_t*)malloc(na * 4); c->in1 = (uint32_t*)malloc(na * 4);
    c->out = (uint32_t*)malloc(na * 4); c->op = (uint8_t*)malloc(na);
    c->node_wire = (uint32_t*)malloc((size_t)(n_nodes ? n_nodes : 1) * 4);
    int rc = ORC_OK;
    orc_map wire_map = {0}, prod_map = {0}, out_set = {0};
    uint32_t* prod_flat = NULL; uint8_t* is_out = NULL;
    if (!c->sorted || !c->in0 || !c->in1 || !c->out || !c->op || !c->node_wire) { rc = ORC_NOMEM; goto fail; }
    memset(c->node_wire, 0xFF, (size_t)(n_nodes ? n_nodes : 1) * 4);

    /* in != out check (compiler.rs:363-383) */
    {
        uint8_t* is_in = (uint8_t*)calloc(n_nodes ? n_nodes : 1, 1);
        if (!is_in) { rc = ORC_NOMEM; goto fail; }
        for (uint32_t i = 0; i < n_in; ++i) is_in[in_nodes[i]] = 1;
        for (uint32_t i = 0; i < n_out; ++i) if (is_in[out_nodes[i]]) { free(is_in); rc = ORC_INCONSISTENCY; goto fail; }
        free(is_in);
    }

    uint32_t next_wire = 0;
    if (mode == 0) {
        if ((rc = map_init(&wire_map, 3 * n + n_in + n_out)) || (rc = map_init(&prod_map, n)) ||
            (rc = map_init(&out_set, n_out))) goto fail;
        for (uint32_t i = 0; i < n_in; ++i) map_insert(&wire_map, in_nodes[i], next_wire++);     /* :392-395 */
        for (uint64_t g = 0; g < n; ++g) map_insert(&prod_map, out[g], (uint32_t)g);             /* :403-406 */
    } else {
        prod_flat = (uint32_t*)malloc((size_t)(n_nodes ? n_nodes : 1) * 4);
        is_out = (uint8_t*)calloc(n_nodes ? n_nodes : 1, 1);
        if (!prod_flat || !is_out) { rc = ORC_NOMEM; goto fail; }
        memset(prod_flat, 0xFF, (size_t)(n_nodes ? n_nodes : 1) * 4);
        for (uint32_t i = 0; i < n_in; ++i) c->node_wire[in_nodes[i]] = next_wire++;
        for (uint64_t g = 0; g < n; ++g) prod_flat[out[g]] = (uint32_t)g;
    }

    deps_env env = {lh, rh, mode, &prod_map, prod_flat};
    rc = topo_sort(n, &env, c->sorted, cycle_at);                                                 /* :408-421 */
    if (rc) goto fail;

    if (mode == 0) {
        for (uint32_t i = 0; i < n_out; ++i) map_insert(&out_set, out_nodes[i], 1);                /* :423 */
        for (uint64_t k = 0; k < n; ++k) {                                                        /* :427-443 */
            uint64_t g = c->sorted[k];
            uint32_t nodes[3] = {lh[g], rh[g], out[g]}, tmp;
            for (int j = 0; j < 3; ++j) {
                if (map_get(&out_set, nodes[j], &tmp)) continue;
                if (map_get(&wire_map, nodes[j], &tmp)) continue;
                map_insert(&wire_map, nodes[j], next_wire++);
            }
        }
        for (uint32_t i = 0; i < n_out; ++i) map_insert(&wire_map, out_nodes[i], next_wire++);     /* :446-449 */
        for (uint64_t k = 0; k < n; ++k) {                                                        /* :453-464 */
            uint64_t g = c->sorted[k];
            uint32_t a = ORC_NONE, b = ORC_NONE, o = ORC_NONE;
            map_get(&wire_map, lh[g], &a); map_get(&wire_map, rh[g], &b); map_get(&wire_map, out[g], &o);
            c->in0[k] = a; c->in1[k] = b; c->out[k] = o; c->op[k] = op[g];
        }
        for (uint64_t i = 0; i <= wire_map.mask; ++i)
            if (wire_map.keys[i] != ORC_NONE) c->node_wire[wire_map.keys[i]] = wire_map.vals[i];
    } else {
        for (uint32_t i = 0; i < n_out; ++i) is_out[out_nodes[i]] = 1;
        for (uint64_t k = 0; k < n; ++k) {
            uint64_t g = c->sorted[k];
            uint32_t nodes[3] = {lh[g], rh[g], out[g]};
            for (int j = 0; j < 3; ++j) {
                if (is_out[nodes[j]]) continue;
                if (c->node_wire[nodes[j]] != ORC_NONE) continue;
                c->node_wire[nodes[j]] = next_wire++;
            }
        }
        for (uint32_t i = 0; i < n_out; ++i) c->node_wire[out_nodes[i]] = next_wire++;
        for (uint64_t k = 0; k < n; ++k) {
            uint64_t g = c->sorted[k];
            c->in0[k] = c->node_wire[lh[g]]; c->in1[k] = c->node_wire[rh[g]]; c->out[k] = c->node_wire[out[g]];
            c->op[k] = op[g];
        }
    }
    c->wire_count = next_wire;                                                                    /* :479 */
    map_free(&wire_map); map_free(&prod_map); map_free(&out_set); free(prod_flat); free(is_out);
    *result = c;
    return ORC_OK;
fail:
    map_free(&wire_map); map_free(&prod_map); map_free(&out_set); free(prod_flat); free(is_out);
    orc_free_circuit(c);
    return rc;
}

/* ------------------------------------------------------------------------------------------ */
/* Frozen bit-blast spec, procedural form.                                       */
/* ------------------------------------------------------------------------------------------ */
typedef struct orc_bool {
    uint64_t n_gates;
    uint64_t wire_count;
    uint32_t width, n_in, n_out;
    uint32_t *in0, *in1, *out;
    uint8_t* op;
} orc_bool;

void orc_free_bool(orc_bool* b) {
    if (!b) return;
    free(b->in0); free(b->in1); free(b->out); free(b->op); free(b);
}

typedef struct {
    uint32_t *in0, *in1, *out; uint8_t* op;   /* NULL => counting mode */
    uint64_t k;                                /* gates emitted so far  */
    uint64_t aux_base;                         /* first aux wire of this arithmetic gate */
    uint64_t aux_used;
    int overflow;
} emitter;

#define NEWW ORC_NONE /* "allocate a fresh aux wire" marker for a destination */

static uint32_t alloc_aux(emitter* e) {
    uint64_t wv = e->aux_base + e->aux_used++;
    if (wv >= 0xFFFFFFFFull) { e->overflow = 1; return 0; }
    return (uint32_t)wv;
}
static uint32_t gate(emitter* e, int op, uint32_t a, uint32_t b, uint32_t dst) {
    if (dst == NEWW) dst = alloc_aux(e);
    if (e->in0) { e->in0[e->k] = a; e->in1[e->k] = b; e->out[e->k] = dst; e->op[e->k] = (uint8_t)op; }
    e->k++;
    return dst;
}
#define XOR_(a, b, d) gate(e, B_XOR, (a), (b), (d))
#define AND_(a, b, d) gate(e, B_AND, (a), (b), (d))
#define INV_(a, d) gate(e, B_INV, (a), (a), (d))

/* D = P + Q mod 2^m; dst[i] == NEWW allocates.  Results written back into dst[]. */
static void add_bits(emitter* e, const uint32_t* P, const uint32_t* Q, uint32_t m, uint32_t* dst) {
    dst[0] = XOR_(P[0], Q[0], dst[0]);
    if (m == 1) return;
    uint32_t c = AND_(P[0], Q[0], NEWW);
    for (uint32_t i = 1; i + 1 < m; ++i) {
        uint32_t x = XOR_(P[i], c, NEWW);
        uint32_t y = XOR_(Q[i], c, NEWW);
        dst[i] = XOR_(x, Q[i], dst[i]);
        uint32_t t = AND_(x, y, NEWW);
        c = XOR_(c, t, NEWW);
    }
    uint32_t t = XOR_(P[m - 1], Q[m - 1], NEWW);
    dst[m - 1] = XOR_(t, c, dst[m - 1]);
}

/* D = P - Q mod 2^m */
static void sub_bits(emitter* e, const uint32_t* P, const uint32_t* Q, uint32_t m, uint32_t* dst) {
    dst[0] = XOR_(P[0], Q[0], dst[0]);
    if (m == 1) return;
    uint32_t br = AND_(dst[0], Q[0], NEWW);
    for (uint32_t i = 1; i + 1 < m; ++i) {
        uint32_t t = XOR_(P[i], Q[i], NEWW);
        uint32_t u = XOR_(Q[i], br, NEWW);
        dst[i] = XOR_(t, br, dst[i]);
        uint32_t v = AND_(t, u, NEWW);
        br = XOR_(br, v, NEWW);
    }
    uint32_t t = XOR_(P[m - 1], Q[m - 1], NEWW);
    dst[m - 1] = XOR_(t, br, dst[m - 1]);
}

/* D = P - Q over m bits (all D fresh aux) and the final borrow (P < Q) -> returned wire */
static uint32_t sub_borrow(emitter* e, const uint32_t* P, const uint32_t* Q, uint32_t m, uint32_t* D) {
    D[0] = XOR_(P[0], Q[0], NEWW);
    uint32_t br = AND_(D[0], Q[0], NEWW);
    for (uint32_t i = 1; i < m; ++i) {
        uint32_t t = XOR_(P[i], Q[i], NEWW);
        uint32_t u = XOR_(Q[i], br, NEWW);
        D[i] = XOR_(t, br, NEWW);
        uint32_t v = AND_(t, u, NEWW);
        br = XOR_(br, v, NEWW);
    }
    return br;
}

/* (P < Q) unsigned -> dst */
static uint32_t ult(emitter* e, const uint32_t* P, const uint32_t* Q, uint32_t m, uint32_t dst) {
    uint32_t t = XOR_(P[0], Q[0], NEWW);
    if (m == 1) return AND_(t, Q[0], dst);
    uint32_t br = AND_(t, Q[0], NEWW);
    for (uint32_t i = 1; i < m; ++i) {
        uint32_t t2 = XOR_(P[i], Q[i], NEWW);
        uint32_t u = XOR_(Q[i], br, NEWW);
        uint32_t v = AND_(t2, u, NEWW);
        br = XOR_(br, v, (i == m - 1) ? dst : NEWW);
    }
    return br;
}

/* (P == Q) -> dst */
static uint32_t eq_bits(emitter* e, const uint32_t* P, const uint32_t* Q, uint32_t m, uint32_t dst, uint32_t* scratch) {
    for (uint32_t i = 0; i < m; ++i) {
        uint32_t d = XOR_(P[i], Q[i], NEWW);
        scratch[i] = INV_(d, (m == 1) ? dst : NEWW);
    }
    uint32_t acc = scratch[0];
    for (uint32_t i = 1; i < m; ++i) acc = AND_(acc, scratch[i], (i == m - 1) ? dst : NEWW);
    return acc;
}

/* (P == 0) -> fresh aux wire */
static uint32_t is_zero(emitter* e, const uint32_t* P, uint32_t m, uint32_t* scratch) {
    for (uint32_t i = 0; i < m; ++i) scratch[i] = INV_(P[i], NEWW);
    uint32_t acc = scratch[0];
    for (uint32_t i = 1; i < m; ++i) acc = AND_(acc, scratch[i], NEWW);
    return acc;
}

static void zero_fill(emitter* e, const uint32_t* A, const uint32_t* O, uint32_t w) {
    for (uint32_t i = 1; i < w; ++i) XOR_(A[0], A[0], O[i]);
}

/* D = P * Q mod 2^m ; dst[i] == NEWW allocates; results written back into dst[] */
static void mul_bits(emitter* e, const uint32_t* P, const uint32_t* Q, uint32_t m, uint32_t* dst,
                     uint32_t* acc, uint32_t* pp, uint32_t* dtmp) {
    for (uint32_t i = 0; i < m; ++i) acc[i] = AND_(P[i], Q[0], (i == 0) ? dst[0] : NEWW);
    dst[0] = acc[0];
    for (uint32_t j = 1; j < m; ++j) {
        for (uint32_t i = j; i < m; ++i) pp[i] = AND_(P[i - j], Q[j], NEWW);
        for (uint32_t i = j; i < m; ++i) dtmp[i] = (i == j) ? dst[j] : NEWW;
        add_bits(e, acc + j, pp + j, m - j, dtmp + j);
        for (uint32_t i = j; i < m; ++i) acc[i] = dtmp[i];
        dst[j] = acc[j];
    }
}

static uint32_t ceil_log2(uint32_t w) { uint32_t k = 0; while ((1ull << k) < w) ++k; return k; }

/* one arithmetic gate -> boolean gates.  A,B,O: w boolean wire ids each. */
static void blast_gate(emitter* e, int op, uint32_t w, const uint32_t* A, const uint32_t* B, const uint32_t* O,
                       uint32_t* s0, uint32_t* s1, uint32_t* s2, uint32_t* s3, uint32_t* s4, uint32_t* s5) {
    switch (op) {
    case AXor: for (uint32_t i = 0; i < w; ++i) XOR_(A[i], B[i], O[i]); break;
    case ABitAnd: for (uint32_t i = 0; i < w; ++i) AND_(A[i], B[i], O[i]); break;
    case ABitOr:
        for (uint32_t i = 0; i < w; ++i) {
            uint32_t t = XOR_(A[i], B[i], NEWW);
            uint32_t u = AND_(A[i], B[i], NEWW);
            XOR_(t, u, O[i]);
        }
        break;
    case AAdd: for (uint32_t i = 0; i < w; ++i) s0[i] = O[i]; add_bits(e, A, B, w, s0); break;
    case ASub: for (uint32_t i = 0; i < w; ++i) s0[i] = O[i]; sub_bits(e, A, B, w, s0); break;
    case AMul: for (uint32_t i = 0; i < w; ++i) s0[i] = O[i]; mul_bits(e, A, B, w, s0, s1, s2, s3); break;
    case ALt: ult(e, A, B, w, O[0]); zero_fill(e, A, O, w); break;
    case AGt: ult(e, B, A, w, O[0]); zero_fill(e, A, O, w); break;
    case AGEq: { uint32_t r = ult(e, A, B, w, NEWW); INV_(r, O[0]); zero_fill(e, A, O, w); } break;
    case ALEq: { uint32_t r = ult(e, B, A, w, NEWW); INV_(r, O[0]); zero_fill(e, A, O, w); } break;
    case AEq: eq_bits(e, A, B, w, O[0], s0); zero_fill(e, A, O, w); break;
    case ANeq: { uint32_t r = eq_bits(e, A, B, w, NEWW, s0); INV_(r, O[0]); zero_fill(e, A, O, w); } break;
    case ABoolOr: {
        uint32_t za = is_zero(e, A, w, s0), zb = is_zero(e, B, w, s0);
        uint32_t t = AND_(za, zb, NEWW);
        INV_(t, O[0]); zero_fill(e, A, O, w);
    } break;
    case ABoolAnd: {
        uint32_t za = is_zero(e, A, w, s0), zb = is_zero(e, B, w, s0);
        uint32_t na = INV_(za, NEWW), nb = INV_(zb, NEWW);
        AND_(na, nb, O[0]); zero_fill(e, A, O, w);
    } break;
    case AShiftL:
    case AShiftR: {
        uint32_t K = ceil_log2(w);
        uint32_t* cur = s0; uint32_t* nxt = s1;
        for (uint32_t i = 0; i < w; ++i) cur[i] = A[i];
        for (uint32_t k = 0; k < K; ++k) {
            uint32_t sh = 1u << k, s = B[k];
            for (uint32_t i = 0; i < w; ++i) {
                int in_range = (op == AShiftL) ? (i >= sh) : ((uint64_t)i + sh < w);
                if (in_range) {
                    uint32_t src = (op == AShiftL) ? cur[i - sh] : cur[i + sh];
                    uint32_t t = XOR_(cur[i], src, NEWW);
                    uint32_t mm = AND_(s, t, NEWW);
                    nxt[i] = XOR_(cur[i], mm, NEWW);
                } else {
                    uint32_t mm = AND_(s, cur[i], NEWW);
                    nxt[i] = XOR_(cur[i], mm, NEWW);
                }
            }
            uint32_t* tswap = cur; cur = nxt; nxt = tswap;
        }
        for (uint32_t j = K; j < w; ++j) s2[j] = INV_(B[j], NEWW);
        uint32_t acc = s2[K];
        for (uint32_t j = K + 1; j < w; ++j) acc = AND_(acc, s2[j], NEWW);
        for (uint32_t i = 0; i < w; ++i) AND_(cur[i], acc, O[i]);
    } break;
    case ADiv:
    case AIntDiv:
    case AMod: {
        int want_q = (op != AMod);
        uint32_t z = XOR_(A[0], A[0], NEWW);
        uint32_t* R = s0;   /* w   */
        uint32_t* Rp = s1;  /* w+1 */
        uint32_t* Qx = s2;  /* w+1 */
        uint32_t* D = s3;   /* w+1 */
        for (uint32_t j = 0; j < w; ++j) { R[j] = z; Qx[j] = B[j]; }
        Qx[w] = z;
        for (uint32_t it = 0; it < w; ++it) {
            uint32_t i = w - 1 - it;
            Rp[0] = A[i];
            for (uint32_t j = 1; j <= w; ++j) Rp[j] = R[j - 1];
            uint32_t borrow = sub_borrow(e, Rp, Qx, w + 1, D);
            if (want_q) INV_(borrow, O[i]);
            int last = (it == w - 1);
            if (last && want_q) break;
            for (uint32_t j = 0; j < w; ++j) {
                uint32_t t = XOR_(D[j], Rp[j], NEWW);
                uint32_t mm = AND_(borrow, t, NEWW);
                R[j] = XOR_(D[j], mm, (last && !want_q) ? O[j] : NEWW);
            }
        }
    } break;
    case APow: {
        uint32_t z = XOR_(A[0], A[0], NEWW);
        uint32_t one = INV_(z, NEWW);
        uint32_t* res = s0; uint32_t* base = s1; uint32_t* sq = s2; uint32_t* prod = s3;
        {
            uint32_t t = XOR_(A[0], one, NEWW);
            uint32_t mm = AND_(B[0], t, NEWW);
            res[0] = XOR_(one, mm, (w == 1) ? O[0] : NEWW);
            for (uint32_t j = 1; j < w; ++j) res[j] = AND_(B[0], A[j], NEWW);
        }
        for (uint32_t j = 0; j < w; ++j) base[j] = A[j];
        /* scratch for mul_bits lives past s3: caller provides s4 (acc,pp) and s5 (dtmp) sized 2w / w */
        for (uint32_t i = 1; i < w; ++i) {
            for (uint32_t j = 0; j < w; ++j) sq[j] = NEWW;
            mul_bits(e, base, base, w, sq, s4, s4 + w, s5);
            for (uint32_t j = 0; j < w; ++j) base[j] = sq[j];
            for (uint32_t j = 0; j < w; ++j) prod[j] = NEWW;
            mul_bits(e, res, base, w, prod, s4, s4 + w, s5);
            for (uint32_t j = 0; j < w; ++j) {
                uint32_t t = XOR_(prod[j], res[j], NEWW);
                uint32_t mm = AND_(B[i], t, NEWW);
                res[j] = XOR_(res[j], mm, (i == w - 1) ? O[j] : NEWW);
            }
        }
    } break;
    default: break;
    }
}

/* template size T(op,w) and aux wire count AUX(op,w) of the spec */
int orc_template_size(int op, uint32_t w, uint64_t* n_gates, uint64_t* n_aux) {
    if (op < 0 || op >= A_NUM_OPS || w == 0) return ORC_ARG;
    uint32_t m = w + 2;
    uint32_t* buf = (uint32_t*)malloc((size_t)m * 4 * 10);
    if (!buf) return ORC_NOMEM;
    uint32_t *A = buf, *B = buf + m, *O = buf + 2 * m;
    for (uint32_t i = 0; i < w; ++i) { A[i] = i; B[i] = w + i; O[i] = 2 * w + i; }
    emitter e = {NULL, NULL, NULL, NULL, 0, 3ull * w, 0, 0};
    blast_gate(&e, op, w, A, B, O, buf + 3 * m, buf + 4 * m, buf + 5 * m, buf + 6 * m, buf + 7 * m, buf + 9 * m);
    free(buf);
    *n_gates = e.k; *n_aux = e.aux_used;
    return ORC_OK;
}

/*
 * boolify(circuit, w) under the frozen spec.  Wire layout:
 *   M = wire_count - n_out.   bit b of arithmetic wire W:
 *     W <  M : W*w + b
 *     W >= M : M*w + aux_total + (W-M)*w + b            (outputs last)
 *   aux wires of the gate at sorted position p: M*w + auxoff[p] + k
 */
/* gates of the arithmetic gates at sorted positions [first, first+count) only, with the GLOBAL wire/aux
 * numbering of the whole circuit (for slice-wise parity checks at sizes where the full boolean circuit would
 * not fit in host memory).  *gate_first = global index of the first boolean gate of the slice. */
int orc_boolify_range(const orc_circuit* c, uint32_t w, uint64_t first, uint64_t count, orc_bool** result,
                      uint64_t* gate_first) {
    *result = NULL;
    if (w == 0 || w > 64 || first > c->n || count > c->n - first) return ORC_ARG;
    uint64_t tg[A_NUM_OPS], ta[A_NUM_OPS];
    for (int op = 0; op < A_NUM_OPS; ++op) { int rc = orc_template_size(op, w, &tg[op], &ta[op]); if (rc) return rc; }
    uint64_t G = 0, AUX = 0, g_first = 0, a_first = 0, g_slice = 0;
    for (uint64_t p = 0; p < c->n; ++p) {
        if (p == first) { g_first = G; a_first = AUX; }
        if (p >= first && p < first + count) g_slice += tg[c->op[p]];
        G += tg[c->op[p]]; AUX += ta[c->op[p]];
    }
    if (first == c->n) { g_first = G; a_first = AUX; }
    uint64_t M = (uint64_t)c->wire_count - c->n_out;
    uint64_t total_wires = (uint64_t)c->wire_count * w + AUX;
    if (total_wires >= 0xFFFFFFFFull) return ORC_OVERFLOW;
    orc_bool* b = (orc_bool*)calloc(1, sizeof(orc_bool));
    if (!b) return ORC_NOMEM;
    uint64_t ga = g_slice ? g_slice : 1;
    b->in0 = (uint32_t*)malloc(ga * 4); b->in1 = (uint32_t*)malloc(ga * 4); b->out = (uint32_t*)malloc(ga * 4);
    b->op = (uint8_t*)malloc(ga);
    uint32_t m = w + 2;
    uint32_t* buf = (uint32_t*)malloc((size_t)m * 4 * 10);
    if (!b->in0 || !b->in1 || !b->out || !b->op || !buf) { free(buf); orc_free_bool(b); return ORC_NOMEM; }
    b->n_gates = g_slice; b->wire_count = total_wires; b->width = w; b->n_in = c->n_in; b->n_out = c->n_out;
    uint32_t *A = buf, *B = buf + m, *O = buf + 2 * m;
    emitter e = {b->in0, b->in1, b->out, b->op, 0, M * w, 0, 0};
    uint64_t out_base = M * w + AUX;
    uint64_t auxoff = a_first;
    for (uint64_t p = first; p < first + count; ++p) {
        uint32_t wa = c->in0[p], wb = c->in1[p], wo = c->out[p];
        for (uint32_t i = 0; i < w; ++i) {
            A[i] = (uint32_t)(wa < M ? (uint64_t)wa * w + i : out_base + ((uint64_t)wa - M) * w + i);
            B[i] = (uint32_t)(wb < M ? (uint64_t)wb * w + i : out_base + ((uint64_t)wb - M) * w + i);
            O[i] = (uint32_t)(wo < M ? (uint64_t)wo * w + i : out_base + ((uint64_t)wo - M) * w + i);
        }
        e.aux_base = M * w + auxoff; e.aux_used = 0;
        blast_gate(&e, c->op[p], w, A, B, O, buf + 3 * m, buf + 4 * m, buf + 5 * m, buf + 6 * m, buf + 7 * m, buf + 9 * m);
        auxoff += ta[c->op[p]];
    }
    free(buf);
    if (e.overflow || e.k != g_slice) { orc_free_bool(b); return ORC_OVERFLOW; }
    *result = b;
    if (gate_first) *gate_first = g_first;
    return ORC_OK;
}

int orc_boolify(const orc_circuit* c, uint32_t w, orc_bool** result) {
    return orc_boolify_range(c, w, 0, c->n, result, NULL);
}

/* boolean wire of (arithmetic wire W, bit b) under the layout above */
uint64_t orc_bool_wire(const orc_circuit* c, const orc_bool* b, uint32_t W, uint32_t bit) {
    uint64_t M = (uint64_t)c->wire_count - c->n_out, w = b->width;
    uint64_t aux = b->wire_count - (uint64_t)c->wire_count * w;
    return W < M ? (uint64_t)W * w + bit : M * w + aux + ((uint64_t)W - M) * w + bit;
}

/* ------------------------------------------------------------------------------------------ */
/* Evaluators.  Semantics = tests/integration.rs:94-115 taken mod 2^w (unsigned); where the     */
/* reference would panic (overflow, /0, shift >= w) the value is DEFINED BY OUR SPEC (unpinned):  */
/* wrap-around; x/0 = 2^w-1, x%0 = x; shift by >= w gives 0.                                     */
/* ------------------------------------------------------------------------------------------ */
static uint64_t maskw(uint32_t w) { return w >= 64 ? ~0ull : ((1ull << w) - 1); }

uint64_t orc_eval_op(int op, uint64_t a, uint64_t b, uint32_t w) {
    uint64_t mk = maskw(w), r = 0;
    a &= mk; b &= mk;
    switch (op) {
    case AAdd: r = a + b; break;
    case ASub: r = a - b; break;
    case AMul: r = a * b; break;
    case ADiv: case AIntDiv: r = b ? a / b : mk; break;
    case AMod: r = b ? a % b : a; break;
    case AEq: r = a == b; break;
    case ANeq: r = a != b; break;
    case AGEq: r = a >= b; break;
    case AGt: r = a > b; break;
    case ALEq: r = a <= b; break;
    case ALt: r = a < b; break;
    case AXor: r = a ^ b; break;
    case ABitOr: r = a | b; break;
    case ABitAnd: r = a & b; break;
    case ABoolOr: r = (a != 0) || (b != 0); break;
    case ABoolAnd: r = (a != 0) && (b != 0); break;
    case AShiftL: r = b >= w ? 0 : a << b; break;
    case AShiftR: r = b >= w ? 0 : a >> b; break;
    case APow: { uint64_t base = a, acc = 1, ex = b; while (ex) { if (ex & 1) acc *= base; base *= base; ex >>= 1; } r = acc; } break;
    default: r = 0;
    }
    return r & mk;
}

/* arithmetic evaluation of T test vectors: wires[W*T + t]; inputs/constants pre-filled by the caller */
int orc_eval_arith(uint64_t n, const uint32_t* in0, const uint32_t* in1, const uint32_t* out, const uint8_t* op,
                   uint32_t w, uint64_t T, uint64_t* wires) {
    for (uint64_t p = 0; p < n; ++p)
        for (uint64_t t = 0; t < T; ++t)
            wires[(uint64_t)out[p] * T + t] = orc_eval_op(op[p], wires[(uint64_t)in0[p] * T + t], wires[(uint64_t)in1[p] * T + t], w);
    return ORC_OK;
}

/* bit-parallel boolean evaluation: wires[v] carries 64 test vectors */
int orc_eval_bool(uint64_t n, const uint32_t* in0, const uint32_t* in1, const uint32_t* out, const uint8_t* op,
                  uint64_t* wires) {
    for (uint64_t k = 0; k < n; ++k) {
        uint64_t a = wires[in0[k]], b = wires[in1[k]];
        wires[out[k]] = op[k] == B_XOR ? (a ^ b) : op[k] == B_AND ? (a & b) : ~a;
    }
    return ORC_OK;
}

/* order-sensitive checksum of a u32 / u8 stream (FNV-1a 64 over little-endian bytes); used for
 * "checksum of checksums" parity at sizes too big to copy around */
uint64_t orc_fnv1a(const void* data, uint64_t nbytes, uint64_t seed) {
    const uint8_t* p = (const uint8_t*)data;
    uint64_t h = seed ? seed : 0xcbf29ce484222325ULL;
    for (uint64_t i = 0; i < nbytes; ++i) { h ^= p[i]; h *= 0x100000001b3ULL; }
    return h;
}

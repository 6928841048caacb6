// c2a_host.hpp — C++ host mirror of the reference's flat-gate-graph API, above the C ABI of include/c2a.h.
//
// The reference is Rust; its toolchain is not in this image, so the host side is C++ (header-only).  Names,
// argument meaning and error texts follow the reference so code written against it reads the same:
//
//   reference (Rust)                                         here (namespace c2a::host)
//   ------------------------------------------------------   ------------------------------------------------
//   AGateType + strum Display/EnumString  a_gate_type.rs:6-28   AGateType, to_string(), parse_gate_type()
//   Compiler::{new, add_inputs, add_outputs, add_signal,        Compiler::{Compiler, add_inputs, add_outputs,
//     get_signals, add_gate, add_connection}                      add_signal, get_signals, add_gate, add_connection}
//                                      compiler.rs:117-278
//   Compiler::build_circuit() -> Result<BristolCircuit, CircuitError>  compiler.rs:321-494
//                                                             Compiler::build_circuit() -> BristolCircuit (throws CircuitError)
//   CircuitError::{CyclicDependency, Inconsistency, ...}  compiler.rs:550-575   CircuitError (kind + same Display text)
//   boolify(&circuit, width)                  main.rs:30-32   boolify(compiler, circuit, width)   (frozen spec)
//   circuit.write_bristol(&mut File)          main.rs:34-35   BristolCircuit::write_bristol(std::ostream&)
//   to_string_pretty(&circuit.info)           main.rs:43-44   BristolCircuit::info_json()
//
// What differs is where the work happens: build_circuit keeps the name maps and the two name-level checks on the host
// (compiler.rs:323-383) and sends Compiler.gates as SoA through c2a_load_gates / c2a_topo_sort / c2a_assign_wires /
// c2a_emit_gates (HIP kernels on the MI355X).  The gate-graph builder keeps a signal->node index and a forwarding
// table instead of the reference's per-call scans over all nodes and gates (compiler.rs:185-195, :219-226, :260-270).
// Canonical IO order: ascending signal id (the reference iterates std HashMaps there).
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <optional>
#include <ostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/c2a.h"

namespace c2a {
namespace host {

// ---- a_gate_type.rs:6-28 -------------------------------------------------------------------------------------------
enum class AGateType : uint8_t {
    AAdd, ADiv, AEq, AGEq, AGt, ALEq, ALt, AMul, ANeq, ASub, AXor, APow, AIntDiv, AMod, AShiftL, AShiftR, ABoolOr,
    ABoolAnd, ABitOr, ABitAnd
};
inline const char* to_string(AGateType t) {
    static const char* const names[] = {"AAdd", "ADiv", "AEq", "AGEq", "AGt", "ALEq", "ALt", "AMul", "ANeq", "ASub", "AXor",
                                        "APow", "AIntDiv", "AMod", "AShiftL", "AShiftR", "ABoolOr", "ABoolAnd", "ABitOr",
                                        "ABitAnd"};
    return names[static_cast<uint8_t>(t)];
}
inline std::optional<AGateType> parse_gate_type(const std::string& s) {
    for (uint8_t i = 0; i < C2A_NUM_GATE_TYPES; ++i)
        if (s == to_string(static_cast<AGateType>(i))) return static_cast<AGateType>(i);
    return std::nullopt;
}

// ---- compiler.rs:550-575 -------------------------------------------------------------------------------------------
class CircuitError : public std::runtime_error {
public:
    enum Kind { CannotMergeConstantNodes, CannotMergeOutputNodes, SignalAlreadyDeclared, CyclicDependency, Inconsistency,
                Backend };
    CircuitError(Kind k, const std::string& display, std::string msg = {})
        : std::runtime_error(display), kind(k), message(std::move(msg)) {}
    Kind kind;
    std::string message;     // the `message` field of CyclicDependency / Inconsistency
    static CircuitError cyclic(uint64_t i) {
        const std::string m = "detected at i=" + std::to_string(i);                 // topological_sort.rs:36
        return CircuitError(CyclicDependency, "Cyclic dependency: " + m, m);
    }
    static CircuitError inconsistency(const std::string& m) { return CircuitError(Inconsistency, "Inconsistency: " + m, m); }
};

// ---- bristol-circuit types as used at compiler.rs:456-493 ------------------------------------------------------------
struct ConstantInfo {
    std::string value;
    size_t wire_index;
    bool operator==(const ConstantInfo& o) const { return value == o.value && wire_index == o.wire_index; }
};
struct CircuitInfo {
    std::map<std::string, size_t> input_name_to_wire_index;        // sorted maps: deterministic JSON
    std::map<std::string, ConstantInfo> constants;
    std::map<std::string, size_t> output_name_to_wire_index;
};
struct Gate {
    std::vector<size_t> inputs, outputs;
    std::string op;
};
struct BristolCircuit {
    size_t wire_count = 0;
    CircuitInfo info;
    // gates as SoA (what the GPU hands back); gate(k) materialises the reference's Gate struct
    std::vector<uint32_t> in0, in1, out;
    std::vector<uint8_t> op;
    bool boolean = false;                                          // op names: AGateType vs XOR/AND/INV
    std::optional<std::pair<std::vector<size_t>, std::vector<size_t>>> io_widths;   // None for arithmetic (:492)
    std::vector<uint32_t> sorted_gate_ids;

    size_t gates_on_device = 0;     // > 0: the SoA was left in HBM (fetch = false); this is the gate count
    size_t n_gates() const { return gates_on_device ? gates_on_device : op.size(); }
    std::string op_name(size_t k) const {
        static const char* const b[] = {"XOR", "AND", "INV"};
        return boolean ? b[op[k]] : to_string(static_cast<AGateType>(op[k]));
    }
    Gate gate(size_t k) const {
        Gate g;
        g.inputs = (boolean && op[k] == C2A_INV) ? std::vector<size_t>{in0[k]} : std::vector<size_t>{in0[k], in1[k]};
        g.outputs = {out[k]};
        g.op = op_name(k);
        return g;
    }
    // circuit.txt (main.rs:34-35).  Bristol fashion; the crate's exact text is not pinned by any reference test
    // (SURVEY Appendix C.2): '{ngates} {nwires}' / '{n_in} {widths..}' / '{n_out} {widths..}' / blank / one gate per line.
    void write_bristol(std::ostream& w) const {
        const size_t n_in = info.input_name_to_wire_index.size(), n_out = info.output_name_to_wire_index.size();
        std::vector<size_t> iw(n_in, 1), ow(n_out, 1);
        if (io_widths) { iw = io_widths->first; ow = io_widths->second; }
        w << n_gates() << ' ' << wire_count << '\n' << iw.size();
        for (size_t x : iw) w << ' ' << x;
        w << '\n' << ow.size();
        for (size_t x : ow) w << ' ' << x;
        w << "\n\n";
        for (size_t k = 0; k < n_gates(); ++k) {
            if (boolean && op[k] == C2A_INV) w << "1 1 " << in0[k] << ' ' << out[k] << ' ' << op_name(k) << '\n';
            else w << "2 1 " << in0[k] << ' ' << in1[k] << ' ' << out[k] << ' ' << op_name(k) << '\n';
        }
    }
    // the same file with the gate lines printed on the GPU (c2a_format_bristol) and streamed chunk by chunk: the boolean
    // circuit of the 10 M-gate config is ~27 GB of text.  `ctx` must still hold the circuit this object describes.
    size_t write_bristol_gpu(std::ostream& w, c2a_ctx* ctx, size_t chunk_gates = size_t(1) << 24) const {
        const size_t n_in = info.input_name_to_wire_index.size(), n_out = info.output_name_to_wire_index.size();
        std::vector<size_t> iw(n_in, 1), ow(n_out, 1);
        if (io_widths) { iw = io_widths->first; ow = io_widths->second; }
        std::ostringstream h;
        h << n_gates() << ' ' << wire_count << '\n' << iw.size();
        for (size_t x : iw) h << ' ' << x;
        h << '\n' << ow.size();
        for (size_t x : ow) h << ' ' << x;
        h << "\n\n";
        w << h.str();
        size_t total = h.str().size();
        std::vector<char> buf;
        for (size_t first = 0; first < n_gates(); first += chunk_gates) {
            const size_t cnt = std::min(chunk_gates, n_gates() - first);
            uint64_t need = 0, got = 0;
            int rc = c2a_format_bristol(ctx, boolean ? 1 : 0, first, cnt, nullptr, 0, &need);
            if (rc != C2A_OK) throw std::runtime_error(std::string("c2a_format_bristol: ") + c2a_last_error(ctx));
            buf.resize(need);
            rc = c2a_format_bristol(ctx, boolean ? 1 : 0, first, cnt, buf.data(), need, &got);
            if (rc != C2A_OK) throw std::runtime_error(std::string("c2a_format_bristol: ") + c2a_last_error(ctx));
            w.write(buf.data(), (std::streamsize)got);
            total += got;
        }
        return total;
    }
    // circuit_info.json (main.rs:43-44), serde_json::to_string_pretty layout
    std::string info_json() const {
        auto esc = [](const std::string& s) {
            std::string o = "\"";
            for (char ch : s) { if (ch == '"' || ch == '\\') o += '\\'; o += ch; }
            return o + "\"";
        };
        std::ostringstream j;
        j << "{\n  \"input_name_to_wire_index\": {";
        bool first = true;
        for (auto& kv : info.input_name_to_wire_index) { j << (first ? "\n" : ",\n") << "    " << esc(kv.first) << ": " << kv.second; first = false; }
        j << (first ? "" : "\n  ") << "},\n  \"constants\": {";
        first = true;
        for (auto& kv : info.constants) {
            j << (first ? "\n" : ",\n") << "    " << esc(kv.first) << ": {\n      \"value\": " << esc(kv.second.value)
              << ",\n      \"wire_index\": " << kv.second.wire_index << "\n    }";
            first = false;
        }
        j << (first ? "" : "\n  ") << "},\n  \"output_name_to_wire_index\": {";
        first = true;
        for (auto& kv : info.output_name_to_wire_index) { j << (first ? "\n" : ",\n") << "    " << esc(kv.first) << ": " << kv.second; first = false; }
        j << (first ? "" : "\n  ") << "}\n}";
        return j.str();
    }
};

// ---- compiler.rs:16-36, 85-115 -----------------------------------------------------------------------------------------
struct Signal { std::string name; std::optional<uint32_t> value; };
struct Node { bool is_const = false, is_out = false; std::vector<uint32_t> signals; };
struct ArithmeticGate { AGateType op; uint32_t lh_in, rh_in, out; };

class Backend {          // RAII owner of one c2a context
public:
    explicit Backend(int device = 0) {
        const int rc = c2a_create(1, &device, &ctx_);
        if (rc != C2A_OK) throw CircuitError(CircuitError::Backend, "c2a_create failed with status " + std::to_string(rc) +
                                                                      ": no usable HIP device (no CPU fallback)");
    }
    ~Backend() { c2a_destroy(ctx_); }
    Backend(const Backend&) = delete;
    Backend& operator=(const Backend&) = delete;
    c2a_ctx* get() const { return ctx_; }
    void check(int rc, uint64_t cycle_at = 0) const {
        if (rc == C2A_OK) return;
        if (rc == C2A_ERR_CYCLIC) throw CircuitError::cyclic(cycle_at);
        if (rc == C2A_ERR_INCONSISTENCY) throw CircuitError::inconsistency("a node is used for both input and output");
        throw CircuitError(CircuitError::Backend, std::string("c2a status ") + std::to_string(rc) + ": " + c2a_last_error(ctx_));
    }
private:
    c2a_ctx* ctx_ = nullptr;
};

class Compiler {
public:
    explicit Compiler(Backend& be) : be_(be) {}

    void add_inputs(const std::map<uint32_t, std::string>& m) { for (auto& kv : m) inputs_[kv.first] = kv.second; }     // :130-132
    void add_outputs(const std::map<uint32_t, std::string>& m) { for (auto& kv : m) outputs_[kv.first] = kv.second; }   // :134-136

    void add_signal(uint32_t id, const std::string& name, std::optional<uint32_t> value) {                              // :139-161
        if (signals_.count(id)) throw CircuitError(CircuitError::SignalAlreadyDeclared, "Signal already declared");
        signals_[id] = Signal{name, value};
        const uint32_t nid = get_node_id();
        Node nd; nd.is_const = value.has_value(); nd.signals = {id};
        nodes_[nid] = std::move(nd);
        sig_node_[id] = nid;
    }
    std::map<uint32_t, std::string> get_signals(const std::string& filter) const {                                      // :163-171
        std::map<uint32_t, std::string> r;
        for (auto& kv : signals_) if (kv.second.name.compare(0, filter.size(), filter) == 0) r[kv.first] = kv.second.name;
        return r;
    }
    void add_gate(AGateType t, uint32_t lhs, uint32_t rhs, uint32_t out_sig) {                                          // :174-209
        const uint32_t n0 = node_of(lhs), n1 = node_of(rhs), n2 = node_of(out_sig);
        nodes_.at(n2).is_out = true;                        // .at() throws where the reference's unwrap() panics
        gates_.push_back(ArithmeticGate{t, n0, n1, n2});
    }
    void add_connection(uint32_t a, uint32_t b) {                                                                       // :213-278
        const uint32_t na = node_of(a), nb = node_of(b);
        if (na == nb) return;
        // an unknown signal id leaves the reference's scan on its placeholder `(0, &Node::new())` (:215-228): the merge
        // then goes through with an empty node 0 (node ids start at 1, so 0 is never a real node)
        static const Node kEmpty{};
        const Node& A = na ? nodes_.at(na) : kEmpty;
        const Node& B = nb ? nodes_.at(nb) : kEmpty;
        if (A.is_out && B.is_out) throw CircuitError(CircuitError::CannotMergeOutputNodes, "Cannot merge output nodes");
        if (A.is_const && B.is_const) throw CircuitError(CircuitError::CannotMergeConstantNodes, "Cannot merge constant nodes");
        Node merged;
        merged.is_out = A.is_out || B.is_out;
        merged.is_const = A.is_const || B.is_const;
        merged.signals = A.signals;
        merged.signals.insert(merged.signals.end(), B.signals.begin(), B.signals.end());
        const uint32_t mid = get_node_id();
        for (uint32_t old : {na, nb}) {
            if (old) { fwd_[old] = mid; nodes_.erase(old); }   // the reference rewrites every gate here (:260-270); resolved lazily
            else                                               // ... except for the placeholder id 0: rewritten now, a later gate keeps its 0
                for (auto& g : gates_) { if (g.lh_in == 0) g.lh_in = mid; if (g.rh_in == 0) g.rh_in = mid; if (g.out == 0) g.out = mid; }
        }
        for (uint32_t s : merged.signals) sig_node_[s] = mid;
        nodes_[mid] = std::move(merged);
    }
    std::vector<ArithmeticGate> gates() const {            // Vec<ArithmeticGate> with merges applied (:113)
        std::vector<ArithmeticGate> r;
        for (auto g : gates_) { g.lh_in = resolve(g.lh_in); g.rh_in = resolve(g.rh_in); g.out = resolve(g.out); r.push_back(g); }
        return r;
    }
    uint32_t node_count() const { return node_count_; }

    // report.json (main.rs:22, :46-47): compiler.rs:287-319 + :502-531 — nodes split into inputs (not the output of any gate)
    // and outputs (gate outputs that no gate reads), each sorted by node id; per node the names of its signals (those
    // containing "random_" are dropped, :519) and the value of its last valued signal.  serde_json::to_string_pretty layout.
    std::string report_json(const std::string& value_type = "sint") const {
        std::vector<uint32_t> in_nodes, out_nodes;
        std::unordered_map<uint32_t, bool> read;
        for (const auto& g : gates_) { read[resolve(g.lh_in)] = true; read[resolve(g.rh_in)] = true; }    // (:300-304 scans per node)
        for (const auto& kv : nodes_) {
            if (!kv.second.is_out) in_nodes.push_back(kv.first);
            else if (!read.count(kv.first)) out_nodes.push_back(kv.first);
        }
        std::sort(in_nodes.begin(), in_nodes.end());
        std::sort(out_nodes.begin(), out_nodes.end());
        auto esc = [](const std::string& s) {
            std::string o = "\"";
            for (char ch : s) { if (ch == '"' || ch == '\\') o += '\\'; o += ch; }
            return o + "\"";
        };
        auto reports = [&](const std::vector<uint32_t>& ids) {
            std::ostringstream j;
            if (ids.empty()) return std::string("[]");
            j << "[";
            bool first = true;
            for (uint32_t id : ids) {
                std::vector<std::string> names;
                std::optional<uint32_t> value;
                for (uint32_t sid : nodes_.at(id).signals) {
                    const Signal& sg = signals_.at(sid);
                    if (sg.name.find("random_") == std::string::npos) names.push_back(sg.name);
                    if (sg.value) value = sg.value;
                }
                j << (first ? "\n" : ",\n") << "    {\n      \"id\": " << id << ",\n      \"names\": ";
                if (names.empty()) j << "[]";
                else {
                    j << "[";
                    for (size_t k = 0; k < names.size(); ++k) j << (k ? ",\n" : "\n") << "        " << esc(names[k]);
                    j << "\n      ]";
                }
                j << ",\n      \"value\": ";
                if (value) j << *value; else j << "null";
                j << "\n    }";
                first = false;
            }
            j << "\n  ]";
            return j.str();
        };
        return "{\n  \"inputs\": " + reports(in_nodes) + ",\n  \"outputs\": " + reports(out_nodes) + ",\n  \"value_type\": \"" + value_type + "\"\n}";
    }

    // compiler.rs:321-494
    BristolCircuit build_circuit() const {
        // -- name maps + the two name-level checks (:323-383), canonical order = ascending signal id
        std::vector<std::pair<std::string, uint32_t>> inputs, outputs;
        std::map<std::string, std::pair<uint32_t, std::string>> constants;
        std::map<std::string, int> seen_in, seen_out;
        for (auto& kv : sig_node_sorted()) {
            const uint32_t sid = kv.first, nid = kv.second;
            auto it = inputs_.find(sid);
            if (it != inputs_.end()) {
                if (seen_in[it->second]++) throw CircuitError::inconsistency("Duplicate input " + it->second);        // :335-339
                inputs.emplace_back(it->second, nid);
            }
            auto ot = outputs_.find(sid);
            if (ot != outputs_.end()) {
                if (seen_out[ot->second]++) throw CircuitError::inconsistency("Duplicate output " + ot->second);      // :345-349
                outputs.emplace_back(ot->second, nid);
            }
            const Signal& sg = signals_.at(sid);
            if (sg.value) constants[sg.name + "_" + std::to_string(sid)] = {nid, std::to_string(*sg.value)};          // :354-359
        }
        std::unordered_map<uint32_t, std::string> node_to_input;
        for (auto& p : inputs) node_to_input[p.second] = p.first;
        for (auto& p : outputs) {                                                                                     // :363-383
            auto it = node_to_input.find(p.second);
            if (it != node_to_input.end())
                throw CircuitError::inconsistency("Node " + std::to_string(p.second) + " used for both input " + it->second +
                                                  " and output " + p.first);
        }
        // -- Compiler.gates as SoA through the C ABI: :385-464 run on the GPU
        const size_t n = gates_.size();
        std::vector<uint32_t> lh(n), rh(n), out(n), in_nodes, out_nodes;
        std::vector<uint8_t> op(n);
        for (size_t g = 0; g < n; ++g) {
            lh[g] = resolve(gates_[g].lh_in); rh[g] = resolve(gates_[g].rh_in); out[g] = resolve(gates_[g].out);
            op[g] = static_cast<uint8_t>(gates_[g].op);
        }
        for (auto& p : inputs) in_nodes.push_back(p.second);
        for (auto& p : outputs) out_nodes.push_back(p.second);
        const uint32_t n_nodes = node_count_ + 1;
        c2a_ctx* c = be_.get();
        be_.check(c2a_load_gates(c, n, lh.data(), rh.data(), out.data(), op.data(), n_nodes, (uint32_t)in_nodes.size(),
                                 in_nodes.data(), (uint32_t)out_nodes.size(), out_nodes.data()));
        BristolCircuit bc;
        bc.sorted_gate_ids.resize(n);
        uint64_t cycle_at = 0;
        be_.check(c2a_topo_sort(c, bc.sorted_gate_ids.data(), &cycle_at), cycle_at);                                  // :408-421
        std::vector<uint32_t> node_to_wire(n_nodes);
        uint32_t wire_count = 0;
        be_.check(c2a_assign_wires(c, node_to_wire.data(), &wire_count));                                             // :388-449
        bc.in0.resize(n); bc.in1.resize(n); bc.out.resize(n); bc.op.resize(n);
        be_.check(c2a_emit_gates(c, bc.in0.data(), bc.in1.data(), bc.out.data(), bc.op.data()));                      // :451-464
        bc.wire_count = wire_count;
        for (auto& kv : constants) {                                                                                  // :466-476
            const uint32_t w = node_to_wire[kv.second.first];
            if (w == C2A_NO_WIRE) throw std::out_of_range("constant node has no wire (the reference panics here)");
            bc.info.constants[kv.first] = ConstantInfo{kv.second.second, w};
        }
        for (auto& p : inputs) bc.info.input_name_to_wire_index[p.first] = node_to_wire[p.second];
        for (auto& p : outputs) bc.info.output_name_to_wire_index[p.first] = node_to_wire[p.second];
        return bc;
    }

    Backend& backend() const { return be_; }

private:
    uint32_t get_node_id() { return ++node_count_; }                                                                  // :497-500
    uint32_t node_of(uint32_t sig) const { auto it = sig_node_.find(sig); return it == sig_node_.end() ? 0u : it->second; }
    uint32_t resolve(uint32_t nid) const {
        auto it = fwd_.find(nid);
        while (it != fwd_.end()) { nid = it->second; it = fwd_.find(nid); }
        return nid;
    }
    std::map<uint32_t, uint32_t> sig_node_sorted() const { return std::map<uint32_t, uint32_t>(sig_node_.begin(), sig_node_.end()); }

    Backend& be_;
    uint32_t node_count_ = 0;
    std::map<uint32_t, std::string> inputs_, outputs_;
    std::map<uint32_t, Signal> signals_;
    std::unordered_map<uint32_t, Node> nodes_;
    std::unordered_map<uint32_t, uint32_t> sig_node_, fwd_;
    std::vector<ArithmeticGate> gates_;
};

// boolify(&circuit, width) (main.rs:30-32) of the circuit last built on `compiler`'s backend.  With fetch=false the
// boolean SoA stays in HBM (only counts and the rebased info are returned).
inline BristolCircuit boolify(const Compiler& compiler, const BristolCircuit& circuit, uint32_t width, bool fetch = true) {
    Backend& be = compiler.backend();
    c2a_bool_info bi;
    be.check(c2a_boolify(be.get(), width, &bi));
    auto wire = [&](size_t W) -> size_t {
        return W < bi.m_wires ? W * width : (size_t)bi.m_wires * width + bi.aux_total + (W - bi.m_wires) * width;
    };
    BristolCircuit b;
    b.boolean = true;
    b.wire_count = bi.wire_count;
    for (auto& kv : circuit.info.input_name_to_wire_index) b.info.input_name_to_wire_index[kv.first] = wire(kv.second);
    for (auto& kv : circuit.info.output_name_to_wire_index) b.info.output_name_to_wire_index[kv.first] = wire(kv.second);
    for (auto& kv : circuit.info.constants) b.info.constants[kv.first] = ConstantInfo{kv.second.value, wire(kv.second.wire_index)};
    b.io_widths = std::make_pair(std::vector<size_t>(bi.n_in, width), std::vector<size_t>(bi.n_out, width));
    if (fetch) {
        b.in0.resize(bi.n_gates); b.in1.resize(bi.n_gates); b.out.resize(bi.n_gates); b.op.resize(bi.n_gates);
        be.check(c2a_bool_read(be.get(), 0, bi.n_gates, b.in0.data(), b.in1.data(), b.out.data(), b.op.data()));
    } else b.gates_on_device = bi.n_gates;
    return b;
}

// boolify(&circuit, width) (main.rs:30-32) of ANY arithmetic BristolCircuit — one the host built itself (the reference's own
// Compiler::build_circuit, src/compiler.rs:321-494), not necessarily the last one built on this backend: the gates go over as SoA of
// wire ids (c2a_load_circuit) and only the bit-blast runs on the GPU.  The choice for deep and narrow circuits.
inline BristolCircuit boolify_circuit(Backend& be, const BristolCircuit& circuit, uint32_t width, bool fetch = true) {
    const uint32_t n_in = (uint32_t)circuit.info.input_name_to_wire_index.size(), n_out = (uint32_t)circuit.info.output_name_to_wire_index.size();
    be.check(c2a_load_circuit(be.get(), circuit.op.size(), circuit.in0.data(), circuit.in1.data(), circuit.out.data(), circuit.op.data(),
                              (uint32_t)circuit.wire_count, n_in, n_out));
    c2a_bool_info bi;
    be.check(c2a_boolify(be.get(), width, &bi));
    auto wire = [&](size_t W) -> size_t {
        return W < bi.m_wires ? W * width : (size_t)bi.m_wires * width + bi.aux_total + (W - bi.m_wires) * width;
    };
    BristolCircuit b;
    b.boolean = true;
    b.wire_count = bi.wire_count;
    for (auto& kv : circuit.info.input_name_to_wire_index) b.info.input_name_to_wire_index[kv.first] = wire(kv.second);
    for (auto& kv : circuit.info.output_name_to_wire_index) b.info.output_name_to_wire_index[kv.first] = wire(kv.second);
    for (auto& kv : circuit.info.constants) b.info.constants[kv.first] = ConstantInfo{kv.second.value, wire(kv.second.wire_index)};
    b.io_widths = std::make_pair(std::vector<size_t>(bi.n_in, width), std::vector<size_t>(bi.n_out, width));
    if (fetch) {
        b.in0.resize(bi.n_gates); b.in1.resize(bi.n_gates); b.out.resize(bi.n_gates); b.op.resize(bi.n_gates);
        be.check(c2a_bool_read(be.get(), 0, bi.n_gates, b.in0.data(), b.in1.data(), b.out.data(), b.op.data()));
    } else b.gates_on_device = bi.n_gates;
    return b;
}

}  // namespace host
}  // namespace c2a

// integration.cpp — the reference's integration tests (tests/integration.rs:279-441) written against the C++ host
// mirror (circom-2-arithc_amd/host/c2a_host.hpp) and run through the C ABI.  The reference compiles a .circom file
// first; here each test replays the calls its unroller makes on `Compiler` for that file.
// Built twice by tests/test_cpp_host.py: against libc2a_emul.so (CPU suite) and libc2a_hip.so (-m gpu).
#include <cstdio>
#include <functional>
#include <map>
#include <string>

#include "../../circom-2-arithc_amd/host/c2a_host.hpp"

using namespace c2a::host;

static int failures = 0;
#define CHECK(cond)                                                                        \
    do {                                                                                   \
        if (!(cond)) { std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } \
    } while (0)

// ArithmeticCircuit::run of tests/integration.rs:191-237 with the per-op semantics of :94-115 (u32)
static std::map<std::string, uint32_t> run(const BristolCircuit& c, const std::map<std::string, uint32_t>& inputs) {
    std::vector<uint32_t> w(c.wire_count, 0);
    for (auto& kv : inputs) w[c.info.input_name_to_wire_index.at(kv.first)] = kv.second;
    for (auto& kv : c.info.constants) w[kv.second.wire_index] = (uint32_t)std::stoul(kv.second.value);
    for (size_t k = 0; k < c.n_gates(); ++k) {
        const uint32_t a = w[c.in0[k]], b = w[c.in1[k]];
        uint32_t r = 0;
        switch (static_cast<AGateType>(c.op[k])) {
        case AGateType::AAdd: r = a + b; break;            case AGateType::ADiv: r = a / b; break;
        case AGateType::AEq: r = a == b; break;            case AGateType::AGEq: r = a >= b; break;
        case AGateType::AGt: r = a > b; break;             case AGateType::ALEq: r = a <= b; break;
        case AGateType::ALt: r = a < b; break;             case AGateType::AMul: r = a * b; break;
        case AGateType::ANeq: r = a != b; break;           case AGateType::ASub: r = a - b; break;
        case AGateType::AXor: r = a ^ b; break;
        case AGateType::APow: { r = 1; for (uint32_t i = 0; i < b; ++i) r *= a; } break;
        case AGateType::AIntDiv: r = a / b; break;         case AGateType::AMod: r = a % b; break;
        case AGateType::AShiftL: r = a << b; break;        case AGateType::AShiftR: r = a >> b; break;
        case AGateType::ABoolOr: r = (a != 0 || b != 0); break;
        case AGateType::ABoolAnd: r = (a != 0 && b != 0); break;
        case AGateType::ABitOr: r = a | b; break;          case AGateType::ABitAnd: r = a & b; break;
        }
        w[c.out[k]] = r;
    }
    std::map<std::string, uint32_t> out;
    for (auto& kv : c.info.output_name_to_wire_index) out[kv.first] = w[kv.second];
    return out;
}

// the same harness on the GPU: c2a_eval on the circuit the back end still holds (named inputs in, named outputs out)
static std::map<std::string, uint32_t> run_gpu(Backend& be, const BristolCircuit& c, const std::map<std::string, uint32_t>& inputs, bool boolean) {
    const size_t n_in = c.info.input_name_to_wire_index.size(), n_out = c.info.output_name_to_wire_index.size();
    std::vector<uint64_t> in(n_in, 0), out(n_out, 0), cv;
    std::vector<uint32_t> cw;
    for (auto& kv : inputs) in[c.info.input_name_to_wire_index.at(kv.first)] = kv.second;
    for (auto& kv : c.info.constants) { cw.push_back((uint32_t)kv.second.wire_index); cv.push_back(std::stoull(kv.second.value)); }
    be.check(c2a_eval(be.get(), boolean ? 1 : 0, 32, 1, in.data(), (uint32_t)cw.size(), cw.data(), cv.data(), out.data()));
    std::map<std::string, uint32_t> r;
    for (auto& kv : c.info.output_name_to_wire_index) r[kv.first] = (uint32_t)out[kv.second - (c.wire_count - n_out)];
    return r;
}

// `out_k <== x_a op x_b`: random signal, gate, connection (process.rs:466-475, :266-269)
static void binop(Compiler& ac, uint32_t& next_sid, AGateType op, uint32_t a, uint32_t b, uint32_t out_sid) {
    const uint32_t r = next_sid++;
    ac.add_signal(r, "0.random_" + std::to_string(r), std::nullopt);
    ac.add_gate(op, a, b, r);
    ac.add_connection(r, out_sid);
}

static void simulation_test(const char* name, Backend& be, const std::function<void(Compiler&)>& build,
                            const std::map<std::string, uint32_t>& inputs, const std::map<std::string, uint32_t>& expected) {
    std::printf("%s\n", name);
    Compiler compiler(be);
    build(compiler);
    const BristolCircuit circuit = compiler.build_circuit();
    const auto outputs = run(circuit, inputs);
    for (auto& kv : expected) { CHECK(outputs.count(kv.first) == 1); CHECK(outputs.at(kv.first) == kv.second); }
    const auto gpu = run_gpu(be, circuit, inputs, false);
    for (auto& kv : expected) { CHECK(gpu.count(kv.first) == 1); CHECK(gpu.at(kv.first) == kv.second); }
    boolify(compiler, circuit, 32);                         // --boolify-width 32: the boolean circuit gives the same answers
    const auto gpu_b = run_gpu(be, circuit, inputs, true);
    for (auto& kv : expected) { CHECK(gpu_b.count(kv.first) == 1); CHECK(gpu_b.at(kv.first) == kv.second); }
}

int main() {
    Backend be(0);

    simulation_test("test_add_zero", be, [](Compiler& ac) {                          // integration.rs:279-286
        ac.add_signal(0, "0.in", std::nullopt); ac.add_signal(1, "0.out", std::nullopt);
        ac.add_signal(2, "0.const_signal_0", 0u);
        uint32_t sid = 3; binop(ac, sid, AGateType::AAdd, 0, 2, 1);
        ac.add_inputs(ac.get_signals("0.in")); ac.add_outputs(ac.get_signals("0.out"));
    }, {{"0.in", 42}}, {{"0.out", 42}});

    simulation_test("test_sum", be, [](Compiler& ac) {                               // integration.rs:365-372
        ac.add_signal(0, "0.a", std::nullopt); ac.add_signal(1, "0.b", std::nullopt); ac.add_signal(2, "0.out", std::nullopt);
        uint32_t sid = 3; binop(ac, sid, AGateType::AAdd, 0, 1, 2);
        ac.add_inputs(ac.get_signals("0.a")); ac.add_inputs(ac.get_signals("0.b")); ac.add_outputs(ac.get_signals("0.out"));
    }, {{"0.a", 3}, {"0.b", 5}}, {{"0.out", 8}});

    simulation_test("test_x_eq_x", be, [](Compiler& ac) {                            // integration.rs:375-382
        ac.add_signal(0, "0.x", std::nullopt); ac.add_signal(1, "0.out", std::nullopt);
        uint32_t sid = 2; binop(ac, sid, AGateType::AEq, 0, 0, 1);
        ac.add_inputs(ac.get_signals("0.x")); ac.add_outputs(ac.get_signals("0.out"));
    }, {{"0.x", 37}}, {{"0.out", 1}});

    {   // integration.rs:335-362
        std::map<std::string, uint32_t> in, exp;
        for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) {
            const std::string ij = "[" + std::to_string(i) + "][" + std::to_string(j) + "]";
            in["0.a" + ij] = 2; in["0.b" + ij] = 2; exp["0.out" + ij] = 4;
        }
        simulation_test("test_matrix_element_multiplication", be, [](Compiler& ac) {
            uint32_t sid = 0;
            for (const char* nm : {"a", "b", "out"})
                for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j)
                    ac.add_signal(sid++, std::string("0.") + nm + "[" + std::to_string(i) + "][" + std::to_string(j) + "]", std::nullopt);
            for (uint32_t k = 0; k < 4; ++k) binop(ac, sid, AGateType::AMul, k, 4 + k, 8 + k);
            ac.add_inputs(ac.get_signals("0.a")); ac.add_inputs(ac.get_signals("0.b")); ac.add_outputs(ac.get_signals("0.out"));
        }, in, exp);
    }

    {   // test_infix_ops, integration.rs:289-332
        struct S { const char* out; AGateType op; uint32_t a, b, expect; };
        const S stmts[] = {
            {"mul_2_3", AGateType::AMul, 2, 3, 6}, {"idiv_4_3", AGateType::AIntDiv, 4, 3, 1}, {"add_3_4", AGateType::AAdd, 3, 4, 7},
            {"sub_4_1", AGateType::ASub, 4, 1, 3}, {"pow_2_4", AGateType::APow, 2, 4, 16}, {"mod_5_3", AGateType::AMod, 5, 3, 2},
            {"shl_5_1", AGateType::AShiftL, 5, 1, 10}, {"shr_5_1", AGateType::AShiftR, 5, 1, 2}, {"leq_2_3", AGateType::ALEq, 2, 3, 1},
            {"leq_3_3", AGateType::ALEq, 3, 3, 1}, {"leq_4_3", AGateType::ALEq, 4, 3, 0}, {"geq_2_3", AGateType::AGEq, 2, 3, 0},
            {"geq_3_3", AGateType::AGEq, 3, 3, 1}, {"geq_4_3", AGateType::AGEq, 4, 3, 1}, {"lt_2_3", AGateType::ALt, 2, 3, 1},
            {"lt_3_3", AGateType::ALt, 3, 3, 0}, {"lt_4_3", AGateType::ALt, 4, 3, 0}, {"gt_2_3", AGateType::AGt, 2, 3, 0},
            {"gt_3_3", AGateType::AGt, 3, 3, 0}, {"gt_4_3", AGateType::AGt, 4, 3, 1}, {"eq_2_3", AGateType::AEq, 2, 3, 0},
            {"eq_3_3", AGateType::AEq, 3, 3, 1}, {"neq_2_3", AGateType::ANeq, 2, 3, 1}, {"neq_3_3", AGateType::ANeq, 3, 3, 0},
            {"or_0_1", AGateType::ABoolOr, 0, 1, 1}, {"and_0_1", AGateType::ABoolAnd, 0, 1, 0}, {"bit_or_1_3", AGateType::ABitOr, 1, 3, 3},
            {"bit_and_1_3", AGateType::ABitAnd, 1, 3, 1}, {"bit_xor_1_3", AGateType::AXor, 1, 3, 2}};
        std::map<std::string, uint32_t> in, exp;
        for (uint32_t i = 0; i < 6; ++i) in["0.x" + std::to_string(i)] = i;
        for (auto& st : stmts) exp[std::string("0.") + st.out] = st.expect;
        simulation_test("test_infix_ops", be, [&](Compiler& ac) {
            for (uint32_t i = 0; i < 6; ++i) ac.add_signal(i, "0.x" + std::to_string(i), std::nullopt);
            uint32_t k = 0;
            for (auto& st : stmts) ac.add_signal(6 + k++, std::string("0.") + st.out, std::nullopt);
            uint32_t sid = 6 + 29; k = 0;
            for (auto& st : stmts) binop(ac, sid, st.op, st.a, st.b, 6 + k++);
            for (uint32_t i = 0; i < 6; ++i) ac.add_inputs(ac.get_signals("0.x" + std::to_string(i)));
            for (auto& st : stmts) ac.add_outputs(ac.get_signals(std::string("0.") + st.out));
        }, in, exp);
    }

    {   // test_constant_sum, integration.rs:393-415
        std::printf("test_constant_sum\n");
        Compiler ac(be);
        ac.add_signal(0, "0.out", std::nullopt); ac.add_signal(1, "0.const_signal_8", 8u); ac.add_connection(1, 0);
        ac.add_outputs(ac.get_signals("0.out"));
        const BristolCircuit circuit = ac.build_circuit();
        CHECK(circuit.info.constants.size() == 1);
        CHECK(circuit.info.constants.count("0.const_signal_8_1") == 1);
        CHECK((circuit.info.constants.at("0.const_signal_8_1") == ConstantInfo{"8", 0}));
    }
    {   // test_direct_output, integration.rs:417-441
        std::printf("test_direct_output\n");
        Compiler ac(be);
        ac.add_signal(0, "0.out", std::nullopt); ac.add_signal(1, "0.const_signal_42", 42u); ac.add_connection(1, 0);
        ac.add_outputs(ac.get_signals("0.out"));
        const BristolCircuit circuit = ac.build_circuit();
        CHECK((circuit.info.output_name_to_wire_index == std::map<std::string, size_t>{{"0.out", 0}}));
        CHECK(circuit.info.constants.size() == 1);
        CHECK((circuit.info.constants.at("0.const_signal_42_1") == ConstantInfo{"42", 0}));
    }
    {   // builder errors, compiler.rs:584-795 (unit tests of add_connection's two merge errors) + a dependency cycle
        std::printf("test_errors\n");
        Compiler ac(be);
        ac.add_signal(0, "0.a", std::nullopt); ac.add_signal(1, "0.b", 1u); ac.add_signal(2, "0.c", 2u);
        try { ac.add_connection(1, 2); CHECK(false); } catch (const CircuitError& e) { CHECK(std::string(e.what()) == "Cannot merge constant nodes"); }
        try { ac.add_signal(0, "0.dup", std::nullopt); CHECK(false); } catch (const CircuitError& e) { CHECK(std::string(e.what()) == "Signal already declared"); }
        Compiler cyc(be);
        cyc.add_signal(0, "0.x", std::nullopt); cyc.add_signal(1, "0.y", std::nullopt); cyc.add_signal(2, "0.z", std::nullopt);
        cyc.add_gate(AGateType::AAdd, 0, 2, 1);       // y = x + z
        cyc.add_gate(AGateType::AAdd, 1, 0, 2);       // z = y + x   -> cycle
        try { cyc.build_circuit(); CHECK(false); } catch (const CircuitError& e) {
            CHECK(e.kind == CircuitError::CyclicDependency);
            CHECK(std::string(e.what()) == "Cyclic dependency: detected at i=0");
        }
    }
    {   // --boolify-width: same answers from the boolean circuit (spot check through the text writer's data)
        std::printf("test_boolify_sum\n");
        Compiler ac(be);
        ac.add_signal(0, "0.a", std::nullopt); ac.add_signal(1, "0.b", std::nullopt); ac.add_signal(2, "0.out", std::nullopt);
        uint32_t sid = 3; binop(ac, sid, AGateType::AAdd, 0, 1, 2);
        ac.add_inputs(ac.get_signals("0.a")); ac.add_inputs(ac.get_signals("0.b")); ac.add_outputs(ac.get_signals("0.out"));
        const BristolCircuit circuit = ac.build_circuit();
        const BristolCircuit b = boolify(ac, circuit, 8);
        CHECK(b.io_widths.has_value() && b.io_widths->first == (std::vector<size_t>{8, 8}) && b.io_widths->second == (std::vector<size_t>{8}));
        std::vector<uint8_t> w(b.wire_count, 0);
        const uint32_t av = 200, bv = 100;
        for (int i = 0; i < 8; ++i) { w[b.info.input_name_to_wire_index.at("0.a") + i] = (av >> i) & 1; w[b.info.input_name_to_wire_index.at("0.b") + i] = (bv >> i) & 1; }
        for (size_t k = 0; k < b.n_gates(); ++k)
            w[b.out[k]] = b.op[k] == C2A_XOR ? (w[b.in0[k]] ^ w[b.in1[k]]) : b.op[k] == C2A_AND ? (w[b.in0[k]] & w[b.in1[k]]) : (uint8_t)(1 - w[b.in0[k]]);
        uint32_t got = 0;
        for (int i = 0; i < 8; ++i) got |= (uint32_t)w[b.info.output_name_to_wire_index.at("0.out") + i] << i;
        CHECK(got == ((av + bv) & 0xFF));
        // boolify(&circuit, width) of a circuit handed over as it is (c2a_load_circuit): the same boolean circuit, gate for gate
        std::printf("test_boolify_of_a_host_built_circuit\n");
        const BristolCircuit b2 = boolify_circuit(be, circuit, 8);
        CHECK(b2.wire_count == b.wire_count && b2.in0 == b.in0 && b2.in1 == b.in1 && b2.out == b.out && b2.op == b.op);
        CHECK(b2.info.input_name_to_wire_index == b.info.input_name_to_wire_index && b2.info.output_name_to_wire_index == b.info.output_name_to_wire_index);
    }
    std::printf(failures ? "%d check(s) FAILED\n" : "all checks passed\n", failures);
    return failures ? 1 : 0;
}

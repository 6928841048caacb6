// c2a_templates.h — host-side generator of the per-(op,width) bit-blast templates.
//
// Stands in for the per-gate expansion inside boolify(&circuit, width) (src/main.rs:30-32; the crate's
// source is absent from the reference tree, so the expansion is OUR frozen spec).  A template is the
// gate list of one arithmetic gate with symbolic wires: A[i] / B[i] operand bits, O[i] result bits,
// X[k] gate-private aux wires numbered in allocation order.  The GPU map kernel (k_boolify) rebases
// the symbols per arithmetic gate.  Semantics: unsigned arithmetic mod 2^w, matching
// tests/integration.rs:94-115 wherever that is defined.
#pragma once
#include <cstdint>
#include <vector>

namespace c2a {

enum : uint32_t { kRefA = 0u << 30, kRefB = 1u << 30, kRefO = 2u << 30, kRefX = 3u << 30 };
enum : uint32_t { kXor = 0, kAnd = 1, kInv = 2 };

struct TemplateEntry { uint32_t in0, in1, out, op; };

class TemplateBuilder {
public:
    explicit TemplateBuilder(uint32_t w) : w_(w) {}
    std::vector<TemplateEntry> gates;
    uint32_t aux = 0;

    void build(uint32_t op);

private:
    using W = uint32_t;              // symbolic wire
    using Vec = std::vector<W>;
    static constexpr W NEW = 0xFFFFFFFFu;
    uint32_t w_;

    W A(uint32_t i) const { return kRefA | i; }
    W B(uint32_t i) const { return kRefB | i; }
    W O(uint32_t i) const { return kRefO | i; }
    W emit(uint32_t op, W a, W b, W dst) {
        if (dst == NEW) dst = kRefX | aux++;
        gates.push_back({a, b, dst, op});
        return dst;
    }
    W XOR(W a, W b, W d = NEW) { return emit(kXor, a, b, d); }
    W AND(W a, W b, W d = NEW) { return emit(kAnd, a, b, d); }
    W INV(W a, W d = NEW) { return emit(kInv, a, a, d); }

    Vec bitsA() const { Vec v(w_); for (uint32_t i = 0; i < w_; ++i) v[i] = A(i); return v; }
    Vec bitsB() const { Vec v(w_); for (uint32_t i = 0; i < w_; ++i) v[i] = B(i); return v; }
    Vec bitsO() const { Vec v(w_); for (uint32_t i = 0; i < w_; ++i) v[i] = O(i); return v; }

    // D = P + Q (mod 2^m); dst[i]==NEW allocates; returns the result wires
    Vec add(const Vec& P, const Vec& Q, Vec dst) {
        const uint32_t m = (uint32_t)P.size();
        dst[0] = XOR(P[0], Q[0], dst[0]);
        if (m == 1) return dst;
        W c = AND(P[0], Q[0]);
        for (uint32_t i = 1; i + 1 < m; ++i) {
            W x = XOR(P[i], c);
            W y = XOR(Q[i], c);
            dst[i] = XOR(x, Q[i], dst[i]);
            W t = AND(x, y);
            c = XOR(c, t);
        }
        W t = XOR(P[m - 1], Q[m - 1]);
        dst[m - 1] = XOR(t, c, dst[m - 1]);
        return dst;
    }
    // D = P - Q (mod 2^m)
    Vec sub(const Vec& P, const Vec& Q, Vec dst) {
        const uint32_t m = (uint32_t)P.size();
        dst[0] = XOR(P[0], Q[0], dst[0]);
        if (m == 1) return dst;
        W br = AND(dst[0], Q[0]);
        for (uint32_t i = 1; i + 1 < m; ++i) {
            W t = XOR(P[i], Q[i]);
            W u = XOR(Q[i], br);
            dst[i] = XOR(t, br, dst[i]);
            W v = AND(t, u);
            br = XOR(br, v);
        }
        W t = XOR(P[m - 1], Q[m - 1]);
        dst[m - 1] = XOR(t, br, dst[m - 1]);
        return dst;
    }
    // D = P - Q over all m bits (fresh wires) + final borrow (P < Q)
    W sub_borrow(const Vec& P, const Vec& Q, Vec& D) {
        const uint32_t m = (uint32_t)P.size();
        D.assign(m, 0);
        D[0] = XOR(P[0], Q[0]);
        W br = AND(D[0], Q[0]);
        for (uint32_t i = 1; i < m; ++i) {
            W t = XOR(P[i], Q[i]);
            W u = XOR(Q[i], br);
            D[i] = XOR(t, br);
            W v = AND(t, u);
            br = XOR(br, v);
        }
        return br;
    }
    // (P < Q) unsigned -> dst
    W ult(const Vec& P, const Vec& Q, W dst) {
        const uint32_t m = (uint32_t)P.size();
        W t = XOR(P[0], Q[0]);
        if (m == 1) return AND(t, Q[0], dst);
        W br = AND(t, Q[0]);
        for (uint32_t i = 1; i < m; ++i) {
            W t2 = XOR(P[i], Q[i]);
            W u = XOR(Q[i], br);
            W v = AND(t2, u);
            br = XOR(br, v, i == m - 1 ? dst : NEW);
        }
        return br;
    }
    // (P == Q) -> dst
    W eq(const Vec& P, const Vec& Q, W dst) {
        const uint32_t m = (uint32_t)P.size();
        Vec nz(m);
        for (uint32_t i = 0; i < m; ++i) {
            W d = XOR(P[i], Q[i]);
            nz[i] = INV(d, m == 1 ? dst : NEW);
        }
        W acc = nz[0];
        for (uint32_t i = 1; i < m; ++i) acc = AND(acc, nz[i], i == m - 1 ? dst : NEW);
        return acc;
    }
    // (P == 0) -> fresh wire
    W is_zero(const Vec& P) {
        const uint32_t m = (uint32_t)P.size();
        Vec nz(m);
        for (uint32_t i = 0; i < m; ++i) nz[i] = INV(P[i]);
        W acc = nz[0];
        for (uint32_t i = 1; i < m; ++i) acc = AND(acc, nz[i]);
        return acc;
    }
    void zero_fill() { for (uint32_t i = 1; i < w_; ++i) XOR(A(0), A(0), O(i)); }
    // D = P * Q (mod 2^m), shift-and-add rows
    Vec mul(const Vec& P, const Vec& Q, Vec dst) {
        const uint32_t m = (uint32_t)P.size();
        Vec acc(m);
        for (uint32_t i = 0; i < m; ++i) acc[i] = AND(P[i], Q[0], i == 0 ? dst[0] : NEW);
        dst[0] = acc[0];
        for (uint32_t j = 1; j < m; ++j) {
            Vec pp(m - j), lo(acc.begin() + j, acc.end()), d(m - j, NEW);
            for (uint32_t i = j; i < m; ++i) pp[i - j] = AND(P[i - j], Q[j]);
            d[0] = dst[j];
            Vec r = add(lo, pp, d);
            for (uint32_t i = j; i < m; ++i) acc[i] = r[i - j];
            dst[j] = acc[j];
        }
        return dst;
    }
};

inline void TemplateBuilder::build(uint32_t op) {
    const uint32_t w = w_;
    gates.clear();
    aux = 0;
    switch (op) {
    case 10 /*AXor*/: for (uint32_t i = 0; i < w; ++i) XOR(A(i), B(i), O(i)); break;
    case 19 /*ABitAnd*/: for (uint32_t i = 0; i < w; ++i) AND(A(i), B(i), O(i)); break;
    case 18 /*ABitOr*/:
        for (uint32_t i = 0; i < w; ++i) { W t = XOR(A(i), B(i)); W u = AND(A(i), B(i)); XOR(t, u, O(i)); }
        break;
    case 0 /*AAdd*/: add(bitsA(), bitsB(), bitsO()); break;
    case 9 /*ASub*/: sub(bitsA(), bitsB(), bitsO()); break;
    case 7 /*AMul*/: mul(bitsA(), bitsB(), bitsO()); break;
    case 6 /*ALt*/: ult(bitsA(), bitsB(), O(0)); zero_fill(); break;
    case 4 /*AGt*/: ult(bitsB(), bitsA(), O(0)); zero_fill(); break;
    case 3 /*AGEq*/: { W r = ult(bitsA(), bitsB(), NEW); INV(r, O(0)); zero_fill(); } break;
    case 5 /*ALEq*/: { W r = ult(bitsB(), bitsA(), NEW); INV(r, O(0)); zero_fill(); } break;
    case 2 /*AEq*/: eq(bitsA(), bitsB(), O(0)); zero_fill(); break;
    case 8 /*ANeq*/: { W r = eq(bitsA(), bitsB(), NEW); INV(r, O(0)); zero_fill(); } break;
    case 16 /*ABoolOr*/: {
        W za = is_zero(bitsA()), zb = is_zero(bitsB());
        W t = AND(za, zb);
        INV(t, O(0));
        zero_fill();
    } break;
    case 17 /*ABoolAnd*/: {
        W za = is_zero(bitsA()), zb = is_zero(bitsB());
        W na = INV(za), nb = INV(zb);
        AND(na, nb, O(0));
        zero_fill();
    } break;
    case 14 /*AShiftL*/:
    case 15 /*AShiftR*/: {
        const bool left = op == 14;
        uint32_t K = 0;
        while ((1ull << K) < w) ++K;
        Vec cur = bitsA(), nxt(w);
        for (uint32_t k = 0; k < K; ++k) {
            const uint32_t sh = 1u << k;
            const W s = B(k);
            for (uint32_t i = 0; i < w; ++i) {
                const bool in_range = left ? (i >= sh) : ((uint64_t)i + sh < w);
                if (in_range) {
                    const W src = left ? cur[i - sh] : cur[i + sh];
                    W t = XOR(cur[i], src);
                    W m = AND(s, t);
                    nxt[i] = XOR(cur[i], m);
                } else {
                    W m = AND(s, cur[i]);
                    nxt[i] = XOR(cur[i], m);
                }
            }
            cur.swap(nxt);
        }
        Vec nb(w);
        for (uint32_t j = K; j < w; ++j) nb[j] = INV(B(j));
        W acc = nb[K];
        for (uint32_t j = K + 1; j < w; ++j) acc = AND(acc, nb[j]);
        for (uint32_t i = 0; i < w; ++i) AND(cur[i], acc, O(i));
    } break;
    case 1 /*ADiv*/:
    case 12 /*AIntDiv*/:
    case 13 /*AMod*/: {
        const bool want_q = op != 13;
        const W z = XOR(A(0), A(0));
        Vec R(w, z), Qx(w + 1), Rp(w + 1), D;
        for (uint32_t j = 0; j < w; ++j) Qx[j] = B(j);
        Qx[w] = z;
        for (uint32_t it = 0; it < w; ++it) {
            const uint32_t i = w - 1 - it;
            Rp[0] = A(i);
            for (uint32_t j = 1; j <= w; ++j) Rp[j] = R[j - 1];
            const W borrow = sub_borrow(Rp, Qx, D);
            if (want_q) INV(borrow, O(i));
            const bool last = it == w - 1;
            if (last && want_q) break;
            for (uint32_t j = 0; j < w; ++j) {
                W t = XOR(D[j], Rp[j]);
                W m = AND(borrow, t);
                R[j] = XOR(D[j], m, (last && !want_q) ? O(j) : NEW);
            }
        }
    } break;
    case 11 /*APow*/: {
        const W z = XOR(A(0), A(0));
        const W one = INV(z);
        Vec res(w), base = bitsA();
        {
            W t = XOR(A(0), one);
            W m = AND(B(0), t);
            res[0] = XOR(one, m, w == 1 ? O(0) : NEW);
            for (uint32_t j = 1; j < w; ++j) res[j] = AND(B(0), A(j));
        }
        for (uint32_t i = 1; i < w; ++i) {
            base = mul(base, base, Vec(w, NEW));
            Vec prod = mul(res, base, Vec(w, NEW));
            for (uint32_t j = 0; j < w; ++j) {
                W t = XOR(prod[j], res[j]);
                W m = AND(B(i), t);
                res[j] = XOR(res[j], m, i == w - 1 ? O(j) : NEW);
            }
        }
    } break;
    default: break;
    }
}

}  // namespace c2a

// c2a_wave.h — the wave-level primitives of gfx950 that the dataflow launch (c2a_peel.h) is written in, each next to its
// twin for the host emulation of the CPU test-suite (tests/emul/hip_emul.h, -DC2A_EMULATE): lane reads and writes, DPP row
// shifts, uniformity hints, wave priority, kernel arguments in scalar registers of their own, branch weights, compiler
// fences for pending loads, SCALAR-MEMORY atomics and stores with their fixed result registers.  Everything that differs
// between the product build and the emulated one lives HERE, so that the kernels themselves read the same in both.
#pragma once
#include "c2a_platform.h"

namespace c2a {

// wave-uniform pick of one lane's value (v_readlane: a few cycles; ds_bpermute through __shfl costs ~100)
__device__ __forceinline__ u32 rdlane(u32 v, u32 j) {
#ifdef C2A_EMULATE
    return __shfl(v, (int)j, 64);
#else
    return (u32)__builtin_amdgcn_readlane((int)v, (int)j);
#endif
}
// v with lane j replaced by the wave-uniform x (v_writelane: one instruction; a compare + select costs three)
template <int J> __device__ __forceinline__ u32 wrlane_c(u32 x, u32 v) {
#ifdef C2A_EMULATE
    return (threadIdx.x & 63u) == (u32)J ? x : v;
#else
    asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(__builtin_amdgcn_readfirstlane((int)x)), "n"(J));
    return v;
#endif
}
// lane i takes the value of lane i + 8 of its row of 16 lanes (one DPP move; lanes 8..15 of a row keep their own)
__device__ __forceinline__ u32 row_shl8(u32 v) {
#ifdef C2A_EMULATE
    const u32 l = threadIdx.x & 63u;
    const u32 up = (u32)__shfl(v, (int)((l & 15u) < 8u ? l + 8u : l), 64);
    return up;
#else
    return (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x108, 0xF, 0xF, false);
#endif
}
// every lane takes the value of lane K of its quad (four neighbouring lanes): one DPP move
template <int K> __device__ __forceinline__ u32 quad_bcast(u32 v) {
#ifdef C2A_EMULATE
    const u32 l = threadIdx.x & 63u;
    return (u32)__shfl(v, (int)((l & ~3u) + K), 64);
#else
    return (u32)__builtin_amdgcn_update_dpp((int)v, (int)v, K * 0x55, 0xF, 0xF, false);      // quad_perm:[K,K,K,K]
#endif
}
// a (all lanes) if the wave-uniform j is not 0, else b — without a branch
__device__ __forceinline__ u32 select_uniform(u32 j, u32 a, u32 b) {
#ifdef C2A_EMULATE
    return j ? a : b;
#else
    const u64 m = (j & 1u) ? ~0ull : 0ull;
    u32 r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(b), "v"(a), "s"(m));
    return r;
#endif
}
// x + 1 unless bit B of the wave-uniform f is set (s_bitcmp0 + s_addc: two scalar instructions; as `x + (cond ? 1 : 0)` the compiler
// makes a 64-bit select and a second compare of it: four)
template <int B> __device__ __forceinline__ u32 inc_unless_bit(u32 x, u32 f) {
#ifdef C2A_EMULATE
    return x + (((f >> B) & 1u) ? 0u : 1u);
#else
    u32 r;
    asm("s_bitcmp0_b32 %2, %3\n\ts_addc_u32 %0, %1, 0" : "=s"(r) : "s"(x), "s"(f), "n"(B) : "scc");
    return r;
#endif
}
__device__ __forceinline__ u32 uniform(u32 v) {
#ifdef C2A_EMULATE
    return v;
#else
    return (u32)__builtin_amdgcn_readfirstlane((int)v);
#endif
}
__device__ __forceinline__ uint4 uniform4(const uint4& v) { return make_uint4(uniform(v.x), uniform(v.y), uniform(v.z), uniform(v.w)); }
__device__ __forceinline__ u64 rdlane64(u64 v, u32 j) { return (u64)rdlane((u32)v, j) | ((u64)rdlane((u32)(v >> 32), j) << 32); }

__device__ __forceinline__ ull c2a_now() {
#ifdef C2A_EMULATE
    return 0;
#else
    return wall_clock64();      // constant 100 MHz
#endif
}
__device__ __forceinline__ void peel_sleep(int units) {
#ifndef C2A_EMULATE
    if (units <= 1) __builtin_amdgcn_s_sleep(1); else if (units <= 4) __builtin_amdgcn_s_sleep(4); else if (units <= 16) __builtin_amdgcn_s_sleep(16); else if (units <= 64) __builtin_amdgcn_s_sleep(64); else __builtin_amdgcn_s_sleep(127);
#else
    (void)units;
    hipemu_wave_yield();        // (the emulation interleaves the waves of a workgroup at these points only)
#endif
}
// Where one lane does something and the wave then LEAVES a loop (break / return), the lanes must be seen to meet again
// first: otherwise the compiler threads the jump into both sides of the `if (lane == 0)`, the loop exit becomes a join of
// a divergent branch, and every value carried around that loop is handled as divergent (vector registers, masked code).
template <int P> __device__ __forceinline__ void wave_priority_() {
#if !defined(C2A_EMULATE) && !defined(C2A_NO_PRIO)
    __builtin_amdgcn_s_setprio(P);
#endif
}
#define wave_priority(p) wave_priority_<p>()
__device__ __forceinline__ void wave_join() {
#ifndef C2A_EMULATE
    __builtin_amdgcn_wave_barrier();
#endif
}
// A kernel argument in scalar registers OF ITS OWN.  The argument block arrives as one 16-register load; under register
// pressure the allocator spills and reloads that block as a unit — sixteen v_readlane for every pointer the chain step
// touches (measured: 315 -> 180 vector instructions per gate).  A copy through an opaque s_mov makes each pointer its own
// two-register value.  (The copy is made on a GLOBAL-address-space pointer and cast back: a generic pointer of unknown
// origin would turn every access through it into a flat_ instruction.)
template <class T> __device__ __forceinline__ T* own_sgprs(T* p) {
#ifdef C2A_EMULATE
    return p;
#else
    typedef __attribute__((address_space(1))) T* G;
    G q;
    asm volatile("s_mov_b64 %0, %1" : "=s"(q) : "s"((G)p));
    return (T*)q;
#endif
}
// A pointer READ FROM MEMORY, as a wave-uniform global pointer (STATS only).
__device__ __forceinline__ ull* uniform_ptr(ull* p) {
#ifdef C2A_EMULATE
    return p;
#else
    const u64 v = (u64)p;
    const u64 u = (u64)(u32)__builtin_amdgcn_readfirstlane((u32)v) | ((u64)(u32)__builtin_amdgcn_readfirstlane((u32)(v >> 32)) << 32);
    return (ull*)(__attribute__((address_space(1))) ull*)u;
#endif
}
__device__ __forceinline__ u32 own_sgpr(u32 v) {
#ifdef C2A_EMULATE
    return v;
#else
    u32 q;
    asm volatile("s_mov_b32 %0, %1" : "=s"(q) : "s"(v));
    return q;
#endif
}
// Branch weights for the block layout: a TAKEN branch costs a wave its instruction buffer (~16+ clocks against 4 for one
// that falls through), and a chain step runs some fifty branches — the hot path should fall through, the cold code
// (re-reads, deep trees, the cold list loop, giving up) sit out of line.
#ifndef C2A_HINTS
#define C2A_HINTS 1
#endif
#if C2A_HINTS
#define C2A_LIKELY(x) __builtin_expect(!!(x), 1)
#define C2A_UNLIKELY(x) __builtin_expect(!!(x), 0)
#else
#define C2A_LIKELY(x) (x)
#define C2A_UNLIKELY(x) (x)
#endif
// the compiler must treat v as used (and redefined) here: pins the wait for a pending load to this point
#ifdef C2A_EMULATE
#define C2A_PIN(v) ((void)0)
#define C2A_OPAQUE(v) ((void)0)
#define C2A_OPAQUE_S(v) ((void)0)
#else
#define C2A_PIN(v) asm volatile("" : "+v"(v) :: "memory")
#define C2A_OPAQUE(v) asm volatile("" : "+v"(v))      /* the compiler knows nothing about v from here on (no hoisting of what is computed from it) */
#define C2A_OPAQUE_S(v) asm volatile("" : "+s"(v))    /* ... a wave-uniform value (keeps two blocks that end alike from being merged) */
#endif

// SCALAR TICKETS.  A returning atomic of ONE lane does not need the vector memory path: gfx950 still has the scalar-memory
// atomics of the gfx9 family (s_atomic_add ... glc; executed in L2 like the vector ones, exact next to them on the same
// word — tools/ubench/satomic.hip).  Measured on MI355X, scattered words, 2 048 waves: 520 ns against 880 ns for a one-lane
// global_atomic_add, and they are counted by lgkmcnt, NOT by vmcnt: waiting for a ticket does not wait for vector loads
// issued after it (vmcnt is in order, and where the number of loads behind the atomic depends on a branch the compiler
// waits for the smallest count: the wait for a hand-off ticket was a wait for the next step's static loads).
// The compiler knows no scalar atomics, so they are inline assembly, and it cannot know that their result registers are
// in flight: a copy or a spill of one before the wait would read garbage.  Hence the results land in FIXED registers
// s97..s101 that the compiler never touches — the kernel is compiled with amdgpu_num_sgpr: registers beyond that budget
// are RESERVED in the allocator (SIRegisterInfo::getReservedRegs) — and are read (s_mov) only after the wait.  The
// clobber lists make the kernel descriptor cover them (tests/test_abi_exports.py reads it back from the built library).
//   s100, s101   the two fill[] tickets of the step in flight (ONE pair: a step reads it at its top, then issues the next)
//   s[98:99]     hand-off: tickets taken so far on the array, producer side | consumer side (the pre-op value)
//   s97          hand-off: BEGIN unit counted (value unused: the wait is what matters)
// Under emulation the "registers" are a small array per lane and the atomics go through lane 0.
// (Same-box A/B of the peel stage while this went in: one-lane vector atomics 10.42 ms; scalar ones through C++ helpers 10.61 —
// the helpers' branches and the spills of a smaller register budget cost more than the shorter round trip brings —; the
// hand-written sequences below 10.28; with the branch weights 9.97 / 9.70.)
constexpr int kSregBase = 97, kSregBegin = 97, kSregPush = 98, kSregFill0 = 100, kSregFill1 = 101;
#define C2A_SREG_BUDGET 103      /* 6 of the budget are VCC, FLAT_SCRATCH, XNACK_MASK: s0..s96 for the compiler */
#define C2A_SREG_CLOBBERS "s97", "s98", "s99", "s100", "s101"
// SCALAR STORES for one-lane stores of wave-uniform data (s_store_dword[x4]: gfx9 family, still there on gfx950).  They go
// through the scalar data cache, which is WRITE-BACK: nothing is visible to anybody else until s_dcache_wb — fine for data
// that only later launches read, provided every wave writes the cache back before it ends (tools/ubench/sstore.hip: 0 of
// 3.3 M results wrong with the write-back, 75 % lost without; neighbouring words written on different CUs both survive).
#ifdef C2A_EMULATE
__device__ __forceinline__ void sstore_x4(uint4* p, u32 a, u32 b, u32 c, u32 d) { if ((threadIdx.x & 63u) == 0) *p = make_uint4(a, b, c, d); }
__device__ __forceinline__ void sstore_x1(u32* p, u32 a) { if ((threadIdx.x & 63u) == 0) *p = a; }
__device__ __forceinline__ void sstore_x1_at(u32* base, u32 byte_off, u32 a) { if ((threadIdx.x & 63u) == 0) base[byte_off / 4u] = a; }
__device__ __forceinline__ void sstore_flush() {}
#else
typedef u32 c2a_v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void sstore_x4(uint4* p, u32 a, u32 b, u32 c, u32 d) {
    typedef __attribute__((address_space(1))) uint4* G;
    c2a_v4u q = {uniform(a), uniform(b), uniform(c), uniform(d)};
    asm volatile("s_store_dwordx4 %0, %1, 0x0" :: "s"(q), "s"((G)p) : "memory");
}
__device__ __forceinline__ void sstore_x1(u32* p, u32 a) {
    typedef __attribute__((address_space(1))) u32* G;
    asm volatile("s_store_dword %0, %1, 0x0" :: "s"(uniform(a)), "s"((G)p) : "memory");
}
// (the word at base + a 32-bit byte offset in a register: no 64-bit address arithmetic)
__device__ __forceinline__ void sstore_x1_at(u32* base, u32 byte_off, u32 a) {
    typedef __attribute__((address_space(1))) u32* G;
    asm volatile("s_store_dword %0, %1, %2" :: "s"(uniform(a)), "s"((G)base), "s"(uniform(byte_off)) : "memory");
}
__device__ __forceinline__ void sstore_flush() { asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory"); }
#endif
#ifdef C2A_EMULATE
#define C2A_PEEL_KERNEL_ATTR
struct SRegs { u32 r[8]; };
template <int R> __device__ __forceinline__ void sreg_add32(SRegs& sr, u32* p, u32 v) {
    u32 t = 0;
    if ((threadIdx.x & 63u) == 0) t = atomicAdd(p, v);
    sr.r[R - kSregBase] = (u32)__shfl((int)t, 0, 64);
}
template <int R> __device__ __forceinline__ void sreg_add64(SRegs& sr, u64* p, u64 v) {
    ull t = 0;
    if ((threadIdx.x & 63u) == 0) t = atomicAdd(reinterpret_cast<ull*>(p), (ull)v);
    sr.r[R - kSregBase] = (u32)__shfl((int)(u32)t, 0, 64); sr.r[R - kSregBase + 1] = (u32)__shfl((int)(u32)(t >> 32), 0, 64);
}
template <int R> __device__ __forceinline__ void sreg_inc64(SRegs& sr, u64* p) { sreg_add64<R>(sr, p, 1ull); }
template <int R> __device__ __forceinline__ void sreg_inc32(SRegs& sr, u32* p) { sreg_add32<R>(sr, p, 1u); }
template <int R> __device__ __forceinline__ void sreg_set(SRegs& sr, u32 v) { sr.r[R - kSregBase] = v; }
template <int R> __device__ __forceinline__ u32 sreg_get(const SRegs& sr) { return sr.r[R - kSregBase]; }
__device__ __forceinline__ void sreg_wait() {}
// the fill[] ticket of one producer, branch-free: a producer with other consumers (cnt > 1) takes a ticket on its own
// word; any other adds 0 to this wave's DUMMY word behind fill[n] (always 0: "ticket 0 of cnt" reads as the last one
// for cnt 1 and as nothing for cnt 0)
template <int R> __device__ __forceinline__ void sfill_take(SRegs& sr, u32* fill, u32 dep, u32 cnt, u32 dummy_idx) {
    u32 t = 0;
    if ((threadIdx.x & 63u) == 0) t = atomicAdd(&fill[cnt > 1u ? dep : dummy_idx], cnt > 1u ? 1u : 0u);
    sr.r[R - kSregBase] = (u32)__shfl((int)t, 0, 64);
}
template <int R0, int R1> __device__ __forceinline__ u32 sfill_claims(const SRegs& sr, u32 cnt0, u32 cnt1) {
    return (sr.r[R0 - kSregBase] + 1u == cnt0 ? 1u : 0u) | (sr.r[R1 - kSregBase] + 1u == cnt1 ? 2u : 0u);
}
#else
#define C2A_PEEL_KERNEL_ATTR __attribute__((amdgpu_num_sgpr(C2A_SREG_BUDGET)))
struct SRegs {};
template <int R> __device__ __forceinline__ void sreg_add32(SRegs&, u32* p, u32 v) {
    typedef __attribute__((address_space(1))) u32* G;
    asm volatile("s_mov_b32 s%c2, %1\n\ts_atomic_add s%c2, %0, 0x0 glc" :: "s"((G)p), "s"(v), "n"(R) : "memory", C2A_SREG_CLOBBERS);
}
template <int R> __device__ __forceinline__ void sreg_add64(SRegs&, u64* p, u64 v) {
    typedef __attribute__((address_space(1))) u64* G;
    static_assert((R & 1) == 0, "an aligned register pair");
    asm volatile("s_mov_b64 s[%c2:%c3], %1\n\ts_atomic_add_x2 s[%c2:%c3], %0, 0x0 glc" :: "s"((G)p), "s"(v), "n"(R), "n"(R + 1) : "memory", C2A_SREG_CLOBBERS);
}
// (+ 1 on the low word of a pair / on a word: the constant goes in as an immediate — as an operand the compiler puts it in a
// register first, one more move on the claim path)
template <int R> __device__ __forceinline__ void sreg_inc64(SRegs&, u64* p) {
    typedef __attribute__((address_space(1))) u64* G;
    static_assert((R & 1) == 0, "an aligned register pair");
    asm volatile("s_mov_b64 s[%c1:%c2], 1\n\ts_atomic_add_x2 s[%c1:%c2], %0, 0x0 glc" :: "s"((G)p), "n"(R), "n"(R + 1) : "memory", C2A_SREG_CLOBBERS);
}
template <int R> __device__ __forceinline__ void sreg_inc32(SRegs&, u32* p) {
    typedef __attribute__((address_space(1))) u32* G;
    asm volatile("s_mov_b32 s%c1, 1\n\ts_atomic_add s%c1, %0, 0x0 glc" :: "s"((G)p), "n"(R) : "memory", C2A_SREG_CLOBBERS);
}
template <int R> __device__ __forceinline__ void sreg_set(SRegs&, u32 v) { asm volatile("s_mov_b32 s%c1, %0" :: "s"(v), "n"(R) : C2A_SREG_CLOBBERS); }
template <int R> __device__ __forceinline__ u32 sreg_get(const SRegs&) { u32 v; asm volatile("s_mov_b32 %0, s%c1" : "=s"(v) : "n"(R)); return v; }
__device__ __forceinline__ void sreg_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// (five scalar instructions, no branch; the word's byte offset goes in a register: fill[] stays below 4 GB — n < 2^30,
// checked by the host)
template <int R> __device__ __forceinline__ void sfill_take(SRegs&, u32* fill, u32 dep, u32 cnt, u32 dummy_idx) {
    typedef __attribute__((address_space(1))) u32* G;
    u32 off;
    asm volatile("s_cmp_gt_u32 %2, 1\n\ts_cselect_b32 s%c5, 1, 0\n\ts_cselect_b32 %0, %3, %4\n\ts_lshl_b32 %0, %0, 2\n\ts_atomic_add s%c5, %1, %0 glc"
                 : "=&s"(off) : "s"((G)fill), "s"(uniform(cnt)), "s"(uniform(dep)), "s"(uniform(dummy_idx)), "n"(R) : "memory", "scc", C2A_SREG_CLOBBERS);
}
// both tickets are back: bit l = producer l is claimed (its ticket was the last of cnt)
template <int R0, int R1> __device__ __forceinline__ u32 sfill_claims(const SRegs&, u32 cnt0, u32 cnt1) {
    u32 r, t;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_add_u32 %0, s%c4, 1\n\ts_cmp_eq_u32 %0, %2\n\ts_cselect_b32 %0, 1, 0\n\t"
                 "s_add_u32 %1, s%c5, 1\n\ts_cmp_eq_u32 %1, %3\n\ts_cselect_b32 %1, 2, 0\n\ts_or_b32 %0, %0, %1"
                 : "=&s"(r), "=&s"(t) : "s"(uniform(cnt0)), "s"(uniform(cnt1)), "n"(R0), "n"(R1) : "memory", "scc");
    return r;
}
#endif

}  // namespace c2a

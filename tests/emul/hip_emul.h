// hip_emul.h — host emulation of the small HIP subset used by circom-2-arithc_amd/csrc.
//
// TEST INFRASTRUCTURE ONLY.  There is no GPU in the build container, so the kernel sources are also
// compiled with g++ against this header (-DC2A_EMULATE) into tests/emul/libc2a_emul.so and driven
// through the same C ABI by the `-m "not gpu"` tests.  The product library (libc2a_hip.so) never
// includes this file and the Python host layer never loads the emulated library.
//
// Model: one OS thread; a launch runs blocks sequentially; the threads of a block are fibers (a 12-instruction
// x86-64 stack switch: glibc's swapcontext makes a sigprocmask system call per switch) that run round-robin
// between __syncthreads() calls.  Atomics are plain read-modify-writes.
// Wave-level intrinsics (__ballot, __shfl*, __any, __all) rendezvous the 64 fibers of a wave.
#pragma once
#if !defined(__x86_64__)
#error "hip_emul.h: the fiber switch is written for x86-64"
#endif

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }

// void hipemu_switch(void** save_sp, void* load_sp): park the caller (callee-saved registers on its stack, stack
// pointer in *save_sp) and resume the context whose stack pointer is load_sp
extern "C" void hipemu_switch(void** save_sp, void* load_sp);
__asm__(R"(
.text
.weak hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
.size hipemu_switch,.-hipemu_switch
)");

namespace hipemu {
struct Fiber {
    void* sp = nullptr;
    char* stack = nullptr;
    dim3 tid;
    bool done = false;
    int wait_kind = 0;  // 0 none, 1 block barrier, 2 wave rendezvous, 3 the wave lets the other waves of the block run
};
struct State {
    dim3 tid, bid, bdim, gdim;
    Fiber* cur = nullptr;
    void* sched_sp = nullptr;
    std::function<void()> body;
    // wave rendezvous scratch
    unsigned long long wave_vals[64];
    unsigned wave_arrived = 0;
    unsigned wave_epoch = 0;
};
inline State& st() { static State s; return s; }
inline void trampoline() {
    State& s = st();
    s.body();
    s.cur->done = true;
    hipemu_switch(&s.cur->sp, s.sched_sp);
    std::abort();      // a finished fiber is never resumed
}
inline void yield(int kind) {
    State& s = st();
    if (!s.cur) { std::fprintf(stderr, "hip_emul: sync primitive used in a NOSYNC launch\n"); std::abort(); }
    s.cur->wait_kind = kind;
    hipemu_switch(&s.cur->sp, s.sched_sp);
}
constexpr size_t kStack = 256 * 1024;
// fiber stacks are reused by every launch (a 256 KB malloc is an mmap + page faults each time)
inline char* stack_pool(unsigned t) {
    static std::vector<char*> pool;
    while (pool.size() <= t) pool.push_back((char*)std::malloc(kStack));
    return pool[t];
}

template <class F>
void run_block_fibers(unsigned nthreads, dim3 bdim, F&& per_thread) {
    State& s = st();
    std::vector<Fiber> fibers(nthreads);
    for (unsigned t = 0; t < nthreads; ++t) {
        Fiber& f = fibers[t];
        f.stack = stack_pool(t);
        f.tid = dim3(t % bdim.x, (t / bdim.x) % bdim.y, t / (bdim.x * bdim.y));
        // first switch "returns" into trampoline with the stack aligned as after a call
        void** top = reinterpret_cast<void**>((reinterpret_cast<uintptr_t>(f.stack) + kStack) & ~uintptr_t(15));
        top[-1] = nullptr;                                   // trampoline's (unused) return address
        top[-2] = reinterpret_cast<void*>(&trampoline);
        for (int r = 3; r <= 8; ++r) top[-r] = nullptr;      // rbp rbx r12 r13 r14 r15
        f.sp = top - 8;
    }
    s.body = per_thread;
    // Waves are scheduled as units so that wave rendezvous complete: run each wave's fibers round-robin
    // until every fiber of the wave is done, parked at a block barrier, or has stepped aside for the other waves
    // (hipemu_wave_yield: a polling loop's back-off — the only points where the waves of a block interleave); then the
    // next wave; repeat.  A barrier opens when nothing else in the block can run.
    unsigned nwaves = (nthreads + 63) / 64;
    for (;;) {
        bool any_alive = false;
        for (unsigned wv = 0; wv < nwaves; ++wv) {
            unsigned lo = wv * 64, hi = std::min(nthreads, lo + 64);
            for (;;) {
                bool progressed = false;
                for (unsigned t = lo; t < hi; ++t) {
                    Fiber& f = fibers[t];
                    if (f.done || f.wait_kind == 1 || f.wait_kind == 3) continue;
                    f.wait_kind = 0;
                    s.cur = &f; s.tid = f.tid;
                    hipemu_switch(&s.sched_sp, f.sp);
                    progressed = true;
                }
                bool all_parked = true;
                for (unsigned t = lo; t < hi; ++t) if (!fibers[t].done && fibers[t].wait_kind != 1 && fibers[t].wait_kind != 3) all_parked = false;
                if (all_parked) break;
                if (!progressed) { std::fprintf(stderr, "hip_emul: wave deadlock\n"); std::abort(); }
            }
        }
        bool any_aside = false;
        for (unsigned t = 0; t < nthreads; ++t) if (!fibers[t].done) { any_alive = true; if (fibers[t].wait_kind == 3) { any_aside = true; fibers[t].wait_kind = 0; } }
        if (!any_alive) break;
        if (!any_aside) for (unsigned t = 0; t < nthreads; ++t) if (!fibers[t].done) fibers[t].wait_kind = 0;      // the barrier opens
    }
    s.cur = nullptr;
}
}  // namespace hipemu

#define threadIdx (hipemu::st().tid)
#define blockIdx (hipemu::st().bid)
#define blockDim (hipemu::st().bdim)
#define gridDim (hipemu::st().gdim)
#define warpSize 64

inline void __syncthreads() { hipemu::yield(1); }
// every thread of a wave, in uniform control flow: the other waves of the block run until they step aside or finish
inline void hipemu_wave_yield() { if (hipemu::st().cur) hipemu::yield(3); }
inline void __threadfence() {}
inline void __threadfence_block() {}

// ---- wave intrinsics (all 64 lanes of the wave must call them convergently) ----
namespace hipemu {
inline unsigned lane_id() { State& s = st(); return (s.tid.x + s.tid.y * s.bdim.x + s.tid.z * s.bdim.x * s.bdim.y) & 63; }
inline unsigned wave_width() {
    State& s = st();
    unsigned nt = s.bdim.x * s.bdim.y * s.bdim.z;
    unsigned flat = s.tid.x + s.tid.y * s.bdim.x + s.tid.z * s.bdim.x * s.bdim.y;
    unsigned lo = flat & ~63u;
    return std::min(64u, nt - lo);
}
// deposit v, wait for the lanes of the `width`-aligned subgroup (whole wave for width 64), return a pointer to the
// wave's 64 value slots (only the caller's subgroup slots are meaningful; valid until that subgroup's next rendezvous)
struct SubSync { unsigned arrived = 0; unsigned epoch = 0; };
inline const unsigned long long* rendezvous(unsigned long long v, unsigned width = 64) {
    State& s = st();
    static unsigned long long snapshot[64];
    static SubSync sync[7][64];                     // [log2 width][subgroup]
    unsigned lg = 0; while ((1u << lg) < width) ++lg;
    const unsigned lane = lane_id();
    const unsigned sub = lane / width, base = sub * width;
    unsigned members = std::min(wave_width() > base ? wave_width() - base : 0u, width);
    SubSync& y = sync[lg][sub];
    const unsigned my_epoch = y.epoch;
    s.wave_vals[lane] = v;
    y.arrived++;
    if (y.arrived == members) {
        for (unsigned i = base; i < base + width; ++i) snapshot[i] = i < base + members ? s.wave_vals[i] : 0;
        y.arrived = 0;
        y.epoch++;
    } else {
        while (y.epoch == my_epoch) yield(2);
    }
    return snapshot;
}
}  // namespace hipemu
inline unsigned long long __ballot(int pred) {
    const unsigned long long* v = hipemu::rendezvous(pred ? 1 : 0);
    unsigned long long m = 0;
    for (unsigned i = 0; i < 64; ++i) if (v[i]) m |= 1ull << i;
    // second rendezvous so nobody overwrites the snapshot before all lanes have read it
    unsigned long long r = m; hipemu::rendezvous(0);
    return r;
}
inline int __any(int pred) { return __ballot(pred) != 0; }
inline int __all(int pred) { unsigned w = hipemu::wave_width(); unsigned long long full = w == 64 ? ~0ull : ((1ull << w) - 1); return (__ballot(pred) & full) == full; }
template <class T> inline T __shfl(T val, int src, int width = 64) {
    static_assert(sizeof(T) <= 8, "shfl");
    unsigned long long bits = 0; std::memcpy(&bits, &val, sizeof(T));
    const unsigned long long* v = hipemu::rendezvous(bits, (unsigned)width);
    unsigned lane = hipemu::lane_id();
    unsigned base = lane & ~(unsigned)(width - 1);
    unsigned long long got = v[base + ((unsigned)src & (unsigned)(width - 1))];
    hipemu::rendezvous(0, (unsigned)width);
    T out; std::memcpy(&out, &got, sizeof(T)); return out;
}
template <class T> inline T __shfl_up(T val, unsigned delta, int width = 64) {
    unsigned lane = hipemu::lane_id(); unsigned in = lane & (unsigned)(width - 1);
    T o = __shfl(val, in >= delta ? (int)(in - delta) : (int)in, width);
    return in >= delta ? o : val;
}
template <class T> inline T __shfl_down(T val, unsigned delta, int width = 64) {
    unsigned lane = hipemu::lane_id(); unsigned in = lane & (unsigned)(width - 1);
    T o = __shfl(val, in + delta < (unsigned)width ? (int)(in + delta) : (int)in, width);
    return in + delta < (unsigned)width ? o : val;
}
template <class T> inline T __shfl_xor(T val, int mask, int width = 64) {
    unsigned lane = hipemu::lane_id(); unsigned in = lane & (unsigned)(width - 1);
    return __shfl(val, (int)(in ^ (unsigned)mask), width);
}
inline int __popcll(unsigned long long x) { return __builtin_popcountll(x); }
inline int __popc(unsigned x) { return __builtin_popcount(x); }
inline int __ffsll(unsigned long long x) { return __builtin_ffsll((long long)x); }
inline int __ffs(unsigned x) { return __builtin_ffs((int)x); }
inline int __clz(unsigned x) { return x ? __builtin_clz(x) : 32; }
inline int __clzll(unsigned long long x) { return x ? __builtin_clzll(x) : 64; }
inline unsigned __lane_id() { return hipemu::lane_id(); }

// ---- atomics ----
template <class T> inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <class T> inline T atomicSub(T* p, T v) { T o = *p; *p = o - v; return o; }
template <class T> inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <class T> inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }
template <class T> inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> inline T atomicCAS(T* p, T c, T v) { T o = *p; if (o == c) *p = v; return o; }

// ---- scoped atomic load/store builtins (sc1 accesses on the GPU; plain here) + misc ----
#ifndef __HIP_MEMORY_SCOPE_AGENT
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#endif
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) ((void)(*(p) = (v)))
inline void __builtin_amdgcn_s_sleep(int) {}
inline unsigned long long wall_clock64() { return 0; }

// ---- runtime ----
typedef int hipError_t;
typedef int hipStream_t;
struct hipEventImpl { std::chrono::steady_clock::time_point t; };
typedef hipEventImpl* hipEvent_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorPeerAccessAlreadyEnabled = 704 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
struct hipDeviceProp_t { int multiProcessorCount; char name[64]; char gcnArchName[64]; size_t totalGlobalMem; };

inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hip_emul error"; }
inline hipError_t hipGetLastError() { return hipSuccess; }
// the two "devices" are one address space; HIPEMU_NO_PEER=1 (tests) says they cannot map each other: the library then gathers
inline hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = std::getenv("HIPEMU_NO_PEER") ? 0 : 1; return hipSuccess; }
inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
inline hipError_t hipSetDevice(int) { return hipSuccess; }
inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 2; return hipSuccess; }      // two "devices" so that the multi-device path can run
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    p->multiProcessorCount = 4; std::strcpy(p->name, "hip_emul"); std::strcpy(p->gcnArchName, "emul"); p->totalGlobalMem = 1ull << 34; return hipSuccess;
}
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); if (*p) std::memset(*p, 0xA5, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
constexpr unsigned hipHostMallocPortable = 1, hipHostMallocMapped = 2;
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = std::calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
inline hipError_t hipHostFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipHostGetDevicePointer(void** d, void* h, unsigned) { *d = h; return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t) { if (n) std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamCreate(hipStream_t* s) { *s = 0; return hipSuccess; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipEventImpl(); return hipSuccess; }
inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess;
}
inline hipError_t hipMemGetInfo(size_t* f, size_t* t) { *f = 1ull << 33; *t = 1ull << 34; return hipSuccess; }

// ---- launches ----
// kernels that use __syncthreads / wave intrinsics: fibers
template <class K, class... A>
inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t, hipStream_t, A... args) {
    hipemu::State& s = hipemu::st();
    s.gdim = grid; s.bdim = block;
    unsigned nthreads = block.x * block.y * block.z;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                s.bid = dim3(bx, by, bz);
                hipemu::run_block_fibers(nthreads, block, [&]() { kernel(args...); });
            }
}
// kernels whose workgroups talk to each other while they run (c2a_peel2.h: claim waves feed tournament waves through
// mailboxes in memory): EVERY block of the grid is alive at once; the waves of all blocks take turns, a wave runs until
// its fibers are done, at a block barrier or have stepped aside (hipemu_wave_yield: the back-off of a polling loop).
// C2A_EMUL_SEED=<n> shuffles the order of the waves in every pass and lets a wave sit a pass out now and then: a different
// interleaving of the protocol per seed (a randomized-schedule stress of what only a real GPU runs truly in parallel).
template <class K, class... A>
inline void hipemuLaunchConcurrent(K kernel, dim3 grid, dim3 block, size_t, hipStream_t, A... args) {
    using namespace hipemu;
    State& s = st();
    s.gdim = grid; s.bdim = block;
    const unsigned nthreads = block.x * block.y * block.z, nblocks = grid.x;
    const unsigned wpb = (nthreads + 63) / 64, nwaves = nblocks * wpb;
    std::vector<Fiber> fibers((size_t)nblocks * nthreads);
    std::vector<dim3> bids(fibers.size());
    for (unsigned b = 0; b < nblocks; ++b)
        for (unsigned t = 0; t < nthreads; ++t) {
            Fiber& f = fibers[(size_t)b * nthreads + t];
            f.stack = stack_pool(b * nthreads + t);
            f.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            bids[(size_t)b * nthreads + t] = dim3(b, 0, 0);
            void** top = reinterpret_cast<void**>((reinterpret_cast<uintptr_t>(f.stack) + kStack) & ~uintptr_t(15));
            top[-1] = nullptr;
            top[-2] = reinterpret_cast<void*>(&trampoline);
            for (int r = 3; r <= 8; ++r) top[-r] = nullptr;
            f.sp = top - 8;
        }
    s.body = [&]() { kernel(args...); };
    unsigned long long rng = 0;
    if (const char* e = std::getenv("C2A_EMUL_SEED")) rng = std::strtoull(e, nullptr, 10) * 0x9E3779B97F4A7C15ull + 1;
    auto next_rand = [&]() { rng ^= rng << 13; rng ^= rng >> 7; rng ^= rng << 17; return rng; };
    std::vector<unsigned> order(nwaves);
    unsigned long long passes = 0, pass_limit = 50000000ull;
    if (const char* e = std::getenv("C2A_EMUL_PASS_LIMIT")) pass_limit = std::strtoull(e, nullptr, 10);
    for (unsigned i = 0; i < nwaves; ++i) order[i] = i;
    for (;;) {
        if (rng) for (unsigned i = nwaves; i > 1; --i) std::swap(order[i - 1], order[next_rand() % i]);
        bool ran_any = false;
        for (unsigned oi = 0; oi < nwaves; ++oi) {
            const unsigned wv = order[oi];
            if (rng && (next_rand() & 3u) == 0 && ran_any) continue;          // (this wave sits the pass out)
            const unsigned b = wv / wpb, lo = b * nthreads + (wv % wpb) * 64, hi = std::min(b * nthreads + nthreads, lo + 64);
            unsigned long long spins = 0;
            for (;;) {
                if (++spins == 4000000000ull) {
                    std::fprintf(stderr, "hip_emul: block %u never steps aside (a polling loop without a back-off?) kinds:", b);
                    for (unsigned t = lo; t < hi; ++t) std::fprintf(stderr, " %d", fibers[t].done ? 9 : fibers[t].wait_kind);
                    std::fprintf(stderr, "\n");
                    std::abort();
                }
                bool progressed = false;
                for (unsigned t = lo; t < hi; ++t) {
                    Fiber& f = fibers[t];
                    if (f.done || f.wait_kind == 1 || f.wait_kind == 3) continue;
                    f.wait_kind = 0;
                    s.cur = &f; s.tid = f.tid; s.bid = bids[t];
                    hipemu_switch(&s.sched_sp, f.sp);
                    progressed = true; ran_any = true;
                }
                bool all_parked = true;
                for (unsigned t = lo; t < hi; ++t) if (!fibers[t].done && fibers[t].wait_kind != 1 && fibers[t].wait_kind != 3) all_parked = false;
                if (all_parked) break;
                if (!progressed) { std::fprintf(stderr, "hip_emul: wave deadlock\n"); std::abort(); }
            }
        }
        // end of a pass: waves that stepped aside may go on; a block barrier opens when every live fiber of the block is at it
        bool any_alive = false;
        for (unsigned b = 0; b < nblocks; ++b) {
            bool at_barrier = true, alive = false;
            for (unsigned t = b * nthreads; t < (b + 1) * nthreads; ++t) {
                Fiber& f = fibers[t];
                if (f.done) continue;
                alive = true;
                if (f.wait_kind == 3) f.wait_kind = 0;
                if (f.wait_kind != 1) at_barrier = false;
            }
            if (alive && at_barrier) for (unsigned t = b * nthreads; t < (b + 1) * nthreads; ++t) if (!fibers[t].done) fibers[t].wait_kind = 0;
            any_alive = any_alive || alive;
        }
        if (!any_alive) break;
        if (++passes == pass_limit) {
            std::fprintf(stderr, "hip_emul: concurrent launch still alive after %llu passes:", passes);
            for (unsigned b = 0; b < nblocks; ++b) {
                unsigned alive = 0;
                for (unsigned t = b * nthreads; t < (b + 1) * nthreads; ++t) alive += !fibers[t].done;
                std::fprintf(stderr, " b%u:%u", b, alive);
            }
            std::fprintf(stderr, "\n");
            std::abort();
        }
    }
    s.cur = nullptr;
}
// kernels with no intra-block synchronisation: plain loops (fast path)
template <class K, class... A>
inline void hipemuLaunchNoSync(K kernel, dim3 grid, dim3 block, size_t, hipStream_t, A... args) {
    hipemu::State& s = hipemu::st();
    s.gdim = grid; s.bdim = block; s.cur = nullptr;
    for (unsigned bx = 0; bx < grid.x; ++bx) {
        s.bid = dim3(bx, 0, 0);
        for (unsigned tx = 0; tx < block.x; ++tx) { s.tid = dim3(tx, 0, 0); kernel(args...); }
    }
}

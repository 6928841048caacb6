// Would a chain step split over TWO waves of one workgroup run faster than on one?  Wave A: the claim side of a step (a returning
// scalar atomic issued, ~60 dependent instructions, the wait for the atomic, ~40 more, a message to B through LDS); wave B: the
// tournament / record side (~210 dependent instructions, a 512-byte record load issued one message ahead, a record store).
// Against: ONE wave doing all of it in sequence (what k_peel does).  W workgroups per CU, every CU busy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__device__ __forceinline__ unsigned long long now() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
#define REP4(x) x x x x
#define REP10(x) REP4(x) REP4(x) x x
#define REP20(x) REP10(x) REP10(x)
// 4 instructions, dependent, mixed (like the step: readlane -> salu -> valu -> valu)
#define UNIT asm volatile("v_readlane_b32 %1, %0, 3\n\ts_add_u32 %1, %1, 1\n\tv_add_u32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1" : "+v"(v), "+s"(s0) :: "scc");
template <int MODE> __global__ void __launch_bounds__(128) k(unsigned long long* out, unsigned* tick, unsigned long long* rec, int iters) {
    __shared__ unsigned s_seq, s_done, s_pay[4][64];
    // LDS through inline assembly: a volatile (or atomic) LDS access makes the compiler wait for every global access in flight
    auto lds_ld = [](const unsigned* p) -> unsigned { unsigned r; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"((unsigned)(size_t)p) : "memory"); return r; };
    auto lds_st = [](unsigned* p, unsigned x) { asm volatile("ds_write_b32 %0, %1" :: "v"((unsigned)(size_t)p), "v"(x) : "memory"); };
    const unsigned lane = threadIdx.x & 63u, wv = (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned v = threadIdx.x, s0 = blockIdx.x;
    asm volatile("s_mov_b32 %0, %0" : "+s"(s0));
    if (threadIdx.x == 0) { lds_st(&s_seq, 0); lds_st(&s_done, 0); }
    __syncthreads();
    unsigned* my_tick = tick + (size_t)blockIdx.x * 64;
    unsigned long long* my_rec = rec + (size_t)blockIdx.x * 64 * 8;
    const unsigned long long t0 = now();
    if (MODE == 0) {                       // one wave does everything (wave 1 idles at once)
        if (wv == 0) {
            unsigned long long w = 0;
            for (int i = 1; i <= iters; ++i) {
                unsigned t;
                asm volatile("s_mov_b32 %0, 1\n\ts_atomic_add %0, %1, 0x0 glc" : "=&s"(t) : "s"(my_tick) : "memory");
                REP10(UNIT) REP4(UNIT)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(t) : "scc");
                REP10(UNIT)
                const unsigned long long wn = __hip_atomic_load(&my_rec[((i & 7) * 64) + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                REP20(UNIT) REP20(UNIT) REP10(UNIT) REP4(UNIT)      // ~216 instructions: tournament + record
                v += (unsigned)(wn ^ (wn >> 32));                                  // (the load is consumed behind them)
                __hip_atomic_store(&my_rec[(((i + 3) & 7) * 64) + lane], (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            v += (unsigned)w;
        }
    } else {
        if (wv == 0) {                     // A: claims
            for (int i = 1; i <= iters; ++i) {
                if (MODE != 3) {
                unsigned t;
                asm volatile("s_mov_b32 %0, 1\n\ts_atomic_add %0, %1, 0x0 glc" : "=&s"(t) : "s"(my_tick) : "memory");
                REP10(UNIT) REP4(UNIT)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(t) : "scc");
                }
                while ((unsigned)i - (unsigned)__builtin_amdgcn_readfirstlane((int)lds_ld(&s_done)) > 3u) __builtin_amdgcn_s_sleep(1);      // ring of 4
                if (MODE != 3) { REP10(UNIT) }
                lds_st(&s_pay[i & 3][lane], v);
                asm volatile("" ::: "memory");
                if (lane == 0) lds_st(&s_seq, (unsigned)i);
            }
        } else {                           // B: tournaments
            unsigned long long w = 0;
            for (int i = 1; i <= iters; ++i) {
                while ((unsigned)__builtin_amdgcn_readfirstlane((int)lds_ld(&s_seq)) < (unsigned)i) __builtin_amdgcn_s_sleep(1);
                v += lds_ld(&s_pay[i & 3][lane]);
                if (MODE != 2) {
                const unsigned long long wn = __hip_atomic_load(&my_rec[((i & 7) * 64) + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                REP20(UNIT) REP20(UNIT) REP10(UNIT) REP4(UNIT)
                v += (unsigned)(wn ^ (wn >> 32));
                __hip_atomic_store(&my_rec[(((i + 3) & 7) * 64) + lane], (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                if (lane == 0) lds_st(&s_done, (unsigned)i);
            }
            v += (unsigned)w;
        }
    }
    const unsigned long long t1 = now();
    if (lane == 0 && (MODE == 0 ? wv == 0 : wv == 1)) out[blockIdx.x] = t1 - t0;
    if (v + s0 == 0x12345) out[0] = 1;
}
template <int MODE> void run(const char* name, unsigned long long* d, unsigned* tick, unsigned long long* rec, int n_cu) {
    printf("%-40s", name);
    for (int w : {1, 2, 4, 8}) {
        const int grid = n_cu * w, iters = 3000;
        hipMemset(tick, 0, (size_t)grid * 256);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(128), 0, 0, d, tick, rec, iters);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid);
        hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("  wg/CU %d: %5.0f ns/iter (p90 %5.0f)", w, h[grid / 2] * 10.0 / iters, h[grid * 9 / 10] * 10.0 / iters);
    }
    printf("\n");
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int maxg = p.multiProcessorCount * 8;
    unsigned long long* d; hipMalloc(&d, maxg * 8);
    unsigned* tick; hipMalloc(&tick, (size_t)maxg * 256);
    unsigned long long* rec; hipMalloc(&rec, (size_t)maxg * 64 * 8 * 8); hipMemset(rec, 0, (size_t)maxg * 64 * 8 * 8);
    run<0>("one wave: claim + tournament in sequence", d, tick, rec, p.multiProcessorCount);
    run<1>("two waves: A claims, B tournaments (LDS)", d, tick, rec, p.multiProcessorCount);
    run<2>("  A's side alone (B only acknowledges)", d, tick, rec, p.multiProcessorCount);
    run<3>("  B's side alone (A only sends)", d, tick, rec, p.multiProcessorCount);
    return 0;
}

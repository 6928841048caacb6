// How fast does ONE wave get through a dependent instruction stream, and what do its neighbours on the CU cost it?
// The chain step of the dataflow peel is ~400 instructions, two thirds of them scalar: if the CU's one scalar unit is what 8
// waves per CU share, a wave's time per instruction is a THROUGHPUT figure and fewer scalar instructions (or fewer waves) help;
// if it is latency, only a shorter dependency chain does.  Streams: dependent SALU, independent SALU (4 chains), dependent
// VALU, VALU -> SGPR -> VALU ping-pong (v_readlane / v_mov), taken branches.  Waves per CU 1, 2, 4, 8, 16 (64-thread workgroups).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
template <int MODE> __global__ void k(unsigned long long* out, int iters, unsigned seed) {
    unsigned s0 = seed, s1 = seed * 3u + 1u, s2 = seed + 7u, s3 = seed ^ 5u, v = threadIdx.x + seed;
    asm volatile("s_mov_b32 %0, %0" : "+s"(s0)); asm volatile("s_mov_b32 %0, %0" : "+s"(s1));
    asm volatile("s_mov_b32 %0, %0" : "+s"(s2)); asm volatile("s_mov_b32 %0, %0" : "+s"(s3));
    const unsigned long long t0 = now();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) { REP64(asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");) }
        if (MODE == 1) { REP16(asm volatile("s_add_u32 %0, %0, %4\n\ts_add_u32 %1, %1, %4\n\ts_add_u32 %2, %2, %4\n\ts_add_u32 %3, %3, %4" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : "s"(seed) : "scc");) }
        if (MODE == 2) { REP64(asm volatile("v_add_u32 %0, %0, %1" : "+v"(v) : "s"(s1));) }
        if (MODE == 3) { REP16(asm volatile("v_readlane_b32 %1, %0, 3\n\ts_add_u32 %1, %1, 1\n\tv_add_u32 %0, %0, %1\n\tv_xor_b32 %0, %0, %1" : "+v"(v), "+s"(s0) :: "scc");) }
        if (MODE == 4) { REP16(asm volatile("s_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:\n\ts_add_u32 %0, %0, 1\n\ts_cmp_lg_u32 %0, 0\n\ts_cbranch_scc1 2f\n\ts_nop 0\n2:" : "+s"(s0) :: "scc");) }    // 2 taken branches + 3 salu per unit = 4 units of 16 -> counted as 64 "instructions" (5 per unit really)
        if (MODE == 5) { REP16(asm volatile("s_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_add_u32 %0, %0, 1\n1:\n\ts_cmp_eq_u32 %0, 0\n\ts_cbranch_scc1 2f\n\ts_nop 0\n2:" : "+s"(s0) :: "scc");) }    // the same with branches NOT taken
        if (MODE == 6) { REP16(asm volatile("v_cmp_ne_u32 vcc, %0, %1\n\ts_and_b64 vcc, vcc, exec\n\tv_cndmask_b32 %0, %0, %1, vcc\n\tv_add_u32 %0, %0, %2" : "+v"(v) : "v"(threadIdx.x), "s"(s1) : "vcc");) }
    }
    const unsigned long long t1 = now();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (s0 + s1 + s2 + s3 + v == 0x12345) out[0] = 1;
}
template <int MODE> void run(const char* name, unsigned long long* d, int n_cu) {
    printf("%-34s", name);
    for (int w : {1, 2, 4, 8, 16, 32}) {
        const int grid = n_cu * w, iters = 2000;
        k<MODE><<<grid, 64>>>(d, iters, 3);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(grid);
        hipMemcpy(h.data(), d, grid * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("  w%-2d %5.2f ns", w, h[grid / 2] * 10.0 / (iters * 64.0));
    }
    printf("   (median wave, per instruction of the unrolled body)\n");
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    unsigned long long* d; hipMalloc(&d, 256 * 32 * 8 * 4);
    printf("%s, %d CUs, waves per CU ->\n", p.name, p.multiProcessorCount);
    run<0>("dependent SALU chain", d, p.multiProcessorCount);
    run<1>("4 independent SALU chains", d, p.multiProcessorCount);
    run<2>("dependent VALU chain", d, p.multiProcessorCount);
    run<3>("readlane->salu->valu->valu", d, p.multiProcessorCount);
    run<4>("cmp+taken branch (5 instr / 4)", d, p.multiProcessorCount);
    run<5>("cmp+untaken branch (~5 / 4)", d, p.multiProcessorCount);
    run<6>("v_cmp->s_and->cndmask->add", d, p.multiProcessorCount);
    return 0;
}

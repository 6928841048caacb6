// Cost of the dataflow peel's publish step for one wave (2048 waves in flight, 8 per CU):
//  A: 512-byte write-through store (8 B per lane, sc1)  -> s_waitcnt vmcnt(0)              [store acknowledged]
//  B: returning atomicAdd on a scattered word (lanes 0,1)                                   [commit ticket]
//  C: A then B with the wait in between (the shipped protocol)
//  D: store and atomic issued back to back, ONE wait (what self-validating data would allow)
//  E: 16-byte record store (2 lanes) + 512-byte string store + atomic back to back, one wait
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
__device__ __forceinline__ unsigned long long now(unsigned& dep) {
    unsigned long long t;
    asm volatile("s_nop 0\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(dep) :: "memory");
    return t;
}
__global__ void __launch_bounds__(64) k(unsigned long long* strs, unsigned* words, const unsigned* idx, int mode, int iters, unsigned long long* out) {
    const unsigned wave = blockIdx.x, lane = threadIdx.x;
    unsigned dep = lane;
    unsigned long long total = 0;
    for (int it = 0; it < iters; ++it) {
        const unsigned i = idx[(wave * 37 + it * 101) & 0xFFFFF];
        unsigned long long* s = strs + (size_t)(i & 0x3FFFFF) * 64 + lane;
        unsigned* w = words + (size_t)((i * 2654435761u) >> 8) * 16;
        const unsigned long long t0 = now(dep);
        if (mode == 0 || mode == 2 || mode == 3 || mode == 4) __hip_atomic_store(s, (unsigned long long)dep + it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mode == 4 && lane < 2) __hip_atomic_store(strs + (size_t)((i + 7777) & 0x3FFFFF) * 64 + lane, (unsigned long long)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (mode == 0 || mode == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (mode >= 1 && lane < 2) dep += atomicAdd(w + lane * 4096, 1u);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long t1 = now(dep);
        total += t1 - t0;
    }
    if (lane == 0) out[wave] = total;
    if (dep == 0xFFFFFFFF) out[0] = 0;
}
int main() {
    const size_t nstr = 4u << 20;                 // 2 GB of strings
    unsigned long long* strs; hipMalloc(&strs, nstr * 512);
    unsigned* words; hipMalloc(&words, (size_t)(1u << 24) * 64 + (1 << 20)); hipMemset(words, 0, (size_t)(1u << 24) * 64);
    std::vector<unsigned> h(1 << 20); unsigned s = 5; for (auto& x : h) { s = s * 1664525u + 1013904223u; x = s >> 4; }
    unsigned* idx; hipMalloc(&idx, h.size() * 4); hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const int W = 2048, iters = 200;
    unsigned long long* out; hipMalloc(&out, W * 8); std::vector<unsigned long long> ho(W);
    const char* names[] = {"A store 512 B + ack", "B returning atomic", "C store, ack, atomic (shipped)", "D store + atomic, one wait", "E record + string + atomic, one wait"};
    for (int rep = 0; rep < 2; ++rep) for (int mode = 0; mode < 5; ++mode) {
        k<<<W, 64>>>(strs, words, idx, mode, iters, out);
        hipMemcpy(ho.data(), out, W * 8, hipMemcpyDeviceToHost);
        double sum = 0; for (auto v : ho) sum += v;
        if (rep) printf("%-40s %6.0f ns per step (mean over %d waves x %d steps)\n", names[mode], sum / W / iters * 10.0, W, iters);
    }
    return 0;
}

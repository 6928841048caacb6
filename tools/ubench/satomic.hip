// Scalar returning atomics (s_atomic_add ... glc: SMEM path, counted by lgkmcnt — NOT by vmcnt, so a wait for the ticket
// does not wait for vector loads issued after it) on gfx950: do they exist in hardware, are they coherent with the vector
// atomics on the same word (both execute in L2), and what does one cost next to a one-lane global_atomic_add?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__device__ __forceinline__ unsigned long long now() {
    unsigned long long t;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}
__device__ __forceinline__ unsigned s_add_ret(unsigned* p, unsigned v) {
    typedef __attribute__((address_space(1))) unsigned* G;
    unsigned r = v;
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(r) : "s"((G)p) : "memory");
    return r;
}
__device__ __forceinline__ unsigned long long s_add_ret64(unsigned long long* p, unsigned long long v) {
    typedef __attribute__((address_space(1))) unsigned long long* G;
    unsigned long long r = v;
    asm volatile("s_atomic_add_x2 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(r) : "s"((G)p) : "memory");
    return r;
}
// mode 0: scalar atomic on a shared counter; 1: one-lane vector atomic on the same counter; 2: scalar on scattered words;
// 3: vector on scattered words; 4: scalar 64-bit on the shared counter pair
__global__ void k(unsigned* ctr, unsigned* scat, const unsigned* idx, int mode, unsigned* got, unsigned long long* dt) {
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned r = 0;
    const unsigned i = __builtin_amdgcn_readfirstlane(idx[wave]);
    const unsigned long long t0 = now();
    if (mode == 0) r = s_add_ret(ctr, 1u);
    else if (mode == 1) { if (lane == 0) r = atomicAdd(ctr, 1u); r = __builtin_amdgcn_readfirstlane(r); }
    else if (mode == 2) r = s_add_ret(scat + i, 1u);
    else if (mode == 3) { if (lane == 0) r = atomicAdd(scat + i, 1u); r = __builtin_amdgcn_readfirstlane(r); }
    else r = (unsigned)s_add_ret64(reinterpret_cast<unsigned long long*>(ctr), 1ull | (1ull << 32));
    const unsigned long long t1 = now();
    if (lane == 0) { got[wave] = r; dt[wave] = t1 - t0; }
}
int main() {
    const size_t words = 64u << 20;
    unsigned *ctr, *scat, *idx, *got; unsigned long long* dt;
    hipMalloc(&ctr, 256); hipMalloc(&scat, words * 4); hipMemset(scat, 0, words * 4);
    const int max_waves = 16384;
    std::vector<unsigned> h(max_waves); unsigned s = 7;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (s >> 4) % (unsigned)(words / 32) * 32; }
    hipMalloc(&idx, max_waves * 4); hipMemcpy(idx, h.data(), max_waves * 4, hipMemcpyHostToDevice);
    hipMalloc(&got, max_waves * 4); hipMalloc(&dt, max_waves * 8);
    std::vector<unsigned> hg(max_waves); std::vector<unsigned long long> hd(max_waves);
    const char* names[] = {"scalar, one counter ", "vector, one counter ", "scalar, scattered   ", "vector, scattered   ", "scalar x2, one pair "};
    for (int mode : {1, 0, 4, 3, 2}) for (int waves : {1, 64, 2048, 16384}) {
        hipMemset(ctr, 0, 256);
        const int blocks = (waves + 3) / 4, threads = waves >= 4 ? 256 : waves * 64;
        k<<<blocks, threads>>>(ctr, scat, idx, mode, got, dt);
        if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed: %s\n", names[mode], hipGetErrorString(hipGetLastError())); return 1; }
        hipMemcpy(hg.data(), got, waves * 4, hipMemcpyDeviceToHost); hipMemcpy(hd.data(), dt, waves * 8, hipMemcpyDeviceToHost);
        unsigned c[2]; hipMemcpy(c, ctr, 8, hipMemcpyDeviceToHost);
        bool perm = true;
        if (mode == 0 || mode == 1 || mode == 4) { std::sort(hg.begin(), hg.begin() + waves); for (int j = 0; j < waves; ++j) perm = perm && hg[j] == (unsigned)j; perm = perm && c[0] == (unsigned)waves; }
        std::sort(hd.begin(), hd.begin() + waves);
        printf("%s %5d waves: median %5.0f ns  p90 %5.0f  max %5.0f   %s (counter %u %u)\n", names[mode], waves, hd[waves / 2] * 10.0, hd[waves * 9 / 10] * 10.0, hd[waves - 1] * 10.0,
               perm ? "tickets exact" : "TICKETS WRONG", c[0], c[1]);
    }
    // mixed: scalar and vector atomics on the SAME word from different waves
    hipMemset(ctr, 0, 256);
    k<<<1024, 256>>>(ctr, scat, idx, 0, got, dt); k<<<1024, 256>>>(ctr, scat, idx, 1, got, dt);
    hipDeviceSynchronize(); unsigned c; hipMemcpy(&c, ctr, 4, hipMemcpyDeviceToHost);
    printf("scalar then vector on one word: %u (expect 8192)\n", c);
    return 0;
}

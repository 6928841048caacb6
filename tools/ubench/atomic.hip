// Latency of ONE returning atomicAdd per lane on scattered addresses, as a function of how many are in flight
// (the peel issues two per frontier gate per level: the fill[] tickets).  Compared with a plain load of the same word.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__device__ __forceinline__ unsigned long long now(unsigned& dep) {
    unsigned long long t;
    asm volatile("s_nop 0\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(dep) :: "memory");
    return t;
}
// mode 0: returning atomic add, 1: plain load, 2: returning atomic on a 64-byte-private line (no line sharing), 3: two dependent atomics
__global__ void k(unsigned* buf, const unsigned* idx, int lanes_per_wave, int mode, unsigned long long* out) {
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned v = lane;
    const unsigned i = idx[wave * 2 + (lane & 1)];
    unsigned long long t0 = now(v);
    if ((int)lane < lanes_per_wave) {
        if (mode == 0 || mode == 2) v += atomicAdd(&buf[i], 1u);
        else if (mode == 1) v += buf[i];
        else { unsigned a = atomicAdd(&buf[i], 1u); v += atomicAdd(&buf[(i + a * 16 + 4096) & 0x3FFFFF0], 1u); }
    }
    unsigned long long t1 = now(v);
    if (lane == 0) out[wave] = t1 - t0;
    if (v == 0xFFFFFFFF) out[0] = 0;
}
int main() {
    const size_t words = 64u << 20;   // 256 MB
    unsigned* buf; hipMalloc(&buf, words * 4); hipMemset(buf, 0, words * 4);
    const int max_waves = 16384;
    std::vector<unsigned> h(max_waves * 2); unsigned s = 99;
    for (auto& x : h) { s = s * 1664525u + 1013904223u; x = (s >> 4) % (unsigned)(words / 16) * 16; }
    unsigned* idx; hipMalloc(&idx, h.size() * 4); hipMemcpy(idx, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    unsigned long long* out; hipMalloc(&out, max_waves * 8);
    std::vector<unsigned long long> ho(max_waves);
    const char* names[] = {"atomic", "load  ", "atomic", "2 dependent atomics"};
    for (int mode : {1, 0, 3}) for (int waves : {1, 64, 512, 2048, 8192}) for (int rep = 0; rep < 2; ++rep) {
        const int blocks = (waves + 7) / 8, threads = waves >= 8 ? 512 : waves * 64;
        k<<<blocks, threads>>>(buf, idx, 2, mode, out);
        hipMemcpy(ho.data(), out, waves * 8, hipMemcpyDeviceToHost);
        std::sort(ho.begin(), ho.begin() + waves);
        if (rep) printf("%s  %5d waves x 2 lanes: median %5.0f ns  p90 %5.0f  max %5.0f\n", names[mode], waves, ho[waves / 2] * 10.0, ho[waves * 9 / 10] * 10.0, ho[waves - 1] * 10.0);
    }
    return 0;
}

// How long does a wave wait for K coalesced 512-byte strings that the PREVIOUS launch wrote (the peel's round trip 3)?
// Launch A: every wave writes its string (8 B per lane).  Launch B (back to back): every wave loads K strings of
// "recent" nodes (random among the N written) and reports the time from issue to arrival.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__device__ __forceinline__ unsigned long long now(unsigned& dep) {
    unsigned long long t;
    asm volatile("s_nop 0\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(dep) :: "memory");
    return t;
}
__global__ void writer(unsigned long long* pstr, unsigned base, unsigned gen) {
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    pstr[(size_t)(base + wave) * 64 + lane] = ((unsigned long long)gen << 32) | (wave * 64 + lane);
}
template <int K>
__global__ void reader(const unsigned long long* pstr, const unsigned* pick, unsigned base, unsigned gen, unsigned long long* out, unsigned* bad) {
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    unsigned dep = lane;
    unsigned idx[K];
#pragma unroll
    for (int k = 0; k < K; ++k) idx[k] = pick[wave * 8 + k];
    dep += idx[0];
    unsigned long long t0 = now(dep);
    unsigned long long v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = pstr[(size_t)(base + idx[k]) * 64 + lane];
    unsigned acc = 0;
#pragma unroll
    for (int k = 0; k < K; ++k) { acc += (unsigned)v[k]; if ((unsigned)(v[k] >> 32) != gen || (unsigned)v[k] != idx[k] * 64 + lane) atomicAdd(bad, 1u); }
    dep += acc;
    unsigned long long t1 = now(dep);
    if (lane == 0) out[wave] = t1 - t0;
    if (dep == 0xFFFFFFFF) out[0] = 0;
}
int main() {
    const unsigned N = 2048;                       // strings per level
    const size_t total = 10u << 20;                // 10 M strings = 5 GB like the peel
    unsigned long long* pstr; hipMalloc(&pstr, total * 512);
    std::vector<unsigned> hp(N * 8); unsigned s = 7;
    for (auto& x : hp) { s = s * 1664525u + 1013904223u; x = (s >> 8) % N; }
    unsigned* pick; hipMalloc(&pick, hp.size() * 4); hipMemcpy(pick, hp.data(), hp.size() * 4, hipMemcpyHostToDevice);
    unsigned long long* out; hipMalloc(&out, N * 8); unsigned* bad; hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    std::vector<unsigned long long> ho(N);
    for (int rep = 0; rep < 3; ++rep) for (int K : {1, 2, 5}) {
        const unsigned base = (unsigned)((rep * 3 + K) * 100000u);
        writer<<<N / 8, 512>>>(pstr, base, 100 + rep);
        if (K == 1) reader<1><<<N / 8, 512>>>(pstr, pick, base, 100 + rep, out, bad);
        else if (K == 2) reader<2><<<N / 8, 512>>>(pstr, pick, base, 100 + rep, out, bad);
        else reader<5><<<N / 8, 512>>>(pstr, pick, base, 100 + rep, out, bad);
        hipMemcpy(ho.data(), out, N * 8, hipMemcpyDeviceToHost);
        unsigned hb; hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost);
        std::sort(ho.begin(), ho.end());
        printf("rep %d: %d strings per wave, %u waves: median %4.0f ns  p90 %4.0f  max %4.0f  (mismatches %u)\n", rep, K, N, ho[N / 2] * 10.0, ho[N * 9 / 10] * 10.0, ho[N - 1] * 10.0, hb);
    }
    return 0;
}

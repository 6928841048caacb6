// Pointer-chase latency microbenchmark (one lane, dependent loads) at several working-set sizes, plus
// the cost of a returned atomic and of an empty dependent kernel launch.  Diagnostics only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>
#include <random>
#include <chrono>

__global__ void chase(const unsigned* __restrict__ next, unsigned start, int hops, unsigned* out) {
    unsigned x = start;
    for (int i = 0; i < hops; ++i) x = next[x];
    *out = x;
}
__global__ void chase_atomic(unsigned* ctr, int hops, unsigned* out) {
    unsigned x = 0;
    for (int i = 0; i < hops; ++i) x += atomicAdd(ctr + (x & 1023u) * 16, 1u);
    *out = x;
}
__global__ void empty(unsigned* p) { if (threadIdx.x == 9999) *p = 1; }

int main() {
    std::mt19937 rng(1);
    unsigned* d_out; hipMalloc(&d_out, 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (size_t mb : {1, 8, 64, 512, 4096}) {
        size_t n = mb * 1024 * 1024 / 64;           // one hop per 64-byte line
        std::vector<unsigned> perm(n); std::iota(perm.begin(), perm.end(), 0u); std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<unsigned> nxt(n * 16, 0);
        for (size_t i = 0; i < n; ++i) nxt[(size_t)perm[i] * 16] = perm[(i + 1) % n] * 16;
        unsigned* d; hipMalloc(&d, n * 64); hipMemcpy(d, nxt.data(), n * 64, hipMemcpyHostToDevice);
        int hops = 20000;
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(a); chase<<<1, 64>>>(d, perm[0] * 16, hops, d_out); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (rep) printf("chase  working set %5zu MiB: %.1f ns/hop\n", mb, ms * 1e6 / hops);
        }
        hipFree(d);
    }
    unsigned* ctr; hipMalloc(&ctr, 1024 * 64); hipMemset(ctr, 0, 1024 * 64);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a); chase_atomic<<<1, 64>>>(ctr, 20000, d_out); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep) printf("returned atomicAdd chain: %.1f ns/op\n", ms * 1e6 / 20000);
    }
    for (int blocks : {1, 64, 512, 2048}) {
        hipEventRecord(a);
        for (int i = 0; i < 2000; ++i) empty<<<blocks, 256>>>(d_out);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("empty kernel x2000 back-to-back, %4d blocks: %.2f us/launch\n", blocks, ms * 1e3 / 2000);
    }
    return 0;
}

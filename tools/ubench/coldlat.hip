// Dependent-hop latency on COLD random lines, one lane: what one step of the dataflow launch waits for.
//   plain load (static data: gstat, clist) / sc1 load (node records) / returning atomic add (tickets on fill[])
//   over 40 MB (the size of fill[] at 10 M gates) and 2 GB, one hop per 128-byte line, random order.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <random>
#include <algorithm>

template <int MODE>
__global__ void k_chase(unsigned* ring, unsigned start, int hops, unsigned long long* t_out, unsigned* sink) {
    unsigned long long t0 = wall_clock64();
    unsigned i = start;
    for (int h = 0; h < hops; ++h) {
        unsigned* p = ring + (size_t)i * 32;
        if (MODE == 0) i = *(volatile unsigned*)p;
        else if (MODE == 1) i = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else i = atomicAdd(p, 0u);
    }
    unsigned long long t1 = wall_clock64();
    t_out[blockIdx.x] = t1 - t0; sink[blockIdx.x] = i;
}

int main() {
    const size_t sizes[2] = {40ull << 20, 2048ull << 20};
    for (int sz = 0; sz < 2; ++sz) {
        const size_t lines = sizes[sz] / 128;
        std::vector<unsigned> perm(lines);
        for (size_t i = 0; i < lines; ++i) perm[i] = (unsigned)i;
        std::mt19937_64 rng(1234);
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<unsigned> ring(lines * 32, 0);
        for (size_t i = 0; i < lines; ++i) ring[(size_t)perm[i] * 32] = perm[(i + 1) % lines];
        unsigned *d, *sink; unsigned long long* t;
        hipMalloc(&d, lines * 128); hipMalloc(&sink, 64); hipMalloc(&t, 64);
        const char* names[3] = {"plain load", "sc1 load", "returning atomic add"};
        for (int mode = 0; mode < 3; ++mode) {
            hipMemcpy(d, ring.data(), lines * 128, hipMemcpyHostToDevice);     // (rewritten by the host every time: cold for the GPU caches)
            hipDeviceSynchronize();
            const int hops = 20000;
            if (mode == 0) k_chase<0><<<1, 1>>>(d, perm[0], hops, t, sink);
            else if (mode == 1) k_chase<1><<<1, 1>>>(d, perm[0], hops, t, sink);
            else k_chase<2><<<1, 1>>>(d, perm[0], hops, t, sink);
            hipDeviceSynchronize();
            unsigned long long ht = 0; hipMemcpy(&ht, t, 8, hipMemcpyDeviceToHost);
            printf("%5zu MB, %-22s %6.0f ns per dependent hop\n", sizes[sz] >> 20, names[mode], ht * 10.0 / hops);
        }
        hipFree(d); hipFree(sink); hipFree(t);
    }
    return 0;
}

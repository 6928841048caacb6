// Does the speed of a store stream depend on WHERE hipMalloc put the buffer?  Rounds of allocations (with frees between them);
// per buffer a pure 16-byte-per-lane store sweep alone, then the four buffers together in k_boolify's mix.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((vector_size(16)));
__global__ void fill1(unsigned* a, size_t groups) {
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (size_t)gridDim.x * blockDim.x) {
        unsigned v = (unsigned)g;
        *(u32x4*)(a + 4 * g) = u32x4{v, v + 1, v + 2, v + 3};
    }
}
__global__ void fill4(unsigned* a, unsigned* b, unsigned* c, unsigned char* d, size_t groups) {
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (size_t)gridDim.x * blockDim.x) {
        unsigned v = (unsigned)g;
        *(u32x4*)(a + 4 * g) = u32x4{v, v + 1, v + 2, v + 3};
        *(u32x4*)(b + 4 * g) = u32x4{v, v, v, v};
        *(u32x4*)(c + 4 * g) = u32x4{v + 7, v, v, v};
        *(unsigned*)(d + 4 * g) = v;
    }
}
// k_boolify's shape: block b owns the groups [b * S, (b + 1) * S) of every stream and walks them 256 groups at a time
__global__ void chunk4(unsigned* a, unsigned* b, unsigned* c, unsigned char* d, size_t groups, unsigned S, int mask) {
    const size_t lo = (size_t)blockIdx.x * S, hi = lo + S < groups ? lo + S : groups;
    for (size_t g = lo + threadIdx.x; g < hi; g += blockDim.x) {
        unsigned v = (unsigned)g;
        if (mask & 1) *(u32x4*)(a + 4 * g) = u32x4{v, v + 1, v + 2, v + 3};
        if (mask & 2) *(u32x4*)(b + 4 * g) = u32x4{v, v, v, v};
        if (mask & 4) *(u32x4*)(c + 4 * g) = u32x4{v + 7, v, v, v};
        if (mask & 8) *(unsigned*)(d + 4 * g) = v;
    }
}
static hipEvent_t e0, e1;
template <class F> float timed(F f) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) { hipEventRecord(e0); f(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms; }
    return best;
}
int main() {
    const size_t G = 742172195ull / 4;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<void*> hog;
    for (int round = 0; round < 8; ++round) {
        unsigned *a, *b, *c; unsigned char* d;
        hipMalloc(&a, G * 16); hipMalloc(&b, G * 16); hipMalloc(&c, G * 16); hipMalloc(&d, G * 4);
        float ta = timed([&] { fill1<<<65536, 256>>>(a, G); }), tb = timed([&] { fill1<<<65536, 256>>>(b, G); }), tc = timed([&] { fill1<<<65536, 256>>>(c, G); });
        float td = timed([&] { fill1<<<65536, 256>>>((unsigned*)d, G / 4); });
        float t4 = timed([&] { fill4<<<65536, 256>>>(a, b, c, d, G); });
        const unsigned S = 4736;      // 256 arithmetic gates x 74 boolean gates / 4
        const unsigned nb = (unsigned)((G + S - 1) / S);
        float c15 = timed([&] { chunk4<<<nb, 256>>>(a, b, c, d, G, S, 15); });
        float c1 = timed([&] { chunk4<<<nb, 256>>>(a, b, c, d, G, S, 1); }), c2 = timed([&] { chunk4<<<nb, 256>>>(a, b, c, d, G, S, 2); }), c4 = timed([&] { chunk4<<<nb, 256>>>(a, b, c, d, G, S, 4); });
        float c7 = timed([&] { chunk4<<<nb, 256>>>(a, b, c, d, G, S, 7); });
        float k15 = timed([&] { chunk4<<<(unsigned)((G + 4 * S - 1) / (4 * S)), 1024>>>(a, b, c, d, G, 4 * S, 15); });
        printf("round %d: chunked (block = 76 KB of each stream): all four %.3f ms | a %.3f b %.3f c %.3f | a+b+c %.3f | 1024 threads x 303 KB: %.3f\n", round, c15, c1, c2, c4, c7, k15);
        printf("round %d: alone a %.0f b %.0f c %.0f d %.0f GB/s | together %.3f ms = %.0f GB/s | a %p b %p c %p d %p\n", round, G * 16.0 / ta / 1e6, G * 16.0 / tb / 1e6,
               G * 16.0 / tc / 1e6, G * 4.0 / td / 1e6, t4, G * 52.0 / t4 / 1e6, (void*)a, (void*)b, (void*)c, (void*)d);
        hipFree(a); hipFree(b); hipFree(c); hipFree(d);
        if (round % 3 == 1) { void* h; hipMalloc(&h, (size_t)5 << 30); hog.push_back(h); }      // (something stays behind now and then)
        if (round == 5) { for (void* h : hog) hipFree(h); hog.clear(); }
    }
    return 0;
}

// Three 16-byte store streams + one 4-byte one written in lock-step (k_boolify's mix) into ONE allocation: how does the time
// depend on the distance between the arrays' starts?  Array k starts at k * (size rounded up to 32 MB) + k * skew.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((vector_size(16)));
__global__ void chunk4(unsigned* a, unsigned* b, unsigned* c, unsigned char* d, size_t groups, unsigned S) {
    const size_t lo = (size_t)blockIdx.x * S, hi = lo + S < groups ? lo + S : groups;
    for (size_t g = lo + threadIdx.x; g < hi; g += blockDim.x) {
        unsigned v = (unsigned)g;
        *(u32x4*)(a + 4 * g) = u32x4{v, v + 1, v + 2, v + 3};
        *(u32x4*)(b + 4 * g) = u32x4{v, v, v, v};
        *(u32x4*)(c + 4 * g) = u32x4{v + 7, v, v, v};
        *(unsigned*)(d + 4 * g) = v;
    }
}
int main(int argc, char** argv) {
    const size_t G = 742172195ull / 4, R = ((G * 16 + (32u << 20) - 1) >> 25) << 25;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const unsigned S = 4736, nb = (unsigned)((G + S - 1) / S);
    for (int round = 0; round < 3; ++round) {
        char* slab; hipMalloc(&slab, 4 * R + (64u << 20));
        void* hog = nullptr; if (round == 1) hipMalloc(&hog, (size_t)7 << 30);
        printf("slab %p:", (void*)slab);
        const size_t skews[] = {0, 256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1u << 20, 2u << 20, 3u << 20, 4u << 20, 5u << 20, 8u << 20, 12u << 20, 16u << 20};
        for (size_t sk : skews) {
            unsigned* a = (unsigned*)slab; unsigned* b = (unsigned*)(slab + R + sk); unsigned* c = (unsigned*)(slab + 2 * R + 2 * sk); unsigned char* d = (unsigned char*)(slab + 3 * R + 3 * sk);
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) { hipEventRecord(e0); chunk4<<<nb, 256>>>(a, b, c, d, G, S); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (rep && ms < best) best = ms; }
            printf(" %zu:%.3f", sk, best);
        }
        printf("\n");
        hipFree(slab); if (hog) hipFree(hog);
    }
    return 0;
}

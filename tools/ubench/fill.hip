// Write-stream ceiling: 3 u32x4 streams + 1 u32 stream (the store mix of k_boolify), no reads.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((vector_size(16)));
__global__ void fill4(unsigned* a, unsigned* b, unsigned* c, unsigned char* d, size_t groups) {
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < groups; g += (size_t)gridDim.x * blockDim.x) {
        unsigned v = (unsigned)g;
        *(u32x4*)(a + 4 * g) = u32x4{v, v + 1, v + 2, v + 3};
        *(u32x4*)(b + 4 * g) = u32x4{v, v, v, v};
        *(u32x4*)(c + 4 * g) = u32x4{v + 7, v, v, v};
        *(unsigned*)(d + 4 * g) = v;
    }
}
__global__ void copy16(const u32x4* in, u32x4* out, size_t n) {
    for (size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x; g < n; g += (size_t)gridDim.x * blockDim.x) out[g] = in[g];
}
int main() {
    size_t G = 742172195ull / 4;
    unsigned *a, *b, *c; unsigned char* d;
    hipMalloc(&a, G * 16); hipMalloc(&b, G * 16); hipMalloc(&c, G * 16); hipMalloc(&d, G * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int blocks : {2048, 8192, 65536, 262144}) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0); fill4<<<blocks, 256>>>(a, b, c, d, G); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("fill (3 x16B + 1 x4B per lane), %6d blocks: %.3f ms  %.1f GB/s\n", blocks, ms, G * 52.0 / ms / 1e6);
        }
    }
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0); copy16<<<65536, 256>>>((const u32x4*)a, (u32x4*)b, G); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep == 2) printf("copy 16B/lane: %.3f ms  %.1f GB/s (read+write)\n", ms, G * 32.0 / ms / 1e6);
    }
    hipEventRecord(e0); hipMemsetAsync(a, 0, G * 16, 0); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("hipMemset %.3f ms %.1f GB/s\n", ms, G * 16.0 / ms / 1e6);
    return 0;
}

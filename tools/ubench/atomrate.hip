// Throughput of returning atomic adds on scattered words of a 40 MB array (what k_deps does 17.7 M times per build), by scope:
// agent scope (what a counter shared by all XCDs needs) against workgroup scope (performed in the issuing XCD's L2 — only right
// when every atomic on a word comes from one XCD), full chip, 256 threads x 4096 workgroups, 10 M atomics.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int SCOPE, int RET> __global__ void k(unsigned* a, unsigned n_words, unsigned per_thread, unsigned* sink, unsigned window) {
    unsigned x = (blockIdx.x * blockDim.x + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    const unsigned base = window ? (blockIdx.x & 7u) * (n_words / 8u) : 0u, span = window ? n_words / 8u : n_words;
    for (unsigned i = 0; i < per_thread; ++i) {
        x = x * 1664525u + 1013904223u;
        unsigned* p = a + base + (x >> 7) % span;
        if (RET) acc += __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, SCOPE);
        else __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, SCOPE);
    }
    if (acc == 0xFFFFFFFFu) *sink = acc;
}
template <int SCOPE, int RET> void run(const char* name, unsigned* a, unsigned n_words, unsigned* sink, unsigned window) {
    const unsigned grid = 4096, per = 10;
    hipMemset(a, 0, (size_t)n_words * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<SCOPE, RET>), dim3(grid), dim3(256), 0, 0, a, n_words, per, sink, window);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<SCOPE, RET>), dim3(grid), dim3(256), 0, 0, a, n_words, per, sink, window);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-58s %6.3f ms for %.1f M atomics = %5.1f G/s\n", name, ms, grid * 256.0 * per / 1e6, grid * 256.0 * per / ms / 1e6);
}
int main() {
    unsigned n_words = 10u << 20; unsigned* a; hipMalloc(&a, (size_t)n_words * 4); unsigned* sink; hipMalloc(&sink, 4);
    run<__HIP_MEMORY_SCOPE_AGENT, 1>("agent scope, returning, whole array", a, n_words, sink, 0);
    run<__HIP_MEMORY_SCOPE_AGENT, 0>("agent scope, not returning, whole array", a, n_words, sink, 0);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, 1>("workgroup scope, returning, whole array", a, n_words, sink, 0);
    run<__HIP_MEMORY_SCOPE_AGENT, 1>("agent scope, returning, an eighth of the array per XCD", a, n_words, sink, 1);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, 1>("workgroup scope, returning, an eighth of the array per XCD", a, n_words, sink, 1);
    run<__HIP_MEMORY_SCOPE_WORKGROUP, 0>("workgroup scope, not returning, an eighth per XCD", a, n_words, sink, 1);
    return 0;
}

// Does a kernel boundary cost instruction-cache misses?  One wave runs ~16 KB of straight-line ALU code three times
// inside a launch (pass 0 = cold I$, passes 1-2 = warm), and the launch is repeated back to back.
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned long long now(unsigned& dep) {
    unsigned long long t;
    asm volatile("s_nop 0\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "+v"(dep) :: "memory");
    return t;
}
template <int N> __device__ __forceinline__ unsigned body(unsigned x) {
#pragma unroll
    for (int i = 0; i < N; ++i) { x = x * 1664525u + (unsigned)(1013904223u + i * 77u); x ^= x >> 13; }
    return x;
}
__global__ void k(unsigned long long* out, unsigned seed, int passes) {
    unsigned x = seed + threadIdx.x;
    for (int p = 0; p < passes; ++p) {
        unsigned long long c0 = now(x);
        x = body<1000>(x);
        unsigned long long c1 = now(x);
        if (threadIdx.x == 0 && blockIdx.x == 0) out[p] = c1 - c0;
    }
    if (x == 12345) out[7] = x;
}
int main() {
    unsigned long long* out; hipMalloc(&out, 64); unsigned long long h[8];
    for (int rep = 0; rep < 4; ++rep) {
        k<<<1, 64>>>(out, rep, 3);
        if (rep == 2) k<<<1, 64>>>(out, rep, 3);      // back to back, no host sync between
        hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("launch %d: pass0 %.2f us, pass1 %.2f us, pass2 %.2f us\n", rep, h[0] / 100.0, h[1] / 100.0, h[2] / 100.0);
    }
    k<<<256, 64>>>(out, 9, 3); hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
    printf("256 WGs: pass0 %.2f us, pass1 %.2f us, pass2 %.2f us\n", h[0] / 100.0, h[1] / 100.0, h[2] / 100.0);
    return 0;
}

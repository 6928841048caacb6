// Calibration of the HBM byte model: KNOWN numbers of accesses, in the access patterns the build's kernels have,
// over a 2 GB array (eight times the 256 MB Infinity Cache), one kernel per pattern — run under rocprofv3 --pmc (tools/pmc_calibrate.sh)
// to see what FETCH_SIZE / WRITE_SIZE / TCC_EA0_RDREQ[_32B] / TCC_EA0_WRREQ[_64B] report PER ACCESS for each.  The guide calibrates
// FETCH_SIZE for wide coalesced streams only (x 2 on gfx950); everything scattered is calibrated here.
//   stream_read16 / stream_write16   coalesced 16 B per lane (the guide's reference point)
//   gather4 / gather8 / gather16     one random aligned word of 4 / 8 / 16 bytes per lane
//   gather_line512                   a wave reads 512 contiguous bytes at a random 512-aligned place (k_peel's node records)
//   scatter4 / scatter16             one random aligned store per lane;  scatter_line512: a wave writes 512 contiguous bytes
//   atomic_ret4                      one random returning atomic add per lane (k_deps' tickets)
// Every kernel: 4096 x 256 threads x PER accesses = N known exactly; addresses from an LCG (no reuse beyond chance).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef unsigned u32; typedef unsigned long long u64;
constexpr u32 GRID = 4096, BLOCK = 256, PER = 16;
__device__ __forceinline__ u32 lcg(u32& x) { x = x * 1664525u + 1013904223u; return x; }
__global__ void stream_read16(const uint4* a, u64 n16, u32* sink) {
    u32 acc = 0;
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n16; i += (u64)GRID * BLOCK) { const uint4 v = a[i]; acc += v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void stream_write16(uint4* a, u64 n16) {
    for (u64 i = (u64)blockIdx.x * BLOCK + threadIdx.x; i < n16; i += (u64)GRID * BLOCK) a[i] = make_uint4((u32)i, 1u, 2u, 3u);
}
template <class T> __global__ void gather(const T* a, u64 n_elems, u32* sink) {
    u32 x = (blockIdx.x * BLOCK + threadIdx.x) * 2654435761u + 12345u, acc = 0;
    for (u32 i = 0; i < PER; ++i) { const u64 r = ((u64)lcg(x) << 16) ^ lcg(x); const T v = a[r % n_elems]; acc += ((const u32*)&v)[0]; }
    if (acc == 0x12345678u) *sink = acc;
}
template <class T> __global__ void scatter(T* a, u64 n_elems) {
    u32 x = (blockIdx.x * BLOCK + threadIdx.x) * 2654435761u + 54321u;
    for (u32 i = 0; i < PER; ++i) { const u64 r = ((u64)lcg(x) << 16) ^ lcg(x); T v; memset(&v, 0, sizeof(T)); ((u32*)&v)[0] = x; a[r % n_elems] = v; }
}
__global__ void gather_line512(const u64* a, u64 n_lines, u32* sink) {      // one wave = one 512-byte line per access
    const u32 wave = (blockIdx.x * BLOCK + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    u32 x = wave * 2654435761u + 999u; u64 acc = 0;
    for (u32 i = 0; i < PER; ++i) { const u64 r = ((u64)lcg(x) << 16) ^ lcg(x); acc += a[(r % n_lines) * 64 + lane]; }
    if (acc == 0x12345678ull) *sink = (u32)acc;
}
__global__ void scatter_line512(u64* a, u64 n_lines) {
    const u32 wave = (blockIdx.x * BLOCK + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    u32 x = wave * 2654435761u + 777u;
    for (u32 i = 0; i < PER; ++i) { const u64 r = ((u64)lcg(x) << 16) ^ lcg(x); a[(r % n_lines) * 64 + lane] = x + lane; }
}
__global__ void atomic_ret4(u32* a, u64 n_elems, u32* sink) {
    u32 x = (blockIdx.x * BLOCK + threadIdx.x) * 2654435761u + 4242u, acc = 0;
    for (u32 i = 0; i < PER; ++i) { const u64 r = ((u64)lcg(x) << 16) ^ lcg(x); acc += atomicAdd(&a[r % n_elems], 1u); }
    if (acc == 0x12345678u) *sink = acc;
}
int main(int argc, char** argv) {
    const u64 bytes = 2ull << 30;
    void* a; if (hipMalloc(&a, bytes) != hipSuccess) { printf("hipMalloc failed\n"); return 1; }
    u32* sink; hipMalloc(&sink, 4);
    hipMemset(a, 0, bytes);
    const u64 N = (u64)GRID * BLOCK * PER;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
#define RUN(label, known, unit_bytes, ...) do { hipEventRecord(e0); __VA_ARGS__; hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); \
    printf("%-18s %12llu accesses x %3u B = %8.1f MB useful, %7.3f ms\n", label, (u64)(known), (u32)(unit_bytes), (double)(known) * (unit_bytes) / 1e6, ms); } while (0)
    RUN("stream_read16", bytes / 16, 16, hipLaunchKernelGGL(stream_read16, dim3(GRID), dim3(BLOCK), 0, 0, (const uint4*)a, bytes / 16, sink));
    RUN("stream_write16", bytes / 16, 16, hipLaunchKernelGGL(stream_write16, dim3(GRID), dim3(BLOCK), 0, 0, (uint4*)a, bytes / 16));
    RUN("gather4", N, 4, hipLaunchKernelGGL(gather<u32>, dim3(GRID), dim3(BLOCK), 0, 0, (const u32*)a, bytes / 4, sink));
    RUN("gather8", N, 8, hipLaunchKernelGGL(gather<u64>, dim3(GRID), dim3(BLOCK), 0, 0, (const u64*)a, bytes / 8, sink));
    RUN("gather16", N, 16, hipLaunchKernelGGL(gather<uint4>, dim3(GRID), dim3(BLOCK), 0, 0, (const uint4*)a, bytes / 16, sink));
    RUN("gather_line512", N / 64, 512, hipLaunchKernelGGL(gather_line512, dim3(GRID), dim3(BLOCK), 0, 0, (const u64*)a, bytes / 512, sink));
    RUN("scatter4", N, 4, hipLaunchKernelGGL(scatter<u32>, dim3(GRID), dim3(BLOCK), 0, 0, (u32*)a, bytes / 4));
    RUN("scatter16", N, 16, hipLaunchKernelGGL(scatter<uint4>, dim3(GRID), dim3(BLOCK), 0, 0, (uint4*)a, bytes / 16));
    RUN("scatter_line512", N / 64, 512, hipLaunchKernelGGL(scatter_line512, dim3(GRID), dim3(BLOCK), 0, 0, (u64*)a, bytes / 512));
    RUN("atomic_ret4", N, 4, hipLaunchKernelGGL(atomic_ret4, dim3(GRID), dim3(BLOCK), 0, 0, (u32*)a, bytes / 4, sink));
    hipDeviceSynchronize();
    return 0;
}

// What does a dependent hop cost, and why?  (single lane unless stated)
//  A. 1 MB ring: first pass (cold), second pass in the same launch (L2 hit), then the same ring in the NEXT launch
//     (does a kernel boundary drop clean lines from the L2?) and after another kernel WROTE the ring (dirty lines)
//  B. page-spread ring: 4096 nodes, each in its own 2 MB page over 8 GB; second pass = cache hit + TLB miss
//  C. loaded: every wave of a 256 x 1024 launch chases its own 64-node ring (lines written by the previous launch)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdint>

__global__ void chase(const unsigned* ring, unsigned start, int hops, int passes, unsigned long long* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned p = start;
    for (int k = 0; k < passes; ++k) {
        unsigned long long c0 = wall_clock64();
        for (int i = 0; i < hops; ++i) p = ring[p];
        unsigned long long c1 = wall_clock64();
        out[k] = c1 - c0;
    }
    out[7] = p;
}
__global__ void chase64(const unsigned long long* ring, unsigned long long start, int hops, int passes, unsigned long long* out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    unsigned long long p = start;
    for (int k = 0; k < passes; ++k) {
        unsigned long long c0 = wall_clock64();
        for (int i = 0; i < hops; ++i) p = ring[p];
        unsigned long long c1 = wall_clock64();
        out[k] = c1 - c0;
    }
    out[7] = p;
}
__global__ void rewrite(unsigned* ring, size_t words) {     // rewrites every word with its own value (dirties the lines)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) ring[i] = ring[i] + 0;
}
__global__ void rewrite_volatile(volatile unsigned* ring, size_t words) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) { unsigned v = ring[i]; ring[i] = v; }
}
// loaded chase: wave w follows next[] through its own 64 nodes (node = 128-byte line, lane 0 reads, then all lanes use it)
__global__ void __launch_bounds__(1024) loaded(const unsigned* ring, int hops, unsigned long long* out, int sc1) {
    const unsigned wave = (blockIdx.x * 1024 + threadIdx.x) >> 6;
    unsigned p = wave * 64 * 32;
    unsigned long long c0 = wall_clock64();
    if (sc1) for (int i = 0; i < hops; ++i) p = __hip_atomic_load(&ring[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else for (int i = 0; i < hops; ++i) p = ring[p];
    unsigned long long c1 = wall_clock64();
    if ((threadIdx.x & 63) == 0) out[wave] = (c1 - c0) + (p == 0xFFFFFFFFu);
}

static std::vector<unsigned> perm_of(size_t n, unsigned seed) {
    std::vector<unsigned> perm(n); for (size_t i = 0; i < n; ++i) perm[i] = (unsigned)i;
    unsigned s = seed; for (size_t i = n - 1; i > 0; --i) { s = s * 1664525u + 1013904223u; size_t j = s % (i + 1); std::swap(perm[i], perm[j]); }
    return perm;
}

int main() {
    unsigned long long* out; hipMalloc(&out, 1 << 20); unsigned long long h[8];
    {   // A
        const size_t lines = 8192, stride = 32;
        std::vector<unsigned> hb(lines * stride, 0); auto perm = perm_of(lines, 12345);
        for (size_t i = 0; i < lines; ++i) hb[(size_t)perm[i] * stride] = perm[(i + 1) % lines] * stride;
        unsigned* ring; hipMalloc(&ring, hb.size() * 4); hipMemcpy(ring, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
        chase<<<1, 64>>>(ring, 0, 8192, 2, out); hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("A 1MB ring: cold %.0f ns/hop, same launch again %.0f", h[0] * 10.0 / 8192, h[1] * 10.0 / 8192);
        chase<<<1, 64>>>(ring, 0, 8192, 1, out); hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf(", next launch %.0f", h[0] * 10.0 / 8192);
        chase<<<1, 64>>>(ring, 0, 8192, 1, out); chase<<<1, 64>>>(ring, 0, 8192, 1, out); hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf(", back-to-back launch %.0f", h[0] * 10.0 / 8192);
        rewrite_volatile<<<256, 256>>>(ring, hb.size()); chase<<<1, 64>>>(ring, 0, 8192, 2, out); hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf(", after a 256-WG launch rewrote it %.0f then %.0f\n", h[0] * 10.0 / 8192, h[1] * 10.0 / 8192);
        hipFree(ring);
    }
    {   // B
        const size_t nodes = 4096, page_words = (2u << 20) / 8;       // u64 ring, one node per 2 MB page
        unsigned long long* ring; hipMalloc(&ring, nodes * (2u << 20)); hipMemset(ring, 0, nodes * (2u << 20));
        auto perm = perm_of(nodes, 777);
        std::vector<unsigned long long> idx(nodes);
        for (size_t i = 0; i < nodes; ++i) {
            const unsigned long long from = (unsigned long long)perm[i] * page_words + (perm[i] % 128) * 16, to = (unsigned long long)perm[(i + 1) % nodes] * page_words + (perm[(i + 1) % nodes] % 128) * 16;
            hipMemcpy(ring + from, &to, 8, hipMemcpyHostToDevice);
        }
        const unsigned long long start = (unsigned long long)perm[0] * page_words + (perm[0] % 128) * 16;
        chase64<<<1, 64>>>(ring, start, 4096, 3, out); hipMemcpy(h, out, 64, hipMemcpyDeviceToHost);
        printf("B one node per 2MB page over 8 GB: cold %.0f ns/hop, again %.0f, again %.0f (cache hit + TLB miss)\n", h[0] * 10.0 / 4096, h[1] * 10.0 / 4096, h[2] * 10.0 / 4096);
        hipFree(ring);
    }
    {   // C
        const size_t waves = 256 * 16, nodes = waves * 64, stride = 32;
        std::vector<unsigned> hb(nodes * stride, 0);
        for (size_t w = 0; w < waves; ++w) { auto perm = perm_of(64, (unsigned)w + 1); for (size_t i = 0; i < 64; ++i) hb[(w * 64 + perm[i]) * stride] = (unsigned)((w * 64 + perm[(i + 1) % 64]) * stride); }
        unsigned* ring; hipMalloc(&ring, hb.size() * 4); hipMemcpy(ring, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
        std::vector<unsigned long long> ho(waves);
        for (int sc1 = 0; sc1 < 2; ++sc1) for (int grid : {256, 32, 1}) {
            rewrite_volatile<<<256, 256>>>(ring, hb.size());
            loaded<<<grid, 1024>>>(ring, 64, out, sc1); hipMemcpy(ho.data(), out, waves * 8, hipMemcpyDeviceToHost);
            double sum = 0, mx = 0; const size_t nw = (size_t)grid * 16; for (size_t w = 0; w < nw; ++w) { sum += ho[w]; mx = std::max<double>(mx, ho[w]); }
            printf("C loaded %s, %3d WGs x 16 waves, 64 hops over freshly rewritten lines: mean %.0f ns/hop, slowest wave %.0f\n", sc1 ? "sc1  " : "plain", grid, sum / nw * 10 / 64, mx * 10 / 64);
            loaded<<<grid, 1024>>>(ring, 64, out, sc1); hipMemcpy(ho.data(), out, waves * 8, hipMemcpyDeviceToHost);
            sum = 0; mx = 0; for (size_t w = 0; w < nw; ++w) { sum += ho[w]; mx = std::max<double>(mx, ho[w]); }
            printf("         again (next launch, lines clean):                                   mean %.0f ns/hop, slowest wave %.0f\n", sum / nw * 10 / 64, mx * 10 / 64);
        }
    }
    return 0;
}

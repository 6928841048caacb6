// Can a persistent kernel confine itself to ONE XCD (shared, coherent L2) and run a cheap barrier loop?
//  - HW_REG_XCC_ID readout, census of workgroups per XCD
//  - barrier among the workgroups of XCD 0 (L2 atomics, no fences), verified producer->consumer hand-off with
//    plain stores + nt (L1-bypassing) loads, cost per barrier iteration
//  - dependent-hop latency of nt loads on data written by another CU of the same XCD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
    // s_getreg_b32 hwreg(HW_REG_XCC_ID = 20, offset 0, size 4)
    return __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 0xF;
}

struct Ctl { unsigned joined, bystanders, P, arrive, gen, errors, census[8], pad[16]; };

__device__ __forceinline__ unsigned ld_nt(const unsigned* p) { return __builtin_nontemporal_load(p); }

template <int MODE>   // 0: plain store + nt load, 1: plain store + sc1 (relaxed agent) load, 2: sc1 store + sc1 load
__global__ void __launch_bounds__(256) persistent(Ctl* ctl, unsigned* slots, unsigned* chain, int iters, unsigned long long* t_out) {
    __shared__ unsigned s_P, s_rank;
    const unsigned x = xcc_id();
    if (threadIdx.x == 0) {
        atomicAdd(&ctl->census[x & 7], 1u);
        if (x == 0) s_rank = atomicAdd(&ctl->joined, 1u); else atomicAdd(&ctl->bystanders, 1u);
    }
    __syncthreads();
    if (x != 0) return;
    if (threadIdx.x == 0) {
        while (__hip_atomic_load(&ctl->joined, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) +
               __hip_atomic_load(&ctl->bystanders, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) __builtin_amdgcn_s_sleep(1);
        s_P = __hip_atomic_load(&ctl->joined, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    const unsigned P = s_P, rank = s_rank;
    unsigned long long t0 = wall_clock64();
    unsigned errors = 0;
    for (int it = 1; it <= iters; ++it) {
        // produce: every thread writes one word of this workgroup's 1 KB record
        if (MODE == 2) __hip_atomic_store(&slots[rank * 256 + threadIdx.x], (unsigned)it * 1000u + rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else slots[rank * 256 + threadIdx.x] = (unsigned)it * 1000u + rank;
        __builtin_amdgcn_s_waitcnt(0);           // stores issued... (vmcnt(0))
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // barrier: one counter + generation, relaxed agent atomics, no fences
        if (threadIdx.x == 0) {
            const unsigned a = atomicAdd(&ctl->arrive, 1u);
            if (a == P * (unsigned)it - 1) __hip_atomic_store(&ctl->gen, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(&ctl->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)it) {}
        }
        __syncthreads();
        // consume the next workgroup's record with L1-bypassing loads
        const unsigned nb = (rank + 1) % P;
        const unsigned v = MODE == 0 ? ld_nt(&slots[nb * 256 + threadIdx.x]) : __hip_atomic_load(&slots[nb * 256 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v != (unsigned)it * 1000u + nb) { ++errors; if (v == (unsigned)(it - 1) * 1000u + nb) atomicAdd(&ctl->pad[0], 1u); else if (v == 0) atomicAdd(&ctl->pad[1], 1u); else atomicAdd(&ctl->pad[2], 1u); if (it <= 4) atomicAdd(&ctl->pad[3], 1u); }
    }
    unsigned long long t1 = wall_clock64();
    if (errors) atomicAdd(&ctl->errors, errors);
    if (rank == 0 && threadIdx.x == 0) { t_out[0] = t1 - t0; ctl->P = P; }
    // dependent nt-load chase over a buffer written (just now) by workgroup rank+1's threads? use prebuilt chain
    if (rank == 0 && threadIdx.x == 0) {
        unsigned long long c0 = wall_clock64();
        unsigned p = 0;
        for (int i = 0; i < 4096; ++i) p = ld_nt(&chain[p]);
        unsigned long long c1 = wall_clock64();
        t_out[1] = c1 - c0; t_out[2] = p;
    }
}

int main() {
    Ctl* ctl; hipMalloc(&ctl, sizeof(Ctl)); hipMemset(ctl, 0, sizeof(Ctl));
    unsigned* slots; hipMalloc(&slots, 256 * 256 * 4); hipMemset(slots, 0, 256 * 256 * 4);
    // chain: random cycle over 64 MB of 64-byte lines
    size_t n = 1 << 20; std::vector<unsigned> h(n * 16, 0); std::vector<unsigned> perm(n);
    for (size_t i = 0; i < n; ++i) perm[i] = (unsigned)i;
    unsigned s = 12345; for (size_t i = n - 1; i > 0; --i) { s = s * 1664525u + 1013904223u; size_t j = s % (i + 1); std::swap(perm[i], perm[j]); }
    for (size_t i = 0; i < n; ++i) h[(size_t)perm[i] * 16] = perm[(i + 1) % n] * 16;
    unsigned* chain; hipMalloc(&chain, n * 64); hipMemcpy(chain, h.data(), n * 64, hipMemcpyHostToDevice);
    unsigned long long* t; hipMalloc(&t, 64); hipMemset(t, 0, 64);
    for (int run = 0; run < 9; ++run) {
        const int iters = 2000, mode = run % 3;
        hipMemset(ctl, 0, sizeof(Ctl));
        if (run >= 3) { hipFree(slots); hipMalloc(&slots, 256 * 256 * 4); if (run < 6) hipMemset(slots, 0, 256 * 256 * 4); }
        if (mode == 0) persistent<0><<<256, 256>>>(ctl, slots, chain, iters, t);
        else if (mode == 1) persistent<1><<<256, 256>>>(ctl, slots, chain, iters, t);
        else persistent<2><<<256, 256>>>(ctl, slots, chain, iters, t);
        hipError_t e = hipDeviceSynchronize();
        Ctl hc; unsigned long long ht[3]; hipMemcpy(&hc, ctl, sizeof(Ctl), hipMemcpyDeviceToHost); hipMemcpy(ht, t, 24, hipMemcpyDeviceToHost);
        printf("%s census:", hipGetErrorString(e)); for (int k = 0; k < 8; ++k) printf(" %u", hc.census[k]);
        printf(" | mode %d P=%u errors=%u (prev-iter %u, zero %u, other %u, in first 4 iters %u) | %.2f us/iter | nt chase %.1f ns/hop\n", mode, hc.P, hc.errors,
               hc.pad[0], hc.pad[1], hc.pad[2], hc.pad[3], ht[0] * 10.0 / iters / 1000.0, ht[1] * 10.0 / 4096);
    }
    return 0;
}

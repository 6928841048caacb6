// Single-XCD persistent execution, second look: what do L2-level (not memory-side) exchanges cost?
//  T1 barrier among the P workgroups of one XCD: arrive = atomic add, wait = (a) sc1 load poll, (b) RMW poll, (c) plain-load poll after buffer_inv
//  T2 dependent hop latency of one lane: 1 MB ring (L2 resident) and 256 MB ring (memory), plain / sc1 / nt loads
//  T3 hand-off: plain store -> vmcnt(0) -> barrier -> buffer_inv sc1 -> plain load, verified
//  T4 same-address atomic throughput: P workgroups x 1 lane x 256 returning atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 0xF; }
struct Ctl { unsigned joined, bystanders, P, arrive, gen, errors, hot, pad[25]; };
#define LD_SC1(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ST_SC1(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
__device__ __forceinline__ void inv_l1() { asm volatile("buffer_inv sc1" ::: "memory"); }

template <int POLL>
__device__ __forceinline__ void barrier(Ctl* ctl, unsigned P, unsigned epoch) {
    if (threadIdx.x == 0) {
        const unsigned a = atomicAdd(&ctl->arrive, 1u);
        if (a == P * epoch - 1) ST_SC1(&ctl->gen, epoch);
        unsigned spins = 0;
        if (POLL == 0) { while (LD_SC1(&ctl->gen) < epoch && ++spins < (1u << 22)) {} }
        else if (POLL == 1) { while (atomicOr(&ctl->gen, 0u) < epoch && ++spins < (1u << 22)) {} }
        else { while (++spins < (1u << 22)) { inv_l1(); if (*(volatile unsigned*)&ctl->gen >= epoch) break; } }
        if (spins >= (1u << 22)) atomicAdd(&ctl->pad[0], 1u);
    }
    __syncthreads();
}

template <int POLL, bool ONE_XCD>
__global__ void __launch_bounds__(256) k(Ctl* ctl, unsigned* slots, unsigned* ring_small, unsigned* ring_big, int iters, unsigned long long* t_out) {
    __shared__ unsigned s_P, s_rank;
    const unsigned x = xcc_id();
    const bool mine = !ONE_XCD || x == 0;
    if (threadIdx.x == 0) { if (mine) s_rank = atomicAdd(&ctl->joined, 1u); else atomicAdd(&ctl->bystanders, 1u); }
    __syncthreads();
    if (!mine) return;
    if (threadIdx.x == 0) {
        while (LD_SC1(&ctl->joined) + LD_SC1(&ctl->bystanders) < gridDim.x) __builtin_amdgcn_s_sleep(1);
        s_P = LD_SC1(&ctl->joined);
    }
    __syncthreads();
    const unsigned P = s_P, rank = s_rank;
    unsigned epoch = 0;
    // T1
    unsigned long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) barrier<POLL>(ctl, P, ++epoch);
    unsigned long long t1 = wall_clock64();
    // T3: hand-off flavours, one barrier per iteration (records double-buffered by iteration parity)
    //   0: plain store -> sc1 load   1: sc1 store -> sc1 load   2: plain store -> buffer_inv sc1 -> plain load
    unsigned long long th[4];
    unsigned errs[3] = {0, 0, 0};
    th[0] = wall_clock64();
    for (int flavour = 0; flavour < 3; ++flavour) {
        for (int it = 1; it <= iters; ++it) {
            unsigned* rec = slots + (it & 1) * 65536 + flavour * 131072;
            const unsigned val = (unsigned)it * 1000u + rank;
            if (flavour == 1) ST_SC1(&rec[rank * 256 + threadIdx.x], val); else rec[rank * 256 + threadIdx.x] = val;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            barrier<POLL>(ctl, P, ++epoch);
            const unsigned nb = (rank + 1) % P;
            unsigned v;
            if (flavour == 2) { inv_l1(); v = *(volatile unsigned*)&rec[nb * 256 + threadIdx.x]; }
            else v = LD_SC1(&rec[nb * 256 + threadIdx.x]);
            if (v != (unsigned)it * 1000u + nb) ++errs[flavour];
        }
        th[flavour + 1] = wall_clock64();
    }
    unsigned long long t2 = th[3];
    for (int f = 0; f < 3; ++f) if (errs[f]) atomicAdd(&ctl->pad[1 + f], errs[f]);
    // T5: every workgroup streams 160 KB of a 5 MB region (written by the host long ago) with sc1 loads
    barrier<POLL>(ctl, P, ++epoch);
    unsigned long long t5a = wall_clock64();
    unsigned sum5 = 0;
    for (int rep = 0; rep < 4; ++rep) {
        const uint4* src = (const uint4*)ring_big + (size_t)rank * 10240 + (size_t)rep * 40;
        for (int i = threadIdx.x; i < 10240; i += 256) {
            const unsigned long long lo = __hip_atomic_load((const unsigned long long*)&src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned long long hi = __hip_atomic_load((const unsigned long long*)&src[i] + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sum5 += (unsigned)(lo ^ hi);
        }
        barrier<POLL>(ctl, P, ++epoch);
    }
    unsigned long long t5b = wall_clock64();
    if (sum5 == 12345) atomicAdd(&ctl->pad[5], 1u);
    if (rank == 0 && threadIdx.x == 0) { t_out[9] = th[1] - th[0]; t_out[10] = th[2] - th[1]; t_out[11] = th[3] - th[2]; t_out[12] = t5b - t5a; }
    // T4
    barrier<POLL>(ctl, P, ++epoch);
    unsigned long long t3 = wall_clock64();
    unsigned acc = 0;
    if (threadIdx.x == 0) for (int i = 0; i < 256; ++i) acc += atomicAdd(&ctl->hot, 1u);
    barrier<POLL>(ctl, P, ++epoch);
    unsigned long long t4 = wall_clock64();
    if (rank == 0 && threadIdx.x == 0) {
        t_out[0] = t1 - t0; t_out[1] = t2 - t1; t_out[2] = t4 - t3; ctl->P = P; t_out[15] = acc;
        // T2
        for (int kind = 0; kind < 6; ++kind) {
            unsigned* ring = kind < 3 ? ring_small : ring_big;
            unsigned p = ring[kind * 64 % 8] ;
            for (int i = 0; i < kind * 37; ++i) p = ring[p];
            if (kind < 3) for (int i = 0; i < 8192; ++i) p = ring[p];           // warm L2
            unsigned long long c0 = wall_clock64();
            for (int i = 0; i < 4096; ++i) {
                if (kind % 3 == 0) { p = ring[p]; }
                else if (kind % 3 == 1) p = LD_SC1(&ring[p]);
                else p = __builtin_nontemporal_load(&ring[p]);
            }
            unsigned long long c1 = wall_clock64();
            t_out[3 + kind] = c1 - c0; t_out[14] += p;
        }
    }
}

static unsigned* make_ring(size_t lines, unsigned stride_words) {
    std::vector<unsigned> h(lines * stride_words, 0), perm(lines);
    for (size_t i = 0; i < lines; ++i) perm[i] = (unsigned)i;
    unsigned s = 12345; for (size_t i = lines - 1; i > 0; --i) { s = s * 1664525u + 1013904223u; size_t j = s % (i + 1); std::swap(perm[i], perm[j]); }
    for (size_t i = 0; i < lines; ++i) h[(size_t)perm[i] * stride_words] = perm[(i + 1) % lines] * stride_words;
    unsigned* d; hipMalloc(&d, h.size() * 4); hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    return d;
}

template <int POLL, bool ONE>
static void run(const char* name, Ctl* ctl, unsigned* slots, unsigned* rs, unsigned* rb, unsigned long long* t) {
    const int iters = 2000;
    hipMemset(ctl, 0, sizeof(Ctl)); hipMemset(t, 0, 128);
    k<POLL, ONE><<<256, 256>>>(ctl, slots, rs, rb, iters, t);
    hipError_t e = hipDeviceSynchronize();
    Ctl hc; unsigned long long ht[16]; hipMemcpy(&hc, ctl, sizeof(Ctl), hipMemcpyDeviceToHost); hipMemcpy(ht, t, 128, hipMemcpyDeviceToHost);
    printf("%-28s %s P=%3u | barrier %.2f us | hand-off us/iter (errors): plain->sc1 %.2f (%u)  sc1->sc1 %.2f (%u)  plain->inv->plain %.2f (%u); timeouts %u | 4 x (P x 160 KB sc1 stream + barrier) %.2f us each | 256 hot atomics x P: %.1f ns each | hop ns small plain/sc1/nt %.0f %.0f %.0f  big %.0f %.0f %.0f\n",
           name, hipGetErrorString(e), hc.P, ht[0] * 10.0 / iters / 1000, ht[9] * 10.0 / iters / 1000, hc.pad[1], ht[10] * 10.0 / iters / 1000, hc.pad[2], ht[11] * 10.0 / iters / 1000, hc.pad[3], hc.pad[0], ht[12] * 10.0 / 4 / 1000, ht[2] * 10.0 / (256.0 * hc.P),
           ht[3] * 10.0 / 4096, ht[4] * 10.0 / 4096, ht[5] * 10.0 / 4096, ht[6] * 10.0 / 4096, ht[7] * 10.0 / 4096, ht[8] * 10.0 / 4096);
}

int main() {
    Ctl* ctl; hipMalloc(&ctl, sizeof(Ctl));
    unsigned* slots; hipMalloc(&slots, 3 * 131072 * 4); hipMemset(slots, 0, 3 * 131072 * 4);
    unsigned* rs = make_ring(1 << 13, 32);      // 8192 lines x 128 B = 1 MB
    unsigned* rb = make_ring(1 << 21, 32);      // 256 MB
    unsigned long long* t; hipMalloc(&t, 128);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, true>("one XCD, sc1-load poll", ctl, slots, rs, rb, t);
        run<1, true>("one XCD, RMW poll", ctl, slots, rs, rb, t);
        run<2, true>("one XCD, inv+plain poll", ctl, slots, rs, rb, t);
        run<0, false>("all XCDs, sc1-load poll", ctl, slots, rs, rb, t);
        run<1, false>("all XCDs, RMW poll", ctl, slots, rs, rb, t);
    }
    return 0;
}

// Ping-pong between two waves of the SAME XCD (different CUs) through one line, by memory scope of the accesses:
//   agent scope (sc1: what the dataflow launch uses everywhere — correct across XCDs) against workgroup scope (sc0: the
//   XCD's own L2 is the meeting point; only correct when every party is on that XCD).  Also the latency of a returning
//   atomic add at either scope.  Prints the XCD ids so that the pairing can be checked.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 0xF; }

template <int SCOPE, int SHIFT, bool RMW>
__global__ void __launch_bounds__(64) k_pp(unsigned* flags, unsigned* xcc, int iters, unsigned long long* t_out, unsigned* errs) {
    // wg and wg + 8 land on the same XCD (round-robin dispatch); SHIFT rotates the partner to another XCD
    const unsigned wg = blockIdx.x, side = wg >> 3, pair = side == 0 ? (wg & 7u) : ((wg + SHIFT) & 7u);
    if (threadIdx.x == 0) xcc[wg] = xcc_id();
    unsigned* a = flags + pair * 64;            // ping writes a, pong writes b (separate 128-byte lines)
    unsigned* b = flags + pair * 64 + 32;
    unsigned long long t0 = wall_clock64();
    unsigned bad = 0;
    if (threadIdx.x == 0) {
        for (int it = 1; it <= iters; ++it) {
            if (side == 0) {
                if (RMW) atomicExch(a, (unsigned)it); else __hip_atomic_store(a, (unsigned)it, __ATOMIC_RELAXED, SCOPE);
                unsigned spins = 0;
                while ((RMW ? atomicAdd(b, 0u) : __hip_atomic_load(b, __ATOMIC_RELAXED, SCOPE)) != (unsigned)it && ++spins < (1u << 24)) {}
                if (spins >= (1u << 24)) { ++bad; break; }
            } else {
                unsigned spins = 0;
                while ((RMW ? atomicAdd(a, 0u) : __hip_atomic_load(a, __ATOMIC_RELAXED, SCOPE)) != (unsigned)it && ++spins < (1u << 24)) {}
                if (spins >= (1u << 24)) { ++bad; break; }
                if (RMW) atomicExch(b, (unsigned)it); else __hip_atomic_store(b, (unsigned)it, __ATOMIC_RELAXED, SCOPE);
            }
        }
    }
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { t_out[wg] = t1 - t0; if (bad) atomicAdd(errs, bad); }
}

template <int SCOPE>
__global__ void __launch_bounds__(64) k_rmw(unsigned* ctr, int iters, unsigned long long* t_out) {
    // one lane per workgroup, dependent returning atomics on the workgroup's own line
    unsigned* p = ctr + blockIdx.x * 32;
    unsigned long long t0 = wall_clock64();
    unsigned v = 0;
    if (threadIdx.x == 0) for (int it = 0; it < iters; ++it) v += __hip_atomic_fetch_add(p + (v & 0u), 1u, __ATOMIC_RELAXED, SCOPE);
    unsigned long long t1 = wall_clock64();
    if (threadIdx.x == 0) { t_out[blockIdx.x] = t1 - t0; if (v == 0xFFFFFFFFu) p[1] = v; }
}

int main() {
    const int iters = 20000;
    unsigned *flags, *xcc, *errs; unsigned long long* t;
    hipMalloc(&flags, 8 * 64 * 4); hipMalloc(&xcc, 16 * 4); hipMalloc(&errs, 4); hipMalloc(&t, 16 * 8);
    std::vector<unsigned> hx(16); std::vector<unsigned long long> ht(16);
    for (int variant = 0; variant < 6; ++variant) {
        hipMemset(flags, 0, 8 * 64 * 4); hipMemset(errs, 0, 4);
        const char* name = "";
        switch (variant) {
            case 0: name = "same XCD, sc1 store / sc1 load"; k_pp<__HIP_MEMORY_SCOPE_AGENT, 0, false><<<16, 64>>>(flags, xcc, iters, t, errs); break;
            case 1: name = "other XCD (+1), sc1 store / sc1 load"; k_pp<__HIP_MEMORY_SCOPE_AGENT, 1, false><<<16, 64>>>(flags, xcc, iters, t, errs); break;
            case 2: name = "other XCD (+4), sc1 store / sc1 load"; k_pp<__HIP_MEMORY_SCOPE_AGENT, 4, false><<<16, 64>>>(flags, xcc, iters, t, errs); break;
            case 3: name = "same XCD, atomic exchange / atomic add 0"; k_pp<__HIP_MEMORY_SCOPE_AGENT, 0, true><<<16, 64>>>(flags, xcc, iters, t, errs); break;
            case 4: name = "other XCD (+1), atomic exchange / atomic add 0"; k_pp<__HIP_MEMORY_SCOPE_AGENT, 1, true><<<16, 64>>>(flags, xcc, iters, t, errs); break;
            case 5: name = "same XCD, sc0 store / sc0 load (workgroup scope: NOT coherent between CUs)"; k_pp<__HIP_MEMORY_SCOPE_WORKGROUP, 0, false><<<16, 64>>>(flags, xcc, iters / 100, t, errs); break;
        }
        hipDeviceSynchronize();
        unsigned he = 0;
        hipMemcpy(hx.data(), xcc, 64, hipMemcpyDeviceToHost); hipMemcpy(ht.data(), t, 128, hipMemcpyDeviceToHost); hipMemcpy(&he, errs, 4, hipMemcpyDeviceToHost);
        double ns = 0; for (int i = 0; i < 8; ++i) ns += ht[i] * 10.0 / (variant == 5 ? iters / 100 : iters);
        printf("ping-pong %-75s %6.0f ns per round trip (2 hops), timeouts %u\n", name, ns / 8, he);
    }
    printf("xcc ids of workgroups 0..15:"); for (int i = 0; i < 16; ++i) printf(" %u", hx[i]); printf("\n");
    unsigned* ctr; hipMalloc(&ctr, 16 * 32 * 4);
    for (int scope = 0; scope < 2; ++scope) {
        hipMemset(ctr, 0, 16 * 32 * 4);
        if (scope == 0) k_rmw<__HIP_MEMORY_SCOPE_AGENT><<<16, 64>>>(ctr, iters, t); else k_rmw<__HIP_MEMORY_SCOPE_WORKGROUP><<<16, 64>>>(ctr, iters, t);
        hipDeviceSynchronize();
        hipMemcpy(ht.data(), t, 128, hipMemcpyDeviceToHost);
        double ns = 0; for (int i = 0; i < 16; ++i) ns += ht[i] * 10.0 / iters;
        printf("returning atomic add, %s scope: %.0f ns each\n", scope ? "workgroup" : "agent", ns / 16);
    }
    return 0;
}

// Scalar stores (s_store_dword / s_store_dwordx4: SMEM path, through the write-back scalar data cache) on gfx950 for
// one-lane stores of wave-uniform data: do they exist, does s_dcache_wb before the end of the wave make them visible to
// the next launch, do two waves on different CUs that write NEIGHBOURING words of one line both survive (dirty byte masks),
// and is nothing stale when the same words are rewritten launch after launch?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef __attribute__((address_space(1))) unsigned* G;
typedef unsigned v4u __attribute__((ext_vector_type(4)));

// wave w writes quad[w] = {w, round, w ^ round, ~w} with one s_store_dwordx4 and word[perm(w)] = w + round with one
// s_store_dword (perm pairs waves of different workgroups on neighbouring words)
__global__ void k_write(unsigned* quad, unsigned* word, unsigned n_waves, unsigned round, int wb) {
    const unsigned w = __builtin_amdgcn_readfirstlane((blockIdx.x * blockDim.x + threadIdx.x) >> 6);
    if (w >= n_waves) return;
    v4u q = {w, round, w ^ round, ~w};
    const unsigned qoff = w * 16u;
    const unsigned half = n_waves / 2, j = w < half ? 2 * w : 2 * (w - half) + 1;      // waves w and w + half share a pair of words
    const unsigned woff = __builtin_amdgcn_readfirstlane(j * 4u), wv = __builtin_amdgcn_readfirstlane(w + round);
    asm volatile("s_store_dwordx4 %0, %1, %2\n\ts_store_dword %3, %4, %5" :: "s"(q), "s"((G)quad), "s"(qoff), "s"(wv), "s"((G)word), "s"(woff) : "memory");
    if (wb) asm volatile("s_dcache_wb\n\ts_waitcnt lgkmcnt(0)" ::: "memory");
}
__global__ void k_check(const unsigned* quad, const unsigned* word, unsigned n_waves, unsigned round, unsigned* bad) {
    const unsigned w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_waves) return;
    const unsigned half = n_waves / 2, j = w < half ? 2 * w : 2 * (w - half) + 1;
    const bool ok = quad[4 * w] == w && quad[4 * w + 1] == round && quad[4 * w + 2] == (w ^ round) && quad[4 * w + 3] == ~w && word[j] == w + round;
    if (!ok) atomicAdd(bad, 1u);
}
int main() {
    const unsigned n_waves = 1u << 16;
    unsigned *quad, *word, *bad;
    hipMalloc(&quad, n_waves * 16); hipMalloc(&word, n_waves * 4); hipMalloc(&bad, 4);
    for (int wb = 1; wb >= 0; --wb) {
        unsigned total_bad = 0;
        for (unsigned round = 1; round <= 50; ++round) {
            if (round % 10 == 1) { hipMemset(quad, 0xFF, n_waves * 16); hipMemset(word, 0xFF, n_waves * 4); }      // (as the library does with child[])
            hipMemset(bad, 0, 4);
            k_write<<<n_waves / 4, 256>>>(quad, word, n_waves, round, wb);
            k_check<<<n_waves / 256, 256>>>(quad, word, n_waves, round, bad);
            unsigned b; hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost);
            total_bad += b;
        }
        if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
        printf("scalar stores %s s_dcache_wb: %u of %u wave-results wrong over 50 launches\n", wb ? "with" : "WITHOUT", total_bad, 50 * n_waves);
    }
    return 0;
}

// c2a_platform.h — the one place that decides between the real HIP runtime (product build, hipcc,
// gfx950) and the host emulation used by the CPU test-suite (tests/emul/hip_emul.h, -DC2A_EMULATE).
#pragma once
#include <cstdint>

#ifdef C2A_EMULATE
#include "hip_emul.h"
// kernels without __syncthreads()/wave intrinsics run as plain loops under emulation
#define C2A_LAUNCH_NOSYNC(kernel, grid, block, stream, ...) \
    hipemuLaunchNoSync(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
// kernels whose workgroups exchange data while they run: every block alive at once under emulation
#define C2A_LAUNCH_CONCURRENT(kernel, grid, block, stream, ...) \
    hipemuLaunchConcurrent(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
#else
#include <hip/hip_runtime.h>
#define C2A_LAUNCH_CONCURRENT(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
#define C2A_LAUNCH_NOSYNC(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
#endif
#define C2A_LAUNCH(kernel, grid, block, stream, ...) \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)

typedef uint8_t u8;
typedef uint32_t u32;
typedef uint64_t u64;
typedef int64_t i64;
typedef unsigned long long ull;

#define C2A_NONE 0xFFFFFFFFu

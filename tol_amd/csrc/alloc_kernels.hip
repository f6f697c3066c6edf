// alloc_kernels.hip -- the one device helper of the placed allocator (problem.cpp: device_alloc): how many 32-bit words of a
// freshly mapped block do NOT hold the pattern the allocator wrote there.  Why that is asked: problem.cpp, settle_block().
#include "kernels.h"

#include <hip/hip_runtime.h>

namespace tolfg {

__global__ __launch_bounds__(256) void count_not_kernel(const uint4 *__restrict__ p, size_t vecs, unsigned pattern, unsigned long long *count)
{
    unsigned long long mine = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < vecs; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        mine += (v.x != pattern) + (v.y != pattern) + (v.z != pattern) + (v.w != pattern);
    }
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if ((threadIdx.x & 63) == 0 && mine) atomicAdd(count, mine);
}

hipError_t launch_count_not(const void *p, size_t bytes, unsigned pattern, unsigned long long *count, hipStream_t s)
{
    const size_t vecs = bytes / 16;                  // blocks are whole chunks: a multiple of 16 bytes
    if (vecs == 0) return hipSuccess;
    size_t blocks = (vecs + 255) / 256;
    if (blocks > 2048) blocks = 2048;                // 8 workgroups per CU walk the block
    hipLaunchKernelGGL(count_not_kernel, dim3((unsigned)blocks), dim3(256), 0, s, static_cast<const uint4 *>(p), vecs, pattern, count);
    return hipGetLastError();
}

}  // namespace tolfg

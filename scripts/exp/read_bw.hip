// Experiment: pure streaming-read bandwidth (16-byte loads, UN in flight per lane), no writes except one word per block.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
template <int UN>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ x, long n16, uint32_t* __restrict__ out) {
    u32x4 acc = {0u, 0u, 0u, 0u};
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UN - 1) * stride < n16; i += UN * stride) {
        u32x4 v[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < UN; ++u) acc ^= v[u];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = 1;
}
extern "C" void launch_read(void* stream, const void* x, long bytes, void* out, int blocks, int un) {
    long n16 = bytes / 16;
    if (un == 1) hipLaunchKernelGGL((read_kernel<1>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x, n16, (uint32_t*)out);
    else if (un == 4) hipLaunchKernelGGL((read_kernel<4>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x, n16, (uint32_t*)out);
    else hipLaunchKernelGGL((read_kernel<8>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const u32x4*)x, n16, (uint32_t*)out);
}

// Experiment: can the kernel in front of a weight-streaming GEMM shorten that GEMM's ramp by touching the first bytes each of its blocks
// will stream?  touch_kernel: block i reads `bytes` of the 16 weight rows block i of gemm_skinny_dma_kernel owns (first K columns), with the
// default cache policy, into a checksum.  Block i of both kernels lands on XCD i % 8 (observed dispatch order), so the lines sit in the
// L2 the GEMM block will ask.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
extern "C" __global__ void touch_kernel(const uint16_t* __restrict__ W, long ldw, int N, int kcols, uint32_t* sink) {
    const int row0 = blockIdx.x * 16;
    uint32_t acc = 0;
    const int chunks = kcols / 8;                      // 16-byte chunks per row
    for (int idx = threadIdx.x; idx < 16 * chunks; idx += blockDim.x) {
        const int r = idx / chunks, c = idx % chunks;
        if (row0 + r < N) { u32x4 v = *reinterpret_cast<const u32x4*>(W + (long)(row0 + r) * ldw + c * 8); acc += v[0] ^ v[1] ^ v[2] ^ v[3]; }
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
extern "C" void launch_touch(void* stream, const void* W, long ldw, int N, int kcols, void* sink) {
    hipLaunchKernelGGL(touch_kernel, dim3((N + 15) / 16), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)W, ldw, N, kcols, (uint32_t*)sink);
}

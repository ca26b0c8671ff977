// Experiment: sustained bf16 MFMA rate from REGISTERS ONLY (no LDS, no memory) with the ring kernel's accumulator footprint (128 fp32 per lane,
// two waves per SIMD), for the two instruction shapes - v_mfma_f32_16x16x32_bf16 (8 x 4 tiles, what gemm_bt_ring_kernel issues) and
// v_mfma_f32_32x32x16_bf16 (4 x 2 tiles) - on random and on zero operands.  The prefill GEMM is power-limited on real data:
// does the instruction shape change the clock the chip sustains?   hipcc --offload-arch=gfx950 -O3 -shared -fPIC mfma_power.hip -o mfma_power.so
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
typedef __attribute__((__vector_size__(4 * sizeof(uint32_t)))) uint32_t u32x4;

__device__ __forceinline__ uint32_t rnd(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }
// a bf16 pair with exponents near 1.0 and random mantissas / signs (finite, no denormals): realistic toggling without overflow
__device__ __forceinline__ uint32_t rnd_bf2(uint32_t& s, int zero) {
    if (zero) return 0u;
    const uint32_t r = rnd(s);
    const uint32_t lo = 0x3c00u | (r & 0x83ffu), hi = 0x3c00u | ((r >> 16) & 0x83ffu);      // |x| in [2^-7, 2^-5) roughly, random sign
    return lo | (hi << 16);
}

template <int SHAPE>
__global__ __launch_bounds__(512) void mfma_kernel(float* sink, int iters, int zero) {
    uint32_t s = blockIdx.x * 7919u + threadIdx.x * 104729u + 1u;
    // two operand sets, alternated every k step, so that the matrix pipe sees changing inputs like a K loop does
    union { u32x4 r; bf16x8_t f; } a[2][8], b[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
        for (int i = 0; i < 8; ++i) a[u][i].r = u32x4{rnd_bf2(s, zero), rnd_bf2(s, zero), rnd_bf2(s, zero), rnd_bf2(s, zero)};
#pragma unroll
        for (int i = 0; i < 4; ++i) b[u][i].r = u32x4{rnd_bf2(s, zero), rnd_bf2(s, zero), rnd_bf2(s, zero), rnd_bf2(s, zero)};
    }
    if (SHAPE == 16) {
        f32x4_t acc[4][8];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[u][i].f, a[u][j].f, acc[i][j], 0, 0, 0);
        }
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if (t == 12345.678f) sink[0] = t;
    } else {
        // 32x32x16: the same 64 operand registers as four A (32 rows x 16 k) and two B fragments per k step, two k steps per set:
        // a[u][0..3] / a[u][4..7] = the A fragments of k step 0 / 1, b[u][0..1] / b[u][2..3] likewise
        f32x16_t acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[u][ks * 2 + i].f, a[u][ks * 4 + j].f, acc[i][j], 0, 0, 0);
        }
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (t == 12345.678f) sink[0] = t;
    }
}

// flops per launch: both shapes issue 2 sets x 32 x 16384 (= 2 x 2 x 8 x 32768) flops per wave and iteration
extern "C" void launch_mfma(void* stream, int shape, float* sink, int blocks, int iters, int zero) {
    if (shape == 16) hipLaunchKernelGGL((mfma_kernel<16>), dim3(blocks), dim3(512), 0, (hipStream_t)stream, sink, iters, zero);
    else hipLaunchKernelGGL((mfma_kernel<32>), dim3(blocks), dim3(512), 0, (hipStream_t)stream, sink, iters, zero);
}

// Experiment (r06): what does the fp32-residual epilogue of the 256x256 ring tile cost by its ADDRESS PATTERN alone?
// One 512-thread block per CU, every block walks tiles of a [M][N] fp32 matrix: C = acc + R (4 B read + 4 B written per element), 128 accumulators per lane.
//   pattern 0 (shipped fragment layout, operands swapped): lane = (fg = lane >> 4, m = lane & 15): row 16 mi + m, columns 16 ni + 4 fg .. + 3
//              -> one float4 instruction touches 16 rows x 64 B; lanes next to each other sit in DIFFERENT rows
//   pattern 1 (operands un-swapped, W rows interleaved over the four column tiles): lane = (fg, j): row 16 mi + 4 fg + r, columns 4 j .. 4 j + 3
//              -> one float4 instruction touches 4 rows x 256 B; 16 neighbouring lanes cover one row segment
// GAP > 0 inserts a spin of that many s_sleep units between tiles (the K loop during which HBM idles), so that the bursts of all CUs coincide as in the GEMM.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(512) void epi_kernel(const float* __restrict__ R, float* __restrict__ C, int M, int N, int gap) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wm = w >> 2, wn = w & 3;
    const int fg = lane >> 4, lo = lane & 15;
    f32x4 acc[4][8];
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
        for (int mi = 0; mi < 8; ++mi) acc[ni][mi] = f32x4{(float)lane, (float)ni, (float)mi, 1.0f};
    const int tn = N / 256, tiles = (M / 256) * tn;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int m0 = (t / tn) * 256 + wm * 128, n0 = (t % tn) * 256 + wn * 64;
        if (PAT == 0) {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
                const long row = (long)(m0 + mi * 16 + lo) * N + n0 + 4 * fg;
#pragma unroll
                for (int ni = 0; ni < 4; ++ni) {
                    const f32x4 rr = *reinterpret_cast<const f32x4*>(R + row + ni * 16);
                    *reinterpret_cast<f32x4*>(C + row + ni * 16) = acc[ni][mi] + rr;
                }
            }
        } else {
#pragma unroll
            for (int mi = 0; mi < 8; ++mi) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const long off = (long)(m0 + mi * 16 + 4 * fg + r) * N + n0 + 4 * lo;
                    const f32x4 rr = *reinterpret_cast<const f32x4*>(R + off);
                    *reinterpret_cast<f32x4*>(C + off) = f32x4{acc[0][mi][r], acc[1][mi][r], acc[2][mi][r], acc[3][mi][r]} + rr;
                }
            }
        }
        for (int g = 0; g < gap; ++g) __builtin_amdgcn_s_sleep(127);
        __syncthreads();
    }
}

extern "C" void launch_epi(void* stream, const void* R, void* C, int M, int N, int pat, int gap, int blocks) {
    if (pat == 0) hipLaunchKernelGGL((epi_kernel<0>), dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const float*)R, (float*)C, M, N, gap);
    else hipLaunchKernelGGL((epi_kernel<1>), dim3(blocks), dim3(512), 0, (hipStream_t)stream, (const float*)R, (float*)C, M, N, gap);
}

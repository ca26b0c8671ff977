// Kernels of the VQGAN mask tokenizer (reference models/taming_transformer/modules.py, quantize.py:272-330,
// vqgan.py:54-99) that the GEMM / attention / embedding kernels do not already cover.  Feature maps are token-major
// [b, h*w, C] bf16 like the SegModule path, so every convolution is im2col + MFMA GEMM:
//   im2col3x3s_kernel      3x3 window with stride and top/left padding (Downsample: pad (0,1,0,1), stride 2, modules.py:66-72)
//   groupnorm stats/apply  GroupNorm(32, C, eps 1e-6) (+ swish), two passes: per-(batch, pixel chunk, group) partial sums in a
//                          caller workspace, summed in a FIXED order by the apply pass (deterministic, no atomics)
//   upsample_nearest2x     F.interpolate(scale_factor=2, mode="nearest") (modules.py:50)
//   softmax_rows           AttnBlock softmax over keys (single head of C = 512 channels: scores come from the GEMM)
//   row_sqnorm / vq_argmin nearest codebook entry: argmin_n |e_n|^2 - 2 z.e_n (the |z|^2 term of quantize.py:286-288 is
//                          constant per row), first minimum wins like torch.argmin
#include "common.h"
#include "crab_internal.h"
#include <math.h>

namespace {

__global__ void im2col3x3s_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int h, int w, int C, int stride, int pt, int pl,
                                  int oh, int ow) {
    const int pix = blockIdx.x;                       // b*oh*ow + oy*ow + ox
    const int b = pix / (oh * ow), r = pix % (oh * ow), oy = r / ow, ox = r % ow;
    const int nv = C >> 3;
    for (int i = threadIdx.x; i < 9 * nv; i += blockDim.x) {
        const int tap = i / nv, c8 = i % nv;
        const int yy = oy * stride + tap / 3 - pt, xx = ox * stride + tap % 3 - pl;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) v = *reinterpret_cast<const u32x4*>(in + (((long)b * h + yy) * w + xx) * C + c8 * 8);
        *reinterpret_cast<u32x4*>(out + (long)pix * 9 * C + tap * C + c8 * 8) = v;
    }
}

constexpr int GN_MAXC = 1024;                          // channels per block pass (4 per thread)

// grid (chunks, B); partial[b][chunk][g] = {sum, sum of squares} over the chunk's pixels and the group's channels
template <bool XF>
__global__ __launch_bounds__(256) void groupnorm_stats_kernel(const void* __restrict__ x_, int HW, int C, int G, int rows_per_chunk,
                                                              float* __restrict__ partial) {
    __shared__ float cs[GN_MAXC], cq[GN_MAXC];
    const int tid = threadIdx.x, b = blockIdx.y, ch = blockIdx.x;
    const int p0 = ch * rows_per_chunk, p1 = min(HW, p0 + rows_per_chunk);
    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f}, k[4] = {0.f, 0.f, 0.f, 0.f};
    const long base = (long)b * HW * C;
    const int cpg_ = C / G;
    // r06 (found by scripts/fuzz_ops.py on groups of 2..4 values): sums are taken of x - k, k = the group's first value of the image, so that
    // var = E[(x-k)^2] - E[x-k]^2 does not cancel when the group's spread is small against its mean (shifted-data form; k is the same for every chunk)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = tid + j * 256;
        if (c < C) k[j] = XF ? ((const float*)x_)[base + (c / cpg_) * cpg_] : bf2f(((const bf16_t*)x_)[base + (c / cpg_) * cpg_]);
    }
    for (int p = p0; p < p1; ++p) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = tid + j * 256;
            if (c < C) {
                const float v = (XF ? ((const float*)x_)[base + (long)p * C + c] : bf2f(((const bf16_t*)x_)[base + (long)p * C + c])) - k[j];
                s[j] += v; q[j] += v * v;
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int c = tid + j * 256; if (c < C) { cs[c] = s[j]; cq[c] = q[j]; } }
    __syncthreads();
    if (tid < G) {
        const int cpg = C / G;
        float a = 0.f, bq = 0.f;
        for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) { a += cs[c]; bq += cq[c]; }
        float* o = partial + (((long)b * gridDim.x + ch) * G + tid) * 2;
        o[0] = a; o[1] = bq;
        if (ch == 0)                                     // the shift, after the partial sums of all images: [B * chunks * G * 2 | B * G]
            partial[(long)gridDim.y * gridDim.x * G * 2 + (long)b * G + tid] =
                XF ? ((const float*)x_)[base + tid * cpg] : bf2f(((const bf16_t*)x_)[base + tid * cpg]);
    }
}

// grid (ceil(HW*C/8/256), B)
// WF: weight / bias are fp32 (r06: the GroupNorm parameters are not matrix operands; their bf16 rounding is a systematic 2^-9 error on every channel)
template <bool WF>
__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ out, int HW, int C, int G,
                                                              int chunks, const float* __restrict__ partial, const void* __restrict__ weight_,
                                                              const void* __restrict__ bias_, float eps, int swish) {
    __shared__ float mean_s[64], rstd_s[64];
    const int tid = threadIdx.x, b = blockIdx.y;
    if (tid < G) {
        float a = 0.f, q = 0.f;
        for (int ch = 0; ch < chunks; ++ch) { const float* o = partial + (((long)b * chunks + ch) * G + tid) * 2; a += o[0]; q += o[1]; }
        const float n = (float)HW * (float)(C / G);
        const float m = a / n;
        const float var = fmaxf(q / n - m * m, 0.f);
        mean_s[tid] = m + partial[(long)gridDim.y * chunks * G * 2 + (long)b * G + tid]; rstd_s[tid] = rsqrtf(var + eps);
    }
    __syncthreads();
    const long v8 = (long)blockIdx.x * 256 + tid;
    const long nv = (long)HW * C / 8;
    if (v8 >= nv) return;
    const int c0 = (int)((v8 * 8) % C);
    const long off = (long)b * HW * C + v8 * 8;
    const u32x4 xv = *reinterpret_cast<const u32x4*>(x + off);
    float wf[8], bfv[8];
    if (WF) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { wf[e] = ((const float*)weight_)[c0 + e]; bfv[e] = ((const float*)bias_)[c0 + e]; }
    } else {
        const u32x4 wv = *reinterpret_cast<const u32x4*>((const bf16_t*)weight_ + c0);
        const u32x4 bv = *reinterpret_cast<const u32x4*>((const bf16_t*)bias_ + c0);
#pragma unroll
        for (int e = 0; e < 4; ++e) { wf[2 * e] = lo_bf(wv[e]); wf[2 * e + 1] = hi_bf(wv[e]); bfv[2 * e] = lo_bf(bv[e]); bfv[2 * e + 1] = hi_bf(bv[e]); }
    }
    const int cpg = C / G;
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float y[2];
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf) {
            const int c = c0 + 2 * e + hlf, g = c / cpg;
            const float xx = hlf ? hi_bf(xv[e]) : lo_bf(xv[e]);
            const float ww = wf[2 * e + hlf];
            const float bb = bfv[2 * e + hlf];
            float v = (xx - mean_s[g]) * rstd_s[g] * ww + bb;
            if (swish) v = v / (1.0f + __expf(-v));
            y[hlf] = v;
        }
        o[e] = pack_bf2(y[0], y[1]);
    }
    *reinterpret_cast<u32x4*>(out + off) = o;
}

__global__ void upsample_nearest2x_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int h, int w, int C, long total8) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total8) return;
    const int nv = C >> 3;
    const int c8 = (int)(i % nv);
    long r = i / nv;
    const int X = (int)(r % (2 * w)); r /= (2 * w);
    const int Y = (int)(r % (2 * h));
    const long b = r / (2 * h);
    *reinterpret_cast<u32x4*>(out + i * 8) = *reinterpret_cast<const u32x4*>(in + (((b * h + (Y >> 1)) * w + (X >> 1)) * (long)C) + c8 * 8);
}

// one block per row: out = softmax(scale * in)
__global__ __launch_bounds__(256) void softmax_rows_kernel(const float* __restrict__ in, long ldi, bf16_t* __restrict__ out, long ldo, int N,
                                                           float scale) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* x = in + (long)blockIdx.x * ldi;
    float mx = -INFINITY;
    for (int i = tid; i < N; i += 256) mx = fmaxf(mx, x[i] * scale);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int i = tid; i < N; i += 256) s += __expf(x[i] * scale - mx);
    s = wave_sum(s);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    bf16_t* o = out + (long)blockIdx.x * ldo;
    for (int i = tid; i < N; i += 256) o[i] = f2bf(__expf(x[i] * scale - mx) * inv);
}

__global__ void row_sqnorm_kernel(const bf16_t* __restrict__ e, long lde, int N, int D, float* __restrict__ out) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) { const float v = bf2f(e[(long)n * lde + i]); s += v * v; }
    s = wave_sum(s);
    if (lane == 0) out[n] = s;
}

// one block per row: idx = first argmin_n (e2[n] - 2 dots[m][n])
__global__ __launch_bounds__(256) void vq_argmin_kernel(const float* __restrict__ dots, long ldd, const float* __restrict__ e2, int N,
                                                        int64_t* __restrict__ idx, int64_t offset) {
    __shared__ float bv[4];
    __shared__ int bi[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* d = dots + (long)blockIdx.x * ldd;
    float best = INFINITY; int besti = 0x7fffffff;
    for (int n = tid; n < N; n += 256) {
        const float v = e2[n] - 2.0f * d[n];
        if (v < best) { best = v; besti = n; }                    // ascending n per thread: first minimum kept
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (ov < best || (ov == best && oi < besti)) { best = ov; besti = oi; }
    }
    if (lane == 0) { bv[wv] = best; bi[wv] = besti; }
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < 4; ++k)
            if (bv[k] < best || (bv[k] == best && bi[k] < besti)) { best = bv[k]; besti = bi[k]; }
        idx[blockIdx.x] = (int64_t)besti + offset;
    }
}

// ---- the quantiser in fp32 (r06; quantize.py:272-313): idx[m] = first argmin_n (|z_m|^2 + |e_n|^2) - 2 z_m . e_n with z (the latents of the
// last conv, unrounded) and e (the codebook) in fp32 and the reference's fp32 expression (A + B) - 2 C per entry.  The codebook ids are INDEX
// work: with bf16 operands an id flips wherever the nearest / second-nearest margin is below the bf16 distance error (~1e-2 of |z||e|); in
// fp32 only where it is below fp32 summation noise.  2 M N D flops (2.1 GF per 256 x 256 mask: irrelevant next to the encoder), vector FMAs.
// grid (ceil(M / 16), nsplit): a block owns 16 latent rows (LDS, broadcast reads) and every nsplit-th 64-entry tile of the codebook (staged through
// LDS, row stride D + 1: conflict-free); a wave owns 4 of the rows, a lane one entry of the tile; k ascends in both the dot product and the
// tile walk, so a thread meets its entries in ascending n (first minimum kept by strict <).  part[m][split] = (best d, its n).
constexpr int VQ_RT = 16, VQ_NT = 64, VQ_MAXD = 256;

__global__ __launch_bounds__(256) void vq_nearest_f32_kernel(const float* __restrict__ z, long ldz, const float* __restrict__ e, long lde,
                                                             const float* __restrict__ e2, int M, int N, int D, float* __restrict__ pbest,
                                                             int* __restrict__ pidx) {
    extern __shared__ float vq_lds[];
    float* zs = vq_lds;                                  // [VQ_RT][D]
    float* es = vq_lds + VQ_RT * D;                      // [VQ_NT][D + 1]
    __shared__ float z2s[VQ_RT];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m0 = blockIdx.x * VQ_RT, split = blockIdx.y, nsplit = gridDim.y;
    for (int i = tid; i < VQ_RT * D; i += 256) {
        const int r = i / D, k = i % D;
        zs[i] = (m0 + r < M) ? z[(long)(m0 + r) * ldz + k] : 0.f;
    }
    __syncthreads();
    for (int j = 0; j < 4; ++j) {                        // |z|^2 of the wave's rows (the reference's torch.sum(z ** 2, dim = 1))
        const int r = wv * 4 + j;
        float s = 0.f;
        for (int k = lane; k < D; k += 64) s += zs[r * D + k] * zs[r * D + k];
        s = wave_sum(s);
        if (lane == 0) z2s[r] = s;
    }
    float best[4] = {INFINITY, INFINITY, INFINITY, INFINITY};
    int besti[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
    const int ntiles = (N + VQ_NT - 1) / VQ_NT;
    for (int t = split; t < ntiles; t += nsplit) {
        const int n0 = t * VQ_NT;
        __syncthreads();                                 // the previous tile has been consumed (and z2s is visible)
        for (int i = tid; i < VQ_NT * (D >> 2); i += 256) {
            const int r = i / (D >> 2), k4 = i % (D >> 2);
            float4 v = {0.f, 0.f, 0.f, 0.f};
            if (n0 + r < N) v = *reinterpret_cast<const float4*>(e + (long)(n0 + r) * lde + k4 * 4);
            float* d = es + r * (D + 1) + k4 * 4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        __syncthreads();
        const int n = n0 + lane;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float* er = es + lane * (D + 1);
        const float* zr = zs + wv * 4 * D;
        for (int k = 0; k < D; ++k) {
            const float ev = er[k];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = fmaf(zr[j * D + k], ev, acc[j]);
        }
        if (n < N) {
            const float en = e2[n];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float dv = (z2s[wv * 4 + j] + en) - 2.0f * acc[j];
                if (dv < best[j]) { best[j] = dv; besti[j] = n; }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float b = best[j]; int bi = besti[j];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_xor(b, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov < b || (ov == b && oi < bi)) { b = ov; bi = oi; }
        }
        const int m = m0 + wv * 4 + j;
        if (lane == 0 && m < M) { pbest[(long)m * nsplit + split] = b; pidx[(long)m * nsplit + split] = bi; }
    }
}

__global__ void vq_nearest_finish_kernel(const float* __restrict__ pbest, const int* __restrict__ pidx, int M, int nsplit, int64_t* __restrict__ idx,
                                         int64_t offset) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= M) return;
    float b = INFINITY; int bi = 0x7fffffff;
    for (int s = 0; s < nsplit; ++s) {
        const float v = pbest[(long)m * nsplit + s];
        const int i = pidx[(long)m * nsplit + s];
        if (v < b || (v == b && i < bi)) { b = v; bi = i; }
    }
    idx[m] = (int64_t)bi + offset;
}

__global__ void row_sqnorm_f32_kernel(const float* __restrict__ e, long lde, int N, int D, float* __restrict__ out) {
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (n >= N) return;
    float s = 0.f;
    for (int i = lane; i < D; i += 64) { const float v = e[(long)n * lde + i]; s += v * v; }
    s = wave_sum(s);
    if (lane == 0) out[n] = s;
}

// ---- the PRECISE form of the VQGAN encoder (r06): the codebook ids are index work, and an id flips wherever the nearest / second-nearest margin is
// below the error of the latents - which with bf16 conv operands is ~6e-3 of their scale however the rest is stored (the operand floor,
// oracle/vqgan_oracle.py emulate("floor"): two flips on the reference fixture).  The encoder in front of the quantiser therefore keeps its
// activations in fp32 and feeds the MFMA GEMMs SPLIT operands: x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 mantissa bits), and
//     x . w  ~=  x_hi . w_hi  +  x_lo . w_hi  +  x_hi . w_lo          (the dropped lo . lo term is 2^-18 relative)
// as ONE bf16 GEMM over a 3x longer K: the activation map is written as [hi | lo | hi] channels, the weight as [hi | hi | lo], fp32 accumulation
// inside the matrix pipe - no new GEMM kernel, 3x the FLOPs of a path that is 0.25 TFLOP per mask.
//   split3_kernel            fp32 [M, C] -> bf16 [M, 3C] in pattern 0 ([hi | lo | hi], activations) or 1 ([hi | hi | lo], weights / the B operand)
//   groupnorm_f32_apply      GroupNorm (+ swish) fp32 -> fp32 (statistics as before, from the fp32 map)
//   add_bias_f32             x[m, c] += b[c] in fp32 (the GEMM epilogue's bias is a bf16 operand)
//   softmax_rows_f32         the AttnBlock's row softmax, fp32 -> fp32
__global__ void split3_kernel(const float* __restrict__ x, long ldx, bf16_t* __restrict__ out, long ldo, int M, int C, int Cp, int pattern) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;          // over M * Cp (Cp = C padded to 8: padded channels are zero)
    if (i >= (long)M * Cp) return;
    const int c = (int)(i % Cp);
    const long m = i / Cp;
    const float v = c < C ? x[m * ldx + c] : 0.f;
    const bf16_t hi = f2bf(v);
    const bf16_t lo = f2bf(v - bf2f(hi));
    bf16_t* o = out + m * ldo + c;
    o[0] = hi;
    o[Cp] = pattern ? hi : lo;
    o[2 * Cp] = pattern ? lo : hi;
}

__global__ __launch_bounds__(256) void groupnorm_f32_apply_kernel(const float* __restrict__ x, float* __restrict__ out, int HW, int C, int G, int chunks,
                                                                  const float* __restrict__ partial, const float* __restrict__ weight,
                                                                  const float* __restrict__ bias, float eps, int swish) {
    __shared__ float mean_s[64], rstd_s[64];
    const int tid = threadIdx.x, b = blockIdx.y;
    if (tid < G) {
        float a = 0.f, q = 0.f;
        for (int ch = 0; ch < chunks; ++ch) { const float* o = partial + (((long)b * chunks + ch) * G + tid) * 2; a += o[0]; q += o[1]; }
        const float n = (float)HW * (float)(C / G);
        const float m = a / n;
        const float var = fmaxf(q / n - m * m, 0.f);
        mean_s[tid] = m + partial[(long)gridDim.y * chunks * G * 2 + (long)b * G + tid]; rstd_s[tid] = rsqrtf(var + eps);
    }
    __syncthreads();
    const long i = (long)blockIdx.x * 256 + tid;
    if (i >= (long)HW * C) return;
    const int c = (int)(i % C), g = c / (C / G);
    const long off = (long)b * HW * C + i;
    float v = (x[off] - mean_s[g]) * rstd_s[g] * weight[c] + bias[c];
    if (swish) v = v / (1.0f + expf(-v));
    out[off] = v;
}

__global__ void add_bias_f32_kernel(float* __restrict__ x, long ldx, const float* __restrict__ b, long M, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * C) return;
    x[(i / C) * ldx + i % C] += b[i % C];
}

__global__ __launch_bounds__(256) void softmax_rows_f32_kernel(const float* __restrict__ in, long ldi, float* __restrict__ out, long ldo, int N, float scale) {
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* x = in + (long)blockIdx.x * ldi;
    float mx = -INFINITY;
    for (int i = tid; i < N; i += 256) mx = fmaxf(mx, x[i] * scale);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wv] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int i = tid; i < N; i += 256) s += expf(x[i] * scale - mx);
    s = wave_sum(s);
    if (lane == 0) red[wv] = s;
    __syncthreads();
    const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
    float* o = out + (long)blockIdx.x * ldo;
    for (int i = tid; i < N; i += 256) o[i] = expf(x[i] * scale - mx) * inv;
}

}  // namespace

#define S_(x) ((hipStream_t)(x))
static inline unsigned cdiv_(long a, long b) { return (unsigned)((a + b - 1) / b); }

extern "C" int crab_im2col3x3_strided(crab_ctx* ctx, void* stream, const void* in, void* out, int B, int h, int w, int C, int stride, int pad_top,
                                      int pad_left, int oh, int ow) {
    if (!ctx) return CRAB_E_INVALID;
    if (!in || !out || B <= 0 || h <= 0 || w <= 0 || C <= 0 || (C & 7) || stride <= 0 || oh <= 0 || ow <= 0)
        return crab_fail(ctx, CRAB_E_INVALID, "im2col3x3_strided: bad argument (C must be a multiple of 8)");
    hipLaunchKernelGGL(im2col3x3s_kernel, dim3(B * oh * ow), dim3(256), 0, S_(stream), (const bf16_t*)in, (bf16_t*)out, h, w, C, stride, pad_top,
                       pad_left, oh, ow);
    return crab_check_launch(ctx, "im2col3x3_strided");
}

extern "C" int64_t crab_groupnorm_workspace(int B, int HW, int G) {
    int chunks = (HW + 63) / 64;
    if (chunks > 64) chunks = 64;
    return ((int64_t)B * chunks * G * 2 + (int64_t)B * G) * (int64_t)sizeof(float);          // partial {sum, sum of squares} per chunk + the shift per (image, group)
}

extern "C" int crab_groupnorm_p(crab_ctx* ctx, void* stream, const void* x, void* out, int B, int HW, int C, int G, float eps, const void* weight,
                                const void* bias, int w_fp32, int swish, void* workspace, int64_t workspace_bytes);

extern "C" int crab_groupnorm(crab_ctx* ctx, void* stream, const void* x, void* out, int B, int HW, int C, int G, float eps, const void* weight,
                              const void* bias, int swish, void* workspace, int64_t workspace_bytes) {
    return crab_groupnorm_p(ctx, stream, x, out, B, HW, C, G, eps, weight, bias, 0, swish, workspace, workspace_bytes);
}

extern "C" int crab_groupnorm_p(crab_ctx* ctx, void* stream, const void* x, void* out, int B, int HW, int C, int G, float eps, const void* weight,
                                const void* bias, int w_fp32, int swish, void* workspace, int64_t workspace_bytes) {
    if (!ctx) return CRAB_E_INVALID;
    if (!x || !out || !weight || !bias || !workspace || B <= 0 || HW <= 0 || C <= 0 || G <= 0 || G > 64 || C % G || (C & 7) || C > GN_MAXC)
        return crab_fail(ctx, CRAB_E_INVALID, "groupnorm: bad argument (C % 8 == 0, C <= 1024, G <= 64)");
    if (workspace_bytes < crab_groupnorm_workspace(B, HW, G)) return crab_fail(ctx, CRAB_E_WORKSPACE, "groupnorm: workspace too small");
    int chunks = (HW + 63) / 64;
    if (chunks > 64) chunks = 64;
    const int rows = (HW + chunks - 1) / chunks;
    chunks = (HW + rows - 1) / rows;
    hipLaunchKernelGGL((groupnorm_stats_kernel<false>), dim3(chunks, B), dim3(256), 0, S_(stream), x, HW, C, G, rows, (float*)workspace);
    int rc = crab_check_launch(ctx, "groupnorm_stats_kernel");
    if (rc) return rc;
    if (w_fp32)
        hipLaunchKernelGGL((groupnorm_apply_kernel<true>), dim3(cdiv_((long)HW * C / 8, 256), B), dim3(256), 0, S_(stream), (const bf16_t*)x, (bf16_t*)out, HW, C,
                           G, chunks, (const float*)workspace, weight, bias, eps, swish);
    else
        hipLaunchKernelGGL((groupnorm_apply_kernel<false>), dim3(cdiv_((long)HW * C / 8, 256), B), dim3(256), 0, S_(stream), (const bf16_t*)x, (bf16_t*)out, HW, C,
                           G, chunks, (const float*)workspace, weight, bias, eps, swish);
    return crab_check_launch(ctx, w_fp32 ? "groupnorm_apply_kernel<fp32 params>" : "groupnorm_apply_kernel");
}

extern "C" int crab_upsample_nearest2x(crab_ctx* ctx, void* stream, const void* in, void* out, int B, int h, int w, int C) {
    if (!ctx) return CRAB_E_INVALID;
    if (!in || !out || B <= 0 || h <= 0 || w <= 0 || C <= 0 || (C & 7)) return crab_fail(ctx, CRAB_E_INVALID, "upsample_nearest2x: bad argument");
    const long total8 = (long)B * 4 * h * w * (C / 8);
    hipLaunchKernelGGL(upsample_nearest2x_kernel, dim3(cdiv_(total8, 256)), dim3(256), 0, S_(stream), (const bf16_t*)in, (bf16_t*)out, h, w, C, total8);
    return crab_check_launch(ctx, "upsample_nearest2x");
}

extern "C" int crab_softmax_rows(crab_ctx* ctx, void* stream, const float* in, int64_t ldi, void* out, int64_t ldo, int M, int N, float scale) {
    if (!ctx) return CRAB_E_INVALID;
    if (!in || !out || M <= 0 || N <= 0) return crab_fail(ctx, CRAB_E_INVALID, "softmax_rows: bad argument");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3(M), dim3(256), 0, S_(stream), in, (long)ldi, (bf16_t*)out, (long)ldo, N, scale);
    return crab_check_launch(ctx, "softmax_rows");
}

extern "C" int crab_row_sqnorm(crab_ctx* ctx, void* stream, const void* e, int64_t lde, int N, int D, float* out) {
    if (!ctx) return CRAB_E_INVALID;
    if (!e || !out || N <= 0 || D <= 0) return crab_fail(ctx, CRAB_E_INVALID, "row_sqnorm: bad argument");
    hipLaunchKernelGGL(row_sqnorm_kernel, dim3(cdiv_(N, 4)), dim3(256), 0, S_(stream), (const bf16_t*)e, (long)lde, N, D, out);
    return crab_check_launch(ctx, "row_sqnorm");
}

extern "C" int crab_vq_argmin(crab_ctx* ctx, void* stream, const float* dots, int64_t ldd, const float* e2, int M, int N, int64_t* idx,
                              int64_t offset) {
    if (!ctx) return CRAB_E_INVALID;
    if (!dots || !e2 || !idx || M <= 0 || N <= 0) return crab_fail(ctx, CRAB_E_INVALID, "vq_argmin: bad argument");
    hipLaunchKernelGGL(vq_argmin_kernel, dim3(M), dim3(256), 0, S_(stream), dots, (long)ldd, e2, N, idx, offset);
    return crab_check_launch(ctx, "vq_argmin");
}

extern "C" int crab_row_sqnorm_f32(crab_ctx* ctx, void* stream, const float* e, int64_t lde, int N, int D, float* out) {
    if (!ctx) return CRAB_E_INVALID;
    if (!e || !out || N <= 0 || D <= 0) return crab_fail(ctx, CRAB_E_INVALID, "row_sqnorm_f32: bad argument");
    hipLaunchKernelGGL(row_sqnorm_f32_kernel, dim3(cdiv_(N, 4)), dim3(256), 0, S_(stream), e, (long)lde, N, D, out);
    return crab_check_launch(ctx, "row_sqnorm_f32");
}

static int vq_nsplit(int M, int N) {
    const int rb = (M + VQ_RT - 1) / VQ_RT, nt = (N + VQ_NT - 1) / VQ_NT;
    int ns = (1024 + rb - 1) / rb;                       // ~4 blocks per CU
    if (ns > nt) ns = nt;
    if (ns > 64) ns = 64;
    return ns < 1 ? 1 : ns;
}

// monotone in M and N (one workspace sized for the largest call serves smaller ones): splits * M <= min(64, column tiles) * M and <= 16 384 + M
extern "C" int64_t crab_vq_nearest_f32_workspace(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    const int64_t nt = (N + VQ_NT - 1) / VQ_NT;
    int64_t rows = (nt < 64 ? nt : 64) * (int64_t)M;
    if (rows > 16384 + (int64_t)M) rows = 16384 + (int64_t)M;
    return rows * 8;
}

extern "C" int crab_vq_nearest_f32(crab_ctx* ctx, void* stream, const float* z, int64_t ldz, const float* e, int64_t lde, const float* e2, int M, int N,
                                   int D, int64_t* idx, int64_t offset, void* workspace, int64_t workspace_bytes) {
    if (!ctx) return CRAB_E_INVALID;
    if (!z || !e || !e2 || !idx || M <= 0 || N <= 0 || D <= 0 || D > VQ_MAXD || (D & 3) || (lde & 3) || ((uintptr_t)e & 15))
        return crab_fail(ctx, CRAB_E_INVALID, "vq_nearest_f32: bad argument (D % 4 == 0, D <= 256, 16-byte aligned codebook rows)");
    if (!workspace || workspace_bytes < crab_vq_nearest_f32_workspace(M, N)) return crab_fail(ctx, CRAB_E_WORKSPACE, "vq_nearest_f32: needs crab_vq_nearest_f32_workspace(M, N) bytes");
    const int ns = vq_nsplit(M, N);
    float* pbest = (float*)workspace;
    int* pidx = (int*)((char*)workspace + (int64_t)M * ns * 4);
    const size_t lds = (size_t)(VQ_RT * D + VQ_NT * (D + 1)) * sizeof(float);
    static bool attr = false;
    if (!attr) {
        CRAB_HIP_TRY(ctx, hipFuncSetAttribute((const void*)vq_nearest_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((VQ_RT * VQ_MAXD + VQ_NT * (VQ_MAXD + 1)) * sizeof(float))));
        attr = true;
    }
    hipLaunchKernelGGL(vq_nearest_f32_kernel, dim3((M + VQ_RT - 1) / VQ_RT, ns), dim3(256), lds, S_(stream), z, (long)ldz, e, (long)lde, e2, M, N, D, pbest, pidx);
    int rc = crab_check_launch(ctx, "vq_nearest_f32_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(vq_nearest_finish_kernel, dim3(cdiv_(M, 256)), dim3(256), 0, S_(stream), pbest, pidx, M, ns, idx, offset);
    return crab_check_launch(ctx, "vq_nearest_finish_kernel");
}

// ---- precise (split-operand) encoder entry points, see the kernels' header
extern "C" int crab_split3(crab_ctx* ctx, void* stream, const float* x, int64_t ldx, void* out, int64_t ldo, int M, int C, int pattern) {
    if (!ctx) return CRAB_E_INVALID;
    const int Cp = (C + 7) / 8 * 8;
    if (!x || !out || M <= 0 || C <= 0 || ldo < 3 * Cp || (pattern != 0 && pattern != 1)) return crab_fail(ctx, CRAB_E_INVALID, "split3: bad argument (ldo >= 3 * round_up(C, 8))");
    hipLaunchKernelGGL(split3_kernel, dim3(cdiv_((long)M * Cp, 256)), dim3(256), 0, S_(stream), x, (long)ldx, (bf16_t*)out, (long)ldo, M, C, Cp, pattern);
    return crab_check_launch(ctx, "split3_kernel");
}

extern "C" int crab_groupnorm_f32(crab_ctx* ctx, void* stream, const float* x, float* out, int B, int HW, int C, int G, float eps, const float* weight,
                                  const float* bias, int swish, void* workspace, int64_t workspace_bytes) {
    if (!ctx) return CRAB_E_INVALID;
    if (!x || !out || !weight || !bias || !workspace || B <= 0 || HW <= 0 || C <= 0 || G <= 0 || G > 64 || C % G || C > GN_MAXC)
        return crab_fail(ctx, CRAB_E_INVALID, "groupnorm_f32: bad argument (C <= 1024, G <= 64)");
    if (workspace_bytes < crab_groupnorm_workspace(B, HW, G)) return crab_fail(ctx, CRAB_E_WORKSPACE, "groupnorm_f32: workspace too small");
    int chunks = (HW + 63) / 64;
    if (chunks > 64) chunks = 64;
    const int rows = (HW + chunks - 1) / chunks;
    chunks = (HW + rows - 1) / rows;
    hipLaunchKernelGGL((groupnorm_stats_kernel<true>), dim3(chunks, B), dim3(256), 0, S_(stream), (const void*)x, HW, C, G, rows, (float*)workspace);
    int rc = crab_check_launch(ctx, "groupnorm_stats_kernel<fp32>");
    if (rc) return rc;
    hipLaunchKernelGGL(groupnorm_f32_apply_kernel, dim3(cdiv_((long)HW * C, 256), B), dim3(256), 0, S_(stream), x, out, HW, C, G, chunks, (const float*)workspace,
                       weight, bias, eps, swish);
    return crab_check_launch(ctx, "groupnorm_f32_apply_kernel");
}

extern "C" int crab_add_bias_f32(crab_ctx* ctx, void* stream, float* x, int64_t ldx, const float* bias, int64_t M, int C) {
    if (!ctx) return CRAB_E_INVALID;
    if (!x || !bias || M <= 0 || C <= 0) return crab_fail(ctx, CRAB_E_INVALID, "add_bias_f32: bad argument");
    hipLaunchKernelGGL(add_bias_f32_kernel, dim3(cdiv_(M * C, 256)), dim3(256), 0, S_(stream), x, (long)ldx, bias, (long)M, C);
    return crab_check_launch(ctx, "add_bias_f32_kernel");
}

extern "C" int crab_softmax_rows_f32(crab_ctx* ctx, void* stream, const float* in, int64_t ldi, float* out, int64_t ldo, int M, int N, float scale) {
    if (!ctx) return CRAB_E_INVALID;
    if (!in || !out || M <= 0 || N <= 0) return crab_fail(ctx, CRAB_E_INVALID, "softmax_rows_f32: bad argument");
    hipLaunchKernelGGL(softmax_rows_f32_kernel, dim3(M), dim3(256), 0, S_(stream), in, (long)ldi, out, (long)ldo, N, scale);
    return crab_check_launch(ctx, "softmax_rows_f32");
}

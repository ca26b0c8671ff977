// Small HBM/latency-bound kernels of the SegModule pixel path (reference models/multimodal_encoder.py:268-543, 891-1444):
// feature maps are kept TOKEN-MAJOR ([h*w, C] bf16, row = y*w + x) so that 1x1 convolutions, LayerNorm2d and the
// two-way transformer all run on the GEMM / LayerNorm / attention kernels; the kernels here cover what is left:
// 3x3 im2col, ConvTranspose2d(k=2,s=2) pixel shuffle, bilinear resize, random-Fourier dense positional encoding,
// row-broadcast add, the previous-mask gate and an in-place activation.
#include "common.h"
#include "crab_internal.h"
#include <math.h>

namespace {

// out[(b*h+y)*w+x, (ky*3+kx)*C + c] = in[b, y+ky-1, x+kx-1, c] (zero outside): Conv2d(k=3, pad=1) as one GEMM
__global__ void im2col3x3_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out, int B, int h, int w, int C) {
    const int pix = blockIdx.x;                       // b*h*w + y*w + x
    const int b = pix / (h * w), r = pix % (h * w), y = r / w, x = r % w;
    const int nv = C >> 3;
    for (int i = threadIdx.x; i < 9 * nv; i += blockDim.x) {
        const int tap = i / nv, c8 = i % nv;
        const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (yy >= 0 && yy < h && xx >= 0 && xx < w) v = *reinterpret_cast<const u32x4*>(in + (((long)b * h + yy) * w + xx) * C + c8 * 8);
        *reinterpret_cast<u32x4*>(out + (long)pix * 9 * C + tap * C + c8 * 8) = v;
    }
}

// ConvTranspose2d(k=2,s=2): g[h*w, 4*Co] (columns (dy,dx,co)) -> out[(2h)*(2w), Co] + bias
__global__ void pixel_shuffle2x_kernel(const bf16_t* __restrict__ g, const bf16_t* __restrict__ bias, bf16_t* __restrict__ out, int h, int w, int Co) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)4 * h * w * Co;
    if (idx >= total) return;
    const int co = idx % Co;
    long r = idx / Co;
    const int X = r % (2 * w), Y = r / (2 * w);
    const int y = Y >> 1, dy = Y & 1, x = X >> 1, dx = X & 1;
    float v = bf2f(g[((long)y * w + x) * 4 * Co + (dy * 2 + dx) * Co + co]);
    if (bias) v += bf2f(bias[co]);
    out[idx] = f2bf(v);
}

// bilinear resize, align_corners=False (F.interpolate): in element (c,y,x) at in[c*sc + y*sy + x*sx] (bf16 or fp32),
// out[C,H,W] fp32 contiguous: out = beta*out + alpha*interp
template <typename TI>
__global__ void bilinear_kernel(const TI* __restrict__ in, long sc, long sy, long sx, int C, int h, int w, float* __restrict__ out,
                                int H, int W, float alpha, float beta) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)C * H * W) return;
    const int X = idx % W, Y = (idx / W) % H, c = idx / ((long)W * H);
    float fy = ((float)Y + 0.5f) * ((float)h / (float)H) - 0.5f;
    float fx = ((float)X + 0.5f) * ((float)w / (float)W) - 0.5f;
    fy = fy < 0.f ? 0.f : fy;                                   // PyTorch clamps the source index at 0
    fx = fx < 0.f ? 0.f : fx;
    int y0 = (int)fy, x0 = (int)fx;
    int y1 = y0 + (y0 < h - 1 ? 1 : 0), x1 = x0 + (x0 < w - 1 ? 1 : 0);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    auto ld = [&](int yy, int xx) -> float {
        TI v = in[c * sc + yy * sy + xx * sx];
        return sizeof(TI) == 4 ? *reinterpret_cast<const float*>(&v) : bf2f(*reinterpret_cast<const bf16_t*>(&v));
    };
    float v = (1.f - ly) * ((1.f - lx) * ld(y0, x0) + lx * ld(y0, x1)) + ly * ((1.f - lx) * ld(y1, x0) + lx * ld(y1, x1));
    out[idx] = (beta != 0.f ? beta * out[idx] : 0.f) + alpha * v;
}

// PositionEmbeddingRandom.forward (:825-839): pe[(y*w+x), :] = [sin(2pi * ((2c-1) @ G)), cos(...)], c = ((x+.5)/w, (y+.5)/h)
__global__ void dense_pe_kernel(const float* __restrict__ G, bf16_t* __restrict__ pe, int h, int w, int F) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= h * w * F) return;
    const int f = idx % F, pix = idx / F, y = pix / w, x = pix % w;
    const float cx = 2.f * (((float)x + 0.5f) / (float)w) - 1.f, cy = 2.f * (((float)y + 0.5f) / (float)h) - 1.f;
    const float a = 6.283185307179586f * (cx * G[f] + cy * G[F + f]);
    pe[(long)pix * 2 * F + f] = f2bf(sinf(a));
    pe[(long)pix * 2 * F + F + f] = f2bf(cosf(a));
}

// out[m,:] = a[m,:] + b[m % brows, :]
__global__ void add_rows_kernel(const bf16_t* __restrict__ a, long lda, const bf16_t* __restrict__ b, long ldb, int brows, bf16_t* __restrict__ out,
                                long ldo, int M, int D) {
    const int nv = D >> 3;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)M * nv) return;
    const int m = idx / nv, c = idx % nv;
    u32x4 x = *reinterpret_cast<const u32x4*>(a + (long)m * lda + c * 8);
    u32x4 y = *reinterpret_cast<const u32x4*>(b + (long)(m % brows) * ldb + c * 8);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = pack_bf2(lo_bf(x[j]) + lo_bf(y[j]), hi_bf(x[j]) + hi_bf(y[j]));
    *reinterpret_cast<u32x4*>(out + (long)m * ldo + c * 8) = o;
}

// src[m,:] *= sigmoid(mean_c prev[m, c]) + 1   (:1112-1114), one wave per pixel row
__global__ __launch_bounds__(256) void mask_gate_kernel(const bf16_t* __restrict__ prev, long ldp, int ncls, bf16_t* __restrict__ src, long lds_, int M, int D) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= M) return;
    float s = 0.f;
    for (int c = lane; c < ncls; c += 64) s += bf2f(prev[(long)m * ldp + c]);
    s = wave_sum(s) / (float)ncls;
    const float g = 1.f / (1.f + expf(-s)) + 1.f;
    for (int e = lane; e < D; e += 64) src[(long)m * lds_ + e] = f2bf(bf2f(src[(long)m * lds_ + e]) * g);
}

// out[g,:] = scale * sum_{k<T} in[g*T + k, :]   (fused_pred_embeddings, :388-393: multiseg_scalar = 1/T constants)
__global__ void group_mean_kernel(const bf16_t* __restrict__ in, long ldi, bf16_t* __restrict__ out, long ldo, int G, int T, int D, float scale) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= G * D) return;
    const int g = idx / D, e = idx % D;
    float s = 0.f;
    for (int k = 0; k < T; ++k) s += bf2f(in[(long)(g * T + k) * ldi + e]);
    out[(long)g * ldo + e] = f2bf(scale * s);
}

// What the reference's eval loops make of a predicted mask before writing the PNG (scripts/quick_start.py:313-318: binary tasks
// `(sigmoid(pred) > 0.5) * 255`; utils/avss_utils.py:291-292: `argmax(softmax(pred, dim=classes))`): C == 1 -> 255 where pred > 0 (sigmoid is
// monotone, sigmoid(0) = 0.5), else 0; C > 1 -> the index of the first maximum over the C class planes (softmax is monotone).
__global__ void mask_labels_kernel(const float* __restrict__ pred, int C, long hw, uint8_t* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hw) return;
    if (C == 1) { out[i] = pred[i] > 0.0f ? 255 : 0; return; }
    float best = pred[i];
    int bi = 0;
    for (int c = 1; c < C; ++c) {
        const float v = pred[(long)c * hw + i];
        if (v > best || (v != v && best == best)) { best = v; bi = c; }      // NaN counts as the maximum, first one wins (torch.argmax; as class_areas_kernel)
    }
    out[i] = (uint8_t)bi;
}

__global__ void act_kernel(bf16_t* __restrict__ x, long n, int act) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = f2bf(apply_act(bf2f(x[i]), act));
}

}  // namespace

#define S_(x) ((hipStream_t)(x))
static inline unsigned cdiv_(long a, long b) { return (unsigned)((a + b - 1) / b); }

extern "C" {

int crab_im2col3x3(crab_ctx* ctx, void* stream, const void* in, void* out, int B, int h, int w, int C) {
    if (!ctx) return CRAB_E_INVALID;
    if (!in || !out || B <= 0 || h <= 0 || w <= 0 || (C & 7)) return crab_fail(ctx, CRAB_E_INVALID, "im2col3x3: bad argument");
    hipLaunchKernelGGL(im2col3x3_kernel, dim3(B * h * w), dim3(256), 0, S_(stream), (const bf16_t*)in, (bf16_t*)out, B, h, w, C);
    return crab_check_launch(ctx, "im2col3x3");
}

int crab_pixel_shuffle2x(crab_ctx* ctx, void* stream, const void* g, const void* bias, void* out, int h, int w, int Co) {
    if (!ctx) return CRAB_E_INVALID;
    if (!g || !out || h <= 0 || w <= 0 || Co <= 0) return crab_fail(ctx, CRAB_E_INVALID, "pixel_shuffle2x: bad argument");
    hipLaunchKernelGGL(pixel_shuffle2x_kernel, dim3(cdiv_((long)4 * h * w * Co, 256)), dim3(256), 0, S_(stream), (const bf16_t*)g, (const bf16_t*)bias,
                       (bf16_t*)out, h, w, Co);
    return crab_check_launch(ctx, "pixel_shuffle2x");
}

int crab_bilinear(crab_ctx* ctx, void* stream, const void* in, int in_fp32, int64_t sc, int64_t sy, int64_t sx, int C, int h, int w, float* out,
                  int H, int W, float alpha, float beta) {
    if (!ctx) return CRAB_E_INVALID;
    if (!in || !out || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return crab_fail(ctx, CRAB_E_INVALID, "bilinear: bad argument");
    unsigned blocks = cdiv_((long)C * H * W, 256);
    if (in_fp32) hipLaunchKernelGGL((bilinear_kernel<float>), dim3(blocks), dim3(256), 0, S_(stream), (const float*)in, (long)sc, (long)sy, (long)sx, C, h, w, out, H, W, alpha, beta);
    else hipLaunchKernelGGL((bilinear_kernel<bf16_t>), dim3(blocks), dim3(256), 0, S_(stream), (const bf16_t*)in, (long)sc, (long)sy, (long)sx, C, h, w, out, H, W, alpha, beta);
    return crab_check_launch(ctx, "bilinear");
}

int crab_dense_pe(crab_ctx* ctx, void* stream, const void* G, void* pe, int h, int w, int F) {
    if (!ctx) return CRAB_E_INVALID;
    if (!G || !pe || h <= 0 || w <= 0 || F <= 0) return crab_fail(ctx, CRAB_E_INVALID, "dense_pe: bad argument");
    hipLaunchKernelGGL(dense_pe_kernel, dim3(cdiv_((long)h * w * F, 256)), dim3(256), 0, S_(stream), (const float*)G, (bf16_t*)pe, h, w, F);
    return crab_check_launch(ctx, "dense_pe");
}

int crab_add_rows(crab_ctx* ctx, void* stream, const void* a, int64_t lda, const void* b, int64_t ldb, int brows, void* out, int64_t ldo, int M, int D) {
    if (!ctx) return CRAB_E_INVALID;
    if (!a || !b || !out || M <= 0 || brows <= 0 || (D & 7) || (lda & 7) || (ldb & 7) || (ldo & 7)) return crab_fail(ctx, CRAB_E_INVALID, "add_rows: bad argument");
    hipLaunchKernelGGL(add_rows_kernel, dim3(cdiv_((long)M * (D >> 3), 256)), dim3(256), 0, S_(stream), (const bf16_t*)a, (long)lda, (const bf16_t*)b, (long)ldb,
                       brows, (bf16_t*)out, (long)ldo, M, D);
    return crab_check_launch(ctx, "add_rows");
}

int crab_mask_gate(crab_ctx* ctx, void* stream, const void* prev, int64_t ldp, int ncls, void* src, int64_t lds_, int M, int D) {
    if (!ctx) return CRAB_E_INVALID;
    if (!prev || !src || M <= 0 || ncls <= 0 || D <= 0) return crab_fail(ctx, CRAB_E_INVALID, "mask_gate: bad argument");
    hipLaunchKernelGGL(mask_gate_kernel, dim3(cdiv_(M, 4)), dim3(256), 0, S_(stream), (const bf16_t*)prev, (long)ldp, ncls, (bf16_t*)src, (long)lds_, M, D);
    return crab_check_launch(ctx, "mask_gate");
}

int crab_group_mean(crab_ctx* ctx, void* stream, const void* in, int64_t ldi, void* out, int64_t ldo, int G, int T, int D, float scale) {
    if (!ctx) return CRAB_E_INVALID;
    if (!in || !out || G <= 0 || T <= 0 || D <= 0) return crab_fail(ctx, CRAB_E_INVALID, "group_mean: bad argument");
    hipLaunchKernelGGL(group_mean_kernel, dim3(cdiv_((long)G * D, 256)), dim3(256), 0, S_(stream), (const bf16_t*)in, (long)ldi, (bf16_t*)out, (long)ldo, G, T, D, scale);
    return crab_check_launch(ctx, "group_mean");
}

int crab_mask_labels(crab_ctx* ctx, void* stream, const float* pred, int C, int64_t hw, uint8_t* out) {
    if (!ctx) return CRAB_E_INVALID;
    if (!pred || !out || C <= 0 || C > 255 || hw <= 0) return crab_fail(ctx, CRAB_E_INVALID, "mask_labels: pred [C, hw] fp32, 1 <= C <= 255, out [hw] uint8");
    hipLaunchKernelGGL(mask_labels_kernel, dim3(cdiv_(hw, 256)), dim3(256), 0, S_(stream), pred, C, (long)hw, out);
    return crab_check_launch(ctx, "mask_labels");
}

int crab_act_inplace(crab_ctx* ctx, void* stream, void* x, int64_t n, int act) {
    if (!ctx) return CRAB_E_INVALID;
    if (!x || n <= 0) return crab_fail(ctx, CRAB_E_INVALID, "act_inplace: bad argument");
    hipLaunchKernelGGL(act_kernel, dim3(cdiv_(n, 256)), dim3(256), 0, S_(stream), (bf16_t*)x, (long)n, act);
    return crab_check_launch(ctx, "act_inplace");
}

}  // extern "C"

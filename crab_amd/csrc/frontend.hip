// Input front-end on the device: the step immediately before the hot path, CPU-side in the reference
// (dataset/quick_start_dataset.py:299-343 -> transformers CLIPImageProcessor / Pillow; dataset/audio_processor.py:29-41 ->
// torchaudio.compliance.kaldi.fbank).  Pure HBM / small-FFT work:
//   crab_bicubic_coeffs      host: Pillow's precompute_coeffs + normalize_coeffs_8bpc (22-bit fixed-point taps)
//   resample_u8_kernel       one separable pass of Pillow's 8-bit resampler (horizontal or vertical), uint8 HWC, bit-exact
//   clip_normalize_kernel    centre crop + 1/255 + (x - mean) / std + HWC -> CHW, fp32 or bf16 out
//   kaldi_fbank_kernel       one 25 ms frame per block: DC removal, pre-emphasis, povey window, 512-point FFT in LDS,
//                            power spectrum, 128 mel bins, log, (x - mean) * scale
#include "common.h"
#include "crab_internal.h"
#include <math.h>

namespace {

constexpr int PRECISION_BITS = 32 - 8 - 2;

// grid: (ceil(out_size * other / 256), N); one thread per output pixel (3 or C channels)
template <int C>
__global__ __launch_bounds__(256) void resample_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int H, int W, int out_size,
                                                          int horizontal, const int32_t* __restrict__ bounds,
                                                          const int32_t* __restrict__ kk, int ksize) {
    const int OH = horizontal ? H : out_size, OW = horizontal ? out_size : W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)OH * OW) return;
    const int oy = idx / OW, ox = idx % OW;
    const int o = horizontal ? ox : oy;
    const int x0 = bounds[2 * o], n = bounds[2 * o + 1];
    const int32_t* k = kk + (long)o * ksize;
    const uint8_t* s = src + (long)blockIdx.y * H * W * C;
    int acc[C];
#pragma unroll
    for (int c = 0; c < C; ++c) acc[c] = 1 << (PRECISION_BITS - 1);
    for (int i = 0; i < n; ++i) {
        const uint8_t* px = horizontal ? s + ((long)oy * W + x0 + i) * C : s + ((long)(x0 + i) * W + ox) * C;
        const int w = k[i];
#pragma unroll
        for (int c = 0; c < C; ++c) acc[c] += (int)px[c] * w;
    }
    uint8_t* d = dst + ((long)blockIdx.y * OH * OW + idx) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        int v = acc[c] >> PRECISION_BITS;
        d[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

// one thread per output pixel position (all 3 channels); out [N,3,size,size]
__global__ __launch_bounds__(256) void clip_normalize_kernel(const uint8_t* __restrict__ src, int H, int W, int top, int left, int size,
                                                             void* __restrict__ out, int out_bf16, float m0, float m1, float m2, float s0,
                                                             float s1, float s2, float rescale) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= size * size) return;
    const int y = idx / size, x = idx % size;
    const uint8_t* px = src + ((long)blockIdx.y * H * W + (long)(top + y) * W + left + x) * 3;
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float v = ((float)px[c] * rescale - mean[c]) / sd[c];
        const long o = ((long)blockIdx.y * 3 + c) * size * size + idx;
        if (out_bf16) reinterpret_cast<bf16_t*>(out)[o] = f2bf(v);
        else reinterpret_cast<float*>(out)[o] = v;
    }
}

// ---------------------------------------------------------------------------------------------- kaldi fbank
constexpr int FB_WIN = 400, FB_SHIFT = 160, FB_PAD = 512, FB_BINS = FB_PAD / 2 + 1, FB_MEL = 128;

__device__ __forceinline__ int bitrev9(int v) { return (int)(__brev((unsigned)v) >> 23); }

// grid (frames, waveforms); block 256
__global__ __launch_bounds__(256) void kaldi_fbank_kernel(const float* __restrict__ wave, long ldw, int n_samples, float in_scale,
                                                          float preemph, const float* __restrict__ window,
                                                          const float* __restrict__ mel_t /*[257][128]*/, float* __restrict__ out, int frames,
                                                          float out_sub, float out_scale) {
    __shared__ float re[FB_PAD], im[FB_PAD], twr[FB_PAD / 2], twi[FB_PAD / 2], raw[FB_WIN], red[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int f = blockIdx.x;
    const float* x = wave + (long)blockIdx.y * ldw + (long)f * FB_SHIFT;
    // twiddles exp(-2 pi i k / 512), k < 256
    {
        float s, c;
        sincospif(-(float)tid / 256.0f, &s, &c);
        twr[tid] = c; twi[tid] = s;
    }
    float part = 0.f;
    for (int i = tid; i < FB_WIN; i += 256) { const float v = x[i] * in_scale; raw[i] = v; part += v; }
    part = wave_sum(part);
    if (lane == 0) red[wv] = part;
    __syncthreads();
    const float mean = (red[0] + red[1] + red[2] + red[3]) / (float)FB_WIN;      // remove_dc_offset
    // pre-emphasis (replicate-padded previous sample), povey window, zero padding, bit-reversed placement
    for (int i = tid; i < FB_PAD; i += 256) {
        float v = 0.f;
        if (i < FB_WIN) {
            const float cur = raw[i] - mean, prev = raw[i > 0 ? i - 1 : 0] - mean;
            v = (cur - preemph * prev) * window[i];
        }
        const int j = bitrev9(i);
        re[j] = v; im[j] = 0.f;
    }
    __syncthreads();
    // radix-2 decimation in time: 9 stages, one butterfly per thread
#pragma unroll
    for (int st = 0; st < 9; ++st) {
        const int half = 1 << st;
        const int pos = tid & (half - 1);
        const int i0 = ((tid >> st) << (st + 1)) + pos, i1 = i0 + half;
        const int tw = pos << (8 - st);
        const float wr = twr[tw], wi = twi[tw];
        const float ar = re[i0], ai = im[i0], br = re[i1], bi = im[i1];
        const float tr = br * wr - bi * wi, ti = br * wi + bi * wr;
        re[i0] = ar + tr; im[i0] = ai + ti; re[i1] = ar - tr; im[i1] = ai - ti;
        __syncthreads();
    }
    // power spectrum, bins 0..256, kept in re[]
    float p0 = re[tid] * re[tid] + im[tid] * im[tid];
    float p256 = 0.f;
    if (tid == 0) p256 = re[256] * re[256] + im[256] * im[256];
    __syncthreads();
    re[tid] = p0;
    if (tid == 0) re[256] = p256;
    __syncthreads();
    if (tid < FB_MEL) {
        float acc = 0.f;
        for (int k = 0; k < FB_BINS; ++k) acc += re[k] * mel_t[k * FB_MEL + tid];
        const float e = logf(fmaxf(acc, 1.1920928955078125e-07f));
        out[((long)blockIdx.y * frames + f) * FB_MEL + tid] = (e - out_sub) * out_scale;
    }
}

double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

}  // namespace

extern "C" int crab_bicubic_ksize(int in_size, int out_size) {
    if (in_size <= 0 || out_size <= 0) return CRAB_E_INVALID;
    double scale = (double)in_size / out_size;
    double filterscale = scale < 1.0 ? 1.0 : scale;
    return (int)ceil(2.0 * filterscale) * 2 + 1;
}

extern "C" int crab_bicubic_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk, int ksize) {
    if (in_size <= 0 || out_size <= 0 || !bounds || !kk || ksize < crab_bicubic_ksize(in_size, out_size)) return CRAB_E_INVALID;
    const double scale = (double)in_size / out_size;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale, ss = 1.0 / filterscale;
    double* k = (double*)malloc(sizeof(double) * ksize);
    if (!k) return CRAB_E_INVALID;
    for (int xx = 0; xx < out_size; ++xx) {
        const double center = (xx + 0.5) * scale;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        double ww = 0.0;
        for (int x = 0; x < xmax; ++x) { k[x] = bicubic_filter((x + xmin - center + 0.5) * ss); ww += k[x]; }
        for (int x = 0; x < ksize; ++x) {
            double v = x < xmax ? (ww != 0.0 ? k[x] / ww : k[x]) : 0.0;
            kk[(long)xx * ksize + x] = v < 0 ? (int32_t)(-0.5 + v * (1 << PRECISION_BITS)) : (int32_t)(0.5 + v * (1 << PRECISION_BITS));
        }
        bounds[2 * xx] = xmin; bounds[2 * xx + 1] = xmax;
    }
    free(k);
    return ksize;
}

extern "C" int crab_resample_u8(crab_ctx* ctx, void* stream, const void* src, int N, int H, int W, int C, void* dst, int out_size,
                                int horizontal, const int32_t* bounds, const int32_t* kk, int ksize) {
    if (!ctx) return CRAB_E_INVALID;
    if (!src || !dst || !bounds || !kk || N <= 0 || H <= 0 || W <= 0 || out_size <= 0 || ksize <= 0) return crab_fail(ctx, CRAB_E_INVALID, "resample_u8: bad argument");
    if (C != 3 && C != 1) return crab_fail(ctx, CRAB_E_UNSUPPORTED, "resample_u8: 1 or 3 channels");
    const long npix = horizontal ? (long)H * out_size : (long)out_size * W;
    dim3 grid((unsigned)((npix + 255) / 256), N), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (C == 3) hipLaunchKernelGGL((resample_u8_kernel<3>), grid, block, 0, s, (const uint8_t*)src, (uint8_t*)dst, H, W, out_size, horizontal, bounds, kk, ksize);
    else hipLaunchKernelGGL((resample_u8_kernel<1>), grid, block, 0, s, (const uint8_t*)src, (uint8_t*)dst, H, W, out_size, horizontal, bounds, kk, ksize);
    return crab_check_launch(ctx, "resample_u8_kernel");
}

extern "C" int crab_clip_normalize(crab_ctx* ctx, void* stream, const void* src, int N, int H, int W, int top, int left, int size, void* out,
                                   int out_bf16, const float* mean3, const float* std3, float rescale) {
    if (!ctx) return CRAB_E_INVALID;
    if (!src || !out || !mean3 || !std3 || N <= 0 || size <= 0 || top < 0 || left < 0 || top + size > H || left + size > W)
        return crab_fail(ctx, CRAB_E_INVALID, "clip_normalize: bad argument / crop outside the image");
    dim3 grid((size * size + 255) / 256, N), block(256);
    hipLaunchKernelGGL(clip_normalize_kernel, grid, block, 0, (hipStream_t)stream, (const uint8_t*)src, H, W, top, left, size, out, out_bf16,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2], rescale);
    return crab_check_launch(ctx, "clip_normalize_kernel");
}

extern "C" int crab_kaldi_fbank_frames(int n_samples) { return n_samples < FB_WIN ? 0 : 1 + (n_samples - FB_WIN) / FB_SHIFT; }

extern "C" int crab_kaldi_fbank(crab_ctx* ctx, void* stream, const float* wave, int64_t ldw, int n_wave, int n_samples, float in_scale,
                                float preemphasis, const float* window400, const float* mel_t, float* out, float out_sub, float out_scale) {
    if (!ctx) return CRAB_E_INVALID;
    if (!wave || !window400 || !mel_t || !out || n_wave <= 0 || ldw < n_samples) return crab_fail(ctx, CRAB_E_INVALID, "kaldi_fbank: bad argument");
    const int frames = crab_kaldi_fbank_frames(n_samples);
    if (frames <= 0) return crab_fail(ctx, CRAB_E_INVALID, "kaldi_fbank: fewer than 400 samples");
    dim3 grid(frames, n_wave), block(256);
    hipLaunchKernelGGL(kaldi_fbank_kernel, grid, block, 0, (hipStream_t)stream, wave, (long)ldw, n_samples, in_scale, preemphasis, window400,
                       mel_t, out, frames, out_sub, out_scale);
    return crab_check_launch(ctx, "kaldi_fbank_kernel");
}

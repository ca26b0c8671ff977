// Segmentation metrics of the reference's pixel-task eval loops (SURVEY.md 8 f-1 anchors utils/avss_utils.py:8-96, 379-435): what the loops
// compute from a predicted mask right behind generate_avs - after moving the mask to the host (scripts/quick_start.py:118-119, 198-199,
// 267-268 mask_iou + Eval_Fmeasure; :342 metric_s_for_null; :395 calc_color_miou_fscore).  Here the masks stay where SegModule left them.
//
// HBM-bound integer work: every kernel is one pass over the mask planes that ends in pixel COUNTS (int32, exact, order-independent: LDS /
// global integer atomics), and a single-block finishing launch that forms the fp32 ratios from the counts in the reference's operation
// order (helpers compiled with contraction off: no a*b + c fused into an FMA, the reference's CPU tensor ops round each step).
//   mask_iou            counts per image {pred, target, inter, union, (1-t)(1-p), non-binary target}  ->  sum_n inter/(union + eps) / N, and
//                       metric_s_for_null's sqrt(pred / pixels)                                                (avss_utils.py:8-47)
//   Eval_Fmeasure       per image: histogram of sigmoid(pred) over the pr_num thresholds (binary search in LDS), suffix sums = the
//                       reference's `(y_pred >= th_i)` counts for every i at once (255 passes over the image there), precision / recall /
//                       F per threshold, mean over the images with a non-empty ground truth, max over thresholds   (avss_utils.py:50-96)
//   _batch_miou_fscore  per frame: argmax over the class planes, three class histograms (TP, TP + FP, TP + FN) = the reference's three
//                       torch.histc calls, per-class IoU / F sums in frame order, per-frame mean IoU            (avss_utils.py:379-435)
// Thresholds on sigmoid(pred): `sigmoid(pred) > 0.5` is evaluated as pred > 0 (as crab_mask_labels does for the PNG: the fp32 sigmoid of the
// reference's CPU build and the correctly rounded one disagree with EACH OTHER on logits in (0, 2^-22), nowhere else); the 255-threshold
// comparison uses sigmoid in fp64 rounded once to fp32 (within 1 ulp of whatever vectorised expf the reference's host runs).
#include "common.h"
#include "crab_internal.h"
#include <math.h>

namespace {

constexpr int SM_MAXT = 1024;                                   // thresholds / classes a block keeps in LDS

// The finishing arithmetic rounds after every operation, like the reference's CPU tensor ops: hipcc's default -ffp-contract=fast would fuse
// b2 * P + R into one FMA (HIP's __fadd_rn / __fmul_rn are plain operators in a header compiled under that default, so they do not stop it);
// these helpers are compiled with contraction off.  Division and square root are correctly rounded by default on this toolchain
// (-fhip-fp32-correctly-rounded-divide-sqrt).
#pragma clang fp contract(off)
__device__ __forceinline__ float add_rn(float a, float b) { return a + b; }
__device__ __forceinline__ float sub_rn(float a, float b) { return a - b; }
__device__ __forceinline__ float mul_rn(float a, float b) { return a * b; }
__device__ __forceinline__ float div_rn(float a, float b) { return a / b; }

__device__ __forceinline__ float sigmoid_rn(float x) { return (float)(1.0 / (1.0 + exp(-(double)x))); }

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// counts[n][6] += {pred, target != 0, pred & target, pred | target, !pred & !target, target not in {0, 1}} over a slice of image n.
// V4: four pixels per 16-byte load (hw % 4 == 0 and 16-byte aligned planes: the 224 x 224 masks of the eval loops).
template <bool V4>
__global__ __launch_bounds__(256) void mask_counts_kernel(const float* __restrict__ pred, const float* __restrict__ target, long hw, int* __restrict__ counts) {
    const int n = blockIdx.y;
    const float* p = pred + (long)n * hw;
    const float* t = target ? target + (long)n * hw : nullptr;
    int c[6] = {0, 0, 0, 0, 0, 0};
    auto one = [&](float pv, float tv) {
        const int pb = pv > 0.0f ? 1 : 0, tb = tv != 0.0f ? 1 : 0;
        c[5] += (tv != 0.0f && tv != 1.0f) ? 1 : 0;
        c[0] += pb; c[1] += tb; c[2] += pb & tb; c[3] += pb | tb; c[4] += (1 - pb) & (1 - tb);
    };
    const long step = (long)gridDim.x * blockDim.x, i0 = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (V4) {
        const long nv = hw >> 2;
        for (long i = i0; i < nv; i += step) {
            const f32x4_t pv = reinterpret_cast<const f32x4_t*>(p)[i];
            f32x4_t tv = {0.f, 0.f, 0.f, 0.f};
            if (t) tv = reinterpret_cast<const f32x4_t*>(t)[i];
#pragma unroll
            for (int k = 0; k < 4; ++k) one(pv[k], tv[k]);
        }
    } else {
        for (long i = i0; i < hw; i += step) one(p[i], t ? t[i] : 0.0f);
    }
    // one global atomic per (block, counter): per-wave atomics serialised on the 6 words of an image (25 blocks x 4 waves each: 20 of the launch's 33 us)
    __shared__ int blk[6];
    if (threadIdx.x < 6) blk[threadIdx.x] = 0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int s = wave_sum(c[k]);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(&blk[k], s);
    }
    __syncthreads();
    if (threadIdx.x < 6 && blk[threadIdx.x]) atomicAdd(&counts[n * 6 + threadIdx.x], blk[threadIdx.x]);
}

// out[0] = sum_n inter_n / (union_n + eps) / N  with the empty-target substitution (avss_utils.py:35-45); out[1] = sqrt(sum_n pred_n / (N hw)).
// The per-image ratios are formed by the block's threads side by side (a single thread walking the images pays one memory latency per image:
// 64 images took 40 us); thread 0 then adds them in image order, chunk after chunk.
__global__ __launch_bounds__(256) void mask_iou_finish_kernel(const int* __restrict__ counts, int N, long hw, float eps, int has_target, float* __restrict__ out) {
    __shared__ float term[256];
    __shared__ long long px;
    const int tid = threadIdx.x;
    if (tid == 0) px = 0;
    __syncthreads();
    float acc = 0.f;
    for (int n0 = 0; n0 < N; n0 += 256) {
        const int n = n0 + tid;
        if (n < N) {
            const int* c = counts + n * 6;
            float inter = (float)c[2], uni = (float)c[3];
            if (c[1] == 0) { inter = (float)c[4]; uni = (float)hw; }
            term[tid] = div_rn(inter, add_rn(uni, eps));
            atomicAdd((unsigned long long*)&px, (unsigned long long)c[0]);
        }
        __syncthreads();
        if (tid == 0 && has_target)
            for (int k = 0; k < min(256, N - n0); ++k) acc = add_rn(acc, term[k]);
        __syncthreads();
    }
    if (tid == 0) {
        out[0] = has_target ? div_rn(acc, (float)N) : 0.f;
        out[1] = sqrtf(div_rn((float)px, (float)((long)N * hw)));
    }
}

// Pass 1, FM_SPLIT blocks per image: the histogram of sigmoid(pred) over the threshold table.  Bin k = #{j : th_j <= s} in [0, T]; bin 0 (below
// every threshold) is counted nowhere, bins 1..T go to ge[n][1][k-1] (all pixels) and ge[n][0][k-1] (gt pixels) through an LDS histogram and one
// global atomic per non-empty bin and block; ysum[n] += {gt pixels, gt pixels outside {0, 1}}.  ge / ysum are zeroed by the caller (memset).
// (One block per image did all of this in 58 us at 224 x 224 - fp64 sigmoid + LDS atomics of 49 pixels per thread in a row.)
constexpr int FM_SPLIT_MAX = 16;
__global__ __launch_bounds__(256) void fmeasure_hist_kernel(const float* __restrict__ pred, const float* __restrict__ gt, long hw,
                                                            const float* __restrict__ thresholds, int T, int* __restrict__ ge, int* __restrict__ ysum) {
    __shared__ float th[SM_MAXT];
    __shared__ int cnt[SM_MAXT + 1], tp[SM_MAXT + 1];
    __shared__ int ys, bad;
    const int n = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < T; i += blockDim.x) th[i] = thresholds[i];
    for (int i = tid; i <= T; i += blockDim.x) { cnt[i] = 0; tp[i] = 0; }
    if (tid == 0) { ys = 0; bad = 0; }
    __syncthreads();
    const float* p = pred + (long)n * hw;
    const float* g = gt + (long)n * hw;
    int my_y = 0, my_bad = 0;
    for (long i = (long)blockIdx.x * blockDim.x + tid; i < hw; i += (long)gridDim.x * blockDim.x) {
        const float s = sigmoid_rn(p[i]);
        const float gv = g[i];
        int lo = 0, hi = T;                                     // k = #{j : th_j <= s}  (ascending thresholds; NaN -> 0, as `NaN >= th` is false)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (th[mid] <= s) lo = mid + 1; else hi = mid;
        }
        atomicAdd(&cnt[lo], 1);
        if (gv != 0.0f) { atomicAdd(&tp[lo], 1); ++my_y; if (gv != 1.0f) ++my_bad; }
    }
    my_y = wave_sum(my_y); my_bad = wave_sum(my_bad);
    if ((tid & 63) == 0) { if (my_y) atomicAdd(&ys, my_y); if (my_bad) atomicAdd(&bad, my_bad); }
    __syncthreads();
    if (tid == 0) { if (ys) atomicAdd(&ysum[n * 2], ys); if (bad) atomicAdd(&ysum[n * 2 + 1], bad); }
    for (int k = 1 + tid; k <= T; k += blockDim.x) {
        if (tp[k]) atomicAdd(&ge[((long)n * 2 + 0) * T + k - 1], tp[k]);
        if (cnt[k]) atomicAdd(&ge[((long)n * 2 + 1) * T + k - 1], cnt[k]);
    }
}

// Pass 2, one block per image: the bins become suffix sums in place - ge[n][0][i] = #{gt & sigmoid(pred) >= th_i}, ge[n][1][i] = #{sigmoid(pred) >= th_i}
// = the reference's 255 `(y_pred >= th_i)` sweeps - and fscore[n][i] = (1 + b2) P R / (b2 P + R), NaN -> 0, with P = tp / (count + 1e-20),
// R = tp / (gt pixels + 1e-20)      (avss_utils.py:50-64, 88-89)
__global__ __launch_bounds__(1024) void fmeasure_image_kernel(int T, float one_b2, float b2, int* __restrict__ ge, const int* __restrict__ ysum,
                                                              float* __restrict__ fscore) {
    __shared__ int cnt[SM_MAXT], tp[SM_MAXT];
    const int n = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < T; i += blockDim.x) { tp[i] = ge[((long)n * 2 + 0) * T + i]; cnt[i] = ge[((long)n * 2 + 1) * T + i]; }
    const int ys = ysum[n * 2];
    __syncthreads();
    for (int i = tid; i < T; i += blockDim.x) {
        int c = 0, t = 0;
        for (int k = i; k < T; ++k) { c += cnt[k]; t += tp[k]; }                           // bins i+1 .. T live at indices i .. T-1
        ge[((long)n * 2 + 0) * T + i] = t;
        ge[((long)n * 2 + 1) * T + i] = c;
        const float ft = (float)t;
        const float prec = div_rn(ft, add_rn((float)c, 1e-20f));
        const float rec = div_rn(ft, add_rn((float)ys, 1e-20f));
        float f = div_rn(mul_rn(mul_rn(one_b2, prec), rec), add_rn(mul_rn(b2, prec), rec));
        if (f != f) f = 0.f;
        fscore[(long)n * T + i] = f;
    }
}

// score[i] = sum over the images with a non-empty gt (in image order) of fscore[n][i] / their number; best = {max_i score[i], that number}
__global__ __launch_bounds__(1024) void fmeasure_finish_kernel(const float* __restrict__ fscore, const int* __restrict__ ysum, int N, int T,
                                                               float* __restrict__ score, float* __restrict__ best) {
    __shared__ float mx[1024];
    const int tid = threadIdx.x;
    float m = 0.f;                                              // scores are >= 0 (NaN already 0); no image counted -> zeros(pr_num).max() = 0
    __shared__ int cnt_s;
    if (tid == 0) cnt_s = 0;
    __syncthreads();
    for (int n = tid; n < N; n += blockDim.x)
        if (ysum[n * 2] > 0) atomicAdd(&cnt_s, 1);
    __syncthreads();
    const int cnt = cnt_s;
    for (int i = tid; i < T; i += blockDim.x) {
        float acc = 0.f;
        for (int n0 = 0; n0 < N; n0 += 8) {                     // eight images' loads in flight, then the adds in image order
            float v[8];
            int ok[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = min(n0 + j, N - 1);
                v[j] = fscore[(long)n * T + i];
                ok[j] = ysum[n * 2] > 0;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (n0 + j < N && ok[j]) acc = add_rn(acc, v[j]);
        }
        const float s = cnt ? div_rn(acc, (float)cnt) : 0.f;
        score[i] = s;
        m = fmaxf(m, s);
    }
    mx[tid] = m;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) mx[tid] = fmaxf(mx[tid], mx[tid + o]);
        __syncthreads();
    }
    if (tid == 0) { best[0] = mx[0]; best[1] = (float)cnt; }
}

// areas[f][0][c] += #{argmax == c == target}, [f][1][c] += #{argmax == c, target >= 0}, [f][2][c] += #{target == c} over a slice of frame f.
// The reference shifts both maps by one, zeroes the prediction where target + 1 <= 0, and counts with histc over [1, nclass]: values outside
// that range (a negative or >= nclass label) fall out of every histogram (avss_utils.py:386-402).
// PX pixels per thread (PX = 4: one 16-byte load per class plane, hw % 4 == 0), the class loop unrolled 8 deep so that eight plane loads are
// in flight per thread, marked non-temporal (read once) - a pixel's C values are C separate planes, [C][hw]: with one pixel and one load at a time the pass ran at 3.0 TB/s, vectorised at 5.8, streaming at 6.7.
template <int PX>
__global__ __launch_bounds__(256) void class_areas_kernel(const float* __restrict__ pred, const long long* __restrict__ target, int C, long hw,
                                                          int* __restrict__ areas) {
    __shared__ int h[3][SM_MAXT];
    const int f = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < 3 * SM_MAXT; i += blockDim.x) (&h[0][0])[i] = 0;
    __syncthreads();
    const float* p = pred + (long)f * C * hw;
    const long long* t = target + (long)f * hw;
    typedef __attribute__((ext_vector_type(PX))) float vec_t;
    const long nv = hw / PX;
    for (long i = (long)blockIdx.x * blockDim.x + tid; i < nv; i += (long)gridDim.x * blockDim.x) {
        float best[PX];
        int bi[PX];
        {
            const vec_t v = *reinterpret_cast<const vec_t*>(p + i * PX);
#pragma unroll
            for (int k = 0; k < PX; ++k) { best[k] = PX == 1 ? ((const float*)&v)[0] : v[k]; bi[k] = 0; }
        }
        int c = 1;
        for (; c + 8 <= C; c += 8) {
            vec_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(reinterpret_cast<const vec_t*>(p + (long)(c + j) * hw + i * PX));   // read once: streaming loads
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int k = 0; k < PX; ++k) {
                    const float x = PX == 1 ? ((const float*)&v[j])[0] : v[j][k];
                    if (x > best[k] || (x != x && best[k] == best[k])) { best[k] = x; bi[k] = c + j; }      // (a NaN is the maximum, as for torch.argmax: the first one wins)
                }
        }
        for (; c < C; ++c) {
            const vec_t v = *reinterpret_cast<const vec_t*>(p + (long)c * hw + i * PX);
#pragma unroll
            for (int k = 0; k < PX; ++k) {
                const float x = PX == 1 ? ((const float*)&v)[0] : v[k];
                if (x > best[k] || (x != x && best[k] == best[k])) { best[k] = x; bi[k] = c; }
            }
        }
#pragma unroll
        for (int k = 0; k < PX; ++k) {
            const long long tv = t[i * PX + k];
            if (tv >= 0) {
                atomicAdd(&h[1][bi[k]], 1);
                if (tv < C) {
                    atomicAdd(&h[2][(int)tv], 1);
                    if (tv == bi[k]) atomicAdd(&h[0][bi[k]], 1);
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < 3 * C; i += blockDim.x) {
        const int k = i / C, c = i % C, v = h[k][c];
        if (v) atomicAdd(&areas[((long)f * 3 + k) * C + c], v);
    }
}

// One block.  Frames are taken in chunks of floor(4096 / C): every (frame, class) pair of a chunk gets its IoU, F and union flag from one thread
// (the correctly rounded divisions are ~100 dependent instructions per pair: with one thread per CLASS walking 64 frames the launch took 38 us
// on 71 busy lanes), staged in LDS; then thread c adds its class over the chunk's frames IN FRAME ORDER and thread f adds its frame over the classes
// in class order - the reference's `ious += iou` per frame and `torch.sum(iou)` restated as index-order sums.
constexpr int MF_PAIRS = 4096;
__global__ __launch_bounds__(1024) void miou_finish_kernel(const int* __restrict__ areas, int BF, int C, float one_b2, float b2, float* __restrict__ iou_fc,
                                                           float* __restrict__ ious, float* __restrict__ fscores, float* __restrict__ cls_count,
                                                           float* __restrict__ vid_miou) {
    __shared__ float iou_s[MF_PAIRS], fs_s[MF_PAIRS], un_s[MF_PAIRS];
    const int tid = threadIdx.x, nthr = blockDim.x;
    const int chunk = min(MF_PAIRS / C, nthr);                  // >= 4 frames (C <= 1024), at most one frame per thread
    // per-class running sums live in the registers of thread c (c = tid, tid + nthr, ...: C <= 1024 = nthr, so one class per thread)
    float si = 0.f, sf = 0.f, sc = 0.f;
    for (int f0 = 0; f0 < BF; f0 += chunk) {
        const int nf = min(chunk, BF - f0);
        for (int q = tid; q < nf * C; q += nthr) {
            const int fi = q / C, c = q % C, f = f0 + fi;
            const float ai = (float)areas[((long)f * 3 + 0) * C + c], ap = (float)areas[((long)f * 3 + 1) * C + c], al = (float)areas[((long)f * 3 + 2) * C + c];
            const float au = sub_rn(add_rn(ap, al), ai);
            const float iou = div_rn(ai, add_rn(2.220446049250313e-16f, au));
            const float prec = div_rn(ai, ap), rec = div_rn(ai, al);                    // 0 / 0 = NaN, as in the reference
            float fs = div_rn(mul_rn(mul_rn(one_b2, prec), rec), add_rn(mul_rn(b2, prec), rec));
            if (fs != fs) fs = 0.f;
            iou_fc[(long)f * C + c] = iou;
            iou_s[q] = iou; fs_s[q] = fs; un_s[q] = au != 0.f ? 1.f : 0.f;
        }
        __syncthreads();
        if (tid < C)
            for (int fi = 0; fi < nf; ++fi) {
                si = add_rn(si, iou_s[fi * C + tid]);
                sf = add_rn(sf, fs_s[fi * C + tid]);
                if (un_s[fi * C + tid] != 0.f) sc = add_rn(sc, 1.f);
            }
        // the chunk's frames: one thread per frame, taken from the top of the block so that they do not queue behind the class threads' waves
        const int fi = nthr - 1 - tid;
        if (fi < nf) {
            float s = 0.f;
            int nz = 0;
            for (int c = 0; c < C; ++c) {
                const float v = iou_s[fi * C + c];
                s = add_rn(s, v);
                nz += v != 0.f ? 1 : 0;
            }
            vid_miou[f0 + fi] = div_rn(s, (float)nz);           // no class with a non-zero IoU: 0 / 0 = NaN, as torch.sum(iou) / 0
        }
        __syncthreads();
    }
    if (tid < C) { ious[tid] = si; fscores[tid] = sf; cls_count[tid] = sc; }
}

// AVSS ground truth, colour map -> class ids (dataset/quick_start_dataset.py:63-73 color_mask_to_label, called on the PIL mask right before it
// becomes X_modals['<mask>'], :534-539): out[p] = the FIRST palette entry equal to the pixel's (r, g, b), 0 when none is (the reference stacks one
// equality plane per colour and takes the argmax, which is 0 for an all-zero row).  One thread per pixel, the palette (<= 256 x 3 bytes) in LDS.
__global__ __launch_bounds__(256) void color_to_label_kernel(const uint8_t* __restrict__ rgb, long hw, const uint8_t* __restrict__ palette, int n,
                                                             long long* __restrict__ out) {
    __shared__ uint32_t pal[256];
    for (int i = threadIdx.x; i < n; i += blockDim.x) pal[i] = (uint32_t)palette[i * 3] | ((uint32_t)palette[i * 3 + 1] << 8) | ((uint32_t)palette[i * 3 + 2] << 16);
    __syncthreads();
    const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= hw) return;
    const uint32_t v = (uint32_t)rgb[p * 3] | ((uint32_t)rgb[p * 3 + 1] << 8) | ((uint32_t)rgb[p * 3 + 2] << 16);
    int lab = 0;
    for (int i = n - 1; i >= 0; --i)
        if (pal[i] == v) lab = i;                               // descending: the lowest matching index survives
    out[p] = lab;
}

}  // namespace

#define S_(x) ((hipStream_t)(x))

extern "C" {

int crab_mask_iou(crab_ctx* ctx, void* stream, const float* pred, const float* target, int N, int64_t hw, float eps, int32_t* counts, float* out) {
    if (!ctx) return CRAB_E_INVALID;
    if (!pred || !counts || !out || N <= 0 || hw <= 0 || (int64_t)N * hw >= ((int64_t)1 << 31))
        return crab_fail(ctx, CRAB_E_INVALID, "mask_iou: pred [N, hw] fp32, counts [N, 6] int32, out [2] fp32, N hw < 2^31");
    CRAB_HIP_TRY(ctx, hipMemsetAsync(counts, 0, (size_t)N * 6 * sizeof(int32_t), S_(stream)));
    const bool v4 = (hw & 3) == 0 && (((uintptr_t)pred | (uintptr_t)target) & 15) == 0;
    long bx = (hw + 256 * 16 - 1) / (256 * 16);                // 16 pixels per thread (four 16-byte loads per plane in the vector form)
    if (bx > 1024) bx = 1024;
    if (v4) hipLaunchKernelGGL((mask_counts_kernel<true>), dim3((unsigned)bx, (unsigned)N), dim3(256), 0, S_(stream), pred, target, (long)hw, counts);
    else hipLaunchKernelGGL((mask_counts_kernel<false>), dim3((unsigned)bx, (unsigned)N), dim3(256), 0, S_(stream), pred, target, (long)hw, counts);
    int rc = crab_check_launch(ctx, "mask_counts_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(mask_iou_finish_kernel, dim3(1), dim3(256), 0, S_(stream), (const int*)counts, N, (long)hw, eps, target ? 1 : 0, out);
    return crab_check_launch(ctx, "mask_iou_finish_kernel");
}

int crab_fmeasure(crab_ctx* ctx, void* stream, const float* pred, const float* gt, int N, int64_t hw, const float* thresholds, int T, double beta2,
                  int32_t* ge, int32_t* ysum, float* fscore, float* score, float* best) {
    if (!ctx) return CRAB_E_INVALID;
    if (!pred || !gt || !thresholds || !ge || !ysum || !fscore || !score || !best || N <= 0 || hw <= 0 || hw >= ((int64_t)1 << 31) || T <= 0 || T > SM_MAXT)
        return crab_fail(ctx, CRAB_E_INVALID, "fmeasure: pred / gt [N, hw] fp32, 1 <= T <= 1024 ascending thresholds, hw < 2^31");
    const float one_b2 = (float)(1.0 + beta2), b2 = (float)beta2;
    CRAB_HIP_TRY(ctx, hipMemsetAsync(ge, 0, (size_t)N * 2 * T * sizeof(int32_t), S_(stream)));
    CRAB_HIP_TRY(ctx, hipMemsetAsync(ysum, 0, (size_t)N * 2 * sizeof(int32_t), S_(stream)));
    long split = (hw + 256 * 16 - 1) / (256 * 16);             // ~16 pixels per thread; a few images fill the chip through the split, many through N
    if (split > FM_SPLIT_MAX) split = FM_SPLIT_MAX;
    hipLaunchKernelGGL(fmeasure_hist_kernel, dim3((unsigned)split, (unsigned)N), dim3(256), 0, S_(stream), pred, gt, (long)hw, thresholds, T, (int*)ge, (int*)ysum);
    int rc = crab_check_launch(ctx, "fmeasure_hist_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(fmeasure_image_kernel, dim3((unsigned)N), dim3(1024), 0, S_(stream), T, one_b2, b2, (int*)ge, (const int*)ysum, fscore);
    rc = crab_check_launch(ctx, "fmeasure_image_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(fmeasure_finish_kernel, dim3(1), dim3(1024), 0, S_(stream), (const float*)fscore, (const int*)ysum, N, T, score, best);
    return crab_check_launch(ctx, "fmeasure_finish_kernel");
}

int crab_miou_fscore(crab_ctx* ctx, void* stream, const float* pred, const int64_t* target, int BF, int C, int64_t hw, double beta2, int32_t* areas,
                     float* iou_fc, float* ious, float* fscores, float* cls_count, float* vid_miou) {
    if (!ctx) return CRAB_E_INVALID;
    if (!pred || !target || !areas || !iou_fc || !ious || !fscores || !cls_count || !vid_miou || BF <= 0 || C <= 0 || C > SM_MAXT || hw <= 0 ||
        hw >= ((int64_t)1 << 31))
        return crab_fail(ctx, CRAB_E_INVALID, "miou_fscore: pred [BF, C, hw] fp32, target [BF, hw] int64, 1 <= C <= 1024, hw < 2^31");
    CRAB_HIP_TRY(ctx, hipMemsetAsync(areas, 0, (size_t)BF * 3 * C * sizeof(int32_t), S_(stream)));
    const bool v4 = (hw & 3) == 0 && ((uintptr_t)pred & 15) == 0;
    long bx = ((v4 ? hw / 4 : hw) + 255) / 256;                // one pixel group per thread: it already costs C plane loads
    if (bx > 4096) bx = 4096;
    if (v4) hipLaunchKernelGGL((class_areas_kernel<4>), dim3((unsigned)bx, (unsigned)BF), dim3(256), 0, S_(stream), pred, (const long long*)target, C, (long)hw, (int*)areas);
    else hipLaunchKernelGGL((class_areas_kernel<1>), dim3((unsigned)bx, (unsigned)BF), dim3(256), 0, S_(stream), pred, (const long long*)target, C, (long)hw, (int*)areas);
    int rc = crab_check_launch(ctx, "class_areas_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(miou_finish_kernel, dim3(1), dim3(1024), 0, S_(stream), (const int*)areas, BF, C, (float)(1.0 + beta2), (float)beta2, iou_fc, ious, fscores,
                       cls_count, vid_miou);
    return crab_check_launch(ctx, "miou_finish_kernel");
}

int crab_color_to_label(crab_ctx* ctx, void* stream, const uint8_t* rgb, int64_t hw, const uint8_t* palette, int n, int64_t* out) {
    if (!ctx) return CRAB_E_INVALID;
    if (!rgb || !palette || !out || hw <= 0 || n <= 0 || n > 256) return crab_fail(ctx, CRAB_E_INVALID, "color_to_label: rgb [hw, 3] uint8, palette [n <= 256, 3] uint8, out [hw] int64");
    hipLaunchKernelGGL(color_to_label_kernel, dim3((unsigned)((hw + 255) / 256)), dim3(256), 0, S_(stream), rgb, (long)hw, palette, n, (long long*)out);
    return crab_check_launch(ctx, "color_to_label_kernel");
}

}  // extern "C"

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

// counts[n][6] += {pred, target != 0, pred & target, pred | target, !pred & !target, target not in {0, 1}} over a slice of image n
__global__ __launch_bounds__(256) void mask_counts_kernel(const float* __restrict__ pred, const float* __restrict__ target, long hw, int* __restrict__ counts) {
    const int n = blockIdx.y;
    const float* p = pred + (long)n * hw;
    const float* t = target ? target + (long)n * hw : nullptr;
    int c[6] = {0, 0, 0, 0, 0, 0};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (long)gridDim.x * blockDim.x) {
        const int pb = p[i] > 0.0f ? 1 : 0;
        int tb = 0;
        if (t) {
            const float tv = t[i];
            tb = tv != 0.0f ? 1 : 0;
            c[5] += (tv != 0.0f && tv != 1.0f) ? 1 : 0;
        }
        c[0] += pb; c[1] += tb; c[2] += pb & tb; c[3] += pb | tb; c[4] += (1 - pb) & (1 - tb);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int s = wave_sum(c[k]);
        if ((threadIdx.x & 63) == 0 && s) atomicAdd(&counts[n * 6 + k], s);
    }
}

// out[0] = sum_n inter_n / (union_n + eps) / N  with the empty-target substitution (avss_utils.py:35-45); out[1] = sqrt(sum_n pred_n / (N hw))
__global__ void mask_iou_finish_kernel(const int* __restrict__ counts, int N, long hw, float eps, int has_target, float* __restrict__ out) {
    if (threadIdx.x || blockIdx.x) return;
    float acc = 0.f;
    long px = 0;
    for (int n = 0; n < N; ++n) {
        const int* c = counts + n * 6;
        px += c[0];
        if (has_target) {
            float inter = (float)c[2], uni = (float)c[3];
            if (c[1] == 0) { inter = (float)c[4]; uni = (float)hw; }
            acc = add_rn(acc, div_rn(inter, add_rn(uni, eps)));
        }
    }
    out[0] = has_target ? div_rn(acc, (float)N) : 0.f;
    out[1] = sqrtf(div_rn((float)px, (float)((long)N * hw)));
}

// One block per image.  ge[n][0][i] = #{gt & sigmoid(pred) >= th_i}, ge[n][1][i] = #{sigmoid(pred) >= th_i}; ysum[n] = {gt pixels, gt not in {0,1}};
// fscore[n][i] = (1 + b2) P R / (b2 P + R), NaN -> 0, with P = tp / (count + 1e-20), R = tp / (gt pixels + 1e-20)      (avss_utils.py:50-64, 88-89)
__global__ __launch_bounds__(1024) void fmeasure_image_kernel(const float* __restrict__ pred, const float* __restrict__ gt, long hw,
                                                              const float* __restrict__ thresholds, int T, float one_b2, float b2,
                                                              int* __restrict__ ge, int* __restrict__ ysum, float* __restrict__ fscore) {
    __shared__ float th[SM_MAXT];
    __shared__ int cnt[SM_MAXT + 1], tp[SM_MAXT + 1];
    __shared__ int ys, bad;
    const int n = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < T; i += blockDim.x) th[i] = thresholds[i];
    for (int i = tid; i <= T; i += blockDim.x) { cnt[i] = 0; tp[i] = 0; }
    if (tid == 0) { ys = 0; bad = 0; }
    __syncthreads();
    const float* p = pred + (long)n * hw;
    const float* g = gt + (long)n * hw;
    int my_y = 0, my_bad = 0;
    for (long i = tid; i < hw; i += blockDim.x) {
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
    if (tid == 0) { ysum[n * 2] = ys; ysum[n * 2 + 1] = bad; }
    for (int i = tid; i < T; i += blockDim.x) {
        int c = 0, t = 0;
        for (int k = i + 1; k <= T; ++k) { c += cnt[k]; t += tp[k]; }
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
    int cnt = 0;
    for (int n = 0; n < N; ++n) cnt += ysum[n * 2] > 0 ? 1 : 0;
    for (int i = tid; i < T; i += blockDim.x) {
        float acc = 0.f;
        for (int n = 0; n < N; ++n)
            if (ysum[n * 2] > 0) acc = add_rn(acc, fscore[(long)n * T + i]);
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
__global__ __launch_bounds__(256) void class_areas_kernel(const float* __restrict__ pred, const long long* __restrict__ target, int C, long hw,
                                                          int* __restrict__ areas) {
    __shared__ int h[3][SM_MAXT];
    const int f = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < 3 * SM_MAXT; i += blockDim.x) (&h[0][0])[i] = 0;
    __syncthreads();
    const float* p = pred + (long)f * C * hw;
    const long long* t = target + (long)f * hw;
    for (long i = (long)blockIdx.x * blockDim.x + tid; i < hw; i += (long)gridDim.x * blockDim.x) {
        float best = p[i];
        int bi = 0;
        for (int c = 1; c < C; ++c) {
            const float v = p[(long)c * hw + i];
            if (v > best) { best = v; bi = c; }
        }
        const long long tv = t[i];
        if (tv >= 0) {
            atomicAdd(&h[1][bi], 1);
            if (tv < C) {
                atomicAdd(&h[2][(int)tv], 1);
                if (tv == bi) atomicAdd(&h[0][bi], 1);
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < 3 * C; i += blockDim.x) {
        const int k = i / C, c = i % C, v = h[k][c];
        if (v) atomicAdd(&areas[((long)f * 3 + k) * C + c], v);
    }
}

// Thread c: the per-class sums over the frames in frame order; then thread f: the frame's mean IoU over its classes with a non-zero IoU.
__global__ __launch_bounds__(1024) void miou_finish_kernel(const int* __restrict__ areas, int BF, int C, float one_b2, float b2, float* __restrict__ iou_fc,
                                                           float* __restrict__ ious, float* __restrict__ fscores, float* __restrict__ cls_count,
                                                           float* __restrict__ vid_miou) {
    const int tid = threadIdx.x;
    for (int c = tid; c < C; c += blockDim.x) {
        float si = 0.f, sf = 0.f, sc = 0.f;
        for (int f = 0; f < BF; ++f) {
            const float ai = (float)areas[((long)f * 3 + 0) * C + c], ap = (float)areas[((long)f * 3 + 1) * C + c], al = (float)areas[((long)f * 3 + 2) * C + c];
            const float au = sub_rn(add_rn(ap, al), ai);
            const float iou = div_rn(ai, add_rn(2.220446049250313e-16f, au));
            iou_fc[(long)f * C + c] = iou;
            si = add_rn(si, iou);
            if (au != 0.f) sc = add_rn(sc, 1.f);
            const float prec = div_rn(ai, ap), rec = div_rn(ai, al);                    // 0 / 0 = NaN, as in the reference
            float fs = div_rn(mul_rn(mul_rn(one_b2, prec), rec), add_rn(mul_rn(b2, prec), rec));
            if (fs != fs) fs = 0.f;
            sf = add_rn(sf, fs);
        }
        ious[c] = si; fscores[c] = sf; cls_count[c] = sc;
    }
    __threadfence();                                            // iou_fc rows are read below by other threads of this block
    __syncthreads();
    for (int f = tid; f < BF; f += blockDim.x) {
        float s = 0.f;
        int nz = 0;
        for (int c = 0; c < C; ++c) {
            const float v = iou_fc[(long)f * C + c];
            s = add_rn(s, v);
            nz += v != 0.f ? 1 : 0;
        }
        vid_miou[f] = div_rn(s, (float)nz);                  // no class with a non-zero IoU: 0 / 0 = NaN, as torch.sum(iou) / 0
    }
}

}  // namespace

#define S_(x) ((hipStream_t)(x))

extern "C" {

int crab_mask_iou(crab_ctx* ctx, void* stream, const float* pred, const float* target, int N, int64_t hw, float eps, int32_t* counts, float* out) {
    if (!ctx) return CRAB_E_INVALID;
    if (!pred || !counts || !out || N <= 0 || hw <= 0 || (int64_t)N * hw >= ((int64_t)1 << 31))
        return crab_fail(ctx, CRAB_E_INVALID, "mask_iou: pred [N, hw] fp32, counts [N, 6] int32, out [2] fp32, N hw < 2^31");
    CRAB_HIP_TRY(ctx, hipMemsetAsync(counts, 0, (size_t)N * 6 * sizeof(int32_t), S_(stream)));
    long bx = (hw + 256 * 8 - 1) / (256 * 8);
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(mask_counts_kernel, dim3((unsigned)bx, (unsigned)N), dim3(256), 0, S_(stream), pred, target, (long)hw, counts);
    int rc = crab_check_launch(ctx, "mask_counts_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(mask_iou_finish_kernel, dim3(1), dim3(64), 0, S_(stream), (const int*)counts, N, (long)hw, eps, target ? 1 : 0, out);
    return crab_check_launch(ctx, "mask_iou_finish_kernel");
}

int crab_fmeasure(crab_ctx* ctx, void* stream, const float* pred, const float* gt, int N, int64_t hw, const float* thresholds, int T, double beta2,
                  int32_t* ge, int32_t* ysum, float* fscore, float* score, float* best) {
    if (!ctx) return CRAB_E_INVALID;
    if (!pred || !gt || !thresholds || !ge || !ysum || !fscore || !score || !best || N <= 0 || hw <= 0 || hw >= ((int64_t)1 << 31) || T <= 0 || T > SM_MAXT)
        return crab_fail(ctx, CRAB_E_INVALID, "fmeasure: pred / gt [N, hw] fp32, 1 <= T <= 1024 ascending thresholds, hw < 2^31");
    const float one_b2 = (float)(1.0 + beta2), b2 = (float)beta2;
    hipLaunchKernelGGL(fmeasure_image_kernel, dim3((unsigned)N), dim3(1024), 0, S_(stream), pred, gt, (long)hw, thresholds, T, one_b2, b2, (int*)ge, (int*)ysum, fscore);
    int rc = crab_check_launch(ctx, "fmeasure_image_kernel");
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
    long bx = (hw + 255) / 256;                                // one pixel per thread: a pixel already costs C strided loads
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(class_areas_kernel, dim3((unsigned)bx, (unsigned)BF), dim3(256), 0, S_(stream), pred, (const long long*)target, C, (long)hw, (int*)areas);
    int rc = crab_check_launch(ctx, "class_areas_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(miou_finish_kernel, dim3(1), dim3(1024), 0, S_(stream), (const int*)areas, BF, C, (float)(1.0 + beta2), (float)beta2, iou_fc, ious, fscores,
                       cls_count, vid_miou);
    return crab_check_launch(ctx, "miou_finish_kernel");
}

}  // extern "C"

/*
 * metrics_demo.c -- the segmentation metrics of the pixel-task eval loops through the C ABI of include/crab_hip.h only (no Python, no torch
 * in the process): what a caller of the reference's utils/avss_utils.py (mask_iou :22-47, Eval_Fmeasure :67-96, metric_s_for_null :8-19,
 * calc_color_miou_fscore :422-435; call sites scripts/quick_start.py:118-119, 342, 395) binds instead of moving the masks to the host.
 *
 *   metrics_demo <in.bin> <out.bin> N hw T BF C chw
 * in.bin : pred fp32 [N][hw], gt fp32 [N][hw], thresholds fp32 [T], class logits fp32 [BF][C][chw], class ids int64 [BF][chw]
 * out.bin: counts int32 [N][6], {iou, s} fp32 [2], ge int32 [N][2][T], ysum int32 [N][2], fscore fp32 [N][T], score fp32 [T],
 *          best fp32 [2], areas int32 [BF][3][C], iou_fc fp32 [BF][C], ious / fscores / cls_count fp32 [C] each, vid_miou fp32 [BF]
 * tests/test_c_abi_gpu.py compares every array with crab_amd.avss_utils bit for bit.
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "crab_hip.h"

#define HIP_OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(2); } } while (0)
#define CRAB_OK_(e) do { int r_ = (e); if (r_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #e, r_, crab_last_error(ctx)); exit(3); } } while (0)

static crab_ctx* ctx;
static FILE *fin, *fout;

static void* upload(size_t bytes) {
    void* host = malloc(bytes);
    void* dev = NULL;
    if (fread(host, 1, bytes, fin) != bytes) { fprintf(stderr, "input too short\n"); exit(4); }
    HIP_OK(hipMalloc(&dev, bytes));
    HIP_OK(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
    free(host);
    return dev;
}

static void* dev_alloc(size_t bytes) {
    void* p = NULL;
    HIP_OK(hipMalloc(&p, bytes));
    return p;
}

static void download(const void* dev, size_t bytes) {
    void* host = malloc(bytes);
    HIP_OK(hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
    if (fwrite(host, 1, bytes, fout) != bytes) { fprintf(stderr, "short write\n"); exit(5); }
    free(host);
}

int main(int argc, char** argv) {
    if (argc != 9) { fprintf(stderr, "usage: %s in.bin out.bin N hw T BF C chw\n", argv[0]); return 1; }
    const int N = atoi(argv[3]), T = atoi(argv[5]), BF = atoi(argv[6]), C = atoi(argv[7]);
    const int64_t hw = atoll(argv[4]), chw = atoll(argv[8]);
    fin = fopen(argv[1], "rb");
    fout = fopen(argv[2], "wb");
    if (!fin || !fout) { fprintf(stderr, "cannot open the files\n"); return 1; }
    if (crab_abi_version() < 10) { fprintf(stderr, "libcrab_hip.so is older than ABI 10\n"); return 1; }
    CRAB_OK_(crab_ctx_create(0, &ctx));
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    float* pred = (float*)upload((size_t)N * hw * 4);
    float* gt = (float*)upload((size_t)N * hw * 4);
    float* th = (float*)upload((size_t)T * 4);
    float* cpred = (float*)upload((size_t)BF * C * chw * 4);
    int64_t* ctgt = (int64_t*)upload((size_t)BF * chw * 8);

    /* mask_iou + metric_s_for_null: one counting pass, one finishing launch */
    int32_t* counts = (int32_t*)dev_alloc((size_t)N * 6 * 4);
    float* out2 = (float*)dev_alloc(2 * 4);
    CRAB_OK_(crab_mask_iou(ctx, stream, pred, gt, N, hw, 1e-7f, counts, out2));

    /* Eval_Fmeasure: beta^2 = 0.3, T thresholds */
    int32_t* ge = (int32_t*)dev_alloc((size_t)N * 2 * T * 4);
    int32_t* ysum = (int32_t*)dev_alloc((size_t)N * 2 * 4);
    float* fscore = (float*)dev_alloc((size_t)N * T * 4);
    float* score = (float*)dev_alloc((size_t)T * 4);
    float* best = (float*)dev_alloc(2 * 4);
    CRAB_OK_(crab_fmeasure(ctx, stream, pred, gt, N, hw, th, T, 0.3, ge, ysum, fscore, score, best));

    /* calc_color_miou_fscore */
    int32_t* areas = (int32_t*)dev_alloc((size_t)BF * 3 * C * 4);
    float* iou_fc = (float*)dev_alloc((size_t)BF * C * 4);
    float* ious = (float*)dev_alloc((size_t)C * 4);
    float* fsc = (float*)dev_alloc((size_t)C * 4);
    float* cls = (float*)dev_alloc((size_t)C * 4);
    float* vid = (float*)dev_alloc((size_t)BF * 4);
    CRAB_OK_(crab_miou_fscore(ctx, stream, cpred, ctgt, BF, C, chw, 0.3, areas, iou_fc, ious, fsc, cls, vid));
    CRAB_OK_(crab_sync(ctx, stream));

    download(counts, (size_t)N * 6 * 4);
    download(out2, 8);
    download(ge, (size_t)N * 2 * T * 4);
    download(ysum, (size_t)N * 2 * 4);
    download(fscore, (size_t)N * T * 4);
    download(score, (size_t)T * 4);
    download(best, 8);
    download(areas, (size_t)BF * 3 * C * 4);
    download(iou_fc, (size_t)BF * C * 4);
    download(ious, (size_t)C * 4);
    download(fsc, (size_t)C * 4);
    download(cls, (size_t)C * 4);
    download(vid, (size_t)BF * 4);
    fclose(fout);

    /* a bad argument comes back as an error code with a message, not as a fault */
    if (crab_fmeasure(ctx, stream, pred, gt, N, hw, th, 4096, 0.3, ge, ysum, fscore, score, best) != CRAB_E_INVALID) { fprintf(stderr, "T = 4096 accepted\n"); return 6; }
    printf("iou/s/fmeasure written; last error: %s\n", crab_last_error(ctx));
    crab_ctx_destroy(ctx);
    return 0;
}

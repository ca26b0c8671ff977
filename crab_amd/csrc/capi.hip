// Context management for libcrab_hip.so (see include/crab_hip.h).
#include "crab_internal.h"
#include <stdlib.h>

extern "C" {

int crab_abi_version(void) { return 12; }   // 12: crab_vq_nearest_f32 / crab_row_sqnorm_f32 (codebook ids from fp32 latents, fp32 codebook, fp32 distances), crab_groupnorm_p (fp32 GroupNorm parameters); 11: crab_trace_begin / crab_trace_end (which kernels the entry points launched: tests pin a comparison to a kernel instantiation); 10: segmentation metrics of the pixel-task eval loops (crab_mask_iou, crab_fmeasure, crab_miou_fscore, crab_color_to_label; seg_metrics.hip); 9: the ragged decode batch - crab_gemm_desc.rope_row_off, crab_llama_io.row_off, crab_qkv_rope_split_ragged (several generate() calls of different prompt lengths decode as one right-aligned batch); 8: crab_attn_desc.key_mask / key_mask_ld (general 2-D attention_mask), crab_attn_decode_keymask; 7: fp32 residual stream (crab_gemm_desc.r_fp32, crab_llama_io / crab_enc_io.x_fp32, crab_rmsnorm_f32 / crab_layernorm_f32 / crab_embedding_f32 / crab_cast_rows_*); 6: crab_gemm_desc.rope_S / rope_pos_ids (prefill RoPE in the q|k|v epilogue), crab_gemm_fuses_prefill_rope; 2: fused RoPE / KV-append fields in crab_gemm_desc; 3: next-group router fields; 4: crab_llama_layer*; 5: crab_attn_desc.kv_start, crab_qkv_rope_split_ids, crab_attn_decode_masked

int crab_decode_max_rows(void) { return CRAB_DECODE_MAX_ROWS; }
int crab_attn_split_below(void) { return CRAB_ATTN_SPLIT_BELOW; }
int crab_sizeof_gemm_desc(void) { return (int)sizeof(crab_gemm_desc); }
int crab_sizeof_attn_desc(void) { return (int)sizeof(crab_attn_desc); }

int crab_ctx_create(int device, crab_ctx** out) {
    if (!out) return CRAB_E_INVALID;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return CRAB_E_HIP;
    crab_ctx* c = (crab_ctx*)calloc(1, sizeof(crab_ctx));
    if (!c) return CRAB_E_INVALID;
    c->device = device;
    c->err[0] = 0;
    *out = c;
    return CRAB_OK;
}

void crab_ctx_destroy(crab_ctx* ctx) { free(ctx); }

const char* crab_last_error(crab_ctx* ctx) { return ctx ? ctx->err : "null context"; }

int crab_trace_begin(crab_ctx* ctx) {
    if (!ctx) return CRAB_E_INVALID;
    ctx->trace_n = 0;
    ctx->trace_on = 1;
    return CRAB_OK;
}

int64_t crab_trace_end(crab_ctx* ctx, char* buf, int64_t n) {
    if (!ctx) return -1;
    ctx->trace_on = 0;
    int64_t w = 0;
    for (int i = 0; i < ctx->trace_n; ++i) {
        char line[96];
        const int l = snprintf(line, sizeof(line), "%s\t%ld\n", ctx->trace[i].name, ctx->trace[i].count);
        if (buf && w + l < n) memcpy(buf + w, line, (size_t)l);
        w += l;
    }
    if (buf && n > 0) buf[w < n ? w : n - 1] = 0;
    return w;                                   // bytes needed (without the terminator): a caller with a short buffer asks again
}

int crab_sync(crab_ctx* ctx, void* stream) {
    if (!ctx) return CRAB_E_INVALID;
    CRAB_HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
    return CRAB_OK;
}

}  // extern "C"

/*
 * crab_hip.h -- C ABI of libcrab_hip.so, the MI355X (gfx950) forward path for Crab's multimodal
 * inference stack.  Plain pointers and sizes only; no torch types.  All pointers are DEVICE pointers
 * unless a parameter says "host".  Every entry point returns 0 on success or a negative CRAB_E_* code,
 * never throws and never exits; crab_last_error() gives the message.  The library never allocates,
 * frees or retains caller memory: inputs, outputs, KV caches and workspaces are caller-owned
 * (SURVEY.md 8b "Ownership").  `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream).
 *
 * The reference (GeWu-Lab/Crab) is 100% Python and has no FFI; each entry point cites the reference
 * module whose arithmetic it replaces, i.e. what a maintainer would bind instead of the eager
 * PyTorch op sequence (see INTEGRATION.md for the ctypes stubs).
 *
 * Storage dtype is bf16 (uint16 raw bits == torch.bfloat16); accumulation and softmax/norm math are fp32.  One exception since ABI 7:
 * the RESIDUAL STREAM (decoder x, CLIP tower x, the pre-LayerNorm sums of the post-LN encoders) may be - and in crab_amd is - fp32
 * (crab_gemm_desc.c_fp32 / r_fp32, crab_llama_io.x_fp32, crab_enc_io.x_fp32).
 */
#ifndef CRAB_HIP_H
#define CRAB_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRAB_OK 0
#define CRAB_E_INVALID (-1)   /* bad argument (shape, alignment, null pointer) */
#define CRAB_E_HIP (-2)       /* a HIP runtime call or kernel launch failed */
#define CRAB_E_UNSUPPORTED (-3)
#define CRAB_E_WORKSPACE (-4) /* caller workspace too small */

typedef struct crab_ctx crab_ctx;

int crab_ctx_create(int device, crab_ctx** out);
void crab_ctx_destroy(crab_ctx* ctx);
const char* crab_last_error(crab_ctx* ctx);
int crab_sync(crab_ctx* ctx, void* stream);
int crab_abi_version(void);
/* The launch trace (ABI 11; diagnostics, host-side bookkeeping only): between crab_trace_begin and crab_trace_end every kernel launch an entry
 * point of this context issues is counted under the name of its launch site, template instantiation included where the dispatch has one
 * ("attn_decode_kernel<128>", "attn_fwd32_kernel<128,causal>", "gemm_bt_ring_kernel+rope2", ...).  crab_trace_end stops the trace and
 * writes "name\tcount\n" lines (NUL-terminated) into the HOST buffer `buf` of `n` bytes; it returns the bytes the full listing needs, so a
 * caller whose buffer was short asks again with a larger one (the counts stay until the next crab_trace_begin).  No reference counterpart:
 * the reference is eager PyTorch.  Used by the parity tests to prove WHICH kernel a reference-generated fixture was compared through. */
int crab_trace_begin(crab_ctx* ctx);
int64_t crab_trace_end(crab_ctx* ctx, char* buf /* host */, int64_t n);
/* sizeof(crab_gemm_desc) / sizeof(crab_attn_desc) as compiled into the library: lets a binding verify its struct mirror */
int crab_sizeof_gemm_desc(void);
int crab_sizeof_attn_desc(void);

/* activation codes for epilogues */
enum { CRAB_ACT_NONE = 0, CRAB_ACT_GELU = 1, CRAB_ACT_QUICK_GELU = 2, CRAB_ACT_RELU = 3, CRAB_ACT_SILU = 4,
       /* crab_gemm_bf16 only: B rows interleaved (gate_i, up_i); C[m, j] = silu(v[m, 2j]) * v[m, 2j+1], C has N/2 columns
        * (LlamaMLP, modeling_llama.py:269, with gate_proj / up_proj packed into one GEMM) */
       CRAB_ACT_SWIGLU_PAIR = 5 };

/* ---------------------------------------------------------------------------------------------
 * GEMM:  C[M,N] = res_scale * R[M,N] + act( A[M,K] . B[N,K]^T + A2[M,K2] . B2[N,K2]^T + bias[N] )
 * A, B (and A2, B2) are K-contiguous bf16 ("x @ W^T" with W stored [out,in] exactly as
 * torch.nn.Linear.weight).  fp32 MFMA accumulation.  C is bf16, or fp32 when c_fp32 != 0; R is bf16, or fp32 when r_fp32 != 0
 * (ldr then counts fp32 elements).  c_fp32 = r_fp32 = 1 with R == C is the FP32 RESIDUAL STREAM of the decoder / CLIP tower
 * (x += proj(...) without a bf16 rounding of x per layer; the reference adds its residuals in fp32, models/modeling_llama.py:805-827
 * run with --bf16 False, scripts/quick_start.sh:42-44).
 * The optional second K segment carries the hyper-LoRA update (A2 = routed rank-24 activations,
 * B2 = concatenated lora_B), see crab_hyperlora_mix.
 * Replaces: every nn.Linear / F.linear on the path -- peft_hyper/tuners/lora.py:341 (base linear),
 * transformers CLIP/Llama/Qwen2 projections, models/Qformer.py dense layers, models/beats/backbone.py
 * q/k/v/out/fc1/fc2, and the patch-embedding convolutions once im2col'ed.
 * Batched form: z in [0,batch): z0 = z % nb0, z1 = z / nb0; X += z0*sX0 + z1*sX1 (elements).
 * Requirements: K, K2, lda, ldb, lda2, ldb2 multiples of 8; A/B 16-byte aligned.
 */
typedef struct {
    const void* A; const void* B; void* C;
    const void* bias;  /* bf16 [N] or NULL */
    const void* R;     /* bf16 [M,N] residual or NULL */
    const void* A2; const void* B2; /* optional second K segment, or NULL */
    int64_t lda, ldb, ldc, ldr, lda2, ldb2;
    int32_t M, N, K, K2;
    int32_t act;       /* CRAB_ACT_* applied to (acc + bias) before the residual */
    int32_t c_fp32;
    float res_scale;   /* multiplies R (BEATs deep-norm alpha); 1.0 for a plain residual */
    int32_t batch, nb0;   /* batch <= 1 means unbatched */
    int64_t sA0, sA1, sB0, sB1, sC0, sC1, sR0, sR1, sBias0, sBias1;
    int32_t tune;      /* 0 = automatic kernel choice; >0 forces a variant (benchmarking only) */
    void* workspace;   /* optional caller-owned scratch for split-K partials (decode regime, 16 < M <= 128); NULL = none */
    int64_t workspace_bytes;
    /* optional fused post-RMSNorm: norm_out[M,N] (bf16) = rmsnorm(C) * norm_w (LlamaRMSNorm behind o_proj / down_proj).  C bf16: the norm sees the
     * stored bf16 row and rounds x_hat before the weight multiply (modeling_llama.py:116-117 in bf16).  C fp32 (then R, if given, must be fp32 too:
     * r_fp32): the norm sees the unrounded row, one rounding of x_hat * w. */
    const void* norm_w; void* norm_out; int64_t ld_norm; float norm_eps;
    /* optional fused RoPE + KV-cache append behind the packed q|k|v projection, ONE ROW PER SEQUENCE (decode): the result
     * is what crab_gemm_bf16 followed by crab_qkv_rope_split(B = M, S = 1) on C would leave - C[:, :H*d] = RoPE(q), the
     * k / v rows stored at k_cache / v_cache[b, hk, pos, :], pos = rope_pos0 + rope_pos_dev[0] (modeling_llama.py:204-236,
     * 408-412) - in the split-K regime without a separate pass over C.  rope_tab == NULL disables it.  Needs
     * N == (rope_H + 2*rope_Hk) * rope_d, bf16 C, no activation / residual / post-norm. */
    const float* rope_tab; void* rope_k_cache; void* rope_v_cache; const int32_t* rope_pos_dev;
    int32_t rope_H, rope_Hk, rope_d, rope_Tmax, rope_pos0;
    /* optional hyper-LoRA router of the NEXT projection group, evaluated on the post-norm rows (needs norm_w / norm_out):
     * route_U[M, route_ucols] = crab_hyperlora_route(norm_out, route_RA, ...) - inside the row-owning split-K reduction when
     * that path is taken (the normalised row is already in registers), by a separate pass otherwise (which then needs
     * `workspace`).  route_RA == NULL disables it; route_RA is [pad16(nproj*(nl+r)), N] with row stride route_ldra. */
    const void* route_RA; void* route_U; int64_t route_ldra, route_ldu;
    int32_t route_nproj, route_nl, route_r, route_ucols; float route_scaling;
    /* optional hyper-LoRA of THIS projection evaluated inside the call (single-projection groups: o_proj, down_proj; peft_hyper/tuners/
     * lora.py:338-350) instead of being handed in as the second K segment:  C = ... + lora_scaling * sum_i softmax(A . R^T)_i * B_i (A . A_lora^T).
     * lora_RA [pad16(lora_nl + lora_r), K] = lora_nl route rows then lora_r lora_A rows (row stride lora_ldra); B2 / ldb2 / K2 describe
     * lora_B [N, K2 >= lora_nl * lora_r] (expert i, rank j at column i * lora_r + j) and A2 must be NULL.  The router product rides on the
     * projection's launch as 16 extra weight rows and the update is applied by the row-owning tail that also stores the residual row, its
     * RMSNorm and the next group's router (csrc/rowfin.hip): needs M <= 16, norm_w / norm_out, crab_rowfin_lora_ok(lora_nl, lora_r, N), K2 % 8 == 0, lora_RA
     * padded to 16 rows (the ride-along blocks read 16), and a workspace of
     * crab_rowfin_workspace(M, N) bytes; CRAB_E_UNSUPPORTED otherwise (use crab_hyperlora_route + A2 there).  lora_RA == NULL disables it. */
    const void* lora_RA; int64_t lora_ldra; int32_t lora_nl, lora_r; float lora_scaling;
    /* PREFILL form of the fused RoPE (rope_S > 1 rows per sequence, M = B * rope_S; head_dim 128, rope_pos_dev == NULL): the q and k column
     * tiles of the large-M kernel rotate in their epilogue - q in place in C, k straight into rope_k_cache[b, hk, rope_pos0 + s, :] - with
     * the rotary position rope_pos_ids[b * rope_ld_pos + s] when given (forward()'s position_ids) and rope_pos0 + s otherwise.  The v
     * columns are written to C as usual: the caller finishes with crab_qkv_rope_split(rope_tab = NULL, k_cache = NULL) (v-cache append + V^T).
     * Whether a given call will do this is a pure function of the descriptor: crab_gemm_fuses_prefill_rope(d); when it returns 0 the rope_*
     * fields are ignored at M > 256 and the caller runs the full crab_qkv_rope_split as before.  Bit-identical to that pair. */
    int32_t rope_S; int64_t rope_ld_pos; const int32_t* rope_pos_ids;
    /* optional with rope_S > 1: V^T scratch [B, Hk, 128, rope_vt_ld] (what crab_qkv_rope_split's `vt` is).  When given - and the call fuses at all -
     * the v column tiles append to rope_v_cache and write V^T in the epilogue as well: crab_gemm_fuses_prefill_rope(d) == 2, no split pass left. */
    void* rope_vt; int64_t rope_vt_ld;
    /* R is fp32 [M, ldr] (see the top of this comment).  With norm_w the fused post-norm then reads the fp32 row (c_fp32 must be set too, and
     * the normalised row is bf16(x_hat * w) without the intermediate bf16 rounding of x_hat that the all-bf16 form reproduces).  Unbatched only. */
    int32_t r_fp32;
    /* optional with the DECODE form of the fused RoPE (rope_S <= 1), ABI 9: int32 [M], row m is rotated at position
     * (rope_pos0 + rope_pos_dev[0]) - rope_row_off[m] while its K / V rows still land in cache slot rope_pos0 + rope_pos_dev[0].  This is the
     * RAGGED decode batch: sequences whose prompts have different lengths are right-aligned in one KV cache (sequence m's first key sits in
     * slot rope_row_off[m]) so that one append index serves every row, yet every row keeps the rotary positions 0 .. S_m - 1 of its own
     * generate() call (models/unified_llama.py:262-267 gives every call of the eval loop positions from 0; scripts/finetune/
     * inference_hyper_lora.py:1466-1479).  NULL = no offsets. */
    const int32_t* rope_row_off;
    /* norm_w is fp32 [N] (16-byte aligned) instead of bf16 (ABI 9): the RMSNorm weight is not a matrix operand; kept in fp32 it removes a
     * 2^-9 relative error from every channel of every normalised row.  With the fp32 residual stream only (c_fp32 = r_fp32 = 1). */
    int32_t norm_w_fp32;
} crab_gemm_desc;

/* Rows up to which crab_gemm_bf16 treats a problem as WEIGHT-STREAMING (the decode regime: one row per clip) when a workspace is given:
 * the weights are read once per call whatever M; crab_llama_layers / crab_amd/decoder.py fuse the routers and norms into the reductions up
 * to the same bound.  256 until r03; 512 since r04 (two 256-row groups per launch sharing each weight panel through an XCD's L2). */
#define CRAB_DECODE_MAX_ROWS 512
int crab_decode_max_rows(void);   /* the value compiled into the library (bindings check their mirror of the macro against it) */
int crab_attn_split_below(void);  /* likewise for CRAB_ATTN_SPLIT_BELOW */
int crab_gemm_bf16(crab_ctx* ctx, void* stream, const crab_gemm_desc* d);
int crab_gemm_fuses_prefill_rope(const crab_gemm_desc* d);   /* 0: no; 1: this call rotates q / k and appends k in its epilogue (see rope_S); 2: and handles the v columns (rope_vt) */
/* bytes of crab_gemm_desc.workspace the M <= 16 layer tail needs (fp32 sums + router product + per-slice partials) */
int64_t crab_rowfin_workspace(int M, int N);
/* 1 when the in-call hyper-LoRA form (crab_gemm_desc.lora_RA, M <= 16) can serve an adapter with nl experts of rank r on a projection of N
 * outputs (nl <= 8, nl + r <= 16, nl * r <= 32, N <= 8192, N % 8 == 0); callers pick crab_hyperlora_route + the A2 / B2 segment otherwise
 * (e.g. lora_r = 16: peft_hyper/tuners/lora.py:42-83 makes r a free argument, 8 is only the reference's default) */
int crab_rowfin_lora_ok(int nl, int r, int N);

/* ---------------------------------------------------------------------------------------------
 * hyper-LoRA routing mix (peft_hyper/tuners/lora.py:346-350).
 * T[M, ldt] holds, for each of `nproj` projections sharing the same input x, the skinny product
 * x.[R;A]^T : columns p*(nl+r) .. +nl-1 = route logits, then r columns of lora_A(x).
 * Writes U[M, ldu] bf16 with U[m, p*nl*r + i*r + j] = scaling * softmax_fp32(route)_i * h_j ;
 * columns [nproj*nl*r, ucols) are zero-filled (K padding of the second GEMM segment).
 */
int crab_hyperlora_mix(crab_ctx* ctx, void* stream, const void* T, int64_t ldt, int t_fp32, void* U, int64_t ldu,
                       int M, int nproj, int nl, int r, int ucols, float scaling);

/* Fused router: U = mix(x . [R;A]^T) without materialising T through a GEMM grid.  RA is [tcols, K] bf16 with
 * tcols = round_up(nproj*(nl+r), 16) rows (zero padded).  K is split across blocks, partial products are summed in a
 * fixed order (deterministic).  workspace: crab_hyperlora_route_workspace(M, K, tcols) bytes, caller-owned. */
int64_t crab_hyperlora_route_workspace(int M, int K, int tcols);
int crab_hyperlora_route(crab_ctx* ctx, void* stream, const void* X, int64_t ldx, const void* RA, int64_t ldra, int M, int K,
                         int nproj, int nl, int r, void* U, int64_t ldu, int ucols, float scaling, void* workspace,
                         int64_t workspace_bytes);

/* RMSNorm (models/modeling_llama.py:112-117) and LayerNorm (torch.nn.LayerNorm) over the last dim, bf16 in/out. */
int crab_rmsnorm(crab_ctx* ctx, void* stream, const void* x, int64_t ldx, const void* w, void* y, int64_t ldy,
                 int M, int D, float eps);
int crab_layernorm(crab_ctx* ctx, void* stream, const void* x, int64_t ldx, const void* w, const void* b, void* y,
                   int64_t ldy, int M, int D, float eps);

/* The same over an fp32 input row (the fp32 residual stream): y = bf16(x_hat * w [+ b]), no intermediate rounding. */
int crab_rmsnorm_f32(crab_ctx* ctx, void* stream, const float* x, int64_t ldx, const void* w, void* y, int64_t ldy,
                     int M, int D, float eps);
int crab_layernorm_f32(crab_ctx* ctx, void* stream, const float* x, int64_t ldx, const void* w, const void* b, void* y,
                       int64_t ldy, int M, int D, float eps);
/* LayerNorm with the storage of both the input row (x_fp32) and the PARAMETERS (w_fp32: w and b are fp32 [D], 16-byte aligned) stated by the
 * caller (ABI 9).  The LayerNorm weights / biases of the encoders are not matrix operands: kept in fp32 they cost 4 KB per norm and remove a
 * systematic 2^-9 relative error from every channel of every LayerNorm output (the encoder features move from 2x to 1.2x the
 * bf16-operand floor).  y is bf16. */
int crab_layernorm_p(crab_ctx* ctx, void* stream, const void* x, int x_fp32, int64_t ldx, const void* w, const void* b, int w_fp32, void* y,
                     int64_t ldy, int M, int D, float eps);
/* RMSNorm likewise (the decoder's input_layernorm / post_attention_layernorm / model.norm weights in fp32: crab_llama_layer.norm_w_fp32) */
int crab_rmsnorm_p(crab_ctx* ctx, void* stream, const void* x, int x_fp32, int64_t ldx, const void* w, int w_fp32, void* y, int64_t ldy,
                   int M, int D, float eps);

/* out[t,:] = table[ids[t],:]  (embed_tokens; unified_arch.py:213-214, unified_llama.py:125-127).  Rows with ids[t] < 0 are
 * left untouched (the multimodal splice fills them with projector features, unified_arch.py:283-300); ids >= vocab clamp. */
int crab_embedding(crab_ctx* ctx, void* stream, const int64_t* ids, const void* table, void* out, int64_t ldo,
                   int T, int D, int vocab);
/* fp32 output rows: where the fp32 residual stream of a decode step starts */
int crab_embedding_f32(crab_ctx* ctx, void* stream, const int64_t* ids, const void* table, float* out, int64_t ldo,
                       int T, int D, int vocab);

/* rope table: tab[pos][i] = (cos, sin)(pos * theta^(-2i/d)), fp32 pairs, i < d/2  (modeling_llama.py:130-156) */
int crab_rope_table(crab_ctx* ctx, void* stream, float* tab, int max_pos, int head_dim, float theta);

/* Split a packed projection qkv[T, (H+2Hk)*d] (T = B*S tokens, row stride ldqkv), apply RoPE to q and k
 * (modeling_llama.py:204-236), and scatter:
 *   q  -> rotated in place inside qkv
 *   k  -> k_cache[b, hk, pos, :]      (cache layout [B, Hk, Tmax, d], appended AFTER RoPE, :408-412)
 *   v  -> v_cache[b, hk, pos, :]
 *   v^T-> vt[b, hk, :, s]  ([B, Hk, d, vt_ld]) when vt != NULL (prefill attention operand)
 * pos = pos0 + s, or pos_dev[0] + s when pos_dev != NULL (device-resident decode position).
 * rope_tab == NULL skips the rotation (encoders). */
int crab_qkv_rope_split(crab_ctx* ctx, void* stream, void* qkv, int64_t ldqkv, const float* rope_tab,
                        void* k_cache, void* v_cache, void* vt, int64_t vt_ld, int B, int S, int H, int Hk, int d,
                        int Tmax, int pos0, const int32_t* pos_dev);

/* Same with explicit rotary positions: token (b, s) is rotated by pos_ids[b*ld_pos + s] (the position_ids forward() passes on,
 * models/unified_llama.py:149-160: cumsum(attention_mask) - 1 for a left-padded batch, unified_arch.py:372-373) while its K / V rows
 * still land in cache slot pos0 + s (+ pos_dev[0]).  pos_ids == NULL is crab_qkv_rope_split. */
int crab_qkv_rope_split_ids(crab_ctx* ctx, void* stream, void* qkv, int64_t ldqkv, const float* rope_tab,
                            void* k_cache, void* v_cache, void* vt, int64_t vt_ld, int B, int S, int H, int Hk, int d,
                            int Tmax, int pos0, const int32_t* pos_dev, const int32_t* pos_ids, int64_t ld_pos);

/* The ragged form (ABI 9): sequence b is rotated at slot - row_off[b] (slot = pos0 + pos_dev[0] + s) - see crab_gemm_desc.rope_row_off.
 * row_off == NULL is crab_qkv_rope_split. */
int crab_qkv_rope_split_ragged(crab_ctx* ctx, void* stream, void* qkv, int64_t ldqkv, const float* rope_tab,
                               void* k_cache, void* v_cache, void* vt, int64_t vt_ld, int B, int S, int H, int Hk, int d,
                               int Tmax, int pos0, const int32_t* pos_dev, const int32_t* row_off);

/* Encoder-side split: qkv[T, 3*H*d] -> kbuf[B,H,S,d], vt[B,H,d,vt_ld]; q stays in place (no RoPE). */

/* Flash attention forward (MFMA, LDS-staged K / V^T tiles, fp32 online softmax).
 *   O[b, i, h*d + :] = softmax_j( scale * q_i.k_j + gate[b,h,i] * bias[h,i,j] + mask ) . v_j
 * q: element (b,h,i,:) at q + b*q_bs + h*q_hs + i*q_ss ; k likewise ; vt: (b,h,dd,j) at vt + b*vt_bs + h*vt_hs + dd*vt_ds + j
 * H query heads, Hk key/value heads (GQA: kv head = h / (H/Hk)).  causal != 0 masks j > i + (Skv - Sq).
 * bias (fp32 [H,Sq,Skv]) and gate (fp32 [B,H,Sq]) may be NULL.  head_dim in {32,64,128} (32: no causal/bias form).
 * Replaces: modeling_llama.py:417-445 (prefill), CLIP / Q-Former (Qformer.py:171-277) / BEATs
 * (backbone.py:621-670, gated relative position bias) eager attention.
 */
typedef struct {
    const void* q; const void* k; const void* vt; void* o;
    int64_t q_bs, q_hs, q_ss, k_bs, k_hs, k_ss, vt_bs, vt_hs, vt_ds, o_bs, o_ss;
    const float* bias; const float* gate;
    int32_t B, H, Hk, Sq, Skv, head_dim, causal;
    float scale;
    /* optional [B] int32: keys j < kv_start[b] are masked for sequence b - the left-pad attention_mask that forward() hands to the
     * decoder (models/unified_llama.py:149-160, prepare_multimodal_inputs' mask :344-373).  A query row left without any visible
     * key (a pad row) produces zeros.  NULL = no mask. */
    const int32_t* kv_start;
    /* optional general 2-D attention_mask over the keys (HF accepts any mask, not only left padding: modeling_attn_mask_utils'
     * padding mask AND-ed with the causal one): [B][key_mask_ld] 32-bit words, bit (j & 31) of word j >> 5 set = key j of sequence b is
     * visible; key_mask_ld >= ceil(Skv / 32).  Combines with causal / kv_start; a query row without a visible key produces zeros
     * (the reference's softmax over an all-masked row is implementation-defined).  NULL = no mask.  head_dim 64 / 128. */
    const uint32_t* key_mask; int64_t key_mask_ld;
} crab_attn_desc;
int crab_attn_fwd(crab_ctx* ctx, void* stream, const crab_attn_desc* d);

/* Decode attention: one query row per (b,h) against the KV cache [B,Hk,Tmax,d] (modeling_llama.py:394-445
 * with q_len == 1).  ctx_len keys are visible: ctx_len = ctx_len_host, or ctx_dev[0] + ctx_len_host when
 * ctx_dev != NULL.  q: [B, ldq] row per sequence with head h at column h*d; o likewise. */
int crab_attn_decode(crab_ctx* ctx, void* stream, const void* q, int64_t ldq, const void* k_cache, const void* v_cache,
                     void* o, int64_t ldo, int B, int H, int Hk, int d, int Tmax, int ctx_len_host,
                     const int32_t* ctx_dev, float scale);

/* Same, with the first kv_start[b] cache rows of sequence b invisible (left-pad attention_mask of a padded batch; NULL = none). */
int crab_attn_decode_masked(crab_ctx* ctx, void* stream, const void* q, int64_t ldq, const void* k_cache, const void* v_cache,
                            void* o, int64_t ldo, int B, int H, int Hk, int d, int Tmax, int ctx_len_host,
                            const int32_t* ctx_dev, float scale, const int32_t* kv_start);
/* Same under a general key mask (crab_attn_desc.key_mask's layout; key_mask_ld >= ceil(ctx_len / 32), Tmax-wide with ctx_dev): the
 * one-token shortcut of forward() (models/unified_llama.py:125-127) with an attention_mask that has holes.  Per-(b, h) kernel for any
 * H / Hk; a row without a visible key produces zeros. */
int crab_attn_decode_keymask(crab_ctx* ctx, void* stream, const void* q, int64_t ldq, const void* k_cache, const void* v_cache,
                             void* o, int64_t ldo, int B, int H, int Hk, int d, int Tmax, int ctx_len_host, const int32_t* ctx_dev,
                             float scale, const uint32_t* key_mask, int64_t key_mask_ld);

/* Decode attention of the small-batch layer (B * H < 256 blocks would not fill the chip): RoPE of q and of the new k, the KV-cache
 * append (modeling_llama.py:204-236, 408-412) and the attention over keys 0 .. pos in ONE launch, from the RAW packed q|k|v row
 * (qkv [B, ldqkv]: H query heads, Hk key heads, Hk value heads of d elements, as the projection left them, bias and adapter included).
 * pos = pos0 + pos_dev[0] (pos_dev may be NULL).  q / k are rounded to bf16 after the rotation like crab_qkv_rope_split stores
 * them (the appended cache rows are bit-identical to that pair's; o agrees up to the softmax accumulation order); qkv itself is not modified.  The context is split over up
 * to 8 blocks per (b, h) while B * H < 256 (about one block per CU), merged by the last split to finish (deterministic).  workspace:
 * crab_attn_decode_rope_workspace(B, H, d) bytes whose LAST B * H * 4 bytes (the tickets) the caller zero-fills once; the call
 * leaves them zero.  May be NULL when B * H >= 256.  crab_llama_layers / crab_amd/decoder.py pick this entry point while
 * B * H < CRAB_ATTN_SPLIT_BELOW (at least two context splits per head: the reference's batch sizes 1 .. 8 with 32 heads). */
#define CRAB_ATTN_SPLIT_BELOW 257    /* the fused small-batch attention is used while B * H is below this (512-block target: >= 2 context splits per head) */
int64_t crab_attn_decode_rope_workspace(int B, int H, int d);
int crab_attn_decode_rope(crab_ctx* ctx, void* stream, const void* qkv, int64_t ldqkv, const float* rope_tab, void* k_cache,
                          void* v_cache, void* o, int64_t ldo, int B, int H, int Hk, int d, int Tmax, int pos0,
                          const int32_t* pos_dev, float scale, void* workspace, int64_t workspace_bytes);

/* y[M, I] = silu(gu[:, :I]) * gu[:, I:2I]   (modeling_llama.py:269; gate and up packed side by side) */
int crab_swiglu(crab_ctx* ctx, void* stream, const void* gu, int64_t ldgu, void* y, int64_t ldy, int M, int I);

/* ids[b] = argmax_v logits[b, v] (fp32, first maximum wins, like torch.argmax). suppress >= 0 masks that id. */
int crab_argmax(crab_ctx* ctx, void* stream, const float* logits, int64_t ldl, int64_t* ids, int B, int V, int suppress);

/* Non-overlapping patch im2col: in[N,C,Hh,Ww] (fp32 or bf16) -> out[N*gh*gw, ldo] bf16, k = (c*P + ky)*P + kx,
 * token = (n, gy, gx) row-major; columns [C*P*P, ldo) zero.  CLIP: conv14/14 (HF CLIPVisionEmbeddings),
 * BEATs: conv16/16 on [B,1,L,128] (BEATs.py:148-151; rows beyond gh*P are dropped like the conv does). */
int crab_im2col_patch(crab_ctx* ctx, void* stream, const void* in, int in_fp32, void* out, int64_t ldo, int N, int C,
                      int Hh, int Ww, int P);

/* CLIP token assembly + pre_layrnorm: x[n,0]=cls+pos[0], x[n,1+p]=patch[n,p]+pos[1+p]; y = LN(x) */
int crab_clip_embed_ln(crab_ctx* ctx, void* stream, const void* patch, const void* cls, const void* pos, const void* lnw,
                       const void* lnb, void* y, int N, int P, int D, float eps);
/* the same with pre_layrnorm's weight / bias in fp32 when ln_fp32 != 0 (ABI 9) */
int crab_clip_embed_ln_p(crab_ctx* ctx, void* stream, const void* patch, const void* cls, const void* pos, const void* lnw,
                         const void* lnb, int ln_fp32, void* y, int N, int P, int D, float eps);

/* BEATs helpers (models/beats/backbone.py):
 *  posconv_pad: x[B,n,E] -> xp[G][B][n+Kc-1][E/G] zero padded (Kc/2 in front), the sliding-window GEMM operand
 *  relpos_bias: bias[h,i,j] = table[bucket(j-i)][h]   (:392-430), fp32 out
 *  gru_gate:    gate[b,h,i] = ga*(gb*grep_a[h]-1)+2 from the un-scaled q projection (:650-662), fp32 out */
int crab_beats_posconv_pad(crab_ctx* ctx, void* stream, const void* x, void* xp, int B, int n, int E, int G, int Kc);
int crab_beats_relpos_bias(crab_ctx* ctx, void* stream, const void* table, float* bias, int n, int H, int num_buckets,
                           int max_distance);
int crab_beats_gru_gate(crab_ctx* ctx, void* stream, const void* q, int64_t ldq, const void* gw, const void* gb,
                        const void* grep_a, float* gate, int B, int n, int H, int d);

/* Strided row copy / dtype cast helpers: dst[r, 0:cols] = src[r, 0:cols] */
int crab_copy_rows(crab_ctx* ctx, void* stream, const void* src, int64_t lds, void* dst, int64_t ldd, int rows, int cols);
int crab_copy_rows_batched(crab_ctx* ctx, void* stream, const void* src, int64_t lds, int64_t src_bs, void* dst, int64_t ldd,
                           int64_t dst_bs, int batch, int rows, int cols);
int crab_cast_f32_bf16(crab_ctx* ctx, void* stream, const float* src, void* dst, int64_t n);
/* strided row casts between bf16 activations and the fp32 residual stream: dst[r, 0:cols] = (T)src[r, 0:cols]; cols % 8 == 0 */
int crab_cast_rows_bf16_f32(crab_ctx* ctx, void* stream, const void* src, int64_t lds, float* dst, int64_t ldd, int rows, int cols);
int crab_cast_rows_f32_bf16(crab_ctx* ctx, void* stream, const float* src, int64_t lds, void* dst, int64_t ldd, int rows, int cols);

/* Device-resident greedy decoding step (HF GenerationMixin greedy search as driven by unified_llama.py:262-267,
 * SURVEY.md B.3): tok = argmax(logits[b]) on fp32 logits (first maximum wins); eos is suppressed while
 * step < min_new_tokens; rows already finished emit pad_id; cur_ids[b] = tok (feeds the next embedding lookup),
 * out_ids[b, step] = tok, finished[b] |= (tok == eos).  step is read from device memory so the launch can be
 * replayed from a HIP graph; crab_advance then bumps the position and step words. eos_id < 0 disables EOS. */
int crab_greedy_select(crab_ctx* ctx, void* stream, const float* logits, int64_t ldl, int B, int V, int64_t* cur_ids,
                       int64_t* out_ids, int64_t ld_out, const int32_t* step_dev, int32_t* finished, int eos_id, int pad_id,
                       int min_new_tokens);
int crab_advance(crab_ctx* ctx, void* stream, int32_t* pos_dev, int32_t* step_dev);
/* The same step with SAMPLING (HF GenerationMixin sample mode: the reference never passes do_sample, so Llama-2-chat's generation_config
 * applies - temperature 0.6, top_p 0.9, GenerationConfig's default top_k 50; scripts/quick_start.py:36-43): logits / temperature -> keep
 * the top_k largest (0 = all) -> keep the smallest set of largest tokens whose probability mass reaches top_p (TopPLogitsWarper: token i
 * stays iff the mass of the strictly larger ones is < top_p) -> draw from the renormalised rest.  The draw uses a counter-based generator
 * keyed by (seed, step_dev[0], row): deterministic per seed, replayable from a HIP graph; HF's torch.multinomial stream is not reproduced. */
int crab_sample_select(crab_ctx* ctx, void* stream, const float* logits, int64_t ldl, int B, int V, int64_t* cur_ids,
                       int64_t* out_ids, int64_t ld_out, const int32_t* step_dev, int32_t* finished, int eos_id, int pad_id,
                       int min_new_tokens, float temperature, int top_k, float top_p, uint64_t seed);

/* ---------------------------------------------------------------------------------------------
 * Fused decoder blocks: the launch sequence of ONE LlamaDecoderLayer / Qwen2DecoderLayer with hyper-LoRA
 * adapted projections (models/modeling_llama.py:805-827 = rmsnorm -> self_attn -> residual -> rmsnorm -> mlp -> residual;
 * models/qwen/modeling_qwen2.py:202-317; peft_hyper/tuners/lora.py:338-350) behind one call, so that a C caller runs a
 * prefill pass or a decode step without re-implementing the sequencing of crab_amd/decoder.py:
 *     crab_rmsnorm(x, layer[0] input_layernorm) -> h ; crab_llama_layers(...) ; lm_head = crab_gemm_bf16(h, ...) ;
 *     crab_greedy_select ; crab_advance.
 * The functions only issue launches of the entry points above on `stream` (capturable into a HIP graph); they keep no state.
 *
 * crab_linear_group: the projections that consume one input, packed (crab_amd/peft_hyper.py PackedLinearGroup):
 *     W [N, K] (rows of the members stacked: q|k|v; o; gate|up INTERLEAVED (gate_i, up_i); down), optional bias [N],
 *     RA [tcols, K] = per member nl route rows then r lora_A rows, zero padded to tcols = round_up(nproj*(nl+r), 16),
 *     B2 [N, ucols] = block-structured lora_B (member p, expert i, rank j at column p*nl*r + i*r + j), ucols = round_up(nproj*nl*r, 32).
 *     RA == NULL: plain linear.
 * crab_llama_layer: the four groups of a layer, the RMSNorm weight behind the attention block, and - because the norm that
 *     FOLLOWS a layer is fused into its down-projection epilogue - the NEXT norm's weight (next layer's input_layernorm, or
 *     model.norm after the last layer) and the next layer's q|k|v group (its router is evaluated ahead when M <= CRAB_DECODE_MAX_ROWS).
 * crab_llama_io: caller-owned activations for M = B*S rows.  In: x = residual stream, h = rmsnorm(x) * this layer's
 *     input_layernorm.  Out: x updated, h = rmsnorm(x) * next_norm_w.  qkv / att / act / u / u2 are scratch.
 *     k_cache / v_cache: rows of these B sequences in layer 0's cache [B, Hk, Tmax, d]; layer l lives cache_layer_stride
 *     ELEMENTS further.  Prefill (S rows per sequence at positions pos0 .. pos0+S-1): vt [B, Hk, d, vt_ld] receives V^T for
 *     the flash kernel.  Decode (S == 1): position = pos0 + pos_dev[0] (pos_dev may be NULL), RoPE + KV append are fused
 *     behind the q|k|v GEMM.  u_qkv_ready (in/out): u2 holds the q|k|v router output of the layer about to run.
 *     route_ws: crab_hyperlora_route_workspace(M, max K, max tcols) bytes; splitk_ws: the crab_gemm_desc.workspace (M <= CRAB_DECODE_MAX_ROWS). */
typedef struct {
    const void* W; const void* bias; const void* RA; const void* B2;
    int64_t ldw, ldra, ldb2;
    int32_t N, K, nproj, nl, r, tcols, ucols;
    float scaling;     /* lora_alpha / r */
} crab_linear_group;

typedef struct {
    crab_linear_group qkv, o, gu, down;
    const void* post_attention_norm_w;
    const void* next_norm_w;
    const crab_linear_group* next_qkv;   /* NULL after the last layer */
    int32_t H, Hk, d;
    float rms_eps;
    int32_t norm_w_fp32;                 /* ABI 9: post_attention_norm_w / next_norm_w are fp32 [D] (needs crab_llama_io.x_fp32); 0: bf16 */
} crab_llama_layer;

typedef struct {
    void* x; void* h; void* qkv; void* att; void* act; void* u; void* u2;
    int64_t ldx, ldh, ldqkv, ldatt, ldact, ldu;
    void* route_ws; int64_t route_ws_bytes;
    void* splitk_ws; int64_t splitk_ws_bytes;
    const float* rope_tab;               /* crab_rope_table, >= Tmax positions */
    void* k_cache; void* v_cache; int64_t cache_layer_stride;
    void* vt; int64_t vt_ld;             /* prefill only; vt == NULL selects the decode sequence in crab_llama_layers */
    const int32_t* pos_dev;
    int32_t B, S, Tmax, pos0;
    int32_t u_qkv_ready;
    /* optional, decode only: crab_attn_decode_rope_workspace(B, H, d) bytes whose tickets (the last B * H * 4 bytes) the caller zero-filled
     * once.  When given and B * H < CRAB_ATTN_SPLIT_BELOW (few blocks per head cannot hide the KV stream's latency) the q|k|v projection leaves its raw
     * row and crab_attn_decode_rope does RoPE + KV append + split-context attention in one launch. */
    void* attn_ws; int64_t attn_ws_bytes;
    /* x is the FP32 residual stream: fp32 [M, ldx] (ldx in fp32 elements), read and written by the o_proj / down_proj epilogues and read by
     * the norms; h and everything else stay bf16.  0: x is bf16 (the r01-r03 storage). */
    int32_t x_fp32;
    /* optional, decode only (ABI 9): int32 [B] - the RAGGED decode batch.  Sequence b's first key sits in cache slot row_off[b] (its prompt was
     * prefilled into slots row_off[b] .. row_off[b] + S_b - 1 by a prefill call whose k_cache / v_cache pointers were advanced by row_off[b] * d
     * elements), so one append slot pos0 + pos_dev[0] serves every row: the new token of row b is rotated at slot - row_off[b]
     * (crab_gemm_desc.rope_row_off) and attends keys row_off[b] .. slot (crab_attn_decode_masked's kv_start).  Several generate() calls of the
     * eval loop (each with its own prompt length and left padding) then decode as ONE batch that streams the weights once per step.
     * NULL = every sequence starts at slot 0. */
    const int32_t* row_off;
    /* optional, PREFILL only (ABI 9): the left-pad attention_mask and position_ids of forward() (models/unified_llama.py:149-160) behind the
     * sequencer - pos_ids int32 [B, ld_pos >= S]: token (b, s) is rotated at pos_ids[b * ld_pos + s] (its K / V rows still land in cache slot
     * pos0 + s: crab_gemm_desc.rope_pos_ids / crab_qkv_rope_split_ids); kv_start int32 [B]: keys below kv_start[b] are invisible to sequence b
     * (crab_attn_desc.kv_start).  Together they prefill sequences of DIFFERENT lengths in one call, right-aligned: sequence b padded in front
     * to S rows, kv_start[b] = S - S_b, pos_ids[b][s] = max(s - kv_start[b], 0) - every real token sees exactly the keys and positions of its
     * own call, and its cache rows land where the ragged decode batch (row_off = kv_start) expects them.  NULL = positions pos0 + s, no mask. */
    const int32_t* pos_ids; int64_t ld_pos;
    const int32_t* kv_start;
    /* optional, PREFILL through crab_llama_layers only (ABI 9): the caller needs nothing but the LAST row of every sequence after the last
     * layer (generate(): lm_head reads one row per sequence; the reference computes all S and drops S - 1, modeling_llama.py:1260).  The last
     * layer then projects q|k|v for all rows (the cache needs every row's k / v) and runs attention, o_proj, the MLP and the final norm for
     * the B last rows only, as a one-row-per-sequence step (crab_attn_decode over the S cached keys, the decode-regime GEMMs): the other
     * S - 1 rows of the last layer's attention / MLP are dead compute (2.6 % of a 32-layer prefill).  Out: h[b, :] (row b of h, b < B) =
     * rmsnorm(x_last_row_of_sequence_b) * next_norm_w; x is NOT updated for the last layer; att, act, qkv, h are scratch as before (their
     * first B rows and qkv's storage are reused).  Needs S > 1, M = B * S rows of act with ldact >= H * d, qkv storage >= B * D * 4 bytes
     * (true for every decoder: M * ldqkv * 2 bytes).  0: every row through every layer (forward(): all logits). */
    int32_t last_rows_only;
} crab_llama_io;

int crab_sizeof_llama_layer(void);
int crab_sizeof_llama_io(void);
int crab_llama_layer_prefill(crab_ctx* ctx, void* stream, const crab_llama_layer* layer, crab_llama_io* io, int layer_index);
int crab_llama_layer_decode(crab_ctx* ctx, void* stream, const crab_llama_layer* layer, crab_llama_io* io, int layer_index);
/* all layers of a stack in order (layers[l].next_qkv == &layers[l+1].qkv); prefill when io->vt != NULL, decode otherwise */
int crab_llama_layers(crab_ctx* ctx, void* stream, const crab_llama_layer* layers, int n_layers, crab_llama_io* io);

/* ---------------------------------------------------------------------------------------------
 * Fused encoder blocks: the launch sequence of ONE encoder layer behind one call, so that a C caller runs a whole clip
 * (examples/clip_demo.c) without re-implementing the sequencing of crab_amd/multimodal_encoder.py.  Like crab_llama_layers they only
 * issue launches of the entry points above on `stream` and keep no state.
 *   crab_clip_layer    HF CLIPEncoderLayer as the reference drives it (models/multimodal_encoder.py:52-84; pre-LN, quick-GELU MLP,
 *                      q|k|v packed as one [3D, D] matrix with bias)
 *   crab_beats_layer   models/beats/backbone.py:214-275 (post-LN deep-norm: x = LN(alpha x + sublayer)); attention :432-684 with the gated
 *                      relative position bias: io->bias [H, S, S] fp32 from crab_beats_relpos_bias (once per forward, :131-137),
 *                      io->gate [B, H, S] fp32 scratch for crab_beats_gru_gate; grep_w == NULL: no gate
 *   crab_qformer_layer models/Qformer.py:404-476 with cross_attention_freq = 1 and the query FFN (:483-486): x = the B * S query rows,
 *                      io->enc = the B * enc_rows encoder rows (already layer-normed by the projector, multimodal_encoder.py:119-144, 226-244)
 * crab_dense = one nn.Linear (W [N, K] row stride ldw, bias [N] or NULL); crab_ln = LayerNorm weight / bias / eps (+ fp32: the parameters are fp32).
 * crab_enc_io: caller-owned rows for M = B * S tokens.  x [M, width] in / out (dense rows); a [M, width], y [M, width] scratch;
 *   qkv [max(M, B * enc_rows), 3 * width (2 * width for the Q-Former)]; att [M, width]; f [M, ffn width];
 *   vt: V^T scratch of vt_bytes >= B * H * d * round8(keys) * 2; workspace: the crab_gemm_desc.workspace handed to every GEMM with <= CRAB_DECODE_MAX_ROWS rows. */
typedef struct { const void* W; const void* bias; int64_t ldw; int32_t N, K; } crab_dense;
typedef struct { const void* w; const void* b; float eps; int32_t fp32; /* ABI 9: w and b are fp32 [D] (0: bf16) */ } crab_ln;
typedef struct { crab_ln ln1, ln2; crab_dense qkv, out, fc1, fc2; int32_t H; } crab_clip_layer_w;
typedef struct { crab_dense qkv, out, fc1, fc2; crab_ln ln_attn, ln_final; const void* grep_w; const void* grep_b; const void* grep_a; int32_t H; float alpha; } crab_beats_layer_w;
typedef struct { crab_dense sq, skv, so; crab_ln sln; crab_dense cq, ckv, co; crab_ln cln; crab_dense iq, oq; crab_ln oln; int32_t H; } crab_qformer_layer_w;
typedef struct {
    void* x; void* a; void* y; void* qkv; void* att; void* f; void* vt; int64_t vt_bytes;
    const void* enc; int32_t enc_rows;              /* Q-Former only */
    const float* bias; float* gate;                 /* BEATs only */
    void* workspace; int64_t workspace_bytes;
    int32_t B, S;
    /* crab_clip_layer only (pre-LN residual tower): x is fp32 [M, width]; 0: bf16 */
    int32_t x_fp32;
} crab_enc_io;
int crab_clip_layer(crab_ctx* ctx, void* stream, const crab_clip_layer_w* w, crab_enc_io* io);
int crab_beats_layer(crab_ctx* ctx, void* stream, const crab_beats_layer_w* w, crab_enc_io* io);
int crab_qformer_layer(crab_ctx* ctx, void* stream, const crab_qformer_layer_w* w, crab_enc_io* io);
int crab_sizeof_enc_io(void);
int crab_sizeof_clip_layer_w(void);
int crab_sizeof_beats_layer_w(void);
int crab_sizeof_qformer_layer_w(void);

/* ---------------------------------------------------------------------------------------------
 * SegModule pixel path (models/multimodal_encoder.py:268-543, 891-1444).  Feature maps are token-major [h*w, C] bf16.
 *  im2col3x3       : Conv2d(k=3,pad=1) operand, out[(b,y,x), (ky*3+kx)*C + c]                      (image_feature_neck :316-332)
 *  pixel_shuffle2x : ConvTranspose2d(k=2,s=2) after its GEMM: g[h*w, (dy,dx,co)] -> out[(2h)(2w), Co] + bias  (:937-949)
 *  bilinear        : F.interpolate(mode='bilinear', align_corners=False); strided input (bf16|fp32), fp32 [C,H,W] out,
 *                    out = beta*out + alpha*interp                                                  (:436, :523-531)
 *  dense_pe        : PositionEmbeddingRandom.forward (:825-839), G = positional_encoding_gaussian_matrix fp32 [2,F] -> pe[h*w, 2F]
 *  group_mean      : out[g] = scale * sum_{k<T} in[g*T+k]  (fused_pred_embeddings :388-393)
 *  add_rows        : out[m] = a[m] + b[m % brows]  (queries+query_pe, keys+key_pe, +level_embed, +no_mask_embed)
 *  mask_gate       : src[m,:] *= sigmoid(mean_c prev[m,c]) + 1                                     (:1112-1114)
 *  act_inplace     : x = act(x), CRAB_ACT_* */
int crab_im2col3x3(crab_ctx* ctx, void* stream, const void* in, void* out, int B, int h, int w, int C);
int crab_pixel_shuffle2x(crab_ctx* ctx, void* stream, const void* g, const void* bias, void* out, int h, int w, int Co);
int crab_bilinear(crab_ctx* ctx, void* stream, const void* in, int in_fp32, int64_t sc, int64_t sy, int64_t sx, int C, int h, int w,
                  float* out, int H, int W, float alpha, float beta);
int crab_dense_pe(crab_ctx* ctx, void* stream, const void* G, void* pe, int h, int w, int F);
int crab_add_rows(crab_ctx* ctx, void* stream, const void* a, int64_t lda, const void* b, int64_t ldb, int brows, void* out, int64_t ldo,
                  int M, int D);
int crab_mask_gate(crab_ctx* ctx, void* stream, const void* prev, int64_t ldp, int ncls, void* src, int64_t lds, int M, int D);
int crab_group_mean(crab_ctx* ctx, void* stream, const void* in, int64_t ldi, void* out, int64_t ldo, int G, int T, int D, float scale);
int crab_act_inplace(crab_ctx* ctx, void* stream, void* x, int64_t n, int act);
/* mask_labels: what the eval loops write to the PNG from a predicted mask pred [C, hw] fp32 (SegModule output): C == 1 -> 255 where
 * sigmoid(pred) > 0.5 else 0 (scripts/quick_start.py:313-318); C > 1 -> argmax over the class planes, first maximum
 * (utils/avss_utils.py:291-292 softmax + argmax); out [hw] uint8 */
int crab_mask_labels(crab_ctx* ctx, void* stream, const float* pred, int C, int64_t hw, uint8_t* out);

/* ---------------------------------------------------------------------------------------------
 * Segmentation metrics (SURVEY.md 8 f-1 anchors utils/avss_utils.py:8-96, 379-435): what the reference's pixel-task eval loops compute
 * from a predicted mask right behind generate_avs, on `pred_mask.cpu()` (scripts/quick_start.py:118-119, 198-199, 267-268, 342, 395;
 * scripts/finetune/inference_hyper_lora.py:658-659, 804-805, 981-982, 1063, 1177).  All buffers are DEVICE memory owned by the caller; every
 * pixel count is an exact int32; the fp32 ratios are formed from the counts in the reference's operation order, one rounding per operation.
 *  crab_mask_iou     : replaces mask_iou (avss_utils.py:22-47) and metric_s_for_null (:8-19).  pred [N, hw] fp32 logits (mask where pred > 0 =
 *      sigmoid(pred) > 0.5), target [N, hw] fp32 in {0, 1} or NULL (metric_s only).  counts [N][6] int32 = {pred, target, pred & target,
 *      pred | target, !pred & !target, target pixels outside {0, 1}}; out[0] = sum_n inter_n / (union_n + eps) / N with an empty target's
 *      {inter, union} replaced by {!pred & !target, hw}; out[1] = sqrt(sum_n pred_n / (N hw)).
 *  crab_fmeasure     : replaces Eval_Fmeasure + _eval_pr (:50-96).  pred / gt [N, hw] fp32 (gt in {0, 1}), thresholds [T] ascending fp32
 *      (the reference: torch.linspace(0, 1 - 1e-10, 255)), T <= 1024.  ge [N][2][T] int32 = {#(gt & sigmoid(pred) >= th_i), #(sigmoid(pred) >= th_i)},
 *      ysum [N][2] int32 = {gt pixels, gt pixels outside {0, 1}}, fscore [N][T] = (1 + beta2) P R / (beta2 P + R) with NaN -> 0,
 *      score [T] = mean of fscore over the images whose gt is not empty (image order), best [2] = {max_i score[i], images counted}.
 *  crab_miou_fscore  : replaces calc_color_miou_fscore / _batch_miou_fscore (:379-435).  pred [BF, C, hw] fp32 class logits (argmax of the
 *      logits, first maximum, a NaN counting as the maximum like torch.argmax; the reference takes the argmax of softmax(pred, dim = 1), which is the
 *      same index except where two DISTINCT logits round to the same fp32 probability - then the reference keeps the first of them and this kernel the
 *      larger one: a tie of the probabilities at the top needs logits within ~6e-8 relative of each other), target [BF, hw] int64 class ids (ids outside [0, C) are counted nowhere; a negative id
 *      also removes the pixel's prediction, as the reference's `predict * (target > 0)` does after its +1 shift), C <= 1024.
 *      areas [BF][3][C] int32 = {TP, TP + FP, TP + FN} (the three torch.histc calls), iou_fc [BF][C] = TP / (2.22e-16 + union),
 *      ious / fscores / cls_count [C] = the per-class sums over the frames in frame order, vid_miou [BF] = sum_c iou / #{iou != 0}. */
int crab_mask_iou(crab_ctx* ctx, void* stream, const float* pred, const float* target, int N, int64_t hw, float eps, int32_t* counts, float* out);
int crab_fmeasure(crab_ctx* ctx, void* stream, const float* pred, const float* gt, int N, int64_t hw, const float* thresholds, int T, double beta2,
                  int32_t* ge, int32_t* ysum, float* fscore, float* score, float* best);
int crab_miou_fscore(crab_ctx* ctx, void* stream, const float* pred, const int64_t* target, int BF, int C, int64_t hw, double beta2, int32_t* areas,
                     float* iou_fc, float* ious, float* fscores, float* cls_count, float* vid_miou);
/* color_to_label: the AVSS ground truth from its colour map (dataset/quick_start_dataset.py:63-73 color_mask_to_label, :534-539): rgb [hw, 3] uint8
 * (the PIL mask after `.convert('RGB').resize((224, 224), NEAREST)`), palette [n, 3] uint8 (get_v2_pallete :35-59 = the PASCAL-VOC bit-shuffle table,
 * n = 71), both DEVICE; out [hw] int64 = the first palette index whose colour equals the pixel, 0 when none does. */
int crab_color_to_label(crab_ctx* ctx, void* stream, const uint8_t* rgb, int64_t hw, const uint8_t* palette, int n, int64_t* out);

/* ---------------------------------------------------------------------------------------------
 * Input front-end: what the reference's dataset code does on the CPU right before generate().
 *  crab_bicubic_ksize / crab_bicubic_coeffs (HOST arrays): Pillow 10.4 Resample.c precompute_coeffs + normalize_coeffs_8bpc
 *      for BICUBIC over the whole image: bounds[out][2] = {first tap, tap count}, kk[out][ksize] 22-bit fixed-point taps.
 *      Replaces the coefficient set-up inside `Image.resize` called by transformers' CLIPImageProcessor.resize
 *      (dataset/quick_start_dataset.py:315 `self.video_processor.preprocess(frames, return_tensors='pt')`).
 *  crab_resample_u8 : one separable 8-bit pass (ImagingResampleHorizontal_8bpc / Vertical_8bpc), uint8 [N,H,W,C] -> uint8
 *      [N,H,out,C] (horizontal) or [N,out,W,C]; bounds / kk are DEVICE copies.  Bit-exact with Pillow.
 *  crab_clip_normalize : centre crop + rescale + normalize + HWC->CHW: out[n,c,y,x] = (src[n,top+y,left+x,c]*rescale - mean[c]) / std[c],
 *      fp32 or bf16 (image_processing_clip.py center_crop / rescale / normalize).  mean3 / std3 are HOST arrays.
 *  crab_kaldi_fbank : torchaudio.compliance.kaldi.fbank as called by dataset/audio_processor.py:29-41 (25 ms / 10 ms frames
 *      at 16 kHz, dither 0, DC removal, pre-emphasis, povey window, 512-point FFT, power, 128 mel bins 20..8000 Hz, log),
 *      then (x - out_sub) * out_scale.  wave fp32 [n_wave][ldw], scaled by in_scale (2^15); window400 [400] and
 *      mel_t [257][128] (transposed filterbank incl. the zero Nyquist row) are DEVICE fp32 arrays owned by the caller;
 *      out fp32 [n_wave][frames][128], frames = crab_kaldi_fbank_frames(n_samples) = 1 + (n - 400) / 160. */
int crab_bicubic_ksize(int in_size, int out_size);
int crab_bicubic_coeffs(int in_size, int out_size, int32_t* bounds, int32_t* kk, int ksize);
int crab_resample_u8(crab_ctx* ctx, void* stream, const void* src, int N, int H, int W, int C, void* dst, int out_size, int horizontal,
                     const int32_t* bounds, const int32_t* kk, int ksize);
int crab_clip_normalize(crab_ctx* ctx, void* stream, const void* src, int N, int H, int W, int top, int left, int size, void* out,
                        int out_bf16, const float* mean3, const float* std3, float rescale);
int crab_kaldi_fbank_frames(int n_samples);
int crab_kaldi_fbank(crab_ctx* ctx, void* stream, const float* wave, int64_t ldw, int n_wave, int n_samples, float in_scale,
                     float preemphasis, const float* window400, const float* mel_t, float* out, float out_sub, float out_scale);

/* ---------------------------------------------------------------------------------------------
 * VQGAN mask tokenizer (models/multimodal_encoder.py:546-601 MaskEncoder, taming_transformer/modules.py,
 * quantize.py:272-330, vqgan.py:54-99).  Token-major [b, h*w, C] bf16 feature maps; convolutions = im2col + crab_gemm_bf16.
 *  im2col3x3_strided : out[(b,oy,ox), (ky*3+kx)*C + c] = in[b, oy*stride+ky-pad_top, ox*stride+kx-pad_left, c] (0 outside);
 *                      Downsample (modules.py:66-72: F.pad (0,1,0,1) + Conv2d(k3, s2)) is stride 2, pads 0.
 *  groupnorm         : GroupNorm(G, C, eps) over (h*w, C/G) per (b, g) (+ x*sigmoid(x) when swish): Normalize / nonlinearity
 *                      (modules.py:29-35); workspace = crab_groupnorm_workspace bytes (fixed-order partial sums of x - k, k = the group's first
 *                      value of the image: the shifted-data variance, exact to fp32 rounding for groups of any size and spread; may run in place).
 *  upsample_nearest2x: F.interpolate(scale_factor=2, mode="nearest") (modules.py:50).
 *  softmax_rows      : out bf16 = softmax(scale * in fp32) per row (AttnBlock, modules.py:176-178).
 *  row_sqnorm        : out[n] = sum_d e[n,d]^2 (codebook norms, quantize.py:287).
 *  vq_argmin         : idx[m] = offset + first argmin_n (e2[n] - 2 dots[m,n])  (quantize.py:286-290; dots = z . e^T from the GEMM). */
int crab_im2col3x3_strided(crab_ctx* ctx, void* stream, const void* in, void* out, int B, int h, int w, int C, int stride, int pad_top,
                           int pad_left, int oh, int ow);
int64_t crab_groupnorm_workspace(int B, int HW, int G);
int crab_groupnorm(crab_ctx* ctx, void* stream, const void* x, void* out, int B, int HW, int C, int G, float eps, const void* weight,
                   const void* bias, int swish, void* workspace, int64_t workspace_bytes);
int crab_upsample_nearest2x(crab_ctx* ctx, void* stream, const void* in, void* out, int B, int h, int w, int C);
int crab_softmax_rows(crab_ctx* ctx, void* stream, const float* in, int64_t ldi, void* out, int64_t ldo, int M, int N, float scale);
int crab_row_sqnorm(crab_ctx* ctx, void* stream, const void* e, int64_t lde, int N, int D, float* out);
int crab_vq_argmin(crab_ctx* ctx, void* stream, const float* dots, int64_t ldd, const float* e2, int M, int N, int64_t* idx, int64_t offset);
/* r06 (ABI 12): the quantiser in fp32 - codebook ids are INDEX work.  z [M, D] fp32 (the unrounded latents of quant_conv), e [N, D] fp32 (the
 * codebook as the checkpoint holds it), e2 = crab_row_sqnorm_f32(e): idx[m] = offset + first argmin_n (|z_m|^2 + e2[n]) - 2 z_m . e_n, the fp32
 * expression of VectorQuantizer2.forward (quantize.py:286-290); equals the reference's ids wherever its nearest / second-nearest margin exceeds
 * fp32 summation noise GIVEN THE SAME LATENTS.  D % 4 == 0, D <= 256; workspace = crab_vq_nearest_f32_workspace(M, N) bytes.
 * crab_groupnorm_p = crab_groupnorm with fp32 weight / bias when w_fp32 != 0 (GroupNorm parameters are not matrix operands). */
int crab_row_sqnorm_f32(crab_ctx* ctx, void* stream, const float* e, int64_t lde, int N, int D, float* out);
int64_t crab_vq_nearest_f32_workspace(int M, int N);
int crab_vq_nearest_f32(crab_ctx* ctx, void* stream, const float* z, int64_t ldz, const float* e, int64_t lde, const float* e2, int M, int N, int D,
                        int64_t* idx, int64_t offset, void* workspace, int64_t workspace_bytes);
int crab_groupnorm_p(crab_ctx* ctx, void* stream, const void* x, void* out, int B, int HW, int C, int G, float eps, const void* weight, const void* bias,
                     int w_fp32, int swish, void* workspace, int64_t workspace_bytes);
/* r06 (ABI 12): the PRECISE form of the encoder in front of the quantiser (MaskEncoder.encode_mask; vqgan.py:54-63, modules.py:342-433).  Codebook
 * ids are index work and flip wherever the latents' error exceeds the nearest / second-nearest margin; bf16 conv operands alone cost ~6e-3 of the
 * latents' scale (the operand floor).  The precise form keeps activations in fp32 and gives the MFMA GEMM split operands x = hi + lo:
 *   crab_split3        fp32 [M, C] -> bf16 [M, 3 * round_up(C, 8)]: pattern 0 = [hi | lo | hi] (the A side), 1 = [hi | hi | lo] (the B side), so that
 *                      crab_gemm_bf16 over the tripled K forms x_hi.w_hi + x_lo.w_hi + x_hi.w_lo with fp32 accumulation (~2^-17 relative);
 *   crab_groupnorm_f32 GroupNorm(G, C, eps) (+ swish), fp32 in / out / parameters;  crab_add_bias_f32: x[m, c] += bias[c];
 *   crab_softmax_rows_f32: the AttnBlock's row softmax (modules.py:178-181), fp32 in / out. */
int crab_split3(crab_ctx* ctx, void* stream, const float* x, int64_t ldx, void* out, int64_t ldo, int M, int C, int pattern);
int crab_groupnorm_f32(crab_ctx* ctx, void* stream, const float* x, float* out, int B, int HW, int C, int G, float eps, const float* weight,
                       const float* bias, int swish, void* workspace, int64_t workspace_bytes);
int crab_add_bias_f32(crab_ctx* ctx, void* stream, float* x, int64_t ldx, const float* bias, int64_t M, int C);
int crab_softmax_rows_f32(crab_ctx* ctx, void* stream, const float* in, int64_t ldi, float* out, int64_t ldo, int M, int N, float scale);

/* ---------------------------------------------------------------------------------------------
 * Multi-GPU: per-clip sharding, one process per GPU, full weight replica; the ONLY exchange is a gather of fixed-size result
 * records to a root rank.  crab_gather_results is that gather over RCCL (ncclGather: every peer sends over its own xGMI link to the root);
 * RCCL is dlopen()ed at the first crab_dist_* call (CRAB_RCCL_LIB overrides the name), so a single-GPU caller never loads it.
 * The reference has no collective on this path (scripts/finetune/inference_hyper_lora.py:1466-1479 is a single-process loop);
 * crab_amd/parallel.py:gather_results is the torch.distributed form of the same exchange.
 *   crab_dist_unique_id : rank 0 makes the 128-byte rendezvous id; the CALLER ships it to the other ranks (file, socket, environment)
 *   crab_dist_init      : collective over all ranks: the communicator on ctx's device
 *   crab_gather_results : send [bytes_per_rank] (device) from every rank -> recv [world * bytes_per_rank] (device) on `root`, rank order; a
 *                         record is whatever the caller packs, e.g. {int64 clip_id, int64 ids[n_new]} per clip (+ fp32 first-step logits);
 *                         asynchronous on `stream`; recv may be NULL on the other ranks
 */
#define CRAB_DIST_ID_BYTES 128
typedef struct crab_comm crab_comm;
int crab_dist_unique_id(crab_ctx* ctx, void* id_out /* host, CRAB_DIST_ID_BYTES */);
int crab_dist_init(crab_ctx* ctx, const void* id /* host */, int world, int rank, crab_comm** out);
int crab_gather_results(crab_ctx* ctx, void* stream, crab_comm* comm, const void* send, int64_t bytes_per_rank, void* recv, int root);
int crab_dist_world(const crab_comm* comm);
int crab_dist_rank(const crab_comm* comm);
void crab_dist_destroy(crab_comm* comm);

#ifdef __cplusplus
}
#endif
#endif

// Encoder-layer sequencing behind the C-ABI: crab_clip_layer / crab_beats_layer / crab_qformer_layer (include/crab_hip.h, SURVEY.md 8b
// "fused blocks").  Host code only: every launch goes through the library's own entry points, in the order crab_amd/multimodal_encoder.py
// issued them per layer before this file existed, so the Python modules (which now make ONE call per layer) and a C-only caller
// (examples/clip_demo.c) produce the same bits.
//
//   CLIP ViT layer   (HF CLIPEncoderLayer as used by models/multimodal_encoder.py:52-84; pre-LN, quick-GELU):
//       a = LN1(x); qkv = a.Wqkv^T + b; att = softmax(q k^T / sqrt(d)) v; x2 = att.Wo^T + b + x; a = LN2(x2); f = quick_gelu(a.W1^T + b);
//       x = f.W2^T + b + x2
//       io->x_fp32: x and x2 (= io->y) are fp32 - the residual stream of the 23-layer pre-LN tower is never rounded to bf16
//   BEATs layer      (models/beats/backbone.py:214-275 post-LN deep-norm; attention :432-684 with the gated relative position bias):
//       qkv = x.Wqkv^T + b; gate = gru_gate(q); att = softmax(q k^T / sqrt(d) + gate * bias) v; x = LN(att.Wo^T + b + alpha x);
//       x = LN(gelu(x.W1^T + b).W2^T + b + alpha x)
//       io->x_fp32 (BEATs and Q-Former, post-LN): the pre-LN sums (io->y) are fp32 and the LayerNorm reads them unrounded; x itself stays bf16
//       (it is the next GEMM's operand anyway)
//   Q-Former layer   (models/Qformer.py:404-476 BertLayer with cross_attention_freq = 1, query branch :483-486):
//       self-attention over the nq query rows, post-LN; cross-attention to the m encoder rows (K / V from `enc`), post-LN; query FFN
//       (intermediate_query / output_query), post-LN
#include "crab_internal.h"
#include <math.h>

namespace {

int dense(crab_ctx* ctx, void* stream, const crab_enc_io* io, const crab_dense* w, const void* a, int64_t lda, void* c, int64_t ldc, int M, int act,
          const void* residual, int64_t ldr, float res_scale, int c_fp32 = 0, int r_fp32 = 0) {
    crab_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.A = a; d.lda = lda; d.B = w->W; d.ldb = w->ldw; d.C = c; d.ldc = ldc; d.bias = w->bias;
    d.R = residual; d.ldr = ldr; d.res_scale = res_scale; d.c_fp32 = c_fp32; d.r_fp32 = residual ? r_fp32 : 0;
    d.M = M; d.N = w->N; d.K = w->K; d.act = act; d.batch = 1; d.nb0 = 1;
    if (M <= CRAB_DECODE_MAX_ROWS) { d.workspace = io->workspace; d.workspace_bytes = io->workspace_bytes; }     // the rule of crab_amd/ops.py: gemm()
    return crab_gemm_bf16(ctx, stream, &d);
}

// bidirectional MHA over token-major projections: q rows [B*Sq, ldq] (head h at column q_col0 + h*d), k rows [B*Skv, ldk] (k_col0 + h*d);
// V^T is materialised from the packed rows `vsrc` whose v heads follow `v_skip` heads of d columns
int attention(crab_ctx* ctx, void* stream, const crab_enc_io* io, const uint16_t* q, int64_t ldq, const uint16_t* k, int64_t ldk,
              void* vsrc, int64_t ldv, int v_skip, int B, int H, int Sq, int Skv, int d, const float* bias, const float* gate) {
    const int Sp = (Skv + 7) / 8 * 8;
    if ((int64_t)B * H * d * Sp * 2 > io->vt_bytes) return crab_fail(ctx, CRAB_E_WORKSPACE, "encoder layer: vt scratch smaller than B * H * d * round8(Skv) * 2 bytes");
    int rc = crab_qkv_rope_split(ctx, stream, vsrc, ldv, nullptr, nullptr, nullptr, io->vt, Sp, B, Skv, v_skip, H, d, 1, 0, nullptr);
    if (rc) return rc;
    crab_attn_desc a;
    memset(&a, 0, sizeof(a));
    a.q = q; a.k = k; a.vt = io->vt; a.o = io->att;
    a.q_bs = (int64_t)Sq * ldq; a.q_hs = d; a.q_ss = ldq;
    a.k_bs = (int64_t)Skv * ldk; a.k_hs = d; a.k_ss = ldk;
    a.vt_bs = (int64_t)H * d * Sp; a.vt_hs = (int64_t)d * Sp; a.vt_ds = Sp;
    a.o_bs = (int64_t)Sq * H * d; a.o_ss = (int64_t)H * d;
    a.bias = bias; a.gate = gate;
    a.B = B; a.H = H; a.Hk = H; a.Sq = Sq; a.Skv = Skv; a.head_dim = d; a.causal = 0;
    a.scale = (float)(1.0 / sqrt((double)d));           // == d ** -0.5 rounded once, as the Python modules pass it
    return crab_attn_fwd(ctx, stream, &a);
}

// LayerNorm of bf16 rows, or of fp32 rows (the fp32 residual stream / fp32 pre-LN sums)
int ln(crab_ctx* ctx, void* stream, int x_fp32, const void* x, int64_t ldx, const crab_ln* w, void* y, int64_t ldy, int M, int D) {
    return crab_layernorm_p(ctx, stream, x, x_fp32, ldx, w->w, w->b, w->fp32, y, ldy, M, D, w->eps);
}

int check_io(crab_ctx* ctx, const crab_enc_io* io, bool need_f, const char* who) {
    char msg[160];
    if (!io || !io->x || !io->a || !io->qkv || !io->att || !io->vt || !io->y || (need_f && !io->f) || io->B <= 0 || io->S <= 0) {
        snprintf(msg, sizeof(msg), "%s: x, a, qkv, att, vt, y%s and positive B, S are required", who, need_f ? ", f" : "");
        return crab_fail(ctx, CRAB_E_INVALID, msg);
    }
    return CRAB_OK;
}

}  // namespace

extern "C" {

int crab_clip_layer(crab_ctx* ctx, void* stream, const crab_clip_layer_w* w, crab_enc_io* io) {
    if (!ctx) return CRAB_E_INVALID;
    if (!w) return crab_fail(ctx, CRAB_E_INVALID, "clip_layer: null weights");
    int rc = check_io(ctx, io, true, "clip_layer");
    if (rc) return rc;
    const int D = w->out.N, H = w->H, M = io->B * io->S;
    if (H <= 0 || D % H || w->qkv.N != 3 * D || w->qkv.K != D || w->out.K != D || w->fc1.K != D || w->fc2.N != D || w->fc2.K != w->fc1.N)
        return crab_fail(ctx, CRAB_E_INVALID, "clip_layer: shapes do not chain (qkv [3D, D], out [D, D], fc1 [I, D], fc2 [D, I])");
    const int d = D / H;
    uint16_t* qkv = (uint16_t*)io->qkv;
    const int xf = io->x_fp32 ? 1 : 0;
    if ((rc = ln(ctx, stream, xf, io->x, D, &w->ln1, io->a, D, M, D))) return rc;
    if ((rc = dense(ctx, stream, io, &w->qkv, io->a, D, qkv, 3 * D, M, CRAB_ACT_NONE, nullptr, 0, 1.0f))) return rc;
    if ((rc = attention(ctx, stream, io, qkv, 3 * D, qkv + D, 3 * D, qkv, 3 * D, H, io->B, H, io->S, io->S, d, nullptr, nullptr))) return rc;
    if ((rc = dense(ctx, stream, io, &w->out, io->att, D, io->y, D, M, CRAB_ACT_NONE, io->x, D, 1.0f, xf, xf))) return rc;  // y = x + attn
    if ((rc = ln(ctx, stream, xf, io->y, D, &w->ln2, io->a, D, M, D))) return rc;
    if ((rc = dense(ctx, stream, io, &w->fc1, io->a, D, io->f, w->fc1.N, M, CRAB_ACT_QUICK_GELU, nullptr, 0, 1.0f))) return rc;
    return dense(ctx, stream, io, &w->fc2, io->f, w->fc1.N, io->x, D, M, CRAB_ACT_NONE, io->y, D, 1.0f, xf, xf);            // x = y + mlp
}

int crab_beats_layer(crab_ctx* ctx, void* stream, const crab_beats_layer_w* w, crab_enc_io* io) {
    if (!ctx) return CRAB_E_INVALID;
    if (!w) return crab_fail(ctx, CRAB_E_INVALID, "beats_layer: null weights");
    int rc = check_io(ctx, io, true, "beats_layer");
    if (rc) return rc;
    const int E = w->out.N, H = w->H, M = io->B * io->S;
    if (H <= 0 || E % H || w->qkv.N != 3 * E || w->qkv.K != E || w->out.K != E || w->fc1.K != E || w->fc2.N != E || w->fc2.K != w->fc1.N)
        return crab_fail(ctx, CRAB_E_INVALID, "beats_layer: shapes do not chain (qkv [3E, E], out [E, E], fc1 [F, E], fc2 [E, F])");
    if (w->grep_w && (!w->grep_b || !w->grep_a || !io->gate || !io->bias)) return crab_fail(ctx, CRAB_E_INVALID, "beats_layer: the gated bias needs grep_b, grep_a, io->gate and io->bias");
    const int d = E / H;
    uint16_t* qkv = (uint16_t*)io->qkv;
    if ((rc = dense(ctx, stream, io, &w->qkv, io->x, E, qkv, 3 * E, M, CRAB_ACT_NONE, nullptr, 0, 1.0f))) return rc;
    const float* gate = nullptr;
    if (w->grep_w) {
        if ((rc = crab_beats_gru_gate(ctx, stream, qkv, 3 * E, w->grep_w, w->grep_b, w->grep_a, io->gate, io->B, io->S, H, d))) return rc;
        gate = io->gate;
    }
    if ((rc = attention(ctx, stream, io, qkv, 3 * E, qkv + E, 3 * E, qkv, 3 * E, H, io->B, H, io->S, io->S, d, io->bias, gate))) return rc;
    const int yf = io->x_fp32 ? 1 : 0;
    if ((rc = dense(ctx, stream, io, &w->out, io->att, E, io->y, E, M, CRAB_ACT_NONE, io->x, E, w->alpha, yf, 0))) return rc;       // y = alpha x + attn
    if ((rc = ln(ctx, stream, yf, io->y, E, &w->ln_attn, io->x, E, M, E))) return rc;
    if ((rc = dense(ctx, stream, io, &w->fc1, io->x, E, io->f, w->fc1.N, M, CRAB_ACT_GELU, nullptr, 0, 1.0f))) return rc;
    if ((rc = dense(ctx, stream, io, &w->fc2, io->f, w->fc1.N, io->y, E, M, CRAB_ACT_NONE, io->x, E, w->alpha, yf, 0))) return rc;
    return ln(ctx, stream, yf, io->y, E, &w->ln_final, io->x, E, M, E);
}

int crab_qformer_layer(crab_ctx* ctx, void* stream, const crab_qformer_layer_w* w, crab_enc_io* io) {
    if (!ctx) return CRAB_E_INVALID;
    if (!w) return crab_fail(ctx, CRAB_E_INVALID, "qformer_layer: null weights");
    int rc = check_io(ctx, io, true, "qformer_layer");
    if (rc) return rc;
    if (!io->enc || io->enc_rows <= 0) return crab_fail(ctx, CRAB_E_INVALID, "qformer_layer: enc [B * enc_rows, enc_width] is required");
    const int h = w->so.N, H = w->H, nq = io->S, m = io->enc_rows, B = io->B, M = B * nq;
    if (H <= 0 || h % H || w->sq.N != h || w->sq.K != h || w->skv.N != 2 * h || w->skv.K != h || w->so.K != h || w->cq.N != h || w->cq.K != h ||
        w->ckv.N != 2 * h || w->co.N != h || w->co.K != h || w->iq.K != h || w->oq.N != h || w->oq.K != w->iq.N)
        return crab_fail(ctx, CRAB_E_INVALID, "qformer_layer: shapes do not chain (query [h, h], key|value [2h, .], dense [h, h], FFN [i, h] / [h, i])");
    const int d = h / H;
    uint16_t* kv = (uint16_t*)io->qkv;                  // [max(B nq, B m), 2h]
    // ---- self-attention over the query rows (Qformer.py:171-277), post-LN (:287-291)
    if ((rc = dense(ctx, stream, io, &w->sq, io->x, h, io->a, h, M, CRAB_ACT_NONE, nullptr, 0, 1.0f))) return rc;
    if ((rc = dense(ctx, stream, io, &w->skv, io->x, h, kv, 2 * h, M, CRAB_ACT_NONE, nullptr, 0, 1.0f))) return rc;
    if ((rc = attention(ctx, stream, io, (const uint16_t*)io->a, h, kv, 2 * h, kv, 2 * h, 0, B, H, nq, nq, d, nullptr, nullptr))) return rc;
    const int yf = io->x_fp32 ? 1 : 0;
    if ((rc = dense(ctx, stream, io, &w->so, io->att, h, io->y, h, M, CRAB_ACT_NONE, io->x, h, 1.0f, yf, 0))) return rc;
    if ((rc = ln(ctx, stream, yf, io->y, h, &w->sln, io->x, h, M, h))) return rc;
    // ---- cross-attention to the encoder rows
    if ((rc = dense(ctx, stream, io, &w->cq, io->x, h, io->a, h, M, CRAB_ACT_NONE, nullptr, 0, 1.0f))) return rc;
    if ((rc = dense(ctx, stream, io, &w->ckv, io->enc, w->ckv.K, kv, 2 * h, B * m, CRAB_ACT_NONE, nullptr, 0, 1.0f))) return rc;
    if ((rc = attention(ctx, stream, io, (const uint16_t*)io->a, h, kv, 2 * h, kv, 2 * h, 0, B, H, nq, m, d, nullptr, nullptr))) return rc;
    if ((rc = dense(ctx, stream, io, &w->co, io->att, h, io->y, h, M, CRAB_ACT_NONE, io->x, h, 1.0f, yf, 0))) return rc;
    if ((rc = ln(ctx, stream, yf, io->y, h, &w->cln, io->x, h, M, h))) return rc;
    // ---- query FFN
    if ((rc = dense(ctx, stream, io, &w->iq, io->x, h, io->f, w->iq.N, M, CRAB_ACT_GELU, nullptr, 0, 1.0f))) return rc;
    if ((rc = dense(ctx, stream, io, &w->oq, io->f, w->iq.N, io->y, h, M, CRAB_ACT_NONE, io->x, h, 1.0f, yf, 0))) return rc;
    return ln(ctx, stream, yf, io->y, h, &w->oln, io->x, h, M, h);
}

int crab_sizeof_enc_io(void) { return (int)sizeof(crab_enc_io); }
int crab_sizeof_clip_layer_w(void) { return (int)sizeof(crab_clip_layer_w); }
int crab_sizeof_beats_layer_w(void) { return (int)sizeof(crab_beats_layer_w); }
int crab_sizeof_qformer_layer_w(void) { return (int)sizeof(crab_qformer_layer_w); }

}  // extern "C"

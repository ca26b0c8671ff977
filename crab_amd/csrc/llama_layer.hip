// Decoder-layer sequencing behind the C-ABI: crab_llama_layer_prefill / crab_llama_layer_decode / crab_llama_layers
// (include/crab_hip.h, SURVEY.md 8b "fused blocks").  Host code only: every launch goes through the library's own entry points,
// in the order crab_amd/decoder.py issued them per layer before this file existed (and still does when a per-launch profiler is
// attached), so the two sequencers are interchangeable bit for bit (tests/test_model_gpu.py::test_native_layer_sequencer_*).
//
// One layer (models/modeling_llama.py:805-827, models/qwen/modeling_qwen2.py:202-317; hyper-LoRA peft_hyper/tuners/lora.py:338-350):
//     u    = route(h) for q|k|v                     (skipped when the previous layer's epilogue produced it: io->u_qkv_ready)
//     qkv  = [h | u] . [Wqkv | Bcat]^T (+ bias)      decode: RoPE + KV append ride on this GEMM;  prefill: crab_qkv_rope_split
//     att  = flash attention (prefill, causal) | KV-streaming attention (decode, ctx read from pos_dev)
//     x   += [att | u] . [Wo | Bcat]^T ; h = rmsnorm(x) * post_attention_layernorm   (+ the gate|up router ahead, M <= CRAB_DECODE_MAX_ROWS)
//     act  = silu(gate(h)) * up(h)                   one GEMM over the interleaved gate|up rows, SwiGLU in its epilogue
//     x   += [act | u] . [Wdown | Bcat]^T ; h = rmsnorm(x) * next_norm_w              (+ the next layer's q|k|v router ahead)
// io->x_fp32: x is fp32 (the residual adds of modeling_llama.py:805-827 are never rounded to bf16); h and every GEMM operand stay bf16.
#include "crab_internal.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>
using std::max;

namespace {

struct GroupCall {
    const void* x; int64_t ldx;              // input rows [M, K]
    void* out; int64_t ldc;                  // output rows
    const void* residual; int64_t ldr;       // optional residual (added after the activation)
    int act;
    const void* norm_w; void* norm_out; int64_t ld_norm; float eps;   // optional fused post-RMSNorm
    const crab_linear_group* route_next; void* route_u;                // optional router of the next group on the post-norm rows
    const void* u_ready;                     // router output of THIS group already computed by a producer epilogue
    bool rope;                               // decode: RoPE + KV append fused behind this (q|k|v) GEMM
    int rope_prefill_S;                      // prefill: rows per sequence; q / k rotate in the projection's epilogue when the library can (crab_gemm_fuses_prefill_rope)
    int* fused_prefill_rope;                 // out: whether it did
};

int check_group(crab_ctx* ctx, const crab_linear_group* g, const char* name) {
    if (!g->W || g->N <= 0 || g->K <= 0 || g->ldw < g->K) {
        char msg[128];
        snprintf(msg, sizeof(msg), "llama_layer: group %s needs W, positive N / K and ldw >= K", name);
        return crab_fail(ctx, CRAB_E_INVALID, msg);
    }
    if (g->RA && (!g->B2 || g->nproj < 1 || g->nl < 1 || g->r < 1 || g->tcols < g->nproj * (g->nl + g->r) || (g->tcols & 15) ||
                  g->ucols < g->nproj * g->nl * g->r || (g->ucols & 7))) {
        char msg[160];
        snprintf(msg, sizeof(msg), "llama_layer: adapter of group %s needs B2, tcols = pad16(nproj (nl + r)) and ucols >= nproj nl r, multiple of 8", name);
        return crab_fail(ctx, CRAB_E_INVALID, msg);
    }
    return CRAB_OK;
}

// PackedLinearGroup.__call__ (crab_amd/peft_hyper.py): router (unless ready) + the K-extended GEMM with its fused epilogues
int run_group(crab_ctx* ctx, void* stream, const crab_linear_group* g, const crab_llama_io* io, const crab_llama_layer* L, int M,
              const GroupCall& c, void* k_cache, void* v_cache) {
    crab_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.A = c.x; d.lda = c.ldx;
    d.B = g->W; d.ldb = g->ldw;
    d.C = c.out; d.ldc = c.ldc;
    d.bias = g->bias;
    d.R = c.residual; d.ldr = c.ldr;
    if (c.residual && io->x_fp32) d.c_fp32 = d.r_fp32 = 1;        // the fp32 residual stream: x is read and written unrounded (o_proj, down_proj)
    d.M = M; d.N = g->N; d.K = g->K;
    d.act = c.act;
    d.res_scale = 1.0f;
    d.batch = 1; d.nb0 = 1;
    if (M <= CRAB_DECODE_MAX_ROWS) { d.workspace = io->splitk_ws; d.workspace_bytes = io->splitk_ws_bytes; }
    if (c.norm_w) { d.norm_w = c.norm_w; d.norm_out = c.norm_out; d.ld_norm = c.ld_norm; d.norm_eps = c.eps; d.norm_w_fp32 = L->norm_w_fp32; }
    if (c.route_next && c.route_next->RA && c.norm_w && M <= CRAB_DECODE_MAX_ROWS) {
        const crab_linear_group* n = c.route_next;
        d.route_RA = n->RA; d.route_ldra = n->ldra; d.route_U = c.route_u; d.route_ldu = io->ldu;
        d.route_nproj = n->nproj; d.route_nl = n->nl; d.route_r = n->r; d.route_ucols = n->ucols; d.route_scaling = n->scaling;
    }
    if (c.rope) {
        d.rope_tab = io->rope_tab; d.rope_k_cache = k_cache; d.rope_v_cache = v_cache; d.rope_pos_dev = io->pos_dev;
        d.rope_H = L->H; d.rope_Hk = L->Hk; d.rope_d = L->d; d.rope_Tmax = io->Tmax; d.rope_pos0 = io->pos0;
        d.rope_row_off = io->row_off;                              // ragged decode batch: row b rotates at slot - row_off[b]
    }
    if (c.rope_prefill_S > 1 && !io->pos_dev) {
        d.rope_tab = io->rope_tab; d.rope_k_cache = k_cache; d.rope_v_cache = v_cache; d.rope_pos_dev = nullptr;
        d.rope_H = L->H; d.rope_Hk = L->Hk; d.rope_d = L->d; d.rope_Tmax = io->Tmax; d.rope_pos0 = io->pos0; d.rope_S = c.rope_prefill_S;
        d.rope_vt = io->vt; d.rope_vt_ld = io->vt_ld;
        d.rope_pos_ids = io->pos_ids; d.rope_ld_pos = io->ld_pos;      // explicit rotary positions (NULL: pos0 + s)
        if (c.fused_prefill_rope) *c.fused_prefill_rope = crab_gemm_fuses_prefill_rope(&d);      // a function of shapes / pointers set above only
    }
    if (g->RA && !c.u_ready && M <= 16 && c.norm_w && g->nproj == 1 && crab_rowfin_enabled() && crab_rowfin_lora_ok(g->nl, g->r, g->N)) {
        // the reference's batch sizes, o_proj / down_proj: the group's [R;A] rows ride on the projection's launch and the update is applied
        // by the wide layer tail (rowfin.hip) - same choice as PackedLinearGroup.__call__ (crab_amd/peft_hyper.py)
        d.lora_RA = g->RA; d.lora_ldra = g->ldra; d.lora_nl = g->nl; d.lora_r = g->r; d.lora_scaling = g->scaling;
        d.B2 = g->B2; d.ldb2 = g->ldb2; d.K2 = g->ucols;
        return crab_gemm_bf16(ctx, stream, &d);
    }
    if (g->RA) {
        const void* u = c.u_ready;
        if (!u) {
            int rc = crab_hyperlora_route(ctx, stream, c.x, c.ldx, g->RA, g->ldra, M, g->K, g->nproj, g->nl, g->r, io->u, io->ldu, g->ucols,
                                          g->scaling, io->route_ws, io->route_ws_bytes);
            if (rc) return rc;
            u = io->u;
        }
        d.A2 = u; d.lda2 = io->ldu; d.B2 = g->B2; d.ldb2 = g->ldb2; d.K2 = g->ucols;
    }
    return crab_gemm_bf16(ctx, stream, &d);
}

int check_io(crab_ctx* ctx, const crab_llama_layer* L, const crab_llama_io* io, bool prefill) {
    if (!ctx) return CRAB_E_INVALID;
    if (!L || !io) return crab_fail(ctx, CRAB_E_INVALID, "llama_layer: null layer / io");
    if (L->H <= 0 || L->Hk <= 0 || L->d <= 0 || L->H % L->Hk) return crab_fail(ctx, CRAB_E_INVALID, "llama_layer: H, Hk, d must be positive, H % Hk == 0");
    if (!L->post_attention_norm_w || !L->next_norm_w) return crab_fail(ctx, CRAB_E_INVALID, "llama_layer: post_attention_norm_w / next_norm_w missing");
    if (L->norm_w_fp32 && !io->x_fp32) return crab_fail(ctx, CRAB_E_INVALID, "llama_layer: fp32 norm weights need the fp32 residual stream (x_fp32)");
    int rc;
    if ((rc = check_group(ctx, &L->qkv, "qkv")) || (rc = check_group(ctx, &L->o, "o")) || (rc = check_group(ctx, &L->gu, "gate|up")) ||
        (rc = check_group(ctx, &L->down, "down")))
        return rc;
    const int D = L->o.N, I = L->down.K, Nq = (L->H + 2 * L->Hk) * L->d;
    if (L->qkv.N != Nq || L->qkv.K != D || L->o.K != L->H * L->d || L->gu.K != D || L->gu.N != 2 * I || L->down.N != D)
        return crab_fail(ctx, CRAB_E_INVALID, "llama_layer: group shapes do not chain (qkv [(H + 2 Hk) d, D], o [D, H d], gate|up [2 I, D], down [D, I])");
    if (io->B <= 0 || io->S <= 0 || io->Tmax <= 0 || io->pos0 < 0) return crab_fail(ctx, CRAB_E_INVALID, "llama_layer: B, S, Tmax must be positive");
    if (!io->x || !io->h || !io->qkv || !io->att || !io->act || !io->k_cache || !io->v_cache || !io->rope_tab)
        return crab_fail(ctx, CRAB_E_INVALID, "llama_layer: x, h, qkv, att, act, k_cache, v_cache and rope_tab are required");
    if (io->x_fp32 && (((uintptr_t)io->x & 15) || (io->ldx & 3))) return crab_fail(ctx, CRAB_E_INVALID, "llama_layer: the fp32 residual stream needs 16-byte aligned rows");
    if (io->ldx < D || io->ldh < D || io->ldqkv < Nq || io->ldatt < L->H * L->d || io->ldact < I)
        return crab_fail(ctx, CRAB_E_INVALID, "llama_layer: a leading dimension is smaller than its row");
    const bool lora = L->qkv.RA || L->o.RA || L->gu.RA || L->down.RA;
    if (lora && (!io->u || !io->u2 || !io->route_ws)) return crab_fail(ctx, CRAB_E_INVALID, "llama_layer: adapted groups need u, u2 and route_ws");
    const int ucmax = max(max(L->qkv.RA ? L->qkv.ucols : 0, L->o.RA ? L->o.ucols : 0), max(L->gu.RA ? L->gu.ucols : 0, L->down.RA ? L->down.ucols : 0));
    if (lora && (io->ldu < ucmax || (io->ldu & 7))) return crab_fail(ctx, CRAB_E_INVALID, "llama_layer: ldu must cover the widest group's ucols and be a multiple of 8");
    if (prefill) {
        if (!io->vt || io->vt_ld < io->S) return crab_fail(ctx, CRAB_E_INVALID, "llama_layer_prefill: vt [B, Hk, d, vt_ld >= S] is required");
        if (io->pos0 + io->S > io->Tmax) return crab_fail(ctx, CRAB_E_INVALID, "llama_layer_prefill: rows do not fit the KV cache");
        if (io->row_off) return crab_fail(ctx, CRAB_E_INVALID, "llama_layer_prefill: row_off is a decode field (a prefill into a right-aligned cache advances k_cache / v_cache instead)");
        if (io->pos_ids && io->ld_pos < io->S) return crab_fail(ctx, CRAB_E_INVALID, "llama_layer_prefill: ld_pos < S");
    } else {
        if (io->S != 1) return crab_fail(ctx, CRAB_E_INVALID, "llama_layer_decode: one row per sequence (S == 1)");
        if (io->pos_ids || io->kv_start) return crab_fail(ctx, CRAB_E_INVALID, "llama_layer_decode: pos_ids / kv_start are prefill fields (a ragged decode step takes row_off)");
        if (!io->pos_dev && io->pos0 >= io->Tmax) return crab_fail(ctx, CRAB_E_INVALID, "llama_layer_decode: position outside the KV cache");
    }
    return CRAB_OK;
}

int run_layer(crab_ctx* ctx, void* stream, const crab_llama_layer* L, crab_llama_io* io, int layer_index, bool prefill, bool last_rows = false) {
    const int B = io->B, S = io->S, M = B * S;
    const int H = L->H, Hk = L->Hk, d = L->d;
    uint16_t* kc = (uint16_t*)io->k_cache + (int64_t)layer_index * io->cache_layer_stride;
    uint16_t* vc = (uint16_t*)io->v_cache + (int64_t)layer_index * io->cache_layer_stride;
    const float scale = (float)(1.0 / sqrt((double)d));   // double, then rounded once: what crab_amd/decoder.py passes
    int rc;
    // ---- q|k|v
    // small batch (B * H blocks cannot fill 256 CUs): the projection leaves its raw row and ONE launch does RoPE + KV append + attention with
    // the context split over several blocks per head (crab_attn_decode_rope) - same choice as crab_amd/decoder.py
    // (a ragged batch - io->row_off - takes the general pair: projection with the per-row rotary offset, then the attention with a first visible key per row)
    const bool fuse_attn = !prefill && !io->row_off && io->attn_ws && (long)B * H < CRAB_ATTN_SPLIT_BELOW && (d == 64 || d == 128) && (io->ldqkv & 7) == 0 &&
                           io->attn_ws_bytes >= crab_attn_decode_rope_workspace(B, H, d);
    GroupCall q{};
    q.x = io->h; q.ldx = io->ldh; q.out = io->qkv; q.ldc = io->ldqkv; q.act = CRAB_ACT_NONE;
    q.u_ready = (io->u_qkv_ready && L->qkv.RA) ? io->u2 : nullptr;
    q.rope = !prefill && !fuse_attn;
    int fused_rope = 0;
    if (prefill) { q.rope_prefill_S = S; q.fused_prefill_rope = &fused_rope; }
    if ((rc = run_group(ctx, stream, &L->qkv, io, L, M, q, kc, vc))) return rc;
    if (prefill && last_rows) {
        // ---- the LAST layer of a generate() prefill (io->last_rows_only): the cache has every row's k / v now; everything downstream is needed
        // for the last row of each sequence only.  One-row-per-sequence step over buffers the layer no longer needs:
        //   ql  = act[0:B, 0:H d]   (the rotated q of the last rows)      att = att[0:B]      xl = qkv's storage as [B, D] residual rows (fp32 | bf16)
        //   hl  = h[0:B] (post-attention norm, then overwritten by the final norm = the result)                 actl = act[0:B]
        if (fused_rope != 2 &&
            (rc = crab_qkv_rope_split_ids(ctx, stream, io->qkv, io->ldqkv, fused_rope ? nullptr : io->rope_tab, fused_rope ? nullptr : kc, vc, nullptr,
                                          0, B, S, H, Hk, d, io->Tmax, io->pos0, io->pos_dev, fused_rope ? nullptr : io->pos_ids, io->ld_pos)))
            return rc;                                                  // (no V^T: the flash kernel does not run in this layer)
        const int D = L->o.N;
        const uint16_t* qlast = (const uint16_t*)io->qkv + (int64_t)(S - 1) * io->ldqkv;
        if ((rc = crab_copy_rows(ctx, stream, qlast, (int64_t)S * io->ldqkv, io->act, io->ldact, B, H * d))) return rc;
        if ((rc = crab_attn_decode_masked(ctx, stream, io->act, io->ldact, kc, vc, io->att, io->ldatt, B, H, Hk, d, io->Tmax, io->pos0 + S, nullptr,
                                          scale, io->kv_start)))
            return rc;
        // residual rows of the last tokens, gathered into qkv's storage (fp32: D floats = 2 D 16-bit words per row)
        const int wpr = io->x_fp32 ? 2 * D : D;
        const uint16_t* xlast = (const uint16_t*)io->x + (int64_t)(S - 1) * io->ldx * (io->x_fp32 ? 2 : 1);
        if ((rc = crab_copy_rows(ctx, stream, xlast, (int64_t)S * io->ldx * (io->x_fp32 ? 2 : 1), io->qkv, wpr, B, wpr))) return rc;
        crab_llama_io t = *io;                                          // the B-row view the group calls below see
        t.S = 1; t.x = io->qkv; t.ldx = D;
        const bool ahead = L->gu.RA != nullptr && B <= CRAB_DECODE_MAX_ROWS;
        GroupCall o{};
        o.x = io->att; o.ldx = io->ldatt; o.out = t.x; o.ldc = t.ldx; o.residual = t.x; o.ldr = t.ldx; o.act = CRAB_ACT_NONE;
        o.norm_w = L->post_attention_norm_w; o.norm_out = io->h; o.ld_norm = io->ldh; o.eps = L->rms_eps;
        if (ahead) { o.route_next = &L->gu; o.route_u = io->u2; }
        if ((rc = run_group(ctx, stream, &L->o, &t, L, B, o, nullptr, nullptr))) return rc;
        GroupCall g{};
        g.x = io->h; g.ldx = io->ldh; g.out = io->act; g.ldc = io->ldact; g.act = CRAB_ACT_SWIGLU_PAIR;
        g.u_ready = ahead ? io->u2 : nullptr;
        if ((rc = run_group(ctx, stream, &L->gu, &t, L, B, g, nullptr, nullptr))) return rc;
        GroupCall w{};
        w.x = io->act; w.ldx = io->ldact; w.out = t.x; w.ldc = t.ldx; w.residual = t.x; w.ldr = t.ldx; w.act = CRAB_ACT_NONE;
        w.norm_w = L->next_norm_w; w.norm_out = io->h; w.ld_norm = io->ldh; w.eps = L->rms_eps;
        if ((rc = run_group(ctx, stream, &L->down, &t, L, B, w, nullptr, nullptr))) return rc;
        io->u_qkv_ready = 0;
        return CRAB_OK;
    }
    if (prefill) {
        // fused_rope 1: q and k already rotated (k in the cache) by the projection's epilogue, only the v columns are left (cache append + V^T);
        // 2: those too
        if (fused_rope != 2 &&
            (rc = crab_qkv_rope_split_ids(ctx, stream, io->qkv, io->ldqkv, fused_rope ? nullptr : io->rope_tab, fused_rope ? nullptr : kc, vc, io->vt,
                                          io->vt_ld, B, S, H, Hk, d, io->Tmax, io->pos0, io->pos_dev, fused_rope ? nullptr : io->pos_ids, io->ld_pos)))
            return rc;
        crab_attn_desc a;
        memset(&a, 0, sizeof(a));
        a.q = io->qkv; a.k = kc; a.vt = io->vt; a.o = io->att;
        a.q_bs = (int64_t)S * io->ldqkv; a.q_hs = d; a.q_ss = io->ldqkv;
        a.k_bs = (int64_t)Hk * io->Tmax * d; a.k_hs = (int64_t)io->Tmax * d; a.k_ss = d;
        a.vt_bs = (int64_t)Hk * d * io->vt_ld; a.vt_hs = (int64_t)d * io->vt_ld; a.vt_ds = io->vt_ld;
        a.o_bs = (int64_t)S * io->ldatt; a.o_ss = io->ldatt;
        a.B = B; a.H = H; a.Hk = Hk; a.Sq = S; a.Skv = io->pos0 + S; a.head_dim = d; a.causal = 1; a.scale = scale;
        a.kv_start = io->kv_start;                                  // left-pad mask (NULL: every key visible)
        if ((rc = crab_attn_fwd(ctx, stream, &a))) return rc;
    } else if (fuse_attn) {
        if ((rc = crab_attn_decode_rope(ctx, stream, io->qkv, io->ldqkv, io->rope_tab, kc, vc, io->att, io->ldatt, B, H, Hk, d, io->Tmax, io->pos0,
                                        io->pos_dev, scale, io->attn_ws, io->attn_ws_bytes)))
            return rc;
    } else {
        if ((rc = crab_attn_decode_masked(ctx, stream, io->qkv, io->ldqkv, kc, vc, io->att, io->ldatt, B, H, Hk, d, io->Tmax, io->pos0 + 1,
                                          io->pos_dev, scale, io->row_off)))
            return rc;
    }
    // ---- o: x += o(att); h = rmsnorm(x) * post_attention_layernorm (+ the gate|up router ahead in the decode regime)
    const bool ahead_gu = L->gu.RA != nullptr && M <= CRAB_DECODE_MAX_ROWS;
    GroupCall o{};
    o.x = io->att; o.ldx = io->ldatt; o.out = io->x; o.ldc = io->ldx; o.residual = io->x; o.ldr = io->ldx; o.act = CRAB_ACT_NONE;
    o.norm_w = L->post_attention_norm_w; o.norm_out = io->h; o.ld_norm = io->ldh; o.eps = L->rms_eps;
    if (ahead_gu) { o.route_next = &L->gu; o.route_u = io->u2; }
    if ((rc = run_group(ctx, stream, &L->o, io, L, M, o, nullptr, nullptr))) return rc;
    // ---- gate|up with SwiGLU in the epilogue
    GroupCall g{};
    g.x = io->h; g.ldx = io->ldh; g.out = io->act; g.ldc = io->ldact; g.act = CRAB_ACT_SWIGLU_PAIR;
    g.u_ready = ahead_gu ? io->u2 : nullptr;
    if ((rc = run_group(ctx, stream, &L->gu, io, L, M, g, nullptr, nullptr))) return rc;
    // ---- down: x += down(act); h = rmsnorm(x) * next_norm_w (+ the next layer's q|k|v router ahead)
    const bool ahead_q = L->next_qkv != nullptr && L->next_qkv->RA != nullptr && M <= CRAB_DECODE_MAX_ROWS;
    GroupCall w{};
    w.x = io->act; w.ldx = io->ldact; w.out = io->x; w.ldc = io->ldx; w.residual = io->x; w.ldr = io->ldx; w.act = CRAB_ACT_NONE;
    w.norm_w = L->next_norm_w; w.norm_out = io->h; w.ld_norm = io->ldh; w.eps = L->rms_eps;
    if (ahead_q) { w.route_next = L->next_qkv; w.route_u = io->u2; }
    if ((rc = run_group(ctx, stream, &L->down, io, L, M, w, nullptr, nullptr))) return rc;
    io->u_qkv_ready = ahead_q ? 1 : 0;
    return CRAB_OK;
}

}  // namespace

extern "C" {

int crab_sizeof_llama_layer(void) { return (int)sizeof(crab_llama_layer); }
int crab_sizeof_llama_io(void) { return (int)sizeof(crab_llama_io); }

int crab_llama_layer_prefill(crab_ctx* ctx, void* stream, const crab_llama_layer* layer, crab_llama_io* io, int layer_index) {
    int rc = check_io(ctx, layer, io, true);
    if (rc) return rc;
    if (layer_index < 0) return crab_fail(ctx, CRAB_E_INVALID, "llama_layer_prefill: negative layer index");
    return run_layer(ctx, stream, layer, io, layer_index, true);
}

int crab_llama_layer_decode(crab_ctx* ctx, void* stream, const crab_llama_layer* layer, crab_llama_io* io, int layer_index) {
    int rc = check_io(ctx, layer, io, false);
    if (rc) return rc;
    if (layer_index < 0) return crab_fail(ctx, CRAB_E_INVALID, "llama_layer_decode: negative layer index");
    return run_layer(ctx, stream, layer, io, layer_index, false);
}

int crab_llama_layers(crab_ctx* ctx, void* stream, const crab_llama_layer* layers, int n_layers, crab_llama_io* io) {
    if (!ctx) return CRAB_E_INVALID;
    if (!layers || n_layers <= 0 || !io) return crab_fail(ctx, CRAB_E_INVALID, "llama_layers: null / empty layer table");
    const bool prefill = io->vt != nullptr;
    for (int l = 0; l < n_layers; ++l) {
        int rc = check_io(ctx, &layers[l], io, prefill);
        if (rc) return rc;
    }
    const bool last_rows = prefill && io->last_rows_only != 0;
    if (last_rows) {
        const crab_llama_layer* L = &layers[n_layers - 1];
        if (io->S <= 1 || io->ldact < L->H * L->d || (io->ldact & 7) || (int64_t)io->B * io->S * io->ldqkv * 2 < (int64_t)io->B * L->o.N * (io->x_fp32 ? 4 : 2))
            return crab_fail(ctx, CRAB_E_INVALID, "llama_layers: last_rows_only needs S > 1, ldact >= H d (multiple of 8) and qkv storage of at least B rows of the residual stream");
    }
    for (int l = 0; l < n_layers; ++l) {
        int rc = run_layer(ctx, stream, &layers[l], io, l, prefill, last_rows && l == n_layers - 1);
        if (rc) return rc;
    }
    return CRAB_OK;
}

}  // extern "C"

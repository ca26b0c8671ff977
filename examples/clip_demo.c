/*
 * clip_demo.c -- ONE WHOLE CLIP through libcrab_hip.so from C: CLIP ViT tower -> VLProjector (Q-Former) and BEATs -> ALProjector
 * (Q-Former), the splice of prepare_multimodal_inputs, decoder prefill and greedy decode, with nothing but the C ABI of
 * include/crab_hip.h and the HIP runtime (no Python, no torch in the process).  It is the host-language-neutral form of
 * UnifiedForCausalLM.generate (reference: models/unified_llama.py:244-267 -> models/unified_arch.py:217-406 ->
 * models/multimodal_encoder.py:52-84, 119-144, 168-186, 226-244) for one sample with one <video> and one <audio> block, and the body of
 * tests/test_c_abi_gpu.py::test_c_caller_runs_a_whole_clip, which compares its inputs_embeds, ids and logits bit for bit with the Python
 * modules (the same entry points in the same order: crab_clip_layer / crab_beats_layer / crab_qformer_layer / crab_llama_layers).
 *
 *   run:  clip_demo <blob.bin> <out.bin> <embeds_out.bin> use_graph
 *
 * blob.bin: int32 cfg[40] (see enum below); video fp32 [T_v,3,img,img]; audio fp32 [T_a,L_a,mel]; prompt ids int64 [n_text]; then bf16:
 *   CLIP: patch weight [Dc, pad32(3 P P)], class_embedding [Dc], position_embedding [1 + (img/P)^2, Dc], pre_layrnorm w, b; per layer
 *         (Lc = highest selected hidden state): ln1 w, b; q|k|v W [3Dc, Dc], b; out W, b; ln2 w, b; fc1 W [Ic, Dc], b; fc2 W [Dc, Ic], b
 *   VLProjector: visual_ln w, b [Dc]; Q-Former (enc width Dc); MLP W0 [D, hq], b0, W2 [D, D], b2
 *   BEATs: patch weight [emb, P_b^2]; layer_norm w, b [emb]; post_extract_proj W [E, emb], b; folded pos_conv weight [G][E/G][Kc E/G];
 *         pos_conv bias [E]; encoder.layer_norm w, b [E]; relative_attention_bias [buckets, Hb]; per layer: q|k|v W [3E, E], b; out W, b;
 *         fc1 W [F, E], b; fc2 W [E, F], b; self_attn_layer_norm w, b; final_layer_norm w, b; grep_linear W [8, E/Hb], b [8]; grep_a [Hb]
 *   ALProjector: audio_ln w, b [E]; Q-Former (enc width E); MLP;   then embed_tokens [V, D] (for the text rows of the splice)
 *   Q-Former = embeddings.LayerNorm w, b [hq]; query tokens [nq, hq]; per layer: query W, b; key|value W [2hq, hq], b; output.dense W, b;
 *         output.LayerNorm w, b; cross query W, b; cross key|value W [2hq, enc], b; cross output.dense W, b; cross LayerNorm w, b;
 *         intermediate_query W [iq, hq], b; output_query W [hq, iq], b; output_query LayerNorm w, b
 *   decoder: as examples/demo_common.h
 */
#include <math.h>

#include "demo_common.h"

enum { C_D, C_I, C_L, C_H, C_HK, C_V, C_NL, C_R, C_NNEW, C_QKVBIAS,                    /* decoder */
       C_DC, C_IC, C_LC, C_HC, C_IMG, C_PATCH,                                         /* CLIP */
       C_EMB, C_E, C_F, C_LB, C_HB, C_PB, C_KC, C_G, C_BUCKETS, C_MAXDIST, C_MEL,      /* BEATs */
       C_HQ, C_HHQ, C_IQ, C_LQ, C_NQ,                                                  /* Q-Formers */
       C_TV, C_TA, C_LA, C_NTEXT, C_VIDEO_ID, C_AUDIO_ID, C_N };

static void* next_raw(size_t bytes) {
    void* host = malloc(bytes);
    void* dev = NULL;
    if (fread(host, 1, bytes, blob) != bytes) { fprintf(stderr, "blob too short\n"); exit(4); }
    HIP_OK(hipMalloc(&dev, bytes));
    HIP_OK(hipMemcpy(dev, host, bytes, hipMemcpyHostToDevice));
    free(host);
    return dev;
}

static crab_dense next_dense(int N, int K) {
    crab_dense d;
    memset(&d, 0, sizeof(d));
    d.N = N; d.K = K; d.ldw = K;
    d.W = next_bf16((size_t)N * K);
    d.bias = next_bf16((size_t)N);
    return d;
}

/* LayerNorm parameters travel in fp32 (crab_ln.fp32 = 1, ABI 9), as crab_amd's encoder modules hold them: they are not matrix operands */
static crab_ln next_ln(int n, float eps) {
    crab_ln l;
    l.w = next_raw((size_t)n * 4); l.b = next_raw((size_t)n * 4); l.eps = eps; l.fp32 = 1;
    return l;
}

static void* gws;                       /* the crab_gemm_desc.workspace every GEMM with <= 256 rows gets (the rule of crab_amd/ops.py) */
static int64_t gws_bytes;

static void gemm(hipStream_t s, const void* A, int64_t lda, const crab_dense* w, void* C, int64_t ldc, int M, int act) {
    crab_gemm_desc d;
    memset(&d, 0, sizeof(d));
    d.A = A; d.lda = lda; d.B = w->W; d.ldb = w->ldw; d.C = C; d.ldc = ldc; d.bias = w->bias;
    d.M = M; d.N = w->N; d.K = w->K; d.act = act; d.res_scale = 1.0f; d.batch = 1; d.nb0 = 1;
    if (M <= 256) { d.workspace = gws; d.workspace_bytes = gws_bytes; }
    CRAB_OK_(crab_gemm_bf16(ctx, s, &d));
}

typedef struct { crab_ln emb_ln; void* query; crab_qformer_layer_w* layers; crab_dense mlp0, mlp2; crab_ln in_ln; } projector;

static projector load_projector(int enc_w, int hq, int Hhq, int iq, int Lq, int nq, int D) {
    projector p;
    p.in_ln = next_ln(enc_w, 1e-5f);
    p.emb_ln = next_ln(hq, 1e-12f);
    p.query = next_bf16((size_t)nq * hq);
    p.layers = (crab_qformer_layer_w*)calloc(Lq, sizeof(crab_qformer_layer_w));
    for (int l = 0; l < Lq; ++l) {
        crab_qformer_layer_w* w = &p.layers[l];
        w->sq = next_dense(hq, hq); w->skv = next_dense(2 * hq, hq); w->so = next_dense(hq, hq); w->sln = next_ln(hq, 1e-12f);
        w->cq = next_dense(hq, hq); w->ckv = next_dense(2 * hq, enc_w); w->co = next_dense(hq, hq); w->cln = next_ln(hq, 1e-12f);
        w->iq = next_dense(iq, hq); w->oq = next_dense(hq, iq); w->oln = next_ln(hq, 1e-12f);
        w->H = Hhq;
    }
    p.mlp0 = next_dense(D, hq); p.mlp2 = next_dense(D, D);
    return p;
}

/* VLProjector.forward / ALProjector.forward (multimodal_encoder.py:119-144, 226-244): LN -> Q-Former over B blocks of m encoder rows ->
 * Linear, GELU, Linear; returns [B * nq, D] */
static void* run_projector(hipStream_t s, const projector* p, const void* feat, int B, int m, int enc_w, int hq, int iq, int Lq, int nq, int D, int Hhq) {
    const int Menc = B * m, M = B * nq, rows = Menc > M ? Menc : M;
    void* enc = dev_alloc((size_t)Menc * enc_w * 2);
    CRAB_OK_(crab_layernorm_p(ctx, s, feat, 0, enc_w, p->in_ln.w, p->in_ln.b, p->in_ln.fp32, enc, enc_w, Menc, enc_w, p->in_ln.eps));
    void* z0 = dev_alloc((size_t)nq * hq * 2);
    CRAB_OK_(crab_layernorm_p(ctx, s, p->query, 0, hq, p->emb_ln.w, p->emb_ln.b, p->emb_ln.fp32, z0, hq, nq, hq, p->emb_ln.eps));
    void* z = z0;
    if (B > 1) {                                          /* the learned query tokens, broadcast to every block */
        z = dev_alloc((size_t)M * hq * 2);
        CRAB_OK_(crab_copy_rows_batched(ctx, s, z0, hq, 0, z, hq, (int64_t)nq * hq, B, nq, hq));
    }
    crab_enc_io io;
    memset(&io, 0, sizeof(io));
    const int keys = nq > m ? nq : m;
    /* x_fp32: the pre-LN sums (io.y) of the post-LN Q-Former are fp32, as crab_amd/multimodal_encoder.py keeps them */
    io.x = z; io.a = dev_alloc((size_t)M * hq * 2); io.y = dev_alloc((size_t)M * hq * 4); io.att = dev_alloc((size_t)M * hq * 2); io.x_fp32 = 1;
    io.f = dev_alloc((size_t)M * iq * 2); io.qkv = dev_alloc((size_t)rows * 2 * hq * 2);
    io.vt_bytes = (int64_t)B * hq * pad_to(keys, 8) * 2; io.vt = dev_alloc((size_t)io.vt_bytes);
    io.enc = enc; io.enc_rows = m; io.workspace = gws; io.workspace_bytes = gws_bytes; io.B = B; io.S = nq;
    (void)Hhq;
    for (int l = 0; l < Lq; ++l) CRAB_OK_(crab_qformer_layer(ctx, s, &p->layers[l], &io));
    void* y0 = dev_alloc((size_t)M * D * 2);
    void* y1 = dev_alloc((size_t)M * D * 2);
    gemm(s, z, hq, &p->mlp0, y0, D, M, CRAB_ACT_GELU);
    gemm(s, y0, D, &p->mlp2, y1, D, M, CRAB_ACT_NONE);
    return y1;
}

int main(int argc, char** argv) {
    if (argc != 5) { fprintf(stderr, "usage: %s blob out embeds_out use_graph\n", argv[0]); return 1; }
    const int use_graph = atoi(argv[4]);
    blob = fopen(argv[1], "rb");
    if (!blob) { perror(argv[1]); return 1; }
    int32_t c[C_N];
    if (fread(c, 4, C_N, blob) != C_N) { fprintf(stderr, "blob too short\n"); return 4; }
    HIP_OK(hipSetDevice(0));
    if (crab_ctx_create(0, &ctx) != 0) { fprintf(stderr, "crab_ctx_create failed\n"); return 1; }
    hipStream_t s;
    HIP_OK(hipStreamCreate(&s));
    gws_bytes = 64 << 20;
    gws = dev_alloc((size_t)gws_bytes);
    const int D = c[C_D], Tv = c[C_TV], Ta = c[C_TA], La = c[C_LA], mel = c[C_MEL], ntext = c[C_NTEXT], nq = c[C_NQ];

    /* ---- inputs */
    const int img = c[C_IMG], P = c[C_PATCH], gp = img / P, Pn = gp * gp, T = Pn + 1;
    float* video = (float*)next_raw((size_t)Tv * 3 * img * img * 4);
    float* audio = (float*)next_raw((size_t)Ta * La * mel * 4);
    int64_t* ids_h = (int64_t*)malloc((size_t)ntext * 8);
    if (fread(ids_h, 8, ntext, blob) != (size_t)ntext) { fprintf(stderr, "blob too short\n"); return 4; }

    /* ================= CLIP ViT tower (multimodal_encoder.py:52-84 over HF CLIPVisionModel) ================= */
    const int Dc = c[C_DC], Ic = c[C_IC], Lc = c[C_LC], Hc = c[C_HC], Kp = pad_to(3 * P * P, 32);
    crab_dense patch_w;
    memset(&patch_w, 0, sizeof(patch_w));
    patch_w.N = Dc; patch_w.K = Kp; patch_w.ldw = Kp; patch_w.W = next_bf16((size_t)Dc * Kp);
    void* cls = next_bf16(Dc);
    void* pos = next_bf16((size_t)T * Dc);
    crab_ln pre_ln = next_ln(Dc, 1e-5f);
    crab_clip_layer_w* cl = (crab_clip_layer_w*)calloc(Lc, sizeof(crab_clip_layer_w));
    for (int l = 0; l < Lc; ++l) {
        cl[l].ln1 = next_ln(Dc, 1e-5f);
        cl[l].qkv = next_dense(3 * Dc, Dc); cl[l].out = next_dense(Dc, Dc);
        cl[l].ln2 = next_ln(Dc, 1e-5f);
        cl[l].fc1 = next_dense(Ic, Dc); cl[l].fc2 = next_dense(Dc, Ic);
        cl[l].H = Hc;
    }
    const int Mv = Tv * T;
    void* vbf = dev_alloc((size_t)Tv * 3 * img * img * 2);
    CRAB_OK_(crab_cast_f32_bf16(ctx, s, video, vbf, (int64_t)Tv * 3 * img * img));                   /* the bf16 model takes bf16 pixels */
    void* patches = dev_alloc((size_t)Tv * Pn * Kp * 2);
    CRAB_OK_(crab_im2col_patch(ctx, s, vbf, 0, patches, Kp, Tv, 3, img, img, P));
    void* pe = dev_alloc((size_t)Tv * Pn * Dc * 2);
    gemm(s, patches, Kp, &patch_w, pe, Dc, Tv * Pn, CRAB_ACT_NONE);                                  /* conv14/14, no bias */
    void* hc = dev_alloc((size_t)Mv * Dc * 2);
    CRAB_OK_(crab_clip_embed_ln_p(ctx, s, pe, cls, pos, pre_ln.w, pre_ln.b, pre_ln.fp32, hc, Tv, Pn, Dc, pre_ln.eps));
    {
        crab_enc_io io;
        memset(&io, 0, sizeof(io));
        /* the residual stream of the pre-LN tower is fp32 (crab_enc_io.x_fp32): x and the mid-layer row y; the kept state goes back to bf16 */
        float* hx = (float*)dev_alloc((size_t)Mv * Dc * 4);
        CRAB_OK_(crab_cast_rows_bf16_f32(ctx, s, hc, Dc, hx, Dc, Mv, Dc));
        io.x = hx; io.a = dev_alloc((size_t)Mv * Dc * 2); io.y = dev_alloc((size_t)Mv * Dc * 4); io.att = dev_alloc((size_t)Mv * Dc * 2); io.x_fp32 = 1;
        io.qkv = dev_alloc((size_t)Mv * 3 * Dc * 2); io.f = dev_alloc((size_t)Mv * Ic * 2);
        io.vt_bytes = (int64_t)Tv * Dc * pad_to(T, 8) * 2; io.vt = dev_alloc((size_t)io.vt_bytes);
        io.workspace = gws; io.workspace_bytes = gws_bytes; io.B = Tv; io.S = T;
        for (int l = 0; l < Lc; ++l) CRAB_OK_(crab_clip_layer(ctx, s, &cl[l], &io));
        if (Lc > 0) CRAB_OK_(crab_cast_rows_f32_bf16(ctx, s, hx, Dc, hc, Dc, Mv, Dc));
    }
    void* vfeat = dev_alloc((size_t)Tv * Pn * Dc * 2);                                               /* drop CLS (select_feature 'patch') */
    CRAB_OK_(crab_copy_rows_batched(ctx, s, (const uint16_t*)hc + Dc, Dc, (int64_t)T * Dc, vfeat, Dc, (int64_t)Pn * Dc, Tv, Pn, Dc));
    /* ================= VLProjector ================= */
    const int hq = c[C_HQ], Hhq = c[C_HHQ], iq = c[C_IQ], Lq = c[C_LQ];
    projector vl = load_projector(Dc, hq, Hhq, iq, Lq, nq, D);
    void* vtok = run_projector(s, &vl, vfeat, Tv, Pn, Dc, hq, iq, Lq, nq, D, Hhq);                   /* [Tv * nq, D] */

    /* ================= BEATs (models/beats/BEATs.py:134-182, backbone.py) ================= */
    const int emb = c[C_EMB], E = c[C_E], F = c[C_F], Lb = c[C_LB], Hb = c[C_HB], Pb = c[C_PB], Kc = c[C_KC], G = c[C_G];
    const int n = (La / Pb) * (mel / Pb), Ma = Ta * n, cg = E / G, npad = n + Kc - 1, db = E / Hb;
    crab_dense bpatch;
    memset(&bpatch, 0, sizeof(bpatch));
    bpatch.N = emb; bpatch.K = Pb * Pb; bpatch.ldw = Pb * Pb; bpatch.W = next_bf16((size_t)emb * Pb * Pb);
    crab_ln b_ln = next_ln(emb, 1e-5f);
    crab_dense post = next_dense(E, emb);
    void* pcw = next_bf16((size_t)G * cg * Kc * cg);
    void* pcb = next_bf16(E);
    crab_ln enc_ln = next_ln(E, 1e-5f);
    void* table = next_bf16((size_t)c[C_BUCKETS] * Hb);
    crab_beats_layer_w* bl = (crab_beats_layer_w*)calloc(Lb, sizeof(crab_beats_layer_w));
    const float alpha = (float)pow(2.0 * Lb, 0.25);
    for (int l = 0; l < Lb; ++l) {
        bl[l].qkv = next_dense(3 * E, E); bl[l].out = next_dense(E, E); bl[l].fc1 = next_dense(F, E); bl[l].fc2 = next_dense(E, F);
        bl[l].ln_attn = next_ln(E, 1e-5f); bl[l].ln_final = next_ln(E, 1e-5f);
        bl[l].grep_w = next_bf16((size_t)8 * db); bl[l].grep_b = next_bf16(8); bl[l].grep_a = next_bf16(Hb);
        bl[l].H = Hb; bl[l].alpha = alpha;
    }
    void* abf = dev_alloc((size_t)Ta * La * mel * 2);
    CRAB_OK_(crab_cast_f32_bf16(ctx, s, audio, abf, (int64_t)Ta * La * mel));
    void* apatch = dev_alloc((size_t)Ma * Pb * Pb * 2);
    CRAB_OK_(crab_im2col_patch(ctx, s, abf, 0, apatch, Pb * Pb, Ta, 1, La, mel, Pb));
    void* af = dev_alloc((size_t)Ma * emb * 2);
    gemm(s, apatch, Pb * Pb, &bpatch, af, emb, Ma, CRAB_ACT_NONE);
    void* ax0 = dev_alloc((size_t)Ma * emb * 2);
    CRAB_OK_(crab_layernorm_p(ctx, s, af, 0, emb, b_ln.w, b_ln.b, b_ln.fp32, ax0, emb, Ma, emb, b_ln.eps));
    void* ax = dev_alloc((size_t)Ma * E * 2);
    gemm(s, ax0, emb, &post, ax, E, Ma, CRAB_ACT_NONE);
    void* xp = dev_alloc((size_t)G * Ta * npad * cg * 2);                                            /* x + gelu(pos_conv(x)): sliding-window GEMM */
    CRAB_OK_(crab_beats_posconv_pad(ctx, s, ax, xp, Ta, n, E, G, Kc));
    void* ay = dev_alloc((size_t)Ma * E * 4);                                                        /* the pre-LN sum, fp32 */
    {
        crab_gemm_desc g;
        memset(&g, 0, sizeof(g));
        g.A = xp; g.B = pcw; g.C = ay; g.bias = pcb; g.R = ax; g.c_fp32 = 1;
        g.lda = cg; g.ldb = (int64_t)Kc * cg; g.ldc = E; g.ldr = E;
        g.M = n; g.N = cg; g.K = Kc * cg; g.act = CRAB_ACT_GELU; g.res_scale = 1.0f;
        g.batch = G * Ta; g.nb0 = Ta;
        g.sA0 = (int64_t)npad * cg; g.sA1 = (int64_t)Ta * npad * cg; g.sB0 = 0; g.sB1 = (int64_t)cg * Kc * cg;
        g.sC0 = (int64_t)n * E; g.sC1 = cg; g.sR0 = (int64_t)n * E; g.sR1 = cg; g.sBias0 = 0; g.sBias1 = cg;
        CRAB_OK_(crab_gemm_bf16(ctx, s, &g));
    }
    void* bx = dev_alloc((size_t)Ma * E * 2);
    CRAB_OK_(crab_layernorm_p(ctx, s, ay, 1, E, enc_ln.w, enc_ln.b, enc_ln.fp32, bx, E, Ma, E, enc_ln.eps));
    float* relb = (float*)dev_alloc((size_t)Hb * n * n * 4);
    CRAB_OK_(crab_beats_relpos_bias(ctx, s, table, relb, n, Hb, c[C_BUCKETS], c[C_MAXDIST]));
    {
        crab_enc_io io;
        memset(&io, 0, sizeof(io));
        io.x = bx; io.a = dev_alloc((size_t)Ma * E * 2); io.y = dev_alloc((size_t)Ma * E * 4); io.att = dev_alloc((size_t)Ma * E * 2); io.x_fp32 = 1;
        io.qkv = dev_alloc((size_t)Ma * 3 * E * 2); io.f = dev_alloc((size_t)Ma * F * 2);
        io.vt_bytes = (int64_t)Ta * E * pad_to(n, 8) * 2; io.vt = dev_alloc((size_t)io.vt_bytes);
        io.bias = relb; io.gate = (float*)dev_alloc((size_t)Ta * Hb * n * 4);
        io.workspace = gws; io.workspace_bytes = gws_bytes; io.B = Ta; io.S = n;
        for (int l = 0; l < Lb; ++l) CRAB_OK_(crab_beats_layer(ctx, s, &bl[l], &io));
    }
    /* ================= ALProjector ================= */
    projector al = load_projector(E, hq, Hhq, iq, Lq, nq, D);
    void* atok = run_projector(s, &al, bx, Ta, n, E, hq, iq, Lq, nq, D, Hhq);                        /* [Ta * nq, D] */

    /* ================= splice (prepare_multimodal_inputs, unified_arch.py:273-319): text spans embedded, <video> / <audio> replaced ======= */
    const int S = ntext - 2 + nq * Tv + nq * Ta;
    int64_t* tok = (int64_t*)malloc((size_t)S * 8);
    void* embeds = dev_alloc((size_t)S * D * 2);
    int cur = 0;
    for (int i = 0; i < ntext; ++i) {
        if (ids_h[i] == c[C_VIDEO_ID] || ids_h[i] == c[C_AUDIO_ID]) {
            const int isv = ids_h[i] == c[C_VIDEO_ID], rows = nq * (isv ? Tv : Ta);
            CRAB_OK_(crab_copy_rows(ctx, s, isv ? vtok : atok, D, (uint16_t*)embeds + (size_t)cur * D, D, rows, D));
            for (int j = 0; j < rows; ++j) tok[cur + j] = -1;                                        /* rows the embedding lookup leaves alone */
            cur += rows;
        } else {
            tok[cur++] = ids_h[i];
        }
    }
    if (cur != S) { fprintf(stderr, "splice length %d != %d\n", cur, S); return 5; }
    int64_t* tok_d = (int64_t*)dev_alloc((size_t)S * 8);
    HIP_OK(hipMemcpyAsync(tok_d, tok, (size_t)S * 8, hipMemcpyHostToDevice, s));
    const int I = c[C_I], L = c[C_L], H = c[C_H], Hk = c[C_HK], V = c[C_V], nl = c[C_NL], r = c[C_R];
    void* embed_tokens = next_bf16((size_t)V * D);                                                   /* a copy for the text rows of the splice */
    CRAB_OK_(crab_embedding(ctx, s, tok_d, embed_tokens, embeds, D, S, D, V));
    HIP_OK(hipStreamSynchronize(s));
    {
        uint16_t* eh = (uint16_t*)malloc((size_t)S * D * 2);
        HIP_OK(hipMemcpy(eh, embeds, (size_t)S * D * 2, hipMemcpyDeviceToHost));
        FILE* eo = fopen(argv[3], "wb");
        if (!eo) { perror(argv[3]); return 1; }
        fwrite(eh, 2, (size_t)S * D, eo);
        fclose(eo);
        free(eh);
    }
    printf("clip_demo: encoders + splice ok, S = %d\n", S);
    /* ================= decoder: prefill + greedy decode ================= */
    return run_decoder(embeds, argv[2], D, I, L, H, Hk, V, nl, r, 1, S, c[C_NNEW], c[C_QKVBIAS], use_graph);
}

/*
 * demo_common.h -- shared by examples/decode_demo.c and examples/clip_demo.c: blob reader, device allocation, and run_decoder(): prefill +
 * greedy decode of a hyper-LoRA Llama / Qwen2 decoder stack through the C ABI of include/crab_hip.h only (the host-language-neutral
 * form of crab_amd/decoder.py's GenerationEngine; reference: models/unified_llama.py:244-267 generate -> HF greedy search over
 * models/modeling_llama.py:805-827 layers).
 *
 * Decoder part of the blob (all bf16, in this order): per layer: for each group (q|k|v, o, gate|up, down): W [N,K], bias [N] (q|k|v only,
 * when qkv_bias), RA [tcols,K], B2 [N,ucols]; input_layernorm [D]; post_attention_layernorm [D]; then model.norm [D]; lm_head [V,D];
 * embed_tokens [V,D].   out: ids int64 [B, n_new], then the fp32 logits of every step [n_new][B,V].
 */
#ifndef CRAB_DEMO_COMMON_H
#define CRAB_DEMO_COMMON_H
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "crab_hip.h"

#define HIP_OK(e) do { hipError_t e_ = (e); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(e_)); exit(2); } } while (0)
#define CRAB_OK_(e) do { int r_ = (e); if (r_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #e, r_, crab_last_error(ctx)); exit(3); } } while (0)

static crab_ctx* ctx;
static FILE* blob;

static int pad_to(int n, int m) { return (n + m - 1) / m * m; }

/* next `elems` bf16 elements of the blob -> fresh device buffer */
static void* next_bf16(size_t elems) {
    void* host = malloc(elems * 2);
    void* dev = NULL;
    if (fread(host, 2, elems, blob) != elems) { fprintf(stderr, "blob too short\n"); exit(4); }
    HIP_OK(hipMalloc(&dev, elems * 2));
    HIP_OK(hipMemcpy(dev, host, elems * 2, hipMemcpyHostToDevice));
    free(host);
    return dev;
}

/* next `elems` fp32 elements (norm parameters) */
static void* next_f32(size_t elems) { return next_bf16(elems * 2); }

static void* dev_alloc(size_t bytes) {
    void* p = NULL;
    HIP_OK(hipMalloc(&p, bytes));
    HIP_OK(hipMemset(p, 0, bytes));
    return p;
}

static void load_group(crab_linear_group* g, int N, int K, int nproj, int nl, int r, int bias) {
    memset(g, 0, sizeof(*g));
    g->N = N; g->K = K; g->ldw = K;
    g->W = next_bf16((size_t)N * K);
    if (bias) g->bias = next_bf16((size_t)N);
    g->nproj = nproj; g->nl = nl; g->r = r;
    g->tcols = pad_to(nproj * (nl + r), 16);
    g->ucols = pad_to(nproj * nl * r, 32);
    g->RA = next_bf16((size_t)g->tcols * K); g->ldra = K;
    g->B2 = next_bf16((size_t)N * g->ucols); g->ldb2 = g->ucols;
    g->scaling = 16.0f / (float)r;                    /* lora_alpha = 16 (peft_hyper LoraConfig default of the reference scripts) */
}


/* prefill of `embeds` [B, S, D] (device, bf16) + n_new greedy tokens; reads the decoder weights from the blob (see the header comment) */
static int run_decoder(void* embeds, const char* outpath, int D, int I, int L, int H, int Hk, int V, int nl, int r, int B, int S, int n_new,
                       int qkv_bias, int use_graph) {
    const int d = D / H, Nq = (H + 2 * Hk) * d;
    const int Tmax = pad_to(S + n_new, 64), Sp = pad_to(S, 8), Mp = B * S;
    const float eps = 1e-5f;
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));

    /* ---- weights (borrowed by the layer table for the lifetime of the run) */
    crab_llama_layer* layers = (crab_llama_layer*)calloc(L, sizeof(crab_llama_layer));
    void** ln_in = (void**)calloc(L, sizeof(void*));
    for (int l = 0; l < L; ++l) {
        load_group(&layers[l].qkv, Nq, D, 3, nl, r, qkv_bias);
        load_group(&layers[l].o, D, H * d, 1, nl, r, 0);
        load_group(&layers[l].gu, 2 * I, D, 2, nl, r, 0);          /* rows interleaved (gate_i, up_i) as crab_amd packs them */
        load_group(&layers[l].down, D, I, 1, nl, r, 0);
        ln_in[l] = next_f32(D);                                /* RMSNorm weights travel in fp32 (crab_llama_layer.norm_w_fp32, ABI 9) */
        layers[l].post_attention_norm_w = next_f32(D);
        layers[l].norm_w_fp32 = 1;
        layers[l].H = H; layers[l].Hk = Hk; layers[l].d = d; layers[l].rms_eps = eps;
    }
    void* final_norm = next_f32(D);
    void* lm_head = next_bf16((size_t)V * D);
    void* embed_tokens = next_bf16((size_t)V * D);
    for (int l = 0; l < L; ++l) {
        layers[l].next_norm_w = l + 1 < L ? ln_in[l + 1] : final_norm;
        layers[l].next_qkv = l + 1 < L ? &layers[l + 1].qkv : NULL;
    }
    fclose(blob);

    /* ---- caller-owned activations, caches and workspaces */
    const int ucols_max = pad_to(3 * nl * r, 32), tcols_max = pad_to(3 * (nl + r), 16);
    const int Kmax = D > I ? D : I;
    crab_llama_io io;
    memset(&io, 0, sizeof(io));
    io.x = dev_alloc((size_t)Mp * D * 4); io.ldx = D; io.x_fp32 = 1;      /* the residual stream is fp32 (crab_llama_io.x_fp32), like crab_amd/decoder.py keeps it */
    io.h = dev_alloc((size_t)Mp * D * 2); io.ldh = D;
    io.qkv = dev_alloc((size_t)Mp * Nq * 2); io.ldqkv = Nq;
    io.att = dev_alloc((size_t)Mp * H * d * 2); io.ldatt = H * d;
    io.act = dev_alloc((size_t)Mp * I * 2); io.ldact = I;
    io.u = dev_alloc((size_t)Mp * ucols_max * 2); io.u2 = dev_alloc((size_t)Mp * ucols_max * 2); io.ldu = ucols_max;
    io.route_ws_bytes = crab_hyperlora_route_workspace(Mp, Kmax, tcols_max);
    io.route_ws = dev_alloc((size_t)io.route_ws_bytes);
    io.splitk_ws_bytes = 64 << 20;
    io.splitk_ws = dev_alloc((size_t)io.splitk_ws_bytes);
    float* rope = (float*)dev_alloc((size_t)Tmax * d * sizeof(float));
    CRAB_OK_(crab_rope_table(ctx, stream, rope, Tmax, d, 10000.0f));
    io.rope_tab = rope;
    const size_t cache_layer = (size_t)B * Hk * Tmax * d;
    io.k_cache = dev_alloc(cache_layer * L * 2); io.v_cache = dev_alloc(cache_layer * L * 2); io.cache_layer_stride = (int64_t)cache_layer;
    void* vt = dev_alloc((size_t)B * Hk * d * Sp * 2);
    io.Tmax = Tmax; io.pos0 = 0;

    void* hn = dev_alloc((size_t)B * D * 2);
    float* logits = (float*)dev_alloc((size_t)B * V * sizeof(float));
    int64_t* cur_ids = (int64_t*)dev_alloc((size_t)B * 8);
    int64_t* out_ids = (int64_t*)dev_alloc((size_t)B * n_new * 8);
    int32_t* finished = (int32_t*)dev_alloc((size_t)B * 4);
    int32_t* pos_dev = (int32_t*)dev_alloc(4);
    int32_t* step_dev = (int32_t*)dev_alloc(4);
    float* all_logits = (float*)malloc((size_t)n_new * B * V * sizeof(float));

    crab_gemm_desc head;                                   /* lm_head on the B last rows, fp32 logits */
    memset(&head, 0, sizeof(head));
    head.A = hn; head.lda = D; head.B = lm_head; head.ldb = D; head.C = logits; head.ldc = V; head.M = B; head.N = V; head.K = D;
    head.c_fp32 = 1; head.res_scale = 1.0f; head.batch = 1; head.nb0 = 1;
    head.workspace = io.splitk_ws; head.workspace_bytes = io.splitk_ws_bytes;

    /* ---- prefill: x = embeds; h = rmsnorm(x) * layer 0 input_layernorm; all layers; logits of the last row of each sequence */
    CRAB_OK_(crab_cast_rows_bf16_f32(ctx, stream, embeds, D, (float*)io.x, D, Mp, D));
    CRAB_OK_(crab_rmsnorm_p(ctx, stream, io.x, 1, D, ln_in[0], 1, io.h, D, Mp, D, eps));
    io.B = B; io.S = S; io.vt = vt; io.vt_ld = Sp; io.pos_dev = NULL; io.u_qkv_ready = 0;
    /* lm_head needs one row per sequence: the last layer runs its attention / o_proj / MLP for the B last rows only and leaves
     * rmsnorm(x_last) * model.norm in h[0:B] (crab_llama_io.last_rows_only, ABI 9; what crab_amd/decoder.py::prefill does for generate()) */
    io.last_rows_only = S > 1 ? 1 : 0;
    CRAB_OK_(crab_llama_layers(ctx, stream, layers, L, &io));
    if (io.last_rows_only) CRAB_OK_(crab_copy_rows(ctx, stream, io.h, D, hn, D, B, D));
    else CRAB_OK_(crab_copy_rows(ctx, stream, (const uint16_t*)io.h + (size_t)(S - 1) * D, (int64_t)S * D, hn, D, B, D));
    io.last_rows_only = 0;
    CRAB_OK_(crab_gemm_bf16(ctx, stream, &head));
    {
        int32_t p0 = S - 1;
        HIP_OK(hipMemcpyAsync(pos_dev, &p0, 4, hipMemcpyHostToDevice, stream));
    }
    CRAB_OK_(crab_greedy_select(ctx, stream, logits, V, B, V, cur_ids, out_ids, n_new, step_dev, finished, -1, 0, 0));
    CRAB_OK_(crab_advance(ctx, stream, pos_dev, step_dev));
    HIP_OK(hipMemcpyAsync(all_logits, logits, (size_t)B * V * 4, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));

    /* ---- decode: one row per sequence; position and step live in device memory, so the step can be captured once and replayed */
    io.S = 1; io.vt = NULL; io.vt_ld = 0; io.pos_dev = pos_dev;
    head.A = io.h;                                         /* h after the stack = model.norm(x): exactly the B rows lm_head needs */
    if ((long)B * H < CRAB_ATTN_SPLIT_BELOW) {
        /* a small batch cannot fill the chip with one attention block per head: hand crab_llama_layers the scratch of the fused
         * RoPE + KV append + split-context attention (crab_attn_decode_rope); its tickets are zeroed ONCE, the kernel leaves them zero */
        io.attn_ws_bytes = crab_attn_decode_rope_workspace(B, H, d);
        io.attn_ws = dev_alloc((size_t)io.attn_ws_bytes);
        HIP_OK(hipMemsetAsync(io.attn_ws, 0, (size_t)io.attn_ws_bytes, stream));
    }
    hipGraph_t graph = NULL;
    hipGraphExec_t exec = NULL;
    for (int step = 1; step < n_new; ++step) {
        /* step 1 runs eagerly (it also loads the decode-only kernels), step 2 is captured, instantiated and launched, later steps replay */
        const int capture = use_graph && step == 2;
        if (!exec) {
            if (capture) HIP_OK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
            CRAB_OK_(crab_embedding_f32(ctx, stream, cur_ids, embed_tokens, (float*)io.x, D, B, D, V));
            CRAB_OK_(crab_rmsnorm_p(ctx, stream, io.x, 1, D, ln_in[0], 1, io.h, D, B, D, eps));
            io.u_qkv_ready = 0;
            CRAB_OK_(crab_llama_layers(ctx, stream, layers, L, &io));
            CRAB_OK_(crab_gemm_bf16(ctx, stream, &head));
            CRAB_OK_(crab_greedy_select(ctx, stream, logits, V, B, V, cur_ids, out_ids, n_new, step_dev, finished, -1, 0, 0));
            CRAB_OK_(crab_advance(ctx, stream, pos_dev, step_dev));
            if (capture) {
                HIP_OK(hipStreamEndCapture(stream, &graph));
                HIP_OK(hipGraphInstantiate(&exec, graph, NULL, NULL, 0));
                HIP_OK(hipGraphLaunch(exec, stream));
            }
        } else {
            HIP_OK(hipGraphLaunch(exec, stream));
        }
        HIP_OK(hipMemcpyAsync(all_logits + (size_t)step * B * V, logits, (size_t)B * V * 4, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
    }

    int64_t* ids = (int64_t*)malloc((size_t)B * n_new * 8);
    HIP_OK(hipMemcpy(ids, out_ids, (size_t)B * n_new * 8, hipMemcpyDeviceToHost));
    FILE* out = fopen(outpath, "wb");
    if (!out) { perror(outpath); return 1; }
    fwrite(ids, 8, (size_t)B * n_new, out);
    fwrite(all_logits, 4, (size_t)n_new * B * V, out);
    fclose(out);
    printf("decode_demo ok: B %d S %d new %d graph %d first ids", B, S, n_new, use_graph);
    for (int s = 0; s < n_new && s < 8; ++s) printf(" %lld", (long long)ids[s]);
    printf("\n");
    if (exec) { HIP_OK(hipGraphExecDestroy(exec)); HIP_OK(hipGraphDestroy(graph)); }
    crab_ctx_destroy(ctx);
    return 0;
}
#endif

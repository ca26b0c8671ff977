/*
 * decode_demo.c -- a C caller of libcrab_hip.so: prefill + greedy decode of a hyper-LoRA Llama / Qwen2 decoder stack with nothing
 * but the C ABI of include/crab_hip.h and the HIP runtime (no Python, no torch).  It is the host-language-neutral form of
 * crab_amd/decoder.py's GenerationEngine (reference: models/unified_llama.py:244-267 generate -> HF greedy search over
 * models/modeling_llama.py:805-827 layers), and the body of tests/test_c_abi_gpu.py, which feeds it the packed weights of a tiny
 * model and compares its ids / logits bit for bit with the Python engine.
 *
 *   build:  gcc -O1 -std=c99 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include examples/decode_demo.c -o decode_demo \
 *               -Lcrab_amd -lcrab_hip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/crab_amd -Wl,-rpath,/opt/rocm/lib
 *   run:    decode_demo <blob.bin> <out.bin> D I L H Hk V nl r B S n_new qkv_bias use_graph
 *
 * blob.bin (all bf16 unless noted, in this order): embeds [B,S,D]; per layer: for each group (q|k|v, o, gate|up, down): W [N,K],
 * bias [N] (q|k|v only, when qkv_bias), RA [tcols,K], B2 [N,ucols]; input_layernorm [D]; post_attention_layernorm [D];
 * then model.norm [D]; lm_head [V,D]; embed_tokens [V,D].
 * out.bin: ids int64 [B, n_new], then the fp32 logits of every step [n_new][B,V].
 */
#include "demo_common.h"

int main(int argc, char** argv) {
    if (argc != 16) { fprintf(stderr, "usage: %s blob out D I L H Hk V nl r B S n_new qkv_bias use_graph\n", argv[0]); return 1; }
    const int D = atoi(argv[3]), I = atoi(argv[4]), L = atoi(argv[5]), H = atoi(argv[6]), Hk = atoi(argv[7]), V = atoi(argv[8]);
    const int nl = atoi(argv[9]), r = atoi(argv[10]), B = atoi(argv[11]), S = atoi(argv[12]), n_new = atoi(argv[13]);
    const int qkv_bias = atoi(argv[14]), use_graph = atoi(argv[15]);
    blob = fopen(argv[1], "rb");
    if (!blob) { perror(argv[1]); return 1; }
    HIP_OK(hipSetDevice(0));
    if (crab_ctx_create(0, &ctx) != 0) { fprintf(stderr, "crab_ctx_create failed\n"); return 1; }
    void* embeds = next_bf16((size_t)B * S * D);
    return run_decoder(embeds, argv[2], D, I, L, H, Hk, V, nl, r, B, S, n_new, qkv_bias, use_graph);
}

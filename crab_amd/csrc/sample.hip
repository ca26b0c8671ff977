// Sampling step of the decode loop (the reference never passes do_sample, so HF applies the checkpoint's generation_config: Llama-2-chat
// ships do_sample = true, temperature 0.6, top_p 0.9, and GenerationConfig's default top_k = 50; scripts/quick_start.py:36-43,
// SURVEY.md appendix A.7).  HF order (transformers 4.37.2 generation/utils.py _get_logits_warper, logits_process.py): temperature ->
// top-k -> top-p -> softmax -> multinomial.  One block per row, no sort:
//   * top-k : the k-th largest scaled logit by a 32-step bisection on the order-preserving integer image of the floats (exact);
//   * top-p : TopPLogitsWarper removes, in ascending order, the tokens whose cumulative probability stays <= 1 - top_p, i.e. it keeps
//             token i iff the mass of the strictly larger tokens is < top_p: the kept set is {x >= t} for the largest t whose mass is
//             >= top_p, found by the same bisection on masses;
//   * draw  : u ~ U[0,1) from a counter-based generator keyed by (seed, step, row); threads own contiguous index ranges, a block scan
//             finds the range, its owner walks it.  Deterministic for a given seed; torch.multinomial's stream cannot be reproduced,
//             parity is distributional (tests/test_ops_gpu.py).
// EOS / min_new_tokens / finished-row bookkeeping as greedy_select_kernel (ops.hip).
#include "common.h"
#include "crab_internal.h"
#include <math.h>

namespace {

__device__ __forceinline__ uint32_t fkey(float f) {            // order-preserving float -> uint (NaN-free input)
    const uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

template <typename T>
__device__ __forceinline__ T block_sum(T v, T* sh) {            // 1024 threads, fixed tree: deterministic
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    T t = sh[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) t += sh[w];
    return t;
}

__global__ __launch_bounds__(1024) void sample_select_kernel(const float* __restrict__ logits, long ldl, int V, int64_t* __restrict__ cur_ids,
                                                             int64_t* __restrict__ out_ids, long ld_out, const int* __restrict__ step_dev,
                                                             int* __restrict__ finished, int eos_id, int pad_id, int min_new, float inv_t,
                                                             int top_k, float top_p, unsigned long long seed) {
    __shared__ float shf[16];
    __shared__ int shi[16];
    __shared__ float pre[1024];
    __shared__ int s_tok;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int step = step_dev[0];
    const int suppress = (eos_id >= 0 && step < min_new) ? eos_id : -1;
    const float* row = logits + (long)b * ldl;
    const int chunk = (V + 1023) / 1024, i0 = tid * chunk, i1 = min(V, i0 + chunk);
#define XVAL(i_) ((i_) == suppress ? -INFINITY : row[i_] * inv_t)
    float mx = -INFINITY;
    for (int i = i0; i < i1; ++i) mx = fmaxf(mx, XVAL(i));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((tid & 63) == 0) shf[tid >> 6] = mx;
    __syncthreads();
    mx = shf[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) mx = fmaxf(mx, shf[w]);
    // ---- top-k: tk = key of the k-th largest value
    uint32_t tk = 0u;
    if (top_k > 0 && top_k < V) {
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = tk | (1u << bit);
            int c = 0;
            for (int i = i0; i < i1; ++i) c += fkey(XVAL(i)) >= cand;
            if (block_sum<int>(c, shi) >= top_k) tk = cand;
        }
    }
    float z = 0.f;
    for (int i = i0; i < i1; ++i) { const float x = XVAL(i); if (fkey(x) >= tk) z += __expf(x - mx); }
    const float Z = block_sum<float>(z, shf);
    // ---- top-p: tp = largest key whose mass (within the top-k set) is >= top_p * Z
    uint32_t tp = tk;
    if (top_p < 1.0f) {
        const float need = top_p * Z;
        uint32_t t = 0u;
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t cand = t | (1u << bit);
            float m = 0.f;
            for (int i = i0; i < i1; ++i) { const float x = XVAL(i); const uint32_t k = fkey(x); if (k >= cand && k >= tk) m += __expf(x - mx); }
            if (block_sum<float>(m, shf) >= need) t = cand;
        }
        tp = t > tk ? t : tk;
    }
    // ---- draw inside the kept set {key >= tp}
    float mine = 0.f;
    for (int i = i0; i < i1; ++i) { const float x = XVAL(i); if (fkey(x) >= tp) mine += __expf(x - mx); }
    __syncthreads();
    pre[tid] = mine;
    __syncthreads();
    if (tid == 0) {
        unsigned long long s = seed + 0x9E3779B97F4A7C15ull * (unsigned long long)(step + 1) + 0xD1B54A32D192ED03ull * (unsigned long long)(b + 1);
        s = (s ^ (s >> 30)) * 0xBF58476D1CE4E5B9ull; s = (s ^ (s >> 27)) * 0x94D049BB133111EBull; s ^= s >> 31;     // splitmix64 finaliser
        const float u01 = (float)(s >> 40) * (1.0f / 16777216.0f);
        float total = 0.f;
        for (int t = 0; t < 1024; ++t) total += pre[t];
        const float r = u01 * total;
        float acc = 0.f, acc_owner = 0.f;
        int owner = -1;
        for (int t = 0; t < 1024; ++t) {
            if (pre[t] > 0.f) { owner = t; acc_owner = acc; if (acc + pre[t] > r) break; acc += pre[t]; }
        }
        acc = acc_owner;
        // walk the owner's range (owner is the last non-empty range when rounding pushed r past the total)
        int tok = -1;
        const int j0 = owner * chunk, j1 = min(V, j0 + chunk);
        for (int i = j0; i < j1; ++i) {
            const float x = XVAL(i);
            if (fkey(x) >= tp) { tok = i; acc += __expf(x - mx); if (acc > r) break; }
        }
        s_tok = tok < 0 ? 0 : tok;
    }
    __syncthreads();
    if (tid == 0) {
        int tok = finished[b] ? pad_id : s_tok;
        if (eos_id >= 0 && tok == eos_id) finished[b] = 1;
        cur_ids[b] = tok;
        out_ids[(long)b * ld_out + step] = tok;
    }
#undef XVAL
}

}  // namespace

extern "C" int crab_sample_select(crab_ctx* ctx, void* stream, const float* logits, int64_t ldl, int B, int V, int64_t* cur_ids, int64_t* out_ids,
                                  int64_t ld_out, const int32_t* step_dev, int32_t* finished, int eos_id, int pad_id, int min_new_tokens,
                                  float temperature, int top_k, float top_p, uint64_t seed) {
    if (!ctx) return CRAB_E_INVALID;
    if (!logits || !cur_ids || !out_ids || !step_dev || !finished || B <= 0 || V <= 0) return crab_fail(ctx, CRAB_E_INVALID, "sample_select: bad argument");
    if (!(temperature > 0.f) || !(top_p > 0.f) || top_p > 1.0f || top_k < 0) return crab_fail(ctx, CRAB_E_INVALID, "sample_select: temperature > 0, 0 < top_p <= 1, top_k >= 0");
    hipLaunchKernelGGL(sample_select_kernel, dim3(B), dim3(1024), 0, (hipStream_t)stream, logits, (long)ldl, V, cur_ids, out_ids, (long)ld_out, step_dev,
                       finished, eos_id, pad_id, min_new_tokens, 1.0f / temperature, top_k, top_p, (unsigned long long)seed);
    return crab_check_launch(ctx, "sample_select");
}

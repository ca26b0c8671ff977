// Attention kernels for gfx950.
//
// attn_fwd_kernel: flash-style forward, one 256-thread block per (64 query rows, head, batch); each of the
// 4 waves owns 16 query rows.  K tiles [64 keys][HD] (XOR-swizzled) and V^T tiles [HD][64 keys] (row padded
// to 68) are staged in LDS; S^T = K.Q^T and O^T = V^T.P^T run on v_mfma_f32_16x16x32_bf16, so that
//   * each lane's 16 scores of a 64-key tile all belong to ONE query row (col = lane&15): the online
//     softmax needs two cross-lane shuffles (xor 16, 32) per tile instead of a 16-lane reduction,
//   * the exponentiated scores already sit in the register layout the P^T (B-operand) fragment needs, and
//   * the output accumulator holds 4 consecutive head-dim elements of one query row per lane (8-byte stores).
// The contraction index of the second MFMA is mapped as e -> key 16*(2kk + (e>>2)) + 4*(lane>>4) + (e&3); the
// V^T fragment is gathered with the same map (two ds_read_b64), so no transpose or LDS round trip of P.
//
// attn_decode_kernel: one query row per (b,h) streaming the KV cache once (HBM-bound), 16-lane groups
// per key row, fp32 online softmax, cross-group merge through LDS.
#include "common.h"
#include "crab_internal.h"
#include <math.h>
#include <stdlib.h>

namespace {

struct AttnP {
    const bf16_t* q; const bf16_t* k; const bf16_t* vt; bf16_t* o;
    long q_bs, q_hs, q_ss, k_bs, k_hs, k_ss, vt_bs, vt_hs, vt_ds, o_bs, o_ss;
    const float* bias; const float* gate;
    const int* kv_start;                    // optional [B]: keys j < kv_start[b] are masked (left-pad attention_mask)
    const uint32_t* key_mask; long key_mask_ld;   // optional [B][key_mask_ld] words: bit (j & 31) of word j >> 5 = key j is visible (any 2-D attention_mask)
    int B, H, Hk, Sq, Skv, causal;
    float scale;
};

constexpr int KT = 64;          // keys per tile
constexpr int VT_LD = 68;       // padded V^T row (bf16 elements): 136 B, conflict-free ds_read_b64 gathers

template <int HD>
__device__ __forceinline__ int k_swz(int row, int chunk) {
    // 16-byte chunk swizzle; HD=128: 16 chunks per 256-B row; HD=64: 8 chunks per 128-B row; HD=32: 4 chunks per 64-B
    // row with the permutation P = {0,2,3,1} of (row>>2)&3 (conflict-free for the gfx950 ds_read_b128 lane groups)
    return HD == 128 ? (chunk ^ (row & 15)) : (HD == 64 ? (chunk ^ ((row >> 1) & 7)) : (chunk ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3)));
}

template <int HD, bool CAUSAL, bool BIAS>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnP p) {
    constexpr int CPR = HD / 8;             // 16-byte chunks per K row
    constexpr int KS = HD / 32;             // MFMA k-steps over the head dim
    constexpr int DT = HD / 16;             // output d tiles
    __shared__ __attribute__((aligned(16))) bf16_t lk[KT * HD];
    __shared__ __attribute__((aligned(16))) bf16_t lv[HD * VT_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    // 1-D grid, XCD-aware: workgroup ids are dealt round-robin to the 8 XCDs, so with a (q block, head, batch) grid the q blocks
    // of one head landed on 8 different L2s and each re-read the head's K / V through the fabric (FETCH_SIZE 2x the operands).
    // xcd_remap gives every XCD a contiguous range of logical ids: a head's q blocks share one L2.
    const int nqb = (p.Sq + 63) >> 6;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);
    const int b = lid / (nqb * p.H), h = (lid / nqb) % p.H;
    const int hk = h / (p.H / p.Hk);
    const int q0 = (lid % nqb) * 64;
    const int qrow = q0 + wave * 16 + fr;                       // this lane's query row
    const int qload = qrow < p.Sq ? qrow : p.Sq - 1;
    const int koff = p.Skv - p.Sq;                              // causal offset (0 for prefill)

    const bf16_t* qp = p.q + (long)b * p.q_bs + (long)h * p.q_hs + (long)qload * p.q_ss;
    const bf16_t* kp = p.k + (long)b * p.k_bs + (long)hk * p.k_hs;
    const bf16_t* vp = p.vt + (long)b * p.vt_bs + (long)hk * p.vt_hs;

    bf16x8_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 32 + fg * 8);

    f32x4_t oacc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i) oacc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m_run = -1e30f, l_run = 0.f;

    float gate = 1.f;
    const float* biasrow = nullptr;
    if (BIAS) {
        if (p.gate) gate = p.gate[((long)b * p.H + h) * p.Sq + qload];
        biasrow = p.bias + ((long)h * p.Sq + qload) * p.Skv;
    }

    int kv_end = p.Skv;
    if (CAUSAL) {
        int last = q0 + 63 + koff;                               // last key visible to this block
        if (last + 1 < kv_end) kv_end = last + 1;
    }
    const int ntiles = (kv_end + KT - 1) / KT;
    // left-pad mask (forward() with attention_mask, unified_llama.py:149-160): keys below kv_start[b] are invisible; whole tiles
    // below it are skipped.  A query row that sees no key at all (a pad row) ends with l_run == 0 and stores zeros.
    const int ks0 = p.kv_start ? p.kv_start[b] : 0;
    const int t0 = ks0 / KT;
    // general 2-D attention_mask (holes anywhere): one visibility bit per key, tested on every tile (block-uniform pointer test)
    const uint32_t* km = p.key_mask ? p.key_mask + (long)b * p.key_mask_ld : nullptr;

    // Software pipeline: the global loads of tile t+1 are issued (into registers) right after tile t has been stored to LDS,
    // so their latency runs under tile t's MFMAs and softmax.  Synchronous staging left the waves parked 70 % of the time
    // (PMC: SQ_WAIT_ANY / SQ_WAVE_CYCLES, MFMA busy 11 %).
    constexpr int NKV = (KT * CPR) / 256, NVV = (HD * 8) / 256;
    u32x4 kreg[NKV], vreg[NVV];
    // Loads are UNCONDITIONAL (addresses clamped into the operands, nothing selected or branched on before the data is used): with
    // `row < Skv ? load : 0` and a scalar loop for the ragged V^T chunk the compiler could not keep the loads of the next tile in
    // flight across the MFMAs - every iteration ended in s_waitcnt vmcnt(0) right behind the loads it had just issued, i.e. one full
    // memory latency per 64-key tile (ISA of r02's kernel; 9500 cycles per tile against ~1000 of MFMA).  Keys >= Skv: the K row is a
    // copy of the last valid row (its scores are masked), the V^T chunk is masked to zero at the LDS store (0 * stale bits stays 0).
#define ATT_LOAD_TILE(T_)                                                                                      \
    {                                                                                                          \
        const int kvl_ = (T_) * KT;                                                                            \
        _Pragma("unroll") for (int i = 0; i < NKV; ++i) {                                                      \
            const int idx = tid + i * 256;                                                                     \
            const int row = idx / CPR, c = idx % CPR;                                                          \
            const int kr = min(kvl_ + row, p.Skv - 1);                                                         \
            kreg[i] = *reinterpret_cast<const u32x4*>(kp + (long)kr * p.k_ss + c * 8);                         \
        }                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < NVV; ++i) {                                                      \
            const int idx = tid + i * 256;                                                                     \
            const int row = idx >> 3, c = idx & 7;                                                             \
            int kc = kvl_ + c * 8;                         /* vt_ds is a multiple of 8 >= Skv: a chunk that starts below Skv is in the row */ \
            kc = kc < p.Skv ? kc : 0;                                                                          \
            vreg[i] = *reinterpret_cast<const u32x4*>(vp + (long)row * p.vt_ds + kc);                          \
        }                                                                                                      \
    }
    // ragged last tile (wave-uniform test, VALU only): zero the V^T elements of keys >= Skv
#define ATT_MASK_V(T_)                                                                                         \
    if ((T_) * KT + KT > p.Skv) {                                                                              \
        _Pragma("unroll") for (int i = 0; i < NVV; ++i) {                                                      \
            const int c = (tid + i * 256) & 7;                                                                 \
            const int rem = p.Skv - ((T_) * KT + c * 8);   /* valid elements of this chunk (<= 0: none) */      \
            _Pragma("unroll") for (int w = 0; w < 4; ++w) {                                                    \
                const uint32_t m = (2 * w < rem ? 0x0000ffffu : 0u) | (2 * w + 1 < rem ? 0xffff0000u : 0u);    \
                vreg[i][w] &= m;                                                                               \
            }                                                                                                  \
        }                                                                                                      \
    }
    // (a block that sees no key at all - every row a left-pad row - falls through the loop and stores zeros)
    if (t0 < ntiles) ATT_LOAD_TILE(t0)
    for (int t = t0; t < ntiles; ++t) {
        const int kv0 = t * KT;
        __syncthreads();                                         // previous tile fully consumed
        // ---- registers -> LDS: K tile (KT rows x CPR swizzled chunks), V^T tile (HD rows x 8 chunks, padded rows)
        ATT_MASK_V(t)
#pragma unroll
        for (int i = 0; i < NKV; ++i) {
            const int idx = tid + i * 256;
            const int row = idx / CPR, c = idx % CPR;
            *reinterpret_cast<u32x4*>(lk + row * HD + (k_swz<HD>(row, c) << 3)) = kreg[i];
        }
#pragma unroll
        for (int i = 0; i < NVV; ++i) {
            const int idx = tid + i * 256;
            const int row = idx >> 3, c = idx & 7;
            bf16_t* d = lv + row * VT_LD + c * 8;
            *reinterpret_cast<u32x2*>(d) = u32x2{vreg[i][0], vreg[i][1]};
            *reinterpret_cast<u32x2*>(d + 4) = u32x2{vreg[i][2], vreg[i][3]};
        }
        __syncthreads();
        if (t + 1 < ntiles) ATT_LOAD_TILE(t + 1);

        // ---- S^T = K . Q^T : 4 key sub-tiles of 16
        f32x4_t s[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            s[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                int row = j * 16 + fr;
                int chunk = ks * 4 + fg;
                bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(lk + row * HD + (k_swz<HD>(row, chunk) << 3));
                s[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[ks], s[j], 0, 0, 0);
            }
        }
        // ---- scale, bias, mask; lane holds keys kv0 + 16j + 4fg + r for its query row.  Scores are kept in the log2 domain
        // (scale * log2(e) folded into one multiply, exp2 instead of exp); the per-element mask runs only on tiles that
        // can contain masked keys (the diagonal tile of a causal block, the ragged last tile): after the software
        // pipelining the kernel is VALU-bound (PMC: VALU active 30 % of wave cycles at two waves per SIMD).
        const float sc2 = p.scale * 1.4426950408889634f;
        const bool need_mask = (kv0 + KT > p.Skv) || (CAUSAL && (kv0 + KT - 1 > q0 + wave * 16 + koff)) || (kv0 < ks0) || km;
        float tmax = -1e30f;
        if (BIAS || need_mask) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int kv = kv0 + j * 16 + fg * 4 + r;
                    float v = s[j][r] * sc2;
                    if (BIAS) { if (kv < p.Skv) v += gate * biasrow[kv] * 1.4426950408889634f; }
                    bool ok = kv < p.Skv && kv >= ks0;
                    if (CAUSAL) ok = ok && (kv <= qrow + koff);
                    if (km) ok = ok && ((km[min(kv, p.Skv - 1) >> 5] >> (kv & 31)) & 1u);
                    v = ok ? v : -INFINITY;
                    s[j][r] = v;
                    tmax = fmaxf(tmax, v);
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                s[j] *= sc2;
                tmax = fmaxf(tmax, fmaxf(fmaxf(s[j][0], s[j][1]), fmaxf(s[j][2], s[j][3])));
            }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);                // stays at the finite -1e30 while every key so far is masked
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float e = __builtin_amdgcn_exp2f(s[j][r] - m_new);     // raw v_exp_f32 (exp2f() adds range handling)
                s[j][r] = e;
                psum += e;
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int i = 0; i < DT; ++i) oacc[i] *= alpha;

        // ---- O^T += V^T . P^T : two 32-key contraction chunks
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            union { bf16x8_t v; uint32_t w[4]; } pf;
            pf.w[0] = pack_bf2(s[2 * kk][0], s[2 * kk][1]);
            pf.w[1] = pack_bf2(s[2 * kk][2], s[2 * kk][3]);
            pf.w[2] = pack_bf2(s[2 * kk + 1][0], s[2 * kk + 1][1]);
            pf.w[3] = pack_bf2(s[2 * kk + 1][2], s[2 * kk + 1][3]);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                const bf16_t* vr = lv + (dt * 16 + fr) * VT_LD + kk * 32 + fg * 4;
                union { bf16x8_t v; u32x2 h[2]; } vf;
                vf.h[0] = *reinterpret_cast<const u32x2*>(vr);
                vf.h[1] = *reinterpret_cast<const u32x2*>(vr + 16);
                oacc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf.v, pf.v, oacc[dt], 0, 0, 0);
            }
        }
    }
    // ---- finish: total row sum over the 4 lane groups, normalise, store 4 consecutive d per lane
    l_run += __shfl_xor(l_run, 16, 64);
    l_run += __shfl_xor(l_run, 32, 64);
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;        // no visible key (left-pad query row): zeros, never read by a valid row
    if (qrow < p.Sq) {
        bf16_t* op = p.o + (long)b * p.o_bs + (long)qrow * p.o_ss + (long)h * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            u32x2 w;
            w[0] = pack_bf2(oacc[dt][0] * inv, oacc[dt][1] * inv);
            w[1] = pack_bf2(oacc[dt][2] * inv, oacc[dt][3] * inv);
            *reinterpret_cast<u32x2*>(op + dt * 16 + fg * 4) = w;
        }
    }
}


// ---------------------------------------------------------------------------------------------- forward, 128 query rows per block
// attn_fwd32_kernel: the same flash forward on v_mfma_f32_32x32x16_bf16 with 32 query rows per wave (128 per block) for Sq > 64 (decoder
// prefill, CLIP).  Why: with 16 rows per wave every 16-cycle MFMA needs a fresh 1-KiB K or V^T fragment from LDS - 64 B/clk per SIMD,
// 256 B/clk per CU against the 128 the LDS delivers - and every 64-key tile costs two block barriers; the 16-row kernel ran the
// S = 702 prefill at 0.34 PFLOP/s (13 % of the matrix peak).  Here a 1-KiB fragment feeds a 32-cycle MFMA (half the LDS bytes per
// flop), the online softmax needs ONE cross-lane exchange (lane ^ 32) because the 32 scores a lane holds per tile all belong to one
// query row, the K / V^T tiles are double-buffered in LDS (one barrier per tile, next tile's global loads in flight under the
// MFMAs), the output rescale is skipped while no row of the wave raised its maximum, and a wave whose rows cannot see a causal tile
// (or lie beyond Sq) skips its arithmetic.  Heavy (late) causal query blocks are scheduled first.
// Fragment maps (l = lane, c = l & 31, g = l >> 5):
//   S^T = K.Q^T   A = K rows (key 32 j + c, head-dim 16 ks + 8 g + e)      B = Q (query c, same head-dim)      D[r]: key 8 (r >> 2) + 4 g + (r & 3), query c
//   O^T += V^T.P^T  contraction index (g, e) <-> key 32 j + 16 kk + 8 (e >> 2) + 4 g + (e & 3): B = the lane's own exp'd scores
//                 s[j][8 kk + e] (no shuffle, no LDS round trip), A = V^T row d = 32 dt + c gathered with the same map (two ds_read_b64)
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256, 2) void attn_fwd32_kernel(AttnP p) {
    constexpr int CPR = HD / 8;             // 16-byte chunks per K row
    constexpr int KS = HD / 16;             // MFMA k-steps over the head dim
    constexpr int DT = HD / 32;             // output d tiles
    constexpr int KBUF = KT * HD, VBUF = HD * VT_LD;
    __shared__ __attribute__((aligned(16))) bf16_t lk[2 * KBUF];
    __shared__ __attribute__((aligned(16))) bf16_t lv[2 * VBUF];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fc = lane & 31, fg = lane >> 5;
    const int nqb = (p.Sq + 127) >> 7;
    const int lid = xcd_remap(blockIdx.x, gridDim.x);           // a head's query blocks share one XCD's L2 (K / V re-reads)
    const int b = lid / (nqb * p.H), h = (lid / nqb) % p.H;
    const int hk = h / (p.H / p.Hk);
    const int q0 = (nqb - 1 - lid % nqb) * 128;                 // longest causal blocks first
    const int qw0 = q0 + wave * 32;                             // first query row of this wave
    const int qrow = qw0 + fc;
    const int qload = qrow < p.Sq ? qrow : p.Sq - 1;
    const int koff = p.Skv - p.Sq;
    const bool wave_on = qw0 < p.Sq;

    const bf16_t* qp = p.q + (long)b * p.q_bs + (long)h * p.q_hs + (long)qload * p.q_ss;
    const bf16_t* kp = p.k + (long)b * p.k_bs + (long)hk * p.k_hs;
    const bf16_t* vp = p.vt + (long)b * p.vt_bs + (long)hk * p.vt_hs;

    bf16x8_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qp + ks * 16 + fg * 8);

    f32x16_t oacc[DT];
#pragma unroll
    for (int i = 0; i < DT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) oacc[i][r] = 0.f;
    float m_run = -1e30f, l_run = 0.f;

    int kv_end = p.Skv;
    if (CAUSAL) kv_end = min(kv_end, q0 + 128 + koff);
    const int ntiles = (kv_end + KT - 1) / KT;
    const int ks0 = p.kv_start ? p.kv_start[b] : 0;             // left-pad mask: keys below it are invisible, whole tiles skipped
    const int t0 = ks0 / KT;
    const uint32_t* km = p.key_mask ? p.key_mask + (long)b * p.key_mask_ld : nullptr;      // general 2-D attention_mask: a visibility bit per key
    constexpr int NKV = (KT * CPR) / 256, NVV = (HD * 8) / 256;
    u32x4 kreg[NKV], vreg[NVV];
#define ATT_STORE_TILE(BUF_, T_)                                                                               \
    {                                                                                                          \
        ATT_MASK_V(T_)                                                                                         \
        bf16_t* lk_ = lk + (BUF_) * KBUF;                                                                      \
        bf16_t* lv_ = lv + (BUF_) * VBUF;                                                                      \
        _Pragma("unroll") for (int i = 0; i < NKV; ++i) {                                                      \
            const int idx = tid + i * 256;                                                                     \
            const int row = idx / CPR, c = idx % CPR;                                                          \
            *reinterpret_cast<u32x4*>(lk_ + row * HD + (k_swz<HD>(row, c) << 3)) = kreg[i];                    \
        }                                                                                                      \
        _Pragma("unroll") for (int i = 0; i < NVV; ++i) {                                                      \
            const int idx = tid + i * 256;                                                                     \
            const int row = idx >> 3, c = idx & 7;                                                             \
            bf16_t* d_ = lv_ + row * VT_LD + c * 8;                                                            \
            *reinterpret_cast<u32x2*>(d_) = u32x2{vreg[i][0], vreg[i][1]};                                     \
            *reinterpret_cast<u32x2*>(d_ + 4) = u32x2{vreg[i][2], vreg[i][3]};                                 \
        }                                                                                                      \
    }
    if (t0 >= ntiles) {                                         // block-uniform: every row is a left-pad row that sees no key -> zeros
        if (qrow < p.Sq) {
            bf16_t* op = p.o + (long)b * p.o_bs + (long)qrow * p.o_ss + (long)h * HD;
#pragma unroll
            for (int c = 0; c < HD / 4; ++c) *reinterpret_cast<u32x2*>(op + c * 4 + 0) = u32x2{0u, 0u};
        }
        return;
    }
    // straight-line from here: the wait in front of the first LDS store also covers the (older) Q fragment loads on EVERY path into the
    // loop, so the loop body carries no vmcnt wait except the one for the tile it is about to store
    ATT_LOAD_TILE(t0)
    ATT_STORE_TILE(0, t0)
    if (t0 + 1 < ntiles) ATT_LOAD_TILE(t0 + 1)
    __syncthreads();
    const float sc2 = p.scale * 1.4426950408889634f;            // scores in the log2 domain: one multiply, exp2 instead of exp
    for (int t = t0; t < ntiles; ++t) {
        const int cur = (t - t0) & 1;
        const int kv0 = t * KT;
        if (t + 1 < ntiles) {
            // tile t+1 (in registers since the previous iteration) -> the other buffer, whose readers all passed the last barrier;
            // then the loads of tile t+2 go out and fly under this tile's MFMAs
            ATT_STORE_TILE(cur ^ 1, t + 1)
            if (t + 2 < ntiles) ATT_LOAD_TILE(t + 2)
        }
        const bool skip = !wave_on || (CAUSAL && kv0 > qw0 + 31 + koff);       // wave-uniform: nothing of this tile is visible to the wave
        if (!skip) {
            const bf16_t* lkc = lk + cur * KBUF;
            const bf16_t* lvc = lv + cur * VBUF;
            // ---- S^T = K . Q^T : 2 key sub-tiles of 32, the two accumulator chains interleaved (an MFMA never waits for the one issued
            // right before it)
            f32x16_t s[2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[j][r] = 0.f;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int row = j * 32 + fc;
                    const int chunk = ks * 2 + fg;
                    bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(lkc + row * HD + (k_swz<HD>(row, chunk) << 3));
                    s[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], s[j], 0, 0, 0);
                }
            }
            // ---- mask (only on tiles that can hold masked keys: the diagonal of a causal block, the ragged last tile, the tile the
            // left-pad boundary falls in); lane holds keys kv0 + 32 j + 8 (r >> 2) + 4 g + (r & 3) of its query row.  The scores stay RAW:
            // the running maximum is kept in the scaled log2 domain and the scale rides in the exponent's fma, exp2(s * sc2 - m).
            const bool need_mask = (kv0 + KT > p.Skv) || (CAUSAL && (kv0 + KT - 1 > qw0 + koff)) || (kv0 < ks0) || km;
            if (need_mask) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kv = kv0 + j * 32 + (r >> 2) * 8 + fg * 4 + (r & 3);
                        bool ok = kv < p.Skv && kv >= ks0;
                        if (CAUSAL) ok = ok && (kv <= qrow + koff);
                        if (km) ok = ok && ((km[min(kv, p.Skv - 1) >> 5] >> (kv & 31)) & 1u);
                        s[j][r] = ok ? s[j][r] : -INFINITY;
                    }
            }
            float tmax = -INFINITY;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; r += 2) tmax = fmaxf(tmax, fmaxf(s[j][r], s[j][r + 1]));
            tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
            const float m_new = fmaxf(m_run, tmax * sc2);      // stays at the finite -1e30 while every key so far is masked (sc2 > 0)
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
            const bool raised = m_new != m_run;
            m_run = m_new;
            float psum = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = __builtin_amdgcn_exp2f(__builtin_fmaf(s[j][r], sc2, -m_new));
                    s[j][r] = e;
                    psum += e;
                }
            l_run = l_run * alpha + psum;
            if (__any(raised)) {                                 // alpha == 1 on every lane otherwise
#pragma unroll
                for (int i = 0; i < DT; ++i) oacc[i] *= alpha;
            }
            // ---- O^T += V^T . P^T : four 16-key contraction chunks
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    union { bf16x8_t v; uint32_t w[4]; } pf;
                    pf.w[0] = pack_bf2(s[j][8 * kk + 0], s[j][8 * kk + 1]);
                    pf.w[1] = pack_bf2(s[j][8 * kk + 2], s[j][8 * kk + 3]);
                    pf.w[2] = pack_bf2(s[j][8 * kk + 4], s[j][8 * kk + 5]);
                    pf.w[3] = pack_bf2(s[j][8 * kk + 6], s[j][8 * kk + 7]);
#pragma unroll
                    for (int dt = 0; dt < DT; ++dt) {
                        const bf16_t* vr = lvc + (dt * 32 + fc) * VT_LD + j * 32 + kk * 16 + fg * 4;
                        union { bf16x8_t v; u32x2 hh[2]; } vf;
                        vf.hh[0] = *reinterpret_cast<const u32x2*>(vr);
                        vf.hh[1] = *reinterpret_cast<const u32x2*>(vr + 8);
                        oacc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf.v, pf.v, oacc[dt], 0, 0, 0);
                    }
                }
        }
        __syncthreads();                                         // tile t consumed by every wave, tile t+1 visible in the other buffer
    }
    // ---- finish: the row sum lives in the two lanes of a query (l, l ^ 32); normalise, store 4 consecutive d per (dt, r >> 2)
    l_run += __shfl_xor(l_run, 32, 64);
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;        // no visible key (left-pad query row): zeros, never read by a valid row
    if (qrow < p.Sq) {
        bf16_t* op = p.o + (long)b * p.o_bs + (long)qrow * p.o_ss + (long)h * HD;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                u32x2 w;
                w[0] = pack_bf2(oacc[dt][4 * g4 + 0] * inv, oacc[dt][4 * g4 + 1] * inv);
                w[1] = pack_bf2(oacc[dt][4 * g4 + 2] * inv, oacc[dt][4 * g4 + 3] * inv);
                *reinterpret_cast<u32x2*>(op + dt * 32 + g4 * 8 + fg * 4) = w;
            }
    }
}
#undef ATT_LOAD_TILE
#undef ATT_MASK_V
#undef ATT_STORE_TILE

// ---------------------------------------------------------------------------------------------- decode
// block = 256 threads = 16 groups of 16 lanes; group gidx handles keys gidx, gidx+16, ...; each lane owns
// EPL = HD/16 consecutive head-dim elements.  Every K / V row is read exactly once per step by exactly one block, so the
// loads carry the non-temporal hint (global_load ... nt): measured 562 -> 537 us per launch in the benchmark (6.2 -> 6.5 TB/s;
// 6.8 TB/s in isolation) - the stream no longer displaces the weights and activations the neighbouring GEMMs keep in L2 / MALL.
template <int HD>
__global__ __launch_bounds__(256) void attn_decode_kernel(const bf16_t* __restrict__ q, long ldq, const bf16_t* __restrict__ kc,
                                                          const bf16_t* __restrict__ vc, bf16_t* __restrict__ o, long ldo, int H, int Hk,
                                                          int Tmax, int ctx_host, const int* __restrict__ ctx_dev, float scale,
                                                          const int* __restrict__ kv_start) {
    constexpr int EPL = HD / 16;
    __shared__ float sm[16], sl[16];
    __shared__ float so[16][HD];
    const int tid = threadIdx.x;
    const int grp = tid >> 4, sub = tid & 15;
    const int b = blockIdx.y, h = blockIdx.x;
    const int hk = h / (H / Hk);
    const int ks0 = kv_start ? kv_start[b] : 0;                 // left-pad mask: the first ks0 cache rows of this sequence are invisible
    const int ctx = ctx_host + (ctx_dev ? ctx_dev[0] : 0) - ks0;
    const bf16_t* qp = q + (long)b * ldq + (long)h * HD + sub * EPL;
    float qv[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) qv[e] = bf2f(qp[e]) * scale;
    const bf16_t* kb = kc + (((long)b * Hk + hk) * (long)Tmax + ks0) * HD + sub * EPL;
    const bf16_t* vb = vc + (((long)b * Hk + hk) * (long)Tmax + ks0) * HD + sub * EPL;
    float m = -1e30f, l = 0.f, acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;

    if (EPL == 8) {
        // TWO keys per trip (j and j + 16) with the next pair requested before the current one is consumed: every lane keeps four
        // 16-byte K and four 16-byte V loads in flight, and the loop-carried online-softmax chain (max, two exps, rescale of the
        // accumulators) is paid once per two keys
        const u32x4 z4 = {0u, 0u, 0u, 0u};
        u32x4 k0 = z4, v0 = z4, k1 = z4, v1 = z4;
        if (grp < ctx) {
            k0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kb + (long)grp * HD));
            v0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vb + (long)grp * HD));
        }
        if (grp + 16 < ctx) {
            k1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kb + (long)(grp + 16) * HD));
            v1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vb + (long)(grp + 16) * HD));
        }
        for (int j = grp; j < ctx; j += 32) {
            u32x4 kn0 = z4, vn0 = z4, kn1 = z4, vn1 = z4;
            if (j + 32 < ctx) {
                kn0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kb + (long)(j + 32) * HD));
                vn0 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vb + (long)(j + 32) * HD));
            }
            if (j + 48 < ctx) {
                kn1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kb + (long)(j + 48) * HD));
                vn1 = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vb + (long)(j + 48) * HD));
            }
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s0 += qv[2 * e] * lo_bf(k0[e]) + qv[2 * e + 1] * hi_bf(k0[e]);
                s1 += qv[2 * e] * lo_bf(k1[e]) + qv[2 * e + 1] * hi_bf(k1[e]);
            }
            s0 = row16_sum(s0); s1 = row16_sum(s1);
            const bool has1 = j + 16 < ctx;                      // group-uniform
            const float mn = fmaxf(m, has1 ? fmaxf(s0, s1) : s0);
            const float a = __expf(m - mn), p0 = __expf(s0 - mn), p1 = has1 ? __expf(s1 - mn) : 0.f;
            l = l * a + (p0 + p1);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[2 * e] = acc[2 * e] * a + (p0 * lo_bf(v0[e]) + p1 * lo_bf(v1[e]));
                acc[2 * e + 1] = acc[2 * e + 1] * a + (p0 * hi_bf(v0[e]) + p1 * hi_bf(v1[e]));
            }
            m = mn;
            k0 = kn0; v0 = vn0; k1 = kn1; v1 = vn1;
        }
    } else {
    for (int j = grp; j < ctx; j += 16) {
        float kx[EPL], vx[EPL];
        {
            u32x2 kw = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(kb + (long)j * HD));
            u32x2 vw = __builtin_nontemporal_load(reinterpret_cast<const u32x2*>(vb + (long)j * HD));
#pragma unroll
            for (int e = 0; e < 2; ++e) { kx[2 * e] = lo_bf(kw[e]); kx[2 * e + 1] = hi_bf(kw[e]); vx[2 * e] = lo_bf(vw[e]); vx[2 * e + 1] = hi_bf(vw[e]); }
        }
        float sdot = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) sdot += qv[e] * kx[e];
        sdot = row16_sum(sdot);
        float mn = fmaxf(m, sdot);
        float a = __expf(m - mn), pw = __expf(sdot - mn);
        l = l * a + pw;
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[e] = acc[e] * a + pw * vx[e];
        m = mn;
    }
    }
    if (sub == 0) { sm[grp] = m; sl[grp] = l; }
#pragma unroll
    for (int e = 0; e < EPL; ++e) so[grp][sub * EPL + e] = acc[e];
    __syncthreads();
    if (tid < HD) {
        float M = -1e30f;
#pragma unroll
        for (int g = 0; g < 16; ++g) M = fmaxf(M, sm[g]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            float w = __expf(sm[g] - M);
            L += sl[g] * w;
            O += so[g][tid] * w;
        }
        o[(long)b * ldo + (long)h * HD + tid] = f2bf(O / L);
    }
}


// ---------------------------------------------------------------------------------------------- decode under a general key mask
// forward()'s one-token shortcut (models/unified_llama.py:125-127) with a 2-D attention_mask that has holes anywhere (HF accepts any
// mask: modeling_attn_mask_utils' padding mask AND-ed with the causal one): the per-(b, h) kernel above with a visibility bit per cache
// row - bit (j & 31) of word j >> 5 of the sequence's mask row; invisible rows are skipped before their loads.  A rare path (the eval
// loop never masks the cache), kept out of the benchmark kernels: plain loads, one key per 16-lane group and trip.
template <int HD>
__global__ __launch_bounds__(256) void attn_decode_keymask_kernel(const bf16_t* __restrict__ q, long ldq, const bf16_t* __restrict__ kc,
                                                                  const bf16_t* __restrict__ vc, bf16_t* __restrict__ o, long ldo, int H, int Hk,
                                                                  int Tmax, int ctx_host, const int* __restrict__ ctx_dev, float scale,
                                                                  const uint32_t* __restrict__ key_mask, long key_mask_ld) {
    constexpr int EPL = HD / 16, WPL = EPL / 2;
    __shared__ float sm[16], sl[16];
    __shared__ float so[16][HD];
    const int tid = threadIdx.x;
    const int grp = tid >> 4, sub = tid & 15;
    const int b = blockIdx.y, h = blockIdx.x;
    const int hk = h / (H / Hk);
    const int ctx = ctx_host + (ctx_dev ? ctx_dev[0] : 0);
    const uint32_t* km = key_mask + (long)b * key_mask_ld;
    const bf16_t* qp = q + (long)b * ldq + (long)h * HD + sub * EPL;
    float qv[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) qv[e] = bf2f(qp[e]) * scale;
    const bf16_t* kb = kc + ((long)b * Hk + hk) * (long)Tmax * HD + sub * EPL;
    const bf16_t* vb = vc + ((long)b * Hk + hk) * (long)Tmax * HD + sub * EPL;
    float m = -1e30f, l = 0.f, acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    for (int j = grp; j < ctx; j += 16) {
        if (!((km[j >> 5] >> (j & 31)) & 1u)) continue;          // group-uniform
        uint32_t kw[WPL], vw[WPL];
#pragma unroll
        for (int w = 0; w < WPL; ++w) {
            kw[w] = reinterpret_cast<const uint32_t*>(kb + (long)j * HD)[w];
            vw[w] = reinterpret_cast<const uint32_t*>(vb + (long)j * HD)[w];
        }
        float sdot = 0.f;
#pragma unroll
        for (int w = 0; w < WPL; ++w) sdot += qv[2 * w] * lo_bf(kw[w]) + qv[2 * w + 1] * hi_bf(kw[w]);
        sdot = row16_sum(sdot);
        const float mn = fmaxf(m, sdot);
        const float a = __expf(m - mn), pw = __expf(sdot - mn);
        l = l * a + pw;
#pragma unroll
        for (int w = 0; w < WPL; ++w) {
            acc[2 * w] = acc[2 * w] * a + pw * lo_bf(vw[w]);
            acc[2 * w + 1] = acc[2 * w + 1] * a + pw * hi_bf(vw[w]);
        }
        m = mn;
    }
    if (sub == 0) { sm[grp] = m; sl[grp] = l; }
#pragma unroll
    for (int e = 0; e < EPL; ++e) so[grp][sub * EPL + e] = acc[e];
    __syncthreads();
    if (tid < HD) {
        float M = -1e30f;
#pragma unroll
        for (int g = 0; g < 16; ++g) M = fmaxf(M, sm[g]);
        float L = 0.f, O = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) {
            const float w = __expf(sm[g] - M);
            L += sl[g] * w;
            O += so[g][tid] * w;
        }
        o[(long)b * ldo + (long)h * HD + tid] = f2bf(L > 0.f ? O / L : 0.f);     // no visible key at all (a pad row): zeros, like the prefill kernels
    }
}


// ---------------------------------------------------------------------------------------------- decode, small batch (B * H <= 256)
// attn_decode_rope_kernel: the decode attention of the reference's own batch sizes (1 and 8 clips, scripts/quick_start.py:43,
// inference_hyper_lora.py:1477) with what surrounded it folded in:
//   * RoPE of q and of the new k (modeling_llama.py:204-236) and the KV-cache append (:408-412) happen HERE, from the raw packed q|k|v
//     row: the separate rotate / scatter launch (4.8 us of a 135 us layer) is gone.  Lane `sub` owns EPL consecutive dims of the head, its
//     rotation partner (dim +- d/2) is lane sub ^ 8 of the same 16-lane group: one shuffle per element.  q and k are rounded to bf16
//     after the rotation exactly like the stored form (rope_lo / rope_hi, common.h), so the result is bit-identical to the unfused pair.
//     The new key / value never travel through the cache for THIS step: the split that owns position `pos` takes them from registers,
//     and the block (h % G == 0, split 0) appends them for the following steps - no inter-block dependency.
//   * the context is SPLIT over blockIdx.z when (b, h) alone cannot fill the chip (one clip = 32 blocks on 256 CUs: 19 us per layer at
//     0.7 TB/s): every split writes its (max, sum, weighted V sum) with write-through stores, takes a ticket on a per-(b, h) counter
//     and the LAST one merges the splits in split order - nobody waits, the order of arrival does not matter (deterministic).
//     The counters must be zero at entry; the kernel leaves them zero (the caller zero-fills the workspace once).
template <int HD, int NG = 16>                              // NG 16-lane key groups per block: 16 (256 threads) or 32 (512 threads, see the launch)
__global__ __launch_bounds__(NG * 16) void attn_decode_rope_kernel(const bf16_t* __restrict__ qkv, long ldq, const float* __restrict__ tab,
                                                               bf16_t* __restrict__ kc, bf16_t* __restrict__ vc, bf16_t* __restrict__ o, long ldo,
                                                               int H, int Hk, int Tmax, int pos0, const int* __restrict__ pos_dev, float scale,
                                                               float* part, unsigned* counters) {
    constexpr int EPL = HD / 16, WPL = EPL / 2;                 // elements / 32-bit words per lane
    __shared__ float sm[NG], sl[NG];
    __shared__ float so[NG][HD];
    __shared__ unsigned s_old;
    const int tid = threadIdx.x;
    const int grp = tid >> 4, sub = tid & 15;
    const int h = blockIdx.x, b = blockIdx.y, z = blockIdx.z, nsplit = gridDim.z;
    const int G = H / Hk, hk = h / G;
    const int pos = pos0 + (pos_dev ? pos_dev[0] : 0);         // the token being decoded: keys 0 .. pos are visible
    // ---- q, new k (rotated, rounded like the stored form) and new v of this head
    const bool hi_half = sub >= 8;
    const float* cs = tab + ((long)pos * (HD / 2) + (sub & 7) * EPL) * 2;
    const bf16_t* row = qkv + (long)b * ldq;
    uint32_t qw[WPL], kw[WPL], vw[WPL];
#pragma unroll
    for (int i = 0; i < WPL; ++i) {
        qw[i] = reinterpret_cast<const uint32_t*>(row + (long)h * HD + sub * EPL)[i];
        kw[i] = reinterpret_cast<const uint32_t*>(row + (long)(H + hk) * HD + sub * EPL)[i];
        vw[i] = reinterpret_cast<const uint32_t*>(row + (long)(H + Hk + hk) * HD + sub * EPL)[i];
    }
    // ---- this split's keys: cached rows [kb0, kb0 + nc), and the new key from registers when pos falls in the split
    const long crow = ((long)b * Hk + hk) * (long)Tmax;
    const int chunk = (pos + nsplit) / nsplit;                  // ceil((pos + 1) / nsplit)
    const int kb0 = z * chunk, ke0 = min(pos + 1, kb0 + chunk);
    const int nc = min(ke0, pos) - kb0;
    const bool has_new = pos >= kb0 && pos < ke0;
    const bf16_t* kb = kc + (crow + kb0) * HD + sub * EPL;
    const bf16_t* vb = vc + (crow + kb0) * HD + sub * EPL;
    // HD = 128: the first NB cached keys of every 16-lane group (8 x 16 = 128 keys per block: a whole split at the context lengths of the
    // benchmark) are requested HERE, unconditionally (clamped rows, masked later), in the same memory round trip as the q|k|v row and the
    // rotation table: r02's two-keys-per-trip loop with conditional loads cost one round trip per 32 keys plus one for the prologue (ISA: the
    // compiler waited for the prefetched pair at the bottom of every trip) - ~5 of this kernel's 12 us at one clip
    constexpr int NB = 8;
    u32x4 kk[EPL == 8 ? NB : 1], vv[EPL == 8 ? NB : 1];
    if (EPL == 8) {
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            // clamped in ABSOLUTE rows: an empty split (kb0 beyond pos; nc <= 0) must not read past this head's cache rows - it reads row
            // max(pos - 1, 0) or an earlier one, whose value is never used
            const int ja = max(min(kb0 + grp + NG * u, kb0 + nc - 1), 0) - kb0;
            kk[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kb + (long)ja * HD));
            vv[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vb + (long)ja * HD));
        }
    }
    float qv[EPL], kn[EPL], vn[EPL];
    uint32_t kst[WPL];
#pragma unroll
    for (int i = 0; i < WPL; ++i) {
        float r2[2][2];                                          // [q | k][lo | hi element of the word]
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const float c = cs[2 * (2 * i + e)], sn = cs[2 * (2 * i + e) + 1];
            const float xq = e ? hi_bf(qw[i]) : lo_bf(qw[i]), xk = e ? hi_bf(kw[i]) : lo_bf(kw[i]);
            const float pq = row_xor8(xq), pk = row_xor8(xk);
            r2[0][e] = hi_half ? rope_hi(pq, xq, c, sn) : rope_lo(xq, pq, c, sn);
            r2[1][e] = hi_half ? rope_hi(pk, xk, c, sn) : rope_lo(xk, pk, c, sn);
        }
        const uint32_t qs = pack_bf2(r2[0][0], r2[0][1]);
        kst[i] = pack_bf2(r2[1][0], r2[1][1]);
        qv[2 * i] = lo_bf(qs) * scale; qv[2 * i + 1] = hi_bf(qs) * scale;
        kn[2 * i] = lo_bf(kst[i]); kn[2 * i + 1] = hi_bf(kst[i]);
        vn[2 * i] = lo_bf(vw[i]); vn[2 * i + 1] = hi_bf(vw[i]);
    }
    if (z == 0 && h % G == 0 && grp == 0) {                     // append for the following steps
#pragma unroll
        for (int i = 0; i < WPL; ++i) {
            reinterpret_cast<uint32_t*>(kc + (crow + pos) * HD + sub * EPL)[i] = kst[i];
            reinterpret_cast<uint32_t*>(vc + (crow + pos) * HD + sub * EPL)[i] = vw[i];
        }
    }
    float m = -1e30f, l = 0.f, acc[EPL];
#pragma unroll
    for (int e = 0; e < EPL; ++e) acc[e] = 0.f;
    if (EPL == 8) {
        // NB keys per group and trip: NB scores, ONE running-max update and rescale, then the weighted V rows; keys beyond the split are
        // masked (probability 0, V row replaced by zeros: the clamped row may hold anything)
        for (int j0 = 0; j0 < nc; j0 += NG * NB) {
            if (j0 > 0) {
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int jc = min(j0 + grp + NG * u, nc - 1);
                    kk[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kb + (long)jc * HD));
                    vv[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vb + (long)jc * HD));
                }
            }
            float sc[NB], mx = -1e30f;
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                float d = 0.f;
#pragma unroll
                for (int e = 0; e < 4; ++e) d += qv[2 * e] * lo_bf(kk[u][e]) + qv[2 * e + 1] * hi_bf(kk[u][e]);
                d = row16_sum(d);
                sc[u] = j0 + grp + NG * u < nc ? d : -INFINITY;  // group-uniform
                mx = fmaxf(mx, sc[u]);
            }
            const float mn = fmaxf(m, mx);
            const float a = __expf(m - mn);
            float ps = 0.f, t[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) t[e] = 0.f;
#pragma unroll
            for (int u = 0; u < NB; ++u) {
                const bool ok = j0 + grp + NG * u < nc;
                const float pw = __expf(sc[u] - mn);            // exp(-inf) = 0 for the masked keys
                ps += pw;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t w = ok ? vv[u][e] : 0u;
                    t[2 * e] += pw * lo_bf(w);
                    t[2 * e + 1] += pw * hi_bf(w);
                }
            }
            l = l * a + ps;
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[e] = acc[e] * a + t[e];
            m = mn;
        }
    } else {
    for (int j = grp; j < nc; j += NG) {
        uint32_t kk2[WPL], vv2[WPL];
#pragma unroll
        for (int i = 0; i < WPL; ++i) {
            kk2[i] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(kb + (long)j * HD) + i);
            vv2[i] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(vb + (long)j * HD) + i);
        }
        float sdot = 0.f;
#pragma unroll
        for (int i = 0; i < WPL; ++i) sdot += qv[2 * i] * lo_bf(kk2[i]) + qv[2 * i + 1] * hi_bf(kk2[i]);
        sdot = row16_sum(sdot);
        const float mn = fmaxf(m, sdot);
        const float a = __expf(m - mn), pw = __expf(sdot - mn);
        l = l * a + pw;
#pragma unroll
        for (int i = 0; i < WPL; ++i) {
            acc[2 * i] = acc[2 * i] * a + pw * lo_bf(vv2[i]);
            acc[2 * i + 1] = acc[2 * i + 1] * a + pw * hi_bf(vv2[i]);
        }
        m = mn;
    }
    }
    {
        float sdot = 0.f;
#pragma unroll
        for (int e = 0; e < EPL; ++e) sdot += qv[e] * kn[e];
        sdot = row16_sum(sdot);
        if (has_new && grp == 0) {
            const float mn = fmaxf(m, sdot);
            const float a = __expf(m - mn), pw = __expf(sdot - mn);
            l = l * a + pw;
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[e] = acc[e] * a + pw * vn[e];
            m = mn;
        }
    }
    if (sub == 0) { sm[grp] = m; sl[grp] = l; }
#pragma unroll
    for (int e = 0; e < EPL; ++e) so[grp][sub * EPL + e] = acc[e];
    __syncthreads();
    float Mx = -1e30f, L = 0.f, O = 0.f;
    if (tid < HD) {
#pragma unroll
        for (int g = 0; g < NG; ++g) Mx = fmaxf(Mx, sm[g]);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const float w = __expf(sm[g] - Mx);
            L += sl[g] * w;
            O += so[g][tid] * w;
        }
    }
    if (nsplit == 1) {
        if (tid < HD) o[(long)b * ldo + (long)h * HD + tid] = f2bf(O / L);
        return;
    }
    // ---- publish this split, take a ticket, the last one merges
    float* mine = part + ((long)(b * H + h) * nsplit + z) * (HD + 2);
    if (tid < HD) {
        __hip_atomic_store(mine + tid, O, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid == 0) {
            __hip_atomic_store(mine + HD, Mx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mine + HD + 1, L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_old = __hip_atomic_fetch_add(counters + b * H + h, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_old != (unsigned)(nsplit - 1)) return;
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        counters[b * H + h] = 0u;                               // ready for the next step
    }
    __syncthreads();
    if (tid < HD) {
        // every partial of every split is requested before the first is used (<= 8 splits: the workspace is sized for 8): the two loops over
        // a runtime split count compiled to one dependent load after the other - up to 16 serial trips to memory on the critical path
        const float* base = part + (long)(b * H + h) * nsplit * (HD + 2);
        constexpr int MAXS = 8;
        float mv[MAXS], lv[MAXS], ov[MAXS];
#pragma unroll
        for (int s2 = 0; s2 < MAXS; ++s2) {
            const float* bs = base + min(s2, nsplit - 1) * (HD + 2);
            mv[s2] = bs[HD]; lv[s2] = bs[HD + 1]; ov[s2] = bs[tid];
        }
        float Mt = -1e30f;
#pragma unroll
        for (int s2 = 0; s2 < MAXS; ++s2)
            if (s2 < nsplit) Mt = fmaxf(Mt, mv[s2]);
        float Lt = 0.f, Ot = 0.f;
#pragma unroll
        for (int s2 = 0; s2 < MAXS; ++s2) {                     // split order, whatever the order of arrival
            if (s2 < nsplit) {
                const float w = __expf(mv[s2] - Mt);
                Lt += lv[s2] * w;
                Ot += ov[s2] * w;
            }
        }
        o[(long)b * ldo + (long)h * HD + tid] = f2bf(Ot / Lt);
    }
}

// ---------------------------------------------------------------------------------------------- decode, grouped-query
// One block per (b, kv head): every K and V row is read ONCE for the G query heads that share it (the per-(b,h) kernel
// above re-reads it G times; Qwen2-7B: G = 7).  Keys are processed in chunks of GQ_CH:
//   A. scores on the matrix pipe (r03): a wave takes 16 keys per tile, S[head][key] = Q[16 x 128] . K^T with the G query heads as rows 0..G-1
//      of the A operand (rows >= G are zero) and the K rows loaded STRAIGHT into B-fragment shape (lane (key = l & 15, g = l >> 4) reads the
//      16 bytes at d = 32 ks + 8 g of its key row for each of the 4 k-steps) - 4 MFMAs per 16 keys replace 56 fmas + a 15-exchange transpose
//      butterfly per key row and lane, and the fp32 copy of q (56 VGPRs at G = 7) is gone -> scores[key][8] in LDS
//   B. per head: chunk max, running max / rescale factor, p = exp(s - m) in place, running sum (once per chunk)
//   C. each 16-lane group takes a V row, reads its 8 probabilities (two broadcast float4 LDS reads) and accumulates
//      acc[g][8 dims] += p[g] * v: no per-key exponentials or rescales
// and the 16 groups are merged through LDS at the end.
constexpr int GQ_CH = 512;

template <int HD, int G>
__global__ __launch_bounds__(256, (G <= 4 ? 3 : 2)) void attn_decode_gqa_kernel(const bf16_t* __restrict__ q, long ldq, const bf16_t* __restrict__ kc,
                                                              const bf16_t* __restrict__ vc, bf16_t* __restrict__ o, long ldo, int Hk,
                                                              int Tmax, int ctx_host, const int* __restrict__ ctx_dev, float scale,
                                                              const int* __restrict__ kv_start) {
    static_assert(HD == 128 && G >= 2 && G <= 8, "grouped decode: head_dim 128, 2..8 query heads per kv head");
    constexpr int EPL = 8;
    // scores [GQ_CH][8] fp32 (16 KB) during the chunks, reused as the merge buffer [4 waves][G][HD] fp32 at the end (the four
    // 16-lane groups of a wave are merged by shuffles first): ~30 KB per block keeps 5 blocks = 20 waves resident per CU -
    // this kernel lives on loads in flight (with the 57 KB version only 2 blocks fitted: 2.7 TB/s)
    constexpr int SMEM_F = (4 * G * HD > GQ_CH * 8) ? 4 * G * HD : GQ_CH * 8;
    __shared__ __attribute__((aligned(16))) float sbuf[SMEM_F];
    __shared__ float red[4][8];
    __shared__ float m_run[8], l_run[8], alpha_s[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = tid >> 4, sub = tid & 15;
    const int b = blockIdx.y, hk = blockIdx.x;
    const int ks0 = kv_start ? kv_start[b] : 0;
    const int ctx = ctx_host + (ctx_dev ? ctx_dev[0] : 0) - ks0;
    // A operand of the score MFMAs: row (l & 15) = query head (zero for rows >= G), k index 8 (l >> 4) + e <-> d = 32 ks + 8 (l >> 4) + e
    const int fr = lane & 15, fg = lane >> 4;
    bf16x8_t qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        u32x4 w = {0u, 0u, 0u, 0u};
        if (fr < G) w = *reinterpret_cast<const u32x4*>(q + (long)b * ldq + (long)(hk * G + fr) * HD + ks * 32 + fg * 8);
        qf[ks] = __builtin_bit_cast(bf16x8_t, w);
    }
    const bf16_t* kfr = kc + (((long)b * Hk + hk) * (long)Tmax + ks0) * HD + fg * 8;      // + key * HD + 32 ks
    const bf16_t* vb = vc + (((long)b * Hk + hk) * (long)Tmax + ks0) * HD + sub * EPL;
    float acc[G][EPL];
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < EPL; ++e) acc[g][e] = 0.f;
    if (tid < 8) { m_run[tid] = -1e30f; l_run[tid] = 0.f; }

    for (int c0 = 0; c0 < ctx; c0 += GQ_CH) {
        const int cn = min(GQ_CH, ctx - c0);
        __syncthreads();                                  // previous chunk's probabilities consumed, m_run/l_run visible
        // ---- A: scores.  Wave w takes the 16-key tiles w, w + 4, ... of the chunk, two tiles per trip with all eight 16-byte K loads of the
        // trip issued before the first MFMA (rows beyond the chunk are clamped and not stored)
        const int rounds = (cn + 15) >> 4;
        const int ntile = rounds;                               // 16-key tiles in this chunk
        constexpr int TPT = G <= 4 ? 2 : 4;                     // tiles per trip: 8 or 16 K loads (128 / 256 bytes) in flight per lane
        for (int t0 = wave; t0 < ntile; t0 += 4 * TPT) {
            u32x4 kw[TPT][4];
#pragma unroll
            for (int u = 0; u < TPT; ++u) {
                const int jc = min((t0 + 4 * u) * 16 + fr, cn - 1);
                const bf16_t* kr = kfr + (long)(c0 + jc) * HD;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) kw[u][ks] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(kr + ks * 32));
            }
#pragma unroll
            for (int u = 0; u < TPT; ++u) {
                if (t0 + 4 * u < ntile) {                       // wave-uniform
                    f32x4_t sc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks)
                        sc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qf[ks], __builtin_bit_cast(bf16x8_t, kw[u][ks]), sc, 0, 0, 0);
                    // D: lane (key = l & 15, heads 4 (l >> 4) + r): lane groups 0 / 1 hold heads 0-3 / 4-7 of their key
                    const int j = (t0 + 4 * u) * 16 + fr;
                    if (fg < 2 && j < cn) *reinterpret_cast<f32x4_t*>(sbuf + j * 8 + fg * 4) = sc * scale;
                }
            }
        }
        __syncthreads();
        // the first trip of V rows of phase C is requested HERE: it flies under phase B (four barriers, the exponentials, no memory traffic)
        constexpr int RPT = G <= 4 ? 4 : 6;                     // V rows per trip: 64 / 96 bytes in flight per lane (8 rows spill at G >= 7)
        u32x4 vw[RPT];
#pragma unroll
        for (int u = 0; u < RPT; ++u) {
            const int jc = min(u * 16 + grp, cn - 1);
            vw[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vb + (long)(c0 + jc) * HD));
        }
        // ---- B: per-head chunk max -> running max, p = exp(s - m) in place, running sum
        {
            const int g = tid & 7;
            float mx = -1e30f;
            for (int j = tid >> 3; j < cn; j += 32) mx = fmaxf(mx, sbuf[j * 8 + g]);
            mx = fmaxf(mx, row_xor8(mx)); mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            if (lane < 8) red[wave][lane] = mx;
            __syncthreads();
            const float m_old = m_run[g];
            const float m_new = fmaxf(fmaxf(fmaxf(red[0][g], red[1][g]), fmaxf(red[2][g], red[3][g])), m_old);
            float sum = 0.f;
            for (int j = tid >> 3; j < cn; j += 32) {
                const float pv = __expf(sbuf[j * 8 + g] - m_new);
                sbuf[j * 8 + g] = pv;
                sum += pv;
            }
            sum += row_xor8(sum); sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
            __syncthreads();                                             // red[] max values consumed
            if (lane < 8) red[wave][lane] = sum;
            __syncthreads();
            if (tid < 8) {
                const float a = __expf(m_old - m_new);
                alpha_s[tid] = a;
                l_run[tid] = l_run[tid] * a + red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
                m_run[tid] = m_new;
            }
            __syncthreads();
        }
        // ---- C: rescale (once per chunk) and accumulate P.V
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const float a = alpha_s[g];
#pragma unroll
            for (int e = 0; e < EPL; ++e) acc[g][e] *= a;
        }
        for (int r0 = 0; r0 < rounds; r0 += RPT) {
            if (r0 > 0) {
#pragma unroll
                for (int u = 0; u < RPT; ++u) {
                    const int jc = min((r0 + u) * 16 + grp, cn - 1);
                    vw[u] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(vb + (long)(c0 + jc) * HD));
                }
            }
#pragma unroll
            for (int u = 0; u < RPT; ++u) {
                // the row's 8 probabilities are read right before use (two broadcast float4 LDS reads)
                const int j = (r0 + u) * 16 + grp, jc = min(j, cn - 1);
                f32x4_t p0 = *reinterpret_cast<const f32x4_t*>(sbuf + jc * 8);
                f32x4_t p1 = *reinterpret_cast<const f32x4_t*>(sbuf + jc * 8 + 4);
                if (j >= cn) { p0 = f32x4_t{0.f, 0.f, 0.f, 0.f}; p1 = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
                float vx[EPL];
#pragma unroll
                for (int e = 0; e < 4; ++e) { vx[2 * e] = lo_bf(vw[u][e]); vx[2 * e + 1] = hi_bf(vw[u][e]); }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float pg = g < 4 ? p0[g & 3] : p1[g & 3];
#pragma unroll
                    for (int e = 0; e < EPL; ++e) acc[g][e] += pg * vx[e];
                }
            }
        }
    }
    __syncthreads();
    // ---- merge the 16 key groups (all share the running max): the 4 groups of a wave by shuffles, the 4 waves through LDS
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
        for (int e = 0; e < EPL; ++e) {
            float v = acc[g][e];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            acc[g][e] = v;
        }
    if (lane < 16) {
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int e = 0; e < EPL; ++e) sbuf[(wave * G + g) * HD + sub * EPL + e] = acc[g][e];
    }
    __syncthreads();
    for (int idx = tid; idx < G * HD; idx += 256) {
        const float O = (sbuf[idx] + sbuf[G * HD + idx]) + (sbuf[2 * G * HD + idx] + sbuf[3 * G * HD + idx]);
        const int g = idx / HD;
        o[(long)b * ldo + (long)(hk * G) * HD + idx] = f2bf(O / l_run[g]);
    }
}

}  // namespace

extern "C" int crab_attn_fwd(crab_ctx* ctx, void* stream, const crab_attn_desc* d) {
    if (!ctx) return CRAB_E_INVALID;
    if (!d || !d->q || !d->k || !d->vt || !d->o) return crab_fail(ctx, CRAB_E_INVALID, "attn_fwd: null operand");
    if (d->B <= 0 || d->H <= 0 || d->Hk <= 0 || d->H % d->Hk || d->Sq <= 0 || d->Skv <= 0)
        return crab_fail(ctx, CRAB_E_INVALID, "attn_fwd: bad shape");
    if (d->head_dim != 32 && d->head_dim != 64 && d->head_dim != 128) return crab_fail(ctx, CRAB_E_UNSUPPORTED, "attn_fwd: head_dim must be 32, 64 or 128");
    if (d->head_dim == 32 && (d->causal || d->bias)) return crab_fail(ctx, CRAB_E_UNSUPPORTED, "attn_fwd: head_dim 32 only without mask/bias");
    if ((d->q_ss & 7) || (d->q_hs & 7) || (d->q_bs & 7) || (d->k_ss & 7) || (d->k_hs & 7) || (d->k_bs & 7) || (d->vt_ds & 7) ||
        (d->vt_hs & 7) || (d->vt_bs & 7) || (d->o_ss & 3) || (d->o_bs & 3))
        return crab_fail(ctx, CRAB_E_INVALID, "attn_fwd: strides must keep 16-byte alignment");
    if (((uintptr_t)d->q & 15) || ((uintptr_t)d->k & 15) || ((uintptr_t)d->vt & 15) || ((uintptr_t)d->o & 7))
        return crab_fail(ctx, CRAB_E_INVALID, "attn_fwd: pointer alignment");
    if (d->causal && d->Skv < d->Sq) return crab_fail(ctx, CRAB_E_INVALID, "attn_fwd: causal needs Skv >= Sq");
    if ((d->bias == nullptr) && d->gate) return crab_fail(ctx, CRAB_E_INVALID, "attn_fwd: gate without bias");
    if (d->key_mask && (d->key_mask_ld < (d->Skv + 31) / 32 || d->head_dim == 32))
        return crab_fail(ctx, CRAB_E_INVALID, "attn_fwd: key_mask needs key_mask_ld >= ceil(Skv / 32) words per sequence (head_dim 64 / 128)");
    AttnP p;
    p.q = (const bf16_t*)d->q; p.k = (const bf16_t*)d->k; p.vt = (const bf16_t*)d->vt; p.o = (bf16_t*)d->o;
    p.q_bs = d->q_bs; p.q_hs = d->q_hs; p.q_ss = d->q_ss; p.k_bs = d->k_bs; p.k_hs = d->k_hs; p.k_ss = d->k_ss;
    p.vt_bs = d->vt_bs; p.vt_hs = d->vt_hs; p.vt_ds = d->vt_ds; p.o_bs = d->o_bs; p.o_ss = d->o_ss;
    p.bias = d->bias; p.gate = d->gate; p.kv_start = d->kv_start; p.key_mask = d->key_mask; p.key_mask_ld = d->key_mask_ld; p.B = d->B; p.H = d->H; p.Hk = d->Hk; p.Sq = d->Sq; p.Skv = d->Skv;
    p.causal = d->causal; p.scale = d->scale;
    dim3 grid(((d->Sq + 63) / 64) * d->H * d->B), block(256);
    hipStream_t s = (hipStream_t)stream;
    const bool hb = d->bias != nullptr;
    if (d->causal && hb) return crab_fail(ctx, CRAB_E_UNSUPPORTED, "attn_fwd: causal + bias not instantiated");
    // Sq > 64 without bias (decoder prefill, CLIP): 128 query rows per block on the 32x32x16 MFMA.  CRAB_ATTN_FWD32=0 keeps the 64-row kernel (A/B runs)
    static const int fwd32 = []() { const char* e = getenv("CRAB_ATTN_FWD32"); return !(e && e[0] == '0'); }();
    if (fwd32 && !hb && d->Sq > 64 && d->scale > 0.f && (d->head_dim == 128 || d->head_dim == 64)) {
        dim3 grid32(((d->Sq + 127) / 128) * d->H * d->B);
        if (d->head_dim == 128) {
            if (d->causal) hipLaunchKernelGGL((attn_fwd32_kernel<128, true>), grid32, block, 0, s, p);
            else hipLaunchKernelGGL((attn_fwd32_kernel<128, false>), grid32, block, 0, s, p);
        } else {
            if (d->causal) hipLaunchKernelGGL((attn_fwd32_kernel<64, true>), grid32, block, 0, s, p);
            else hipLaunchKernelGGL((attn_fwd32_kernel<64, false>), grid32, block, 0, s, p);
        }
        return crab_check_launch(ctx, d->head_dim == 128 ? (d->causal ? "attn_fwd32_kernel<128,causal>" : "attn_fwd32_kernel<128>")
                                                         : (d->causal ? "attn_fwd32_kernel<64,causal>" : "attn_fwd32_kernel<64>"));
    }
    if (d->head_dim == 32) {
        hipLaunchKernelGGL((attn_fwd_kernel<32, false, false>), grid, block, 0, s, p);
    } else if (d->head_dim == 128) {
        if (d->causal) hipLaunchKernelGGL((attn_fwd_kernel<128, true, false>), grid, block, 0, s, p);
        else if (hb) hipLaunchKernelGGL((attn_fwd_kernel<128, false, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((attn_fwd_kernel<128, false, false>), grid, block, 0, s, p);
    } else {
        if (d->causal) hipLaunchKernelGGL((attn_fwd_kernel<64, true, false>), grid, block, 0, s, p);
        else if (hb) hipLaunchKernelGGL((attn_fwd_kernel<64, false, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((attn_fwd_kernel<64, false, false>), grid, block, 0, s, p);
    }
    return crab_check_launch(ctx, d->head_dim == 32 ? "attn_fwd_kernel<32>" : d->head_dim == 128 ? (d->causal ? "attn_fwd_kernel<128,causal>" : hb ? "attn_fwd_kernel<128,bias>" : "attn_fwd_kernel<128>")
                                                                            : (d->causal ? "attn_fwd_kernel<64,causal>" : hb ? "attn_fwd_kernel<64,bias>" : "attn_fwd_kernel<64>"));
}

extern "C" int64_t crab_attn_decode_rope_workspace(int B, int H, int d) {
    // up to 8 splits of (d + 2) fp32 per (b, h) + one counter per (b, h); the counters (at the END) must be zero at the first call
    return ((int64_t)B * H * 8 * (d + 2) * 4 + 255) / 256 * 256 + (int64_t)B * H * 4;
}

extern "C" int crab_attn_decode_rope(crab_ctx* ctx, void* stream, const void* qkv, int64_t ldqkv, const float* rope_tab, void* k_cache,
                                     void* v_cache, void* o, int64_t ldo, int B, int H, int Hk, int d, int Tmax, int pos0,
                                     const int32_t* pos_dev, float scale, void* workspace, int64_t workspace_bytes) {
    if (!ctx) return CRAB_E_INVALID;
    if (!qkv || !rope_tab || !k_cache || !v_cache || !o || B <= 0 || H <= 0 || Hk <= 0 || H % Hk) return crab_fail(ctx, CRAB_E_INVALID, "attn_decode_rope: bad argument");
    if (d != 64 && d != 128) return crab_fail(ctx, CRAB_E_UNSUPPORTED, "attn_decode_rope: head_dim must be 64 or 128");
    if ((ldqkv & 7) || ((uintptr_t)qkv & 15) || ((uintptr_t)k_cache & 15) || ((uintptr_t)v_cache & 15)) return crab_fail(ctx, CRAB_E_INVALID, "attn_decode_rope: alignment");
    if (!pos_dev && (pos0 < 0 || pos0 >= Tmax)) return crab_fail(ctx, CRAB_E_INVALID, "attn_decode_rope: position outside the KV cache");
    // split while (b, h) alone leaves CUs idle: up to 8 splits, ~two blocks per CU (r03, after every load of the kernel moved into its first
    // memory round trip: a block keeps 64 KB of K / V in flight, two of them saturate a CU's queue).  Decode step at 2 / 3 / 4 / 8 clips with a
    // 256- vs 512-block target: 3.865 / 4.015 / 4.092 / 4.514 vs 3.820 / 3.948 / 3.994 / 4.482 ms (768: slower again); one clip has 8 splits
    // either way.  CRAB_ATTN_BLOCKS overrides the target (A/B runs).
    int nsplit = 1;
    static const int tgt = []() { const char* e = getenv("CRAB_ATTN_BLOCKS"); return e ? atoi(e) : 512; }();
    if ((long)B * H < tgt) { nsplit = (int)(tgt / ((long)B * H)); if (nsplit > 8) nsplit = 8; }
    if (nsplit > 1 && (!workspace || workspace_bytes < crab_attn_decode_rope_workspace(B, H, d)))
        return crab_fail(ctx, CRAB_E_WORKSPACE, "attn_decode_rope: needs crab_attn_decode_rope_workspace(B, H, d) bytes (counters zeroed once)");
    float* part = (float*)workspace;
    unsigned* counters = workspace ? (unsigned*)((char*)workspace + ((int64_t)B * H * 8 * (d + 2) * 4 + 255) / 256 * 256) : nullptr;
    hipStream_t s = (hipStream_t)stream;
    // TWO splits (half of the chip's worth of (b, h) pairs: 5 .. 8 clips of 32 heads - the reference's eval batch) run as ONE block of 32 key
    // groups per pair instead: the same rows in flight per CU, merged through LDS - no partial stores, arrival ticket and reload by the last of
    // two blocks (r04: decode step at batch 8 4.36-4.39 -> 4.31-4.32 ms).  CRAB_ATTN_WIDE=0 keeps the two-split form (A/B runs).
    static const int wide_on = []() { const char* e = getenv("CRAB_ATTN_WIDE"); return !(e && e[0] == '0'); }();
    if (wide_on && nsplit == 2 && d == 128) {
        hipLaunchKernelGGL((attn_decode_rope_kernel<128, 32>), dim3(H, B, 1), dim3(512), 0, s, (const bf16_t*)qkv, (long)ldqkv, rope_tab, (bf16_t*)k_cache,
                           (bf16_t*)v_cache, (bf16_t*)o, (long)ldo, H, Hk, Tmax, pos0, pos_dev, scale, (float*)nullptr, (unsigned*)nullptr);
        return crab_check_launch(ctx, "attn_decode_rope_kernel<128,32>");
    }
    dim3 grid(H, B, nsplit), block(256);
    if (d == 128)
        hipLaunchKernelGGL((attn_decode_rope_kernel<128>), grid, block, 0, s, (const bf16_t*)qkv, (long)ldqkv, rope_tab, (bf16_t*)k_cache,
                           (bf16_t*)v_cache, (bf16_t*)o, (long)ldo, H, Hk, Tmax, pos0, pos_dev, scale, part, counters);
    else
        hipLaunchKernelGGL((attn_decode_rope_kernel<64>), grid, block, 0, s, (const bf16_t*)qkv, (long)ldqkv, rope_tab, (bf16_t*)k_cache,
                           (bf16_t*)v_cache, (bf16_t*)o, (long)ldo, H, Hk, Tmax, pos0, pos_dev, scale, part, counters);
    return crab_check_launch(ctx, d == 128 ? "attn_decode_rope_kernel<128>" : "attn_decode_rope_kernel<64>");
}

extern "C" int crab_attn_decode_keymask(crab_ctx* ctx, void* stream, const void* q, int64_t ldq, const void* k_cache, const void* v_cache,
                                        void* o, int64_t ldo, int B, int H, int Hk, int d, int Tmax, int ctx_len_host, const int32_t* ctx_dev,
                                        float scale, const uint32_t* key_mask, int64_t key_mask_ld) {
    if (!ctx) return CRAB_E_INVALID;
    if (!q || !k_cache || !v_cache || !o || !key_mask || B <= 0 || H <= 0 || Hk <= 0 || H % Hk) return crab_fail(ctx, CRAB_E_INVALID, "attn_decode_keymask: bad argument");
    if (d != 64 && d != 128) return crab_fail(ctx, CRAB_E_UNSUPPORTED, "attn_decode_keymask: head_dim must be 64 or 128");
    if (!ctx_dev && (ctx_len_host <= 0 || ctx_len_host > Tmax)) return crab_fail(ctx, CRAB_E_INVALID, "attn_decode_keymask: ctx_len out of range");
    if (key_mask_ld < ((ctx_dev ? Tmax : ctx_len_host) + 31) / 32)
        return crab_fail(ctx, CRAB_E_INVALID, "attn_decode_keymask: key_mask_ld must cover the visible context (ceil(ctx_len / 32) words, Tmax with ctx_dev)");
    dim3 grid(H, B), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (d == 128)
        hipLaunchKernelGGL((attn_decode_keymask_kernel<128>), grid, block, 0, s, (const bf16_t*)q, (long)ldq, (const bf16_t*)k_cache, (const bf16_t*)v_cache,
                           (bf16_t*)o, (long)ldo, H, Hk, Tmax, ctx_len_host, ctx_dev, scale, key_mask, (long)key_mask_ld);
    else
        hipLaunchKernelGGL((attn_decode_keymask_kernel<64>), grid, block, 0, s, (const bf16_t*)q, (long)ldq, (const bf16_t*)k_cache, (const bf16_t*)v_cache,
                           (bf16_t*)o, (long)ldo, H, Hk, Tmax, ctx_len_host, ctx_dev, scale, key_mask, (long)key_mask_ld);
    return crab_check_launch(ctx, "attn_decode_keymask");
}

extern "C" int crab_attn_decode_masked(crab_ctx* ctx, void* stream, const void* q, int64_t ldq, const void* k_cache, const void* v_cache,
                                       void* o, int64_t ldo, int B, int H, int Hk, int d, int Tmax, int ctx_len_host, const int32_t* ctx_dev,
                                       float scale, const int32_t* kv_start) {
    if (!ctx) return CRAB_E_INVALID;
    if (!q || !k_cache || !v_cache || !o || B <= 0 || H <= 0 || Hk <= 0 || H % Hk) return crab_fail(ctx, CRAB_E_INVALID, "attn_decode: bad argument");
    if (d != 64 && d != 128) return crab_fail(ctx, CRAB_E_UNSUPPORTED, "attn_decode: head_dim must be 64 or 128");
    if (!ctx_dev && (ctx_len_host <= 0 || ctx_len_host > Tmax)) return crab_fail(ctx, CRAB_E_INVALID, "attn_decode: ctx_len out of range");
    dim3 grid(H, B), block(256);
    hipStream_t s = (hipStream_t)stream;
    // grouped-query heads: read each K/V row once per kv head when there are enough (b, kv head) blocks to fill the chip
    const int G = H / Hk;
    if (d == 128 && (G == 2 || G == 4 || G == 7 || G == 8) && (long)B * Hk >= 256 && (ldq & 7) == 0 && ((uintptr_t)q & 15) == 0) {
        dim3 gg(Hk, B);
#define CRAB_GQA(G_)                                                                                                             \
    hipLaunchKernelGGL((attn_decode_gqa_kernel<128, G_>), gg, block, 0, s, (const bf16_t*)q, (long)ldq, (const bf16_t*)k_cache,   \
                       (const bf16_t*)v_cache, (bf16_t*)o, (long)ldo, Hk, Tmax, ctx_len_host, ctx_dev, scale, kv_start)
        if (G == 2) CRAB_GQA(2); else if (G == 4) CRAB_GQA(4); else if (G == 7) CRAB_GQA(7); else CRAB_GQA(8);
#undef CRAB_GQA
        return crab_check_launch(ctx, G == 2 ? "attn_decode_gqa_kernel<128,2>" : G == 4 ? "attn_decode_gqa_kernel<128,4>" : G == 7 ? "attn_decode_gqa_kernel<128,7>" : "attn_decode_gqa_kernel<128,8>");
    }
    if (d == 128)
        hipLaunchKernelGGL((attn_decode_kernel<128>), grid, block, 0, s, (const bf16_t*)q, (long)ldq, (const bf16_t*)k_cache,
                           (const bf16_t*)v_cache, (bf16_t*)o, (long)ldo, H, Hk, Tmax, ctx_len_host, ctx_dev, scale, kv_start);
    else
        hipLaunchKernelGGL((attn_decode_kernel<64>), grid, block, 0, s, (const bf16_t*)q, (long)ldq, (const bf16_t*)k_cache,
                           (const bf16_t*)v_cache, (bf16_t*)o, (long)ldo, H, Hk, Tmax, ctx_len_host, ctx_dev, scale, kv_start);
    return crab_check_launch(ctx, d == 128 ? "attn_decode_kernel<128>" : "attn_decode_kernel<64>");
}

extern "C" int crab_attn_decode(crab_ctx* ctx, void* stream, const void* q, int64_t ldq, const void* k_cache, const void* v_cache,
                                void* o, int64_t ldo, int B, int H, int Hk, int d, int Tmax, int ctx_len_host, const int32_t* ctx_dev,
                                float scale) {
    return crab_attn_decode_masked(ctx, stream, q, ldq, k_cache, v_cache, o, ldo, B, H, Hk, d, Tmax, ctx_len_host, ctx_dev, scale, nullptr);
}

// Weight-streaming kernels for the decode regime (M = clips in flight, <= 128 rows) and for the hyper-LoRA
// router, where the work is HBM-bound on the weight matrix and a tiled GEMM grid cannot fill 256 CUs.
//
// gemm_skinny_dma_kernel (M <= 16, the reference's batch sizes 1 and 8):  C[M,N] = res_scale*R + act(A.B^T + A2.B2^T + bias).
//   One block per 16 weight rows, 8 waves splitting K; each wave streams its K slice HBM -> LDS by LDS-DMA in whole cache lines
//   through a private 4-slot ring and reads it back as MFMA fragments (described at the kernel).  5-6.3 TB/s on the wide shapes.
// gemm_skinny_kernel<MT, NT> (16 < M <= 128 without a split-K workspace; M <= 16 only under tune 1 / 2 / 4 for A/B runs):
//   One block per 16*NT weight rows (N/16 blocks: 256 for N=4096, i.e. one per CU, 768/1376/2001 for the wider
//   projections), 8 waves per block splitting K between them.  Each wave streams its K-slice of the 16 weight
//   rows straight into MFMA A-fragments (global_load_dwordx4, no LDS: every weight byte is used exactly once) and
//   the matching activation columns into B-fragments (L2-resident, M*K*2 bytes), accumulating 16 x 16*MT fp32.
//   The 8 partial tiles are reduced through LDS and the epilogue (bias, activation, residual, bf16/fp32 store)
//   runs once.  No split-K partials ever travel through HBM and the summation order is fixed (deterministic).
//
// lora_t_partial_kernel + lora_mix_reduce_kernel: the hyper-LoRA router
//   T = x.[R;A]^T (N <= 48) is far too skinny for either GEMM grid; it is split over K into `nslices` partial
//   products (grid nslices x M/64), and the second kernel sums the slices in a fixed order, applies the fp32
//   softmax over the route logits and writes U = scaling * p_i * h_j in bf16 (peft_hyper/tuners/lora.py:346-350).
// lora_route_row_kernel: the same router as ONE row-owning launch, used for 64 < M <= 256 (one row per clip at decode).
#include "common.h"
#include "crab_internal.h"
#include <stdlib.h>

namespace {

struct SkinnyP {
    const bf16_t* A; const bf16_t* B; void* C; const bf16_t* bias; const bf16_t* R;
    const bf16_t* A2; const bf16_t* B2;
    long lda, ldb, ldc, ldr, lda2, ldb2;
    int M, N, K, K2, act, c_fp32;
    float res_scale;
    // gemm_skinny_dma_kernel only: 16 extra weight rows Bx [16][K] behind the N rows of B (the projection's own hyper-LoRA [R;A]):
    // SKX blocks past the last block of B each take one K range of them and store a partial product Tx[e][m][0..16) = A . Bx^T over
    // that range (raw fp32 sums; the consumer adds the SKX partials in order) - the router product rides on the projection's launch
    // instead of two launches of its own.  K is split because a 257th block as long as the others shares a CU with one of them and
    // both then stream at half rate: o / down measured 13.3 -> 17.5 us with ONE extra full-K block.
    const bf16_t* Bx; long ldbx; float* Tx;
    // gemm_skinny_dma_kernel only, packed q|k|v projection of ONE row per sequence (decode): RoPE of q / k and the KV-cache append in the
    // epilogue (the result of crab_gemm_bf16 followed by crab_qkv_rope_split(B = M, S = 1), bit for bit).  A rotation pair is (dim i,
    // dim i + d/2) of one head, 64 weight rows apart: a block therefore takes rows [8j, 8j+8) and [d/2 + 8j, d/2 + 8j + 8) of a q / k head
    // (v heads keep 16 consecutive rows), so both partners of every pair sit in its 16-column tile, two lane groups apart.
    const float* rope_tab; bf16_t* rope_kc; bf16_t* rope_vc; const int* rope_pos_dev;
    int rope_H, rope_Hk, rope_d, rope_Tmax, rope_pos0;
    const int* rope_row_off;                 // ragged decode batch: row m rotates at slot - rope_row_off[m] (crab_gemm_desc.rope_row_off)
};

constexpr int SK_WAVES = 8;
constexpr int SKX = 8;                           // extra blocks (K ranges) of the ride-along router rows

// MT: 16-row activation tiles (M <= 16*MT).  NT: 16-row weight tiles per block (block covers 16*NT rows of W).
// Every wave covers all NT weight tiles and all MT activation tiles over its own K slice, so per k-step it issues
// NT weight loads (HBM) + MT activation loads (L2) for NT*MT MFMAs: NT trades grid size (N/(16 NT) blocks) against
// L2 request pressure of the replicated activation reads (MT/NT activation bytes per weight byte).
template <int MT, int NT>
__global__ __launch_bounds__(SK_WAVES * 64) void gemm_skinny_kernel(SkinnyP p) {
    constexpr int U = (NT + MT) <= 3 ? 4 : ((NT + MT) <= 6 ? 2 : 1);       // k-steps in flight per wave
    __shared__ __attribute__((aligned(16))) float red[SK_WAVES][NT * MT][64][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int n0 = blockIdx.x * 16 * NT;

    const int ks1 = (p.K + 31) >> 5;
    const int ks2 = p.A2 ? (p.K2 + 31) >> 5 : 0;
    const int ks = ks1 + ks2;
    const int s_begin = (int)((long)ks * wave / SK_WAVES), s_end = (int)((long)ks * (wave + 1) / SK_WAVES);
    const int nst = s_end - s_begin;
    // rotate the k-step order per block: concurrent blocks then touch different activation lines at any instant
    const int rot = nst > 0 ? (int)(blockIdx.x % (unsigned)nst) : 0;

    f32x4_t acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    long woff[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) woff[j] = min(n0 + j * 16 + fr, p.N - 1);      // clamped: rows >= N are never stored
    int mrow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) mrow[i] = min(i * 16 + fr, p.M - 1);

    const u32x4 z4 = {0u, 0u, 0u, 0u};
    for (int s = 0; s < nst; s += U) {
        u32x4 wv[U][NT], xv[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = s + u < nst;
            int q = s + u + rot;
            if (q >= nst) q -= nst;
            const int st = live ? s_begin + q : s_begin;
            const bool seg2 = st >= ks1;
            const bf16_t* Bp = seg2 ? p.B2 : p.B;
            const bf16_t* Ap = seg2 ? p.A2 : p.A;
            const long lb = seg2 ? p.ldb2 : p.ldb, la = seg2 ? p.lda2 : p.lda;
            const int Kseg = seg2 ? p.K2 : p.K;
            const int k = ((seg2 ? st - ks1 : st) << 5) + fg * 8;
            const bool ok = live && k < Kseg;
            const int kc = ok ? k : 0;
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                u32x4 w = *reinterpret_cast<const u32x4*>(Bp + woff[j] * lb + kc);
                wv[u][j] = ok ? w : z4;
            }
#pragma unroll
            for (int i = 0; i < MT; ++i) xv[u][i] = *reinterpret_cast<const u32x4*>(Ap + (long)mrow[i] * la + kc);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                union { u32x4 r; bf16x8_t f; } wf;
                wf.r = wv[u][j];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    union { u32x4 r; bf16x8_t f; } xf;
                    xf.r = xv[u][i];
                    acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf.f, xf.f, acc[j][i], 0, 0, 0);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) *reinterpret_cast<f32x4_t*>(&red[wave][j * MT + i][lane][0]) = acc[j][i];
    __syncthreads();

    // reduce over waves in fixed order + epilogue: (tile t = j*MT+i, lane l): row m = 16i + (l&15), cols n0+16j+4*(l>>4)+r
    for (int idx = tid; idx < NT * MT * 64; idx += SK_WAVES * 64) {
        const int t = idx >> 6, l = idx & 63;
        const int j = t / MT, i = t % MT;
        const int m = i * 16 + (l & 15);
        const int n = n0 + j * 16 + (l >> 4) * 4;
        if (m >= p.M || n >= p.N) continue;
        f32x4_t v = *reinterpret_cast<const f32x4_t*>(&red[0][t][l][0]);
#pragma unroll
        for (int w = 1; w < SK_WAVES; ++w) v += *reinterpret_cast<const f32x4_t*>(&red[w][t][l][0]);
        if (p.act == ACT_SWIGLU_PAIR) {                 // interleaved (gate, up) columns -> two outputs at column n/2
            float t[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = v[r] + (p.bias ? bf2f(p.bias[n + r]) : 0.f);
            const float o0 = t[0] / (1.0f + __expf(-t[0])) * t[1], o1 = t[2] / (1.0f + __expf(-t[2])) * t[3];
            const long oc = (long)m * p.ldc + (n >> 1);
            if (p.c_fp32 & CF_C32) { reinterpret_cast<float*>(p.C)[oc] = o0; reinterpret_cast<float*>(p.C)[oc + 1] = o1; }
            else { reinterpret_cast<bf16_t*>(p.C)[oc] = f2bf(o0); reinterpret_cast<bf16_t*>(p.C)[oc + 1] = f2bf(o1); }
            continue;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (n + r >= p.N) break;
            float x = v[r];
            if (p.bias) x += bf2f(p.bias[n + r]);
            x = apply_act(x, p.act);
            if (p.R) x += p.res_scale * ld_res(p.R, (long)m * p.ldr + n + r, p.c_fp32);
            if (p.c_fp32 & CF_C32) reinterpret_cast<float*>(p.C)[(long)m * p.ldc + n + r] = x;
            else reinterpret_cast<bf16_t*>(p.C)[(long)m * p.ldc + n + r] = f2bf(x);
        }
    }
}

// ---------------------------------------------------------------------------------------------- M <= 16: weights through LDS
// gemm_skinny_dma_kernel: the same decomposition as gemm_skinny_kernel<1, 1> (one block per 16 weight rows, 8 waves splitting K,
// partial tiles reduced through LDS in wave order), but the weight stream goes HBM -> LDS by LDS-DMA in WHOLE cache lines
// (1-KiB pieces of 8 rows x 128 B, non-temporal) into a wave-private ring of NS 64-wide K slots and is read back as MFMA fragments
// (ds_read_b128, XOR swizzle, no conflicts).  The register-direct loads of the older kernel are fragment shaped - 16 rows x 64 B,
// half a line per row per instruction - and only 4 of them are in flight per wave: at M = 8 it streams the decoder projections at
// 2.3-3.5 TB/s (profiles/README.md).  Here a wave keeps NS x 2 KiB in flight with no register cost, and nothing in the K loop
// synchronises waves (each wave reads only the ring it fills).  The activation fragments (16 rows x 64 B per k step, L2 / TCP hits)
// still go straight to registers, issued together with the DMA of their slot so that one counted vmcnt covers both.
__device__ __attribute__((aligned(16))) uint32_t g_zero_page_sk[64];       // zero-initialised device memory (256 B)

typedef __attribute__((address_space(3))) void* sk_lds_vptr;
typedef const __attribute__((address_space(1))) void* sk_gbl_vptr;

template <int NT, int NS>
__global__ __launch_bounds__(SK_WAVES * 64) void gemm_skinny_dma_kernel(SkinnyP p, float* __restrict__ part) {
    constexpr int SLOT = NT * 16 * 64;                                  // elements: NT x 16 weight rows x 64 k
    __shared__ __attribute__((aligned(16))) bf16_t ring[SK_WAVES][NS][SLOT];
    __shared__ __attribute__((aligned(16))) float red[SK_WAVES][NT][64][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int nbN = (p.N + 16 * NT - 1) / (16 * NT);
    const bool extra = (int)blockIdx.x >= nbN;                          // block-uniform: this block owns a K range of Bx, not rows of B
    const int xe = (int)blockIdx.x - nbN;                               // which K range
    const bf16_t* Bw = extra ? p.Bx : p.B;
    const long ldbw = extra ? p.ldbx : p.ldb;
    const int Nw = extra ? 16 : p.N;
    const int n0 = extra ? 0 : (int)blockIdx.x * 16 * NT;
    // fused RoPE: block -> (head hh, 16-row group j); q / k heads take the two 8-row halves of rotation pairs (see SkinnyP)
    const bool rope = NT == 1 && p.rope_tab != nullptr && !extra;
    const int r_bpd = rope ? p.rope_d >> 4 : 1;
    const int r_hh = (int)blockIdx.x / r_bpd, r_j = (int)blockIdx.x % r_bpd;
    const bool r_qk = rope && r_hh < p.rope_H + p.rope_Hk;
    const int r_base = r_hh * p.rope_d, r_half = p.rope_d >> 1;

    const int nk1 = (p.K + 63) >> 6;
    const int nk2 = (p.A2 && !extra) ? (p.K2 + 63) >> 6 : 0;
    const int kx0 = extra ? (int)((long)nk1 * xe / SKX) : 0;            // first K slot of this block
    const int ks = extra ? (int)((long)nk1 * (xe + 1) / SKX) - kx0 : nk1 + nk2;
    const int s_begin = kx0 + (int)((long)ks * wave / SK_WAVES), s_end = kx0 + (int)((long)ks * (wave + 1) / SK_WAVES);
    const int nst = s_end - s_begin;

    // staging coordinates of the 2 NT pieces of a slot: lane l -> row i*8 + (l >> 3), LDS chunk l & 7 (lane-linear, as LDS-DMA writes),
    // source chunk (l & 7) ^ ((row >> 1) & 7): the inverse of the swizzle the fragment reads apply
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_sk);
    long off1[2 * NT], off2[2 * NT];
    int kc[2 * NT];
#pragma unroll
    for (int i = 0; i < 2 * NT; ++i) {
        const int row = i * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((row >> 1) & 7);
        long wrow = min(n0 + row, Nw - 1);                              // clamped: rows >= N are never stored
        if (r_qk) wrow = r_base + (row >> 3) * r_half + 8 * r_j + (row & 7);
        kc[i] = c * 8;
        off1[i] = wrow * ldbw + c * 8;
        off2[i] = wrow * p.ldb2 + c * 8;
    }
    const long xrow1 = (long)min(fr, p.M - 1) * p.lda, xrow2 = (long)min(fr, p.M - 1) * p.lda2;
    bf16_t* myring = &ring[wave][0][0];
    // fragment element offsets inside a 16-row tile: row fr, source chunk ks*4 + fg -> LDS chunk (ks*4 + fg) ^ ((fr >> 1) & 7)
    const int fofs0 = fr * 64 + ((fg ^ ((fr >> 1) & 7)) << 3);
    const int fofs1 = fofs0 ^ 32;

    u32x4 xv[NS][2];
#define SKD_STAGE(ST_, U_)                                                                                \
    {                                                                                                     \
        const int st_ = (ST_);                                                                            \
        const bool s2_ = st_ >= nk1;                                                                      \
        const int k0_ = (s2_ ? st_ - nk1 : st_) << 6;                                                     \
        const int Ks_ = s2_ ? p.K2 : p.K;                                                                 \
        _Pragma("unroll") for (int i = 0; i < 2 * NT; ++i) {                                              \
            const bf16_t* src_ = (s2_ ? p.B2 + off2[i] : Bw + off1[i]) + k0_;                             \
            src_ = (k0_ + kc[i] < Ks_) ? src_ : zero;                                                     \
            __builtin_amdgcn_global_load_lds((sk_gbl_vptr)src_, (sk_lds_vptr)(myring + (U_) * SLOT + i * 512), 16, 0, 2);   \
        }                                                                                                 \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                   \
            const int k_ = k0_ + h * 32 + fg * 8;                                                         \
            const int kk_ = k_ < Ks_ ? k_ : 0;              /* the weight chunk is zero there: any finite activation does */   \
            xv[U_][h] = *reinterpret_cast<const u32x4*>((s2_ ? p.A2 + xrow2 : p.A + xrow1) + kk_);        \
        }                                                                                                 \
    }

    f32x4_t acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NS; ++u)
        if (u < nst) SKD_STAGE(s_begin + u, u)
    for (int s = 0; s < nst; s += NS) {
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            if (s + u < nst) {
                // slot s+u has landed once at most the NS-1 younger slots (2 NT + 2 vector-memory instructions each) are outstanding
                if (s + u + NS - 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * (2 * NT + 2)) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                union { u32x4 r; bf16x8_t f; } w0[NT], w1[NT], x0, x1;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    w0[j].r = *reinterpret_cast<const u32x4*>(myring + u * SLOT + j * 1024 + fofs0);
                    w1[j].r = *reinterpret_cast<const u32x4*>(myring + u * SLOT + j * 1024 + fofs1);
                }
                x0.r = xv[u][0]; x1.r = xv[u][1];
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w0[j].f, x0.f, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1[j].f, x1.f, acc[j], 0, 0, 0);
                }
                // the fragments are in registers (the MFMAs consumed them): the ring slot may be refilled
                asm volatile("" ::: "memory");
                if (s + u + NS < nst) SKD_STAGE(s_begin + s + u + NS, u)
            }
        }
    }
#undef SKD_STAGE
#pragma unroll
    for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4_t*>(&red[wave][j][lane][0]) = acc[j];
    __syncthreads();
    // reduce over waves in fixed order + epilogue: (tile j, lane l): row m = l & 15, cols n0 + 16 j + 4*(l >> 4) + r
    if (tid < NT * 64) {
        const int j = tid >> 6, l = tid & 63;
        const int m = l & 15, n = n0 + j * 16 + (l >> 4) * 4;
        if (rope) {
            // every lane of the wave takes part in the partner exchange, rows >= M included
            f32x4_t v = *reinterpret_cast<const f32x4_t*>(&red[0][0][l][0]);
#pragma unroll
            for (int w = 1; w < SK_WAVES; ++w) v += *reinterpret_cast<const f32x4_t*>(&red[w][0][l][0]);
            const int cg = l >> 4;
            const int d = p.rope_d, pos = p.rope_pos0 + (p.rope_pos_dev ? p.rope_pos_dev[0] : 0);
            // local columns cg*4 .. +3 of this block's tile -> column of the packed projection
            const int dim = r_qk ? (cg >> 1) * r_half + 8 * r_j + (cg & 1) * 4 : 16 * r_j + cg * 4;
            const int col = r_base + dim;
            float x[4], px[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                x[r] = bf2f(f2bf(v[r] + (p.bias ? bf2f(p.bias[col + r]) : 0.f)));      // what the projection stores: the rotation reads bf16
                px[r] = __shfl_xor(x[r], 32, 64);                                           // the lane two column groups away, same row
            }
            if (m >= p.M) return;
            bf16_t* crow = reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + col;
            const uint32_t raw0 = pack_bf2(x[0], x[1]), raw1 = pack_bf2(x[2], x[3]);
            if (!r_qk) {                                                 // value head: cache append, the projection row keeps the value
                const int hk = r_hh - p.rope_H - p.rope_Hk;
                *reinterpret_cast<u32x2*>(crow) = u32x2{raw0, raw1};
                *reinterpret_cast<u32x2*>(p.rope_vc + (((long)m * p.rope_Hk + hk) * p.rope_Tmax + pos) * d + dim) = u32x2{raw0, raw1};
                return;
            }
            const int idim = 8 * r_j + (cg & 1) * 4;                     // index of the rotation pair (first-half dim)
            const int rp = pos - (p.rope_row_off ? p.rope_row_off[m] : 0);   // rotary position (m < M here); the cache slot stays `pos`
            const float* cs = p.rope_tab + ((long)rp * r_half + idim) * 2;
            float o[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float c = cs[2 * r], sn = cs[2 * r + 1];
                o[r] = cg < 2 ? rope_lo(x[r], px[r], c, sn) : rope_hi(px[r], x[r], c, sn);
            }
            const u32x2 ow = {pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3])};
            if (r_hh < p.rope_H) {
                *reinterpret_cast<u32x2*>(crow) = ow;                    // q rotated in place of the projection row
            } else {
                const int hk = r_hh - p.rope_H;
                *reinterpret_cast<u32x2*>(crow) = u32x2{raw0, raw1};     // the row keeps the un-rotated key like the unfused pair
                *reinterpret_cast<u32x2*>(p.rope_kc + (((long)m * p.rope_Hk + hk) * p.rope_Tmax + pos) * d + dim) = ow;
            }
            return;
        }
        if (m >= p.M || n >= Nw) return;
        f32x4_t v = *reinterpret_cast<const f32x4_t*>(&red[0][j][l][0]);
#pragma unroll
        for (int w = 1; w < SK_WAVES; ++w) v += *reinterpret_cast<const f32x4_t*>(&red[w][j][l][0]);
        if (extra) {                                    // partial router product x . [R;A]^T over this block's K range, fp32 [SKX][16][16]
            *reinterpret_cast<f32x4_t*>(p.Tx + ((long)xe * 16 + m) * 16 + n) = v;
            return;
        }
        if (part) {                                     // fp32 act(sum + bias) + res_scale * R (unrounded) for the row-owning reduction
            float* o = part + (long)m * p.N + n;        // kernel of gemm.hip, slab layout [M][N]; never with the SwiGLU pair epilogue
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (n + r >= p.N) break;
                float x = v[r];
                if (p.bias) x += bf2f(p.bias[n + r]);
                x = apply_act(x, p.act);
                if (p.R) x += p.res_scale * ld_res(p.R, (long)m * p.ldr + n + r, p.c_fp32);
                o[r] = x;
            }
            return;
        }
        if (p.act == ACT_SWIGLU_PAIR) {                 // interleaved (gate, up) columns -> two outputs at column n/2
            float t[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) t[r] = v[r] + (p.bias ? bf2f(p.bias[n + r]) : 0.f);
            const float o0 = t[0] / (1.0f + __expf(-t[0])) * t[1], o1 = t[2] / (1.0f + __expf(-t[2])) * t[3];
            const long oc = (long)m * p.ldc + (n >> 1);
            if (p.c_fp32 & CF_C32) { reinterpret_cast<float*>(p.C)[oc] = o0; reinterpret_cast<float*>(p.C)[oc + 1] = o1; }
            else { reinterpret_cast<bf16_t*>(p.C)[oc] = f2bf(o0); reinterpret_cast<bf16_t*>(p.C)[oc + 1] = f2bf(o1); }
            return;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (n + r >= p.N) break;
            float x = v[r];
            if (p.bias) x += bf2f(p.bias[n + r]);
            x = apply_act(x, p.act);
            if (p.R) x += p.res_scale * ld_res(p.R, (long)m * p.ldr + n + r, p.c_fp32);
            if (p.c_fp32 & CF_C32) reinterpret_cast<float*>(p.C)[(long)m * p.ldc + n + r] = x;
            else reinterpret_cast<bf16_t*>(p.C)[(long)m * p.ldc + n + r] = f2bf(x);
        }
    }
}

// ---------------------------------------------------------------------------------------------- hyper-LoRA router
// grid (nslices, ceil(M/(64*MT))); 4 waves, wave w owns MT 16-row tiles (rows 16*(w*MT + i) ..) of the block's 64*MT rows; all
// waves share the K slice.  MT = 1 in the decode regime; MT = 4 for prefill-sized M: every [R;A] fragment a lane loads is then
// used for four row tiles, which cuts the dominant (L2) operand traffic of this skinny product by 4.
template <int NT, int MT>
__global__ __launch_bounds__(256) void lora_t_partial_kernel(const bf16_t* __restrict__ X, long ldx, const bf16_t* __restrict__ RA, long ldra,
                                                             float* __restrict__ part, int M, int K, int kslice) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int fr = lane & 15, fg = lane >> 4;
    const int sl = blockIdx.x;
    const int m0 = (blockIdx.y * 4 + wave) * 16 * MT;
    if (m0 >= M) return;
    int mrow[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) mrow[i] = min(m0 + i * 16 + fr, M - 1);
    const int k_begin = sl * kslice, k_end = min(K, k_begin + kslice);
    f32x4_t acc[NT][MT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int i = 0; i < MT; ++i) acc[j][i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const u32x4 z4 = {0u, 0u, 0u, 0u};
    union Frag { u32x4 r; bf16x8_t f; };
    // K step = 64: lane (fr, fg) loads the 32 contiguous bytes k = k0 + 16 fg .. +15 of its row (a wave instruction then
    // covers whole 128-byte lines) and feeds them to two MFMAs; both operands use the same k permutation, so the sum
    // is the plain dot product.  Three steps are in flight per wave (loads of step s+2 issued before the MFMAs of s).
    Frag xf[3][MT][2], wf[3][NT][2];
    // Full 64-wide steps are loaded WITHOUT the K-tail mask: a select on the loaded registers makes the compiler wait for the
    // loads right after issuing them (s_waitcnt + v_cndmask ahead of the previous step's MFMAs), which serialises the pipeline.
    // The ragged tail (K slice not a multiple of 64) is one masked, unpipelined step after the loop.
#define RT_LOAD(B_, K0_)                                                                                   \
    {                                                                                                      \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                    \
            const int k_ = (K0_) + fg * 16 + h * 8;                                                        \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                 \
                xf[B_][i][h].r = *reinterpret_cast<const u32x4*>(X + (long)mrow[i] * ldx + k_);            \
            _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                 \
                wf[B_][j][h].r = *reinterpret_cast<const u32x4*>(RA + (long)(j * 16 + fr) * ldra + k_);    \
        }                                                                                                  \
    }
#define RT_MMA(B_)                                                                                         \
    _Pragma("unroll") for (int h = 0; h < 2; ++h)                                                          \
        _Pragma("unroll") for (int j = 0; j < NT; ++j)                                                     \
            _Pragma("unroll") for (int i = 0; i < MT; ++i)                                                 \
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[B_][j][h].f, xf[B_][i][h].f, acc[j][i], 0, 0, 0);
    const int k_full = k_begin + ((k_end - k_begin) & ~63);       // end of the whole 64-wide steps
    int k0 = k_begin;
    if (k0 + 5 * 64 <= k_full) {
        // steady state: three steps per trip, every load unconditional (a branch around a load makes the wait-count pass
        // assume the worst at the join and wait for loads that were just issued)
        RT_LOAD(0, k0);
        RT_LOAD(1, k0 + 64);
        for (; k0 + 5 * 64 <= k_full; k0 += 192) {
            RT_LOAD(2, k0 + 128);
            RT_MMA(0);
            RT_LOAD(0, k0 + 192);
            RT_MMA(1);
            RT_LOAD(1, k0 + 256);
            RT_MMA(2);
        }
        RT_MMA(0);                                                // steps k0 and k0 + 64 are already loaded
        RT_MMA(1);
        k0 += 128;
    }
    for (; k0 < k_full; k0 += 64) {                               // at most four remaining whole steps
        RT_LOAD(0, k0);
        RT_MMA(0);
    }
    if (k_full < k_end) {                                         // masked tail step
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int k_ = k_full + fg * 16 + h * 8;
            const bool ok_ = k_ < k_end;
            const int kc_ = ok_ ? k_ : 0;
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                xf[0][i][h].r = *reinterpret_cast<const u32x4*>(X + (long)mrow[i] * ldx + kc_);
                if (!ok_) xf[0][i][h].r = z4;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) wf[0][j][h].r = *reinterpret_cast<const u32x4*>(RA + (long)(j * 16 + fr) * ldra + kc_);
        }
        RT_MMA(0);
    }
#undef RT_LOAD
#undef RT_MMA
    // D[row = t-col 4fg+r][col = m fr]; part layout [slice][m][NT*16]
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = m0 + i * 16 + fr;
        if (m < M) {
            float* o = part + ((long)sl * M + m) * (NT * 16);
#pragma unroll
            for (int j = 0; j < NT; ++j) *reinterpret_cast<f32x4_t*>(o + j * 16 + fg * 4) = acc[j][i];
        }
    }
}

// one block per 4 rows: phase 1 sums the K-slices (thread per (row, t-column), fixed order, coalesced over columns),
// phase 2 = softmax over the route logits and the rank-r mix, one thread per (row, projection)
__global__ __launch_bounds__(256) void lora_mix_reduce_kernel(const float* __restrict__ part, int nslices, int tcols, bf16_t* __restrict__ U,
                                                              long ldu, int M, int nproj, int nl, int r, int ucols, float scaling) {
    __shared__ float T[4][64];
    const int m0 = blockIdx.x * 4;
    const int tid = threadIdx.x;
    {
        const int row = tid >> 6, c = tid & 63;
        const int m = m0 + row;
        if (m < M && c < tcols) {
            float acc = 0.f;
            const float* q = part + (long)m * tcols + c;
            const long st = (long)M * tcols;
            int s = 0;
            for (; s + 8 <= nslices; s += 8) {                    // 8 independent loads in flight, summed in slice order
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = q[(s + i) * st];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc += v[i];
            }
            for (; s < nslices; ++s) acc += q[s * st];
            T[row][c] = acc;
        }
    }
    __syncthreads();
    const int per_row = nproj + 1;
    if (tid >= 4 * per_row) return;
    const int row = tid / per_row, pj = tid % per_row;
    const int m = m0 + row;
    if (m >= M) return;
    bf16_t* u = U + (long)m * ldu;
    const int used = nproj * nl * r;
    if (pj == nproj) {
        for (int c = used; c < ucols; ++c) u[c] = 0;
        return;
    }
    const float* t = &T[row][pj * (nl + r)];
    float e[8], mx = -INFINITY;
    for (int i = 0; i < nl; ++i) mx = fmaxf(mx, t[i]);
    float sum = 0.f;
    for (int i = 0; i < nl; ++i) { e[i] = expf(t[i] - mx); sum += e[i]; }
    const float inv = 1.0f / sum;
    for (int i = 0; i < nl; ++i)
        for (int j = 0; j < r; ++j) u[pj * nl * r + i * r + j] = f2bf(scaling * e[i] * inv * t[nl + j]);
}


// Row-owning router for the decode regime (64 < M <= 256, one row per clip): ONE launch instead of the partial-product + mix pair.
// One block per row keeps the row in registers (8 columns = 16 bytes per lane per chunk), walks the used rows of [R;A] with all
// of a trip's loads in flight, reduces the per-thread partials through LDS in a fixed order (deterministic), then one thread per
// projection applies the fp32 softmax over the route logits and writes u = scaling * p_i * (x A^T)_j in bf16
// (peft_hyper/tuners/lora.py:346-350).  Same arithmetic as the router fused into splitk_epilogue_norm_kernel (gemm.hip).
template <int MAXQ>                                            // K <= MAXQ * 2048
__global__ __launch_bounds__(256) void lora_route_row_kernel(const bf16_t* __restrict__ X, long ldx, const bf16_t* __restrict__ RA, long ldra, int K,
                                                             bf16_t* __restrict__ U, long ldu, int nproj, int nl, int r, int ucols, float scaling) {
    __shared__ float tp[48][257];
    __shared__ float tq[48][4];
    __shared__ float T[48];
    const int m = blockIdx.x, tid = threadIdx.x;
    float xv[MAXQ][8];
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int n = (tid + q * 256) * 8;
        u32x4 w = {0u, 0u, 0u, 0u};
        if (n < K) w = *reinterpret_cast<const u32x4*>(X + (long)m * ldx + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) { xv[q][2 * e] = lo_bf(w[e]); xv[q][2 * e + 1] = hi_bf(w[e]); }
    }
    const int rows = nproj * (nl + r);                         // <= 48
#define ROUTE_TRIPS(RPT_, NTRIPS_)                                                                        \
    for (int tr = 0; tr < (NTRIPS_); ++tr) {                                                              \
        const int c0 = tr * (RPT_);                                                                       \
        float p[RPT_];                                                                                    \
        _Pragma("unroll") for (int cc = 0; cc < (RPT_); ++cc) p[cc] = 0.f;                                \
        _Pragma("unroll") for (int q = 0; q < MAXQ; ++q) {                                                \
            const int n = (tid + q * 256) * 8;                                                            \
            if (n < K) {                                                                                  \
                u32x4 w[RPT_];                                                                            \
                _Pragma("unroll") for (int cc = 0; cc < (RPT_); ++cc)                                     \
                    w[cc] = c0 + cc < rows ? *reinterpret_cast<const u32x4*>(RA + (long)(c0 + cc) * ldra + n) : u32x4{0u, 0u, 0u, 0u};   \
                _Pragma("unroll") for (int cc = 0; cc < (RPT_); ++cc) {                                   \
                    float a = 0.f;                                                                        \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                         \
                        a += xv[q][2 * e] * lo_bf(w[cc][e]) + xv[q][2 * e + 1] * hi_bf(w[cc][e]);         \
                    p[cc] += a;                                                                           \
                }                                                                                         \
            }                                                                                             \
        }                                                                                                 \
        _Pragma("unroll") for (int cc = 0; cc < (RPT_); ++cc)                                             \
            if (c0 + cc < rows) tp[c0 + cc][tid] = p[cc];                                                 \
    }
    if (nl + r == 11) { ROUTE_TRIPS(11, nproj) }
    else { ROUTE_TRIPS(8, (rows + 7) / 8) }
#undef ROUTE_TRIPS
    __syncthreads();
    if (tid < rows * 4) {
        const int c = tid >> 2, qt = tid & 3;
        float a = 0.f;
        for (int i = qt * 64; i < qt * 64 + 64; ++i) a += tp[c][i];
        tq[c][qt] = a;
    }
    __syncthreads();
    if (tid < rows) T[tid] = (tq[tid][0] + tq[tid][1]) + (tq[tid][2] + tq[tid][3]);
    __syncthreads();
    if (tid > nproj) return;
    bf16_t* u = U + (long)m * ldu;
    const int used = nproj * nl * r;
    if (tid == nproj) {
        for (int c = used; c < ucols; ++c) u[c] = 0;
        return;
    }
    const float* t = &T[tid * (nl + r)];
    float e[8], mx = -INFINITY;
    for (int i = 0; i < nl; ++i) mx = fmaxf(mx, t[i]);
    float sum = 0.f;
    for (int i = 0; i < nl; ++i) { e[i] = expf(t[i] - mx); sum += e[i]; }
    const float inv = 1.0f / sum;
    for (int i = 0; i < nl; ++i)
        for (int j = 0; j < r; ++j) u[tid * nl * r + i * r + j] = f2bf(scaling * e[i] * inv * t[nl + j]);
}

// The same router for TWO rows per block (r05; 256 < M <= CRAB_DECODE_MAX_ROWS, K > 8192): what bounds the one-row kernel at these M is the L2 traffic of
// [R;A] - every block re-reads all of it (down group: 33 rows x 11008 x 2 B = 0.7 MB x 512 blocks = 370 MB per launch, ~20 us) - so every chunk of
// [R;A] a thread loads serves both rows of the block.  Per (row, router row) the arithmetic and its order are those of lora_route_row_kernel
// (per-thread partial over the thread's chunks in q order, 4 quarter sums of 64 threads in index order, (q0 + q1) + (q2 + q3)): bit-identical
// results.  The LDS reduction runs per trip (RPT router rows x 2 rows: 23 KB instead of 2 x 49 KB for whole rows).
template <int MAXQ>
__global__ __launch_bounds__(256) void lora_route_row2_kernel(const bf16_t* __restrict__ X, long ldx, const bf16_t* __restrict__ RA, long ldra, int M, int K,
                                                              bf16_t* __restrict__ U, long ldu, int nproj, int nl, int r, int ucols, float scaling) {
    constexpr int RMAX = 11;
    __shared__ float tp[2][RMAX][257];
    __shared__ float tq[2][RMAX][4];
    __shared__ float T[2][48];
    const int tid = threadIdx.x;
    const int m0 = blockIdx.x * 2;
    float xv[2][MAXQ][8];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int m = min(m0 + rr, M - 1);                     // (an odd M: the last block computes its only row twice and stores it once)
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int n = (tid + q * 256) * 8;
            u32x4 w = {0u, 0u, 0u, 0u};
            if (n < K) w = *reinterpret_cast<const u32x4*>(X + (long)m * ldx + n);
#pragma unroll
            for (int e = 0; e < 4; ++e) { xv[rr][q][2 * e] = lo_bf(w[e]); xv[rr][q][2 * e + 1] = hi_bf(w[e]); }
        }
    }
    const int rows = nproj * (nl + r);                         // <= 48
#define ROUTE2_TRIPS(RPT_, NTRIPS_)                                                                       \
    for (int tr = 0; tr < (NTRIPS_); ++tr) {                                                              \
        const int c0 = tr * (RPT_);                                                                       \
        float p[2][RPT_];                                                                                 \
        _Pragma("unroll") for (int cc = 0; cc < (RPT_); ++cc) { p[0][cc] = 0.f; p[1][cc] = 0.f; }         \
        _Pragma("unroll") for (int q = 0; q < MAXQ; ++q) {                                                \
            const int n = (tid + q * 256) * 8;                                                            \
            if (n < K) {                                                                                  \
                u32x4 w[RPT_];                                                                            \
                _Pragma("unroll") for (int cc = 0; cc < (RPT_); ++cc)                                     \
                    w[cc] = c0 + cc < rows ? *reinterpret_cast<const u32x4*>(RA + (long)(c0 + cc) * ldra + n) : u32x4{0u, 0u, 0u, 0u};   \
                _Pragma("unroll") for (int rr = 0; rr < 2; ++rr)                                          \
                _Pragma("unroll") for (int cc = 0; cc < (RPT_); ++cc) {                                   \
                    float a = 0.f;                                                                        \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                         \
                        a += xv[rr][q][2 * e] * lo_bf(w[cc][e]) + xv[rr][q][2 * e + 1] * hi_bf(w[cc][e]); \
                    p[rr][cc] += a;                                                                       \
                }                                                                                         \
            }                                                                                             \
        }                                                                                                 \
        _Pragma("unroll") for (int rr = 0; rr < 2; ++rr)                                                  \
        _Pragma("unroll") for (int cc = 0; cc < (RPT_); ++cc) tp[rr][cc][tid] = p[rr][cc];                \
        __syncthreads();                                                                                  \
        if (tid < 2 * (RPT_) * 4) {                                                                       \
            const int rr = tid / ((RPT_) * 4), c = (tid % ((RPT_) * 4)) >> 2, qt = tid & 3;               \
            float a = 0.f;                                                                                \
            for (int i = qt * 64; i < qt * 64 + 64; ++i) a += tp[rr][c][i];                               \
            tq[rr][c][qt] = a;                                                                            \
        }                                                                                                 \
        __syncthreads();                                                                                  \
        if (tid < 2 * (RPT_)) {                                                                           \
            const int rr = tid / (RPT_), c = tid % (RPT_);                                                \
            if (c0 + c < rows) T[rr][c0 + c] = (tq[rr][c][0] + tq[rr][c][1]) + (tq[rr][c][2] + tq[rr][c][3]);   \
        }                                                                                                 \
    }
    static_assert(RMAX * 4 * 2 <= 256, "quarter sums: one thread per (row, router row, quarter)");
    if (nl + r == 11) { ROUTE2_TRIPS(11, nproj) }
    else { ROUTE2_TRIPS(8, (rows + 7) / 8) }
#undef ROUTE2_TRIPS
    __syncthreads();
    const int rr = tid >> 7, lt = tid & 127;                   // threads 0..127: row m0, 128..255: row m0 + 1
    const int m = m0 + rr;
    if (m >= M || lt > nproj) return;
    bf16_t* u = U + (long)m * ldu;
    const int used = nproj * nl * r;
    if (lt == nproj) {
        for (int c = used; c < ucols; ++c) u[c] = 0;
        return;
    }
    const float* t = &T[rr][lt * (nl + r)];
    float e[8], mx = -INFINITY;
    for (int i = 0; i < nl; ++i) mx = fmaxf(mx, t[i]);
    float sum = 0.f;
    for (int i = 0; i < nl; ++i) { e[i] = expf(t[i] - mx); sum += e[i]; }
    const float inv = 1.0f / sum;
    for (int i = 0; i < nl; ++i)
        for (int j = 0; j < r; ++j) u[lt * nl * r + i * r + j] = f2bf(scaling * e[i] * inv * t[nl + j]);
}

}  // namespace

float* crab_rowfin_T(const crab_gemm_desc* d);      // rowfin.hip: where the router product of a deferred hyper-LoRA update goes

// M <= 16 with the fused RoPE + KV append requested: gemm_skinny_dma_kernel does it in its epilogue (CRAB_SKINNY_ROPE=0: separate pass)
bool crab_skinny_fuses_rope(const crab_gemm_desc* d) {
    static const int on = []() { const char* e = getenv("CRAB_SKINNY_ROPE"); return !(e && e[0] == '0'); }();
    return on && d->rope_tab && d->M <= 16 && d->tune == 0 && !d->c_fp32 && (d->rope_d & 15) == 0 && d->rope_d >= 16 && (d->ldc & 3) == 0 &&
           (d->ldb & 7) == 0 && (!d->A2 || (d->ldb2 & 7) == 0) && d->N == (d->rope_H + 2 * d->rope_Hk) * d->rope_d &&
           (((uintptr_t)d->C | (uintptr_t)d->rope_k_cache | (uintptr_t)d->rope_v_cache) & 7) == 0;
}

// called from crab_gemm_bf16 (gemm.hip) for unbatched problems with M <= 128
int crab_gemm_skinny_launch(crab_ctx* ctx, hipStream_t s, const crab_gemm_desc* d) {
    SkinnyP p;
    p.A = (const bf16_t*)d->A; p.B = (const bf16_t*)d->B; p.C = d->C; p.bias = (const bf16_t*)d->bias; p.R = (const bf16_t*)d->R;
    p.A2 = (const bf16_t*)d->A2; p.B2 = (const bf16_t*)d->B2;
    p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldr = d->ldr; p.lda2 = d->lda2; p.ldb2 = d->ldb2;
    p.M = d->M; p.N = d->N; p.K = d->K; p.K2 = d->A2 ? d->K2 : 0; p.act = d->act; p.c_fp32 = crab_cflags(d); p.res_scale = d->res_scale;
    p.Bx = nullptr; p.ldbx = 0; p.Tx = nullptr;
    p.rope_tab = nullptr; p.rope_kc = p.rope_vc = nullptr; p.rope_pos_dev = nullptr; p.rope_H = p.rope_Hk = p.rope_d = p.rope_Tmax = p.rope_pos0 = 0;
    p.rope_row_off = nullptr;
    if (crab_skinny_fuses_rope(d)) {
        p.rope_tab = d->rope_tab; p.rope_kc = (bf16_t*)d->rope_k_cache; p.rope_vc = (bf16_t*)d->rope_v_cache; p.rope_pos_dev = d->rope_pos_dev;
        p.rope_H = d->rope_H; p.rope_Hk = d->rope_Hk; p.rope_d = d->rope_d; p.rope_Tmax = d->rope_Tmax; p.rope_pos0 = d->rope_pos0;
        p.rope_row_off = d->rope_row_off;
    }
    // M <= 16: the LDS-DMA ring kernel (tune 1 / 2 / 4 keep the register-direct kernel for A/B runs); d->tune == 9: the same with
    // raw fp32 sums to the workspace (used by crab_gemm_bf16 for its fused reduction epilogues)
    if (d->M <= 16 && (d->tune == 0 || d->tune == 9) && (d->ldb & 7) == 0 && (!d->A2 || (d->ldb2 & 7) == 0)) {
        // <NT = 1, NS = 4>: 16 weight rows per block, 4 slots of 2 KiB per wave, two blocks resident per CU.  Measured alternatives
        // (profiles/README.md): a 6-slot ring changes nothing where a CU holds one block (o, down) and loses where it held two;
        // 32 rows per block (half the activation re-reads) is 10-25 % slower on every shape at M = 1 .. 16.
        float* part = d->tune == 9 ? (float*)d->workspace : nullptr;
        int extra_blocks = 0;
        if (d->tune == 9 && d->lora_RA) {               // the projection's own router rows ride on this launch (rowfin.hip applies them)
            p.Bx = (const bf16_t*)d->lora_RA; p.ldbx = d->lora_ldra; p.Tx = crab_rowfin_T(d);
            extra_blocks = SKX;
        }
        hipLaunchKernelGGL((gemm_skinny_dma_kernel<1, 4>), dim3((d->N + 15) / 16 + extra_blocks), dim3(SK_WAVES * 64), 0, s, p, part);
        return crab_check_launch(ctx, "gemm_skinny_dma_kernel");
    }
    // NT (weight tiles per block): bigger NT = fewer replicated activation reads but a smaller grid.  d->tune forces it.
    int mt = d->M <= 16 ? 1 : (d->M <= 32 ? 2 : (d->M <= 64 ? 4 : 8));
    int nt = d->tune > 0 ? d->tune : 0;
    if (nt == 0) {
        nt = 1;
        if (mt >= 2 && d->N >= 8192) nt = 2;
        if (mt >= 4 && d->N >= 16384) nt = 4;
    }
    if (nt != 1 && nt != 2 && nt != 4) nt = 1;
    if (mt == 8 && nt == 4) nt = 2;                                  // 64 KiB static LDS limit for the reduction buffer
    dim3 grid((d->N + 16 * nt - 1) / (16 * nt)), block(SK_WAVES * 64);
#define SK_LAUNCH(MT_, NT_) hipLaunchKernelGGL((gemm_skinny_kernel<MT_, NT_>), grid, block, 0, s, p)
#define SK_NT(MT_) do { if (nt == 1) SK_LAUNCH(MT_, 1); else if (nt == 2) SK_LAUNCH(MT_, 2); else SK_LAUNCH(MT_, 4); } while (0)
    if (mt == 1) SK_NT(1);
    else if (mt == 2) SK_NT(2);
    else if (mt == 4) SK_NT(4);
    else { if (nt == 1) SK_LAUNCH(8, 1); else SK_LAUNCH(8, 2); }
#undef SK_NT
#undef SK_LAUNCH
    return crab_check_launch(ctx, "gemm_skinny_kernel");
}

// K slices of the router product: enough (slice, 64-row block) pairs for ~4 blocks per CU.  Decode (M <= 256) is capped
// by the 16-slab limit; prefill (M = 11232: 176 row blocks) gets 6 slices - with a single slice its 176 four-wave blocks
// ran one dependent load->MFMA chain per wave over the whole K (111 us per launch, 5 % of the prefill phase).
static int route_slices(int M, int K) {
    int mblocks = (M + 63) / 64;
    int nslices = (1024 + mblocks - 1) / (mblocks > 0 ? mblocks : 1);
    int maxs = (K + 127) / 128;
    if (nslices > maxs) nslices = maxs;
    if (nslices < 1) nslices = 1;
    if (nslices > 16) nslices = 16;
    return nslices;
}

// launch shape shared by the workspace query and the launch: row tiles per wave and K slices
static void route_cfg(int M, int K, int* MT, int* nslices) {
    *MT = M >= 4096 ? 4 : 1;                                      // prefill-sized M: four row tiles per wave
    *nslices = route_slices(*MT == 4 ? (M + 3) / 4 : M, K);       // (slice, row block) pairs for ~4 blocks per CU
}

// r06: the query is MONOTONE in M and K - callers size one workspace for their largest chunk and reuse it for smaller ones, and slices * M is not monotone
// (M = 57 344: 5 slices = 286 720 partial rows; M = 50 000: 6 slices = 300 000 - generate_avs_many with 512 samples failed with "workspace too small" on the
// last chunk of its merged prefill).  Bound of route_cfg's slices * M over every m <= M: slices <= min(16, ceil(K / 128)) and slices <= 1024 / row blocks + 1
// with >= m / 256 row blocks, i.e. slices * m <= 262 144 + m.
extern "C" int64_t crab_hyperlora_route_workspace(int M, int K, int tcols) {
    if (M <= 0 || K <= 0 || tcols <= 0) return 0;
    const int64_t maxs = (K + 127) / 128 < 16 ? (K + 127) / 128 : 16;
    int64_t rows = maxs * (int64_t)M;
    if (rows > 262144 + (int64_t)M) rows = 262144 + (int64_t)M;
    return rows * tcols * (int64_t)sizeof(float);
}

extern "C" int crab_hyperlora_route(crab_ctx* ctx, void* stream, const void* X, int64_t ldx, const void* RA, int64_t ldra, int M, int K,
                                    int nproj, int nl, int r, void* U, int64_t ldu, int ucols, float scaling, void* workspace,
                                    int64_t workspace_bytes) {
    if (!ctx) return CRAB_E_INVALID;
    if (!X || !RA || !U || !workspace || M <= 0 || K <= 0 || (K & 7) || (ldx & 7) || (ldra & 7))
        return crab_fail(ctx, CRAB_E_INVALID, "hyperlora_route: bad argument");
    if (nproj < 1 || nl < 1 || nl > 8 || r < 1 || ucols < nproj * nl * r) return crab_fail(ctx, CRAB_E_INVALID, "hyperlora_route: nproj, r >= 1, 1 <= nl <= 8, ucols >= nproj nl r");
    const int tcols = ((nproj * (nl + r) + 15) / 16) * 16;        // RA must hold tcols rows (zero padded)
    if (crab_hyperlora_route_workspace(M, K, tcols) > workspace_bytes) return crab_fail(ctx, CRAB_E_WORKSPACE, "hyperlora_route: workspace too small");
    if (nl + r > 16 || nproj > 3 || tcols > 64) {
        // outside the fused kernels' shapes (e.g. lora_r = 16: r is a free argument of the reference's LoraConfig, peft_hyper/tuners/lora.py:42-83):
        // the same arithmetic as two general launches - T = X . [R;A]^T in fp32 (the library GEMM, into the workspace) and the routing mix
        crab_gemm_desc g;
        memset(&g, 0, sizeof(g));
        g.A = X; g.lda = ldx; g.B = RA; g.ldb = ldra; g.C = workspace; g.ldc = tcols; g.M = M; g.N = tcols; g.K = K;
        g.c_fp32 = 1; g.res_scale = 1.0f; g.batch = 1; g.nb0 = 1;
        int rc = crab_gemm_bf16(ctx, stream, &g);
        return rc ? rc : crab_hyperlora_mix(ctx, stream, workspace, tcols, 1, U, ldu, M, nproj, nl, r, ucols, scaling);
    }
    // decode regime (one row per clip, 64 < M <= CRAB_DECODE_MAX_ROWS): one row-owning launch (~6 us) instead of the partial-product + mix pair (~11 us)
    // (at M <= 16 the row-owning launch has 1-16 blocks and loses to the pair: 6.09 vs 6.22 clips/s at batch 8, profiles/README.md)
    if (M > 64 && M <= CRAB_DECODE_MAX_ROWS && K <= 6 * 2048 && nl <= 8 && ((uintptr_t)X & 15) == 0 && ((uintptr_t)RA & 15) == 0) {
        hipStream_t s0 = (hipStream_t)stream;
#define RR_LAUNCH(Q_) hipLaunchKernelGGL((lora_route_row_kernel<Q_>), dim3(M), dim3(256), 0, s0, (const bf16_t*)X, (long)ldx, (const bf16_t*)RA, (long)ldra, K, \
                                         (bf16_t*)U, (long)ldu, nproj, nl, r, ucols, scaling)
        // two rows per block beyond 256 rows of a LONG row (r05; the down group, K = 11008: the [R;A] traffic halves - 17.6 -> 12.6 us at 512 rows;
        // at K = 4096 the one-row kernel is not traffic-bound: 9.6 / 11.6 us either way, scripts/exp/route_rows2.py).  Bit-identical.
        // CRAB_ROUTE_ROWS=1 keeps one row per block (A/B runs, tests)
        const char* e1 = getenv("CRAB_ROUTE_ROWS");
        if (M > 256 && K > 4 * 2048 && !(e1 && e1[0] == '1' && e1[1] == 0)) {
            hipLaunchKernelGGL((lora_route_row2_kernel<6>), dim3((M + 1) / 2), dim3(256), 0, s0, (const bf16_t*)X, (long)ldx, (const bf16_t*)RA, (long)ldra, M, K,
                               (bf16_t*)U, (long)ldu, nproj, nl, r, ucols, scaling);
            return crab_check_launch(ctx, "lora_route_row2_kernel");
        }
        if (K <= 2 * 2048) RR_LAUNCH(2); else if (K <= 4 * 2048) RR_LAUNCH(4); else RR_LAUNCH(6);
#undef RR_LAUNCH
        return crab_check_launch(ctx, "lora_route_row_kernel");
    }
    int MT, nslices;
    route_cfg(M, K, &MT, &nslices);
    int mblocks = (M + 64 * MT - 1) / (64 * MT);
    hipStream_t s = (hipStream_t)stream;
    int kslice = (((K + nslices - 1) / nslices) + 63) / 64 * 64;
    nslices = (K + kslice - 1) / kslice;
    dim3 grid(nslices, mblocks), block(256);
    float* part = (float*)workspace;
    const int NT = tcols / 16;
#define RT_LAUNCH(NT_, MT_) hipLaunchKernelGGL((lora_t_partial_kernel<NT_, MT_>), grid, block, 0, s, (const bf16_t*)X, (long)ldx, (const bf16_t*)RA, \
                                               (long)ldra, part, M, K, kslice)
    if (MT == 4) { if (NT == 1) RT_LAUNCH(1, 4); else if (NT == 2) RT_LAUNCH(2, 4); else RT_LAUNCH(3, 4); }
    else { if (NT == 1) RT_LAUNCH(1, 1); else if (NT == 2) RT_LAUNCH(2, 1); else RT_LAUNCH(3, 1); }
#undef RT_LAUNCH
    int rc = crab_check_launch(ctx, "lora_t_partial_kernel");
    if (rc) return rc;
    if (tcols > 64) return crab_fail(ctx, CRAB_E_UNSUPPORTED, "hyperlora_route: tcols <= 64");
    unsigned blocks = (unsigned)((M + 3) / 4);
    hipLaunchKernelGGL(lora_mix_reduce_kernel, dim3(blocks), dim3(256), 0, s, part, nslices, tcols, (bf16_t*)U, (long)ldu, M, nproj, nl, r, ucols, scaling);
    return crab_check_launch(ctx, "lora_mix_reduce_kernel");
}

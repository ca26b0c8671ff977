// bf16 MFMA GEMM, direct-to-LDS staging variant (prefill / encoder regime, M > 128).
//
// Same math, tile (128x128x64, 4 waves of 64x64) and LDS image as gemm_bt_kernel in gemm.hip, but the global->LDS
// copy is done by the LDS-DMA path (`global_load_lds_dwordx4`, 16 B per lane): no staging VGPRs, no ds_write pass, no
// address VALU in the loop.  An LDS-DMA instruction writes wave-uniform base + lane*16, i.e. one linear 1-KiB piece
// (8 tile rows of 128 B) per wave instruction, so the 16-byte XOR swizzle (chunk ^ ((row>>1)&7)) that keeps the
// ds_read_b128 fragment reads conflict-free is applied to the per-lane SOURCE address (lane l of a piece fetches
// logical chunk (l&7) ^ swz(row)) and again on the read side (cdna guide 5.4 rule 21: linear dest + inverse-swizzled
// source + swizzled read).  Out-of-range rows / K tails fetch from a zero page instead, so any M, N and K % 8 == 0
// work without a register path.
//
// Pipeline: two LDS stages; the DMA of tile t+1 is issued before the MFMAs of tile t and drained (vmcnt(0)) at the
// single barrier per K tile.  Two blocks per CU (64 KiB LDS each) overlap one block's drain with the other's MFMAs.
#include "common.h"
#include "crab_internal.h"
#include "gemm_epilogue.h"
#include <stdlib.h>

namespace {

struct GemmGP {
    const bf16_t* A; const bf16_t* B; void* C; const bf16_t* bias; const bf16_t* R;
    const bf16_t* A2; const bf16_t* B2;
    long lda, ldb, ldc, ldr, lda2, ldb2;
    int M, N, K, K2, act, c_fp32;
    float res_scale;
    int nb0;
    long sA0, sA1, sB0, sB1, sC0, sC1, sR0, sR1, sBias0, sBias1;
    int tiles_m, tiles_n;
    int splitk;          // > 1: blockIdx.y = K slice (unbatched), raw fp32 partial tiles to `part` [slice][M][N]
    float* part;
    // prefill q|k|v projection (ring kernel, head_dim 128): q and k column tiles rotate and scatter in the epilogue (crab_gemm_desc.rope_S > 1)
    const float* rope_tab; bf16_t* rope_kc; const int* rope_pos_ids; long rope_ld_pos;
    int rope_S, rope_H, rope_Hk, rope_Tmax, rope_pos0;
    bf16_t* rope_vc; bf16_t* rope_vt; long rope_vt_ld;      // v column tiles: cache append + V^T in the epilogue too (rope_vt != NULL)
};

__device__ __attribute__((aligned(16))) uint32_t g_zero_page[64];      // zero-initialised device memory (256 B)

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* gbl_vptr;

constexpr int GBK = 64;

// NTB: the weight (B) tiles are streamed once - decode regime - and are LDS-DMA'd with the non-temporal hint (aux = 2)
template <int BM, int BN, bool NTB = false>
__global__ __launch_bounds__(256) void gemm_bt_glds_kernel(GemmGP p) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int PA = BM / 8, PB = BN / 8;                  // 1-KiB pieces per operand tile
    constexpr int PPW = (PA + PB) / 4;                       // pieces per wave per K tile
    __shared__ __attribute__((aligned(16))) bf16_t lds[2][(BM + BN) * GBK];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;

    const int z = p.splitk > 1 ? 0 : blockIdx.y;
    const int z0 = z % p.nb0, z1 = z / p.nb0;
    const bf16_t* A = p.A + z0 * p.sA0 + z1 * p.sA1;
    const bf16_t* B = p.B + z0 * p.sB0 + z1 * p.sB1;

    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = xcd_remap(blockIdx.x, nwg);
    const int tm = bid % p.tiles_m, tn = bid / p.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    const int nk1 = (p.K + GBK - 1) / GBK;
    const int nk2 = p.A2 ? (p.K2 + GBK - 1) / GBK : 0;
    const int nk = nk1 + nk2;
    int t_begin = 0, t_end = nk;
    if (p.splitk > 1) {
        t_begin = (int)((long)nk * blockIdx.y / p.splitk);
        t_end = (int)((long)nk * (blockIdx.y + 1) / p.splitk);
    }

    // per-lane staging coordinates, hoisted out of the K loop.  With PA == PB (square tile) waves 0,1 stage the
    // activation tile and waves 2,3 the weight tile, so the operand choice is wave-uniform.
    static_assert(PA == PB && PA % PPW == 0, "operand choice must be uniform per wave");
    const bool isA = wave * PPW < PA;
    const int prow = lane >> 3, pc = lane & 7;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);
    const bf16_t* base1 = isA ? A + (long)m0 * p.lda : B + (long)n0 * p.ldb;
    const bf16_t* base2 = p.A2 ? (isA ? p.A2 + (long)m0 * p.lda2 : p.B2 + (long)n0 * p.ldb2) : zero;
    const int ld1 = (int)(isA ? p.lda : p.ldb), ld2 = (int)(isA ? p.lda2 : p.ldb2);
    const int lim = isA ? p.M - m0 : p.N - n0;              // valid rows of this operand tile
    int off1[PPW], off2[PPW], kc[PPW], ldso[PPW];
    bool rok[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wave * PPW + i;
        const int pr0 = (isA ? q : q - PA) * 8;               // first tile row of the piece
        const int row = pr0 + prow;
        const int c = pc ^ ((row >> 1) & 7);                 // logical chunk this lane fetches (inverse swizzle)
        kc[i] = c * 8;
        rok[i] = row < lim;
        off1[i] = row * ld1 + c * 8;
        off2[i] = row * ld2 + c * 8;
        ldso[i] = (isA ? 0 : BM * GBK) + pr0 * GBK;           // wave-uniform LDS element offset of the piece
    }

#define STAGE(T_, BUF_)                                                                                   \
    {                                                                                                     \
        const int t_ = (T_);                                                                              \
        const bool s2_ = t_ >= nk1;                                                                       \
        const int k0_ = (s2_ ? t_ - nk1 : t_) * GBK;                                                      \
        const int Ks_ = s2_ ? p.K2 : p.K;                                                                 \
        const bf16_t* bs_ = (s2_ ? base2 : base1) + k0_;                                                  \
        _Pragma("unroll") for (int i = 0; i < PPW; ++i) {                                                 \
            const bool ok_ = rok[i] && (k0_ + kc[i] < Ks_);                                               \
            const bf16_t* src_ = bs_ + (s2_ ? off2[i] : off1[i]);                                         \
            src_ = ok_ ? src_ : zero;                                                                     \
            bf16_t* dst_ = &lds[(BUF_)][__builtin_amdgcn_readfirstlane(ldso[i])];                         \
            if constexpr (NTB) {                                                                          \
                if (isA) __builtin_amdgcn_global_load_lds((gbl_vptr)src_, (lds_vptr)dst_, 16, 0, 0);      \
                else __builtin_amdgcn_global_load_lds((gbl_vptr)src_, (lds_vptr)dst_, 16, 0, 2);          \
            } else {                                                                                      \
                __builtin_amdgcn_global_load_lds((gbl_vptr)src_, (lds_vptr)dst_, 16, 0, 0);               \
            }                                                                                             \
        }                                                                                                 \
    }

    f32x4_t acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // carried source pointers for the K tiles wholly inside the first K segment (as in the ring kernel below): one 64-bit
    // add per piece instead of the segment / tail / row selects; tiles are staged in order t_begin, t_begin + 1, ...
    const int t_fast = min(t_end, p.K / GBK);
    const bf16_t* fptr[PPW];
    int fadv[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        fptr[i] = rok[i] ? base1 + (long)t_begin * GBK + off1[i] : zero;
        fadv[i] = rok[i] ? GBK : 0;
    }
#define FSTAGE(BUF_)                                                                                      \
    {                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < PPW; ++i) {                                                 \
            bf16_t* dst_ = &lds[(BUF_)][__builtin_amdgcn_readfirstlane(ldso[i])];                         \
            if constexpr (NTB) {                                                                          \
                if (isA) __builtin_amdgcn_global_load_lds((gbl_vptr)fptr[i], (lds_vptr)dst_, 16, 0, 0);   \
                else __builtin_amdgcn_global_load_lds((gbl_vptr)fptr[i], (lds_vptr)dst_, 16, 0, 2);       \
            } else {                                                                                      \
                __builtin_amdgcn_global_load_lds((gbl_vptr)fptr[i], (lds_vptr)dst_, 16, 0, 0);            \
            }                                                                                             \
            fptr[i] += fadv[i];                                                                           \
        }                                                                                                 \
    }
#define XSTAGE(T_, BUF_)                                                                                  \
    {                                                                                                     \
        if ((T_) < t_fast) FSTAGE(BUF_) else STAGE(T_, BUF_)                                              \
    }

    if (t_begin < t_end) XSTAGE(t_begin, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int fr = lane & 15, fg = lane >> 4;
    for (int t = t_begin; t < t_end; ++t) {
        const int cur = (t - t_begin) & 1;
        if (t + 1 < t_end) XSTAGE(t + 1, cur ^ 1);
        const bf16_t* la_ = &lds[cur][0];
        const bf16_t* lb_ = &lds[cur][BM * GBK];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t wf[TN], xf[TM];
            const int chunk = ks * 4 + fg;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                int row = wn * WN + ni * 16 + fr;
                wf[ni] = *reinterpret_cast<const bf16x8_t*>(lb_ + row * GBK + ((chunk ^ ((row >> 1) & 7)) << 3));
            }
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                int row = wm * WM + mi * 16 + fr;
                xf[mi] = *reinterpret_cast<const bf16x8_t*>(la_ + row * GBK + ((chunk ^ ((row >> 1) & 7)) << 3));
            }
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    if (p.splitk > 1) {          // raw partial tile; reduced (fixed slice order) by the split-K epilogue kernels in gemm.hip
        float* part = p.part + (long)blockIdx.y * p.M * p.N;
        const bool v4 = (p.N & 3) == 0;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int m = m0 + wm * WM + mi * 16 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int n = n0 + wn * WN + ni * 16 + fg * 4;
                if (n >= p.N) continue;
                float* o = part + (long)m * p.N + n;
                if (v4) *reinterpret_cast<f32x4_t*>(o) = acc[ni][mi];
                else
                    for (int r = 0; r < 4 && n + r < p.N; ++r) o[r] = acc[ni][mi][r];
            }
        }
        return;
    }
    // output stage shared with gemm_bt_kernel (gemm_epilogue.h): unguarded + activation-specialised on interior sub-tiles
    gemm_epilogue<TM, TN>(acc, p.act, m0 + wm * WM, n0 + wn * WN, fr, fg, p.M, p.N, p.bias ? p.bias + z0 * p.sBias0 + z1 * p.sBias1 : nullptr,
                          p.R ? p.R + z0 * p.sR0 + z1 * p.sR1 : nullptr, p.ldr, p.res_scale, p.C, z0 * p.sC0 + z1 * p.sC1, p.ldc, p.c_fp32);
}

#undef STAGE
#undef FSTAGE
#undef XSTAGE

// ------------------------------------------------------------------------------------------------------------
// Ring variant for large problems: 256x256 tile, 8 waves (2 x 4, 128x64 per wave, two waves per SIMD), BK = 32, four
// LDS stages (4 x 32 KiB = 128 KiB, one block per CU).  A 256^2 tile halves the L2->LDS bytes per FLOP of the 128^2
// tile (the 128^2 kernel is bound by LDS-DMA latency/L2 bandwidth: ~13 TB/s of L2 traffic at 860 TFLOP/s); the
// LDS-DMA is issued THREE K tiles ahead and retired with a COUNTED wait (s_waitcnt vmcnt(8): the two youngest tiles
// stay in flight across the barrier), so L2/HBM latency is covered by ~3 tiles (~1.3 us) of MFMA work.
// One raw s_barrier per K tile (32 MFMAs per wave).
//   LDS image per stage: [BM + BN rows][32 k] bf16, 64-B rows, 16-B chunk swizzle pc = c ^ P[(row>>2)&3], P = {0,2,3,1}
//   (conflict-free for the 16-lane ds_read_b128 groups {0-3,12-15,20-27}, ... of gfx950).
//   RAW: a tile is read one iteration after the vmcnt+barrier that retired it.  WAR: the stage refilled in iteration
//   t was last read in iteration t-1, whose reads are drained (lgkmcnt(0)) before that iteration's barrier.
// ------------------------------------------------------------------------------------------------------------
constexpr int RBK = 32, RNS = 4;

template <int BM, int BN, int WGM, int WGN, bool NTB = false>
__global__ __launch_bounds__(WGM * WGN * 64) void gemm_bt_ring_kernel(GemmGP p) {
    constexpr int NW = WGM * WGN;
    constexpr int WM = BM / WGM, WN = BN / WGN;
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int PA = BM / 16, PB = BN / 16;                // 1-KiB pieces (16 rows x 64 B) per operand tile
    constexpr int PPW = (PA + PB) / NW;                      // pieces per wave per K tile (4)
    constexpr int STAGE_ELEMS = (BM + BN) * RBK;
    __shared__ __attribute__((aligned(16))) bf16_t lds[RNS * STAGE_ELEMS];
    static_assert(PA == PB && PA % PPW == 0, "operand choice must be uniform per wave");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    // blockIdx.y = batch index, or (splitk > 1, unbatched) the K slice whose raw fp32 partial tile this block produces
    const int sk = p.splitk > 1 ? (int)blockIdx.y : 0;
    const int z = p.splitk > 1 ? 0 : (int)blockIdx.y;
    const int z0 = z % p.nb0, z1 = z / p.nb0;
    const bf16_t* A = p.A + z0 * p.sA0 + z1 * p.sA1;
    const bf16_t* B = p.B + z0 * p.sB0 + z1 * p.sB1;
    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = xcd_remap(blockIdx.x, nwg);
    int tm, tn;
    tile_coords(bid, p.tiles_m, p.tiles_n, 8, tm, tn);          // 8 x 4 tile patch per XCD (32 CUs, one block each)
    const int m0 = tm * BM, n0 = tn * BN;
    const int nk1 = (p.K + RBK - 1) / RBK;
    const int nk2 = p.A2 ? (p.K2 + RBK - 1) / RBK : 0;
    const int nk_per = (nk1 + nk2 + p.splitk - 1) / p.splitk;
    const int t_first = sk * nk_per;                                   // first K tile of this slice (both K segments chained)
    const int nk = min(nk1 + nk2, t_first + nk_per) - t_first;        // K tiles of this slice (>= 1 by construction of splitk)

    const bool isA = wave * PPW < PA;
    const int prow = lane >> 2, pc = lane & 3;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page);
    const bf16_t* base1 = isA ? A + (long)m0 * p.lda : B + (long)n0 * p.ldb;
    const bf16_t* base2 = p.A2 ? (isA ? p.A2 + (long)m0 * p.lda2 : p.B2 + (long)n0 * p.ldb2) : zero;
    const int ld1 = (int)(isA ? p.lda : p.ldb), ld2 = (int)(isA ? p.lda2 : p.ldb2);
    const int lim = isA ? p.M - m0 : p.N - n0;
    int off1[PPW], off2[PPW], kc[PPW], ldso[PPW];
    bool rok[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        const int q = wave * PPW + i;
        const int pr0 = (isA ? q : q - PA) * 16;
        const int row = pr0 + prow;
        const int c = pc ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3);
        kc[i] = c * 8;
        rok[i] = row < lim;
        off1[i] = row * ld1 + c * 8;
        off2[i] = row * ld2 + c * 8;
        ldso[i] = (isA ? 0 : BM * RBK) + pr0 * RBK;
    }
#define RSTAGE(T_)                                                                                        \
    {                                                                                                     \
        const int tl_ = (T_);                                                                             \
        const int t_ = t_first + tl_;                                                                     \
        const int sb_ = (tl_ & (RNS - 1)) * STAGE_ELEMS;                                                  \
        const bool s2_ = t_ >= nk1;                                                                       \
        const int k0_ = (s2_ ? t_ - nk1 : t_) * RBK;                                                      \
        const int Ks_ = s2_ ? p.K2 : p.K;                                                                 \
        const bf16_t* bs_ = (s2_ ? base2 : base1) + k0_;                                                  \
        _Pragma("unroll") for (int i = 0; i < PPW; ++i) {                                                 \
            const bool ok_ = rok[i] && (k0_ + kc[i] < Ks_);                                               \
            const bf16_t* src_ = bs_ + (s2_ ? off2[i] : off1[i]);                                         \
            src_ = ok_ ? src_ : zero;                                                                     \
            bf16_t* dst_ = &lds[sb_ + __builtin_amdgcn_readfirstlane(ldso[i])];                           \
            if constexpr (NTB) {                                                                          \
                if (isA) __builtin_amdgcn_global_load_lds((gbl_vptr)src_, (lds_vptr)dst_, 16, 0, 0);      \
                else __builtin_amdgcn_global_load_lds((gbl_vptr)src_, (lds_vptr)dst_, 16, 0, 2);          \
            } else {                                                                                      \
                __builtin_amdgcn_global_load_lds((gbl_vptr)src_, (lds_vptr)dst_, 16, 0, 0);               \
            }                                                                                             \
        }                                                                                                 \
    }

    // Fast staging for the K tiles that lie wholly inside the first K segment (all but the last few): the per-piece source
    // pointer is carried and advanced by one K tile per call (0 for rows outside the operand: they keep reading the zero
    // page), so a piece costs the LDS-DMA plus one 64-bit add instead of the segment / tail / row selects of RSTAGE.
    // FSTAGE must be called for local tiles 0, 1, 2, ... in order (it is: prologue 0..2, then t + 3).
    const int nfast = max(0, min(nk, p.K / RBK - t_first));
    const bf16_t* fptr[PPW];
    int fadv[PPW];
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
        fptr[i] = rok[i] ? base1 + (long)t_first * RBK + off1[i] : zero;
        fadv[i] = rok[i] ? RBK : 0;
    }
#define FSTAGE(T_)                                                                                        \
    {                                                                                                     \
        const int sbf_ = ((T_) & (RNS - 1)) * STAGE_ELEMS;                                                \
        _Pragma("unroll") for (int i = 0; i < PPW; ++i) {                                                 \
            bf16_t* dst_ = &lds[sbf_ + __builtin_amdgcn_readfirstlane(ldso[i])];                          \
            if constexpr (NTB) {                                                                          \
                if (isA) __builtin_amdgcn_global_load_lds((gbl_vptr)fptr[i], (lds_vptr)dst_, 16, 0, 0);   \
                else __builtin_amdgcn_global_load_lds((gbl_vptr)fptr[i], (lds_vptr)dst_, 16, 0, 2);       \
            } else {                                                                                      \
                __builtin_amdgcn_global_load_lds((gbl_vptr)fptr[i], (lds_vptr)dst_, 16, 0, 0);            \
            }                                                                                             \
            fptr[i] += fadv[i];                                                                           \
        }                                                                                                 \
    }
#define XSTAGE(T_)                                                                                        \
    {                                                                                                     \
        if ((T_) < nfast) FSTAGE(T_) else RSTAGE(T_)                                                      \
    }

    f32x4_t acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // prologue: tiles 0..2 in flight, tile 0 retired
    XSTAGE(0);
    if (nk > 1) XSTAGE(1);
    if (nk > 2) XSTAGE(2);
    static_assert(PPW == 4, "vmcnt(8) below = two tiles of 4 LDS-DMA instructions per wave");
    if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const int fr = lane & 15, fg = lane >> 4;
    // fragment byte offsets inside a stage (constant over the loop)
    int wofs[TN], xofs[TM];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int row = wn * WN + ni * 16 + fr;
        wofs[ni] = BM * RBK + row * RBK + ((fg ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3)) << 3);
    }
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int row = wm * WM + mi * 16 + fr;
        xofs[mi] = row * RBK + ((fg ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3)) << 3);
    }

    // Two wave groups run ONE barrier apart (waves w and w+4 share a SIMD): while one group issues its LDS-DMA and
    // fragment reads, the other runs its 32 MFMAs, so the memory-instruction issue time of one wave is hidden under
    // the MFMAs of its SIMD partner instead of idling the matrix pipe (role split as in the cdna guide's 8-phase
    // schedule, at K-tile granularity).  Every wave executes the same number of barriers.
    const int grp = wave / (NW / 2);
    if (grp == 1) __builtin_amdgcn_s_barrier();
    for (int t = 0; t < nk; ++t) {
        // ---- LOAD segment: fragments of tile t first (12 ds_read_b128 issue in ~50 cycles), THEN the DMA for tile t+3:
        // an LDS-DMA instruction costs ~100 issue cycles (measured with s_memtime stamps), so the fragment reads
        // complete underneath the four DMA issues instead of after them
        const bf16_t* st = &lds[(t & (RNS - 1)) * STAGE_ELEMS];
        bf16x8_t wf[TN], xf[TM];
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) wf[ni] = *reinterpret_cast<const bf16x8_t*>(st + wofs[ni]);
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) xf[mi] = *reinterpret_cast<const bf16x8_t*>(st + xofs[mi]);
        __builtin_amdgcn_sched_barrier(0);
        if (t + 3 < nk) XSTAGE(t + 3);
        // retire tile t+1 (the two youngest tiles stay in flight) and drain this tile's LDS reads BEFORE the barrier:
        // after it the partner group may refill the stage these reads came from
        if (t + 3 < nk) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA segment
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ni = 0; ni < TN; ++ni)
#pragma unroll
            for (int mi = 0; mi < TM; ++mi)
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
#undef RSTAGE
#undef FSTAGE
#undef XSTAGE
    if (p.splitk > 1) {          // raw partial tile; reduced in a fixed slice order by the split-K epilogue kernels (gemm.hip)
        float* part = p.part + (long)sk * p.M * p.N;
        const bool v4 = (p.N & 3) == 0;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int m = m0 + wm * WM + mi * 16 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int n = n0 + wn * WN + ni * 16 + fg * 4;
                if (n >= p.N) continue;
                float* o = part + (long)m * p.N + n;
                if (v4) *reinterpret_cast<f32x4_t*>(o) = acc[ni][mi];
                else
                    for (int r = 0; r < 4 && n + r < p.N; ++r) o[r] = acc[ni][mi][r];
            }
        }
        return;
    }
    if constexpr (BM == 256 && BN == 256) {
        if (p.rope_tab && (n0 / 128 < p.rope_H + p.rope_Hk || p.rope_vt)) {
            // ---- q / k column tile of the prefill q|k|v projection: RoPE (modeling_llama.py:204-236) and the K-cache append (:408-412) HERE
            // instead of a pass over C (qkv_rope_split_tile_kernel read and re-wrote these columns: 2/3 of its 1.4 GB per 35-clip chunk).
            // The tile is two heads of 128; the rotation partner of dim i (< 64) is dim i + 64, held by the wave one column block over,
            // so the tile goes through LDS once: bf16(acc + bias) - the value the unfused pair stores and re-reads - in a [256][256] image
            // (exactly the 128 KB of the ring, idle now) with the 16-byte chunk index XOR-ed with (row & 31), then every thread takes
            // (row, head, 8 dims + their partners), rotates with the pinned rope_lo / rope_hi order and stores q in place / k into the cache.
            // Same arithmetic, same roundings as GEMM -> qkv_rope_split: bit-identical.
            __syncthreads();                                    // every wave is past its last fragment read
            uint32_t* T32 = reinterpret_cast<uint32_t*>(lds);
            const bf16_t* bias = p.bias;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int col = wn * WN + ni * 16 + fg * 4;
                u32x2 bw = {0u, 0u};
                if (bias) bw = *reinterpret_cast<const u32x2*>(bias + n0 + col);
                const float b0 = lo_bf(bw.x), b1 = hi_bf(bw.x), b2 = lo_bf(bw.y), b3 = hi_bf(bw.y);
#pragma unroll
                for (int mi = 0; mi < TM; ++mi) {
                    const int row = wm * WM + mi * 16 + fr;
                    const uint32_t w0 = pack_bf2(acc[ni][mi][0] + b0, acc[ni][mi][1] + b1), w1 = pack_bf2(acc[ni][mi][2] + b2, acc[ni][mi][3] + b3);
                    const int ch = (col >> 3) ^ (row & 31);
                    *reinterpret_cast<u32x2*>(T32 + row * 128 + ch * 4 + ((col >> 2) & 1) * 2) = u32x2{w0, w1};
                }
            }
            __syncthreads();
            const int hh0 = n0 / 128;
            if (hh0 >= p.rope_H + p.rope_Hk) {
                // ---- v column tile (two kv heads): the rows go to the V cache as they are, and transposed to V^T [b, hk, d, vt_ld] for the
                // prefill attention.  V^T is written in octets of 8 consecutive tokens of ONE sequence starting at s % 8 == 0 (16-byte
                // stores); a tile's rows start anywhere in a sequence, so the octets cut by the tile's first / middle / last row or begun
                // before it are written element by element by whichever tile holds each element (the neighbouring row tile writes the rest).
                const int hk0 = hh0 - p.rope_H - p.rope_Hk;
                for (int it = tid; it < 256 * 32; it += NW * 64) {
                    const int ch = it & 31, r = it >> 5;
                    const int m = m0 + r;
                    if (m >= p.M) continue;
                    const int b = m / p.rope_S, sq = m - b * p.rope_S;
                    const u32x4 v = *reinterpret_cast<const u32x4*>(T32 + r * 128 + ((ch ^ (r & 31)) << 2));
                    bf16_t* dst = p.rope_vc + (((long)b * p.rope_Hk + hk0 + (ch >> 4)) * p.rope_Tmax + p.rope_pos0 + sq) * 128 + (ch & 15) * 8;
                    *reinterpret_cast<u32x4*>(dst) = v;
                }
                {
                    const int col = tid & 255, ra = (tid >> 8) * 128, rb = ra + 128;      // one column (head, dim) and half of the rows per thread
                    const int hl = col >> 7, dd = col & 127;
                    const bf16_t* T16 = reinterpret_cast<const bf16_t*>(T32);
                    int m = m0 + ra;
                    if (m < p.M) {
                        int b = m / p.rope_S, sq = m - b * p.rope_S;
                        uint32_t w[4] = {0u, 0u, 0u, 0u};
                        int k0 = sq & 7;                                   // first element of the octet being collected
                        for (int r = ra; r < rb && m < p.M; ++r, ++m) {
                            const int k = sq & 7;
                            const uint32_t e = T16[r * 256 + ((((col >> 3) ^ (r & 31)) << 3) | (col & 7))];
                            w[k >> 1] |= e << ((k & 1) * 16);
                            const bool seq_end = sq == p.rope_S - 1;
                            if (k == 7 || seq_end || r == rb - 1 || m == p.M - 1) {
                                bf16_t* vrow = p.rope_vt + (((long)b * p.rope_Hk + hk0 + hl) * 128 + dd) * p.rope_vt_ld + (sq & ~7);
                                if (k0 == 0 && (k == 7 || seq_end)) {
                                    *reinterpret_cast<u32x4*>(vrow) = u32x4{w[0], w[1], w[2], w[3]};    // whole octet (zeros behind a sequence's last token)
                                } else {
                                    for (int kk = k0; kk <= k; ++kk) vrow[kk] = (bf16_t)(w[kk >> 1] >> ((kk & 1) * 16));
                                }
                                w[0] = w[1] = w[2] = w[3] = 0u;
                                k0 = seq_end ? 0 : ((k + 1) & 7);
                            }
                            if (seq_end) { sq = 0; ++b; } else ++sq;
                        }
                    }
                }
                return;
            }
            for (int it = tid; it < 256 * 2 * 8; it += NW * 64) {
                const int c = it & 7, hl = (it >> 3) & 1, r = it >> 4;
                const int m = m0 + r;
                if (m >= p.M) continue;
                const int b = m / p.rope_S, sq = m - b * p.rope_S;
                const int pos = p.rope_pos0 + sq;
                const int rp = p.rope_pos_ids ? p.rope_pos_ids[(long)b * p.rope_ld_pos + sq] : pos;
                const u32x4 lo = *reinterpret_cast<const u32x4*>(T32 + r * 128 + (((hl * 16 + c) ^ (r & 31)) << 2));
                const u32x4 hi = *reinterpret_cast<const u32x4*>(T32 + r * 128 + (((hl * 16 + 8 + c) ^ (r & 31)) << 2));
                const float* cs = p.rope_tab + 2 * ((long)rp * 64 + c * 8);
                u32x4 olo, ohi;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x1a = lo_bf(lo[e]), x1b = hi_bf(lo[e]), x2a = lo_bf(hi[e]), x2b = hi_bf(hi[e]);
                    const float ca = cs[4 * e], sa = cs[4 * e + 1], cb = cs[4 * e + 2], sb = cs[4 * e + 3];
                    olo[e] = pack_bf2(rope_lo(x1a, x2a, ca, sa), rope_lo(x1b, x2b, cb, sb));
                    ohi[e] = pack_bf2(rope_hi(x1a, x2a, ca, sa), rope_hi(x1b, x2b, cb, sb));
                }
                const int hh = hh0 + hl;
                bf16_t* dst = hh < p.rope_H ? reinterpret_cast<bf16_t*>(p.C) + (long)m * p.ldc + (long)hh * 128 + c * 8
                                            : p.rope_kc + (((long)b * p.rope_Hk + (hh - p.rope_H)) * p.rope_Tmax + pos) * 128 + c * 8;
                *reinterpret_cast<u32x4*>(dst) = olo;
                *reinterpret_cast<u32x4*>(dst + 64) = ohi;
            }
            return;
        }
    }
    // output stage shared with gemm_bt_kernel (gemm_epilogue.h): unguarded + activation-specialised on interior sub-tiles
    gemm_epilogue<TM, TN>(acc, p.act, m0 + wm * WM, n0 + wn * WN, fr, fg, p.M, p.N, p.bias ? p.bias + z0 * p.sBias0 + z1 * p.sBias1 : nullptr,
                          p.R ? p.R + z0 * p.sR0 + z1 * p.sR1 : nullptr, p.ldr, p.res_scale, p.C, z0 * p.sC0 + z1 * p.sC1, p.ldc, p.c_fp32);
}


}  // namespace

// 256x256 ring kernel chosen for an unbatched, unsplit problem (the rule of crab_gemm_glds_launch below)
// measured (profiles/README.md, after the epilogue rewrite): the one-block-per-CU ring kernel beats the 128x128 kernel whenever its grid fills
// at least ~55 % of whole rounds of 256 blocks (M = 5616 x N = 4096: 352 tiles = 69 % of 2 rounds, 930 vs 868 TFLOP/s; M = 2056 x N = 4096 x
// K = 1024: 144 tiles = 56 % of one round, 527 vs 491); r03: also at width 768 (BEATs at 256 clips, M = 122880: q-k-v 458 vs 671 us, o 164 vs 225,
// fc1 888 vs 1236, fc2 (N = 768, K = 3072) 513 vs 795), so N and K only have to reach 768
static bool ring_chosen(const crab_gemm_desc* d) {
    const int batch = d->batch > 1 ? d->batch : 1;
    if (d->tune == 301 || d->tune == 300) return false;
    if (d->tune == 302) return true;
    const long big = (long)((d->M + 255) / 256) * ((d->N + 255) / 256) * batch;
    const long rounds = (big + 255) / 256;
    return big >= 120 && big * 100 >= rounds * 256 * 55 && d->M >= 1024 && d->N >= 768 && d->K >= 768;
}

// include/crab_hip.h: will crab_gemm_bf16(d) rotate q / k and append k in its epilogue (prefill q|k|v projection)?  A pure function of the descriptor.
extern "C" int crab_gemm_fuses_prefill_rope(const crab_gemm_desc* d) {
    static const int on = []() { const char* e = getenv("CRAB_PREFILL_ROPE_FUSED"); return !(e && e[0] == '0'); }();
    if (!on || !d || !d->rope_tab || d->rope_S <= 1 || d->rope_pos_dev || !d->rope_k_cache) return 0;
    if (d->rope_d != 128 || (d->rope_H & 1) || (d->rope_Hk & 1) || d->N != (d->rope_H + 2 * d->rope_Hk) * 128) return 0;
    if (d->M <= 256 || d->act != 0 || d->R || d->c_fp32 || d->norm_w || d->lora_RA || (d->ldc & 7)) return 0;
    if ((((uintptr_t)d->C | (uintptr_t)d->rope_k_cache | (uintptr_t)d->rope_tab) & 15) || (d->bias && ((uintptr_t)d->bias & 7))) return 0;
    if (d->rope_pos_ids && d->rope_ld_pos < d->rope_S) return 0;
    if (d->batch > 1 || !ring_chosen(d)) return 0;
    // 2: the v columns too (V-cache append + V^T): needs the V^T scratch, 16-byte rows, whole octets inside a sequence
    if (d->rope_vt && d->rope_v_cache && (d->rope_vt_ld & 7) == 0 && d->rope_vt_ld >= d->rope_S && d->rope_S >= 8 &&
        (((uintptr_t)d->rope_vt | (uintptr_t)d->rope_v_cache) & 15) == 0)
        return 2;
    return 1;
}

// called from crab_gemm_bf16 (gemm.hip) for the 128x128 tile regime
int crab_gemm_glds_launch(crab_ctx* ctx, hipStream_t s, const crab_gemm_desc* d, int splitk, float* part, int ring_split) {
    GemmGP p;
    p.splitk = splitk > 1 ? splitk : 1; p.part = part; p.rope_tab = nullptr;
    const bool ntb = splitk > 1 && d->tune != 601;                 // decode split-K regime: every weight byte is read once per step
    p.A = (const bf16_t*)d->A; p.B = (const bf16_t*)d->B; p.C = d->C; p.bias = (const bf16_t*)d->bias; p.R = (const bf16_t*)d->R;
    p.A2 = (const bf16_t*)d->A2; p.B2 = (const bf16_t*)d->B2;
    p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldr = d->ldr; p.lda2 = d->lda2; p.ldb2 = d->ldb2;
    p.M = d->M; p.N = d->N; p.K = d->K; p.K2 = d->A2 ? d->K2 : 0; p.act = d->act; p.c_fp32 = crab_cflags(d); p.res_scale = d->res_scale;
    int batch = d->batch > 1 ? d->batch : 1;
    p.nb0 = (batch > 1 && d->nb0 > 0) ? d->nb0 : 1;
    p.sA0 = d->sA0; p.sA1 = d->sA1; p.sB0 = d->sB0; p.sB1 = d->sB1; p.sC0 = d->sC0; p.sC1 = d->sC1;
    p.sR0 = d->sR0; p.sR1 = d->sR1; p.sBias0 = d->sBias0; p.sBias1 = d->sBias1;
    if (batch == 1) { p.sA0 = p.sA1 = p.sB0 = p.sB1 = p.sC0 = p.sC1 = p.sR0 = p.sR1 = p.sBias0 = p.sBias1 = 0; }
    // 256x256 ring kernel when the problem fills the chip with big tiles (ring_chosen above); 128x128 two-stage kernel otherwise
    bool use_big = p.splitk == 1 && ring_chosen(d);
    // decode regime, ring_split (chosen by the cost model in gemm.hip): 256x256 ring kernel with the K slices over blockIdx.y,
    // one round of <= 256 blocks
    if (p.splitk > 1 && ring_split) {
        p.tiles_m = (d->M + 255) / 256; p.tiles_n = (d->N + 255) / 256;
        dim3 grid(p.tiles_m * p.tiles_n, p.splitk);
        if (ntb) hipLaunchKernelGGL((gemm_bt_ring_kernel<256, 256, 2, 4, true>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((gemm_bt_ring_kernel<256, 256, 2, 4>), grid, dim3(512), 0, s, p);
        return crab_check_launch(ctx, "gemm_bt_ring_kernel(split-K)");
    }
    if (use_big) {
        p.tiles_m = (d->M + 255) / 256; p.tiles_n = (d->N + 255) / 256;
        dim3 grid(p.tiles_m * p.tiles_n, batch);
        if (crab_gemm_fuses_prefill_rope(d)) {
            p.rope_tab = d->rope_tab; p.rope_kc = (bf16_t*)d->rope_k_cache; p.rope_pos_ids = d->rope_pos_ids; p.rope_ld_pos = d->rope_ld_pos;
            p.rope_S = d->rope_S; p.rope_H = d->rope_H; p.rope_Hk = d->rope_Hk; p.rope_Tmax = d->rope_Tmax; p.rope_pos0 = d->rope_pos0;
            p.rope_vc = (bf16_t*)d->rope_v_cache; p.rope_vt = crab_gemm_fuses_prefill_rope(d) == 2 ? (bf16_t*)d->rope_vt : nullptr; p.rope_vt_ld = d->rope_vt_ld;
        }
        hipLaunchKernelGGL((gemm_bt_ring_kernel<256, 256, 2, 4>), grid, dim3(512), 0, s, p);
        return crab_check_launch(ctx, !p.rope_tab ? "gemm_bt_ring_kernel" : p.rope_vt ? "gemm_bt_ring_kernel+rope2" : "gemm_bt_ring_kernel+rope1");
    }
    p.tiles_m = (d->M + 127) / 128; p.tiles_n = (d->N + 127) / 128;
    dim3 grid(p.tiles_m * p.tiles_n, p.splitk > 1 ? p.splitk : batch);
    if (ntb) hipLaunchKernelGGL((gemm_bt_glds_kernel<128, 128, true>), grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL((gemm_bt_glds_kernel<128, 128>), grid, dim3(256), 0, s, p);
    return crab_check_launch(ctx, "gemm_bt_glds_kernel");
}

// bf16 MFMA GEMM for the DECODE regime of the projections: 128 < M <= 256 rows (one row per clip), C = A.B^T (+ A2.B2^T).
//
// At M = 256 a projection sits on the ridge: 2*M*N*K flops at ~1 PFLOP/s take about as long as the N*K*2 weight bytes at the
// ~6.4 TB/s a read stream reaches.  What the 256x256 / 128x128 prefill kernels lose here is not the inner loop but
// the decomposition: 256-wide column tiles give 48 (q|k|v) or 86 (gate|up) tiles for 256 CUs, so the K dimension was cut into
// 5 / 2 / 8 / 16 slices of 16-26 K tiles each, every block paid ~20k cycles of prologue + epilogue for ~36k cycles of K loop,
// gate|up ran on 172 of 256 CUs, and 13.4 GB of fp32 partial slabs were written and re-read per decode step - as many bytes as
// the weights themselves.
//
// This kernel makes the tile the whole batch x a NARROW column panel: BM = 256 rows (every clip) x BN = 96 or 64 weight rows,
// one block per CU, so a projection becomes ~256 blocks with at most 2-4 K slices:
//     q|k|v   N 12288 = 128 x 96, 2 slices  -> 256 blocks, 66 K tiles each, 2 slabs  (was 48 x 5: 26 K tiles, 5 slabs)
//     gate|up N 22016 = 230 x 96, no split  -> 230 blocks, 131 K tiles, NO slab: SwiGLU in this kernel's epilogue
//     o, down N  4096 =  64 x 64, 4 slices  -> 256 blocks, 33 / 87 K tiles, 4 slabs  (was 8 / 16 slabs)
// The activation operand (256 x K, 2-5.6 MB) is L2-resident and re-read by every block; the weight panel of a block is
// streamed once from HBM with the non-temporal hint.
//
// Structure: 8 waves as 8(M) x 1(N), each 32 rows x BN columns (TM = 2, TN = BN / 16 MFMA 16x16x32 tiles = 12 or 8 MFMAs per
// 32-wide k step), an NS-slot LDS ring of [256 + BN rows][64 k] images (128-byte rows: whole cache lines) filled by LDS-DMA
// (16 B per lane, 1-KiB pieces of 8 rows x 128 B, inverse-swizzled source, zero page for rows >= M / N and K tails) and retired
// with a counted vmcnt.  Each wave stages four pieces of the activation slot and one or two of the weight slot.
//
// The K loop has ONE barrier per 64-wide slot (two k steps) with the fragment reads running one MFMA group ahead.  A first version
// with the ring kernel's schedule (per 32-wide slot: reads -> wait -> barrier -> 12 MFMAs -> barrier) spent ~1000 cycles per slot
// whatever the ring depth, tile width or even with the LDS-DMA removed (scripts/exp/dec_gemm_anatomy.py): with only 12 MFMAs
// (~200 cycles) between barriers every wave serialises ds_read latency + two barrier round trips per slot.  Here an iteration is
//     MFMA(k step 0 of slot t, fragments read in the previous iteration) with the reads of k step 1 issued behind its first MFMAs
//     -> counted vmcnt + barrier(t) -> reads of k step 0 of slot t+1 -> MFMA(k step 1) interleaved with the LDS-DMA that refills slot t
// so no MFMA group waits for an LDS read issued in its own interval and the matrix pipe idles only across the one barrier.
//   RAW: slot t+1 is read after barrier(t), which every wave reaches after waiting for its own pieces of it.
//   WAR: after barrier(t) both k steps of slot t are in registers (lgkmcnt(0) before the barrier), so its ring slot is refilled.
// What bounds it (profiles/README.md, r02): the 44 KiB a CU takes in per slot through its vector-memory path, i.e. the arithmetic
// intensity of the 256 x 96 tile - not LDS reads, not HBM latency (a deeper weight ring was slower), not the matrix pipe (42 % busy).
#include "common.h"
#include "crab_internal.h"
#include <stdlib.h>
#include "gemm_epilogue.h"

namespace {

struct GemmDP {
    const bf16_t* A; const bf16_t* B; void* C; const bf16_t* bias; const bf16_t* R;
    const bf16_t* A2; const bf16_t* B2;
    long lda, ldb, ldc, ldr, lda2, ldb2;
    int M, N, K, K2, act, c_fp32;
    float res_scale;
    int splitk;          // > 1: blockIdx.y = K slice, raw fp32 partial tiles to `part` [slice][M][N]
    float* part;
};

__device__ __attribute__((aligned(16))) uint32_t g_zero_page_dec[64];      // zero-initialised device memory (256 B)

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* gbl_vptr;

constexpr int DBK = 64;      // K extent of a ring slot: 128-byte rows, so every LDS-DMA piece (8 rows x 128 B) moves whole cache lines

template <int BN, int NS, bool NTB>
__global__ __launch_bounds__(512) void gemm_dec_kernel(GemmDP p) {
    constexpr int BM = 256;
    constexpr int TM = 2, TN = BN / 16;
    constexpr int PA = BM / 8, PB = BN / 8;                  // 1-KiB pieces (8 rows x 128 B) per slot: activations, weights
    constexpr int PAW = PA / 8;                              // activation pieces per wave per slot (4)
    constexpr int PBW = (PB + 7) / 8;                        // weight pieces per wave per slot, waves < PB - 8 * (PBW - 1) take PBW
    constexpr int NPW = PAW + PBW;
    constexpr int SLOT_ELEMS = (BM + BN) * DBK;
    static_assert(BN % 16 == 0 && PBW <= 3 && NS >= 3 && NS * SLOT_ELEMS * 2 <= 160 * 1024, "tile / ring geometry");
    __shared__ __attribute__((aligned(16))) bf16_t lds[NS * SLOT_ELEMS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sk = (int)blockIdx.y;
    const int n0 = (int)blockIdx.x * BN;
    const int nk1 = (p.K + DBK - 1) / DBK;
    const int nk2 = p.A2 ? (p.K2 + DBK - 1) / DBK : 0;
    const int nk_per = (nk1 + nk2 + p.splitk - 1) / p.splitk;
    const int t_first = sk * nk_per;                                  // first K slot of this slice (both K segments chained)
    const int nk = min(nk1 + nk2, t_first + nk_per) - t_first;       // >= 1 by construction of splitk (host)

    // ---- staging coordinates.  Piece i < PAW: activation rows (wave * PAW + i) * 8 ..; piece PAW + j: weight rows (wave + 8 j) * 8 ..
    // (present when wave + 8 j < PB).  A piece is 8 rows x 128 B = whole cache lines; lane l fetches row l >> 3, chunk (l & 7) ^ swz.
    const int nb = (wave < PB - 8 * (PBW - 1)) ? PBW : PBW - 1;      // weight pieces of this wave (wave-uniform)
    const int prow = lane >> 3, pc = lane & 7;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_dec);
    int off1[NPW], off2[NPW], kc[NPW], ldso[NPW];
    bool rok[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const bool isA = i < PAW;
        const int pr0 = isA ? (wave * PAW + i) * 8 : (wave + 8 * (i - PAW)) * 8;
        const int row = pr0 + prow;
        const int c = pc ^ ((row >> 1) & 7);                         // inverse swizzle on the source chunk
        kc[i] = c * 8;
        rok[i] = isA ? row < p.M : (i - PAW < nb && n0 + row < p.N);
        off1[i] = row * (int)(isA ? p.lda : p.ldb) + c * 8;
        off2[i] = row * (int)(isA ? p.lda2 : p.ldb2) + c * 8;
        ldso[i] = __builtin_amdgcn_readfirstlane((isA ? 0 : BM * DBK) + pr0 * DBK);
    }
    const bf16_t* baseA1 = p.A;
    const bf16_t* baseB1 = p.B + (long)n0 * p.ldb;
    const bf16_t* baseA2 = p.A2 ? p.A2 : zero;
    const bf16_t* baseB2 = p.A2 ? p.B2 + (long)n0 * p.ldb2 : zero;

#define DMA_A(SRC_, DST_) __builtin_amdgcn_global_load_lds((gbl_vptr)(SRC_), (lds_vptr)(DST_), 16, 0, 0)
#define DMA_B(SRC_, DST_)                                                                                 \
    {                                                                                                     \
        if constexpr (NTB) __builtin_amdgcn_global_load_lds((gbl_vptr)(SRC_), (lds_vptr)(DST_), 16, 0, 2);  \
        else __builtin_amdgcn_global_load_lds((gbl_vptr)(SRC_), (lds_vptr)(DST_), 16, 0, 0);               \
    }
    // generic staging of local slot TL_ into ring element offset SB_ (K tails, second K segment, rows outside the operands)
#define RSTAGE(TL_, SB_)                                                                                  \
    {                                                                                                     \
        const int t_ = t_first + (TL_);                                                                   \
        const int sb_ = (SB_);                                                                            \
        const bool s2_ = t_ >= nk1;                                                                       \
        const int k0_ = (s2_ ? t_ - nk1 : t_) * DBK;                                                      \
        const int Ks_ = s2_ ? p.K2 : p.K;                                                                 \
        _Pragma("unroll") for (int i = 0; i < NPW; ++i) {                                                 \
            if (i < PAW || i - PAW < nb) {                                                                \
                const bool ok_ = rok[i] && (k0_ + kc[i] < Ks_);                                           \
                const bf16_t* src_ = (i < PAW ? (s2_ ? baseA2 + off2[i] : baseA1 + off1[i]) : (s2_ ? baseB2 + off2[i] : baseB1 + off1[i])) + k0_;   \
                src_ = ok_ ? src_ : zero;                                                                 \
                if (i < PAW) DMA_A(src_, &lds[sb_ + ldso[i]]);                                            \
                else DMA_B(src_, &lds[sb_ + ldso[i]]);                                                    \
            }                                                                                             \
        }                                                                                                 \
    }
    // fast staging for the K slots wholly inside the first K segment: carried per-piece source pointers (one 64-bit add per
    // piece; 0 advance for rows that read the zero page).  Must be called for local slots 0, 1, 2, ... in order.
    const int nfast = max(0, min(nk, p.K / DBK - t_first));
    const bf16_t* fptr[NPW];
    int fadv[NPW];
#pragma unroll
    for (int i = 0; i < NPW; ++i) {
        const bf16_t* b_ = i < PAW ? baseA1 : baseB1;
        fptr[i] = rok[i] ? b_ + (long)t_first * DBK + off1[i] : zero;
        fadv[i] = rok[i] ? DBK : 0;
    }
#define FSTAGE(SB_)                                                                                       \
    {                                                                                                     \
        const int sbf_ = (SB_);                                                                           \
        _Pragma("unroll") for (int i = 0; i < PAW; ++i) {                                                 \
            DMA_A(fptr[i], &lds[sbf_ + ldso[i]]);                                                         \
            fptr[i] += fadv[i];                                                                           \
        }                                                                                                 \
        _Pragma("unroll") for (int j = 0; j < PBW; ++j) {                                                 \
            if (j < nb) {                                                                                 \
                DMA_B(fptr[PAW + j], &lds[sbf_ + ldso[PAW + j]]);                                         \
                fptr[PAW + j] += fadv[PAW + j];                                                           \
            }                                                                                             \
        }                                                                                                 \
    }
#define XSTAGE(TL_, SB_)                                                                                  \
    {                                                                                                     \
        if ((TL_) < nfast) FSTAGE(SB_) else RSTAGE(TL_, SB_)                                              \
    }
    // counted wait: leave the youngest KEEP_ slots (PAW + nb LDS-DMA instructions of THIS wave each) in flight
#define WAIT_KEEP(KEEP_)                                                                                  \
    {                                                                                                     \
        if (nb == PBW) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((PAW + PBW) * (KEEP_)) : "memory");       \
        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((PAW + PBW - 1) * (KEEP_)) : "memory");             \
    }

    f32x4_t acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // prologue: slots 0 .. NS-1 in flight (the carried pointers need the slots staged in order), slot 0 retired
#pragma unroll
    for (int t = 0; t < NS; ++t)
        if (t < nk) XSTAGE(t, t * SLOT_ELEMS);
    if (nk >= NS) WAIT_KEEP(NS - 1)
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    // fragment element offsets inside a slot for the first 32-wide k step (chunks 0-3); the second step is chunk ^ 4 = + 32 elements
    // XOR-ed in, which for this swizzle (chunk ^ ((row >> 1) & 7)) is again a plain XOR with 32 on the element offset
    const int fr = lane & 15, fg = lane >> 4;
    int wofs[TN], xofs[TM];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int row = ni * 16 + fr;
        wofs[ni] = BM * DBK + row * DBK + ((fg ^ ((row >> 1) & 7)) << 3);
    }
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int row = wave * 32 + mi * 16 + fr;
        xofs[mi] = row * DBK + ((fg ^ ((row >> 1) & 7)) << 3);
    }
#define READ_FRAGS(W_, X_, BASE_, KS_)                                                                    \
    {                                                                                                     \
        const bf16_t* st_ = &lds[(BASE_)];                                                                \
        _Pragma("unroll") for (int ni = 0; ni < TN; ++ni) W_[ni] = *reinterpret_cast<const bf16x8_t*>(st_ + (wofs[ni] ^ ((KS_) * 32)));   \
        _Pragma("unroll") for (int mi = 0; mi < TM; ++mi) X_[mi] = *reinterpret_cast<const bf16x8_t*>(st_ + (xofs[mi] ^ ((KS_) * 32)));   \
    }
#define MFMA_GROUP(W_, X_, NI0_, NI1_)                                                                    \
    {                                                                                                     \
        _Pragma("unroll") for (int ni = (NI0_); ni < (NI1_); ++ni)                                        \
            _Pragma("unroll") for (int mi = 0; mi < TM; ++mi)                                             \
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W_[ni], X_[mi], acc[ni][mi], 0, 0, 0);   \
    }

    bf16x8_t w0[TN], x0[TM], w1[TN], x1[TM];
    READ_FRAGS(w0, x0, 0, 0)                                    // k step 0 of slot 0
    int sl = 0;                                                 // t % NS, carried
    int t = 0;
    // ---- steady state: the refill slot lies wholly inside the first K segment (carried pointers, no conditions).  The LDS-DMA
    // instructions of the refill are INTERLEAVED with the MFMAs of k step 1 (one piece per output column tile), so that their issue
    // time - ~50 cycles each, 5-6 per wave per slot - runs under the matrix pipe instead of in front of it (issued as a burst right
    // after the barrier they left the pipe idle on both waves of a SIMD: the two run in lockstep)
    const int n_steady = max(0, nfast - NS);
    for (; t < n_steady; ++t) {
        const int base = sl * SLOT_ELEMS;
        MFMA_GROUP(w0, x0, 0, TN / 3)
        __builtin_amdgcn_sched_barrier(0);
        READ_FRAGS(w1, x1, base, 1)
        __builtin_amdgcn_sched_barrier(0);
        MFMA_GROUP(w0, x0, TN / 3, TN)
        __builtin_amdgcn_sched_barrier(0);
        WAIT_KEEP(NS - 2)
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int nsl = sl + 1 == NS ? 0 : sl + 1;
        READ_FRAGS(w0, x0, nsl * SLOT_ELEMS, 0)                 // k step 0 of slot t+1, one MFMA group ahead
        __builtin_amdgcn_sched_barrier(0);
        // (literal piece indices: a loop-indexed form put the carried pointers into scratch)
#define PIECE(I_)                                                                                         \
    {                                                                                                     \
        if constexpr ((I_) < PAW) {                                                                       \
            DMA_A(fptr[I_], &lds[base + ldso[I_]]);                                                       \
            fptr[I_] += fadv[I_];                                                                         \
        } else if constexpr ((I_) < NPW) {                                                                \
            if ((I_) - PAW < nb) {                                                                        \
                DMA_B(fptr[I_], &lds[base + ldso[I_]]);                                                   \
                fptr[I_] += fadv[I_];                                                                     \
            }                                                                                             \
        }                                                                                                 \
    }
#define STEP(NI_) MFMA_GROUP(w1, x1, NI_, (NI_) + 1)
#define SB __builtin_amdgcn_sched_barrier(0);
        static_assert((TN == 6 && NPW == 6) || (TN == 4 && NPW == 5) || (TN == 10 && NPW == 7), "piece schedule below is written for BN 96 / 64 / 160");
        if constexpr (TN == 6) {
            STEP(0) PIECE(0) SB STEP(1) PIECE(1) SB STEP(2) PIECE(2) SB STEP(3) PIECE(3) SB STEP(4) PIECE(4) SB STEP(5) PIECE(5) SB
        } else if constexpr (TN == 10) {
            STEP(0) PIECE(0) SB STEP(1) PIECE(1) SB STEP(2) SB STEP(3) PIECE(2) SB STEP(4) PIECE(3) SB STEP(5) SB STEP(6) PIECE(4) SB STEP(7) PIECE(5) SB
            STEP(8) PIECE(6) SB STEP(9) SB
        } else {
            STEP(0) PIECE(0) SB STEP(1) PIECE(1) SB STEP(2) PIECE(2) PIECE(3) SB STEP(3) PIECE(4) SB
        }
#undef PIECE
#undef STEP
#undef SB
        sl = nsl;
    }
    // ---- tail: K tail / second K segment slots (generic staging) and the drain
    for (; t < nk; ++t) {
        const int base = sl * SLOT_ELEMS;
        // k step 1 of this slot: its reads are issued BEHIND the first MFMAs of k step 0, so that the compiler's wait for w0 / x0
        // (read one group ago, long complete; it emits lgkmcnt(0) at this loop header whatever the order) does not also wait for
        // them, and they complete under the remaining MFMAs
        MFMA_GROUP(w0, x0, 0, TN / 3)
        __builtin_amdgcn_sched_barrier(0);
        READ_FRAGS(w1, x1, base, 1)
        __builtin_amdgcn_sched_barrier(0);
        MFMA_GROUP(w0, x0, TN / 3, TN)
        __builtin_amdgcn_sched_barrier(0);
        // slot t+1 landed (this wave's pieces), every fragment read of slot t drained -> barrier(t)
        if (t + NS - 1 < nk) { WAIT_KEEP(NS - 2) }
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (t + NS < nk) XSTAGE(t + NS, base)                   // ring slot of slot t is free now (both k steps are in registers)
        const int nsl = sl + 1 == NS ? 0 : sl + 1;
        if (t + 1 < nk) READ_FRAGS(w0, x0, nsl * SLOT_ELEMS, 0) // k step 0 of slot t+1, one MFMA group ahead
        MFMA_GROUP(w1, x1, 0, TN)
        sl = nsl;
    }
#undef RSTAGE
#undef FSTAGE
#undef XSTAGE
#undef WAIT_KEEP
#undef READ_FRAGS
#undef MFMA_GROUP
#undef DMA_A
#undef DMA_B

    const int m_wave = wave * 32;
    if (p.splitk > 1) {          // raw fp32 partial tile; reduced in a fixed slice order by the split-K epilogue kernels (gemm.hip)
        float* part = p.part + (long)sk * p.M * p.N;
        const bool v4 = (p.N & 3) == 0;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int m = m_wave + mi * 16 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int n = n0 + ni * 16 + fg * 4;
                if (n >= p.N) continue;
                float* o = part + (long)m * p.N + n;
                if (v4) *reinterpret_cast<f32x4_t*>(o) = acc[ni][mi];
                else
                    for (int r = 0; r < 4 && n + r < p.N; ++r) o[r] = acc[ni][mi][r];
            }
        }
        return;
    }
    gemm_epilogue<TM, TN>(acc, p.act, m_wave, n0, fr, fg, p.M, p.N, p.bias, p.R, p.ldr, p.res_scale, p.C, 0, p.ldc, p.c_fp32);
}


// ---------------------------------------------------------------------------------------------------------------------------------
// Producer / consumer form of the same kernel: 12 waves, 3 per SIMD.  Waves 0-7 are the consumers of gemm_dec_kernel (8(M) x 1(N),
// 32 rows x BN columns each, same ring, same fragment schedule) and issue NO vector-memory instruction; waves 8-11 - one per SIMD -
// do nothing but stage: 8 activation pieces + BN/32 weight pieces per slot each.
//
// Why (profiles/README.md, "where a slot's cycles go"): in the 8-wave kernel the phase of a slot that carries the refill - 12 MFMAs
// interleaved with 5-6 LDS-DMA instructions per wave - takes 970 ticks on the younger wave of a SIMD against 305 for the same
// MFMAs without them: an LDS-DMA instruction waits ~50 ticks for the CU's address path, a wave issues in order, so the MFMAs behind
// it wait too, and both waves of a SIMD are in that phase together.  The counted vmcnt wait, by contrast, is 5 % of the slot: the
// data is never late, its ISSUE is what stalls the matrix pipe.  With the staging moved to waves that have no MFMAs the stall lands
// on a wave with nothing else to do, and the consumers' loop has no vmcnt, no pointer arithmetic and no K-tail branches.
//
// Synchronisation: one s_barrier per slot, joined by all 12 waves.  barrier(t) means (consumers) "every fragment of slot t is in
// registers" and (producers) "my pieces of slot t+1 have landed"; after it the producers refill the ring position of slot t with
// slot t+NS and the consumers read slot t+1.
template <int BN, int NS, bool NTB>
__global__ __launch_bounds__(768) void gemm_dec_ws_kernel(GemmDP p) {
    constexpr int BM = 256;
    constexpr int TM = 2, TN = BN / 16;
    constexpr int PA = BM / 8, PB = BN / 8;                  // 1-KiB pieces (8 rows x 128 B) per slot: activations, weights
    constexpr int NPROD = 4;
    constexpr int PAP = PA / NPROD, PBP = PB / NPROD;        // pieces per producer wave per slot: 8 + 3 (BN 96) or 8 + 2 (BN 64)
    constexpr int NPP = PAP + PBP;
    constexpr int SLOT_ELEMS = (BM + BN) * DBK;
    static_assert(BN % 32 == 0 && NS >= 3 && NS * SLOT_ELEMS * 2 <= 160 * 1024, "tile / ring geometry");
    __shared__ __attribute__((aligned(16))) bf16_t lds[NS * SLOT_ELEMS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sk = (int)blockIdx.y;
    const int n0 = (int)blockIdx.x * BN;
    const int nk1 = (p.K + DBK - 1) / DBK;
    const int nk2 = p.A2 ? (p.K2 + DBK - 1) / DBK : 0;
    const int nk_per = (nk1 + nk2 + p.splitk - 1) / p.splitk;
    const int t_first = sk * nk_per;                                  // first K slot of this slice (both K segments chained)
    const int nk = min(nk1 + nk2, t_first + nk_per) - t_first;       // >= 1 by construction of splitk (host)

    if (wave >= 8) {
        // =========================================================== producers
        const int pw = wave - 8;
        const int prow = lane >> 3, pc = lane & 7;
        const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_dec);
        // piece i < PAP: activation rows (pw * PAP + i) * 8 ..; piece PAP + j: weight rows (pw + NPROD j) * 8 ..
        int off1[NPP], off2[NPP], kc[NPP], ldso[NPP];
        bool rok[NPP];
#pragma unroll
        for (int i = 0; i < NPP; ++i) {
            const bool isA = i < PAP;
            const int pr0 = isA ? (pw * PAP + i) * 8 : (pw + NPROD * (i - PAP)) * 8;
            const int row = pr0 + prow;
            const int c = pc ^ ((row >> 1) & 7);                         // inverse swizzle on the source chunk
            kc[i] = c * 8;
            rok[i] = isA ? row < p.M : n0 + row < p.N;
            off1[i] = row * (int)(isA ? p.lda : p.ldb) + c * 8;
            off2[i] = row * (int)(isA ? p.lda2 : p.ldb2) + c * 8;
            ldso[i] = __builtin_amdgcn_readfirstlane((isA ? 0 : BM * DBK) + pr0 * DBK);
        }
        const bf16_t* baseA1 = p.A;
        const bf16_t* baseB1 = p.B + (long)n0 * p.ldb;
        const bf16_t* baseA2 = p.A2 ? p.A2 : zero;
        const bf16_t* baseB2 = p.A2 ? p.B2 + (long)n0 * p.ldb2 : zero;
#define DMA_A(SRC_, DST_) __builtin_amdgcn_global_load_lds((gbl_vptr)(SRC_), (lds_vptr)(DST_), 16, 0, 0)
#define DMA_B(SRC_, DST_)                                                                                 \
    {                                                                                                     \
        if constexpr (NTB) __builtin_amdgcn_global_load_lds((gbl_vptr)(SRC_), (lds_vptr)(DST_), 16, 0, 2);  \
        else __builtin_amdgcn_global_load_lds((gbl_vptr)(SRC_), (lds_vptr)(DST_), 16, 0, 0);               \
    }
        // generic staging (K tails, second K segment, rows outside the operands)
#define RSTAGE(TL_, SB_)                                                                                  \
    {                                                                                                     \
        const int t_ = t_first + (TL_);                                                                   \
        const int sb_ = (SB_);                                                                            \
        const bool s2_ = t_ >= nk1;                                                                       \
        const int k0_ = (s2_ ? t_ - nk1 : t_) * DBK;                                                      \
        const int Ks_ = s2_ ? p.K2 : p.K;                                                                 \
        _Pragma("unroll") for (int i = 0; i < NPP; ++i) {                                                 \
            const bool ok_ = rok[i] && (k0_ + kc[i] < Ks_);                                               \
            const bf16_t* src_ = (i < PAP ? (s2_ ? baseA2 + off2[i] : baseA1 + off1[i]) : (s2_ ? baseB2 + off2[i] : baseB1 + off1[i])) + k0_;   \
            src_ = ok_ ? src_ : zero;                                                                     \
            if (i < PAP) DMA_A(src_, &lds[sb_ + ldso[i]]);                                                \
            else DMA_B(src_, &lds[sb_ + ldso[i]]);                                                        \
        }                                                                                                 \
    }
        // fast staging for the K slots wholly inside the first K segment: carried per-piece source pointers, slots in order
        const int nfast = max(0, min(nk, p.K / DBK - t_first));
        const bf16_t* fptr[NPP];
        int fadv[NPP];
#pragma unroll
        for (int i = 0; i < NPP; ++i) {
            const bf16_t* b_ = i < PAP ? baseA1 : baseB1;
            fptr[i] = rok[i] ? b_ + (long)t_first * DBK + off1[i] : zero;
            fadv[i] = rok[i] ? DBK : 0;
        }
#define FSTAGE(SB_)                                                                                       \
    {                                                                                                     \
        const int sbf_ = (SB_);                                                                           \
        _Pragma("unroll") for (int i = 0; i < NPP; ++i) {                                                 \
            if (i < PAP) DMA_A(fptr[i], &lds[sbf_ + ldso[i]]);                                            \
            else DMA_B(fptr[i], &lds[sbf_ + ldso[i]]);                                                    \
            fptr[i] += fadv[i];                                                                           \
        }                                                                                                 \
    }
#define XSTAGE(TL_, SB_)                                                                                  \
    {                                                                                                     \
        if ((TL_) < nfast) FSTAGE(SB_) else RSTAGE(TL_, SB_)                                              \
    }
#pragma unroll
        for (int t = 0; t < NS; ++t)
            if (t < nk) XSTAGE(t, t * SLOT_ELEMS)
        if (nk >= NS) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPP * (NS - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                               // slot 0 is there
        int sl = 0;
        for (int t = 0; t < nk; ++t) {
            // my pieces of slot t+1: at most the NS-2 younger slots may still be in flight
            if (t + NS - 1 < nk) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPP * (NS - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                           // barrier(t): slot t is in the consumers' registers
            if (t + NS < nk) XSTAGE(t + NS, sl * SLOT_ELEMS)
            sl = sl + 1 == NS ? 0 : sl + 1;
        }
#undef RSTAGE
#undef FSTAGE
#undef XSTAGE
#undef DMA_A
#undef DMA_B
        return;
    }

    // =============================================================== consumers
    f32x4_t acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fg = lane >> 4;
    int wofs[TN], xofs[TM];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int row = ni * 16 + fr;
        wofs[ni] = BM * DBK + row * DBK + ((fg ^ ((row >> 1) & 7)) << 3);
    }
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int row = wave * 32 + mi * 16 + fr;
        xofs[mi] = row * DBK + ((fg ^ ((row >> 1) & 7)) << 3);
    }
#define READ_FRAGS(W_, X_, BASE_, KS_)                                                                    \
    {                                                                                                     \
        const bf16_t* st_ = &lds[(BASE_)];                                                                \
        _Pragma("unroll") for (int ni = 0; ni < TN; ++ni) W_[ni] = *reinterpret_cast<const bf16x8_t*>(st_ + (wofs[ni] ^ ((KS_) * 32)));   \
        _Pragma("unroll") for (int mi = 0; mi < TM; ++mi) X_[mi] = *reinterpret_cast<const bf16x8_t*>(st_ + (xofs[mi] ^ ((KS_) * 32)));   \
    }
#define MFMA_GROUP(W_, X_, NI0_, NI1_)                                                                    \
    {                                                                                                     \
        _Pragma("unroll") for (int ni = (NI0_); ni < (NI1_); ++ni)                                        \
            _Pragma("unroll") for (int mi = 0; mi < TM; ++mi)                                             \
                acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W_[ni], X_[mi], acc[ni][mi], 0, 0, 0);   \
    }
    __builtin_amdgcn_s_barrier();                                   // slot 0 is there
    __builtin_amdgcn_sched_barrier(0);
    bf16x8_t w0[TN], x0[TM], w1[TN], x1[TM];
    READ_FRAGS(w0, x0, 0, 0)                                        // k step 0 of slot 0
    int sl = 0;
    for (int t = 0; t < nk; ++t) {
        const int base = sl * SLOT_ELEMS;
        MFMA_GROUP(w0, x0, 0, TN / 3)
        __builtin_amdgcn_sched_barrier(0);
        READ_FRAGS(w1, x1, base, 1)                                 // k step 1, behind the first MFMAs of k step 0
        __builtin_amdgcn_sched_barrier(0);
        MFMA_GROUP(w0, x0, TN / 3, TN)
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // every fragment of slot t is in registers
        __builtin_amdgcn_s_barrier();                               // barrier(t): slot t+1 has landed, slot t may be refilled
        __builtin_amdgcn_sched_barrier(0);
        const int nsl = sl + 1 == NS ? 0 : sl + 1;
        if (t + 1 < nk) READ_FRAGS(w0, x0, nsl * SLOT_ELEMS, 0)     // k step 0 of slot t+1, one MFMA group ahead
        __builtin_amdgcn_sched_barrier(0);
        MFMA_GROUP(w1, x1, 0, TN)
        sl = nsl;
    }
#undef READ_FRAGS
#undef MFMA_GROUP

    const int m_wave = wave * 32;
    if (p.splitk > 1) {          // raw fp32 partial tile; reduced in a fixed slice order by the split-K epilogue kernels (gemm.hip)
        float* part = p.part + (long)sk * p.M * p.N;
        const bool v4 = (p.N & 3) == 0;
#pragma unroll
        for (int mi = 0; mi < TM; ++mi) {
            const int m = m_wave + mi * 16 + fr;
            if (m >= p.M) continue;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                const int n = n0 + ni * 16 + fg * 4;
                if (n >= p.N) continue;
                float* o = part + (long)m * p.N + n;
                if (v4) *reinterpret_cast<f32x4_t*>(o) = acc[ni][mi];
                else
                    for (int r = 0; r < 4 && n + r < p.N; ++r) o[r] = acc[ni][mi][r];
            }
        }
        return;
    }
    gemm_epilogue<TM, TN>(acc, p.act, m_wave, n0, fr, fg, p.M, p.N, p.bias, p.R, p.ldr, p.res_scale, p.C, 0, p.ldc, p.c_fp32);
}

}  // namespace

// Decode-regime launch (called from crab_gemm_bf16, gemm.hip).  bn in {64, 96}; splitk >= 1 (K slices over blockIdx.y, none empty).
int crab_gemm_dec_launch(crab_ctx* ctx, hipStream_t s, const crab_gemm_desc* d, int bn, int splitk, float* part, int nt_weights) {
    GemmDP p;
    p.A = (const bf16_t*)d->A; p.B = (const bf16_t*)d->B; p.C = d->C; p.bias = (const bf16_t*)d->bias; p.R = (const bf16_t*)d->R;
    p.A2 = (const bf16_t*)d->A2; p.B2 = (const bf16_t*)d->B2;
    p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldr = d->ldr; p.lda2 = d->lda2; p.ldb2 = d->ldb2;
    p.M = d->M; p.N = d->N; p.K = d->K; p.K2 = d->A2 ? d->K2 : 0; p.act = d->act; p.c_fp32 = crab_cflags(d); p.res_scale = d->res_scale;
    p.splitk = splitk > 1 ? splitk : 1; p.part = part;
    const int tiles = (d->N + bn - 1) / bn;
    dim3 grid(tiles, p.splitk);
    static const int ws_on = []() { const char* e = getenv("CRAB_DEC_WS"); return !(e && e[0] == '0'); }();
    // nt_weights: 1 = the shipped form (producer / consumer, non-temporal weight loads), 2 = the 8-wave kernel (tune 8xxxx), 0 = the
    // 8-wave kernel with default-policy weight loads
    if (bn == 160) {
        // 160-wide panels (a projection too wide for one round of 96-wide panels): 80 accumulator + 96 fragment VGPRs per consumer wave do
        // not fit three waves per SIMD (108 spills in the producer / consumer form): the 8-wave form, non-temporal weight loads
        hipLaunchKernelGGL((gemm_dec_kernel<160, 3, true>), grid, dim3(512), 0, s, p);
        return crab_check_launch(ctx, "gemm_dec_kernel<160>");
    }
    if (ws_on && nt_weights == 1) {                     // CRAB_DEC_WS=0: the 8-wave kernel process-wide (A/B runs)
        if (bn == 96) hipLaunchKernelGGL((gemm_dec_ws_kernel<96, 3, true>), grid, dim3(768), 0, s, p);
        else if (bn == 64) hipLaunchKernelGGL((gemm_dec_ws_kernel<64, 4, true>), grid, dim3(768), 0, s, p);
        else return crab_fail(ctx, CRAB_E_INVALID, "gemm_dec: bn must be 64, 96 or 160");
        return crab_check_launch(ctx, "gemm_dec_ws_kernel");
    }
    if (bn == 96) {
        if (nt_weights) hipLaunchKernelGGL((gemm_dec_kernel<96, 3, true>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((gemm_dec_kernel<96, 3, false>), grid, dim3(512), 0, s, p);
    } else if (bn == 64) {
        if (nt_weights) hipLaunchKernelGGL((gemm_dec_kernel<64, 4, true>), grid, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((gemm_dec_kernel<64, 4, false>), grid, dim3(512), 0, s, p);
    } else {
        return crab_fail(ctx, CRAB_E_INVALID, "gemm_dec: bn must be 64 or 96");
    }
    return crab_check_launch(ctx, "gemm_dec_kernel");
}

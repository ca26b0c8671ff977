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
    int splitk;          // > 1: blockIdx.y = K slice, raw fp32 partial tiles to `part` [slice][Mtot][N]
    float* part;
    // ROW GROUPS (256 < M <= 512): groups == 2 runs the kernel once per 256-row group INSIDE one launch.  The 1-D block index is decoded so that
    // the two blocks of a weight panel (group 0 / group 1) are 8 apart in dispatch order: the same XCD (workgroups go round-robin over the 8
    // XCDs), resident at the same time on two of its CUs, streaming the SAME weight addresses in step - the second reader of a line finds it
    // in that XCD's L2 (or merges with the miss in flight), so a panel costs one HBM pass for 512 rows.  Mtot = rows of the whole problem.
    int groups, Mtot;
};

// block index -> (panel, row group) for GemmDP.groups == 2 (grid.x = 2 * round_up(panels, 8)); identity otherwise
__device__ __forceinline__ void dec_block_coords(const GemmDP& p, int& panel, int& group) {
    const int L = (int)blockIdx.x;
    if (p.groups > 1) { group = (L >> 3) & 1; panel = ((L >> 4) << 3) | (L & 7); }
    else { group = 0; panel = L; }
}

// the kernel's view of ONE row group: operands, outputs and the slab rows moved to the group's first row (wave-uniform scalar arithmetic)
__device__ __forceinline__ GemmDP dec_group_view(const GemmDP& q, int group) {
    GemmDP p = q;
    const int m0 = group * 256;
    p.M = min(256, q.Mtot - m0);
    if (group) {
        p.A = q.A + (long)m0 * q.lda;
        if (q.A2) p.A2 = q.A2 + (long)m0 * q.lda2;
        p.C = (q.c_fp32 & CF_C32) ? (void*)((float*)q.C + (long)m0 * q.ldc) : (void*)((bf16_t*)q.C + (long)m0 * q.ldc);
        if (q.R) p.R = (q.c_fp32 & CF_R32) ? (const bf16_t*)((const float*)q.R + (long)m0 * q.ldr) : q.R + (long)m0 * q.ldr;
        if (q.part) p.part = q.part + (long)m0 * q.N;
    }
    return p;
}

__device__ __attribute__((aligned(16))) uint32_t g_zero_page_dec[64];      // zero-initialised device memory (256 B)

typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* gbl_vptr;

constexpr int DBK = 64;      // K extent of a ring slot: 128-byte rows, so every LDS-DMA piece (8 rows x 128 B) moves whole cache lines

template <int BN, int NS, bool NTB>
__global__ __launch_bounds__(512) void gemm_dec_kernel(GemmDP pin) {
    int panel_, group_;
    dec_block_coords(pin, panel_, group_);
    if (panel_ * BN >= pin.N) return;                          // padding block of the row-group grid (whole block, before any barrier)
    const GemmDP p = dec_group_view(pin, group_);
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
    const int n0 = panel_ * BN;
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
        float* part = p.part + (long)sk * p.Mtot * p.N;
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
__global__ __launch_bounds__(768) void gemm_dec_ws_kernel(GemmDP pin) {
    int panel_, group_;
    dec_block_coords(pin, panel_, group_);
    if (panel_ * BN >= pin.N) return;                          // padding block of the row-group grid (whole block, before any barrier)
    const GemmDP p = dec_group_view(pin, group_);
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
    const int n0 = panel_ * BN;
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
        float* part = p.part + (long)sk * p.Mtot * p.N;
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
// TWO ROW GROUPS PER BLOCK (256 < M <= 512, r04): one block = one weight panel x BOTH 256-row groups, so a panel is staged ONCE per CU for
// 512 rows.  Why a new kernel and not the pair-of-blocks launch above (GemmDP.groups, kept for A/B): with the two blocks of a panel side by
// side on one XCD the second reader merges with the first reader's miss instead of hitting - every CU still waits one HBM latency per weight
// piece, and a CU's throughput is its bytes in flight / latency (DESIGN.md 3 machine model), so halving the HBM traffic bought 6 %
// (scripts/bench_dec_gemm_groups.py, profiles/README.md r04).  Here the weight bytes a CU takes in per MFMA are halved.
//
// LDS: an ACTIVATION ring of 3 entries of [256 rows][64 k] (32 KiB each) that the sub-slots u = 2 t + g (K slot t, row group g) walk in
// order, and a WEIGHT ring of NW = 4 entries of [BN][64 k]: the weights of slot t stay put for both sub-slots and are prefetched NW slots
// (6+ sub-slot times, several HBM latencies) ahead, the activations (L2 resident, ~0.5 us) 3 sub-slots ahead.  144 KiB at BN = 96.
// 8 waves as 8(M) x 1(N): wave w owns rows [32 w, 32 w + 32) of EACH group - 2 x (2 x TN) accumulator tiles - and stages 4 activation
// pieces per sub-slot plus 1-2 weight pieces per slot; the inner schedule of a sub-slot is gemm_dec_kernel's slot schedule (fragments read
// one MFMA group ahead, ONE barrier per sub-slot, the LDS-DMA issue interleaved with the MFMAs of k step 1), with the weight fragments of
// k step 0 / 1 kept in registers across the two groups.
//
// Order of this wave's LDS-DMA instructions (vmcnt retires them in order):  G(v) = what is issued after barrier(v) = [weights of slot
// t + NW, only when g(v) == 1] then [activations of sub-slot v + 3].  barrier(u) needs sub-slot u + 1 - issued in G(u - 2) - so the wait
// before it keeps |G(u - 1)| instructions in flight: 4 + nb before an even barrier, 4 before an odd one; the weights it may need (slot
// t + 1 at an odd barrier) were issued 2 NW - 2 groups earlier.
//   RAW: sub-slot u + 1 (and slot t + 1's weights) are read after barrier(u), which every wave reaches after waiting for its own pieces.
//   WAR: after barrier(u) every fragment of sub-slot u is in registers (lgkmcnt(0) before it): its activation entry is refilled; after an
//   odd barrier both groups are done with slot t's weights: that entry is refilled.
template <int BN, bool NTB>
__global__ __launch_bounds__(512) void gemm_dec2_kernel(GemmDP p) {
    constexpr int BM = 256, NA = 3, NW = 4;
    constexpr int TM = 2, TN = BN / 16;
    constexpr int PB = BN / 8;                               // 1-KiB weight pieces (8 rows x 128 B) per slot
    constexpr int PAW = 4;                                   // activation pieces per wave per sub-slot (32 pieces / 8 waves)
    constexpr int PBW = (PB + 7) / 8;                        // weight pieces per wave per slot: waves < PB - 8 (PBW - 1) take PBW
    constexpr int A_ELEMS = BM * DBK, W_ELEMS = BN * DBK;
    constexpr int W_BASE = NA * A_ELEMS;
    static_assert(BN % 16 == 0 && PBW <= 2 && (NA * A_ELEMS + NW * W_ELEMS) * 2 <= 160 * 1024, "tile / ring geometry");
    __shared__ __attribute__((aligned(16))) bf16_t lds[NA * A_ELEMS + NW * W_ELEMS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sk = (int)blockIdx.y;
    const int n0 = (int)blockIdx.x * BN;
    const int M1 = p.Mtot - BM;                                       // rows of group 1 (1 .. 256)
    const int nk1 = (p.K + DBK - 1) / DBK;
    const int nk2 = p.A2 ? (p.K2 + DBK - 1) / DBK : 0;
    const int nk_per = (nk1 + nk2 + p.splitk - 1) / p.splitk;
    const int t_first = sk * nk_per;
    const int nk = min(nk1 + nk2, t_first + nk_per) - t_first;       // >= 1 by construction of splitk (host)
    const int nu = 2 * nk;                                            // sub-slots

    // ---- staging coordinates: activation piece i: rows (wave * 4 + i) * 8 .. of a group; weight piece j: rows (wave + 8 j) * 8 ..
    const int nb = (wave < PB - 8 * (PBW - 1)) ? PBW : PBW - 1;      // weight pieces of this wave (wave-uniform, 0 .. 2)
    const int prow = lane >> 3, pc = lane & 7;
    const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_zero_page_dec);
    int aoff1[PAW], aoff2[PAW], akc[PAW], aldso[PAW];
    bool arok[2][PAW];
#pragma unroll
    for (int i = 0; i < PAW; ++i) {
        const int pr0 = (wave * PAW + i) * 8, row = pr0 + prow;
        const int c = pc ^ ((row >> 1) & 7);                         // inverse swizzle on the source chunk
        akc[i] = c * 8;
        arok[0][i] = true; arok[1][i] = row < M1;                     // group 0 has all 256 rows (M > 256)
        aoff1[i] = row * (int)p.lda + c * 8;
        aoff2[i] = row * (int)p.lda2 + c * 8;
        aldso[i] = __builtin_amdgcn_readfirstlane(pr0 * DBK);
    }
    int boff1[PBW], boff2[PBW], bkc[PBW], bldso[PBW];
    bool brok[PBW];
#pragma unroll
    for (int j = 0; j < PBW; ++j) {
        const int pr0 = (wave + 8 * j) * 8, row = pr0 + prow;
        const int c = pc ^ ((row >> 1) & 7);
        bkc[j] = c * 8;
        brok[j] = j < nb && n0 + row < p.N;
        boff1[j] = row * (int)p.ldb + c * 8;
        boff2[j] = row * (int)p.ldb2 + c * 8;
        bldso[j] = __builtin_amdgcn_readfirstlane(pr0 * DBK);
    }
    const bf16_t* baseA1[2] = {p.A, p.A + (long)BM * p.lda};
    const bf16_t* baseA2[2] = {p.A2 ? p.A2 : zero, p.A2 ? p.A2 + (long)BM * p.lda2 : zero};
    const bf16_t* baseB1 = p.B + (long)n0 * p.ldb;
    const bf16_t* baseB2 = p.A2 ? p.B2 + (long)n0 * p.ldb2 : zero;

#define DMA_A(SRC_, DST_) __builtin_amdgcn_global_load_lds((gbl_vptr)(SRC_), (lds_vptr)(DST_), 16, 0, 0)
#define DMA_B(SRC_, DST_)                                                                                 \
    {                                                                                                     \
        if constexpr (NTB) __builtin_amdgcn_global_load_lds((gbl_vptr)(SRC_), (lds_vptr)(DST_), 16, 0, 2);  \
        else __builtin_amdgcn_global_load_lds((gbl_vptr)(SRC_), (lds_vptr)(DST_), 16, 0, 0);               \
    }
    // generic staging (K tails, second K segment, rows outside the operands): activations of (local slot TL_, group G_) into entry EA_
#define RSTAGE_A(TL_, G_, EA_)                                                                            \
    {                                                                                                     \
        const int t_ = t_first + (TL_);                                                                   \
        const bool s2_ = t_ >= nk1;                                                                       \
        const int k0_ = (s2_ ? t_ - nk1 : t_) * DBK;                                                      \
        const int Ks_ = s2_ ? p.K2 : p.K;                                                                 \
        _Pragma("unroll") for (int i = 0; i < PAW; ++i) {                                                 \
            const bool ok_ = arok[G_][i] && (k0_ + akc[i] < Ks_);                                         \
            const bf16_t* src_ = (s2_ ? baseA2[G_] + aoff2[i] : baseA1[G_] + aoff1[i]) + k0_;             \
            src_ = ok_ ? src_ : zero;                                                                     \
            DMA_A(src_, &lds[(EA_) * A_ELEMS + aldso[i]]);                                                \
        }                                                                                                 \
    }
#define RSTAGE_B(TL_, EW_)                                                                                \
    {                                                                                                     \
        const int t_ = t_first + (TL_);                                                                   \
        const bool s2_ = t_ >= nk1;                                                                       \
        const int k0_ = (s2_ ? t_ - nk1 : t_) * DBK;                                                      \
        const int Ks_ = s2_ ? p.K2 : p.K;                                                                 \
        _Pragma("unroll") for (int j = 0; j < PBW; ++j) {                                                 \
            if (j < nb) {                                                                                 \
                const bool ok_ = brok[j] && (k0_ + bkc[j] < Ks_);                                         \
                const bf16_t* src_ = (s2_ ? baseB2 + boff2[j] : baseB1 + boff1[j]) + k0_;                 \
                src_ = ok_ ? src_ : zero;                                                                 \
                DMA_B(src_, &lds[W_BASE + (EW_) * W_ELEMS + bldso[j]]);                                   \
            }                                                                                             \
        }                                                                                                 \
    }
    // fast staging for the K slots wholly inside the first K segment: carried source pointers, one 64-bit add per piece; every (slot, group)
    // and every slot must be staged in order through these
    const int nfast = max(0, min(nk, p.K / DBK - t_first));
    const bf16_t* fa[2][PAW];
    int fadv_a[2][PAW];
    const bf16_t* fb[PBW];
    int fadv_b[PBW];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
        for (int i = 0; i < PAW; ++i) {
            fa[g][i] = arok[g][i] ? baseA1[g] + (long)t_first * DBK + aoff1[i] : zero;
            fadv_a[g][i] = arok[g][i] ? DBK : 0;
        }
#pragma unroll
    for (int j = 0; j < PBW; ++j) {
        fb[j] = brok[j] ? baseB1 + (long)t_first * DBK + boff1[j] : zero;
        fadv_b[j] = brok[j] ? DBK : 0;
    }
#define FSTAGE_A(G_, EA_)                                                                                 \
    {                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < PAW; ++i) {                                                 \
            DMA_A(fa[G_][i], &lds[(EA_) * A_ELEMS + aldso[i]]);                                           \
            fa[G_][i] += fadv_a[G_][i];                                                                   \
        }                                                                                                 \
    }
#define FSTAGE_B(EW_)                                                                                     \
    {                                                                                                     \
        _Pragma("unroll") for (int j = 0; j < PBW; ++j) {                                                 \
            if (j < nb) {                                                                                 \
                DMA_B(fb[j], &lds[W_BASE + (EW_) * W_ELEMS + bldso[j]]);                                  \
                fb[j] += fadv_b[j];                                                                       \
            }                                                                                             \
        }                                                                                                 \
    }
#define XSTAGE_A(TL_, G_, EA_) { if ((TL_) < nfast) FSTAGE_A(G_, EA_) else RSTAGE_A(TL_, G_, EA_) }
#define XSTAGE_B(TL_, EW_) { if ((TL_) < nfast) FSTAGE_B(EW_) else RSTAGE_B(TL_, EW_) }
    // counted wait (+ every LDS read drained): leave the youngest KEEP_ LDS-DMA instructions of THIS wave in flight
#define WAITK(KEEP_) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(KEEP_) : "memory");
#define WAIT_EVEN() { if (nb == 2) WAITK(PAW + 2) else if (nb == 1) WAITK(PAW + 1) else WAITK(PAW) }
#define WAIT_RT(K_)                                                                                       \
    {                                                                                                     \
        switch (K_) {                                                                                     \
            case 0: WAITK(0) break;                                                                       \
            case 1: WAITK(1) break;                                                                       \
            case 2: WAITK(2) break;                                                                       \
            case 4: WAITK(4) break;                                                                       \
            case 5: WAITK(5) break;                                                                       \
            default: WAITK(6) break;                                                                      \
        }                                                                                                 \
    }

    f32x4_t acc0[TN][TM], acc1[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) { acc0[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; acc1[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

    // ---- prologue, in the steady-state order ... A(1), W(NW-1), A(2):  W(0 .. NW-2), A(0), A(1), W(NW-1), A(2); then sub-slot 0 retired
#pragma unroll
    for (int t = 0; t < NW - 1; ++t)
        if (t < nk) XSTAGE_B(t, t)
    XSTAGE_A(0, 0, 0)
    XSTAGE_A(0, 1, 1)
    if (NW - 1 < nk) XSTAGE_B(NW - 1, NW - 1)
    if (1 < nk) XSTAGE_A(1, 0, 2)
    {
        const int keep = PAW + ((NW - 1 < nk) ? nb : 0) + ((1 < nk) ? PAW : 0);      // A(1) + W(NW-1) + A(2) may still fly
        if (keep == 2 * PAW + 2) WAITK(2 * PAW + 2) else if (keep == 2 * PAW + 1) WAITK(2 * PAW + 1) else if (keep == 2 * PAW) WAITK(2 * PAW)
        else WAITK(0)
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    const int fr = lane & 15, fg = lane >> 4;
    int wofs[TN], xofs[TM];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        const int row = ni * 16 + fr;
        wofs[ni] = W_BASE + row * DBK + ((fg ^ ((row >> 1) & 7)) << 3);
    }
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int row = wave * 32 + mi * 16 + fr;
        xofs[mi] = row * DBK + ((fg ^ ((row >> 1) & 7)) << 3);
    }
#define READ_W(W_, EW_, KS_) { _Pragma("unroll") for (int ni = 0; ni < TN; ++ni) W_[ni] = *reinterpret_cast<const bf16x8_t*>(&lds[((EW_) * W_ELEMS + wofs[ni]) ^ ((KS_) * 32)]); }
#define READ_X(X_, EA_, KS_) { _Pragma("unroll") for (int mi = 0; mi < TM; ++mi) X_[mi] = *reinterpret_cast<const bf16x8_t*>(&lds[((EA_) * A_ELEMS + xofs[mi]) ^ ((KS_) * 32)]); }
#define MFMA_GROUP(ACC_, W_, X_, NI0_, NI1_)                                                              \
    {                                                                                                     \
        _Pragma("unroll") for (int ni = (NI0_); ni < (NI1_); ++ni)                                        \
            _Pragma("unroll") for (int mi = 0; mi < TM; ++mi)                                             \
                ACC_[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W_[ni], X_[mi], ACC_[ni][mi], 0, 0, 0);   \
    }
#define SB __builtin_amdgcn_sched_barrier(0);

    bf16x8_t w0[TN], w1[TN], x0[TM], x1[TM];
    READ_W(w0, 0, 0)
    READ_X(x0, 0, 0)                                            // k step 0 of sub-slot 0
    int ea = 0, ew = 0;                                         // activation entry of sub-slot 2 t, weight entry of slot t
    int t = 0;
    // ---- steady state: every refill of the iteration lies wholly inside the first K segment (carried pointers, no conditions)
    const int n_steady = max(0, nfast - NW);
    for (; t < n_steady; ++t) {
        const int ea1 = ea + 1 == NA ? 0 : ea + 1, ea2 = ea1 + 1 == NA ? 0 : ea1 + 1;
        // ===== sub-slot 2 t (group 0)
        MFMA_GROUP(acc0, w0, x0, 0, TN / 3) SB
        READ_W(w1, ew, 1)
        READ_X(x1, ea, 1) SB
        MFMA_GROUP(acc0, w0, x0, TN / 3, TN) SB
        WAIT_EVEN()
        __builtin_amdgcn_s_barrier(); SB
        READ_X(x0, ea1, 0) SB                                   // k step 0 of sub-slot 2 t + 1 (w0 stays: same slot)
        // G(2 t): activations of sub-slot 2 t + 3 = (slot t + 1, group 1) into the entry sub-slot 2 t leaves, interleaved with the MFMAs
#define PIECE_A(G_, I_, EA_) { DMA_A(fa[G_][I_], &lds[(EA_) * A_ELEMS + aldso[I_]]); fa[G_][I_] += fadv_a[G_][I_]; }
#define PIECE_B(J_, EW_) { if ((J_) < nb) { DMA_B(fb[J_], &lds[W_BASE + (EW_) * W_ELEMS + bldso[J_]]); fb[J_] += fadv_b[J_]; } }
#define STEP(ACC_, NI_) MFMA_GROUP(ACC_, w1, x1, NI_, (NI_) + 1)
        static_assert(TN == 6 || TN == 4, "piece schedule below is written for BN 96 / 64");
        if constexpr (TN == 6) {
            STEP(acc0, 0) PIECE_A(1, 0, ea) SB STEP(acc0, 1) PIECE_A(1, 1, ea) SB STEP(acc0, 2) PIECE_A(1, 2, ea) SB STEP(acc0, 3) PIECE_A(1, 3, ea) SB
            STEP(acc0, 4) SB STEP(acc0, 5) SB
        } else {
            STEP(acc0, 0) PIECE_A(1, 0, ea) SB STEP(acc0, 1) PIECE_A(1, 1, ea) SB STEP(acc0, 2) PIECE_A(1, 2, ea) SB STEP(acc0, 3) PIECE_A(1, 3, ea) SB
        }
        // ===== sub-slot 2 t + 1 (group 1)
        MFMA_GROUP(acc1, w0, x0, 0, TN / 3) SB
        READ_X(x1, ea1, 1) SB                                   // (w1 stays)
        MFMA_GROUP(acc1, w0, x0, TN / 3, TN) SB
        WAITK(PAW)
        __builtin_amdgcn_s_barrier(); SB
        const int ew1 = ew + 1 == NW ? 0 : ew + 1;
        READ_W(w0, ew1, 0)
        READ_X(x0, ea2, 0) SB                                   // k step 0 of sub-slot 2 t + 2
        // G(2 t + 1): weights of slot t + NW into the entry slot t leaves, THEN activations of sub-slot 2 t + 4 = (slot t + 2, group 0)
        if constexpr (TN == 6) {
            STEP(acc1, 0) PIECE_B(0, ew) SB STEP(acc1, 1) PIECE_B(1, ew) SB STEP(acc1, 2) PIECE_A(0, 0, ea1) SB STEP(acc1, 3) PIECE_A(0, 1, ea1) SB
            STEP(acc1, 4) PIECE_A(0, 2, ea1) SB STEP(acc1, 5) PIECE_A(0, 3, ea1) SB
        } else {
            STEP(acc1, 0) PIECE_B(0, ew) SB STEP(acc1, 1) PIECE_A(0, 0, ea1) PIECE_A(0, 1, ea1) SB STEP(acc1, 2) PIECE_A(0, 2, ea1) SB STEP(acc1, 3) PIECE_A(0, 3, ea1) SB
        }
        ea = ea2; ew = ew1;
    }
    // ---- tail: K tail / second K segment slots (generic staging, run-time wait counts) and the drain
    for (; t < nk; ++t) {
        const int ea1 = ea + 1 == NA ? 0 : ea + 1, ea2 = ea1 + 1 == NA ? 0 : ea1 + 1;
        const int u = 2 * t;
        // ===== sub-slot u (group 0).  G(u - 1) = [W(t - 1 + NW)] + [A(u + 2)]
        MFMA_GROUP(acc0, w0, x0, 0, TN / 3) SB
        READ_W(w1, ew, 1)
        READ_X(x1, ea, 1) SB
        MFMA_GROUP(acc0, w0, x0, TN / 3, TN) SB
        {
            const int keep = ((t >= 1 && t - 1 + NW < nk) ? nb : 0) + ((u + 2 < nu && u >= 1) ? PAW : 0);
            WAIT_RT(keep)
        }
        __builtin_amdgcn_s_barrier(); SB
        READ_X(x0, ea1, 0) SB
        if (u + 3 < nu) XSTAGE_A(t + 1, 1, ea)                  // G(u): A(u + 3) = (slot t + 1, group 1)
        MFMA_GROUP(acc0, w1, x1, 0, TN) SB
        // ===== sub-slot u + 1 (group 1).  G(u) = [A(u + 3)]
        MFMA_GROUP(acc1, w0, x0, 0, TN / 3) SB
        READ_X(x1, ea1, 1) SB
        MFMA_GROUP(acc1, w0, x0, TN / 3, TN) SB
        {
            const int keep = (u + 3 < nu) ? PAW : 0;
            WAIT_RT(keep)
        }
        __builtin_amdgcn_s_barrier(); SB
        const int ew1 = ew + 1 == NW ? 0 : ew + 1;
        if (t + 1 < nk) {
            READ_W(w0, ew1, 0)
            READ_X(x0, ea2, 0)
        }
        SB
        if (t + NW < nk) XSTAGE_B(t + NW, ew)                   // G(u + 1): W(t + NW), then A(u + 4) = (slot t + 2, group 0)
        if (u + 4 < nu) XSTAGE_A(t + 2, 0, ea1)
        MFMA_GROUP(acc1, w1, x1, 0, TN) SB
        ea = ea2; ew = ew1;
    }
#undef PIECE_A
#undef PIECE_B
#undef STEP
#undef SB
#undef READ_W
#undef READ_X
#undef MFMA_GROUP
#undef WAITK
#undef WAIT_EVEN
#undef WAIT_RT
#undef XSTAGE_A
#undef XSTAGE_B
#undef FSTAGE_A
#undef FSTAGE_B
#undef RSTAGE_A
#undef RSTAGE_B
#undef DMA_A
#undef DMA_B

    // ---- epilogue, one row group at a time (the group's view of the operands: dec_group_view)
    const int m_wave = wave * 32;
#pragma unroll
    for (int g = 0; g < 2; ++g) {
        const GemmDP q = dec_group_view(p, g);
        const f32x4_t (&acc)[TN][TM] = g ? acc1 : acc0;
        if (p.splitk > 1) {      // raw fp32 partial tile; reduced in a fixed slice order by the split-K epilogue kernels (gemm.hip)
            float* part = q.part + (long)sk * p.Mtot * p.N;
            const bool v4 = (p.N & 3) == 0;
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                const int m = m_wave + mi * 16 + fr;
                if (m >= q.M) continue;
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
        } else {
            gemm_epilogue<TM, TN>(acc, q.act, m_wave, n0, fr, fg, q.M, q.N, q.bias, q.R, q.ldr, q.res_scale, q.C, 0, q.ldc, q.c_fp32);
        }
    }
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
    p.Mtot = d->M; p.groups = d->M > 256 ? 2 : 1;
    if (d->M > 512) return crab_fail(ctx, CRAB_E_INVALID, "gemm_dec: at most 512 rows (two row groups)");
    const int tiles = (d->N + bn - 1) / bn;
    // two row groups: grid.x = 2 x panels rounded up to 8, see GemmDP.groups; the weights then use the DEFAULT cache policy (the
    // non-temporal hint would keep the first reader's lines out of the L2 the second reader is meant to hit)
    dim3 grid(p.groups > 1 ? 2 * ((tiles + 7) / 8 * 8) : tiles, p.splitk);
    static const int pair_form = []() { const char* e = getenv("CRAB_DEC_GROUPS"); return (e && e[0] == 'p') ? 1 : 0; }();     // "pair": the A/B form
    if (p.groups > 1 && !pair_form) {
        // the shipped form for 256 < M <= 512: one block per panel (x K slice) holds BOTH row groups (gemm_dec2_kernel)
        if (bn != 96 && bn != 64) return crab_fail(ctx, CRAB_E_UNSUPPORTED, "gemm_dec: the two-row-group kernel has 96- and 64-wide panels");
        dim3 grid2(tiles, p.splitk);
        p.groups = 1;                                       // (plain block index; dec_group_view is applied per group in the epilogue)
        if (bn == 96) hipLaunchKernelGGL((gemm_dec2_kernel<96, true>), grid2, dim3(512), 0, s, p);
        else hipLaunchKernelGGL((gemm_dec2_kernel<64, true>), grid2, dim3(512), 0, s, p);
        return crab_check_launch(ctx, "gemm_dec2_kernel");
    }
    if (p.groups > 1) {
        static const int nt2 = []() { const char* e = getenv("CRAB_DEC_GROUPS_NT"); return (e && e[0] == '1') ? 1 : 0; }();
        if (bn == 160) return crab_fail(ctx, CRAB_E_UNSUPPORTED, "gemm_dec: 160-wide panels have no row-group form");
        if (nt2) {
            if (bn == 96) hipLaunchKernelGGL((gemm_dec_ws_kernel<96, 3, true>), grid, dim3(768), 0, s, p);
            else hipLaunchKernelGGL((gemm_dec_ws_kernel<64, 4, true>), grid, dim3(768), 0, s, p);
        } else {
            if (bn == 96) hipLaunchKernelGGL((gemm_dec_ws_kernel<96, 3, false>), grid, dim3(768), 0, s, p);
            else hipLaunchKernelGGL((gemm_dec_ws_kernel<64, 4, false>), grid, dim3(768), 0, s, p);
        }
        return crab_check_launch(ctx, "gemm_dec_ws_kernel(row groups)");
    }
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

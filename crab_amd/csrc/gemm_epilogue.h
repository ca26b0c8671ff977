// Shared output stage of the MFMA GEMM kernels (gemm.hip, gemm_glds.hip).
//
// Accumulator layout (operands swapped, see gemm.hip): acc[ni][mi][r] is row m_lane + 16*mi, column n_lane + 16*ni + r,
// with m_lane = tile row + (lane & 15) and n_lane = tile column + 4 * (lane >> 4).
//
// Two forms:
//   gemm_epilogue_full    : the wave's whole sub-tile is in range and 8-byte aligned -> no guards, fully unrolled, the
//                           ACTIVATION IS A TEMPLATE PARAMETER.  With a runtime `act` every one of the TM*TN*4 inlined
//                           apply_act() switches (erff, two exps) stays in the instruction stream even when act == NONE:
//                           the 256x256 kernel grew to 29k instructions (~170 KB, the instruction cache is 64 KB per CU
//                           pair) and its no-activation epilogue spent 10-15k cycles jumping over dead code
//                           (s_memtime probe, profiles/README.md).
//   gemm_epilogue_guarded : ragged edges / unaligned outputs, per-element guards (same activation dispatch).
#pragma once
#include "common.h"

namespace crab_epi {

template <int ACT>
__device__ __forceinline__ float act_c(float x) {
    if (ACT == ACT_GELU) return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));
    if (ACT == ACT_QUICK_GELU) return x / (1.0f + __expf(-1.702f * x));
    if (ACT == ACT_RELU) return fmaxf(x, 0.0f);
    if (ACT == ACT_SILU) return x / (1.0f + __expf(-x));
    return x;
}

// WIDE (bf16 C, 16-byte aligned rows): the natural store of this fragment layout is 8 bytes per lane (4 bytes with the SwiGLU pair
// epilogue) - 16 rows x 32 (16) bytes per wave instruction - and the store tail of a 256x256 tile is issue-bound (cdna guide T21).  Lanes
// 16 / 32 apart hold the neighbouring column groups of the same row, so v_permlane16_swap / v_permlane32_swap regroup them into ONE
// 16-byte store per lane: a pair of column tiles -> two swaps, the four SwiGLU column tiles of a row -> a 4x4 transpose in four swaps.
template <int TM, int TN, int ACT, bool WIDE>
__device__ __forceinline__ void full_impl(const f32x4_t (&acc)[TN][TM], int m_lane, int n_lane, const bf16_t* bias, const bf16_t* R, long ldr,
                                          float rs, void* Cv, long coff, long ldc, int c_fp32) {
    float bv[TN][4];
#pragma unroll
    for (int ni = 0; ni < TN; ++ni) {
        u32x2 bw = {0u, 0u};
        if (bias) bw = *reinterpret_cast<const u32x2*>(bias + n_lane + ni * 16);
        bv[ni][0] = lo_bf(bw.x); bv[ni][1] = hi_bf(bw.x); bv[ni][2] = lo_bf(bw.y); bv[ni][3] = hi_bf(bw.y);
    }
    const int fg = threadIdx.x >> 4 & 3;                    // n_lane = tile column + 4 * fg
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const long rowC = coff + (long)(m_lane + mi * 16) * ldc + n_lane;
        const long rowR = (long)(m_lane + mi * 16) * ldr + n_lane;
        if constexpr (ACT == ACT_SWIGLU_PAIR && WIDE && TN == 4) {
            uint32_t v[4];
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const float g0 = acc[ni][mi][0] + bv[ni][0], u0 = acc[ni][mi][1] + bv[ni][1];
                const float g1 = acc[ni][mi][2] + bv[ni][2], u1 = acc[ni][mi][3] + bv[ni][3];
                v[ni] = pack_bf2(g0 / (1.0f + __expf(-g0)) * u0, g1 / (1.0f + __expf(-g1)) * u1);
            }
            // 4x4 transpose over (lane row fg, column tile ni): lane fg ends with tile fg's outputs of the four lane rows = 8 columns
            auto a02 = __builtin_amdgcn_permlane32_swap(v[0], v[2], false, false);
            auto a13 = __builtin_amdgcn_permlane32_swap(v[1], v[3], false, false);
            auto b01 = __builtin_amdgcn_permlane16_swap(a02[0], a13[0], false, false);
            auto b23 = __builtin_amdgcn_permlane16_swap(a02[1], a13[1], false, false);
            const long oc = coff + (long)(m_lane + mi * 16) * ldc + ((n_lane - 4 * fg) >> 1) + fg * 8;
            *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(Cv) + oc) = u32x4{b01[0], b01[1], b23[0], b23[1]};
            continue;
        }
        u32x2 pend = {0u, 0u};
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            if (ACT == ACT_SWIGLU_PAIR) {              // columns (gate, up, gate, up) -> two outputs at column n/2 (no residual)
                const float g0 = acc[ni][mi][0] + bv[ni][0], u0 = acc[ni][mi][1] + bv[ni][1];
                const float g1 = acc[ni][mi][2] + bv[ni][2], u1 = acc[ni][mi][3] + bv[ni][3];
                const float o0 = g0 / (1.0f + __expf(-g0)) * u0, o1 = g1 / (1.0f + __expf(-g1)) * u1;
                const long oc = coff + (long)(m_lane + mi * 16) * ldc + ((n_lane + ni * 16) >> 1);
                if (c_fp32 & CF_C32) *reinterpret_cast<float2*>(reinterpret_cast<float*>(Cv) + oc) = make_float2(o0, o1);
                else *reinterpret_cast<uint32_t*>(reinterpret_cast<bf16_t*>(Cv) + oc) = pack_bf2(o0, o1);
                continue;
            }
            float v0 = act_c<ACT>(acc[ni][mi][0] + bv[ni][0]), v1 = act_c<ACT>(acc[ni][mi][1] + bv[ni][1]);
            float v2 = act_c<ACT>(acc[ni][mi][2] + bv[ni][2]), v3 = act_c<ACT>(acc[ni][mi][3] + bv[ni][3]);
            if (R) {
                if (c_fp32 & CF_R32) {              // fp32 residual stream
                    const float4 rr = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(R) + rowR + ni * 16);
                    v0 += rs * rr.x; v1 += rs * rr.y; v2 += rs * rr.z; v3 += rs * rr.w;
                } else {
                    u32x2 rr = *reinterpret_cast<const u32x2*>(R + rowR + ni * 16);
                    v0 += rs * lo_bf(rr.x); v1 += rs * hi_bf(rr.x); v2 += rs * lo_bf(rr.y); v3 += rs * hi_bf(rr.y);
                }
            }
            if (c_fp32 & CF_C32) {
                *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + rowC + ni * 16) = make_float4(v0, v1, v2, v3);
            } else if constexpr (WIDE && TN % 2 == 0) {
                // column tiles ni (even) and ni + 1 are exchanged between the even and odd lane rows: one 16-byte store per pair
                u32x2 oa; oa.x = pack_bf2(v0, v1); oa.y = pack_bf2(v2, v3);
                if ((ni & 1) == 0) { pend.x = oa.x; pend.y = oa.y; continue; }
                auto rx = __builtin_amdgcn_permlane16_swap(pend.x, oa.x, false, false);
                auto ry = __builtin_amdgcn_permlane16_swap(pend.y, oa.y, false, false);
                const long oc = coff + (long)(m_lane + mi * 16) * ldc + (n_lane - 4 * fg) + (ni - 1 + (fg & 1)) * 16 + (fg >> 1) * 8;
                *reinterpret_cast<u32x4*>(reinterpret_cast<bf16_t*>(Cv) + oc) = u32x4{rx[0], ry[0], rx[1], ry[1]};
            } else {
                u32x2 o; o.x = pack_bf2(v0, v1); o.y = pack_bf2(v2, v3);
                *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(Cv) + rowC + ni * 16) = o;
            }
        }
    }
}

}  // namespace crab_epi

template <int TM, int TN, bool WIDE>
__device__ __forceinline__ void gemm_epilogue_full_w(const f32x4_t (&acc)[TN][TM], int act, int m_lane, int n_lane, const bf16_t* bias,
                                                     const bf16_t* R, long ldr, float rs, void* C, long coff, long ldc, int c_fp32) {
    switch (act) {
        case ACT_NONE: crab_epi::full_impl<TM, TN, ACT_NONE, WIDE>(acc, m_lane, n_lane, bias, R, ldr, rs, C, coff, ldc, c_fp32); break;
        case ACT_GELU: crab_epi::full_impl<TM, TN, ACT_GELU, WIDE>(acc, m_lane, n_lane, bias, R, ldr, rs, C, coff, ldc, c_fp32); break;
        case ACT_QUICK_GELU: crab_epi::full_impl<TM, TN, ACT_QUICK_GELU, WIDE>(acc, m_lane, n_lane, bias, R, ldr, rs, C, coff, ldc, c_fp32); break;
        case ACT_RELU: crab_epi::full_impl<TM, TN, ACT_RELU, WIDE>(acc, m_lane, n_lane, bias, R, ldr, rs, C, coff, ldc, c_fp32); break;
        case ACT_SWIGLU_PAIR: crab_epi::full_impl<TM, TN, ACT_SWIGLU_PAIR, WIDE>(acc, m_lane, n_lane, bias, R, ldr, rs, C, coff, ldc, c_fp32); break;
        default: crab_epi::full_impl<TM, TN, ACT_SILU, WIDE>(acc, m_lane, n_lane, bias, R, ldr, rs, C, coff, ldc, c_fp32); break;
    }
}

// requires: rows m_lane .. m_lane+16*TM-1 < M, columns n_lane .. +16*TN-1 < N, ldc/coff (and ldr) multiples of 4
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_full(const f32x4_t (&acc)[TN][TM], int act, int m_lane, int n_lane, const bf16_t* bias,
                                                   const bf16_t* R, long ldr, float rs, void* C, long coff, long ldc, int c_fp32) {
    // 16-byte stores need bf16 C with 16-byte aligned rows (wave-uniform condition)
    const bool wide = !(c_fp32 & CF_C32) && (ldc & 7) == 0 && (coff & 7) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;
    if (wide) gemm_epilogue_full_w<TM, TN, true>(acc, act, m_lane, n_lane, bias, R, ldr, rs, C, coff, ldc, c_fp32);
    else gemm_epilogue_full_w<TM, TN, false>(acc, act, m_lane, n_lane, bias, R, ldr, rs, C, coff, ldc, c_fp32);
}

namespace crab_epi {
template <int TM, int TN, int ACT>
__device__ __forceinline__ void guarded_impl(const f32x4_t (&acc)[TN][TM], int m_lane, int n_lane, int M, int N,
                                                      const bf16_t* bias, const bf16_t* R, long ldr, float rs, void* Cv, long coff, long ldc,
                                                      int c_fp32) {
    const bool vec_ok = ((ldc & 3) == 0) && ((coff & 3) == 0) && (!R || ((ldr & 3) == 0));
#pragma unroll
    for (int mi = 0; mi < TM; ++mi) {
        const int m = m_lane + mi * 16;
        if (m >= M) continue;
#pragma unroll
        for (int ni = 0; ni < TN; ++ni) {
            const int n = n_lane + ni * 16;
            if (n >= N) continue;
            if (ACT == ACT_SWIGLU_PAIR) {              // N % 4 == 0 is checked by the dispatcher: all four columns are in range
                float t[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) t[r] = acc[ni][mi][r] + (bias ? bf2f(bias[n + r]) : 0.f);
                const float o0 = t[0] / (1.0f + __expf(-t[0])) * t[1], o1 = t[2] / (1.0f + __expf(-t[2])) * t[3];
                const long oc = coff + (long)m * ldc + (n >> 1);
                if (c_fp32 & CF_C32) { reinterpret_cast<float*>(Cv)[oc] = o0; reinterpret_cast<float*>(Cv)[oc + 1] = o1; }
                else { reinterpret_cast<bf16_t*>(Cv)[oc] = f2bf(o0); reinterpret_cast<bf16_t*>(Cv)[oc + 1] = f2bf(o1); }
                continue;
            }
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float x = acc[ni][mi][r];
                if (bias && n + r < N) x += bf2f(bias[n + r]);
                v[r] = act_c<ACT>(x);
            }
            if (n + 3 < N && vec_ok) {
                if (R && (c_fp32 & CF_R32)) {
                    const float4 rr = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(R) + (long)m * ldr + n);
                    v[0] += rs * rr.x; v[1] += rs * rr.y; v[2] += rs * rr.z; v[3] += rs * rr.w;
                } else if (R) {
                    u32x2 rr = *reinterpret_cast<const u32x2*>(R + (long)m * ldr + n);
                    v[0] += rs * lo_bf(rr.x); v[1] += rs * hi_bf(rr.x); v[2] += rs * lo_bf(rr.y); v[3] += rs * hi_bf(rr.y);
                }
                if (c_fp32 & CF_C32) {
                    *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + coff + (long)m * ldc + n) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
                    u32x2 o; o.x = pack_bf2(v[0], v[1]); o.y = pack_bf2(v[2], v[3]);
                    *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(Cv) + coff + (long)m * ldc + n) = o;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (n + r >= N) break;
                    float x = v[r];
                    if (R) x += rs * ld_res(R, (long)m * ldr + n + r, c_fp32);
                    if (c_fp32 & CF_C32) reinterpret_cast<float*>(Cv)[coff + (long)m * ldc + n + r] = x;
                    else reinterpret_cast<bf16_t*>(Cv)[coff + (long)m * ldc + n + r] = f2bf(x);
                }
            }
        }
    }
}
}  // namespace crab_epi

template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue_guarded(const f32x4_t (&acc)[TN][TM], int act, int m_lane, int n_lane, int M, int N,
                                                      const bf16_t* bias, const bf16_t* R, long ldr, float rs, void* C, long coff, long ldc,
                                                      int c_fp32) {
    switch (act) {
        case ACT_NONE: crab_epi::guarded_impl<TM, TN, ACT_NONE>(acc, m_lane, n_lane, M, N, bias, R, ldr, rs, C, coff, ldc, c_fp32); break;
        case ACT_GELU: crab_epi::guarded_impl<TM, TN, ACT_GELU>(acc, m_lane, n_lane, M, N, bias, R, ldr, rs, C, coff, ldc, c_fp32); break;
        case ACT_QUICK_GELU: crab_epi::guarded_impl<TM, TN, ACT_QUICK_GELU>(acc, m_lane, n_lane, M, N, bias, R, ldr, rs, C, coff, ldc, c_fp32); break;
        case ACT_RELU: crab_epi::guarded_impl<TM, TN, ACT_RELU>(acc, m_lane, n_lane, M, N, bias, R, ldr, rs, C, coff, ldc, c_fp32); break;
        case ACT_SWIGLU_PAIR: crab_epi::guarded_impl<TM, TN, ACT_SWIGLU_PAIR>(acc, m_lane, n_lane, M, N, bias, R, ldr, rs, C, coff, ldc, c_fp32); break;
        default: crab_epi::guarded_impl<TM, TN, ACT_SILU>(acc, m_lane, n_lane, M, N, bias, R, ldr, rs, C, coff, ldc, c_fp32); break;
    }
}

// one call site per kernel: picks the unguarded form when this WAVE's sub-tile is interior
template <int TM, int TN>
__device__ __forceinline__ void gemm_epilogue(const f32x4_t (&acc)[TN][TM], int act, int m_wave, int n_wave, int fr, int fg, int M, int N,
                                              const bf16_t* bias, const bf16_t* R, long ldr, float rs, void* C, long coff, long ldc,
                                              int c_fp32) {
    const bool full = (m_wave + TM * 16 <= M) && (n_wave + TN * 16 <= N) && ((ldc & 3) == 0) && ((coff & 3) == 0) &&
                      (!R || ((ldr & 3) == 0));
    if (full) gemm_epilogue_full<TM, TN>(acc, act, m_wave + fr, n_wave + fg * 4, bias, R, ldr, rs, C, coff, ldc, c_fp32);
    else gemm_epilogue_guarded<TM, TN>(acc, act, m_wave + fr, n_wave + fg * 4, M, N, bias, R, ldr, rs, C, coff, ldc, c_fp32);
}

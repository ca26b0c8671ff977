// Shared device helpers for the Crab gfx950 kernels (bf16 storage, fp32 arithmetic).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;   // raw bfloat16 bits; torch.bfloat16 storage is passed as-is

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4_t;    // MFMA 16x16 C/D fragment

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;   // 16-byte vector load/store unit
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN preserved: identical to torch's float -> bfloat16
// fp32 -> bf16, round to nearest even: gfx950's packed hardware conversion (v_cvt_pk_bf16_f32, one instruction per
// PAIR; the software add-0x7fff sequence costs ~8 VALU instructions per value and made the 256x256 GEMM epilogue
// VALU-bound: 18k of its 25k cycles, profiles/README.md)
typedef float cvt_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 cvt_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    cvt_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, cvt_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float lo_bf(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float hi_bf(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// Eight consecutive NORM PARAMETERS (LayerNorm weight / bias) starting at element e (a multiple of 8): bf16 storage (one 16-byte load) or,
// WF, fp32 storage (two) - the non-matrix parameters of the encoders may stay in fp32 (crab_ln.fp32): they are not MFMA operands, and their
// bf16 rounding is a systematic 2^-9 relative error on every channel of every LayerNorm output (scripts/parity_floor.py).
// four consecutive parameters (8-byte bf16 load | 16-byte fp32 load)
template <bool WF>
__device__ __forceinline__ void ld_par4(const void* p, long e, float (&o)[4]) {
    if (WF) {
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(p) + e);
        o[0] = a[0]; o[1] = a[1]; o[2] = a[2]; o[3] = a[3];
    } else {
        typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_;
        const u32x2_ v = *reinterpret_cast<const u32x2_*>(reinterpret_cast<const uint16_t*>(p) + e);      // one 8-byte load
        o[0] = lo_bf(v[0]); o[1] = hi_bf(v[0]); o[2] = lo_bf(v[1]); o[3] = hi_bf(v[1]);
    }
}

template <bool WF>
__device__ __forceinline__ void ld_par8(const void* p, long e, float (&o)[8]) {
    if (WF) {
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(p) + e);
        const f32x4_t b = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(p) + e + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[j] = a[j]; o[4 + j] = b[j]; }
    } else {
        const u32x4 v = *reinterpret_cast<const u32x4*>(reinterpret_cast<const uint16_t*>(p) + e);
#pragma unroll
        for (int j = 0; j < 4; ++j) { o[2 * j] = lo_bf(v[j]); o[2 * j + 1] = hi_bf(v[j]); }
    }
}

// Cross-lane exchange inside a 16-lane row by DPP (data-parallel primitives: the exchange rides on a VALU instruction).  The compiler lowers
// __shfl_xor(v, 1..8) to ds_bpermute_b32, an LDS-crossbar instruction: the key loop of the decode-attention kernels issued 4-15 of them
// per key row (ISA, profiles/README.md r03).  lane ^ 1 and ^ 2 are quad permutes, ^ 8 is a rotation of the row by 8, ^ 4 = half-row
// mirror (i -> 7 - i) followed by a quad reversal (j -> j ^ 3).  Same lanes, same values: results are bit-identical to the shuffles.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float row_xor1(float v) { return dpp_f<0xB1>(v); }      // quad_perm [1,0,3,2]
__device__ __forceinline__ float row_xor2(float v) { return dpp_f<0x4E>(v); }      // quad_perm [2,3,0,1]
__device__ __forceinline__ float row_xor4(float v) { return dpp_f<0x1B>(dpp_f<0x141>(v)); }   // row_half_mirror, then quad_perm [3,2,1,0]
__device__ __forceinline__ float row_xor8(float v) { return dpp_f<0x128>(v); }     // row_ror:8
// butterfly sum over the 16 lanes of a row in the order 8, 4, 2, 1 (what `for (off = 8; off; off >>= 1) v += __shfl_xor(v, off)` computes)
__device__ __forceinline__ float row16_sum(float v) {
    v += row_xor8(v);
    v += row_xor4(v);
    v += row_xor2(v);
    v += row_xor1(v);
    return v;
}

// lane ^ 32 and lane ^ 16 partners without the LDS crossbar: v_permlane32_swap / v_permlane16_swap (gfx950) exchange the upper half (the odd
// 16-lane rows) of one register with the lower half (the even rows) of another; with the same value in both, the two results hold, per lane,
// {own, partner} or {partner, own} - their sum / max is the butterfly step, whichever order (commutative: bit-identical to the shuffle form).
// (the two operands must live in DIFFERENT registers - the instruction swaps in place, and with one register for both the halves are merely
// exchanged within it: the empty asm pins a copy)
__device__ __forceinline__ void xor32_pair(float v, float& a, float& b) {
    unsigned u0 = __builtin_bit_cast(unsigned, v), u1 = u0;
    asm volatile("" : "+v"(u1));
    auto r = __builtin_amdgcn_permlane32_swap(u0, u1, false, false);
    const unsigned ra = r[0], rb = r[1];                    // (element reads BEFORE the casts: bit_cast(float, r[1]) compiled to element 0)
    a = __uint_as_float(ra); b = __uint_as_float(rb);
}
__device__ __forceinline__ void xor16_pair(float v, float& a, float& b) {
    unsigned u0 = __builtin_bit_cast(unsigned, v), u1 = u0;
    asm volatile("" : "+v"(u1));
    auto r = __builtin_amdgcn_permlane16_swap(u0, u1, false, false);
    const unsigned ra = r[0], rb = r[1];
    a = __uint_as_float(ra); b = __uint_as_float(rb);
}
// wave-wide butterflies in the order 32, 16, 8, 4, 2, 1: the two cross-row steps by permlane swaps, the four in-row steps by DPP - no LDS
// instruction at all (same pairs, same order: bit-identical to the all-shuffle form)
__device__ __forceinline__ float wave_sum(float v) {
    float a, b;
    xor32_pair(v, a, b); v = a + b;
    xor16_pair(v, a, b); v = a + b;
    return row16_sum(v);
}
__device__ __forceinline__ float wave_max(float v) {
    float a, b;
    xor32_pair(v, a, b); v = fmaxf(a, b);
    xor16_pair(v, a, b); v = fmaxf(a, b);
    v = fmaxf(v, row_xor8(v));
    v = fmaxf(v, row_xor4(v));
    v = fmaxf(v, row_xor2(v));
    v = fmaxf(v, row_xor1(v));
    return v;
}

// Storage flags carried in the GEMM kernels' `c_fp32` parameter word: bit 0 = C is fp32, bit 1 = R is fp32 (crab_gemm_desc.c_fp32 / r_fp32;
// both with R == C = the fp32 residual stream)
enum { CF_C32 = 1, CF_R32 = 2 };
__device__ __forceinline__ float ld_res(const bf16_t* R, long idx, int flags) {
    return (flags & CF_R32) ? reinterpret_cast<const float*>(R)[idx] : bf2f(R[idx]);
}

// activations selectable in GEMM epilogues / elementwise kernels
// ACT_SWIGLU_PAIR: the weight rows are INTERLEAVED (gate_i, up_i): out[m, j] = silu(v[m, 2j]) * v[m, 2j+1], C has N/2 columns
// (the SwiGLU of `down(silu(gate(x)) * up(x))`, modeling_llama.py:269, fused into the gate|up projection)
enum CrabAct { ACT_NONE = 0, ACT_GELU = 1, ACT_QUICK_GELU = 2, ACT_RELU = 3, ACT_SILU = 4, ACT_SWIGLU_PAIR = 5 };

__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case ACT_GELU: return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f));   // exact erf GELU
        case ACT_QUICK_GELU: return x / (1.0f + __expf(-1.702f * x));                // x*sigmoid(1.702x)
        case ACT_RELU: return fmaxf(x, 0.0f);
        case ACT_SILU: return x / (1.0f + __expf(-x));
        default: return x;
    }
}

// Logical tile id -> (tm, tn) in BANDS of GM tile rows: ids walk the GM rows of a band first, then the band's columns, then the
// next band.  With xcd_remap an XCD owns a contiguous id range and its ~32 resident blocks cover a GM x (32/GM) patch of
// output tiles, so they share GM activation panels and 32/GM weight panels through the XCD's L2.  The plain column-major
// order (32 tiles of ONE column) made every XCD stream every activation panel per column: FETCH_SIZE of the gate|up prefill
// GEMM was 4.5 GB per launch against 0.27 GB of operands (profiles/README.md).  Bijective for ragged grids.
__device__ __forceinline__ void tile_coords(int id, int tiles_m, int tiles_n, int GM, int& tm, int& tn) {
    const int per_band = GM * tiles_n;
    const int band = id / per_band, r = id - band * per_band;
    const int first = band * GM;
    const int rows = min(tiles_m - first, GM);
    tm = first + r % rows;
    tn = r / rows;
}

// RoPE rotation of one (x1, x2) = (x[i], x[i + d/2]) pair, fp32, with the operation order PINNED (one product rounded, then one
// fma): every kernel that rotates (prefill tile kernel, decode kernel, the fused q|k|v split-K reduction) must produce
// bit-identical q / k for identical inputs, and the compiler's free choice of which product to contract differs per kernel.
__device__ __forceinline__ float rope_lo(float x1, float x2, float c, float sn) { return __fmaf_rn(x1, c, -__fmul_rn(x2, sn)); }
__device__ __forceinline__ float rope_hi(float x1, float x2, float c, float sn) { return __fmaf_rn(x2, c, __fmul_rn(x1, sn)); }

// bijective XCD-aware block remap: consecutive logical ids land on the same XCD (private L2),
// valid for any grid size (cdna guide 5.5 T1 bijective form).  Placement is a speed hint only.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int NX = 8;
    int q = nwg / NX, r = nwg % NX;
    int xcd = bid % NX, idx = bid / NX;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// bf16 MFMA GEMM for gfx950:  C = res_scale*R + act(A.B^T + A2.B2^T + bias)
//
// Both operands are K-contiguous ("x @ W^T", W as stored by nn.Linear), so A and B fragments are both
// 16-byte row reads.  The MFMA is issued with the WEIGHT tile as the A operand and the ACTIVATION tile
// as the B operand (computing C^T tiles): with v_mfma_f32_16x16x32_bf16 each lane then owns 4
// consecutive output columns n of one row m, which gives packed 8-byte bf16 / 16-byte fp32 stores.
//
// Tile BMxBNx64, 256 threads = 4 waves (2x2), each wave (BM/2)x(BN/2) as 16x16 MFMA tiles.
// LDS: two stages of [BM+BN][64] bf16, XOR-swizzled at 16-byte granularity (chunk ^ ((row>>1)&7)) so
// the ds_read_b128 fragment reads of 16 consecutive rows hit 16 distinct 16-byte bank slots.
// Global->register->LDS prefetch of tile t+1 overlaps the MFMAs of tile t; one barrier per K tile.
#include "common.h"
#include "crab_internal.h"
#include <stdlib.h>
#include "gemm_epilogue.h"

namespace {

struct GemmP {
    const bf16_t* A; const bf16_t* B; void* C; const bf16_t* bias; const bf16_t* R;
    const bf16_t* A2; const bf16_t* B2;
    long lda, ldb, ldc, ldr, lda2, ldb2;
    int M, N, K, K2, act, c_fp32;
    float res_scale;
    int nb0;
    long sA0, sA1, sB0, sB1, sC0, sC1, sR0, sR1, sBias0, sBias1;
    int tiles_m, tiles_n;
    int splitk;          // > 1: blockIdx.y = K slice, raw fp32 partial tiles go to `part` [slice][M][N]
    float* part;
};

constexpr int BK = 64;

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_bt_kernel(GemmP p) {
    constexpr int WM = BM / 2, WN = BN / 2;
    constexpr int TM = WM / 16, TN = WN / 16;
    constexpr int CA = BM / 32, CB = BN / 32;          // 16-byte chunks per thread per tile
    __shared__ __attribute__((aligned(16))) bf16_t lds[2][(BM + BN) * BK];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    const int z = p.splitk > 1 ? 0 : blockIdx.y;
    const int z0 = z % p.nb0, z1 = z / p.nb0;
    const bf16_t* A = p.A + z0 * p.sA0 + z1 * p.sA1;
    const bf16_t* B = p.B + z0 * p.sB0 + z1 * p.sB1;
    const bf16_t* A2 = p.A2 ? p.A2 + z0 * p.sA0 + z1 * p.sA1 : nullptr;
    const bf16_t* B2 = p.B2 ? p.B2 + z0 * p.sB0 + z1 * p.sB1 : nullptr;

    const int nwg = p.tiles_m * p.tiles_n;
    const int bid = xcd_remap(blockIdx.x, nwg);
    const int tm = bid % p.tiles_m, tn = bid / p.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    const int nk1 = (p.K + BK - 1) / BK;
    const int nk2 = A2 ? (p.K2 + BK - 1) / BK : 0;
    const int nk = nk1 + nk2;
    int t_begin = 0, t_end = nk;
    if (p.splitk > 1) {
        t_begin = (int)((long)nk * blockIdx.y / p.splitk);
        t_end = (int)((long)nk * (blockIdx.y + 1) / p.splitk);
    }

    u32x4 ra[CA], rb[CB];
    const u32x4 zero4 = {0u, 0u, 0u, 0u};

    // per-thread staging coordinates (constant over the K loop)
    const int lrow = tid >> 3, lc = tid & 7;            // chunk i covers row lrow + 32*i, 16-byte column lc

#define GLOAD(T_)                                                                                        \
    {                                                                                                    \
        const int t_ = (T_);                                                                             \
        const bool s2_ = t_ >= nk1;                                                                      \
        const bf16_t* Ap_ = s2_ ? A2 : A;                                                                \
        const bf16_t* Bp_ = s2_ ? B2 : B;                                                                \
        const long la_ = s2_ ? p.lda2 : p.lda, lb_ = s2_ ? p.ldb2 : p.ldb;                              \
        const int Ks_ = s2_ ? p.K2 : p.K;                                                                \
        const int k_ = (s2_ ? (t_ - nk1) : t_) * BK + lc * 8;                                            \
        _Pragma("unroll") for (int i = 0; i < CA; ++i) {                                                 \
            const int gr = m0 + lrow + 32 * i;                                                           \
            ra[i] = (gr < p.M && k_ < Ks_) ? *reinterpret_cast<const u32x4*>(Ap_ + (long)gr * la_ + k_) : zero4; \
        }                                                                                                \
        _Pragma("unroll") for (int i = 0; i < CB; ++i) {                                                 \
            const int gr = n0 + lrow + 32 * i;                                                           \
            rb[i] = (gr < p.N && k_ < Ks_) ? *reinterpret_cast<const u32x4*>(Bp_ + (long)gr * lb_ + k_) : zero4; \
        }                                                                                                \
    }
#define LSTORE(BUF_)                                                                                     \
    {                                                                                                    \
        bf16_t* sa_ = &lds[(BUF_)][0];                                                                   \
        bf16_t* sb_ = &lds[(BUF_)][BM * BK];                                                             \
        _Pragma("unroll") for (int i = 0; i < CA; ++i) {                                                 \
            const int row = lrow + 32 * i;                                                               \
            *reinterpret_cast<u32x4*>(sa_ + row * BK + ((lc ^ ((row >> 1) & 7)) << 3)) = ra[i];          \
        }                                                                                                \
        _Pragma("unroll") for (int i = 0; i < CB; ++i) {                                                 \
            const int row = lrow + 32 * i;                                                               \
            *reinterpret_cast<u32x4*>(sb_ + row * BK + ((lc ^ ((row >> 1) & 7)) << 3)) = rb[i];          \
        }                                                                                                \
    }

    f32x4_t acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (t_begin < t_end) {
        GLOAD(t_begin);
        LSTORE(0);
    }
    __syncthreads();

    const int fr = lane & 15, fg = lane >> 4;
    for (int t = t_begin; t < t_end; ++t) {
        const int cur = (t - t_begin) & 1;
        if (t + 1 < t_end) GLOAD(t + 1);
        const bf16_t* la_ = &lds[cur][0];
        const bf16_t* lb_ = &lds[cur][BM * BK];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8_t wf[TN], xf[TM];
            const int chunk = ks * 4 + fg;
#pragma unroll
            for (int ni = 0; ni < TN; ++ni) {
                int row = wn * WN + ni * 16 + fr;
                wf[ni] = *reinterpret_cast<const bf16x8_t*>(lb_ + row * BK + ((chunk ^ ((row >> 1) & 7)) << 3));
            }
#pragma unroll
            for (int mi = 0; mi < TM; ++mi) {
                int row = wm * WM + mi * 16 + fr;
                xf[mi] = *reinterpret_cast<const bf16x8_t*>(la_ + row * BK + ((chunk ^ ((row >> 1) & 7)) << 3));
            }
#pragma unroll
            for (int ni = 0; ni < TN; ++ni)
#pragma unroll
                for (int mi = 0; mi < TM; ++mi)
                    acc[ni][mi] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[ni], xf[mi], acc[ni][mi], 0, 0, 0);
        }
        if (t + 1 < t_end) LSTORE(cur ^ 1);
        __syncthreads();
    }

    if (p.splitk > 1) {          // raw partial tile, reduced by splitk_epilogue_kernel in a fixed slice order
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
    // output stage shared with the LDS-DMA kernels (gemm_epilogue.h): unguarded + activation-specialised on interior sub-tiles
    gemm_epilogue<TM, TN>(acc, p.act, m0 + wm * WM, n0 + wn * WN, fr, fg, p.M, p.N, p.bias ? p.bias + z0 * p.sBias0 + z1 * p.sBias1 : nullptr,
                          p.R ? p.R + z0 * p.sR0 + z1 * p.sR1 : nullptr, p.ldr, p.res_scale, p.C, z0 * p.sC0 + z1 * p.sC1, p.ldc, p.c_fp32);
}

#undef GLOAD
#undef LSTORE

// sum of the K-slice partials (fixed order) + bias / activation / residual epilogue; one thread per 4 columns
__global__ void splitk_epilogue_kernel(const float* __restrict__ part, int S, int M, int N, const bf16_t* __restrict__ bias, int act,
                                       const bf16_t* __restrict__ R, long ldr, float res_scale, void* __restrict__ C, long ldc, int c_fp32) {
    const int nq = (N + 3) >> 2;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)M * nq) return;
    const int m = idx / nq, n = (idx % nq) * 4;
    const long MN = (long)M * N;
    const float* q = part + (long)m * N + n;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const int w = min(4, N - n);
    if (w == 4 && (N & 3) == 0) {
        for (int s = 0; s < S; ++s) {
            f32x4_t t = *reinterpret_cast<const f32x4_t*>(q + s * MN);
            v[0] += t[0]; v[1] += t[1]; v[2] += t[2]; v[3] += t[3];
        }
    } else {
        for (int s = 0; s < S; ++s)
            for (int r = 0; r < w; ++r) v[r] += q[s * MN + r];
    }
    if (act == ACT_SWIGLU_PAIR) {                         // interleaved (gate, up) columns -> N/2 outputs (N % 4 == 0, no residual)
        if (bias) { for (int r = 0; r < 4; ++r) v[r] += bf2f(bias[n + r]); }
        const float o0 = v[0] / (1.0f + __expf(-v[0])) * v[1], o1 = v[2] / (1.0f + __expf(-v[2])) * v[3];
        const long oc = (long)m * ldc + (n >> 1);
        if (c_fp32 & CF_C32) { reinterpret_cast<float*>(C)[oc] = o0; reinterpret_cast<float*>(C)[oc + 1] = o1; }
        else { reinterpret_cast<bf16_t*>(C)[oc] = f2bf(o0); reinterpret_cast<bf16_t*>(C)[oc + 1] = f2bf(o1); }
        return;
    }
    for (int r = 0; r < w; ++r) {
        float x = v[r];
        if (bias) x += bf2f(bias[n + r]);
        x = apply_act(x, act);
        if (R) x += res_scale * ld_res(R, (long)m * ldr + n + r, c_fp32);
        if (c_fp32 & CF_C32) reinterpret_cast<float*>(C)[(long)m * ldc + n + r] = x;
        else reinterpret_cast<bf16_t*>(C)[(long)m * ldc + n + r] = f2bf(x);
    }
}

// SwiGLU-pair fast path of the reduction above (the gate|up projection of the decode step): one thread per 16 interleaved
// (gate, up) columns = 4 x 16-byte loads per slice and ONE 16-byte store of 8 outputs.  Same summation order per column.
__global__ __launch_bounds__(256) void splitk_epilogue_swiglu8_kernel(const float* __restrict__ part, int S, int M, int N,
                                                                       const bf16_t* __restrict__ bias, bf16_t* __restrict__ C, long ldc) {
    const int nq = N >> 4;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)M * nq) return;
    const int m = (int)(idx / nq), n = (int)(idx % nq) * 16;
    const long MN = (long)M * N;
    const float* q = part + (long)m * N + n;
    f32x4_t v[4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int s = 0; s < S; ++s) {
        f32x4_t t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) t[i] = *reinterpret_cast<const f32x4_t*>(q + s * MN + 4 * i);
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] += t[i];
    }
    if (bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[i][r] += bf2f(bias[n + 4 * i + r]);
    }
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float o0 = v[i][0] / (1.0f + __expf(-v[i][0])) * v[i][1], o1 = v[i][2] / (1.0f + __expf(-v[i][2])) * v[i][3];
        o[i] = pack_bf2(o0, o1);
    }
    *reinterpret_cast<u32x4*>(C + (long)m * ldc + (n >> 1)) = o;
}

struct RouteP {                                   // router of the next projection group (crab_gemm_desc.route_*)
    const bf16_t* RA; bf16_t* U; long ldra, ldu; int nproj, nl, r, ucols; float scaling;
};

// Row-owning variant of the split-K epilogue: one block per output row sums the K-slice partials (fixed order), applies
// bias / activation / residual, writes C (bf16) AND the RMS-normalised row rmsnorm(C)*w for the next projection
// (LlamaRMSNorm, modeling_llama.py:112-117, fused behind o_proj / down_proj in the decode regime).  N <= 8192.
// XF: R and C are the FP32 residual stream (ldr / ldc in fp32 elements): the row is stored unrounded, the norm sees that fp32 row and the
// normalised row is bf16(x * rstd * w) without the intermediate rounding of x_hat.
template <bool XF, bool WF = false>                           // WF: the norm weight is fp32 (crab_gemm_desc.norm_w_fp32)
__global__ __launch_bounds__(256) void splitk_epilogue_norm_kernel(const float* __restrict__ part, int S, int M, int N,
                                                                    const bf16_t* __restrict__ bias, int act, const bf16_t* __restrict__ R,
                                                                    long ldr, float res_scale, bf16_t* __restrict__ C, long ldc,
                                                                    const void* __restrict__ nw, float eps, bf16_t* __restrict__ H, long ldh,
                                                                    RouteP rt) {
    __shared__ float red[4];
    const int m = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // gridDim.y = P > 1 (few rows, launch_norm_epilogue): P blocks share a row; each recomputes the row and its norm (identical
    // arithmetic), block y == 0 stores C and H, and block y evaluates the router of projection y only.  The inputs must then be
    // read-only for the launch: no residual operand (the producer folded it into `part`), so that C may alias nothing it reads.
    const int py = (int)blockIdx.y, P = (int)gridDim.y;
    const long MN = (long)M * N;
    constexpr int MAXQ = 4;                                   // 4 chunks of 8 columns per thread: N <= 8192; every access is 16 B wide
    float xv[MAXQ][8];
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int n = (tid + q * 256) * 8;
        if (n < N) {
            f32x4_t v0 = {0.f, 0.f, 0.f, 0.f}, v1 = {0.f, 0.f, 0.f, 0.f};
            const float* pp = part + (long)m * N + n;
            int s = 0;
            for (; s + 4 <= S; s += 4) {                          // 4 slices (8 loads) in flight, summed in slice order
                f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(pp + (s + 0) * MN), b0 = *reinterpret_cast<const f32x4_t*>(pp + (s + 0) * MN + 4);
                f32x4_t a1 = *reinterpret_cast<const f32x4_t*>(pp + (s + 1) * MN), b1 = *reinterpret_cast<const f32x4_t*>(pp + (s + 1) * MN + 4);
                f32x4_t a2 = *reinterpret_cast<const f32x4_t*>(pp + (s + 2) * MN), b2 = *reinterpret_cast<const f32x4_t*>(pp + (s + 2) * MN + 4);
                f32x4_t a3 = *reinterpret_cast<const f32x4_t*>(pp + (s + 3) * MN), b3 = *reinterpret_cast<const f32x4_t*>(pp + (s + 3) * MN + 4);
                v0 += a0; v0 += a1; v0 += a2; v0 += a3;
                v1 += b0; v1 += b1; v1 += b2; v1 += b3;
            }
            for (; s < S; ++s) {
                v0 += *reinterpret_cast<const f32x4_t*>(pp + s * MN);
                v1 += *reinterpret_cast<const f32x4_t*>(pp + s * MN + 4);
            }
            u32x4 rr = {0u, 0u, 0u, 0u}, bb = {0u, 0u, 0u, 0u};
            f32x4_t rf0 = {0.f, 0.f, 0.f, 0.f}, rf1 = {0.f, 0.f, 0.f, 0.f};
            if (XF) {
                if (R) {
                    rf0 = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(R) + (long)m * ldr + n);
                    rf1 = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(R) + (long)m * ldr + n + 4);
                }
            } else if (R) rr = *reinterpret_cast<const u32x4*>(R + (long)m * ldr + n);
            if (bias) bb = *reinterpret_cast<const u32x4*>(bias + n);
            u32x4 o;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                float x = r < 4 ? v0[r] : v1[r - 4];
                const uint32_t bw = bb[r >> 1], rw = rr[r >> 1];
                if (bias) x += (r & 1) ? hi_bf(bw) : lo_bf(bw);
                x = apply_act(x, act);
                if (XF) {
                    if (R) x += res_scale * (r < 4 ? rf0[r & 3] : rf1[r & 3]);
                    xv[q][r] = x;                              // fp32 residual stream: stored and normalised unrounded
                } else {
                    if (R) x += res_scale * ((r & 1) ? hi_bf(rw) : lo_bf(rw));
                    xv[q][r] = bf2f(f2bf(x));                  // the norm sees the stored bf16 value
                }
                ss += xv[q][r] * xv[q][r];
            }
            if (XF) {
                if (py == 0) {
                    float* cx = reinterpret_cast<float*>(C) + (long)m * ldc + n;
                    *reinterpret_cast<f32x4_t*>(cx) = f32x4_t{xv[q][0], xv[q][1], xv[q][2], xv[q][3]};
                    *reinterpret_cast<f32x4_t*>(cx + 4) = f32x4_t{xv[q][4], xv[q][5], xv[q][6], xv[q][7]};
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = pack_bf2(xv[q][2 * r], xv[q][2 * r + 1]);
                if (py == 0) *reinterpret_cast<u32x4*>(C + (long)m * ldc + n) = o;
            }
        }
    }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    const float rstd = rsqrtf((red[0] + red[1] + red[2] + red[3]) / (float)N + eps);
#pragma unroll
    for (int q = 0; q < MAXQ; ++q) {
        const int n = (tid + q * 256) * 8;
        if (n < N) {
            float wv[8];
            ld_par8<WF>(nw, (long)n, wv);
            float h8[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) h8[r] = (XF ? xv[q][r] * rstd : bf2f(f2bf(xv[q][r] * rstd))) * wv[r];
            u32x4 o;
#pragma unroll
            for (int r = 0; r < 4; ++r) o[r] = pack_bf2(h8[2 * r], h8[2 * r + 1]);
            if (py == 0) *reinterpret_cast<u32x4*>(H + (long)m * ldh + n) = o;
#pragma unroll
            for (int r = 0; r < 8; ++r) xv[q][r] = bf2f(f2bf(h8[r]));          // the router sees the stored bf16 row
        }
    }
    if (!rt.RA) return;
    // ---- hyper-LoRA router of the next projection group on this row (peft_hyper/tuners/lora.py:346-350): t = h . [R;A]^T,
    // u = scaling * softmax(t_route) (x) t_A.  Partials per thread -> LDS -> fixed-order sums (deterministic).
    __shared__ float tp[64][257];
    __shared__ float tq[64][4];
    __shared__ float T[64];
    const int tcols = ((rt.nproj * (rt.nl + rt.r) + 15) / 16) * 16;
    // RPT router rows per trip, all their loads in flight together.  Every adapter of the model has nl + r = 3 + 8 = 11 rows per
    // projection: one trip per projection reads exactly the used rows (33 of the 48 padded ones for q|k|v, 22 of 32 for gate|up);
    // other shapes walk the padded table 16 rows at a time.
#define ROUTE_TRIPS(RPT_, NTRIPS_)                                                                        \
    for (int tr = 0; tr < (NTRIPS_); ++tr) {                                                              \
        const int c0 = tr * (RPT_);                                                                       \
        float p[RPT_];                                                                                    \
        _Pragma("unroll") for (int cc = 0; cc < (RPT_); ++cc) p[cc] = 0.f;                                \
        _Pragma("unroll") for (int q = 0; q < MAXQ; ++q) {                                                \
            const int n = (tid + q * 256) * 8;                                                            \
            if (n < N) {                                                                                  \
                u32x4 w[RPT_];                                                                            \
                _Pragma("unroll") for (int cc = 0; cc < (RPT_); ++cc)                                     \
                    w[cc] = *reinterpret_cast<const u32x4*>(rt.RA + (long)(c0 + cc) * rt.ldra + n);       \
                _Pragma("unroll") for (int cc = 0; cc < (RPT_); ++cc) {                                   \
                    float a = 0.f;                                                                        \
                    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                         \
                        a += xv[q][2 * e] * lo_bf(w[cc][e]) + xv[q][2 * e + 1] * hi_bf(w[cc][e]);         \
                    p[cc] += a;                                                                           \
                }                                                                                         \
            }                                                                                             \
        }                                                                                                 \
        _Pragma("unroll") for (int cc = 0; cc < (RPT_); ++cc) tp[c0 + cc][tid] = p[cc];                   \
    }
    const int used_rows = rt.nproj * (rt.nl + rt.r);
    if (P > 1) {                                             // projection py only (host guarantees nl + r == 11, P == nproj)
        for (int tr = py; tr == py; ++tr) {
            const int c0 = tr * 11;
            float p[11];
#pragma unroll
            for (int cc = 0; cc < 11; ++cc) p[cc] = 0.f;
#pragma unroll
            for (int q = 0; q < MAXQ; ++q) {
                const int n = (tid + q * 256) * 8;
                if (n < N) {
                    u32x4 w[11];
#pragma unroll
                    for (int cc = 0; cc < 11; ++cc) w[cc] = *reinterpret_cast<const u32x4*>(rt.RA + (long)(c0 + cc) * rt.ldra + n);
#pragma unroll
                    for (int cc = 0; cc < 11; ++cc) {
                        float a = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e) a += xv[q][2 * e] * lo_bf(w[cc][e]) + xv[q][2 * e + 1] * hi_bf(w[cc][e]);
                        p[cc] += a;
                    }
                }
            }
#pragma unroll
            for (int cc = 0; cc < 11; ++cc) tp[c0 + cc][tid] = p[cc];
        }
    } else if (rt.nl + rt.r == 11) {
        ROUTE_TRIPS(11, rt.nproj)
        for (int c = used_rows; c < tcols; ++c) tp[c][tid] = 0.f;
    } else {
        ROUTE_TRIPS(16, tcols / 16)
    }
#undef ROUTE_TRIPS
    __syncthreads();
    if (tid < tcols * 4 && (P == 1 || ((tid >> 2) >= py * 11 && (tid >> 2) < py * 11 + 11))) {
        const int c = tid >> 2, qt = tid & 3;
        float a = 0.f;
        for (int i = qt * 64; i < qt * 64 + 64; ++i) a += tp[c][i];
        tq[c][qt] = a;
    }
    __syncthreads();
    if (tid < tcols) T[tid] = (tq[tid][0] + tq[tid][1]) + (tq[tid][2] + tq[tid][3]);
    __syncthreads();
    if (tid > rt.nproj || (P > 1 && tid != py && !(tid == rt.nproj && py == 0))) return;
    bf16_t* u = rt.U + (long)m * rt.ldu;
    const int used = rt.nproj * rt.nl * rt.r;
    if (tid == rt.nproj) {
        for (int c = used; c < rt.ucols; ++c) u[c] = 0;
        return;
    }
    const float* t = &T[tid * (rt.nl + rt.r)];
    float e[8], mx = -INFINITY;
    for (int i = 0; i < rt.nl; ++i) mx = fmaxf(mx, t[i]);
    float sum = 0.f;
    for (int i = 0; i < rt.nl; ++i) { e[i] = expf(t[i] - mx); sum += e[i]; }
    const float inv = 1.0f / sum;
    for (int i = 0; i < rt.nl; ++i)
        for (int j = 0; j < rt.r; ++j) u[tid * rt.nl * rt.r + i * rt.r + j] = f2bf(rt.scaling * e[i] * inv * t[rt.nl + j]);
}
}  // namespace

int crab_gemm_skinny_launch(crab_ctx* ctx, hipStream_t s, const crab_gemm_desc* d);   // skinny.hip
bool crab_skinny_fuses_rope(const crab_gemm_desc* d);                                   // skinny.hip: RoPE + KV append in the M <= 16 epilogue
bool crab_rowfin_ok(const crab_gemm_desc* d);                                           // rowfin.hip: the M <= 16 layer tail
int crab_rowfin_launch(crab_ctx* ctx, hipStream_t s, const crab_gemm_desc* d);
int crab_gemm_glds_launch(crab_ctx* ctx, hipStream_t s, const crab_gemm_desc* d, int splitk, float* part, int ring_split);     // gemm_glds.hip
int crab_gemm_dec_launch(crab_ctx* ctx, hipStream_t s, const crab_gemm_desc* d, int bn, int splitk, float* part, int nt_weights);  // gemm_decode.hip

// Decode-regime decomposition (128 < M <= 256): column-panel width BN in {96, 64} and K slices for gemm_dec_kernel (one block per
// CU).  Measured on MI355X (profiles/README.md r02, scripts/bench_dec_gemm.py): a block moves its LDS-DMA bytes - the activation
// rows of its K slice, re-read from L2 by every block, plus its weight panel - at ~58 GB/s per CU whatever the panel width or the
// number of active CUs, so the K loop costs bytes_per_block / 58 GB/s; ~7 us (BN 64) / ~9 us (BN 96: 3-slot ring) of prologue +
// epilogue per round; the reduction kernel
// ~4 us + the fp32 slabs written and read back at ~4.5 TB/s while they stay MALL / L2 resident (<= 40 MB) and ~2 TB/s beyond.
static void dec_choose(const crab_gemm_desc* d, int nks /* 64-wide K slots */, int* bn_out, int* split_out) {
    double best = 1e30;
    for (int bi = 0; bi < 3; ++bi) {
        // 160-wide panels (r03), only for projections too wide for ONE round of 96-wide panels: Qwen2-7B gate|up (37888 columns = 395 panels =
        // 2 rounds, the second 54 % empty) runs as one round of 237 (97.8 -> 71.9 us), the lm_heads in 1 instead of 2 (Llama, 114 -> 88 us) and 4
        // instead of 7 rounds (Qwen2, 403 -> 357 us); narrower projections lose (activation bytes per block stay, blocks get fewer)
        const int bn = bi == 0 ? 96 : (bi == 1 ? 64 : 160);
        if (bn == 160 && (d->N + 95) / 96 <= 256) continue;
        const long tiles = (d->N + bn - 1) / bn;
        for (int sp = 1; sp <= 8 && sp * 4 <= nks; ++sp) {
            const long blocks = tiles * sp;
            const double rounds = (double)((blocks + 255) / 256);
            const int per = (nks + sp - 1) / sp;
            if ((long)(sp - 1) * per >= nks) continue;                   // an empty slice
            const double slab = (double)sp * d->M * d->N * 4.0;
            if (sp > 1 && slab > (double)d->workspace_bytes) continue;
            const double bytes = (256.0 + bn) * per * 64 * 2;
            const double t = rounds * ((bn == 64 ? 7.0 : 10.0) + bytes / 58.0e3) + (sp > 1 ? 4.0 + 2.0 * slab / (slab <= 40.0e6 ? 4.5e6 : 2.0e6) : 0.0);
            if (t < best) { best = t; *bn_out = bn; *split_out = sp; }
        }
    }
}


// The same choice for the two-row-group kernel (gemm_dec2_kernel, 256 < M <= 512), calibrated on scripts/bench_dec2_cfg.py (profiles/README.md
// r04): a K slot costs 0.7 + 0.8 f us with 96-wide panels and 0.45 + 0.55 f us with 64-wide ones, f = the fraction of the 256 CUs streaming in
// the round (the weight stream slows every CU down as more of them pull on HBM), ~12 us of prologue + epilogue per round; the reduction
// ~4 us + one pass over the slabs at ~4 TB/s; a q|k|v projection without slices pays a separate RoPE / KV-append pass (~12 us), an o / down projection without
// slices a separate norm + router pass (~25 us).  Picks q|k|v 96 x 2, o / down 64 x 4, gate|up 96 x 1, lm_head 64 x 1 at Llama-2-7B widths.
static void dec2_choose(const crab_gemm_desc* d, int nks, int* bn_out, int* split_out) {
    double best = 1e30;
    for (int bi = 0; bi < 2; ++bi) {
        const int bn = bi == 0 ? 96 : 64;
        const long tiles = (d->N + bn - 1) / bn;
        for (int sp = 1; sp <= 8 && sp * 4 <= nks; ++sp) {
            const long blocks = tiles * sp;
            const int per = (nks + sp - 1) / sp;
            if ((long)(sp - 1) * per >= nks) continue;                   // an empty slice
            const double slab = (double)sp * d->M * d->N * 4.0;
            if (sp > 1 && slab > (double)d->workspace_bytes) continue;
            double t = 0.0;
            for (long left = blocks; left > 0; left -= 256) {
                const double f = (double)(left < 256 ? left : 256) / 256.0;
                t += 12.0 + per * (bn == 96 ? 0.7 + 0.8 * f : 0.45 + 0.55 * f);
            }
            if (sp > 1) t += 4.0 + slab / 4.0e6;
            else t += d->rope_tab ? 12.0 : (d->norm_w ? 25.0 : 0.0);
            if (t < best) { best = t; *bn_out = bn; *split_out = sp; }
        }
    }
}


// Split-K reduction of the packed q|k|v projection fused with RoPE and the KV-cache append (decode: one row per sequence).
// One thread per (row, head, 4 dims of the first half): it also owns the matching 4 dims of the second half, i.e. the
// rotation partners.  Sums are rounded to bf16 before the rotation, exactly like the unfused pair (reduction kernel ->
// bf16 C -> qkv_rope_split_kernel), so prefill and decode see identical k / q values for identical inputs.
template <int G>                                                // dims per thread and half: 4 (8-byte stores) or 8 (16-byte stores)
__global__ __launch_bounds__(256) void splitk_epilogue_rope_kernel(const float* __restrict__ part, int S, int M, int N,
                                                                    const bf16_t* __restrict__ bias, bf16_t* __restrict__ C, long ldc,
                                                                    const float* __restrict__ tab, bf16_t* __restrict__ kc,
                                                                    bf16_t* __restrict__ vc, const int* __restrict__ pos_dev, int pos0,
                                                                    int H, int Hk, int d, int Tmax, const int* __restrict__ row_off) {
    constexpr int V = G / 4;                                    // float4 per half
    const int half = d >> 1, gpd = half / G;                    // G-dim groups per half head
    const int nh = H + 2 * Hk;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)M * nh * gpd) return;
    const int j = (int)(idx % gpd);
    const int hh = (int)((idx / gpd) % nh);
    const int m = (int)(idx / ((long)gpd * nh));
    const int n1 = hh * d + G * j, n2 = n1 + half;
    const long MN = (long)M * N;
    const float* q = part + (long)m * N;
    f32x4_t a[V], b[V];
#pragma unroll
    for (int i = 0; i < V; ++i) { a[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; b[i] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    int s = 0;
    for (; s + 2 <= S; s += 2) {                                // two slices (4 V loads) in flight, summed in slice order
        f32x4_t ta[2][V], tb[2][V];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < V; ++i) {
                ta[u][i] = *reinterpret_cast<const f32x4_t*>(q + (s + u) * MN + n1 + 4 * i);
                tb[u][i] = *reinterpret_cast<const f32x4_t*>(q + (s + u) * MN + n2 + 4 * i);
            }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int i = 0; i < V; ++i) { a[i] += ta[u][i]; b[i] += tb[u][i]; }
    }
    for (; s < S; ++s)
#pragma unroll
        for (int i = 0; i < V; ++i) {
            a[i] += *reinterpret_cast<const f32x4_t*>(q + s * MN + n1 + 4 * i);
            b[i] += *reinterpret_cast<const f32x4_t*>(q + s * MN + n2 + 4 * i);
        }
    float x1[G], x2[G];
#pragma unroll
    for (int r = 0; r < G; ++r) {
        x1[r] = bf2f(f2bf(a[r >> 2][r & 3] + (bias ? bf2f(bias[n1 + r]) : 0.f)));
        x2[r] = bf2f(f2bf(b[r >> 2][r & 3] + (bias ? bf2f(bias[n2 + r]) : 0.f)));
    }
    const int pos = (pos_dev ? pos_dev[0] : 0) + pos0;
    bf16_t* dst1; bf16_t* dst2;
    if (hh < H) { dst1 = C + (long)m * ldc + n1; dst2 = C + (long)m * ldc + n2; }
    else {
        const int hk = (hh - H) % Hk;
        bf16_t* cache = hh < H + Hk ? kc : vc;
        dst1 = cache + (((long)m * Hk + hk) * Tmax + pos) * d + G * j;
        dst2 = dst1 + half;
    }
    float o1[G], o2[G];
    if (hh < H + Hk) {
        // ragged decode batch (crab_gemm_desc.rope_row_off): the row is rotated at slot - row_off[m], its K / V rows stay in slot `pos`
        const int rp = pos - (row_off ? row_off[m] : 0);
        const float* t = tab + 2 * ((long)rp * half + G * j);    // (cos, sin) pairs of dims G j .. G j + G - 1
#pragma unroll
        for (int r = 0; r < G; ++r) {
            const float c = t[2 * r], sn = t[2 * r + 1];
            o1[r] = rope_lo(x1[r], x2[r], c, sn);
            o2[r] = rope_hi(x1[r], x2[r], c, sn);
        }
    } else {
#pragma unroll
        for (int r = 0; r < G; ++r) { o1[r] = x1[r]; o2[r] = x2[r]; }
    }
    uint32_t w1[G / 2], w2[G / 2];
#pragma unroll
    for (int r = 0; r < G / 2; ++r) { w1[r] = pack_bf2(o1[2 * r], o1[2 * r + 1]); w2[r] = pack_bf2(o2[2 * r], o2[2 * r + 1]); }
    if (G == 8) {
        *reinterpret_cast<u32x4*>(dst1) = u32x4{w1[0], w1[1], w1[G / 2 - 2], w1[G / 2 - 1]};
        *reinterpret_cast<u32x4*>(dst2) = u32x4{w2[0], w2[1], w2[G / 2 - 2], w2[G / 2 - 1]};
    } else {
        *reinterpret_cast<u32x2*>(dst1) = u32x2{w1[0], w1[1]};
        *reinterpret_cast<u32x2*>(dst2) = u32x2{w2[0], w2[1]};
    }
}

// The row-owning reduction (splitk_epilogue_norm_kernel): conditions and launch, shared by the split-K paths and the M <= 16 path
static bool norm_epilogue_ok(const crab_gemm_desc* d) {
    return d->norm_w && (d->N & 7) == 0 && d->N <= 8192 && (d->ldc & 7) == 0 && (d->ld_norm & 7) == 0 &&
           (((uintptr_t)d->C | (uintptr_t)d->norm_out | (uintptr_t)d->norm_w | (uintptr_t)d->R | (uintptr_t)d->bias | (uintptr_t)d->route_RA) & 15) == 0 &&
           (!d->R || (d->ldr & 7) == 0) && (!d->route_RA || ((d->route_ldra & 7) == 0 && d->route_nl <= 8 &&
                                                             d->route_nproj * (d->route_nl + d->route_r) <= 64));
}

static int launch_norm_epilogue(crab_ctx* ctx, hipStream_t s, const crab_gemm_desc* d, const float* part, int splitk, bool folded = false) {
    // folded: `part` already holds act(sum + bias) + res_scale * R in fp32 (the M <= 16 producer): bias / residual are not read
    // again, which makes every input read-only and lets nproj blocks share a row (one projection of the next router each)
    if (folded) {
        RouteP rf;
        rf.RA = (const bf16_t*)d->route_RA; rf.U = (bf16_t*)d->route_U; rf.ldra = d->route_ldra; rf.ldu = d->route_ldu;
        rf.nproj = d->route_nproj; rf.nl = d->route_nl; rf.r = d->route_r; rf.ucols = d->route_ucols; rf.scaling = d->route_scaling;
        const int P = (rf.RA && rf.nproj > 1 && rf.nl + rf.r == 11) ? rf.nproj : 1;
#define NE_LAUNCH(XF_, WF_) hipLaunchKernelGGL((splitk_epilogue_norm_kernel<XF_, WF_>), dim3(d->M, P), dim3(256), 0, s, part, splitk, d->M, d->N, (const bf16_t*)nullptr, ACT_NONE, \
                           (const bf16_t*)nullptr, 0L, 1.0f, (bf16_t*)d->C, (long)d->ldc, d->norm_w, d->norm_eps,                      \
                           (bf16_t*)d->norm_out, (long)d->ld_norm, rf)
        if (d->norm_w_fp32) NE_LAUNCH(true, true); else if (d->c_fp32) NE_LAUNCH(true, false); else NE_LAUNCH(false, false);
#undef NE_LAUNCH
        return crab_check_launch(ctx, "splitk_epilogue_norm_kernel");
    }
    RouteP rt;
    rt.RA = (const bf16_t*)d->route_RA; rt.U = (bf16_t*)d->route_U; rt.ldra = d->route_ldra; rt.ldu = d->route_ldu;
    rt.nproj = d->route_nproj; rt.nl = d->route_nl; rt.r = d->route_r; rt.ucols = d->route_ucols; rt.scaling = d->route_scaling;
#define NE_LAUNCH(XF_, WF_) hipLaunchKernelGGL((splitk_epilogue_norm_kernel<XF_, WF_>), dim3(d->M), dim3(256), 0, s, part, splitk, d->M, d->N, (const bf16_t*)d->bias, d->act, \
                       (const bf16_t*)d->R, (long)d->ldr, d->res_scale, (bf16_t*)d->C, (long)d->ldc, d->norm_w, d->norm_eps,             \
                       (bf16_t*)d->norm_out, (long)d->ld_norm, rt)
    if (d->norm_w_fp32) NE_LAUNCH(true, true); else if (d->c_fp32) NE_LAUNCH(true, false); else NE_LAUNCH(false, false);
#undef NE_LAUNCH
    return crab_check_launch(ctx, "splitk_epilogue_norm_kernel");
}

// unfused form of the optional post-RMSNorm (paths whose epilogue does not own whole rows)
static int post_norm(crab_ctx* ctx, void* stream, const crab_gemm_desc* d) {
    if (d->rope_tab && d->rope_S > 1) return CRAB_OK;   // prefill form: fused in the large-M epilogue or left to the caller (crab_gemm_fuses_prefill_rope)
    if (d->rope_tab)          // paths whose epilogue did not fuse the RoPE / KV append: the separate pass over C
        return crab_qkv_rope_split_ragged(ctx, stream, d->C, d->ldc, d->rope_tab, d->rope_k_cache, d->rope_v_cache, nullptr, 0, d->M, 1, d->rope_H,
                                          d->rope_Hk, d->rope_d, d->rope_Tmax, d->rope_pos0, d->rope_pos_dev, d->rope_row_off);
    if (!d->norm_w) return CRAB_OK;
    int rc = crab_rmsnorm_p(ctx, stream, d->C, d->c_fp32, d->ldc, d->norm_w, d->norm_w_fp32, d->norm_out, d->ld_norm, d->M, d->N, d->norm_eps);
    if (rc || !d->route_RA) return rc;
    return crab_hyperlora_route(ctx, stream, d->norm_out, d->ld_norm, d->route_RA, d->route_ldra, d->M, d->N, d->route_nproj, d->route_nl,
                                d->route_r, d->route_U, d->route_ldu, d->route_ucols, d->route_scaling, d->workspace, d->workspace_bytes);
}

static int crab_dec_min_rows() {
    static const int v = []() { const char* e = getenv("CRAB_DEC_MIN_ROWS"); const int x = e ? atoi(e) : 64; return x < 16 ? 16 : x; }();
    return v;
}

extern "C" int crab_gemm_bf16(crab_ctx* ctx, void* stream, const crab_gemm_desc* d) {
    if (!ctx) return CRAB_E_INVALID;
    if (d && d->norm_w && (!d->norm_out || d->batch > 1)) return crab_fail(ctx, CRAB_E_INVALID, "gemm: post-norm needs norm_out, no batch");
    // storage of the residual stream: R and C are bf16 (r01-r03) or, with c_fp32 = r_fp32 = 1, the fp32 residual stream.  The post-norm reads
    // C, so an fp32 C with a bf16 R (or the reverse) under a post-norm is refused rather than guessed at
    if (d && d->r_fp32 && (!d->R || d->batch > 1 || (d->ldr & 3) || ((uintptr_t)d->R & 15)))
        return crab_fail(ctx, CRAB_E_INVALID, "gemm: r_fp32 needs R (16-byte aligned, ldr % 4 == 0), no batch");
    if (d && d->norm_w && d->R && (!!d->c_fp32 != !!d->r_fp32))
        return crab_fail(ctx, CRAB_E_INVALID, "gemm: with a post-norm R and C must have the same storage (both bf16, or c_fp32 = r_fp32 = 1)");
    if (d && d->norm_w && d->norm_w_fp32 && (!d->c_fp32 || ((uintptr_t)d->norm_w & 15)))
        return crab_fail(ctx, CRAB_E_INVALID, "gemm: norm_w_fp32 needs the fp32 residual stream (c_fp32) and a 16-byte aligned norm_w");
    if (d && d->norm_w && d->c_fp32 && ((d->ldc & 3) || ((uintptr_t)d->C & 15)))
        return crab_fail(ctx, CRAB_E_INVALID, "gemm: an fp32 C under a post-norm needs 16-byte aligned rows");
    if (!d || !d->A || !d->B || !d->C) return crab_fail(ctx, CRAB_E_INVALID, "gemm: null operand");
    if (d->route_RA && (!d->norm_w || !d->route_U || d->route_nproj < 1 || d->route_nl < 1 || d->route_r < 1))
        return crab_fail(ctx, CRAB_E_INVALID, "gemm: the next-group router needs the fused post-norm, route_U and positive nproj / nl / r");
    if (d->rope_tab) {
        if (d->c_fp32 || d->act != ACT_NONE || d->R || d->norm_w || d->batch > 1 || !d->rope_k_cache || !d->rope_v_cache ||
            d->rope_H <= 0 || d->rope_Hk <= 0 || d->rope_d <= 0 || d->N != (d->rope_H + 2 * d->rope_Hk) * d->rope_d)
            return crab_fail(ctx, CRAB_E_INVALID, "gemm: fused rope needs bf16 C, N == (H + 2 Hk) d, caches, no act / residual / post-norm / batch");
        if (d->rope_row_off && d->rope_S > 1)
            return crab_fail(ctx, CRAB_E_INVALID, "gemm: rope_row_off belongs to the decode form (one row per sequence); a prefill call advances its cache pointers instead");
    }
    // prefill form of the fused RoPE (rope_S > 1 rows per sequence): only the large-M kernel implements it; when it will not (shape, alignment,
    // kernel choice - crab_gemm_fuses_prefill_rope, the caller asks the same question and then runs crab_qkv_rope_split itself) the rope
    // fields are dropped here so that no decode-form path (one row per sequence) ever sees them
    crab_gemm_desc plain;
    if (d->rope_tab && d->rope_S > 1 && !crab_gemm_fuses_prefill_rope(d)) { plain = *d; plain.rope_tab = nullptr; d = &plain; }
    if (d->act == ACT_SWIGLU_PAIR && ((d->N & 3) || d->R || d->norm_w || d->batch > 1 || (d->ldc & 1)))
        return crab_fail(ctx, CRAB_E_INVALID, "gemm: swiglu-pair epilogue needs N % 4 == 0, even ldc, no residual / post-norm / batch");
    if (d->M <= 0 || d->N <= 0 || d->K <= 0) return crab_fail(ctx, CRAB_E_INVALID, "gemm: non-positive dimension");
    if ((d->K & 7) || (d->lda & 7) || (d->ldb & 7)) return crab_fail(ctx, CRAB_E_INVALID, "gemm: K/lda/ldb must be multiples of 8");
    if (((uintptr_t)d->A & 15) || ((uintptr_t)d->B & 15)) return crab_fail(ctx, CRAB_E_INVALID, "gemm: A/B must be 16-byte aligned");
    if ((d->A2 != nullptr) != (d->B2 != nullptr) && !(d->lora_RA && !d->A2)) return crab_fail(ctx, CRAB_E_INVALID, "gemm: A2/B2 must be given together");
    if (d->lora_RA) {
        // in-call hyper-LoRA of a single-projection group: only the M <= 16 tail (rowfin.hip) evaluates it
        const bool rowfin_on = crab_rowfin_enabled();
        if (d->A2 || !d->B2 || d->lora_nl < 1 || d->lora_r < 1 || (d->lora_ldra & 7) || ((uintptr_t)d->lora_RA & 15))
            return crab_fail(ctx, CRAB_E_INVALID, "gemm: lora_RA needs B2 = lora_B without A2, positive lora_nl / lora_r, aligned lora_RA");
        if (d->K2 & 7) return crab_fail(ctx, CRAB_E_INVALID, "gemm: lora_RA needs K2 % 8 == 0 (lora_B is read in 16-byte chunks) and lora_RA padded to 16 rows (rows beyond lora_nl + lora_r zero)");
        if (!rowfin_on || d->tune != 0 || d->batch > 1 || !crab_rowfin_ok(d) || (d->ldb & 7))
            return crab_fail(ctx, CRAB_E_UNSUPPORTED, "gemm: the in-call hyper-LoRA (lora_RA) is evaluated by the M <= 16 layer tail only: needs M <= 16, "
                                                      "the fused post-norm (norm_w / norm_out), crab_rowfin_lora_ok(lora_nl, lora_r, N), a workspace of crab_rowfin_workspace(M, N) bytes");
    }
    if (d->A2) {
        if (d->K2 <= 0 || (d->K2 & 7) || (d->lda2 & 7) || (d->ldb2 & 7))
            return crab_fail(ctx, CRAB_E_INVALID, "gemm: K2/lda2/ldb2 must be positive multiples of 8");
        if (((uintptr_t)d->A2 & 15) || ((uintptr_t)d->B2 & 15)) return crab_fail(ctx, CRAB_E_INVALID, "gemm: A2/B2 alignment");
        if (d->batch > 1) return crab_fail(ctx, CRAB_E_UNSUPPORTED, "gemm: second K segment is not batched");
    }
    // ---- weight-streaming regime (decode): M <= 256 rows
    //   M <= 16          : LDS-free skinny kernel (skinny.hip), activations replicated per 16 weight rows
    //   16 < M <= 256    : tiled kernel with split-K over blockIdx.y (needs the caller's workspace) so that
    //                      >= ~2 blocks/CU stream disjoint weight panels; partials reduced in a fixed order
    int splitk = 1, sk_bm = 0, sk_bn = 0, ring_split = 0;
    const int nk_all = (d->K + 63) / 64 + (d->A2 ? (d->K2 + 63) / 64 : 0);
    //   256 < M <= 512   : (r04) the same panel kernel over TWO 256-row groups in one launch (gemm_decode.hip: the two blocks of a weight
    //                      panel run side by side on one XCD and share the panel through its L2) - a decode step of up to 512 clips streams
    //                      the weights once.  Only with a workspace, automatic tuning and never for the prefill form of the fused RoPE.
    static const int dec_on_ = []() { const char* e = getenv("CRAB_DEC_GEMM"); return !(e && e[0] == '0'); }();
    const bool dec512 = d->M > 256 && d->M <= CRAB_DECODE_MAX_ROWS && d->batch <= 1 && d->workspace != nullptr && dec_on_ &&
                        (d->tune == 0 || (d->tune >= 70000 && d->tune < 80000) || (d->tune >= 400 && d->tune < 500)) &&   // 7BBSS: a forced (panel width, K slices); 4SS: the 256 x 256 ring kernel with SS K slices (benchmarking)
                        nk_all >= 8 && !(d->rope_tab && d->rope_S > 1) && (int64_t)d->M * d->N * 4 <= d->workspace_bytes;
    if (d->batch <= 1 && (d->M <= 256 || dec512) && (d->M <= 128 || d->workspace != nullptr)) {
        bool want_split = d->M > 16 && d->workspace != nullptr && nk_all >= 8;
        if (d->tune >= 1 && d->tune <= 4) want_split = false;                      // forced skinny NT
        if (d->tune >= 100) want_split = d->workspace != nullptr;
        if (!want_split && d->M <= 128) {
            // M <= 16 with a fused post-norm (o / down of a decoder layer at the reference's batch sizes): the skinny kernel leaves raw
            // fp32 sums in the workspace (one "slice") and the row-owning reduction kernel applies bias / residual, stores C, the
            // normalised row and the next group's router - one launch instead of rmsnorm + the router's two (CRAB_SKINNY_FUSED=0: off)
            static const int fused_on = []() { const char* e = getenv("CRAB_SKINNY_FUSED"); return !(e && e[0] == '0'); }();
            if (fused_on && crab_rowfin_enabled() && d->tune == 0 && d->M <= 16 && crab_rowfin_ok(d) && (d->ldb & 7) == 0) {
                // r03: the wide two-launch tail (rowfin.hip) - the projection's own router rows ride on the GEMM launch (d->lora_RA), the
                // update, the residual row, its RMSNorm and the NEXT group's router follow in two launches of N / 64 blocks each
                crab_gemm_desc raw = *d;
                raw.tune = 9;
                raw.B2 = d->A2 ? d->B2 : nullptr;                          // a deferred update is not a K segment of the product
                int rc = crab_gemm_skinny_launch(ctx, (hipStream_t)stream, &raw);
                return rc ? rc : crab_rowfin_launch(ctx, (hipStream_t)stream, d);
            }
            if (fused_on && d->tune == 0 && d->M <= 16 && norm_epilogue_ok(d) && d->workspace &&
                (int64_t)d->M * d->N * 4 <= d->workspace_bytes && (d->ldb & 7) == 0) {
                crab_gemm_desc raw = *d;
                raw.tune = 9;                                               // fp32 act(sum + bias) + residual to raw.workspace, no store of C
                int rc = crab_gemm_skinny_launch(ctx, (hipStream_t)stream, &raw);
                return rc ? rc : launch_norm_epilogue(ctx, (hipStream_t)stream, d, (const float*)d->workspace, 1, true);
            }
            int rc = crab_gemm_skinny_launch(ctx, (hipStream_t)stream, d);
            if (rc || crab_skinny_fuses_rope(d)) return rc;             // M <= 16: the rotation and the cache append ran in the epilogue
            return post_norm(ctx, stream, d);
        }
        if (want_split) sk_bm = (d->M <= 64 && d->M <= crab_dec_min_rows()) ? 64 : 128;      // (above the panel kernel's row floor: its regime, sk_bm = 128)
    }
    if (sk_bm) {
        sk_bn = (sk_bm == 64 && d->N <= 4096) ? 64 : 128;          // narrow outputs: 64-wide tiles double the grid instead of the split
        if (d->tune >= 200) sk_bn = 64;
        else if (d->tune >= 100) sk_bn = 128;
        if (sk_bm == 128 && sk_bn == 64) sk_bn = 128;
        long tiles = (long)((d->N + sk_bn - 1) / sk_bn) * ((d->M + sk_bm - 1) / sk_bm);
        splitk = tiles >= 300 ? 1 : (int)((640 + tiles - 1) / tiles);        // a full wave of blocks needs no split (and no partial traffic)
        if (sk_bm == 128 && d->tune < 100) {
            // 128-row tiles (64 < M <= 256): pick the split from a cost model calibrated on the four decoder projections at
            // M = 256 (profiles/README.md): two blocks are resident per CU (512 slots), a K tile of 64 costs ~1.35 us per
            // block when two share a CU, ~6 us of prologue + epilogue per block, and the reduction kernel ~4 us + one pass
            // over each fp32 slab.  qkv (192 tiles): split 4 = 768 blocks = 1.5 rounds -> 66 us, split 2 = 384 blocks -> 59 us.
            double best = 1e30;
            const double slab_us = (double)d->M * d->N * 4.0 / 5.0e6;             // one slab at ~5 TB/s, in us
            for (int sp = 1; sp <= 8 && sp <= nk_all / 4 + (nk_all < 4); ++sp) {
                const long blocks = tiles * sp;
                const double rounds = (double)((blocks + 511) / 512);
                double t = rounds * (6.0 + 1.35 * (double)((nk_all + sp - 1) / sp)) + (sp > 1 ? 4.0 + 0.7 * sp * slab_us : 0.0);
                if (blocks < 256) t *= 256.0 / (double)blocks * 0.5 + 0.5;        // too few blocks: weight panels stream too slowly
                if (t < best) { best = t; splitk = sp; }
            }
        }
        if (d->tune >= 100) splitk = d->tune % 100;
        if (d->tune >= 400 && d->tune < 500 && sk_bm == 128) ring_split = 1;      // benchmarking: 256x256 ring kernel, S K-slices
        if (d->tune < 100 && d->M >= 192 && d->M <= 256 && d->N >= 10240) {
            // wide projections at M ~ 256 (q|k|v: 48 tiles x 5 slices, gate|up: 86 tiles x 2): the 256x256 ring kernel, one
            // block per CU in a single round, beats the 128x128 kernel by 7 / 11 us (profiles/README.md); the narrow ones
            // (o, down: 16 tiles) would need 16 slabs and do not gain
            const long t256 = (long)((d->M + 255) / 256) * ((d->N + 255) / 256);
            int sr = (int)(256 / t256);
            if (sr > 5) sr = 5;
            if (sr >= 2) { ring_split = 1; splitk = sr; }
        }
        if (!ring_split && splitk > nk_all / 4) splitk = nk_all / 4;
        if (splitk < 1) splitk = 1;
        if (splitk > 8 && d->tune < 100) splitk = 8;               // partial slabs: S*M*N*4 bytes written and re-read
        if (splitk > 32) splitk = 32;
        while (splitk > 1 && (int64_t)splitk * d->M * d->N * 4 > d->workspace_bytes) --splitk;
        if (ring_split) {                                              // slices of whole 32-wide K tiles, none empty
            const int nk32 = (d->K + 31) / 32 + (d->A2 ? (d->K2 + 31) / 32 : 0);
            const int per = (nk32 + splitk - 1) / splitk;
            splitk = (nk32 + per - 1) / per;
            if (splitk < 2) { ring_split = 0; splitk = 1; }
        }
    }
    // ---- decode regime, 128 < M <= 256: the batch-tall narrow-panel kernel (gemm_decode.hip); tune 70000 + BN * 100 + S forces
    // a decomposition (80000 + ...: on the 8-wave kernel instead of the producer / consumer one), any other non-zero tune keeps the older kernels (A/B runs), CRAB_DEC_GEMM=0 disables it process-wide
    int dec_bn = 0;
    // r06: the panel kernel's row floor is 64 (r02-r05: 128).  64 < M <= 128 used to take the 128 x 128 LDS-DMA kernel with K slices (two blocks per CU,
    // each re-reading its activation tile per weight tile); the panel kernel streams every weight byte once for the whole batch whatever M is (the row
    // fragments beyond M are masked).  Measured, 32 layers + lm_head of Llama-2-7B (scripts/exp/dec_min_rows.py, profiles/r06_dec_min_rows.txt):
    // M = 72: 4.89 -> 4.47 ms, 96: 5.06 -> 4.47, 128: 5.36 -> 4.58.  CRAB_DEC_MIN_ROWS=128 restores the old floor (A/B runs).
    const int dec_min_rows = crab_dec_min_rows();
    if (sk_bm == 128 && d->M > dec_min_rows && d->workspace) {
        static const int dec_on = []() { const char* e = getenv("CRAB_DEC_GEMM"); return !(e && e[0] == '0'); }();
        const int nk32 = (d->K + 63) / 64 + (d->A2 ? (d->K2 + 63) / 64 : 0);      // K slots of the panel kernel (64 wide)
        if (d->tune >= 70000 && d->tune < 90000) {                      // 8xxxx: the same decomposition on the 8-wave kernel (A/B, tests)
            dec_bn = (d->tune / 100) % 100; splitk = d->tune % 100;
            if (splitk < 1) splitk = 1;
            const int per = (nk32 + splitk - 1) / splitk;
            splitk = (nk32 + per - 1) / per;                              // no empty slice
            while (splitk > 1 && (int64_t)splitk * d->M * d->N * 4 > d->workspace_bytes) --splitk;
            const int per2 = (nk32 + splitk - 1) / splitk;                // the clamp may leave a remainder slice empty: normalise again
            splitk = (nk32 + per2 - 1) / per2;
        } else if (d->tune == 91601) {                                  // A/B: 160-wide panels, one K slice
            dec_bn = 160; splitk = 1;
        } else if (d->tune == 0 && dec_on) {
            if (d->M > 256) dec2_choose(d, nk32, &dec_bn, &splitk);      // two row groups per block: its own calibration, 96- / 64-wide panels only
            else dec_choose(d, nk32, &dec_bn, &splitk);
        }
        if (dec_bn) ring_split = 0;
    }
    GemmP p;
    p.splitk = splitk; p.part = (float*)d->workspace;
    p.A = (const bf16_t*)d->A; p.B = (const bf16_t*)d->B; p.C = d->C;
    p.bias = (const bf16_t*)d->bias; p.R = (const bf16_t*)d->R;
    p.A2 = (const bf16_t*)d->A2; p.B2 = (const bf16_t*)d->B2;
    p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ldr = d->ldr; p.lda2 = d->lda2; p.ldb2 = d->ldb2;
    p.M = d->M; p.N = d->N; p.K = d->K; p.K2 = d->A2 ? d->K2 : 0; p.act = d->act; p.c_fp32 = crab_cflags(d);
    p.res_scale = d->res_scale;
    int batch = d->batch > 1 ? d->batch : 1;
    p.nb0 = (batch > 1 && d->nb0 > 0) ? d->nb0 : 1;
    p.sA0 = d->sA0; p.sA1 = d->sA1; p.sB0 = d->sB0; p.sB1 = d->sB1; p.sC0 = d->sC0; p.sC1 = d->sC1;
    p.sR0 = d->sR0; p.sR1 = d->sR1; p.sBias0 = d->sBias0; p.sBias1 = d->sBias1;
    if (batch == 1) { p.sA0 = p.sA1 = p.sB0 = p.sB1 = p.sC0 = p.sC1 = p.sR0 = p.sR1 = p.sBias0 = p.sBias1 = 0; }
    hipStream_t s = (hipStream_t)stream;
    if (sk_bm) {
        p.tiles_m = (d->M + sk_bm - 1) / sk_bm; p.tiles_n = (d->N + sk_bn - 1) / sk_bn;
        dim3 grid(p.tiles_m * p.tiles_n, splitk);
        if (sk_bm == 64 && sk_bn == 128) hipLaunchKernelGGL((gemm_bt_kernel<64, 128>), grid, dim3(256), 0, s, p);
        else if (sk_bm == 64) hipLaunchKernelGGL((gemm_bt_kernel<64, 64>), grid, dim3(256), 0, s, p);
        else if (d->tune == 300) hipLaunchKernelGGL((gemm_bt_kernel<128, 128>), grid, dim3(256), 0, s, p);
        else if (dec_bn) {
            int rc2 = crab_gemm_dec_launch(ctx, s, d, dec_bn, splitk, p.part, d->tune >= 80000 ? 2 : 1);
            if (rc2) return rc2;
        }
        else {                                                        // 128-row tiles: LDS-DMA staged kernel, K split over blockIdx.y
            int rc2 = crab_gemm_glds_launch(ctx, s, d, splitk, p.part, ring_split);
            if (rc2) return rc2;
        }
        int rc = crab_check_launch(ctx, "gemm_bt_kernel(split-K)");
        if (rc) return rc;
        if (splitk == 1) return post_norm(ctx, stream, d);
        if (d->rope_tab && (d->rope_d & 7) == 0 && (d->ldc & 3) == 0 && (d->N & 3) == 0) {
            const bool g8 = (d->rope_d & 15) == 0 && (d->ldc & 7) == 0 &&
                            (((uintptr_t)d->C | (uintptr_t)d->rope_k_cache | (uintptr_t)d->rope_v_cache) & 15) == 0;
            const long nthr = (long)d->M * (d->rope_H + 2 * d->rope_Hk) * (d->rope_d >> (g8 ? 4 : 3));
#define ROPE_EPI(G_) hipLaunchKernelGGL((splitk_epilogue_rope_kernel<G_>), dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, p.part, splitk, \
                                        d->M, d->N, p.bias, (bf16_t*)d->C, (long)d->ldc, d->rope_tab, (bf16_t*)d->rope_k_cache,                 \
                                        (bf16_t*)d->rope_v_cache, d->rope_pos_dev, d->rope_pos0, d->rope_H, d->rope_Hk, d->rope_d, d->rope_Tmax,        \
                                        d->rope_row_off)
            if (g8) ROPE_EPI(8); else ROPE_EPI(4);
#undef ROPE_EPI
            return crab_check_launch(ctx, "splitk_epilogue_rope_kernel");
        }
        if (norm_epilogue_ok(d)) return launch_norm_epilogue(ctx, s, d, p.part, splitk);
        if (d->act == ACT_SWIGLU_PAIR && !d->c_fp32 && (d->N & 15) == 0 && (d->ldc & 7) == 0 && ((uintptr_t)d->C & 15) == 0) {
            const long nthr8 = (long)d->M * (d->N >> 4);
            hipLaunchKernelGGL(splitk_epilogue_swiglu8_kernel, dim3((unsigned)((nthr8 + 255) / 256)), dim3(256), 0, s, p.part, splitk, d->M, d->N,
                               p.bias, (bf16_t*)d->C, (long)d->ldc);
            rc = crab_check_launch(ctx, "splitk_epilogue_swiglu8_kernel");
            if (rc) return rc;
            return post_norm(ctx, stream, d);
        }
        long nthr = (long)d->M * ((d->N + 3) / 4);
        hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((nthr + 255) / 256)), dim3(256), 0, s, p.part, splitk, d->M, d->N, p.bias, d->act,
                           p.R, (long)d->ldr, d->res_scale, d->C, (long)d->ldc, crab_cflags(d));
        rc = crab_check_launch(ctx, "splitk_epilogue_kernel");
        if (rc) return rc;
        return post_norm(ctx, stream, d);
    }
    // tile choice: 128x128 when it fills the chip, 64x64 for small / skinny problems
    long big_tiles = (long)((d->M + 127) / 128) * ((d->N + 127) / 128) * batch;
    bool small = (d->M <= 64) || (d->N <= 64) || big_tiles < 192;
    if (!small && d->tune != 300) {
        // 128x128 tiles: LDS-DMA staged kernel (gemm_glds.hip); tune == 300 keeps the register-staged variant for A/B runs
        int rc = crab_gemm_glds_launch(ctx, s, d, 1, nullptr, 0);
        return rc ? rc : post_norm(ctx, stream, d);
    }
    if (!small) {
        p.tiles_m = (d->M + 127) / 128; p.tiles_n = (d->N + 127) / 128;
        dim3 grid(p.tiles_m * p.tiles_n, batch);
        hipLaunchKernelGGL((gemm_bt_kernel<128, 128>), grid, dim3(256), 0, s, p);
    } else {
        p.tiles_m = (d->M + 63) / 64; p.tiles_n = (d->N + 63) / 64;
        dim3 grid(p.tiles_m * p.tiles_n, batch);
        hipLaunchKernelGGL((gemm_bt_kernel<64, 64>), grid, dim3(256), 0, s, p);
    }
    int rc = crab_check_launch(ctx, "gemm_bt_kernel");
    return rc ? rc : post_norm(ctx, stream, d);
}

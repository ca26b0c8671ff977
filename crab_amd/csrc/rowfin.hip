// Small-batch layer tail (M <= 16 rows: the reference's batch sizes 1 and 8, scripts/quick_start.py:43,
// scripts/finetune/inference_hyper_lora.py:1477): what follows the o_proj / down_proj product of a decoder layer
// (models/modeling_llama.py:805-827 residual adds + LlamaRMSNorm :112-117; hyper-LoRA peft_hyper/tuners/lora.py:338-350).
//
// r02 ran this as FOUR launches per projection - router partial product, router mix, the GEMM, a row-owning reduction with
// 1-3 blocks per row (8 us: one block walks 90 KB of router rows) - 35 us of every 135 us layer at batch 1.  Kernel boundaries are
// cheap on this chip (1.2-1.9 us, guide price list "boundary"); a grid barrier is not (4.1+ us, "barrier-xcd"); what costs is a
// small kernel that is also NARROW.  So the tail is two WIDE launches behind a GEMM that already carries the router product:
//
//   gemm_skinny_dma_kernel   S = act(A.W^T + bias) + R in fp32, and - 16 extra weight rows = the group's own [R;A] - T = A.[R;A]^T
//   rowfin_apply_kernel      column slices of 64: y = S + scaling * sum_i softmax(T_route)_i B_i T_A   (the hyper-LoRA update of THIS
//                            projection, applied here because T only exists once the GEMM launch is over), x = y stored (fp32 stream, or bf16),
//                            partial sum of squares per (slice, row); r04 (<.., true> instantiation): ALSO the partial router products of the
//                            NEXT group on its slice, formed on the unnormalised row times the norm weight - t = rstd * ((y (.) w) . [R;A]^T), rstd
//                            being a per-row scalar - with the 256 threads split as (router row, 16 of the 64 columns) over the live rows only
//   rowfin_norm_mix_kernel   column slices of 64: rstd from the partial sums of squares, h = rmsnorm(x) * w stored; every block also takes one
//                            (row, projection) pair of the next group: its slices' partials summed in a fixed order, times rstd, fp32 softmax,
//                            u = scaling * p (x) (h A^T).  One memory round trip; no block waits for another.
// r03's second launch (rowfin_route_kernel, kept behind CRAB_ROWFIN_TAIL=1 for A/B runs) formed the partials itself, AFTER rstd, and needed an arrival
// chain for them - drained write-through partial stores, a returning atomic, the last arriver's acquire + reload: five dependent memory round trips,
// 8.4 us at one clip against 5.3 for the kernel above (the apply launch grows from 5.5 to 6.2 us): 117 -> 112 us per layer, 3.76 -> 3.60 ms per step.
// Nothing spins in either form.  Deterministic (fixed orders).
#include "common.h"
#include "crab_internal.h"
#include <stdlib.h>

namespace {

constexpr int RF_CW = 64;                        // columns per block
constexpr int RF_TJ = 64;                        // router rows per (slice, row) record of the partial buffer

constexpr int RF_SKX = 8;                        // K-range partials of the ride-along router product (SKX of skinny.hip)
constexpr int RF_G = 4;                          // slice groups of the last-arriver reduction

struct RowfinApplyP {
    const float* S; const float* T;              // S [M][N] fp32; T [RF_SKX][16][16] partial router products: nl route logits then r lora_A
    const bf16_t* B2; long ldb2; int k2;         // lora_B [N][k2 >= nl * r]                                    (T == NULL: no adapter)
    bf16_t* X; long ldx;                         // out: the row as stored (residual stream): bf16, or fp32 in the XF instantiation (ldx in fp32 elements)
    float* ssq;                                  // out: [slices][16] partial sums of squares of the STORED values
    unsigned* counter;                           // arrival counter of the route kernel, zeroed here
    int M, N, nl, r; float scaling;
    // RP instantiation (r04, the two-launch tail without an arrival chain): the NEXT group's router partials are formed HERE, on the un-normalised
    // row times the norm weight - t = rstd * ((y (.) w) . [R;A]^T), rstd being a per-row scalar the second launch applies - so that the second
    // launch has nothing to wait for (rowfin_norm_mix_kernel)
    const void* nw; const bf16_t* RA; long ldra; int used; float* tpart;      // nw: bf16, or fp32 in the WF instantiations (crab_gemm_desc.norm_w_fp32)
};

template <bool XF, bool RP = false, bool WF = false>
__global__ __launch_bounds__(256) void rowfin_apply_kernel(RowfinApplyP p) {
    __shared__ uint32_t b2s[RF_CW][17];          // lora_B rows of this slice, 32 k columns as 16 words + 1 pad word (conflict-free row walk)
    __shared__ float us[16][32];
    __shared__ float ts[16][16];
    __shared__ __attribute__((aligned(16))) bf16_t ras[RP ? RF_TJ : 1][RF_CW];       // RP: the slice of the next group's [R;A] (8 KB)
    const int tid = threadIdx.x, m = tid >> 4, q = tid & 15;
    const int c0 = blockIdx.x * RF_CW, c = c0 + q * 4;
    if (blockIdx.x == 0 && tid == 0) *p.counter = 0u;
    const int nu = p.T ? p.nl * p.r : 0;         // <= 32
    // every global load of the kernel is issued here, before anything waits: one memory round trip deep
    const bool live = m < p.M && c < p.N;
    f32x4_t y = {0.f, 0.f, 0.f, 0.f};
    if (live) y = *reinterpret_cast<const f32x4_t*>(p.S + (long)m * p.N + c);
    u32x4 rv[2] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
    float ww[4] = {0.f, 0.f, 0.f, 0.f};
    const int used16 = RP ? (p.used + 15) & ~15 : 0;
    if (RP) {                                     // unconditional loads at clamped addresses, masked afterwards (see rowfin_route_kernel)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * 256, j = idx >> 3, ch = idx & 7;
            rv[i] = *reinterpret_cast<const u32x4*>(p.RA + (long)min(j, p.used - 1) * p.ldra + min(c0 + ch * 8, p.N - 8));
        }
        ld_par4<WF>(p.nw, min(c, p.N - 4), ww);
    }
    u32x4 bv = {0u, 0u, 0u, 0u};
    float tsum = 0.f;
    if (p.T) {
        const int row = tid >> 2, ch = tid & 3;  // 64 rows x 4 chunks of 8 columns
        if (c0 + row < p.N && ch * 8 < p.k2) bv = *reinterpret_cast<const u32x4*>(p.B2 + (long)(c0 + row) * p.ldb2 + ch * 8);
        if (m < p.M) {
            float tv[RF_SKX];
#pragma unroll
            for (int e = 0; e < RF_SKX; ++e) tv[e] = p.T[((long)e * 16 + m) * 16 + q];
#pragma unroll
            for (int e = 0; e < RF_SKX; ++e) tsum += tv[e];               // K ranges in order
        }
        b2s[row][ch * 4 + 0] = bv[0]; b2s[row][ch * 4 + 1] = bv[1]; b2s[row][ch * 4 + 2] = bv[2]; b2s[row][ch * 4 + 3] = bv[3];
        us[m][q] = 0.f; us[m][q + 16] = 0.f;
        ts[m][q] = tsum;
    }
    if (RP) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int idx = tid + i * 256, j = idx >> 3, ch = idx & 7;
            if (!(j < p.used && c0 + ch * 8 < p.N)) rv[i] = u32x4{0u, 0u, 0u, 0u};
            if (j < used16) *reinterpret_cast<u32x4*>(&ras[j][ch * 8]) = rv[i];
        }
    }
    __syncthreads();
    if (p.T && q == 0 && m < p.M) {
        // the routing mix of lora_mix_reduce_kernel (skinny.hip), same expression: u = bf16(scaling * softmax(route)_i * (x A^T)_j)
        const float* t = &ts[m][0];
        float e[8], mx = -INFINITY;
        for (int i = 0; i < p.nl; ++i) mx = fmaxf(mx, t[i]);
        float sum = 0.f;
        for (int i = 0; i < p.nl; ++i) { e[i] = expf(t[i] - mx); sum += e[i]; }
        const float inv = 1.0f / sum;
        for (int i = 0; i < p.nl; ++i)
            for (int j = 0; j < p.r; ++j) us[m][i * p.r + j] = bf2f(f2bf(p.scaling * e[i] * inv * t[p.nl + j]));
    }
    __syncthreads();
    float ss = 0.f;
    if (live) {
        if (nu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = 0.f;
                for (int k = 0; k < nu; k += 2) {
                    const uint32_t w = b2s[q * 4 + j][k >> 1];
                    a += us[m][k] * lo_bf(w) + us[m][k + 1] * hi_bf(w);      // us[][nu] is zero when nu is odd
                }
                y[j] += a;
            }
        }
        if (XF) {                                 // fp32 residual stream: stored and normalised unrounded
            *reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(p.X) + (long)m * p.ldx + c) = y;
            ss = (y[0] * y[0] + y[1] * y[1]) + (y[2] * y[2] + y[3] * y[3]);
        } else {
            const uint32_t w0 = pack_bf2(y[0], y[1]), w1 = pack_bf2(y[2], y[3]);
            *reinterpret_cast<u32x2*>(p.X + (long)m * p.ldx + c) = u32x2{w0, w1};
            const float x0 = lo_bf(w0), x1 = hi_bf(w0), x2 = lo_bf(w1), x3 = hi_bf(w1);      // the norm sees the stored bf16 values
            ss = (x0 * x0 + x1 * x1) + (x2 * x2 + x3 * x3);
        }
    }
    ss = row16_sum(ss);
    if (q == 0) p.ssq[blockIdx.x * 16 + m] = m < p.M ? ss : 0.f;
    if (RP) {
        // partial router products of this slice on (stored row) * norm weight, fp32.  The row times the weight goes through LDS so that the 256
        // threads split the work as (router row j = tid / 4, 16 of the slice's 64 columns) and walk only the M live rows: 16 fmas + two lane
        // exchanges per row and thread (one clip: 16 fmas in all) instead of a 16-lane reduction per router row in every one of the 16 row groups
        __shared__ __attribute__((aligned(16))) float yws[16][RF_CW];
        f32x4_t hv = {0.f, 0.f, 0.f, 0.f};
        if (live) {
            float x0 = y[0], x1 = y[1], x2 = y[2], x3 = y[3];
            if (!XF) { x0 = bf2f(f2bf(x0)); x1 = bf2f(f2bf(x1)); x2 = bf2f(f2bf(x2)); x3 = bf2f(f2bf(x3)); }      // the stored bf16 row
            hv = f32x4_t{x0 * ww[0], x1 * ww[1], x2 * ww[2], x3 * ww[3]};
        }
        *reinterpret_cast<f32x4_t*>(&yws[m][q * 4]) = hv;
        __syncthreads();
        const int jr = tid >> 2, part = tid & 3;
        if (jr < used16) {                                       // wave-uniform up to the last wave
            float wf[16];
            const u32x4 w0 = *reinterpret_cast<const u32x4*>(&ras[jr][part * 16]), w1 = *reinterpret_cast<const u32x4*>(&ras[jr][part * 16 + 8]);
#pragma unroll
            for (int e = 0; e < 4; ++e) { wf[2 * e] = lo_bf(w0[e]); wf[2 * e + 1] = hi_bf(w0[e]); wf[8 + 2 * e] = lo_bf(w1[e]); wf[8 + 2 * e + 1] = hi_bf(w1[e]); }
            for (int mm = 0; mm < p.M; ++mm) {
                float a = 0.f;
#pragma unroll
                for (int v = 0; v < 4; ++v) {
                    const f32x4_t yv = *reinterpret_cast<const f32x4_t*>(&yws[mm][part * 16 + v * 4]);
                    a += (yv[0] * wf[4 * v] + yv[1] * wf[4 * v + 1]) + (yv[2] * wf[4 * v + 2] + yv[3] * wf[4 * v + 3]);
                }
                a += __shfl_xor(a, 1, 64);
                a += __shfl_xor(a, 2, 64);
                if (part == 0) p.tpart[((long)blockIdx.x * 16 + mm) * RF_TJ + jr] = a;
            }
        }
    }
}

// Second launch of the r04 tail: rstd from the slices' sums of squares, h = rmsnorm(x) * w stored for this block's 64 columns, and - every block takes
// (row, projection) pairs of the next group's router - the slices' partials summed in a fixed order, scaled by rstd, fp32 softmax over the route
// logits, u = scaling * p (x) (h A^T).  One memory round trip, nothing waits on another block (the arrival ticket, the drained partial stores and
// the last arriver's acquire + reload of rowfin_route_kernel are gone: 8.4 -> ~4.5 us at one clip).
struct RowfinMixP {
    const bf16_t* X; long ldx; const float* ssq; int nb;
    const void* nw; float eps; bf16_t* H; long ldh;
    const float* tpart; bf16_t* U; long ldu; int nproj, nl, r, ucols; float scaling;      // tpart == NULL: no next-group router
    int M, N;
};

template <bool XF, bool WF = false>
__global__ __launch_bounds__(256) void rowfin_norm_mix_kernel(RowfinMixP p) {
    __shared__ float rs[16];
    __shared__ float Tg[RF_G][RF_TJ];
    __shared__ float Tm[RF_TJ];
    const int tid = threadIdx.x, m = tid >> 4, q = tid & 15;
    const int c0 = blockIdx.x * RF_CW, c = c0 + q * 4;
    const int nj = p.nl + p.r, npairs = p.tpart ? p.M * p.nproj : 0;
    const int per = (p.nb + RF_G - 1) / RF_G;
    // ---- every global load first: the partial sums of squares, x, w, and the router partials of this block's first (row, projection) pair
    float sp[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sp[i] = p.ssq[min(q + i * 16, p.nb - 1) * 16 + m];       // nb <= 128 (host)
    const bool live = m < p.M && c < p.N;
    u32x2 xw = {0u, 0u};
    f32x4_t xf = {0.f, 0.f, 0.f, 0.f};
    if (XF) xf = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(p.X) + (long)min(m, p.M - 1) * p.ldx + min(c, p.N - 4));
    else xw = *reinterpret_cast<const u32x2*>(p.X + (long)min(m, p.M - 1) * p.ldx + min(c, p.N - 4));
    float ww[4];
    ld_par4<WF>(p.nw, min(c, p.N - 4), ww);
    const int g = tid >> 6, j = tid & 63;
    float tacc = 0.f;
    int k = blockIdx.x;
    constexpr int PER = 128 / RF_G;                              // nb <= 128: at most 32 slices per group, all their loads in flight together
    float tv[PER];
    const bool mine = k < npairs && j < nj;
    {
        const int kk = mine ? k : 0, jc = mine ? j : 0;
        const int mm = kk / p.nproj, pj = kk % p.nproj;
        const float* src = (p.tpart ? p.tpart : p.ssq) + (p.tpart ? (long)mm * RF_TJ + pj * nj + jc : 0);
        const int b0 = g * per;
#pragma unroll
        for (int i = 0; i < PER; ++i) tv[i] = p.tpart ? src[(long)min(b0 + i, p.nb - 1) * 16 * RF_TJ] : 0.f;      // clamped, masked below
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) sp[i] = q + i * 16 < p.nb ? sp[i] : 0.f;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += sp[i];
    ss = row16_sum(ss);
    const float rstd = rsqrtf(ss / (float)p.N + p.eps);
    if (q == 0) rs[m] = rstd;
    if (live) {
        float h0, h1, h2, h3;
        if (XF) {
            h0 = xf[0] * rstd * ww[0]; h1 = xf[1] * rstd * ww[1];
            h2 = xf[2] * rstd * ww[2]; h3 = xf[3] * rstd * ww[3];
        } else {
            h0 = bf2f(f2bf(lo_bf(xw[0]) * rstd)) * ww[0]; h1 = bf2f(f2bf(hi_bf(xw[0]) * rstd)) * ww[1];
            h2 = bf2f(f2bf(lo_bf(xw[1]) * rstd)) * ww[2]; h3 = bf2f(f2bf(hi_bf(xw[1]) * rstd)) * ww[3];
        }
        *reinterpret_cast<u32x2*>(p.H + (long)m * p.ldh + c) = u32x2{pack_bf2(h0, h1), pack_bf2(h2, h3)};
    }
    if (mine) {
        const int b0 = g * per, b1 = min(p.nb, b0 + per);
#pragma unroll
        for (int i = 0; i < PER; ++i) tacc += b0 + i < b1 ? tv[i] : 0.f;                   // slice order inside the group
    }
    // ---- router of the next group: (row, projection) pairs dealt round-robin to the blocks (one pair per block unless nb < M * nproj)
    for (; k < npairs; k += gridDim.x) {
        const int mm = k / p.nproj, pj = k % p.nproj;
        if (k != (int)blockIdx.x) {                             // further pairs of a narrow projection: their partials are loaded here
            tacc = 0.f;
            if (j < nj) {
                const float* src = p.tpart + (long)mm * RF_TJ + pj * nj + j;
                const int b0 = g * per, b1 = min(p.nb, b0 + per);
                for (int b = b0; b < b1; ++b) tacc += src[(long)b * 16 * RF_TJ];
            }
        }
        if (j < nj) Tg[g][j] = tacc;
        __syncthreads();                                        // also orders rs[] (first trip)
        if (tid < nj) Tm[tid] = (((Tg[0][tid] + Tg[1][tid]) + Tg[2][tid]) + Tg[3][tid]) * rs[mm];
        __syncthreads();
        bf16_t* u = p.U + (long)mm * p.ldu;
        if (tid == 0) {
            float e[8], mx = -INFINITY;
            for (int i = 0; i < p.nl; ++i) mx = fmaxf(mx, Tm[i]);
            float sum = 0.f;
            for (int i = 0; i < p.nl; ++i) { e[i] = expf(Tm[i] - mx); sum += e[i]; }
            const float inv = 1.0f / sum;
            for (int i = 0; i < p.nl; ++i)
                for (int jj = 0; jj < p.r; ++jj) u[pj * p.nl * p.r + i * p.r + jj] = f2bf(p.scaling * e[i] * inv * Tm[p.nl + jj]);
        } else if (tid == 64 && pj == 0) {
            for (int cc = p.nproj * p.nl * p.r; cc < p.ucols; ++cc) u[cc] = 0;
        }
        __syncthreads();                                        // Tg / Tm are reused by the next pair
    }
}

struct RowfinRouteP {
    const bf16_t* X; long ldx; const float* ssq; int nb;     // nb column slices (== gridDim.x)
    const bf16_t* nw; float eps; bf16_t* H; long ldh;
    const bf16_t* RA; long ldra; bf16_t* U; long ldu; int nproj, nl, r, ucols; float scaling;   // the NEXT group's router (RA == NULL: none)
    float* tpart; unsigned* counter;                          // [nb][16][RF_TJ] fp32 partial router products; arrival count
    int M, N;
};

template <bool XF>
__global__ __launch_bounds__(256) void rowfin_route_kernel(RowfinRouteP p) {
    __shared__ __attribute__((aligned(16))) bf16_t ras[RF_TJ][RF_CW];       // 8 KB: the slice of [R;A]
    __shared__ __attribute__((aligned(16))) float Tg[RF_G][16][RF_TJ];     // 16 KB: slice-group sums of the last arriver
    __shared__ float Tm[16][RF_TJ];
    __shared__ unsigned s_old;
    const int tid = threadIdx.x, m = tid >> 4, q = tid & 15;
    const int c0 = blockIdx.x * RF_CW, c = c0 + q * 4;
    const int used = p.RA ? p.nproj * (p.nl + p.r) : 0;       // <= RF_TJ (host)
    const int used16 = (used + 15) & ~15;
    // ---- every global load first (one round trip): the [R;A] slice (two 16-byte chunks per thread), the partial sums of squares, x, w.
    // The loads are UNCONDITIONAL (clamped addresses, values masked afterwards): with `cond ? load : 0` the compiler sank the first use into
    // the first conditional block and waited there (vmcnt(0) behind the first 4-byte load: a second, serialised memory round trip; ISA, r03)
    u32x4 rv[2];
    const bf16_t* ra = p.RA ? p.RA : p.nw;                    // no adapter: any readable row (the values are masked), so that no branch
    const long ldra = p.RA ? p.ldra : 0;                      // separates these loads from the ones below
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * 256, j = idx >> 3, ch = idx & 7;
        const int jc = p.RA ? min(j, used - 1) : 0, cc = min(c0 + ch * 8, p.N - 8);
        rv[i] = *reinterpret_cast<const u32x4*>(ra + (long)jc * ldra + cc);
    }
    float sp[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) sp[i] = p.ssq[min(q + i * 16, p.nb - 1) * 16 + m];       // nb <= 128 (host)
    const bool live = m < p.M && c < p.N;
    u32x2 xw = {0u, 0u};
    f32x4_t xf = {0.f, 0.f, 0.f, 0.f};
    if (XF) xf = *reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(p.X) + (long)min(m, p.M - 1) * p.ldx + min(c, p.N - 4));
    else xw = *reinterpret_cast<const u32x2*>(p.X + (long)min(m, p.M - 1) * p.ldx + min(c, p.N - 4));
    u32x2 ww = *reinterpret_cast<const u32x2*>(p.nw + min(c, p.N - 4));
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * 256, j = idx >> 3, ch = idx & 7;
        if (!(j < used && c0 + ch * 8 < p.N)) rv[i] = u32x4{0u, 0u, 0u, 0u};
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) sp[i] = q + i * 16 < p.nb ? sp[i] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int idx = tid + i * 256, j = idx >> 3, ch = idx & 7;
        if (j < used16) *reinterpret_cast<u32x4*>(&ras[j][ch * 8]) = rv[i];
    }
    // ---- rstd of row m: the slices' partial sums of squares in a fixed order (8 strided terms per lane, then a butterfly)
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += sp[i];
    ss = row16_sum(ss);
    const float rstd = rsqrtf(ss / (float)p.N + p.eps);
    float hf[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        float h0, h1, h2, h3;
        if (XF) {
            h0 = xf[0] * rstd * lo_bf(ww[0]); h1 = xf[1] * rstd * hi_bf(ww[0]);
            h2 = xf[2] * rstd * lo_bf(ww[1]); h3 = xf[3] * rstd * hi_bf(ww[1]);
        } else {
            h0 = bf2f(f2bf(lo_bf(xw[0]) * rstd)) * lo_bf(ww[0]); h1 = bf2f(f2bf(hi_bf(xw[0]) * rstd)) * hi_bf(ww[0]);
            h2 = bf2f(f2bf(lo_bf(xw[1]) * rstd)) * lo_bf(ww[1]); h3 = bf2f(f2bf(hi_bf(xw[1]) * rstd)) * hi_bf(ww[1]);
        }
        const uint32_t o0 = pack_bf2(h0, h1), o1 = pack_bf2(h2, h3);
        *reinterpret_cast<u32x2*>(p.H + (long)m * p.ldh + c) = u32x2{o0, o1};
        hf[0] = lo_bf(o0); hf[1] = hi_bf(o0); hf[2] = lo_bf(o1); hf[3] = hi_bf(o1);       // the router sees the stored bf16 row
    }
    if (!p.RA) return;
    __syncthreads();
    // ---- partial router products of this slice: t[m][j] = sum over the slice's columns of h * [R;A][j]; lane q keeps j = 4q .. 4q+3
    f32x4_t keep = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int jj = 0; jj < RF_TJ / 16; ++jj) {
        if (jj * 16 < used16) {
#pragma unroll
            for (int q2 = 0; q2 < 16; ++q2) {
                const u32x2 w = *reinterpret_cast<const u32x2*>(&ras[jj * 16 + q2][q * 4]);
                float a = (hf[0] * lo_bf(w[0]) + hf[1] * hi_bf(w[0])) + (hf[2] * lo_bf(w[1]) + hf[3] * hi_bf(w[1]));
                a = row16_sum(a);
                if (q == jj * 4 + (q2 >> 2)) keep[q2 & 3] = a;
            }
        }
    }
    // one 16-byte write-through (sc1) store per lane, drained by every storing wave before the arrival (guide G16 R1: no release fence)
    if (4 * q < used16) {
        float* dst = p.tpart + ((long)blockIdx.x * 16 + m) * RF_TJ + 4 * q;
        if (m >= p.M) keep = f32x4_t{0.f, 0.f, 0.f, 0.f};
        asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(dst), "v"(keep) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) s_old = __hip_atomic_fetch_add(p.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_old != (unsigned)(p.nb - 1)) return;                 // not the last slice to arrive: done, nobody waits
    // ---- last arriver: every slice's partials are published.  One acquire drops this CU's stale lines, then plain loads:
    // item = (slice group g, row, 4 router rows): its slices' 16-byte records all in flight, summed in slice order; then the groups in order
    if (tid == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    const int n4 = used16 >> 2, per = (p.nb + RF_G - 1) / RF_G;
    for (int idx = tid; idx < RF_G * p.M * n4; idx += 256) {
        const int j4 = idx % n4, mm = (idx / n4) % p.M, g = idx / (n4 * p.M);
        const int b0 = g * per, b1 = min(p.nb, b0 + per);
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        const float* src = p.tpart + (long)mm * RF_TJ + 4 * j4;
        const long st = 16L * RF_TJ;
        int b = b0;
        for (; b + 8 <= b1; b += 8) {
            f32x4_t v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x4_t*>(src + (b + i) * st);
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += v[i];
        }
        for (; b < b1; ++b) acc += *reinterpret_cast<const f32x4_t*>(src + b * st);
        *reinterpret_cast<f32x4_t*>(&Tg[g][mm][4 * j4]) = acc;
    }
    __syncthreads();
    for (int idx = tid; idx < p.M * used16; idx += 256) {
        const int mm = idx / used16, j = idx % used16;
        Tm[mm][j] = ((Tg[0][mm][j] + Tg[1][mm][j]) + Tg[2][mm][j]) + Tg[3][mm][j];
    }
    __syncthreads();
    if (tid == 0) *p.counter = 0u;                             // ready for the next call even without the apply kernel in front
    const int per_row = p.nproj + 1;
    if (tid >= 16 * per_row) return;
    const int mm = tid / per_row, pj = tid % per_row;
    if (mm >= p.M) return;
    bf16_t* u = p.U + (long)mm * p.ldu;
    const int usedu = p.nproj * p.nl * p.r;
    if (pj == p.nproj) {
        for (int cc = usedu; cc < p.ucols; ++cc) u[cc] = 0;
        return;
    }
    const float* t = &Tm[mm][pj * (p.nl + p.r)];
    float e[8], mx = -INFINITY;
    for (int i = 0; i < p.nl; ++i) mx = fmaxf(mx, t[i]);
    float sum = 0.f;
    for (int i = 0; i < p.nl; ++i) { e[i] = expf(t[i] - mx); sum += e[i]; }
    const float inv = 1.0f / sum;
    for (int i = 0; i < p.nl; ++i)
        for (int j = 0; j < p.r; ++j) u[pj * p.nl * p.r + i * p.r + j] = f2bf(p.scaling * e[i] * inv * t[p.nl + j]);
}

}  // namespace

// Workspace carve behind the fp32 sums S [M][N] (all offsets 256-byte aligned): T [RF_SKX][16][16] fp32 | ssq [nb][16] fp32 |
// tpart [nb][16][RF_TJ] fp32 | counter
static inline int64_t rf_align(int64_t x) { return (x + 255) & ~(int64_t)255; }

extern "C" int64_t crab_rowfin_workspace(int M, int N) {
    const int nb = (N + RF_CW - 1) / RF_CW;
    return rf_align((int64_t)M * N * 4) + rf_align(RF_SKX * 16 * 16 * 4) + rf_align((int64_t)nb * 16 * 4) + rf_align((int64_t)nb * 16 * RF_TJ * 4) + 256;
}

float* crab_rowfin_T(const crab_gemm_desc* d) { return (float*)((char*)d->workspace + rf_align((int64_t)d->M * d->N * 4)); }

// ONE predicate for "the in-call hyper-LoRA (crab_gemm_desc.lora_RA) can be evaluated by this tail": shared by crab_rowfin_ok, the layer
// sequencer (llama_layer.hip: run_group) and - through the C-ABI - crab_amd/peft_hyper.py, so that an adapter outside these limits
// (e.g. lora_r = 16: r is a CLI argument of the reference, 8 is only its default) takes the router + K-extension path instead of failing.
extern "C" int crab_rowfin_lora_ok(int nl, int r, int N) {
    return nl >= 1 && r >= 1 && nl <= 8 && nl + r <= 16 && nl * r <= 32 && N <= 128 * RF_CW && (N & 7) == 0;
}

// CRAB_ROWFIN=0 disables the tail process-wide (one parse for the library; crab_amd/ops.py reads the same variable the same way)
bool crab_rowfin_enabled() {
    static const int on = []() { const char* e = getenv("CRAB_ROWFIN"); return !(e && e[0] == '0' && e[1] == 0); }();
    return on != 0;
}

bool crab_rowfin_ok(const crab_gemm_desc* d) {
    if (!d->norm_w || !d->norm_out || d->M > 16 || !d->workspace) return false;
    if (d->norm_w_fp32 && (!d->c_fp32 || ((uintptr_t)d->norm_w & 15))) return false;          // fp32 norm weights: with the fp32 residual stream only
    if (d->c_fp32 && (((uintptr_t)d->C & 15) || (d->R && !d->r_fp32))) return false;      // the fp32 residual stream: R and C both fp32
    if ((d->N & 7) || (d->ldc & 3) || (d->ld_norm & 3) || (((uintptr_t)d->C | (uintptr_t)d->norm_out | (uintptr_t)d->norm_w) & 7)) return false;
    if (crab_rowfin_workspace(d->M, d->N) > d->workspace_bytes || d->N > 128 * RF_CW) return false;
    if (d->route_RA && ((d->route_ldra & 7) || ((uintptr_t)d->route_RA & 15) || d->route_nl > 8 || d->route_nproj * (d->route_nl + d->route_r) > RF_TJ))
        return false;
    // K2 % 8: rowfin_apply loads lora_B in 16-byte chunks while ch * 8 < K2 - a K2 that is not a multiple of 8 would read past the row
    if (d->lora_RA && (!crab_rowfin_lora_ok(d->lora_nl, d->lora_r, d->N) || !d->B2 || (d->ldb2 & 7) || (d->K2 & 7) ||
                       ((uintptr_t)d->B2 & 15) || d->K2 < d->lora_nl * d->lora_r))
        return false;
    return true;
}

// S (and T when d->lora_RA) are in d->workspace, written by the GEMM launch in front (crab_gemm_skinny_launch, tune 9)
int crab_rowfin_launch(crab_ctx* ctx, hipStream_t s, const crab_gemm_desc* d) {
    const int nb = (d->N + RF_CW - 1) / RF_CW;
    char* w = (char*)d->workspace;
    float* S = (float*)w;
    w += rf_align((int64_t)d->M * d->N * 4);
    float* T = (float*)w;
    w += rf_align(RF_SKX * 16 * 16 * 4);
    float* ssq = (float*)w;
    w += rf_align((int64_t)nb * 16 * 4);
    float* tpart = (float*)w;
    w += rf_align((int64_t)nb * 16 * RF_TJ * 4);
    unsigned* counter = (unsigned*)w;
    RowfinApplyP a;
    a.S = S; a.T = d->lora_RA ? T : nullptr;
    a.B2 = (const bf16_t*)d->B2; a.ldb2 = d->ldb2; a.k2 = d->K2;
    a.X = (bf16_t*)d->C; a.ldx = d->ldc; a.ssq = ssq; a.counter = counter;
    a.M = d->M; a.N = d->N; a.nl = d->lora_nl; a.r = d->lora_r; a.scaling = d->lora_scaling;
    a.nw = d->norm_w; a.RA = (const bf16_t*)d->route_RA; a.ldra = d->route_ldra;
    a.used = d->route_RA ? d->route_nproj * (d->route_nl + d->route_r) : 0; a.tpart = tpart;
    // CRAB_ROWFIN_TAIL=1: the r03 pair (apply, then the route kernel with its arrival chain) for A/B runs; default: router partials in the first launch,
    // a second launch that waits for nothing
    static const int chain = []() { const char* e = getenv("CRAB_ROWFIN_TAIL"); return e && e[0] == '1' && e[1] == 0; }();
    if (!chain || d->norm_w_fp32) {                                    // (the r03 chain pair exists for bf16 norm weights only)
        const bool rp = d->route_RA != nullptr;
        if (d->norm_w_fp32) { if (rp) hipLaunchKernelGGL((rowfin_apply_kernel<true, true, true>), dim3(nb), dim3(256), 0, s, a); else hipLaunchKernelGGL((rowfin_apply_kernel<true, false, true>), dim3(nb), dim3(256), 0, s, a); }
        else if (d->c_fp32) { if (rp) hipLaunchKernelGGL((rowfin_apply_kernel<true, true>), dim3(nb), dim3(256), 0, s, a); else hipLaunchKernelGGL((rowfin_apply_kernel<true, false>), dim3(nb), dim3(256), 0, s, a); }
        else { if (rp) hipLaunchKernelGGL((rowfin_apply_kernel<false, true>), dim3(nb), dim3(256), 0, s, a); else hipLaunchKernelGGL((rowfin_apply_kernel<false, false>), dim3(nb), dim3(256), 0, s, a); }
        int rc0 = crab_check_launch(ctx, "rowfin_apply_kernel");
        if (rc0) return rc0;
        RowfinMixP x;
        x.X = (const bf16_t*)d->C; x.ldx = d->ldc; x.ssq = ssq; x.nb = nb;
        x.nw = d->norm_w; x.eps = d->norm_eps; x.H = (bf16_t*)d->norm_out; x.ldh = d->ld_norm;
        x.tpart = rp ? tpart : nullptr; x.U = (bf16_t*)d->route_U; x.ldu = d->route_ldu;
        x.nproj = d->route_nproj; x.nl = d->route_nl; x.r = d->route_r; x.ucols = d->route_ucols; x.scaling = d->route_scaling;
        x.M = d->M; x.N = d->N;
        if (d->norm_w_fp32) hipLaunchKernelGGL((rowfin_norm_mix_kernel<true, true>), dim3(nb), dim3(256), 0, s, x);
        else if (d->c_fp32) hipLaunchKernelGGL((rowfin_norm_mix_kernel<true, false>), dim3(nb), dim3(256), 0, s, x);
        else hipLaunchKernelGGL((rowfin_norm_mix_kernel<false, false>), dim3(nb), dim3(256), 0, s, x);
        return crab_check_launch(ctx, "rowfin_norm_mix_kernel");
    }
    if (d->c_fp32) hipLaunchKernelGGL((rowfin_apply_kernel<true, false>), dim3(nb), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((rowfin_apply_kernel<false, false>), dim3(nb), dim3(256), 0, s, a);
    int rc = crab_check_launch(ctx, "rowfin_apply_kernel");
    if (rc) return rc;
    RowfinRouteP r;
    r.X = (const bf16_t*)d->C; r.ldx = d->ldc; r.ssq = ssq; r.nb = nb;
    r.nw = (const bf16_t*)d->norm_w; r.eps = d->norm_eps; r.H = (bf16_t*)d->norm_out; r.ldh = d->ld_norm;
    r.RA = (const bf16_t*)d->route_RA; r.ldra = d->route_ldra; r.U = (bf16_t*)d->route_U; r.ldu = d->route_ldu;
    r.nproj = d->route_nproj; r.nl = d->route_nl; r.r = d->route_r; r.ucols = d->route_ucols; r.scaling = d->route_scaling;
    r.tpart = tpart; r.counter = counter; r.M = d->M; r.N = d->N;
    if (d->c_fp32) hipLaunchKernelGGL(rowfin_route_kernel<true>, dim3(nb), dim3(256), 0, s, r);
    else hipLaunchKernelGGL(rowfin_route_kernel<false>, dim3(nb), dim3(256), 0, s, r);
    return crab_check_launch(ctx, "rowfin_route_kernel");
}

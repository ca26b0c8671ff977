// HBM-bound helper kernels of the Crab forward path (norms, RoPE/KV scatter, embeddings, hyper-LoRA
// routing mix, SwiGLU, argmax, patch im2col, BEATs relative-position helpers).  All of them stream bf16
// with 16-byte (8 x bf16) vector accesses where the layout allows and do their arithmetic in fp32.
#include "common.h"
#include "crab_internal.h"
#include <math.h>

namespace {

// ------------------------------------------------------------------------------------------------ norms
// One wave per row (4 rows per 256-thread block); the row stays in registers between the two passes.
// MAXV = vectors of 8 per lane the row may need: 16 (D <= 8192) or 4 (D <= 2048: CLIP / BEATs / Q-Former widths - a quarter of the
// registers, so twice the resident waves for rows that are only 2-4 KB long)
template <bool RMS, int MAXV = 16, bool WF = false>            // WF: w / b are fp32 (crab_ln.fp32)
__global__ __launch_bounds__(256) void norm_kernel(const bf16_t* __restrict__ x, long ldx, const void* __restrict__ w,
                                                   const void* __restrict__ b, bf16_t* __restrict__ y, long ldy,
                                                   int M, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const bf16_t* xr = x + (long)row * ldx;
    bf16_t* yr = y + (long)row * ldy;
    u32x4 v[MAXV];
    const int nvec = D >> 3;                       // D % 8 == 0 (checked on the host)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int vi = lane + i * 64;
        if (vi < nvec) {
            v[i] = *reinterpret_cast<const u32x4*>(xr + vi * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = lo_bf(v[i][j]), c = hi_bf(v[i][j]);
                s1 += a + c;
                s2 += a * a + c * c;
            }
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(s2 / (float)D + eps);
    } else {
        mean = s1 / (float)D;
        // second pass over registers for the centred variance (matches torch's numerics better than E[x^2]-m^2)
        float sv = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int vi = lane + i * 64;
            if (vi < nvec) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float a = lo_bf(v[i][j]) - mean, c = hi_bf(v[i][j]) - mean;
                    sv += a * a + c * c;
                }
            }
        }
        sv = wave_sum(sv);
        rstd = rsqrtf(sv / (float)D + eps);
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int vi = lane + i * 64;
        if (vi < nvec) {
            float wv[8], bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            ld_par8<WF>(w, (long)vi * 8, wv);
            if (!RMS && b) ld_par8<WF>(b, (long)vi * 8, bv);
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = (lo_bf(v[i][j]) - mean) * rstd, c = (hi_bf(v[i][j]) - mean) * rstd;
                if (RMS) {   // modeling_llama.py:116-117: weight * x_hat.to(input_dtype)
                    a = bf2f(f2bf(a)); c = bf2f(f2bf(c));
                }
                a = a * wv[2 * j] + bv[2 * j];
                c = c * wv[2 * j + 1] + bv[2 * j + 1];
                o[j] = pack_bf2(a, c);
            }
            *reinterpret_cast<u32x4*>(yr + vi * 8) = o;
        }
    }
}

// The same for an fp32 input row (the fp32 residual stream of the decoder / CLIP tower: x stays fp32 between the projection epilogues,
// only the normalised row that feeds the next MFMA GEMM is bf16).  No intermediate rounding of x_hat: the reference runs this norm in fp32
// (scripts/quick_start.sh:42-44 --bf16 False; modeling_llama.py:112-117 with an fp32 input is exact), so y = bf16(x_hat * w [+ b]).
template <bool RMS, int MAXV = 16, bool WF = false>
__global__ __launch_bounds__(256) void norm_f32in_kernel(const float* __restrict__ x, long ldx, const void* __restrict__ w,
                                                         const void* __restrict__ b, bf16_t* __restrict__ y, long ldy,
                                                         int M, int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (long)row * ldx;
    bf16_t* yr = y + (long)row * ldy;
    f32x4_t v[MAXV][2];
    const int nvec = D >> 3;                       // D % 8 == 0 (checked on the host)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int vi = lane + i * 64;
        if (vi < nvec) {
            v[i][0] = *reinterpret_cast<const f32x4_t*>(xr + vi * 8);
            v[i][1] = *reinterpret_cast<const f32x4_t*>(xr + vi * 8 + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a = v[i][j >> 2][j & 3];
                s1 += a;
                s2 += a * a;
            }
        }
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(s2 / (float)D + eps);
    } else {
        mean = s1 / (float)D;
        float sv = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            int vi = lane + i * 64;
            if (vi < nvec) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float a = v[i][j >> 2][j & 3] - mean;
                    sv += a * a;
                }
            }
        }
        sv = wave_sum(sv);
        rstd = rsqrtf(sv / (float)D + eps);
    }
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        int vi = lane + i * 64;
        if (vi < nvec) {
            float wv[8], bv[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            ld_par8<WF>(w, (long)vi * 8, wv);
            if (!RMS && b) ld_par8<WF>(b, (long)vi * 8, bv);
            u32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = (v[i][j >> 1][(2 * j) & 3] - mean) * rstd, c = (v[i][j >> 1][(2 * j + 1) & 3] - mean) * rstd;
                a = a * wv[2 * j] + bv[2 * j];
                c = c * wv[2 * j + 1] + bv[2 * j + 1];
                o[j] = pack_bf2(a, c);
            }
            *reinterpret_cast<u32x4*>(yr + vi * 8) = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------ embedding
__global__ void embedding_kernel(const int64_t* __restrict__ ids, const bf16_t* __restrict__ table, bf16_t* __restrict__ out,
                                 long ldo, int T, int D, int vocab) {
    const int t = blockIdx.x;
    long id = ids[t];
    if (id < 0) return;                                  // negative id: row is filled by someone else (modality splice)
    if (id >= vocab) id = vocab - 1;
    const bf16_t* src = table + id * (long)D;
    bf16_t* dst = out + (long)t * ldo;
    for (int i = threadIdx.x; i < (D >> 3); i += blockDim.x)
        *reinterpret_cast<u32x4*>(dst + i * 8) = *reinterpret_cast<const u32x4*>(src + i * 8);
}

// fp32 output rows (the decoder's fp32 residual stream starts here in the decode step)
__global__ void embedding_f32_kernel(const int64_t* __restrict__ ids, const bf16_t* __restrict__ table, float* __restrict__ out,
                                     long ldo, int T, int D, int vocab) {
    const int t = blockIdx.x;
    long id = ids[t];
    if (id < 0) return;
    if (id >= vocab) id = vocab - 1;
    const bf16_t* src = table + id * (long)D;
    float* dst = out + (long)t * ldo;
    for (int i = threadIdx.x; i < (D >> 3); i += blockDim.x) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(src + i * 8);
        *reinterpret_cast<f32x4_t*>(dst + i * 8) = f32x4_t{lo_bf(v[0]), hi_bf(v[0]), lo_bf(v[1]), hi_bf(v[1])};
        *reinterpret_cast<f32x4_t*>(dst + i * 8 + 4) = f32x4_t{lo_bf(v[2]), hi_bf(v[2]), lo_bf(v[3]), hi_bf(v[3])};
    }
}

// strided row casts between the bf16 activations and the fp32 residual stream (8 columns per thread)
__global__ void cast_rows_bf16_f32_kernel(const bf16_t* __restrict__ src, long lds_, float* __restrict__ dst, long ldd, int rows, int cols) {
    const int nv = cols >> 3;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)rows * nv) return;
    const int r = idx / nv, c = idx % nv;
    const u32x4 v = *reinterpret_cast<const u32x4*>(src + (long)r * lds_ + c * 8);
    float* d_ = dst + (long)r * ldd + c * 8;
    *reinterpret_cast<f32x4_t*>(d_) = f32x4_t{lo_bf(v[0]), hi_bf(v[0]), lo_bf(v[1]), hi_bf(v[1])};
    *reinterpret_cast<f32x4_t*>(d_ + 4) = f32x4_t{lo_bf(v[2]), hi_bf(v[2]), lo_bf(v[3]), hi_bf(v[3])};
}
__global__ void cast_rows_f32_bf16_kernel(const float* __restrict__ src, long lds_, bf16_t* __restrict__ dst, long ldd, int rows, int cols) {
    const int nv = cols >> 3;
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)rows * nv) return;
    const int r = idx / nv, c = idx % nv;
    const float* s_ = src + (long)r * lds_ + c * 8;
    const f32x4_t a = *reinterpret_cast<const f32x4_t*>(s_), b = *reinterpret_cast<const f32x4_t*>(s_ + 4);
    *reinterpret_cast<u32x4*>(dst + (long)r * ldd + c * 8) = u32x4{pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3])};
}

// ------------------------------------------------------------------------------------------------ RoPE
__global__ void rope_table_kernel(float* tab, int max_pos, int half, float theta, int d) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= max_pos * half) return;
    int pos = idx / half, i = idx % half;
    // inv_freq = 1 / theta^(2i/d) evaluated in fp32 like torch (modeling_llama.py:130-136), angle = pos*inv_freq
    float inv = 1.0f / powf(theta, (float)(2 * i) / (float)d);
    float ang = (float)pos * inv;
    tab[2 * idx] = cosf(ang);
    tab[2 * idx + 1] = sinf(ang);
}

// grid (T, H + 2*Hk); block d/2 threads... one thread handles the pair (i, i+d/2)
__global__ void qkv_rope_split_kernel(bf16_t* __restrict__ qkv, long ldqkv, const float* __restrict__ tab,
                                      bf16_t* __restrict__ kc, bf16_t* __restrict__ vc, bf16_t* __restrict__ vt, long vt_ld,
                                      int S, int H, int Hk, int d, int Tmax, int pos0, const int* __restrict__ pos_dev,
                                      const int* __restrict__ pos_ids, long ld_pos, const int* __restrict__ row_off) {
    const int t = blockIdx.x;            // token index b*S + s
    const int hh = blockIdx.y;           // 0..H-1 q heads, H..H+Hk-1 k heads, H+Hk.. v heads
    const int b = t / S, s = t % S;
    const int half = d >> 1;
    const int i = threadIdx.x;
    if (i >= half) return;
    if (!tab && (hh < H || (hh < H + Hk && !kc))) return;      // encoder use: only V^T is materialised
    const int pos = (pos_dev ? pos_dev[0] : 0) + pos0 + s;       // cache slot
    // rotary position: position_ids of forward(), or - ragged decode batch - the slot minus the sequence's first slot
    const int rp = pos_ids ? pos_ids[(long)b * ld_pos + s] : pos - (row_off ? row_off[b] : 0);
    bf16_t* src = qkv + (long)t * ldqkv + (long)hh * d;
    if (hh < H + Hk) {
        float x1 = bf2f(src[i]), x2 = bf2f(src[i + half]);
        float o1 = x1, o2 = x2;
        if (tab) {
            float c = tab[2 * ((long)rp * half + i)], sn = tab[2 * ((long)rp * half + i) + 1];
            // q*cos + rotate_half(q)*sin, each product rounded as in the bf16 reference? fp32 here, one rounding
            o1 = rope_lo(x1, x2, c, sn);
            o2 = rope_hi(x1, x2, c, sn);
        }
        if (hh < H) {
            src[i] = f2bf(o1);
            src[i + half] = f2bf(o2);
        } else if (kc) {
            const int hk = hh - H;
            bf16_t* dst = kc + (((long)b * Hk + hk) * Tmax + pos) * d;
            dst[i] = f2bf(o1);
            dst[i + half] = f2bf(o2);
        }
    } else {
        const int hk = hh - H - Hk;
        bf16_t v1 = src[i], v2 = src[i + half];
        if (vc) {
            bf16_t* dst = vc + (((long)b * Hk + hk) * Tmax + pos) * d;
            dst[i] = v1;
            dst[i + half] = v2;
        }
        if (vt) {
            bf16_t* dt = vt + ((long)b * Hk + hk) * d * vt_ld;
            dt[(long)i * vt_ld + s] = v1;
            dt[(long)(i + half) * vt_ld + s] = v2;
        }
    }
}


// Tiled variant for S >= 16 (prefill / encoders): one block per (64 tokens, head).  16-byte vector loads/stores for the
// rotation and the cache scatter; V^T goes through an LDS transpose so that vt rows are written as 16-byte pieces
// (the element-wise kernel above writes V^T with 2-byte stores at a stride of vt_ld elements).
template <int D>
__global__ __launch_bounds__(256) void qkv_rope_split_tile_kernel(bf16_t* __restrict__ qkv, long ldqkv, const float* __restrict__ tab,
                                                                  bf16_t* __restrict__ kc, bf16_t* __restrict__ vc, bf16_t* __restrict__ vt,
                                                                  long vt_ld, int S, int H, int Hk, int Tmax, int pos0,
                                                                  const int* __restrict__ pos_ids, long ld_pos) {
    constexpr int HALF = D / 2, CH = D / 8;                     // 16-byte chunks per head row
    __shared__ bf16_t tile[64][D + 2];                          // +2: odd word stride for the transposed reads
    const int tid = threadIdx.x;
    const int s0 = blockIdx.x * 64, hh = blockIdx.y, b = blockIdx.z;
    const long tok0 = (long)b * S + s0;
    if (hh < H + Hk) {
        if (!tab && (hh < H || !kc)) return;                     // encoder use: nothing to do for q / k heads
        const int hk = hh - H;
        for (int idx = tid; idx < 64 * (CH / 2); idx += 256) {   // (token, chunk pair i0..i0+7 | HALF+i0..)
            const int tk = idx / (CH / 2), c = idx % (CH / 2);
            const int s = s0 + tk;
            if (s >= S) continue;
            bf16_t* src = qkv + (tok0 + tk) * ldqkv + (long)hh * D + c * 8;
            u32x4 lo = *reinterpret_cast<const u32x4*>(src);
            u32x4 hi = *reinterpret_cast<const u32x4*>(src + HALF);
            u32x4 olo = lo, ohi = hi;
            const int pos = pos0 + s;                                       // cache slot
            if (tab) {
                const int rp = pos_ids ? pos_ids[(long)b * ld_pos + s] : pos;   // rotary position
                const float* cs = tab + 2 * ((long)rp * HALF + c * 8);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float x1a = lo_bf(lo[e]), x1b = hi_bf(lo[e]), x2a = lo_bf(hi[e]), x2b = hi_bf(hi[e]);
                    float ca = cs[4 * e], sa = cs[4 * e + 1], cb = cs[4 * e + 2], sb = cs[4 * e + 3];
                    olo[e] = pack_bf2(rope_lo(x1a, x2a, ca, sa), rope_lo(x1b, x2b, cb, sb));
                    ohi[e] = pack_bf2(rope_hi(x1a, x2a, ca, sa), rope_hi(x1b, x2b, cb, sb));
                }
            }
            bf16_t* dst = hh < H ? src : kc + (((long)b * Hk + hk) * Tmax + pos) * D + c * 8;
            *reinterpret_cast<u32x4*>(dst) = olo;
            *reinterpret_cast<u32x4*>(dst + HALF) = ohi;
        }
        return;
    }
    const int hk = hh - H - Hk;
    for (int idx = tid; idx < 64 * CH; idx += 256) {
        const int tk = idx / CH, c = idx % CH;
        const int s = s0 + tk;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (s < S) {
            v = *reinterpret_cast<const u32x4*>(qkv + (tok0 + tk) * ldqkv + (long)hh * D + c * 8);
            if (vc) *reinterpret_cast<u32x4*>(vc + (((long)b * Hk + hk) * Tmax + pos0 + s) * D + c * 8) = v;
        }
        uint32_t* t32 = reinterpret_cast<uint32_t*>(&tile[tk][c * 8]);
        t32[0] = v[0]; t32[1] = v[1]; t32[2] = v[2]; t32[3] = v[3];
    }
    if (!vt) return;
    __syncthreads();
    bf16_t* vtb = vt + ((long)b * Hk + hk) * D * vt_ld;
    for (int idx = tid; idx < D * 8; idx += 256) {               // (d, group of 8 tokens)
        const int dd = idx >> 3, g = idx & 7;
        const int s = s0 + g * 8;
        if (s >= vt_ld) continue;
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (uint32_t)tile[g * 8 + 2 * e][dd] | ((uint32_t)tile[g * 8 + 2 * e + 1][dd] << 16);
        if (s + 8 <= vt_ld) *reinterpret_cast<u32x4*>(vtb + (long)dd * vt_ld + s) = o;
    }
}

// ------------------------------------------------------------------------------------------------ hyper-LoRA mix
// one thread per (row, projection): softmax over nl route logits, then scaling * p_i * h_j
template <typename TT>
__global__ void hyperlora_mix_kernel(const TT* __restrict__ T, long ldt, bf16_t* __restrict__ U, long ldu, int M,
                                     int nproj, int nl, int r, int ucols, float scaling) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    const int used = nproj * nl * r;
    const int per_row = nproj + 1;                 // last slot zero-fills the padding columns
    if (idx >= M * per_row) return;
    int m = idx / per_row, p = idx % per_row;
    bf16_t* u = U + (long)m * ldu;
    if (p == nproj) {
        for (int c = used; c < ucols; ++c) u[c] = 0;
        return;
    }
    const TT* t = T + (long)m * ldt + p * (nl + r);
    float lg[8], mx = -INFINITY;
    for (int i = 0; i < nl; ++i) {
        lg[i] = sizeof(TT) == 4 ? (float)((const float*)t)[i] : bf2f(((const bf16_t*)t)[i]);
        mx = fmaxf(mx, lg[i]);
    }
    float sum = 0.f;
    for (int i = 0; i < nl; ++i) { lg[i] = expf(lg[i] - mx); sum += lg[i]; }
    float inv = 1.0f / sum;
    for (int i = 0; i < nl; ++i) {
        float pi = lg[i] * inv;
        for (int j = 0; j < r; ++j) {
            float h = sizeof(TT) == 4 ? (float)((const float*)t)[nl + j] : bf2f(((const bf16_t*)t)[nl + j]);
            u[p * nl * r + i * r + j] = f2bf(scaling * pi * h);
        }
    }
}

// ------------------------------------------------------------------------------------------------ SwiGLU
__global__ void swiglu_kernel(const bf16_t* __restrict__ gu, long ldgu, bf16_t* __restrict__ y, long ldy, int M, int I) {
    const int nv = I >> 3;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)M * nv) return;
    int m = idx / nv, c = idx % nv;
    u32x4 g = *reinterpret_cast<const u32x4*>(gu + (long)m * ldgu + c * 8);
    u32x4 u = *reinterpret_cast<const u32x4*>(gu + (long)m * ldgu + I + c * 8);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float g0 = lo_bf(g[j]), g1 = hi_bf(g[j]);
        o[j] = pack_bf2(g0 / (1.f + __expf(-g0)) * lo_bf(u[j]), g1 / (1.f + __expf(-g1)) * hi_bf(u[j]));
    }
    *reinterpret_cast<u32x4*>(y + (long)m * ldy + c * 8) = o;
}

// ------------------------------------------------------------------------------------------------ argmax
__global__ __launch_bounds__(1024) void argmax_kernel(const float* __restrict__ logits, long ldl, int64_t* __restrict__ ids,
                                                      int V, int suppress) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const int b = blockIdx.x;
    const float* row = logits + (long)b * ldl;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
        float v = (i == suppress) ? -INFINITY : row[i];
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sv[wave] = best; si[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int nw = blockDim.x >> 6;
        for (int w = 1; w < nw; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        ids[b] = (bi == 0x7fffffff) ? 0 : bi;
    }
}

// ------------------------------------------------------------------------------------------------ im2col (stride == kernel)
// One block per (image, patch row): the C*P image rows that feed the gw patches of that patch row are read as whole rows
// (consecutive lanes -> consecutive pixels: coalesced), staged in LDS as bf16, and each patch's K = C*P*P row (+ zero padding up
// to ldo) is written out contiguously.  out[(n, gy, gx), c*P*P + ky*P + kx] = in[n, c, gy*P + ky, gx*P + kx].
template <typename TI>
__global__ __launch_bounds__(256) void im2col_patch_kernel(const TI* __restrict__ in, bf16_t* __restrict__ out, long ldo, int N, int C, int Hh,
                                                           int Ww, int P, int gh, int gw) {
    extern __shared__ bf16_t smem[];
    const int n = blockIdx.x / gh, gy = blockIdx.x % gh;
    const int wu = gw * P;                          // pixels of a row that belong to whole patches
    const int nrows = C * P;
    const int Kp = C * P * P;
    bf16_t* rows = smem;                            // [C*P][wu]
    int* koff = reinterpret_cast<int*>(smem + ((nrows * wu + 1) & ~1));          // k -> offset of (c, ky, kx) inside `rows` (gx = 0)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = threadIdx.x; k < Kp; k += blockDim.x) {
        const int c = k / (P * P), rr = k - c * P * P, ky = rr / P;
        koff[k] = (c * P + ky) * wu + (rr - ky * P);
    }
    // one wave per image row, 4 pixels per lane (8- or 16-byte loads) when the row allows it
    const bool v4 = (wu & 3) == 0 && (Ww & 3) == 0 && ((uintptr_t)in & 15) == 0;
    for (int r = wave; r < nrows; r += 4) {
        const int c = r / P, ky = r - c * P;
        const TI* src = in + (((long)n * C + c) * Hh + gy * P + ky) * Ww;
        bf16_t* dst = rows + r * wu;
        if (v4) {
            for (int x = lane * 4; x < wu; x += 256) {
                bf16_t h[4];
                if (sizeof(TI) == 4) {
                    const float4 f = *reinterpret_cast<const float4*>(src + x);
                    h[0] = f2bf(f.x); h[1] = f2bf(f.y); h[2] = f2bf(f.z); h[3] = f2bf(f.w);
                } else {
                    *reinterpret_cast<uint2*>(h) = *reinterpret_cast<const uint2*>(src + x);
                }
                *reinterpret_cast<uint2*>(dst + x) = *reinterpret_cast<const uint2*>(h);
            }
        } else {
            for (int x = lane; x < wu; x += 64) {
                const TI px = src[x];
                dst[x] = sizeof(TI) == 4 ? f2bf(*reinterpret_cast<const float*>(&px)) : *reinterpret_cast<const bf16_t*>(&px);
            }
        }
    }
    __syncthreads();
    bf16_t* o = out + ((long)(n * gh + gy) * gw) * ldo;
    if ((ldo & 7) == 0 && ((uintptr_t)out & 15) == 0) {            // 16-byte stores: 8 gathered elements per lane
        const int cpr = (int)(ldo >> 3);
        for (int i = threadIdx.x; i < gw * cpr; i += blockDim.x) {
            const int gx = i / cpr, k0 = (i - gx * cpr) * 8;
            union { u32x4 r; bf16_t h[8]; } v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v.h[j] = k0 + j < Kp ? rows[koff[k0 + j] + gx * P] : (bf16_t)0;
            *reinterpret_cast<u32x4*>(o + (long)gx * ldo + k0) = v.r;
        }
    } else {
        for (long i = threadIdx.x; i < (long)gw * ldo; i += blockDim.x) {
            const int gx = (int)(i / ldo), k = (int)(i - (long)gx * ldo);
            o[i] = k < Kp ? rows[koff[k] + gx * P] : (bf16_t)0;
        }
    }
}

// CLIP: assemble [cls | patches] + position embedding, then pre_layrnorm; one wave per token row
template <bool WF>
__global__ __launch_bounds__(256) void clip_embed_ln_kernel(const bf16_t* __restrict__ patch, const bf16_t* __restrict__ cls,
                                                            const bf16_t* __restrict__ pos, const void* __restrict__ lnw,
                                                            const void* __restrict__ lnb, bf16_t* __restrict__ y, int N, int P,
                                                            int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)N * (P + 1)) return;
    const int n = row / (P + 1), tk = row % (P + 1);
    const bf16_t* src = tk == 0 ? cls : patch + ((long)n * P + (tk - 1)) * D;
    const bf16_t* pe = pos + (long)tk * D;
    constexpr int MAXE = 32;                        // D <= 2048
    float v[MAXE];
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        int e = lane + i * 64;
        if (e < D) {
            // the sum feeds the LayerNorm unrounded (r05: the fp32 reference does not round it; r01-r04 rounded it to bf16 like a bf16 model would)
            v[i] = bf2f(src[e]) + bf2f(pe[e]);
            s1 += v[i];
        }
    }
    float mean = wave_sum(s1) / (float)D;
    float sv = 0.f;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        int e = lane + i * 64;
        if (e < D) { float a = v[i] - mean; sv += a * a; }
    }
    float rstd = rsqrtf(wave_sum(sv) / (float)D + eps);
    bf16_t* yr = y + row * D;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        int e = lane + i * 64;
        if (e < D) {
            const float w_ = WF ? reinterpret_cast<const float*>(lnw)[e] : bf2f(reinterpret_cast<const bf16_t*>(lnw)[e]);
            const float b_ = WF ? reinterpret_cast<const float*>(lnb)[e] : bf2f(reinterpret_cast<const bf16_t*>(lnb)[e]);
            yr[e] = f2bf((v[i] - mean) * rstd * w_ + b_);
        }
    }
}

// D % 8 == 0: each lane owns 16-byte chunks (8 consecutive elements), so every global access of the row is a full-width
// coalesced dwordx4 (the scalar kernel above moves 2 bytes per lane and reaches 1.6 TB/s; this one is HBM-copy bound)
template <bool WF>
__global__ __launch_bounds__(256) void clip_embed_ln_vec_kernel(const bf16_t* __restrict__ patch, const bf16_t* __restrict__ cls,
                                                                const bf16_t* __restrict__ pos, const void* __restrict__ lnw,
                                                                const void* __restrict__ lnb, bf16_t* __restrict__ y, int N, int P,
                                                                int D, float eps) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)N * (P + 1)) return;
    const int n = row / (P + 1), tk = row % (P + 1);
    const bf16_t* src = tk == 0 ? cls : patch + ((long)n * P + (tk - 1)) * D;
    const bf16_t* pe = pos + (long)tk * D;
    constexpr int MAXC = 4;                         // D <= 2048: up to 4 chunks of 8 per lane
    union V8 { u32x4 r; bf16_t h[8]; };
    float v[MAXC][8];
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int e = (lane + i * 64) * 8;
        if (e < D) {
            V8 a, b;
            a.r = *reinterpret_cast<const u32x4*>(src + e);
            b.r = *reinterpret_cast<const u32x4*>(pe + e);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[i][j] = bf2f(a.h[j]) + bf2f(b.h[j]);                  // unrounded (see clip_embed_ln_kernel)
                s1 += v[i][j];
            }
        }
    }
    const float mean = wave_sum(s1) / (float)D;
    float sv = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        if ((lane + i * 64) * 8 < D) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float a = v[i][j] - mean; sv += a * a; }
        }
    }
    const float rstd = rsqrtf(wave_sum(sv) / (float)D + eps);
    bf16_t* yr = y + row * D;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int e = (lane + i * 64) * 8;
        if (e < D) {
            V8 o;
            float w8[8], b8[8];
            ld_par8<WF>(lnw, (long)e, w8);
            ld_par8<WF>(lnb, (long)e, b8);
#pragma unroll
            for (int j = 0; j < 8; ++j) o.h[j] = f2bf((v[i][j] - mean) * rstd * w8[j] + b8[j]);
            *reinterpret_cast<u32x4*>(yr + e) = o.r;
        }
    }
}

// ------------------------------------------------------------------------------------------------ BEATs helpers
// x[B,n,E] -> xp[G][B][n+Kc-1][E/G], Kc/2 zero rows in front (Conv1d padding=Kc/2) and Kc/2-1 behind
// (the even-kernel conv emits n+1 steps of which SamePad drops the last, backbone.py:33-46, modules.py:29-40)
__global__ void beats_posconv_pad_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ xp, int B, int n, int E, int G, int Kc) {
    const int cg = E / G;
    const int np = n + Kc - 1;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long total = (long)G * B * np * cg;
    if (idx >= total) return;
    int c = idx % cg;
    long r = idx / cg;
    int tp = r % np; r /= np;
    int b = r % B;
    int g = r / B;
    int t = tp - Kc / 2;
    bf16_t v = 0;
    if (t >= 0 && t < n) v = x[((long)b * n + t) * E + g * cg + c];
    xp[idx] = v;
}

// bias[h,i,j] = table[bucket(j - i)][h], bidirectional T5 buckets (backbone.py:392-430)
__global__ void beats_relpos_bias_kernel(const bf16_t* __restrict__ table, float* __restrict__ bias, int n, int H, int num_buckets,
                                         int max_distance) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    int i = idx / n, j = idx % n;
    int rel = j - i;
    int nb = num_buckets / 2;
    int bucket = rel > 0 ? nb : 0;
    int a = rel < 0 ? -rel : rel;
    int max_exact = nb / 2;
    if (a < max_exact) {
        bucket += a;
    } else {
        // torch: (log(a / max_exact) / log(max_distance / max_exact) * (nb - max_exact)).to(long), fp32 arithmetic
        float v = logf((float)a / (float)max_exact) / (float)log((double)max_distance / (double)max_exact) * (float)(nb - max_exact);
        int large = max_exact + (int)v;
        bucket += large < nb - 1 ? large : nb - 1;
    }
    for (int h = 0; h < H; ++h) bias[((long)h * n + i) * n + j] = bf2f(table[(long)bucket * H + h]);
}

// gate[b,h,i] from the un-scaled q projection (backbone.py:650-662): grep_linear d->8, view(2,4).sum, sigmoid
__global__ __launch_bounds__(128) void beats_gru_gate_kernel(const bf16_t* __restrict__ q, long ldq, const bf16_t* __restrict__ gw,
                                                             const bf16_t* __restrict__ gb, const bf16_t* __restrict__ grep_a,
                                                             float* __restrict__ gate, int B, int n, int H, int d) {
    // grep_linear weight [8, d] staged once per block in LDS as fp32; each thread reads its q row with 16-byte loads (d % 8 == 0,
    // rows 16-byte aligned: checked by the launcher; otherwise element loads)
    __shared__ float wsm[8 * 256];
    for (int i = threadIdx.x; i < 8 * d; i += blockDim.x) wsm[i] = bf2f(gw[i]);
    __syncthreads();
    int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * H * n) return;
    int i = idx % n, h = (idx / n) % H, b = idx / (n * H);
    const bf16_t* qr = q + ((long)b * n + i) * ldq + h * d;
    float o[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = bf2f(gb[k]);
    const bool vec = ((d & 7) == 0) && ((ldq & 7) == 0) && ((reinterpret_cast<uintptr_t>(q) & 15) == 0);
    for (int e0 = 0; e0 < d; e0 += 8) {
        float x[8];
        if (vec) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(qr + e0);
#pragma unroll
            for (int j = 0; j < 4; ++j) { x[2 * j] = lo_bf(v[j]); x[2 * j + 1] = hi_bf(v[j]); }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = e0 + j < d ? bf2f(qr[e0 + j]) : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float s = o[k];
#pragma unroll
            for (int j = 0; j < 8; ++j) s += x[j] * (e0 + j < d ? wsm[k * d + e0 + j] : 0.f);   // same e-ascending order as before
            o[k] = s;
        }
    }
    float sa = o[0] + o[1] + o[2] + o[3], sb = o[4] + o[5] + o[6] + o[7];
    float ga = 1.f / (1.f + expf(-sa)), gb_ = 1.f / (1.f + expf(-sb));
    gate[idx] = ga * (gb_ * bf2f(grep_a[h]) - 1.0f) + 2.0f;
}

__global__ void copy_rows_kernel(const bf16_t* __restrict__ src, long lds_, bf16_t* __restrict__ dst, long ldd, int rows, int cols) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if ((cols & 7) == 0 && (lds_ & 7) == 0 && (ldd & 7) == 0) {
        int nv = cols >> 3;
        if (idx >= (long)rows * nv) return;
        int r = idx / nv, c = idx % nv;
        *reinterpret_cast<u32x4*>(dst + (long)r * ldd + c * 8) = *reinterpret_cast<const u32x4*>(src + (long)r * lds_ + c * 8);
    } else {
        if (idx >= (long)rows * cols) return;
        int r = idx / cols, c = idx % cols;
        dst[(long)r * ldd + c] = src[(long)r * lds_ + c];
    }
}

__global__ void copy_rows_batched_kernel(const bf16_t* __restrict__ src, long lds_, long sbs, bf16_t* __restrict__ dst, long ldd, long dbs,
                                         int rows, int cols) {
    const int z = blockIdx.y;
    const bf16_t* s_ = src + z * sbs;
    bf16_t* d_ = dst + z * dbs;
    int nv = cols >> 3;
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long)rows * nv) return;
    int r = idx / nv, c = idx % nv;
    *reinterpret_cast<u32x4*>(d_ + (long)r * ldd + c * 8) = *reinterpret_cast<const u32x4*>(s_ + (long)r * lds_ + c * 8);
}

// Greedy step (HF GenerationMixin greedy search as driven by unified_llama.py:262-267): argmax over fp32
// logits (first max wins), eos suppressed while step < min_new, finished rows emit pad.  One block per row.
__global__ __launch_bounds__(1024) void greedy_select_kernel(const float* __restrict__ logits, long ldl, int V, int64_t* __restrict__ cur_ids,
                                                             int64_t* __restrict__ out_ids, long ld_out, const int* __restrict__ step_dev,
                                                             int* __restrict__ finished, int eos_id, int pad_id, int min_new) {
    __shared__ float sv[16];
    __shared__ int si[16];
    const int b = blockIdx.x;
    const int step = step_dev[0];
    const int suppress = (eos_id >= 0 && step < min_new) ? eos_id : -1;
    const float* row = logits + (long)b * ldl;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < V; i += blockDim.x) {
        float v = (i == suppress) ? -INFINITY : row[i];
        if (v > best || (v == best && i < bi)) { best = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { sv[wave] = best; si[wave] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        int nw = blockDim.x >> 6;
        for (int w = 1; w < nw; ++w)
            if (sv[w] > best || (sv[w] == best && si[w] < bi)) { best = sv[w]; bi = si[w]; }
        if (bi == 0x7fffffff) bi = 0;
        int tok = finished[b] ? pad_id : bi;
        if (eos_id >= 0 && tok == eos_id) finished[b] = 1;
        cur_ids[b] = tok;
        out_ids[(long)b * ld_out + step] = tok;
    }
}

__global__ void advance_kernel(int* pos_dev, int* step_dev) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { pos_dev[0] += 1; step_dev[0] += 1; }
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = f2bf(src[i]);
}

}  // namespace

#define S_(x) ((hipStream_t)(x))
static inline unsigned cdiv(long a, long b) { return (unsigned)((a + b - 1) / b); }

extern "C" {

// The four public norm entry points and their parameter-storage variant share one launcher: x bf16 | fp32, RMS | LayerNorm, w / b bf16 | fp32
static int norm_launch(crab_ctx* ctx, void* stream, const char* what, const void* x, int x_fp32, int64_t ldx, const void* w, const void* b, int w_fp32,
                       void* y, int64_t ldy, int M, int D, float eps, bool rms) {
    if (!ctx) return CRAB_E_INVALID;
    char msg[160];
    if (!x || !w || !y || M <= 0) { snprintf(msg, sizeof(msg), "%s: bad argument", what); return crab_fail(ctx, CRAB_E_INVALID, msg); }
    if ((D & 7) || D > 8192 || (ldx & (x_fp32 ? 3 : 7)) || (ldy & 7) || (x_fp32 && ((uintptr_t)x & 15)) ||
        (w_fp32 && (((uintptr_t)w | (uintptr_t)b) & 15))) {
        snprintf(msg, sizeof(msg), "%s: D must be a multiple of 8 and <= 8192%s", what, x_fp32 ? ", 16-byte aligned rows" : "");
        return crab_fail(ctx, CRAB_E_INVALID, msg);
    }
    const dim3 grid(cdiv(M, 4)), block(256);
    hipStream_t s = S_(stream);
#define NL(K_, RMS_, MV_, WF_, XT_) hipLaunchKernelGGL((K_<RMS_, MV_, WF_>), grid, block, 0, s, (const XT_*)x, (long)ldx, w, b, (bf16_t*)y, (long)ldy, M, D, eps)
#define NL_X(RMS_, MV_, WF_) do { if (x_fp32) NL(norm_f32in_kernel, RMS_, MV_, WF_, float); else NL(norm_kernel, RMS_, MV_, WF_, bf16_t); } while (0)
#define NL_W(RMS_, MV_) do { if (w_fp32) NL_X(RMS_, MV_, true); else NL_X(RMS_, MV_, false); } while (0)
    if (rms) { if (D <= 2048) NL_W(true, 4); else NL_W(true, 16); }
    else { if (D <= 2048) NL_W(false, 4); else NL_W(false, 16); }
#undef NL_W
#undef NL_X
#undef NL
    return crab_check_launch(ctx, what);
}

int crab_rmsnorm(crab_ctx* ctx, void* stream, const void* x, int64_t ldx, const void* w, void* y, int64_t ldy, int M, int D, float eps) {
    return norm_launch(ctx, stream, "rmsnorm", x, 0, ldx, w, nullptr, 0, y, ldy, M, D, eps, true);
}

int crab_layernorm(crab_ctx* ctx, void* stream, const void* x, int64_t ldx, const void* w, const void* b, void* y, int64_t ldy, int M,
                   int D, float eps) {
    return norm_launch(ctx, stream, "layernorm", x, 0, ldx, w, b, 0, y, ldy, M, D, eps, false);
}

int crab_rmsnorm_f32(crab_ctx* ctx, void* stream, const float* x, int64_t ldx, const void* w, void* y, int64_t ldy, int M, int D, float eps) {
    return norm_launch(ctx, stream, "rmsnorm_f32", x, 1, ldx, w, nullptr, 0, y, ldy, M, D, eps, true);
}

int crab_layernorm_f32(crab_ctx* ctx, void* stream, const float* x, int64_t ldx, const void* w, const void* b, void* y, int64_t ldy, int M,
                       int D, float eps) {
    return norm_launch(ctx, stream, "layernorm_f32", x, 1, ldx, w, b, 0, y, ldy, M, D, eps, false);
}

int crab_rmsnorm_p(crab_ctx* ctx, void* stream, const void* x, int x_fp32, int64_t ldx, const void* w, int w_fp32, void* y, int64_t ldy, int M,
                   int D, float eps) {
    return norm_launch(ctx, stream, "rmsnorm_p", x, x_fp32 ? 1 : 0, ldx, w, nullptr, w_fp32 ? 1 : 0, y, ldy, M, D, eps, true);
}

int crab_layernorm_p(crab_ctx* ctx, void* stream, const void* x, int x_fp32, int64_t ldx, const void* w, const void* b, int w_fp32, void* y,
                     int64_t ldy, int M, int D, float eps) {
    return norm_launch(ctx, stream, "layernorm_p", x, x_fp32 ? 1 : 0, ldx, w, b, w_fp32 ? 1 : 0, y, ldy, M, D, eps, false);
}

int crab_embedding_f32(crab_ctx* ctx, void* stream, const int64_t* ids, const void* table, float* out, int64_t ldo, int T, int D, int vocab) {
    if (!ctx) return CRAB_E_INVALID;
    if (!ids || !table || !out || T <= 0 || (D & 7) || (ldo & 3) || ((uintptr_t)out & 15)) return crab_fail(ctx, CRAB_E_INVALID, "embedding_f32: bad argument");
    hipLaunchKernelGGL(embedding_f32_kernel, dim3(T), dim3(128), 0, S_(stream), ids, (const bf16_t*)table, out, (long)ldo, T, D, vocab);
    return crab_check_launch(ctx, "embedding_f32");
}

int crab_cast_rows_bf16_f32(crab_ctx* ctx, void* stream, const void* src, int64_t lds_, float* dst, int64_t ldd, int rows, int cols) {
    if (!ctx) return CRAB_E_INVALID;
    if (!src || !dst || rows <= 0 || cols <= 0 || (cols & 7) || (lds_ & 7) || (ldd & 3) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15))
        return crab_fail(ctx, CRAB_E_INVALID, "cast_rows_bf16_f32: cols / lds multiples of 8, ldd multiple of 4, 16-byte aligned");
    hipLaunchKernelGGL(cast_rows_bf16_f32_kernel, dim3(cdiv((long)rows * (cols >> 3), 256)), dim3(256), 0, S_(stream), (const bf16_t*)src, (long)lds_,
                       dst, (long)ldd, rows, cols);
    return crab_check_launch(ctx, "cast_rows_bf16_f32");
}

int crab_cast_rows_f32_bf16(crab_ctx* ctx, void* stream, const float* src, int64_t lds_, void* dst, int64_t ldd, int rows, int cols) {
    if (!ctx) return CRAB_E_INVALID;
    if (!src || !dst || rows <= 0 || cols <= 0 || (cols & 7) || (lds_ & 3) || (ldd & 7) || ((uintptr_t)src & 15) || ((uintptr_t)dst & 15))
        return crab_fail(ctx, CRAB_E_INVALID, "cast_rows_f32_bf16: cols / ldd multiples of 8, lds multiple of 4, 16-byte aligned");
    hipLaunchKernelGGL(cast_rows_f32_bf16_kernel, dim3(cdiv((long)rows * (cols >> 3), 256)), dim3(256), 0, S_(stream), src, (long)lds_,
                       (bf16_t*)dst, (long)ldd, rows, cols);
    return crab_check_launch(ctx, "cast_rows_f32_bf16");
}

int crab_embedding(crab_ctx* ctx, void* stream, const int64_t* ids, const void* table, void* out, int64_t ldo, int T, int D, int vocab) {
    if (!ctx) return CRAB_E_INVALID;
    if (!ids || !table || !out || T <= 0 || (D & 7) || (ldo & 7)) return crab_fail(ctx, CRAB_E_INVALID, "embedding: bad argument");
    hipLaunchKernelGGL(embedding_kernel, dim3(T), dim3(128), 0, S_(stream), ids, (const bf16_t*)table, (bf16_t*)out, (long)ldo, T, D, vocab);
    return crab_check_launch(ctx, "embedding");
}

int crab_rope_table(crab_ctx* ctx, void* stream, float* tab, int max_pos, int head_dim, float theta) {
    if (!ctx) return CRAB_E_INVALID;
    if (!tab || max_pos <= 0 || (head_dim & 1)) return crab_fail(ctx, CRAB_E_INVALID, "rope_table: bad argument");
    int half = head_dim / 2;
    hipLaunchKernelGGL(rope_table_kernel, dim3(cdiv((long)max_pos * half, 256)), dim3(256), 0, S_(stream), tab, max_pos, half, theta, head_dim);
    return crab_check_launch(ctx, "rope_table");
}

int crab_qkv_rope_split(crab_ctx* ctx, void* stream, void* qkv, int64_t ldqkv, const float* rope_tab, void* k_cache, void* v_cache,
                        void* vt, int64_t vt_ld, int B, int S, int H, int Hk, int d, int Tmax, int pos0, const int32_t* pos_dev) {
    return crab_qkv_rope_split_ids(ctx, stream, qkv, ldqkv, rope_tab, k_cache, v_cache, vt, vt_ld, B, S, H, Hk, d, Tmax, pos0, pos_dev, nullptr, 0);
}

static int qkv_rope_split_impl(crab_ctx* ctx, void* stream, void* qkv, int64_t ldqkv, const float* rope_tab, void* k_cache, void* v_cache,
                               void* vt, int64_t vt_ld, int B, int S, int H, int Hk, int d, int Tmax, int pos0, const int32_t* pos_dev,
                               const int32_t* pos_ids, int64_t ld_pos, const int32_t* row_off);

int crab_qkv_rope_split_ids(crab_ctx* ctx, void* stream, void* qkv, int64_t ldqkv, const float* rope_tab, void* k_cache, void* v_cache,
                            void* vt, int64_t vt_ld, int B, int S, int H, int Hk, int d, int Tmax, int pos0, const int32_t* pos_dev,
                            const int32_t* pos_ids, int64_t ld_pos) {
    return qkv_rope_split_impl(ctx, stream, qkv, ldqkv, rope_tab, k_cache, v_cache, vt, vt_ld, B, S, H, Hk, d, Tmax, pos0, pos_dev, pos_ids, ld_pos, nullptr);
}

int crab_qkv_rope_split_ragged(crab_ctx* ctx, void* stream, void* qkv, int64_t ldqkv, const float* rope_tab, void* k_cache, void* v_cache,
                               void* vt, int64_t vt_ld, int B, int S, int H, int Hk, int d, int Tmax, int pos0, const int32_t* pos_dev,
                               const int32_t* row_off) {
    return qkv_rope_split_impl(ctx, stream, qkv, ldqkv, rope_tab, k_cache, v_cache, vt, vt_ld, B, S, H, Hk, d, Tmax, pos0, pos_dev, nullptr, 0, row_off);
}

static int qkv_rope_split_impl(crab_ctx* ctx, void* stream, void* qkv, int64_t ldqkv, const float* rope_tab, void* k_cache, void* v_cache,
                               void* vt, int64_t vt_ld, int B, int S, int H, int Hk, int d, int Tmax, int pos0, const int32_t* pos_dev,
                               const int32_t* pos_ids, int64_t ld_pos, const int32_t* row_off) {
    if (!ctx) return CRAB_E_INVALID;
    if (pos_ids && ld_pos < S) return crab_fail(ctx, CRAB_E_INVALID, "qkv_rope_split: ld_pos < S");
    if (!qkv || B <= 0 || S <= 0 || d > 2048 || (d & 1)) return crab_fail(ctx, CRAB_E_INVALID, "qkv_rope_split: bad argument");
    if ((k_cache || v_cache) && !pos_dev && pos0 + S > Tmax) return crab_fail(ctx, CRAB_E_INVALID, "qkv_rope_split: KV cache overflow");
    if (S >= 16 && !pos_dev && !row_off && (d == 32 || d == 64 || d == 128) && (ldqkv & 7) == 0 && (vt == nullptr || (vt_ld & 7) == 0)) {
        dim3 grid((S + 63) / 64, H + 2 * Hk, B);
#define RS_ARGS (bf16_t*)qkv, (long)ldqkv, rope_tab, (bf16_t*)k_cache, (bf16_t*)v_cache, (bf16_t*)vt, (long)vt_ld, S, H, Hk, Tmax, pos0, pos_ids, (long)ld_pos
        if (d == 128) hipLaunchKernelGGL((qkv_rope_split_tile_kernel<128>), grid, dim3(256), 0, S_(stream), RS_ARGS);
        else if (d == 64) hipLaunchKernelGGL((qkv_rope_split_tile_kernel<64>), grid, dim3(256), 0, S_(stream), RS_ARGS);
        else hipLaunchKernelGGL((qkv_rope_split_tile_kernel<32>), grid, dim3(256), 0, S_(stream), RS_ARGS);
#undef RS_ARGS
        return crab_check_launch(ctx, "qkv_rope_split_tile");
    }
    int threads = ((d / 2 + 63) / 64) * 64;
    hipLaunchKernelGGL(qkv_rope_split_kernel, dim3(B * S, H + 2 * Hk), dim3(threads), 0, S_(stream), (bf16_t*)qkv, (long)ldqkv, rope_tab,
                       (bf16_t*)k_cache, (bf16_t*)v_cache, (bf16_t*)vt, (long)vt_ld, S, H, Hk, d, Tmax, pos0, pos_dev, pos_ids, (long)ld_pos, row_off);
    return crab_check_launch(ctx, "qkv_rope_split");
}

int crab_hyperlora_mix(crab_ctx* ctx, void* stream, const void* T, int64_t ldt, int t_fp32, void* U, int64_t ldu, int M, int nproj,
                       int nl, int r, int ucols, float scaling) {
    if (!ctx) return CRAB_E_INVALID;
    if (!T || !U || M <= 0 || nproj <= 0 || nl <= 0 || nl > 8 || r <= 0 || ucols < nproj * nl * r)
        return crab_fail(ctx, CRAB_E_INVALID, "hyperlora_mix: bad argument");
    unsigned blocks = cdiv((long)M * (nproj + 1), 256);
    if (t_fp32)
        hipLaunchKernelGGL((hyperlora_mix_kernel<float>), dim3(blocks), dim3(256), 0, S_(stream), (const float*)T, (long)ldt, (bf16_t*)U,
                           (long)ldu, M, nproj, nl, r, ucols, scaling);
    else
        hipLaunchKernelGGL((hyperlora_mix_kernel<bf16_t>), dim3(blocks), dim3(256), 0, S_(stream), (const bf16_t*)T, (long)ldt,
                           (bf16_t*)U, (long)ldu, M, nproj, nl, r, ucols, scaling);
    return crab_check_launch(ctx, "hyperlora_mix");
}

int crab_swiglu(crab_ctx* ctx, void* stream, const void* gu, int64_t ldgu, void* y, int64_t ldy, int M, int I) {
    if (!ctx) return CRAB_E_INVALID;
    if (!gu || !y || M <= 0 || (I & 7) || (ldgu & 7) || (ldy & 7)) return crab_fail(ctx, CRAB_E_INVALID, "swiglu: bad argument");
    hipLaunchKernelGGL(swiglu_kernel, dim3(cdiv((long)M * (I >> 3), 256)), dim3(256), 0, S_(stream), (const bf16_t*)gu, (long)ldgu,
                       (bf16_t*)y, (long)ldy, M, I);
    return crab_check_launch(ctx, "swiglu");
}

int crab_argmax(crab_ctx* ctx, void* stream, const float* logits, int64_t ldl, int64_t* ids, int B, int V, int suppress) {
    if (!ctx) return CRAB_E_INVALID;
    if (!logits || !ids || B <= 0 || V <= 0) return crab_fail(ctx, CRAB_E_INVALID, "argmax: bad argument");
    hipLaunchKernelGGL(argmax_kernel, dim3(B), dim3(1024), 0, S_(stream), logits, (long)ldl, ids, V, suppress);
    return crab_check_launch(ctx, "argmax");
}

int crab_im2col_patch(crab_ctx* ctx, void* stream, const void* in, int in_fp32, void* out, int64_t ldo, int N, int C, int Hh, int Ww,
                      int P) {
    if (!ctx) return CRAB_E_INVALID;
    if (!in || !out || N <= 0 || P <= 0 || Hh < P || Ww < P || ldo < (int64_t)C * P * P)
        return crab_fail(ctx, CRAB_E_INVALID, "im2col_patch: bad argument");
    int gh = Hh / P, gw = Ww / P;
    const size_t lds = (((size_t)C * P * gw * P + 1) & ~(size_t)1) * sizeof(bf16_t) + (size_t)C * P * P * sizeof(int);
    if (lds > 64 * 1024) return crab_fail(ctx, CRAB_E_UNSUPPORTED, "im2col_patch: a patch row of the image exceeds 64 KiB of LDS");
    if (in_fp32)
        hipLaunchKernelGGL((im2col_patch_kernel<float>), dim3(N * gh), dim3(256), lds, S_(stream), (const float*)in, (bf16_t*)out,
                           (long)ldo, N, C, Hh, Ww, P, gh, gw);
    else
        hipLaunchKernelGGL((im2col_patch_kernel<bf16_t>), dim3(N * gh), dim3(256), lds, S_(stream), (const bf16_t*)in, (bf16_t*)out,
                           (long)ldo, N, C, Hh, Ww, P, gh, gw);
    return crab_check_launch(ctx, "im2col_patch");
}

int crab_clip_embed_ln_p(crab_ctx* ctx, void* stream, const void* patch, const void* cls, const void* pos, const void* lnw, const void* lnb,
                         int ln_fp32, void* y, int N, int P, int D, float eps) {
    if (!ctx) return CRAB_E_INVALID;
    if (!patch || !cls || !pos || !lnw || !lnb || !y || D > 2048) return crab_fail(ctx, CRAB_E_INVALID, "clip_embed_ln: bad argument");
    const bool vec = (D & 7) == 0 && (((uintptr_t)patch | (uintptr_t)cls | (uintptr_t)pos | (uintptr_t)lnw | (uintptr_t)lnb | (uintptr_t)y) & 15) == 0;
    const dim3 grid(cdiv((long)N * (P + 1), 4)), block(256);
#define CE(K_, WF_) hipLaunchKernelGGL((K_<WF_>), grid, block, 0, S_(stream), (const bf16_t*)patch, (const bf16_t*)cls, (const bf16_t*)pos, lnw, lnb, (bf16_t*)y, N, P, D, eps)
    if (vec) { if (ln_fp32) CE(clip_embed_ln_vec_kernel, true); else CE(clip_embed_ln_vec_kernel, false); }
    else { if (ln_fp32) CE(clip_embed_ln_kernel, true); else CE(clip_embed_ln_kernel, false); }
#undef CE
    return crab_check_launch(ctx, "clip_embed_ln");
}

int crab_clip_embed_ln(crab_ctx* ctx, void* stream, const void* patch, const void* cls, const void* pos, const void* lnw, const void* lnb,
                       void* y, int N, int P, int D, float eps) {
    return crab_clip_embed_ln_p(ctx, stream, patch, cls, pos, lnw, lnb, 0, y, N, P, D, eps);
}

int crab_beats_posconv_pad(crab_ctx* ctx, void* stream, const void* x, void* xp, int B, int n, int E, int G, int Kc) {
    if (!ctx) return CRAB_E_INVALID;
    if (!x || !xp || E % G) return crab_fail(ctx, CRAB_E_INVALID, "beats_posconv_pad: bad argument");
    long total = (long)G * B * (n + Kc - 1) * (E / G);
    hipLaunchKernelGGL(beats_posconv_pad_kernel, dim3(cdiv(total, 256)), dim3(256), 0, S_(stream), (const bf16_t*)x, (bf16_t*)xp, B, n, E, G, Kc);
    return crab_check_launch(ctx, "beats_posconv_pad");
}

int crab_beats_relpos_bias(crab_ctx* ctx, void* stream, const void* table, float* bias, int n, int H, int num_buckets, int max_distance) {
    if (!ctx) return CRAB_E_INVALID;
    if (!table || !bias || n <= 0) return crab_fail(ctx, CRAB_E_INVALID, "beats_relpos_bias: bad argument");
    hipLaunchKernelGGL(beats_relpos_bias_kernel, dim3(cdiv((long)n * n, 256)), dim3(256), 0, S_(stream), (const bf16_t*)table, bias, n, H,
                       num_buckets, max_distance);
    return crab_check_launch(ctx, "beats_relpos_bias");
}

int crab_beats_gru_gate(crab_ctx* ctx, void* stream, const void* q, int64_t ldq, const void* gw, const void* gb, const void* grep_a,
                        float* gate, int B, int n, int H, int d) {
    if (!ctx) return CRAB_E_INVALID;
    if (!q || !gw || !gb || !grep_a || !gate || d <= 0 || d > 256) return crab_fail(ctx, CRAB_E_INVALID, "beats_gru_gate: bad argument (head dim <= 256)");
    hipLaunchKernelGGL(beats_gru_gate_kernel, dim3(cdiv((long)B * H * n, 128)), dim3(128), 0, S_(stream), (const bf16_t*)q, (long)ldq,
                       (const bf16_t*)gw, (const bf16_t*)gb, (const bf16_t*)grep_a, gate, B, n, H, d);
    return crab_check_launch(ctx, "beats_gru_gate");
}

int crab_copy_rows(crab_ctx* ctx, void* stream, const void* src, int64_t lds_, void* dst, int64_t ldd, int rows, int cols) {
    if (!ctx) return CRAB_E_INVALID;
    if (!src || !dst || rows <= 0 || cols <= 0) return crab_fail(ctx, CRAB_E_INVALID, "copy_rows: bad argument");
    bool vec = (cols & 7) == 0 && (lds_ & 7) == 0 && (ldd & 7) == 0;
    long total = vec ? (long)rows * (cols >> 3) : (long)rows * cols;
    hipLaunchKernelGGL(copy_rows_kernel, dim3(cdiv(total, 256)), dim3(256), 0, S_(stream), (const bf16_t*)src, (long)lds_, (bf16_t*)dst,
                       (long)ldd, rows, cols);
    return crab_check_launch(ctx, "copy_rows");
}

int crab_copy_rows_batched(crab_ctx* ctx, void* stream, const void* src, int64_t lds_, int64_t sbs, void* dst, int64_t ldd, int64_t dbs,
                           int batch, int rows, int cols) {
    if (!ctx) return CRAB_E_INVALID;
    if (!src || !dst || rows <= 0 || cols <= 0 || batch <= 0) return crab_fail(ctx, CRAB_E_INVALID, "copy_rows_batched: bad argument");
    if ((cols & 7) || (lds_ & 7) || (ldd & 7) || (sbs & 7) || (dbs & 7)) return crab_fail(ctx, CRAB_E_INVALID, "copy_rows_batched: multiples of 8 required");
    hipLaunchKernelGGL(copy_rows_batched_kernel, dim3(cdiv((long)rows * (cols >> 3), 256), batch), dim3(256), 0, S_(stream), (const bf16_t*)src,
                       (long)lds_, (long)sbs, (bf16_t*)dst, (long)ldd, (long)dbs, rows, cols);
    return crab_check_launch(ctx, "copy_rows_batched");
}

int crab_greedy_select(crab_ctx* ctx, void* stream, const float* logits, int64_t ldl, int B, int V, int64_t* cur_ids, int64_t* out_ids,
                       int64_t ld_out, const int32_t* step_dev, int32_t* finished, int eos_id, int pad_id, int min_new_tokens) {
    if (!ctx) return CRAB_E_INVALID;
    if (!logits || !cur_ids || !out_ids || !step_dev || !finished || B <= 0 || V <= 0) return crab_fail(ctx, CRAB_E_INVALID, "greedy_select: bad argument");
    hipLaunchKernelGGL(greedy_select_kernel, dim3(B), dim3(1024), 0, S_(stream), logits, (long)ldl, V, cur_ids, out_ids, (long)ld_out, step_dev,
                       finished, eos_id, pad_id, min_new_tokens);
    return crab_check_launch(ctx, "greedy_select");
}

int crab_advance(crab_ctx* ctx, void* stream, int32_t* pos_dev, int32_t* step_dev) {
    if (!ctx) return CRAB_E_INVALID;
    if (!pos_dev || !step_dev) return crab_fail(ctx, CRAB_E_INVALID, "advance: bad argument");
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, S_(stream), pos_dev, step_dev);
    return crab_check_launch(ctx, "advance");
}

int crab_cast_f32_bf16(crab_ctx* ctx, void* stream, const float* src, void* dst, int64_t n) {
    if (!ctx) return CRAB_E_INVALID;
    if (!src || !dst || n <= 0) return crab_fail(ctx, CRAB_E_INVALID, "cast: bad argument");
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(cdiv(n, 256)), dim3(256), 0, S_(stream), src, (bf16_t*)dst, (long)n);
    return crab_check_launch(ctx, "cast_f32_bf16");
}

}  // extern "C"

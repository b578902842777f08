// bnbwd.hip -- BatchNorm backward fused with the ReLU masks around it, and the element-wise
// gradient fan-out of the residual add.  HBM-bound: every tensor is read once per pass in
// 16-byte channel octets (coalesced pixel rows); per-channel reductions stay in registers, are
// combined through LDS and leave the block as one partial row (no atomics, fixed order).
//
// Replaces cuDNN batch-norm backward + ReLU backward + add backward of
// nn/blocks/__init__.py:41-51,63,78-81,158-180 (and the Conv2dNormActivation triples).
#include <algorithm>
#include <stdlib.h>

#include "common.h"

int rv_col_reduce(const float* partial, int rows, int cols, double* scratch, int* groups, hipStream_t st);

// Register budget of the bandwidth-bound passes: at most 96 VGPRs, so that one of their workgroups (one wave per SIMD) fits on a CU
// BESIDE a resident wgrad3 workgroup (2 waves per SIMD x 200 VGPRs; 28 KB of LDS left) -- the weight-gradient launches run on the
// side stream during the backward sweep, and these passes then use the HBM bandwidth the MFMA-bound kernel leaves idle.
#ifndef RV_EW_WG_PER_CU
#define RV_EW_WG_PER_CU 1
#endif

namespace {

constexpr int kPixPerBlock = 512;

struct BnbArgs {
    const bf16_t* dout;
    const bf16_t* out;  // optional ReLU mask source (the materialised block output)
    const bf16_t* y;    // raw conv output
    const float *scale, *shift, *mean, *invstd, *coef;
    int64_t pixels;
    int c, c8;
    int ld_dout, ld_out, ld_y, ld_dy, ld_dres;
    int flags;
    float* partial;
    bf16_t* dy;
    bf16_t* dres;
};

__device__ __forceinline__ void unpack8(const u32x4 v, float* f) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = bf_lo(v[j]);
        f[2 * j + 1] = bf_hi(v[j]);
    }
}
__device__ __forceinline__ u32x4 pack8(const float* f) {
    u32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = pack_bf2(f[2 * j], f[2 * j + 1]);
    return v;
}

// g = dout * [out > 0] * [scale*y+shift > 0]; xhat = (y - mean) * invstd
struct BnbLoad {
    u32x4 d, y, o;
};
__device__ __forceinline__ BnbLoad load_px(const BnbArgs& a, int64_t px, int c0) {
    BnbLoad r;
    r.d = *(const u32x4*)(a.dout + px * a.ld_dout + c0);
    r.y = *(const u32x4*)(a.y + px * a.ld_y + c0);
    if (a.out) r.o = *(const u32x4*)(a.out + px * a.ld_out + c0);
    return r;
}
__device__ __forceinline__ void masked_grad(const BnbArgs& a, const BnbLoad& r, const float* sc, const float* sh,
                                            const float* mu, const float* is, float* g, float* xhat) {
    float d[8], yv[8];
    unpack8(r.d, d);
    unpack8(r.y, yv);
    if (a.out) {
        float o[8];
        unpack8(r.o, o);
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = o[j] > 0.f ? d[j] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if ((a.flags & RV_BNB_RELU_Z) && !(yv[j] * sc[j] + sh[j] > 0.f)) d[j] = 0.f;
        g[j] = d[j];
        xhat[j] = (yv[j] - mu[j]) * is[j];
    }
}

// Thread layout shared by both passes: thread = (pixel lane, channel octet), so a thread keeps one octet for the whole
// launch and its per-channel constants live in registers; consecutive threads read consecutive 16-byte octets.
__global__ __launch_bounds__(256, RV_EW_WG_PER_CU) void bn_bwd_reduce_kernel(const BnbArgs a) {
    __shared__ float red[256][17];
    const int tid = threadIdx.x;
    const int lanes_px = 256 / a.c8;           // pixels handled per pass
    const int oct = tid % a.c8, pl = tid / a.c8;
    const bool active = pl < lanes_px;
    const int c0 = oct * 8;
    float sc[8], sh[8], mu[8], is[8], s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = a.scale[c0 + j];
        sh[j] = a.shift[c0 + j];
        mu[j] = a.mean[c0 + j];
        is[j] = a.invstd[c0 + j];
        s0[j] = 0.f;
        s1[j] = 0.f;
    }
    const int64_t p0 = (int64_t)blockIdx.x * kPixPerBlock;
    const int64_t p1 = p0 + kPixPerBlock < a.pixels ? p0 + kPixPerBlock : a.pixels;
    if (active) {
        int64_t px = p0 + pl;
        // four pixels (8-12 16-byte loads) in flight per thread
        for (; px + 3 * lanes_px < p1; px += 4 * lanes_px) {
            BnbLoad r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) r[u] = load_px(a, px + u * lanes_px, c0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float g[8], xh[8];
                masked_grad(a, r[u], sc, sh, mu, is, g, xh);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    s0[j] += g[j];
                    s1[j] += g[j] * xh[j];
                }
            }
        }
        for (; px < p1; px += lanes_px) {
            float g[8], xh[8];
            masked_grad(a, load_px(a, px, c0), sc, sh, mu, is, g, xh);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                s0[j] += g[j];
                s1[j] += g[j] * xh[j];
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[tid][j] = s0[j];
        red[tid][8 + j] = s1[j];
    }
    __syncthreads();
    // thread (oct, j in 0..15) sums over the pixel lanes
    for (int i = tid; i < a.c8 * 16; i += 256) {
        const int o = i / 16, j = i - o * 16;
        float s = 0.f;
        for (int l = 0; l < lanes_px; ++l) s += red[l * a.c8 + o][j];
        const int ch = o * 8 + (j & 7);
        a.partial[((int64_t)blockIdx.x * 2 + (j >> 3)) * a.c + ch] = s;
    }
}

// The same reduce pass for the TWO BatchNorms under one block sum out = relu(bn_a(ya) + bn_b(yb)) (BasicBlock with a projection,
// nn/blocks/__init__.py:68-81): both take g = dOut * [out > 0]; one pass over (dOut, out, ya, yb) forms sum g (shared) and
// sum g * xhat for each -- four reads where two separate passes take six.
struct Bnb2Args {
    const bf16_t *dout, *out, *ya, *yb;
    const float *mean_a, *invstd_a, *mean_b, *invstd_b;
    int64_t pixels;
    int c, c8, ld_dout, ld_out, ld_ya, ld_yb;
    float *partial_a, *partial_b;
};
__global__ __launch_bounds__(256, RV_EW_WG_PER_CU) void bn_bwd_reduce2_kernel(const Bnb2Args a) {
    __shared__ float red[256][25];
    const int tid = threadIdx.x;
    const int lanes_px = 256 / a.c8;
    const int oct = tid % a.c8, pl = tid / a.c8;
    const bool active = pl < lanes_px;
    const int c0 = oct * 8;
    float mua[8], isa[8], mub[8], isb[8], s0[8], sa[8], sb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        mua[j] = a.mean_a[c0 + j];
        isa[j] = a.invstd_a[c0 + j];
        mub[j] = a.mean_b[c0 + j];
        isb[j] = a.invstd_b[c0 + j];
        s0[j] = sa[j] = sb[j] = 0.f;
    }
    const int64_t p0 = (int64_t)blockIdx.x * kPixPerBlock;
    const int64_t p1 = p0 + kPixPerBlock < a.pixels ? p0 + kPixPerBlock : a.pixels;
    if (active) {
        auto add = [&](const u32x4 dv, const u32x4 ov, const u32x4 av, const u32x4 bv) {
            float d[8], o[8], ya[8], yb[8];
            unpack8(dv, d);
            unpack8(ov, o);
            unpack8(av, ya);
            unpack8(bv, yb);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float g = o[j] > 0.f ? d[j] : 0.f;
                s0[j] += g;
                sa[j] += g * ((ya[j] - mua[j]) * isa[j]);
                sb[j] += g * ((yb[j] - mub[j]) * isb[j]);
            }
        };
        int64_t px = p0 + pl;
        for (; px + lanes_px < p1; px += 2 * lanes_px) {  // two pixels (eight 16-byte loads) in flight per thread
            u32x4 v[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int64_t q = px + u * lanes_px;
                v[u][0] = *(const u32x4*)(a.dout + q * a.ld_dout + c0);
                v[u][1] = *(const u32x4*)(a.out + q * a.ld_out + c0);
                v[u][2] = *(const u32x4*)(a.ya + q * a.ld_ya + c0);
                v[u][3] = *(const u32x4*)(a.yb + q * a.ld_yb + c0);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) add(v[u][0], v[u][1], v[u][2], v[u][3]);
        }
        for (; px < p1; px += lanes_px)
            add(*(const u32x4*)(a.dout + px * a.ld_dout + c0), *(const u32x4*)(a.out + px * a.ld_out + c0),
                *(const u32x4*)(a.ya + px * a.ld_ya + c0), *(const u32x4*)(a.yb + px * a.ld_yb + c0));
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[tid][j] = s0[j];
        red[tid][8 + j] = sa[j];
        red[tid][16 + j] = sb[j];
    }
    __syncthreads();
    for (int i = tid; i < a.c8 * 24; i += 256) {
        const int o = i / 24, j = i - o * 24;
        float s = 0.f;
        for (int l = 0; l < lanes_px; ++l) s += red[l * a.c8 + o][j];
        const int ch = o * 8 + (j & 7);
        if (j < 8) {
            a.partial_a[((int64_t)blockIdx.x * 2) * a.c + ch] = s;
            a.partial_b[((int64_t)blockIdx.x * 2) * a.c + ch] = s;
        } else if (j < 16) {
            a.partial_a[((int64_t)blockIdx.x * 2 + 1) * a.c + ch] = s;
        } else {
            a.partial_b[((int64_t)blockIdx.x * 2 + 1) * a.c + ch] = s;
        }
    }
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* red, int groups, int c, double inv_count, const float* gamma,
                                       const float* invstd, float* dgamma, float* dbeta, int accumulate, float* coef) {
    __shared__ double part[2][4][64];
    const int cl = threadIdx.x & 63, gl = threadIdx.x >> 6;
    const int ch = blockIdx.x * 64 + cl;
    double s0 = 0.0, s1 = 0.0;
    if (ch < c)
        for (int g = gl; g < groups; g += 4) {
            s0 += red[(int64_t)g * 2 * c + ch];
            s1 += red[(int64_t)g * 2 * c + c + ch];
        }
    part[0][gl][cl] = s0;
    part[1][gl][cl] = s1;
    __syncthreads();
    if (gl != 0 || ch >= c) return;
    s0 = (part[0][0][cl] + part[0][1][cl]) + (part[0][2][cl] + part[0][3][cl]);
    s1 = (part[1][0][cl] + part[1][1][cl]) + (part[1][2][cl] + part[1][3][cl]);
    if (dgamma) dgamma[ch] = (float)((accumulate ? (double)dgamma[ch] : 0.0) + s1);
    if (dbeta) dbeta[ch] = (float)((accumulate ? (double)dbeta[ch] : 0.0) + s0);
    coef[ch] = gamma[ch] * invstd[ch];
    coef[c + ch] = (float)(s0 * inv_count);
    coef[2 * c + ch] = (float)(s1 * inv_count);
}

// single-launch variant of col_reduce + finalize (see bn_reduce_finalize_kernel in misc.hip)
__global__ __launch_bounds__(256) void bn_bwd_reduce_finalize_kernel(const float* partial, int rows, int c, double inv_count, const float* gamma,
                                                                      const float* invstd, float* dgamma, float* dbeta, int accumulate,
                                                                      float* coef, const float* count_dev) {
    if (count_dev) inv_count = 1.0 / (double)count_dev[0];  // SyncBN: global count as a device scalar
    __shared__ double red[2][16][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int ch = blockIdx.x * 16 + cx;
    double s0 = 0.0, s1 = 0.0;
    if (ch < c) {
        const float* p = partial + ch;
        int r = ry;
        for (; r + 48 < rows; r += 64) {
            const float a0 = p[(int64_t)r * 2 * c], b0 = p[(int64_t)r * 2 * c + c];
            const float a1 = p[(int64_t)(r + 16) * 2 * c], b1 = p[(int64_t)(r + 16) * 2 * c + c];
            const float a2 = p[(int64_t)(r + 32) * 2 * c], b2 = p[(int64_t)(r + 32) * 2 * c + c];
            const float a3 = p[(int64_t)(r + 48) * 2 * c], b3 = p[(int64_t)(r + 48) * 2 * c + c];
            s0 += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
            s1 += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
        }
        for (; r < rows; r += 16) {
            s0 += (double)p[(int64_t)r * 2 * c];
            s1 += (double)p[(int64_t)r * 2 * c + c];
        }
    }
    red[0][ry][cx] = s0;
    red[1][ry][cx] = s1;
    __syncthreads();
    for (int half = 8; half > 0; half >>= 1) {
        if (ry < half) {
            red[0][ry][cx] += red[0][ry + half][cx];
            red[1][ry][cx] += red[1][ry + half][cx];
        }
        __syncthreads();
    }
    if (ry != 0 || ch >= c) return;
    s0 = red[0][0][cx];
    s1 = red[1][0][cx];
    if (dgamma) dgamma[ch] = (float)((accumulate ? (double)dgamma[ch] : 0.0) + s1);
    if (dbeta) dbeta[ch] = (float)((accumulate ? (double)dbeta[ch] : 0.0) + s0);
    coef[ch] = gamma[ch] * invstd[ch];
    coef[c + ch] = (float)(s0 * inv_count);
    coef[2 * c + ch] = (float)(s1 * inv_count);
}

// Each workgroup walks ONE CONTIGUOUS pixel range (a grid-stride comb of 4 KB pieces read + wrote at 4.4-4.7 TB/s, contiguous
// ranges reach 5.2-5.5: profiles/r02_hbm_kernels.md).  MODE 1: streaming (non-temporal) loads / stores, chosen for tensors
// that exceed the 256 MB Infinity Cache anyway (+2..25 % there, -15 % on small ones, same file).
template <int MODE>
__device__ __forceinline__ u32x4 ld16(const bf16_t* p) {
    if (MODE & 1) return __builtin_nontemporal_load((const u32x4*)p);
    return *(const u32x4*)p;
}
template <int MODE>
__device__ __forceinline__ void st16(bf16_t* p, const u32x4 v) {
    if (MODE & 1) __builtin_nontemporal_store(v, (u32x4*)p);
    else *(u32x4*)p = v;
}
template <int MODE>
__device__ __forceinline__ BnbLoad load_px_m(const BnbArgs& a, int64_t px, int c0) {
    BnbLoad r;
    r.d = ld16<MODE>(a.dout + px * a.ld_dout + c0);
    r.y = ld16<MODE>(a.y + px * a.ld_y + c0);
    if (a.out) r.o = ld16<MODE>(a.out + px * a.ld_out + c0);
    return r;
}
template <int MODE>
__device__ __forceinline__ void apply_px_m(const BnbArgs& a, int64_t px, int c0, const BnbLoad& r, const u32x4 old,
                                           const float* sc, const float* sh, const float* mu, const float* is,
                                           const float* k0, const float* k1, const float* k2) {
    float g[8], xh[8], o[8];
    masked_grad(a, r, sc, sh, mu, is, g, xh);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = k0[j] * (g[j] - k1[j] - xh[j] * k2[j]);
    st16<MODE>(a.dy + px * a.ld_dy + c0, pack8(o));
    if (a.dres) {
        if (a.flags & RV_BNB_RES_ACCUM) {
            float prev[8];
            unpack8(old, prev);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] += prev[j];
        }
        st16<MODE>(a.dres + px * a.ld_dres + c0, pack8(g));
    }
}

template <int MODE>
__global__ __launch_bounds__(256, RV_EW_WG_PER_CU) void bn_bwd_apply_kernel(const BnbArgs a) {
    const int tid = threadIdx.x;
    const int lanes_px = 256 / a.c8;
    const int oct = tid % a.c8, pl = tid / a.c8;
    if (pl >= lanes_px) return;
    const int c0 = oct * 8;
    float sc[8], sh[8], mu[8], is[8], k0[8], k1[8], k2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = a.scale[c0 + j];
        sh[j] = a.shift[c0 + j];
        mu[j] = a.mean[c0 + j];
        is[j] = a.invstd[c0 + j];
        k0[j] = a.coef[c0 + j];
        k1[j] = a.coef[a.c + c0 + j];
        k2[j] = a.coef[2 * a.c + c0 + j];
    }
    const bool acc = a.dres && (a.flags & RV_BNB_RES_ACCUM);
    const int64_t per = ((a.pixels + gridDim.x - 1) / gridDim.x + lanes_px - 1) / lanes_px * lanes_px;
    const int64_t end = (int64_t)(blockIdx.x + 1) * per < a.pixels ? (int64_t)(blockIdx.x + 1) * per : a.pixels;
    const int64_t step = lanes_px;
    int64_t px = (int64_t)blockIdx.x * per + pl;
    for (; px + step < end; px += 2 * step) {
        const BnbLoad r0 = load_px_m<MODE>(a, px, c0), r1 = load_px_m<MODE>(a, px + step, c0);
        u32x4 o0 = {}, o1 = {};
        if (acc) {
            o0 = *(const u32x4*)(a.dres + px * a.ld_dres + c0);
            o1 = *(const u32x4*)(a.dres + (px + step) * a.ld_dres + c0);
        }
        apply_px_m<MODE>(a, px, c0, r0, o0, sc, sh, mu, is, k0, k1, k2);
        apply_px_m<MODE>(a, px + step, c0, r1, o1, sc, sh, mu, is, k0, k1, k2);
    }
    if (px < end) {
        const BnbLoad r0 = load_px_m<MODE>(a, px, c0);
        u32x4 o0 = {};
        if (acc) o0 = *(const u32x4*)(a.dres + px * a.ld_dres + c0);
        apply_px_m<MODE>(a, px, c0, r0, o0, sc, sh, mu, is, k0, k1, k2);
    }
}

// The LEAN form of the apply pass: thread = (pixel lane, channel QUAD), 8-byte loads, four pixels in flight per thread -- the
// same per-element arithmetic (bit-identical results) in at most 96 VGPRs, so that a workgroup of it (one wave per SIMD) fits on a
// CU BESIDE a resident wgrad3 workgroup (2 waves per SIMD x 200 VGPRs) and streams while that kernel computes
// (engine.py RV3D_OVERLAP=chain).  The per-channel constants are what a thread of these passes spends its registers on (7 per
// channel); halving the channels per thread pays for twice the loads in flight.  F: bit 0 streaming loads / stores, bit 1 ReLU
// mask tensor, bit 2 residual-gradient output, bit 3 ... accumulated onto what is there.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int F>
__global__ __launch_bounds__(256, 5) void bn_bwd_apply_lean_kernel(const BnbArgs a) {
    constexpr bool kNt = F & 1, kOut = F & 2, kRes = F & 4, kAcc = F & 8;
    constexpr int U = 4;
    const int tid = threadIdx.x;
    const int c4 = a.c8 * 2;
    const int lanes_px = 256 / c4;
    const int quad = tid % c4, pl = tid / c4;
    if (pl >= lanes_px) return;
    const int c0 = quad * 4;
    float sc[4], sh[4], mu[4], is[4], k0[4], k1[4], k2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sc[j] = a.scale[c0 + j];
        sh[j] = a.shift[c0 + j];
        mu[j] = a.mean[c0 + j];
        is[j] = a.invstd[c0 + j];
        k0[j] = a.coef[c0 + j];
        k1[j] = a.coef[a.c + c0 + j];
        k2[j] = a.coef[2 * a.c + c0 + j];
    }
    const bool relu_z = (a.flags & RV_BNB_RELU_Z) != 0;
    // 32-bit byte offsets from the (uniform) tensor bases: the loads take the `saddr + voffset` form and the addresses of the
    // pixels in flight cost one register each (64-bit pointers: two, for every (pixel, tensor) pair -- 40 registers)
    auto ld8 = [&](const bf16_t* base, uint32_t off) -> u32x2 {
        const u32x2* p = (const u32x2*)((const char*)base + off);
        return kNt ? __builtin_nontemporal_load(p) : *p;
    };
    auto st8 = [&](bf16_t* base, uint32_t off, const u32x2 v) {
        u32x2* p = (u32x2*)((char*)base + off);
        if (kNt) __builtin_nontemporal_store(v, p);
        else *p = v;
    };
    auto one = [&](uint32_t o_dy, uint32_t o_dr, const u32x2 dv, const u32x2 yv, const u32x2 ov, const u32x2 old) {
        u32x2 dy, dr;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float g[2] = {bf_lo(dv[h]), bf_hi(dv[h])};
            const float y[2] = {bf_lo(yv[h]), bf_hi(yv[h])};
            float o[2];
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int j = 2 * h + e;
                if (kOut && !((e ? bf_hi(ov[h]) : bf_lo(ov[h])) > 0.f)) g[e] = 0.f;
                if (relu_z && !(y[e] * sc[j] + sh[j] > 0.f)) g[e] = 0.f;
                const float xh = (y[e] - mu[j]) * is[j];
                o[e] = k0[j] * (g[e] - k1[j] - xh * k2[j]);
                if (kAcc) g[e] += e ? bf_hi(old[h]) : bf_lo(old[h]);
            }
            dy[h] = pack_bf2(o[0], o[1]);
            dr[h] = pack_bf2(g[0], g[1]);
        }
        st8(a.dy, o_dy, dy);
        if (kRes) st8(a.dres, o_dr, dr);
    };
    const int64_t per = ((a.pixels + gridDim.x - 1) / gridDim.x + lanes_px - 1) / lanes_px * lanes_px;
    const int64_t end = (int64_t)(blockIdx.x + 1) * per < a.pixels ? (int64_t)(blockIdx.x + 1) * per : a.pixels;
    const int64_t first = (int64_t)blockIdx.x * per + pl;
    const int n_px = first < end ? (int)((end - first + lanes_px - 1) / lanes_px) : 0;  // pixels of this thread
    uint32_t o_d = (uint32_t)((first * a.ld_dout + c0) * 2), o_y = (uint32_t)((first * a.ld_y + c0) * 2),
             o_o = kOut ? (uint32_t)((first * a.ld_out + c0) * 2) : 0u, o_dy = (uint32_t)((first * a.ld_dy + c0) * 2),
             o_dr = kRes ? (uint32_t)((first * a.ld_dres + c0) * 2) : 0u;
    const uint32_t s_d = lanes_px * a.ld_dout * 2, s_y = lanes_px * a.ld_y * 2, s_o = lanes_px * a.ld_out * 2, s_dy = lanes_px * a.ld_dy * 2,
                   s_dr = lanes_px * a.ld_dres * 2;  // (uniform: scalar registers)
    int i = 0;
    for (; i + U <= n_px; i += U) {
        u32x2 dv[U], yv[U], ov[U], old[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            dv[u] = ld8(a.dout, o_d + u * s_d);
            yv[u] = ld8(a.y, o_y + u * s_y);
            ov[u] = kOut ? ld8(a.out, o_o + u * s_o) : u32x2{0, 0};
            old[u] = kAcc ? *(const u32x2*)((const char*)a.dres + (o_dr + u * s_dr)) : u32x2{0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) one(o_dy + u * s_dy, o_dr + u * s_dr, dv[u], yv[u], ov[u], old[u]);
        o_d += U * s_d, o_y += U * s_y, o_o += U * s_o, o_dy += U * s_dy, o_dr += U * s_dr;
    }
    for (; i < n_px; ++i) {
        one(o_dy, o_dr, ld8(a.dout, o_d), ld8(a.y, o_y), kOut ? ld8(a.out, o_o) : u32x2{0, 0},
            kAcc ? *(const u32x2*)((const char*)a.dres + o_dr) : u32x2{0, 0});
        o_d += s_d, o_y += s_y, o_o += s_o, o_dy += s_dy, o_dr += s_dr;
    }
}

// ---- lean forms of the other three passes (same layout and register budget as bn_bwd_apply_lean_kernel) ---------------------------
__device__ __forceinline__ u32x2 ldq(const bf16_t* base, uint32_t off) { return *(const u32x2*)((const char*)base + off); }

// one workgroup = kPixPerBlock pixels, like bn_bwd_reduce_kernel; rows [block][2][c]
template <bool OUT>
__global__ __launch_bounds__(256, 5) void bn_bwd_reduce_lean_kernel(const BnbArgs a) {
    __shared__ float red[256][9];
    constexpr int U = 8;
    const int tid = threadIdx.x;
    const int c4 = a.c8 * 2;
    const int lanes_px = 256 / c4;
    const int quad = tid % c4, pl = tid / c4;
    const int c0 = quad * 4;
    float sc[4], sh[4], mu[4], is[4], s0[4], s1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        sc[j] = a.scale[c0 + j];
        sh[j] = a.shift[c0 + j];
        mu[j] = a.mean[c0 + j];
        is[j] = a.invstd[c0 + j];
        s0[j] = 0.f;
        s1[j] = 0.f;
    }
    const bool relu_z = (a.flags & RV_BNB_RELU_Z) != 0;
    const int64_t p0 = (int64_t)blockIdx.x * kPixPerBlock;
    const int64_t p1 = p0 + kPixPerBlock < a.pixels ? p0 + kPixPerBlock : a.pixels;
    const int64_t first = p0 + pl;
    const int n_px = (pl < lanes_px && first < p1) ? (int)((p1 - first + lanes_px - 1) / lanes_px) : 0;
    uint32_t o_d = (uint32_t)((first * a.ld_dout + c0) * 2), o_y = (uint32_t)((first * a.ld_y + c0) * 2),
             o_o = OUT ? (uint32_t)((first * a.ld_out + c0) * 2) : 0u;
    const uint32_t s_d = lanes_px * a.ld_dout * 2, s_y = lanes_px * a.ld_y * 2, s_o = lanes_px * a.ld_out * 2;
    auto add = [&](const u32x2 dv, const u32x2 yv, const u32x2 ov) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int j = 2 * h + e;
                float g = e ? bf_hi(dv[h]) : bf_lo(dv[h]);
                const float y = e ? bf_hi(yv[h]) : bf_lo(yv[h]);
                if (OUT && !((e ? bf_hi(ov[h]) : bf_lo(ov[h])) > 0.f)) g = 0.f;
                if (relu_z && !(y * sc[j] + sh[j] > 0.f)) g = 0.f;
                s0[j] += g;
                s1[j] = fmaf(g, (y - mu[j]) * is[j], s1[j]);  // (explicit: the pair kernel must add the same values in the same way)
            }
    };
    int i = 0;
    for (; i + U <= n_px; i += U) {
        u32x2 dv[U], yv[U], ov[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            dv[u] = ldq(a.dout, o_d + u * s_d);
            yv[u] = ldq(a.y, o_y + u * s_y);
            ov[u] = OUT ? ldq(a.out, o_o + u * s_o) : u32x2{0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) add(dv[u], yv[u], ov[u]);
        o_d += U * s_d, o_y += U * s_y, o_o += U * s_o;
    }
    for (; i < n_px; ++i) {
        add(ldq(a.dout, o_d), ldq(a.y, o_y), OUT ? ldq(a.out, o_o) : u32x2{0, 0});
        o_d += s_d, o_y += s_y, o_o += s_o;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[tid][j] = s0[j];
        red[tid][4 + j] = s1[j];
    }
    __syncthreads();
    for (int k = tid; k < c4 * 8; k += 256) {  // thread (quad, j in 0..7) sums over the pixel lanes
        const int q = k >> 3, j = k & 7;
        float sum = 0.f;
        for (int l = 0; l < lanes_px; ++l) sum += red[l * c4 + q][j];
        a.partial[((int64_t)blockIdx.x * 2 + (j >> 2)) * a.c + q * 4 + (j & 3)] = sum;
    }
}

__global__ __launch_bounds__(256, 5) void bn_bwd_reduce2_lean_kernel(const Bnb2Args a) {
    __shared__ float red[256][13];
    constexpr int U = 4;
    const int tid = threadIdx.x;
    const int c4 = a.c8 * 2;
    const int lanes_px = 256 / c4;
    const int quad = tid % c4, pl = tid / c4;
    const int c0 = quad * 4;
    float mua[4], isa[4], mub[4], isb[4], s0[4], sa[4], sb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        mua[j] = a.mean_a[c0 + j];
        isa[j] = a.invstd_a[c0 + j];
        mub[j] = a.mean_b[c0 + j];
        isb[j] = a.invstd_b[c0 + j];
        s0[j] = sa[j] = sb[j] = 0.f;
    }
    const int64_t p0 = (int64_t)blockIdx.x * kPixPerBlock;
    const int64_t p1 = p0 + kPixPerBlock < a.pixels ? p0 + kPixPerBlock : a.pixels;
    const int64_t first = p0 + pl;
    const int n_px = (pl < lanes_px && first < p1) ? (int)((p1 - first + lanes_px - 1) / lanes_px) : 0;
    uint32_t o_d = (uint32_t)((first * a.ld_dout + c0) * 2), o_o = (uint32_t)((first * a.ld_out + c0) * 2),
             o_a = (uint32_t)((first * a.ld_ya + c0) * 2), o_b = (uint32_t)((first * a.ld_yb + c0) * 2);
    const uint32_t s_d = lanes_px * a.ld_dout * 2, s_o = lanes_px * a.ld_out * 2, s_a = lanes_px * a.ld_ya * 2, s_b = lanes_px * a.ld_yb * 2;
    auto add = [&](const u32x2 dv, const u32x2 ov, const u32x2 av, const u32x2 bv) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int j = 2 * h + e;
                const float g = (e ? bf_hi(ov[h]) : bf_lo(ov[h])) > 0.f ? (e ? bf_hi(dv[h]) : bf_lo(dv[h])) : 0.f;
                s0[j] += g;
                sa[j] = fmaf(g, ((e ? bf_hi(av[h]) : bf_lo(av[h])) - mua[j]) * isa[j], sa[j]);
                sb[j] = fmaf(g, ((e ? bf_hi(bv[h]) : bf_lo(bv[h])) - mub[j]) * isb[j], sb[j]);
            }
    };
    int i = 0;
    for (; i + U <= n_px; i += U) {
        u32x2 dv[U], ov[U], av[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            dv[u] = ldq(a.dout, o_d + u * s_d);
            ov[u] = ldq(a.out, o_o + u * s_o);
            av[u] = ldq(a.ya, o_a + u * s_a);
            bv[u] = ldq(a.yb, o_b + u * s_b);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) add(dv[u], ov[u], av[u], bv[u]);
        o_d += U * s_d, o_o += U * s_o, o_a += U * s_a, o_b += U * s_b;
    }
    for (; i < n_px; ++i) {
        add(ldq(a.dout, o_d), ldq(a.out, o_o), ldq(a.ya, o_a), ldq(a.yb, o_b));
        o_d += s_d, o_o += s_o, o_a += s_a, o_b += s_b;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        red[tid][j] = s0[j];
        red[tid][4 + j] = sa[j];
        red[tid][8 + j] = sb[j];
    }
    __syncthreads();
    for (int k = tid; k < c4 * 12; k += 256) {
        const int q = k / 12, j = k - q * 12;
        float sum = 0.f;
        for (int l = 0; l < lanes_px; ++l) sum += red[l * c4 + q][j];
        const int ch = q * 4 + (j & 3);
        if (j < 4) {
            a.partial_a[((int64_t)blockIdx.x * 2) * a.c + ch] = sum;
            a.partial_b[((int64_t)blockIdx.x * 2) * a.c + ch] = sum;
        } else if (j < 8) {
            a.partial_a[((int64_t)blockIdx.x * 2 + 1) * a.c + ch] = sum;
        } else {
            a.partial_b[((int64_t)blockIdx.x * 2 + 1) * a.c + ch] = sum;
        }
    }
}

// The apply pass for the two BatchNorms under one block sum (see bn_bwd_reduce2_kernel): g = dOut * [out > 0] is formed once,
// dYa / dYb leave together -- four reads and two writes where two apply passes take six and two.
struct Bnb2Apply {
    const bf16_t *dout, *out, *ya, *yb;
    const float *mean_a, *invstd_a, *coef_a, *mean_b, *invstd_b, *coef_b;
    bf16_t *dya, *dyb;
    int64_t pixels;
    int c, c8, ld_dout, ld_out, ld_ya, ld_yb, ld_dya, ld_dyb;
};
template <int MODE>
__global__ __launch_bounds__(256, RV_EW_WG_PER_CU) void bn_bwd_apply2_kernel(const Bnb2Apply a) {
    const int tid = threadIdx.x;
    const int lanes_px = 256 / a.c8;
    const int oct = tid % a.c8, pl = tid / a.c8;
    if (pl >= lanes_px) return;
    const int c0 = oct * 8;
    float mua[8], isa[8], ka0[8], ka1[8], ka2[8], mub[8], isb[8], kb0[8], kb1[8], kb2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        mua[j] = a.mean_a[c0 + j], isa[j] = a.invstd_a[c0 + j];
        ka0[j] = a.coef_a[c0 + j], ka1[j] = a.coef_a[a.c + c0 + j], ka2[j] = a.coef_a[2 * a.c + c0 + j];
        mub[j] = a.mean_b[c0 + j], isb[j] = a.invstd_b[c0 + j];
        kb0[j] = a.coef_b[c0 + j], kb1[j] = a.coef_b[a.c + c0 + j], kb2[j] = a.coef_b[2 * a.c + c0 + j];
    }
    const int64_t per = ((a.pixels + gridDim.x - 1) / gridDim.x + lanes_px - 1) / lanes_px * lanes_px;
    const int64_t end = (int64_t)(blockIdx.x + 1) * per < a.pixels ? (int64_t)(blockIdx.x + 1) * per : a.pixels;
    for (int64_t px = (int64_t)blockIdx.x * per + pl; px < end; px += lanes_px) {
        float d[8], o[8], ya[8], yb[8], ra[8], rb[8];
        unpack8(ld16<MODE>(a.dout + px * a.ld_dout + c0), d);
        unpack8(ld16<MODE>(a.out + px * a.ld_out + c0), o);
        unpack8(ld16<MODE>(a.ya + px * a.ld_ya + c0), ya);
        unpack8(ld16<MODE>(a.yb + px * a.ld_yb + c0), yb);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float g = o[j] > 0.f ? d[j] : 0.f;
            const float xa = (ya[j] - mua[j]) * isa[j], xb = (yb[j] - mub[j]) * isb[j];
            ra[j] = ka0[j] * (g - ka1[j] - xa * ka2[j]);
            rb[j] = kb0[j] * (g - kb1[j] - xb * kb2[j]);
        }
        st16<MODE>(a.dya + px * a.ld_dya + c0, pack8(ra));
        st16<MODE>(a.dyb + px * a.ld_dyb + c0, pack8(rb));
    }
}

__global__ __launch_bounds__(256) void ew_mask_grad_kernel(int64_t pixels, int c8, const bf16_t* dout, int ld_dout,
                                                           const bf16_t* out, int ld_out, bf16_t* d, int ld_d,
                                                           int accumulate) {
    const int64_t total = pixels * c8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t px = i / c8;
        const int c0 = (int)(i - px * c8) * 8;
        float g[8];
        unpack8(*(const u32x4*)(dout + px * ld_dout + c0), g);
        if (out) {
            float o[8];
            unpack8(*(const u32x4*)(out + px * ld_out + c0), o);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] = o[j] > 0.f ? g[j] : 0.f;
        }
        bf16_t* p = d + px * ld_d + c0;
        if (accumulate) {
            float old[8];
            unpack8(*(const u32x4*)p, old);
#pragma unroll
            for (int j = 0; j < 8; ++j) g[j] += old[j];
        }
        *(u32x4*)p = pack8(g);
    }
}

// ---------------------------------------------------------------------------------------------
// Small-K layers without an input gradient (the stem's 3 -> C positional conv on the 9x unfolded grid, the 5 -> C
// feature projections): BatchNorm backward AND the 1x1 conv's weight gradient from ONE pass over (dOut, y, v).
//   dy = k0 (g - S0/n - xhat S1/n),  k0 = gamma*invstd,  S0 = sum g,  S1 = sum g*xhat
//   dW[c][d] = sum_p dy[p,c] v[p,d]
//            = k0 ( R[c][d] - (S0/n) m1[d] - (S1/n) invstd_c ( sum_e W[c,e] M2[e][d] - mean_c m1[d] ) )
// with R[c][d] = sum_p g[p,c] v[p,d] and the data moments m1 = sum_p v, M2 = sum_p v v^T (y = W v is linear in v).
// So dy is never written (2.4 GB on the stem) and the separate apply and weight-gradient passes disappear.
// ---------------------------------------------------------------------------------------------
// RECOMP: y = W v is recomputed from the <= 8 input channels (16 bytes per pixel) instead of read back (2 C bytes per
// pixel: 2.4 GB on the stem's 3 -> 256 positional layer), and the ReLU gate / xhat come from the fp32 value with the
// layer's true statistics rather than from the bf16-rounded activated output.
template <int CIN, bool RECOMP>
__global__ __launch_bounds__(256) void bn_bwd_smallk_reduce_kernel(const BnbArgs a, const bf16_t* v, int ld_v, const bf16_t* w, int ld_w) {
    __shared__ float red[256][9];
    const int tid = threadIdx.x;
    const int lanes_px = 256 / a.c8;
    const int oct = tid % a.c8, pl = tid / a.c8;
    const bool active = pl < lanes_px;
    const int c0 = oct * 8;
    float sc[8], sh[8], mu[8], is[8], s0[8], s1[8], r[CIN][8], wt[RECOMP ? 8 : 1][CIN];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = a.scale[c0 + j];
        sh[j] = a.shift[c0 + j];
        mu[j] = a.mean[c0 + j];
        is[j] = a.invstd[c0 + j];
        s0[j] = 0.f;
        s1[j] = 0.f;
#pragma unroll
        for (int d = 0; d < CIN; ++d) r[d][j] = 0.f;
        if (RECOMP) {
#pragma unroll
            for (int e = 0; e < CIN; ++e) wt[RECOMP ? j : 0][e] = bf2f(w[(int64_t)(c0 + j) * ld_w + e]);
        }
    }
    const int64_t p0 = (int64_t)blockIdx.x * kPixPerBlock;
    const int64_t p1 = p0 + kPixPerBlock < a.pixels ? p0 + kPixPerBlock : a.pixels;
    auto load = [&](int64_t px) {
        BnbLoad l;
        l.d = *(const u32x4*)(a.dout + px * a.ld_dout + c0);
        if (!RECOMP) l.y = *(const u32x4*)(a.y + px * a.ld_y + c0);
        if (a.out) l.o = *(const u32x4*)(a.out + px * a.ld_out + c0);
        return l;
    };
    auto accumulate = [&](const BnbLoad& l, const u32x4 wv) {
        float g[8], xh[8], vv[8];
        unpack8(wv, vv);
        if (RECOMP) {
            float d[8];
            unpack8(l.d, d);
            if (a.out) {
                float o[8];
                unpack8(l.o, o);
#pragma unroll
                for (int j = 0; j < 8; ++j) d[j] = o[j] > 0.f ? d[j] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float y = 0.f;
#pragma unroll
                for (int e = 0; e < CIN; ++e) y += wt[RECOMP ? j : 0][e] * vv[e];
                if ((a.flags & RV_BNB_RELU_Z) && !(y * sc[j] + sh[j] > 0.f)) d[j] = 0.f;
                g[j] = d[j];
                xh[j] = (y - mu[j]) * is[j];
            }
        } else {
            masked_grad(a, l, sc, sh, mu, is, g, xh);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s0[j] += g[j];
            s1[j] += g[j] * xh[j];
#pragma unroll
            for (int d = 0; d < CIN; ++d) r[d][j] += g[j] * vv[d];
        }
    };
    if (active) {
        int64_t px = p0 + pl;
        for (; px + lanes_px < p1; px += 2 * lanes_px) {
            const BnbLoad l0 = load(px), l1 = load(px + lanes_px);
            const u32x4 w0 = *(const u32x4*)(v + px * ld_v), w1 = *(const u32x4*)(v + (px + lanes_px) * ld_v);
            accumulate(l0, w0);
            accumulate(l1, w1);
        }
        for (; px < p1; px += lanes_px) accumulate(load(px), *(const u32x4*)(v + px * ld_v));
    }
    // planes 0 (S0), 1 (S1), 2 + d (R[.][d]): block sums over the pixel lanes, one plane at a time through LDS
    for (int plane = 0; plane < 2 + CIN; ++plane) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float val = plane == 0 ? s0[j] : (plane == 1 ? s1[j] : 0.f);
#pragma unroll
            for (int d = 0; d < CIN; ++d)
                if (plane == 2 + d) val = r[d][j];
            red[tid][j] = val;
        }
        __syncthreads();
        for (int i = tid; i < a.c8 * 8; i += 256) {
            const int o = i >> 3, j = i & 7;
            float s = 0.f;
            for (int l = 0; l < lanes_px; ++l) s += red[l * a.c8 + o][j];
            a.partial[((int64_t)blockIdx.x * (2 + CIN) + plane) * a.c + o * 8 + j] = s;
        }
        __syncthreads();
    }
}

// data moments of the conv input: per block one row of CIN + CIN*CIN sums (m1, M2 row-major)
template <int CIN>
__global__ __launch_bounds__(256) void smallk_moments_kernel(const bf16_t* v, int ld_v, int64_t pixels, float* partial) {
    __shared__ float red[256];
    float m[CIN + CIN * CIN];
#pragma unroll
    for (int i = 0; i < CIN + CIN * CIN; ++i) m[i] = 0.f;
    const int64_t per = (pixels + gridDim.x - 1) / gridDim.x;
    const int64_t p0 = blockIdx.x * per, p1 = p0 + per < pixels ? p0 + per : pixels;
    for (int64_t px = p0 + threadIdx.x; px < p1; px += 256) {
        float vv[8];
        unpack8(*(const u32x4*)(v + px * ld_v), vv);
#pragma unroll
        for (int d = 0; d < CIN; ++d) {
            m[d] += vv[d];
#pragma unroll
            for (int e = 0; e < CIN; ++e) m[CIN + d * CIN + e] += vv[d] * vv[e];
        }
    }
    for (int i = 0; i < CIN + CIN * CIN; ++i) {
        red[threadIdx.x] = m[i];
        __syncthreads();
        for (int h = 128; h > 0; h >>= 1) {
            if ((int)threadIdx.x < h) red[threadIdx.x] += red[threadIdx.x + h];
            __syncthreads();
        }
        if (threadIdx.x == 0) partial[(int64_t)blockIdx.x * (CIN + CIN * CIN) + i] = red[0];
        __syncthreads();
    }
}

// red_g: [groups][(2+CIN)*c] fp64, red_m: [groups_m][CIN+CIN*CIN] fp64, w: bf16 [c][ld_w] (the packed gather image)
template <int CIN>
__global__ void bn_bwd_smallk_finalize_kernel(const double* red_g, int groups, const double* red_m, int groups_m, int c, int cin,
                                              double inv_count, const float* gamma, const float* mean, const float* invstd,
                                              const bf16_t* w, int ld_w, const double* global_s01, const double* count_dev,
                                              float* dgamma, float* dbeta, float* dW) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    if (count_dev) inv_count = 1.0 / *count_dev;  // SyncBN: the all-reduced pixel count, read on device (no host sync)
    double S[2 + CIN];
    for (int p = 0; p < 2 + CIN; ++p) {
        double s = 0.0;
        for (int g = 0; g < groups; ++g) s += red_g[(int64_t)g * (2 + CIN) * c + (int64_t)p * c + ch];
        S[p] = s;
    }
    double m[CIN + CIN * CIN];
    for (int i = 0; i < CIN + CIN * CIN; ++i) {
        double s = 0.0;
        for (int g = 0; g < groups_m; ++g) s += red_m[(int64_t)g * (CIN + CIN * CIN) + i];
        m[i] = s;
    }
    dbeta[ch] = (float)S[0];
    dgamma[ch] = (float)S[1];
    const double k0 = (double)gamma[ch] * (double)invstd[ch];
    // SyncBN: the normalisation coefficients use the all-reduced (sum g, sum g*xhat); everything else stays this rank's
    const double k1 = (global_s01 ? global_s01[ch] : S[0]) * inv_count, k2 = (global_s01 ? global_s01[c + ch] : S[1]) * inv_count;
    for (int d = 0; d < cin; ++d) {
        double ym = 0.0;  // sum_p y[p,ch] v[p,d] = sum_e W[ch,e] M2[e][d]
        for (int e = 0; e < cin; ++e) ym += (double)bf2f(w[(int64_t)ch * ld_w + e]) * m[CIN + e * CIN + d];
        const double xv = (double)invstd[ch] * (ym - (double)mean[ch] * m[d]);
        dW[(int64_t)ch * cin + d] = (float)(k0 * (S[2 + d] - k1 * m[d] - k2 * xv));
    }
}


// ---------------------------------------------------------------------------------------------
// Small-K forward: h = relu?(BatchNorm(W v)) for a 1x1 conv with cin <= 8.  K is so small that the layer is an
// element-wise map writing C channels per pixel, and its batch statistics follow in closed form from the data moments
//   mean_c = w_c . m1 / n,   E[y_c^2] = w_c^T M2 w_c / n      (m1 = sum_p v, M2 = sum_p v v^T)
// so there is no statistics pass over the (9x unfolded, 2.4 GB) output and no raw conv output in HBM at all.
// ---------------------------------------------------------------------------------------------
template <int CIN>
__global__ void smallk_stats_kernel(const double* red_m, int groups_m, int c, int cin, const bf16_t* w, int ld_w, double inv_count,
                                    double unbias, const double* count_dev, const float* gamma, const float* beta, float eps, float momentum,
                                    float* running_mean, float* running_var, float* scale, float* shift, float* mean_out,
                                    float* invstd_out) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    if (count_dev) {  // SyncBN: the all-reduced pixel count travels with the moments and is read here (no host sync)
        const double n = *count_dev;
        inv_count = 1.0 / n;
        unbias = n > 1.0 ? n / (n - 1.0) : 1.0;
    }
    double m[CIN + CIN * CIN];
    for (int i = 0; i < CIN + CIN * CIN; ++i) {
        double s = 0.0;
        for (int g = 0; g < groups_m; ++g) s += red_m[(int64_t)g * (CIN + CIN * CIN) + i];
        m[i] = s;
    }
    double wv[CIN];
    for (int e = 0; e < CIN; ++e) wv[e] = e < cin ? (double)bf2f(w[(int64_t)ch * ld_w + e]) : 0.0;
    double s1 = 0.0, s2 = 0.0;
    for (int e = 0; e < cin; ++e) {
        s1 += wv[e] * m[e];
        for (int f = 0; f < cin; ++f) s2 += wv[e] * wv[f] * m[CIN + e * CIN + f];
    }
    const double mean = s1 * inv_count;
    double var = s2 * inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)eps);
    const double sc = (double)gamma[ch] * invstd;
    scale[ch] = (float)sc;
    shift[ch] = (float)((double)beta[ch] - mean * sc);
    if (mean_out) mean_out[ch] = (float)mean;
    if (invstd_out) invstd_out[ch] = (float)invstd;
    if (running_mean) running_mean[ch] = (float)((1.0 - momentum) * running_mean[ch] + momentum * mean);
    if (running_var) running_var[ch] = (float)((1.0 - momentum) * running_var[ch] + momentum * var * unbias);
}

template <int CIN>
__global__ __launch_bounds__(256) void smallk_apply_kernel(const bf16_t* v, int ld_v, int64_t pixels, int c8, const bf16_t* w, int ld_w,
                                                           const float* scale, const float* shift, int relu, bf16_t* h, int ld_h) {
    const int tid = threadIdx.x;
    const int lanes_px = 256 / c8;
    const int oct = tid % c8, pl = tid / c8;
    if (pl >= lanes_px) return;
    const int c0 = oct * 8;
    float wt[8][CIN], sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = scale[c0 + j];
        sh[j] = shift[c0 + j];
#pragma unroll
        for (int e = 0; e < CIN; ++e) wt[j][e] = bf2f(w[(int64_t)(c0 + j) * ld_w + e]);
    }
    // one contiguous pixel range per workgroup, streaming stores beyond the Infinity Cache (see bn_bwd_apply_kernel)
    const bool nt = pixels * c8 * 16 >= ((int64_t)256 << 20);
    const int64_t per = ((pixels + gridDim.x - 1) / gridDim.x + lanes_px - 1) / lanes_px * lanes_px;
    const int64_t end = (int64_t)(blockIdx.x + 1) * per < pixels ? (int64_t)(blockIdx.x + 1) * per : pixels;
    for (int64_t px = (int64_t)blockIdx.x * per + pl; px < end; px += lanes_px) {
        float vv[8], o[8];
        unpack8(*(const u32x4*)(v + px * ld_v), vv);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float y = 0.f;
#pragma unroll
            for (int e = 0; e < CIN; ++e) y += wt[j][e] * vv[e];
            y = y * sc[j] + sh[j];
            o[j] = relu ? fmaxf(y, 0.f) : y;
        }
        if (nt) __builtin_nontemporal_store(pack8(o), (u32x4*)(h + px * ld_h + c0));
        else *(u32x4*)(h + px * ld_h + c0) = pack8(o);
    }
}

__global__ void sum_rows_f64_kernel(const double* rows, int n_rows, int cols, double* out) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < cols; i += gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int r = 0; r < n_rows; ++r) s += rows[(int64_t)r * cols + i];
        out[i] = s;
    }
}

// the lean (quad layout, <= 96 VGPRs) forms of the four BatchNorm-backward passes are what runs; the octet forms take tensors whose byte
// offsets do not fit 32 bits (A/B of the two and of a one-workgroup-per-CU launch: profiles/r04_ab_notes.md)
constexpr int bnb_lean() { return 2; }

int grid_for(int64_t work) {
    int64_t b = (work + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

int fill(BnbArgs* a, int64_t pixels, int32_t c, const void* dout, int32_t ld_dout, const void* out, int32_t ld_out,
         const void* y, int32_t ld_y, const float* scale, const float* shift, const float* mean, const float* invstd,
         int32_t flags) {
    RV_REQUIRE(dout && (y || (flags & RV_BNB_Y_FROM_INPUT)) && scale && shift && mean && invstd, "bn backward: null argument");
    RV_REQUIRE(c % 8 == 0 && c / 8 <= 256 && ld_dout % 8 == 0 && ld_y % 8 == 0 && (!out || ld_out % 8 == 0), "bn backward: channels / strides must be multiples of 8 (c <= 2048)");
    memset(a, 0, sizeof(*a));
    a->dout = (const bf16_t*)dout;
    a->out = (const bf16_t*)out;
    a->y = (const bf16_t*)y;
    a->scale = scale;
    a->shift = shift;
    a->mean = mean;
    a->invstd = invstd;
    a->pixels = pixels;
    a->c = c;
    a->c8 = c / 8;
    a->ld_dout = ld_dout;
    a->ld_out = ld_out;
    a->ld_y = ld_y;
    a->flags = flags;
    return 0;
}

}  // namespace

extern "C" int32_t rv_bn_bwd_rows(int64_t pixels) { return (int32_t)((pixels + kPixPerBlock - 1) / kPixPerBlock); }

extern "C" int rv_bn_bwd_reduce(int64_t pixels, int32_t c, const void* dout, int32_t ld_dout, const void* out,
                                int32_t ld_out, const void* y, int32_t ld_y, const float* scale, const float* shift,
                                const float* mean, const float* invstd, int32_t flags, float* partial, rvStream stream) {
    BnbArgs a;
    if (fill(&a, pixels, c, dout, ld_dout, out, ld_out, y, ld_y, scale, shift, mean, invstd, flags)) return 1;
    RV_REQUIRE(partial, "rv_bn_bwd_reduce: null partial buffer");
    a.partial = partial;
    if (bnb_lean() && c <= 1024 && pixels * std::max(std::max((int64_t)ld_dout, (int64_t)ld_y), (int64_t)ld_out) * 2 < ((int64_t)1 << 32)) {
        if (out) hipLaunchKernelGGL(bn_bwd_reduce_lean_kernel<true>, dim3(rv_bn_bwd_rows(pixels)), dim3(256), 0, (hipStream_t)stream, a);
        else hipLaunchKernelGGL(bn_bwd_reduce_lean_kernel<false>, dim3(rv_bn_bwd_rows(pixels)), dim3(256), 0, (hipStream_t)stream, a);
        RV_CHECK_LAUNCH("bn_bwd_reduce_lean_kernel");
        return 0;
    }
    hipLaunchKernelGGL(bn_bwd_reduce_kernel, dim3(rv_bn_bwd_rows(pixels)), dim3(256), 0, (hipStream_t)stream, a);
    RV_CHECK_LAUNCH("bn_bwd_reduce_kernel");
    return 0;
}

extern "C" int rv_bn_bwd_reduce_pair(int64_t pixels, int32_t c, const void* dout, int32_t ld_dout, const void* out, int32_t ld_out,
                                     const void* ya, int32_t ld_ya, const float* mean_a, const float* invstd_a, const void* yb, int32_t ld_yb,
                                     const float* mean_b, const float* invstd_b, float* partial_a, float* partial_b, rvStream stream) {
    RV_REQUIRE(dout && out && ya && yb && mean_a && invstd_a && mean_b && invstd_b && partial_a && partial_b, "rv_bn_bwd_reduce_pair: null argument");
    RV_REQUIRE(pixels > 0 && c > 0 && c % 8 == 0 && c <= 2048 && ld_dout % 8 == 0 && ld_out % 8 == 0 && ld_ya % 8 == 0 && ld_yb % 8 == 0 &&
                   ld_dout >= c && ld_out >= c && ld_ya >= c && ld_yb >= c,
               "rv_bn_bwd_reduce_pair: channels / strides must be multiples of 8 (at most 2048 channels)");
    Bnb2Args a{};
    a.dout = (const bf16_t*)dout, a.out = (const bf16_t*)out, a.ya = (const bf16_t*)ya, a.yb = (const bf16_t*)yb;
    a.mean_a = mean_a, a.invstd_a = invstd_a, a.mean_b = mean_b, a.invstd_b = invstd_b;
    a.pixels = pixels, a.c = c, a.c8 = c / 8, a.ld_dout = ld_dout, a.ld_out = ld_out, a.ld_ya = ld_ya, a.ld_yb = ld_yb;
    a.partial_a = partial_a, a.partial_b = partial_b;
    if (bnb_lean() && c <= 1024 && pixels * std::max(std::max((int64_t)ld_dout, (int64_t)ld_out), std::max((int64_t)ld_ya, (int64_t)ld_yb)) * 2 < ((int64_t)1 << 32)) {
        hipLaunchKernelGGL(bn_bwd_reduce2_lean_kernel, dim3(rv_bn_bwd_rows(pixels)), dim3(256), 0, (hipStream_t)stream, a);
        RV_CHECK_LAUNCH("bn_bwd_reduce2_lean_kernel");
        return 0;
    }
    hipLaunchKernelGGL(bn_bwd_reduce2_kernel, dim3(rv_bn_bwd_rows(pixels)), dim3(256), 0, (hipStream_t)stream, a);
    RV_CHECK_LAUNCH("bn_bwd_reduce2_kernel");
    return 0;
}

extern "C" int rv_bn_bwd_finalize(const float* partial, int32_t rows, int32_t c, int64_t count, const float* gamma,
                                  const float* invstd, float* dgamma, float* dbeta, int32_t accumulate, float* coef,
                                  rvStream stream) {
    RV_REQUIRE(partial && gamma && invstd && coef, "rv_bn_bwd_finalize: null argument");
    RV_REQUIRE(count > 0 || (count < 0 && rows == 1), "rv_bn_bwd_finalize: a device-side count (count < 0) needs the single row of all-reduced totals");
    const float* count_dev = count < 0 ? partial + 2 * c : nullptr;
    if (rows <= 2048) {
        hipLaunchKernelGGL(bn_bwd_reduce_finalize_kernel, dim3(rv_ceil_div(c, 16)), dim3(256), 0, (hipStream_t)stream, partial, rows, c,
                           1.0 / (double)count, gamma, invstd, dgamma, dbeta, accumulate, coef, count_dev);
        RV_CHECK_LAUNCH("bn_bwd_reduce_finalize_kernel");
        return 0;
    }
    RV_REQUIRE(count > 0, "rv_bn_bwd_finalize: the two-stage path (> 2048 rows) takes a host-side count only");
    double* scratch = (double*)(partial + (int64_t)rows * 2 * c);
    int groups;
    if (rv_col_reduce(partial, rows, 2 * c, scratch, &groups, (hipStream_t)stream)) return 1;
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(rv_ceil_div(c, 64)), dim3(256), 0, (hipStream_t)stream, scratch, groups, c,
                       1.0 / (double)count, gamma, invstd, dgamma, dbeta, accumulate, coef);
    RV_CHECK_LAUNCH("bn_bwd_finalize_kernel");
    return 0;
}

extern "C" int rv_bn_bwd_apply(int64_t pixels, int32_t c, const void* dout, int32_t ld_dout, const void* out,
                               int32_t ld_out, const void* y, int32_t ld_y, const float* scale, const float* shift,
                               const float* mean, const float* invstd, const float* coef, int32_t flags, void* dy,
                               int32_t ld_dy, void* dres, int32_t ld_dres, rvStream stream) {
    BnbArgs a;
    if (fill(&a, pixels, c, dout, ld_dout, out, ld_out, y, ld_y, scale, shift, mean, invstd, flags)) return 1;
    RV_REQUIRE(coef && dy && ld_dy % 8 == 0 && (!dres || ld_dres % 8 == 0), "rv_bn_bwd_apply: bad outputs");
    a.coef = coef;
    a.dy = (bf16_t*)dy;
    a.ld_dy = ld_dy;
    a.dres = (bf16_t*)dres;
    a.ld_dres = ld_dres;
    // lean form: alone it takes the same time (96.7 against 96.5-96.9 ms
    // per rv-av2 step), beside a weight gradient on the side stream (engine.py RV3D_OVERLAP=chain) it is what fits on the CU
    const int lean = bnb_lean();
    const int64_t ld_max = std::max(std::max((int64_t)ld_dout, (int64_t)ld_y), std::max(std::max((int64_t)ld_out, (int64_t)ld_dy), (int64_t)ld_dres));
    if (lean && c <= 1024 && pixels * ld_max * 2 < ((int64_t)1 << 32)) {  // (32-bit byte offsets)
        const int lanes4 = 256 / (2 * a.c8);
        int64_t blocks4 = (pixels + lanes4 - 1) / lanes4;
        if (blocks4 > 4096) blocks4 = 4096;
        const int f = ((int64_t)pixels * c * 2 >= ((int64_t)256 << 20) ? 1 : 0) | (out ? 2 : 0) | (dres ? 4 : 0) |
                      (dres && (flags & RV_BNB_RES_ACCUM) ? 8 : 0);
        using K = void (*)(const BnbArgs);
        static const K table[16] = {bn_bwd_apply_lean_kernel<0>,  bn_bwd_apply_lean_kernel<1>,  bn_bwd_apply_lean_kernel<2>,  bn_bwd_apply_lean_kernel<3>,
                                    bn_bwd_apply_lean_kernel<4>,  bn_bwd_apply_lean_kernel<5>,  bn_bwd_apply_lean_kernel<6>,  bn_bwd_apply_lean_kernel<7>,
                                    bn_bwd_apply_lean_kernel<4>,  bn_bwd_apply_lean_kernel<5>,  bn_bwd_apply_lean_kernel<6>,  bn_bwd_apply_lean_kernel<7>,
                                    bn_bwd_apply_lean_kernel<12>, bn_bwd_apply_lean_kernel<13>, bn_bwd_apply_lean_kernel<14>, bn_bwd_apply_lean_kernel<15>};
        hipLaunchKernelGGL(table[f], dim3((int)blocks4), dim3(256), 0, (hipStream_t)stream, a);
        RV_CHECK_LAUNCH("bn_bwd_apply_lean_kernel");
        return 0;
    }
    const int lanes_px = 256 / a.c8;
    int64_t blocks = (pixels + lanes_px - 1) / lanes_px;
    if (blocks > 4096) blocks = 4096;
    if ((int64_t)pixels * c * 2 >= ((int64_t)256 << 20))
        hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(bn_bwd_apply_kernel<0>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, a);
    RV_CHECK_LAUNCH("bn_bwd_apply_kernel");
    return 0;
}

extern "C" int rv_bn_bwd_apply_pair(int64_t pixels, int32_t c, const void* dout, int32_t ld_dout, const void* out, int32_t ld_out,
                                    const void* ya, int32_t ld_ya, const float* mean_a, const float* invstd_a, const float* coef_a, void* dya,
                                    int32_t ld_dya, const void* yb, int32_t ld_yb, const float* mean_b, const float* invstd_b,
                                    const float* coef_b, void* dyb, int32_t ld_dyb, rvStream stream) {
    RV_REQUIRE(dout && out && ya && yb && mean_a && invstd_a && coef_a && dya && mean_b && invstd_b && coef_b && dyb, "rv_bn_bwd_apply_pair: null argument");
    RV_REQUIRE(pixels > 0 && c > 0 && c % 8 == 0 && c <= 2048 && ld_dout % 8 == 0 && ld_out % 8 == 0 && ld_ya % 8 == 0 && ld_yb % 8 == 0 &&
                   ld_dya % 8 == 0 && ld_dyb % 8 == 0 && ld_dout >= c && ld_out >= c && ld_ya >= c && ld_yb >= c && ld_dya >= c && ld_dyb >= c,
               "rv_bn_bwd_apply_pair: channels / strides must be multiples of 8 (at most 2048 channels)");
    Bnb2Apply a{};
    a.dout = (const bf16_t*)dout, a.out = (const bf16_t*)out, a.ya = (const bf16_t*)ya, a.yb = (const bf16_t*)yb;
    a.mean_a = mean_a, a.invstd_a = invstd_a, a.coef_a = coef_a, a.mean_b = mean_b, a.invstd_b = invstd_b, a.coef_b = coef_b;
    a.dya = (bf16_t*)dya, a.dyb = (bf16_t*)dyb;
    a.pixels = pixels, a.c = c, a.c8 = c / 8;
    a.ld_dout = ld_dout, a.ld_out = ld_out, a.ld_ya = ld_ya, a.ld_yb = ld_yb, a.ld_dya = ld_dya, a.ld_dyb = ld_dyb;
    // (no lean form of this one: ten constants per channel -- 40 registers at four channels per thread -- leave no room for loads in
    //  flight within 96 registers; it runs beside a weight gradient only where that one leaves more)
    const int lanes_px = 256 / a.c8;
    int64_t blocks = (pixels + lanes_px - 1) / lanes_px;
    if (blocks > 4096) blocks = 4096;
    if ((int64_t)pixels * c * 2 >= ((int64_t)256 << 20))
        hipLaunchKernelGGL(bn_bwd_apply2_kernel<1>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(bn_bwd_apply2_kernel<0>, dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, a);
    RV_CHECK_LAUNCH("bn_bwd_apply2_kernel");
    return 0;
}

extern "C" int rv_ew_mask_grad(int64_t pixels, int32_t c, const void* dout, int32_t ld_dout, const void* out,
                               int32_t ld_out, void* d, int32_t ld_d, int32_t accumulate, rvStream stream) {
    RV_REQUIRE(dout && d, "rv_ew_mask_grad: null argument");
    RV_REQUIRE(c % 8 == 0 && ld_dout % 8 == 0 && ld_d % 8 == 0 && (!out || ld_out % 8 == 0), "rv_ew_mask_grad: channels / strides must be multiples of 8");
    hipLaunchKernelGGL(ew_mask_grad_kernel, dim3(grid_for(pixels * (c / 8))), dim3(256), 0, (hipStream_t)stream, pixels,
                       c / 8, (const bf16_t*)dout, ld_dout, (const bf16_t*)out, ld_out, (bf16_t*)d, ld_d, accumulate);
    RV_CHECK_LAUNCH("ew_mask_grad_kernel");
    return 0;
}

extern "C" int64_t rv_bn_bwd_smallk_workspace_bytes(int64_t pixels, int32_t c, int32_t cin) {
    const int CIN = cin <= 4 ? 4 : 8;
    const int64_t rows = rv_bn_bwd_rows(pixels);
    const int64_t planes = 2 + CIN, mcols = CIN + CIN * CIN;
    // partial rows + fp64 group scratch for both reductions
    int64_t bytes = ((rows + 2 * 64) * planes * c + (1024 + 2 * 64) * mcols) * (int64_t)sizeof(float) + 256;
    bytes = (bytes + 7) & ~(int64_t)7;
    return bytes + (planes * c + 80) * 8;  // + the reduced sums and moments (fp64) of the single-call form
}

// phase A: local sums.  sums = (2 + CIN) * c doubles (planes S0, S1, R[.][d]), moms = CIN + CIN*CIN doubles.
extern "C" int rv_bn_bwd_smallk_sums(int64_t pixels, int32_t c, const void* dout, int32_t ld_dout, const void* out, int32_t ld_out,
                                     const void* y, int32_t ld_y, const float* scale, const float* shift, const float* mean,
                                     const float* invstd, int32_t flags, const void* v, int32_t ld_v, int32_t cin, const void* w_packed,
                                     int32_t ld_w, double* sums, double* moms, void* workspace, rvStream stream) {
    BnbArgs a;
    if (fill(&a, pixels, c, dout, ld_dout, out, ld_out, y, ld_y, scale, shift, mean, invstd, flags)) return 1;
    RV_REQUIRE(v && sums && moms && workspace, "rv_bn_bwd_smallk_sums: null argument");
    const bool recomp = (flags & RV_BNB_Y_FROM_INPUT) != 0;
    RV_REQUIRE(!recomp || (w_packed && ld_w >= cin), "rv_bn_bwd_smallk_sums: RV_BNB_Y_FROM_INPUT needs the packed weight");
    const bf16_t* wq = (const bf16_t*)w_packed;
    RV_REQUIRE(cin >= 1 && cin <= 8 && ld_v % 8 == 0 && ld_v >= 8, "rv_bn_bwd_smallk: 1 <= cin <= 8, input rows of at least 8 channels");
    const int CIN = cin <= 4 ? 4 : 8;
    const int rows = rv_bn_bwd_rows(pixels), planes = 2 + CIN, mcols = CIN + CIN * CIN;
    float* part_g = (float*)workspace;
    double* scr_g = (double*)(part_g + (int64_t)rows * planes * c);
    float* part_m = (float*)(scr_g + (int64_t)64 * planes * c);
    const int mblocks = pixels >= 1024 * 256 ? 1024 : (int)((pixels + 255) / 256);
    double* scr_m = (double*)(part_m + (int64_t)1024 * mcols);
    hipStream_t st = (hipStream_t)stream;
    a.partial = part_g;
    if (CIN == 4) {
        if (recomp) hipLaunchKernelGGL((bn_bwd_smallk_reduce_kernel<4, true>), dim3(rows), dim3(256), 0, st, a, (const bf16_t*)v, ld_v, wq, ld_w);
        else hipLaunchKernelGGL((bn_bwd_smallk_reduce_kernel<4, false>), dim3(rows), dim3(256), 0, st, a, (const bf16_t*)v, ld_v, wq, ld_w);
        hipLaunchKernelGGL(smallk_moments_kernel<4>, dim3(mblocks), dim3(256), 0, st, (const bf16_t*)v, ld_v, pixels, part_m);
    } else {
        if (recomp) hipLaunchKernelGGL((bn_bwd_smallk_reduce_kernel<8, true>), dim3(rows), dim3(256), 0, st, a, (const bf16_t*)v, ld_v, wq, ld_w);
        else hipLaunchKernelGGL((bn_bwd_smallk_reduce_kernel<8, false>), dim3(rows), dim3(256), 0, st, a, (const bf16_t*)v, ld_v, wq, ld_w);
        hipLaunchKernelGGL(smallk_moments_kernel<8>, dim3(mblocks), dim3(256), 0, st, (const bf16_t*)v, ld_v, pixels, part_m);
    }
    RV_CHECK_LAUNCH("bn_bwd_smallk reduce kernels");
    int groups_g, groups_m;
    if (rv_col_reduce(part_g, rows, planes * c, scr_g, &groups_g, st)) return 1;
    if (rv_col_reduce(part_m, mblocks, mcols, scr_m, &groups_m, st)) return 1;
    hipLaunchKernelGGL(sum_rows_f64_kernel, dim3(rv_ceil_div(planes * c, 128)), dim3(128), 0, st, scr_g, groups_g, planes * c, sums);
    hipLaunchKernelGGL(sum_rows_f64_kernel, dim3(1), dim3(128), 0, st, scr_m, groups_m, mcols, moms);
    RV_CHECK_LAUNCH("sum_rows_f64_kernel");
    return 0;
}

// The same phase A for the first positional layer of the MetaKernel stem when its output gradient does not exist in memory:
// dOut = dy2 W2 is formed on the fly by pos_bwd_kernel (posconv.hip), which writes the per-workgroup rows of the planes.
int rv_pos_bwd_launch(int64_t pixels, const void* dy2, const void* w2_scatter, const void* rel, int32_t ld_rel, int32_t cin,
                      const void* w1_packed, int32_t ld_w1, const float* scale1, const float* shift1, const float* mean1,
                      const float* invstd1, float* partial, int32_t planes, int32_t max_rows, int32_t* rows, int32_t c, hipStream_t stream);

extern "C" int rv_pos_backward_sums(int64_t pixels, int32_t c, const void* dy2, const void* w2_scatter, const void* rel, int32_t ld_rel,
                                    int32_t cin, const void* w1_packed, int32_t ld_w1, const float* scale1, const float* shift1,
                                    const float* mean1, const float* invstd1, double* sums, double* moms, void* workspace,
                                    rvStream stream) {
    RV_REQUIRE(dy2 && w2_scatter && rel && w1_packed && scale1 && shift1 && mean1 && invstd1 && sums && moms && workspace,
               "rv_pos_backward_sums: null argument");
    RV_REQUIRE((c == 256 || c == 128) && cin >= 1 && cin <= 3 && ld_rel % 8 == 0 && ld_rel >= 8 && pixels > 0,
               "rv_pos_backward_sums: built for 256 or 128 channels, cin <= 3");
    const int CIN = 4, planes = 2 + CIN, mcols = CIN + CIN * CIN;
    const int rows_max = rv_bn_bwd_rows(pixels);  // the workspace layout of rv_bn_bwd_smallk_workspace_bytes
    float* part_g = (float*)workspace;
    double* scr_g = (double*)(part_g + (int64_t)rows_max * planes * c);
    float* part_m = (float*)(scr_g + (int64_t)64 * planes * c);
    const int mblocks = pixels >= 1024 * 256 ? 1024 : (int)((pixels + 255) / 256);
    double* scr_m = (double*)(part_m + (int64_t)1024 * mcols);
    hipStream_t st = (hipStream_t)stream;
    int rows = 0;
    if (rv_pos_bwd_launch(pixels, dy2, w2_scatter, rel, ld_rel, cin, w1_packed, ld_w1, scale1, shift1, mean1, invstd1, part_g, planes, rows_max,
                          &rows, c, st))
        return 1;
    hipLaunchKernelGGL(smallk_moments_kernel<4>, dim3(mblocks), dim3(256), 0, st, (const bf16_t*)rel, ld_rel, pixels, part_m);
    RV_CHECK_LAUNCH("smallk_moments_kernel");
    int groups_g, groups_m;
    if (rv_col_reduce(part_g, rows, planes * c, scr_g, &groups_g, st)) return 1;
    if (rv_col_reduce(part_m, mblocks, mcols, scr_m, &groups_m, st)) return 1;
    hipLaunchKernelGGL(sum_rows_f64_kernel, dim3(rv_ceil_div(planes * c, 128)), dim3(128), 0, st, scr_g, groups_g, planes * c, sums);
    hipLaunchKernelGGL(sum_rows_f64_kernel, dim3(1), dim3(128), 0, st, scr_m, groups_m, mcols, moms);
    RV_CHECK_LAUNCH("sum_rows_f64_kernel");
    return 0;
}

// phase B: gradients from the sums.  global_s01 (2 * c doubles, optional): all-reduced (sum g, sum g*xhat) under SyncBN.
extern "C" int rv_bn_bwd_smallk_from_sums(int32_t c, int32_t cin, const double* sums, const double* moms, const double* global_s01,
                                          const void* w_packed, int32_t ld_w, const float* gamma, const float* stat_mean,
                                          const float* stat_invstd, int64_t count, float* dgamma, float* dbeta, float* dW,
                                          rvStream stream) {
    RV_REQUIRE(sums && moms && w_packed && gamma && stat_mean && stat_invstd && dgamma && dbeta && dW, "rv_bn_bwd_smallk_from_sums: null argument");
    RV_REQUIRE(cin >= 1 && cin <= 8 && (count > 0 || global_s01), "rv_bn_bwd_smallk_from_sums: bad shape (count < 0 needs global_s01)");
    hipStream_t st = (hipStream_t)stream;
    // count < 0 (SyncBN): the global pixel count is the double behind the all-reduced sums, global_s01[2 c]
    const double* count_dev = count < 0 ? global_s01 + 2 * (int64_t)c : nullptr;
    const double inv = count > 0 ? 1.0 / (double)count : 0.0;
    if (cin <= 4)
        hipLaunchKernelGGL(bn_bwd_smallk_finalize_kernel<4>, dim3(rv_ceil_div(c, 64)), dim3(64), 0, st, sums, 1, moms, 1, c, cin, inv,
                           gamma, stat_mean, stat_invstd, (const bf16_t*)w_packed, ld_w, global_s01, count_dev, dgamma, dbeta, dW);
    else
        hipLaunchKernelGGL(bn_bwd_smallk_finalize_kernel<8>, dim3(rv_ceil_div(c, 64)), dim3(64), 0, st, sums, 1, moms, 1, c, cin, inv,
                           gamma, stat_mean, stat_invstd, (const bf16_t*)w_packed, ld_w, global_s01, count_dev, dgamma, dbeta, dW);
    RV_CHECK_LAUNCH("bn_bwd_smallk_finalize_kernel");
    return 0;
}

extern "C" int rv_bn_bwd_smallk(int64_t pixels, int32_t c, const void* dout, int32_t ld_dout, const void* out, int32_t ld_out,
                                const void* y, int32_t ld_y, const float* scale, const float* shift, const float* mean,
                                const float* invstd, int32_t flags, const void* v, int32_t ld_v, int32_t cin, const void* w_packed,
                                int32_t ld_w, const float* gamma, const float* stat_mean, const float* stat_invstd, int64_t count,
                                float* dgamma, float* dbeta, float* dW, void* workspace, rvStream stream) {
    // (mean, invstd) drive the per-pixel xhat = (y - mean) * invstd; (stat_mean, stat_invstd) are the layer's batch
    // statistics used in the closed-form terms.  They differ when y is the ACTIVATED output h = relu(gamma*xhat + beta)
    // (rv_smallk_forward): then the caller passes scale = 1, shift = 0, mean = beta, invstd = 1/gamma.
    RV_REQUIRE(workspace, "rv_bn_bwd_smallk: null workspace");
    const int CIN = cin <= 4 ? 4 : 8;
    double* sums = (double*)((uint8_t*)workspace + rv_bn_bwd_smallk_workspace_bytes(pixels, c, cin) - (int64_t)((2 + CIN) * c + 80) * 8);
    double* moms = sums + (int64_t)(2 + CIN) * c;
    if (rv_bn_bwd_smallk_sums(pixels, c, dout, ld_dout, out, ld_out, y, ld_y, scale, shift, mean, invstd, flags, v, ld_v, cin, w_packed, ld_w,
                              sums, moms, workspace, stream))
        return 1;
    return rv_bn_bwd_smallk_from_sums(c, cin, sums, moms, nullptr, w_packed, ld_w, gamma, stat_mean ? stat_mean : mean,
                                      stat_invstd ? stat_invstd : invstd, count, dgamma, dbeta, dW, stream);
}

extern "C" int64_t rv_smallk_forward_workspace_bytes(int32_t cin) {
    const int CIN = cin <= 4 ? 4 : 8;
    return (int64_t)(1024 + 2 * 64) * (CIN + CIN * CIN) * (int64_t)sizeof(float) + 256;
}

/* moments of v (device, fp64 group rows in the workspace); groups returned through *groups_m */
static int smallk_moments(const void* v, int32_t ld_v, int64_t pixels, int CIN, void* workspace, double** scr_m, int* groups_m,
                          hipStream_t st) {
    const int mcols = CIN + CIN * CIN;
    float* part_m = (float*)workspace;
    const int mblocks = pixels >= 1024 * 256 ? 1024 : (int)((pixels + 255) / 256);
    *scr_m = (double*)(part_m + (int64_t)1024 * mcols);
    if (CIN == 4)
        hipLaunchKernelGGL(smallk_moments_kernel<4>, dim3(mblocks), dim3(256), 0, st, (const bf16_t*)v, ld_v, pixels, part_m);
    else
        hipLaunchKernelGGL(smallk_moments_kernel<8>, dim3(mblocks), dim3(256), 0, st, (const bf16_t*)v, ld_v, pixels, part_m);
    RV_CHECK_LAUNCH("smallk_moments_kernel");
    return rv_col_reduce(part_m, mblocks, mcols, *scr_m, groups_m, st);
}

extern "C" int rv_smallk_moments(const void* v, int32_t ld_v, int64_t pixels, int32_t cin, double* moments, void* workspace,
                                 rvStream stream) {
    RV_REQUIRE(v && moments && workspace && cin >= 1 && cin <= 8 && ld_v % 8 == 0 && ld_v >= 8, "rv_smallk_moments: bad argument");
    const int CIN = cin <= 4 ? 4 : 8;
    double* scr;
    int groups;
    if (smallk_moments(v, ld_v, pixels, CIN, workspace, &scr, &groups, (hipStream_t)stream)) return 1;
    // sum the (at most 64) group rows into `moments` (CIN + CIN*CIN doubles): reuse the column reducer's layout
    hipLaunchKernelGGL(sum_rows_f64_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, scr, groups, CIN + CIN * CIN, moments);
    RV_CHECK_LAUNCH("sum_rows_f64_kernel");
    return 0;
}

extern "C" int rv_smallk_forward(const void* v, int32_t ld_v, int64_t pixels, int32_t cin, const void* w_packed, int32_t ld_w,
                                 int32_t c, const double* moments, int64_t count, const float* gamma, const float* beta, float eps,
                                 float momentum, float* running_mean, float* running_var, float* scale, float* shift, float* mean,
                                 float* invstd, int32_t relu, void* h, int32_t ld_h, rvStream stream) {
    RV_REQUIRE(v && w_packed && scale && shift, "rv_smallk_forward: null argument");  // h == NULL: statistics only (rv_pos_forward applies)
    RV_REQUIRE(cin >= 1 && cin <= 8 && ld_v % 8 == 0 && ld_v >= 8 && c % 8 == 0 && c / 8 <= 256 && ld_h % 8 == 0, "rv_smallk_forward: bad shape");
    const int CIN = cin <= 4 ? 4 : 8;
    hipStream_t st = (hipStream_t)stream;
    if (moments) {  // training: batch statistics in closed form from the moments (already all-reduced by the caller under SyncBN)
        RV_REQUIRE(gamma && beta && count != 0, "rv_smallk_forward: statistics need gamma, beta and the pixel count");
        // count < 0 (SyncBN): the all-reduced pixel count is the double behind the moments, moments[cin_pad + cin_pad^2]
        const double* count_dev = count < 0 ? moments + (CIN + CIN * CIN) : nullptr;
        const double unbias = count > 1 ? (double)count / (double)(count - 1) : 1.0;
        const double inv = count > 0 ? 1.0 / (double)count : 0.0;
        if (CIN == 4)
            hipLaunchKernelGGL(smallk_stats_kernel<4>, dim3(rv_ceil_div(c, 64)), dim3(64), 0, st, moments, 1, c, cin, (const bf16_t*)w_packed, ld_w,
                               inv, unbias, count_dev, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean, invstd);
        else
            hipLaunchKernelGGL(smallk_stats_kernel<8>, dim3(rv_ceil_div(c, 64)), dim3(64), 0, st, moments, 1, c, cin, (const bf16_t*)w_packed, ld_w,
                               inv, unbias, count_dev, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean, invstd);
        RV_CHECK_LAUNCH("smallk_stats_kernel");
    }
    if (!h) return 0;
    const int c8 = c / 8, lanes_px = 256 / c8;
    int64_t blocks = (pixels + lanes_px - 1) / lanes_px;
    if (blocks > 4096) blocks = 4096;
    if (CIN == 4)
        hipLaunchKernelGGL(smallk_apply_kernel<4>, dim3((int)blocks), dim3(256), 0, st, (const bf16_t*)v, ld_v, pixels, c8, (const bf16_t*)w_packed, ld_w,
                           scale, shift, relu, (bf16_t*)h, ld_h);
    else
        hipLaunchKernelGGL(smallk_apply_kernel<8>, dim3((int)blocks), dim3(256), 0, st, (const bf16_t*)v, ld_v, pixels, c8, (const bf16_t*)w_packed, ld_w,
                           scale, shift, relu, (bf16_t*)h, ld_h);
    RV_CHECK_LAUNCH("smallk_apply_kernel");
    return 0;
}

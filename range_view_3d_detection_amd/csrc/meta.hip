// meta.hip -- MetaKernel stem glue (nn/stems/__init__.py:64-85): the reference's two F.unfold
// calls and the (B, C, 9, H*W) element-wise product, as gather kernels over NHWC bf16 tensors.
// Pure HBM-bound byte movers: one 16-byte channel octet per thread, consecutive threads on
// consecutive octets (coalesced 512-byte pixel rows for C = 256).
//
// Tap order follows F.unfold: k = ky*3 + kx, neighbour = (h + ky - 1, w + kx - 1), zero padded.
#include <stdlib.h>

#include "common.h"

namespace {

__global__ void meta_relative_kernel(const float* cart, int N, int H, int W, bf16_t* rel) {
    const int64_t hw = (int64_t)H * W;
    const int64_t total = (int64_t)N * hw * 9;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(i % 9);
        const int64_t p = i / 9;  // pixel index n*H*W + h*W + w
        const int64_t n = p / hw;
        const int64_t r = p - n * hw;
        const int h = (int)(r / W), w = (int)(r - (int64_t)h * W);
        const int hn = h + k / 3 - 1, wn = w + k % 3 - 1;
        const float* c0 = cart + n * 3 * hw;
        float v[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const float centre = c0[j * hw + r];
            const float nb = (hn >= 0 && hn < H && wn >= 0 && wn < W) ? c0[j * hw + (int64_t)hn * W + wn] : 0.f;
            v[j] = nb - centre;
        }
        u32x4* dst = (u32x4*)(rel + i * 32);
        dst[0] = u32x4{pack_bf2(v[0], v[1]), pack_bf2(v[2], 0.f), 0u, 0u};
        dst[1] = u32x4{0u, 0u, 0u, 0u};
        dst[2] = u32x4{0u, 0u, 0u, 0u};
        dst[3] = u32x4{0u, 0u, 0u, 0u};
    }
}

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

// Work order of the two 9x-grid gather kernels.  Workgroups are dealt round-robin over the 8 XCDs (one L2 each); every pixel
// reads the feature rows h-1, h, h+1 of its 3 x 3 neighbourhood.  XCD x therefore owns a COLUMN STRIP of the image
// (ceil(W/8) columns, all rows, row after row): the three feature rows of its strip (3 x 128 KB at W = 2048, C = 256) stay in its
// L2 while the 2.4 GB position / output tensors stream through, instead of every L2 fetching every feature row nine times.
struct StripItem {
    int64_t p;      // pixel index n*H*W + h*W + w
    int64_t n, r;   // image, h*W + w
    int h, w, k, oc;
    bool ok;
};
__device__ __forceinline__ StripItem strip_item(int64_t q, int xcd, int N, int H, int W, int c8) {
    const int ws = (W + 7) / 8;
    StripItem it;
    it.oc = (int)(q % c8);
    q /= c8;
    it.k = (int)(q % 9);
    q /= 9;
    const int wl = (int)(q % ws);
    const int64_t row = q / ws;  // n*H + h
    it.w = xcd * ws + wl;
    it.n = row / H;
    it.h = (int)(row - it.n * H);
    it.ok = row < (int64_t)N * H && it.w < W;
    it.r = (int64_t)it.h * W + it.w;
    it.p = it.n * H * W + it.r;
    return it;
}

// Items of an XCD's column strip in (row, column, tap) order are consecutive in memory within a row segment.  A workgroup
// takes PIECES of kPieceItems consecutive items (64 KB at C = 256: long contiguous runs read + write ~15 % faster than
// 4 KB pieces, profiles/r02_hbm_kernels.md) round-robin over the strip, so that the workgroups of an XCD still sweep the
// strip row after row together and the three feature rows they need stay in its L2.
constexpr int kPieceItems = 128;

struct MetaItem {
    int64_t pk;   // (pixel, tap) index into the 9x-grid tensors; -1: the item does not exist (column past the image)
    int64_t nbr;  // pixel index of the tap's neighbour, -1: outside the image (zero padding)
};
__device__ __forceinline__ MetaItem meta_item(int64_t it, int xcd, int ws, int H, int W) {
    MetaItem m;
    const int k = (int)(it % 9);
    const int64_t t = it / 9;
    const int w = xcd * ws + (int)(t % ws);
    const int64_t row = t / ws, n = row / H;
    const int h = (int)(row - n * H);
    m.pk = w < W ? ((n * H + h) * (int64_t)W + w) * 9 + k : -1;
    const int hn = h + k / 3 - 1, wn = w + k % 3 - 1;
    m.nbr = (w < W && hn >= 0 && hn < H && wn >= 0 && wn < W) ? (n * H + hn) * (int64_t)W + wn : -1;
    return m;
}

// geo[p][k*C + c] = relu(scale*pos[p*9+k][c] + shift) * feat[nbr_k(p)][c];  thread = (item lane, channel octet)
__global__ __launch_bounds__(256) void meta_modulate_kernel(const bf16_t* pos, const float* scale, const float* shift,
                                                            const bf16_t* feat, int ld_feat, int N, int H, int W, int C,
                                                            bf16_t* geo, int piece) {
    const int c8 = C / 8;
    const int lanes = 256 / c8;
    const int oct = threadIdx.x % c8, pl = threadIdx.x / c8;
    if (pl >= lanes) return;
    const int c0 = oct * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = scale[c0 + j];
        sh[j] = shift[c0 + j];
    }
    const int xcd = blockIdx.x & 7, ws = (W + 7) / 8;
    const int64_t items = (int64_t)N * H * ws * 9, pieces = (items + piece - 1) / piece;
    for (int64_t pc = blockIdx.x >> 3; pc < pieces; pc += gridDim.x >> 3) {
        const int64_t hi = (pc + 1) * piece < items ? (pc + 1) * piece : items;
        // two items per thread in flight (four 16-byte loads): one item's two loads per thread did not cover the HBM latency
        for (int64_t it = pc * piece + pl; it < hi; it += 2 * lanes) {
            MetaItem m[2];
            u32x4 av[2], fv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                m[u] = meta_item(it + u * lanes, xcd, ws, H, W);
                if (it + u * lanes >= hi) m[u].pk = -1;
                av[u] = fv[u] = u32x4{0u, 0u, 0u, 0u};
                if (m[u].pk >= 0 && m[u].nbr >= 0) {
                    av[u] = *(const u32x4*)(pos + m[u].pk * C + c0);
                    fv[u] = *(const u32x4*)(feat + m[u].nbr * ld_feat + c0);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (m[u].pk < 0) continue;
                float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (m[u].nbr >= 0) {
                    float a[8], f[8];
                    unpack8(av[u], a);
                    unpack8(fv[u], f);
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = fmaxf(a[j] * sc[j] + sh[j], 0.f) * f[j];
                }
                *(u32x4*)(geo + m[u].pk * C + c0) = pack8(o);
            }
        }
    }
}

// dpos_act[p*9+k][c] = dgeo[p][k*C+c] * feat[nbr_k(p)][c]
__global__ __launch_bounds__(256) void meta_modulate_bwd_pos_kernel(const bf16_t* dgeo, const bf16_t* feat, int ld_feat,
                                                                    int N, int H, int W, int C, bf16_t* dpos) {
    const int c8 = C / 8;
    const int64_t hw = (int64_t)H * W;
    const int xcd = blockIdx.x & 7;
    const int64_t per_xcd = (int64_t)N * H * ((W + 7) / 8) * 9 * c8, stride = (int64_t)(gridDim.x >> 3) * blockDim.x;
    for (int64_t q = (int64_t)(blockIdx.x >> 3) * blockDim.x + threadIdx.x; q < per_xcd; q += stride) {
        const StripItem it = strip_item(q, xcd, N, H, W, c8);
        if (!it.ok) continue;
        const int oc = it.oc, k = it.k, h = it.h, w = it.w;
        const int64_t n = it.n, pk = it.p * 9 + k;
        const int hn = h + k / 3 - 1, wn = w + k % 3 - 1;
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (hn >= 0 && hn < H && wn >= 0 && wn < W) {
            float g[8], f[8];
            unpack8(*(const u32x4*)(dgeo + pk * C + oc * 8), g);
            unpack8(*(const u32x4*)(feat + (n * hw + (int64_t)hn * W + wn) * ld_feat + oc * 8), f);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = g[j] * f[j];
        }
        *(u32x4*)(dpos + pk * C + oc * 8) = pack8(o);
    }
}

// dfeat[q][c] = sum_k dgeo[p_k][k*C+c] * relu(scale*pos[p_k*9+k][c]+shift),  p_k = q - offset_k (the pixel whose
// k-th neighbour is q)
__global__ __launch_bounds__(256) void meta_modulate_bwd_feat_kernel(const bf16_t* dgeo, const bf16_t* pos,
                                                                     const float* scale, const float* shift, int N, int H,
                                                                     int W, int C, bf16_t* dfeat, int ld_dfeat) {
    const int c8 = C / 8;
    const int64_t hw = (int64_t)H * W;
    const int64_t total = (int64_t)N * hw * c8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int oc = (int)(i % c8);
        const int64_t q = i / c8;
        const int64_t n = q / hw;
        const int64_t r = q - n * hw;
        const int h = (int)(r / W), w = (int)(r - (int64_t)h * W);
        float sc[8], sh[8], acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc[j] = scale[oc * 8 + j];
            sh[j] = shift[oc * 8 + j];
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int hp = h - (k / 3 - 1), wp = w - (k % 3 - 1);
            if (hp < 0 || hp >= H || wp < 0 || wp >= W) continue;
            const int64_t pk = (n * hw + (int64_t)hp * W + wp) * 9 + k;
            float g[8], a[8];
            unpack8(*(const u32x4*)(dgeo + pk * C + oc * 8), g);
            unpack8(*(const u32x4*)(pos + pk * C + oc * 8), a);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += g[j] * fmaxf(a[j] * sc[j] + sh[j], 0.f);
        }
        *(u32x4*)(dfeat + q * ld_dfeat + oc * 8) = pack8(acc);
    }
}

// -----------------------------------------------------------------------------------------------------------------
// Backward of the modulation FUSED with the BatchNorm(+ReLU) backward of the positional layer that feeds it.
//   P = relu(scale*y + shift),  geo[p][k] = P[p][k] * feat[nbr_k(p)]
//   z[p][k]  = dgeo[p][k] * feat[nbr_k(p)] * [P > 0]                 gradient w.r.t. the BatchNorm output
//   dfeat[q] = sum_k dgeo[q - off_k][k] * P[q - off_k][k]
//   dy[p][k] = k0 (z - S0/n - xhat S1/n),  S0 = sum z, S1 = sum z*xhat, xhat = (y - mean) invstd
// Two passes over (dgeo, y) instead of the four the unfused chain made (modulate-backward x2, BatchNorm reduce, apply),
// and z is never written: 5 instead of 9 transfers of the 9x-grid tensor (2.4 GB each at 4 x 64 x 2048 x 256).
// Pass 1 walks TARGET pixels q: every (p, k) whose neighbour lies inside the image is the k-th source of exactly one q, so
// one thread sees dgeo/y of that pair once, multiplies by ITS OWN feat[q] (no neighbour gather at all) for S0/S1 and by P
// for dfeat.  Pairs whose neighbour is outside the image have z = 0 and contribute to neither.
// Thread = (pixel lane, channel octet) as in bnbwd.hip: per-channel constants stay in registers.
// -----------------------------------------------------------------------------------------------------------------
constexpr int kMetaPixPerBlock = 512;  // 1024 partial rows at 4 x 64 x 2048: the single-launch finalize takes them

struct MetaBwdArgs {
    const bf16_t *dgeo, *y, *feat;
    const float *scale, *shift, *mean, *invstd, *coef;
    int N, H, W, C, c8, ld_feat, ld_dfeat, piece;
    bf16_t* dfeat;
    float* partial;
    bf16_t* dy;
};

__global__ __launch_bounds__(256) void meta_bwd_sums_kernel(const MetaBwdArgs a) {
    __shared__ float red[256][17];
    const int tid = threadIdx.x;
    const int lanes_px = 256 / a.c8;
    const int oct = tid % a.c8, pl = tid / a.c8;
    const bool active = pl < lanes_px;
    const int c0 = oct * 8, C = a.C, H = a.H, W = a.W;
    const int64_t hw = (int64_t)H * W, pixels = (int64_t)a.N * hw;
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
    const int64_t p0 = (int64_t)blockIdx.x * kMetaPixPerBlock;
    const int64_t p1 = p0 + kMetaPixPerBlock < pixels ? p0 + kMetaPixPerBlock : pixels;
    if (active) {
        for (int64_t q = p0 + pl; q < p1; q += lanes_px) {
            const int64_t n = q / hw, r = q - n * hw;
            const int h = (int)(r / W), w = (int)(r - (int64_t)h * W);
            u32x4 gv[9], yv[9];
            // all 18 loads of the target in flight before the first use (sources outside the image: zeros)
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int hp = h - (k / 3 - 1), wp = w - (k % 3 - 1);
                const bool in = hp >= 0 && hp < H && wp >= 0 && wp < W;
                const int64_t pk = in ? ((q - (int64_t)(k / 3 - 1) * W - (k % 3 - 1)) * 9 + k) : 0;
                gv[k] = *(const u32x4*)(a.dgeo + pk * C + c0);
                yv[k] = *(const u32x4*)(a.y + pk * C + c0);
                if (!in) gv[k] = u32x4{0u, 0u, 0u, 0u};
            }
            float f[8], acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            unpack8(*(const u32x4*)(a.feat + q * a.ld_feat + c0), f);
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                float g[8], y[8];
                unpack8(gv[k], g);
                unpack8(yv[k], y);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float act = y[j] * sc[j] + sh[j];
                    const float gp = act > 0.f ? g[j] : 0.f;  // dgeo where the ReLU passed
                    acc[j] += gp * act;
                    const float z = gp * f[j];
                    s0[j] += z;
                    s1[j] += z * ((y[j] - mu[j]) * is[j]);
                }
            }
            *(u32x4*)(a.dfeat + q * a.ld_dfeat + c0) = pack8(acc);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[tid][j] = s0[j];
        red[tid][8 + j] = s1[j];
    }
    __syncthreads();
    for (int i = tid; i < a.c8 * 16; i += 256) {
        const int o = i / 16, j = i - o * 16;
        float s = 0.f;
        for (int l = 0; l < lanes_px; ++l) s += red[l * a.c8 + o][j];
        a.partial[((int64_t)blockIdx.x * 2 + (j >> 3)) * C + o * 8 + (j & 7)] = s;
    }
}

// Pass 2: dy for every (p, k), in the XCD column-strip order of the forward gather (feat rows stay in the XCD's L2).
__global__ __launch_bounds__(256) void meta_bwd_apply_kernel(const MetaBwdArgs a) {
    const int tid = threadIdx.x;
    const int lanes_px = 256 / a.c8;
    const int oct = tid % a.c8, pl = tid / a.c8;
    if (pl >= lanes_px) return;
    const int c0 = oct * 8, C = a.C, H = a.H, W = a.W;
    float sc[8], sh[8], mu[8], is[8], k0[8], k1[8], k2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        sc[j] = a.scale[c0 + j];
        sh[j] = a.shift[c0 + j];
        mu[j] = a.mean[c0 + j];
        is[j] = a.invstd[c0 + j];
        k0[j] = a.coef[c0 + j];
        k1[j] = a.coef[C + c0 + j];
        k2[j] = a.coef[2 * C + c0 + j];
    }
    const int xcd = blockIdx.x & 7, ws = (W + 7) / 8;
    const int64_t items = (int64_t)a.N * H * ws * 9, pieces = (items + a.piece - 1) / a.piece;
    for (int64_t pc = blockIdx.x >> 3; pc < pieces; pc += gridDim.x >> 3) {
        const int64_t hi = (pc + 1) * a.piece < items ? (pc + 1) * a.piece : items;
        for (int64_t it = pc * a.piece + pl; it < hi; it += 2 * lanes_px) {  // two items (six 16-byte loads) in flight per thread
            MetaItem m[2];
            u32x4 gv[2], yv[2], fv[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                m[u] = meta_item(it + u * lanes_px, xcd, ws, H, W);
                if (it + u * lanes_px >= hi) m[u].pk = -1;
                gv[u] = yv[u] = fv[u] = u32x4{0u, 0u, 0u, 0u};
                if (m[u].pk >= 0) {
                    gv[u] = *(const u32x4*)(a.dgeo + m[u].pk * C + c0);
                    yv[u] = *(const u32x4*)(a.y + m[u].pk * C + c0);
                    if (m[u].nbr >= 0) fv[u] = *(const u32x4*)(a.feat + m[u].nbr * a.ld_feat + c0);
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                if (m[u].pk < 0) continue;
                const bool in = m[u].nbr >= 0;
                float g[8], y[8], f[8], o[8];
                unpack8(gv[u], g);
                unpack8(yv[u], y);
                unpack8(fv[u], f);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float z = (in && y[j] * sc[j] + sh[j] > 0.f) ? g[j] * f[j] : 0.f;
                    o[j] = k0[j] * (z - k1[j] - (y[j] - mu[j]) * is[j] * k2[j]);
                }
                *(u32x4*)(a.dy + m[u].pk * C + c0) = pack8(o);
            }
        }
    }
}

constexpr int meta_piece() { return kPieceItems; }  // items per contiguous piece

int grid_for(int64_t work) {
    int64_t b = (work + 255) / 256;
    b = (b + 7) & ~(int64_t)7;  // (the strip order deals whole rounds of 8 workgroups)
    return (int)(b < 8 ? 8 : (b > 256 * 16 ? 256 * 16 : b));
}

}  // namespace

extern "C" int rv_meta_relative(const float* cart_nchw, int32_t N, int32_t H, int32_t W, void* rel, rvStream stream) {
    RV_REQUIRE(cart_nchw && rel, "rv_meta_relative: null argument");
    hipLaunchKernelGGL(meta_relative_kernel, dim3(grid_for((int64_t)N * H * W * 9)), dim3(256), 0, (hipStream_t)stream,
                       cart_nchw, N, H, W, (bf16_t*)rel);
    RV_CHECK_LAUNCH("meta_relative_kernel");
    return 0;
}

extern "C" int rv_meta_modulate(const void* pos_raw, const float* scale, const float* shift, const void* feat,
                                int32_t ld_feat, int32_t N, int32_t H, int32_t W, int32_t C, void* geo, rvStream stream) {
    RV_REQUIRE(pos_raw && scale && shift && feat && geo, "rv_meta_modulate: null argument");
    RV_REQUIRE(C % 8 == 0 && ld_feat % 8 == 0, "rv_meta_modulate: channels must be a multiple of 8");
    RV_REQUIRE(C <= 2048, "rv_meta_modulate: at most 2048 channels");
    hipLaunchKernelGGL(meta_modulate_kernel, dim3(grid_for((int64_t)N * H * W * 9 * (C / 8))), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t*)pos_raw, scale, shift, (const bf16_t*)feat, ld_feat, N, H, W, C,
                       (bf16_t*)geo, meta_piece());
    RV_CHECK_LAUNCH("meta_modulate_kernel");
    return 0;
}

extern "C" int rv_meta_modulate_bwd(const void* dgeo, const void* pos_raw, const float* scale, const float* shift,
                                    const void* feat, int32_t ld_feat, int32_t N, int32_t H, int32_t W, int32_t C,
                                    void* dpos_act, void* dfeat, int32_t ld_dfeat, rvStream stream) {
    RV_REQUIRE(dgeo && pos_raw && scale && shift && feat && dpos_act && dfeat, "rv_meta_modulate_bwd: null argument");
    RV_REQUIRE(C % 8 == 0 && ld_feat % 8 == 0 && ld_dfeat % 8 == 0, "rv_meta_modulate_bwd: channels must be a multiple of 8");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(meta_modulate_bwd_pos_kernel, dim3(grid_for((int64_t)N * H * W * 9 * (C / 8))), dim3(256), 0, st,
                       (const bf16_t*)dgeo, (const bf16_t*)feat, ld_feat, N, H, W, C, (bf16_t*)dpos_act);
    hipLaunchKernelGGL(meta_modulate_bwd_feat_kernel, dim3(grid_for((int64_t)N * H * W * (C / 8))), dim3(256), 0, st,
                       (const bf16_t*)dgeo, (const bf16_t*)pos_raw, scale, shift, N, H, W, C, (bf16_t*)dfeat, ld_dfeat);
    RV_CHECK_LAUNCH("meta_modulate_bwd kernels");
    return 0;
}

extern "C" int32_t rv_meta_bwd_rows(int32_t N, int32_t H, int32_t W) {
    return (int32_t)(((int64_t)N * H * W + kMetaPixPerBlock - 1) / kMetaPixPerBlock);
}

extern "C" int rv_meta_modulate_bwd_sums(const void* dgeo, const void* pos_raw, const float* scale, const float* shift,
                                         const float* mean, const float* invstd, const void* feat, int32_t ld_feat, int32_t N,
                                         int32_t H, int32_t W, int32_t C, void* dfeat, int32_t ld_dfeat, float* partial,
                                         rvStream stream) {
    RV_REQUIRE(dgeo && pos_raw && scale && shift && mean && invstd && feat && dfeat && partial,
               "rv_meta_modulate_bwd_sums: null argument");
    RV_REQUIRE(C % 8 == 0 && C <= 2048 && ld_feat % 8 == 0 && ld_dfeat % 8 == 0,
               "rv_meta_modulate_bwd_sums: channels must be a multiple of 8 (at most 2048)");
    MetaBwdArgs a{};
    a.dgeo = (const bf16_t*)dgeo, a.y = (const bf16_t*)pos_raw, a.feat = (const bf16_t*)feat;
    a.scale = scale, a.shift = shift, a.mean = mean, a.invstd = invstd;
    a.N = N, a.H = H, a.W = W, a.C = C, a.c8 = C / 8, a.ld_feat = ld_feat, a.ld_dfeat = ld_dfeat;
    a.dfeat = (bf16_t*)dfeat, a.partial = partial;
    hipLaunchKernelGGL(meta_bwd_sums_kernel, dim3(rv_meta_bwd_rows(N, H, W)), dim3(256), 0, (hipStream_t)stream, a);
    RV_CHECK_LAUNCH("meta_bwd_sums_kernel");
    return 0;
}

extern "C" int rv_meta_modulate_bwd_apply(const void* dgeo, const void* pos_raw, const float* scale, const float* shift,
                                          const float* mean, const float* invstd, const float* coef, const void* feat,
                                          int32_t ld_feat, int32_t N, int32_t H, int32_t W, int32_t C, void* dy, rvStream stream) {
    RV_REQUIRE(dgeo && pos_raw && scale && shift && mean && invstd && coef && feat && dy, "rv_meta_modulate_bwd_apply: null argument");
    RV_REQUIRE(C % 8 == 0 && C <= 2048 && ld_feat % 8 == 0, "rv_meta_modulate_bwd_apply: channels must be a multiple of 8 (at most 2048)");
    MetaBwdArgs a{};
    a.dgeo = (const bf16_t*)dgeo, a.y = (const bf16_t*)pos_raw, a.feat = (const bf16_t*)feat;
    a.scale = scale, a.shift = shift, a.mean = mean, a.invstd = invstd, a.coef = coef;
    a.N = N, a.H = H, a.W = W, a.C = C, a.c8 = C / 8, a.ld_feat = ld_feat;
    a.dy = (bf16_t*)dy;
    a.piece = meta_piece();
    hipLaunchKernelGGL(meta_bwd_apply_kernel, dim3(grid_for((int64_t)N * H * W * 9 * (C / 8))), dim3(256), 0, (hipStream_t)stream, a);
    RV_CHECK_LAUNCH("meta_bwd_apply_kernel");
    return 0;
}

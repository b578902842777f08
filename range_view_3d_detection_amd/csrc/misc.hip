// misc.hip -- error state, weight packing, BatchNorm finalisation, element-wise combine and
// layout conversion kernels (all HBM-bound byte movers: coalesced 16-byte accesses, no MFMA).
#include <stdarg.h>

#include <stdlib.h>

#include "common.h"
#include "tapconv.h"

// ---------------------------------------------------------------------------------------------
// error state
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void rv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* rv_last_error(void) { return g_err; }
extern "C" int rv_version(void) { return 100; }
extern "C" int32_t rv_pad_channels(int32_t c) { return rv_pad32(c); }

// ---------------------------------------------------------------------------------------------
// weight packing
// ---------------------------------------------------------------------------------------------
namespace {

struct PackArgs {
    const float* T;
    bf16_t* out;
    int cu, cv, cu_pad, cv_pad, kh, kw;
    int scatter;
    TapTable tt;
    int phases;
    // FOLDED gather image of a stride-s layer (fold_s > 0): the fine tensor V is read as (N, H, W/s, s*cv_pad) -- the s fine
    // pixels of a coarse pixel are contiguous in NHWC -- which turns the stride-s gather into a STRIDE-1 one with
    // kw' = jmax - jmin + 1 column taps over s*cv_pad channels: tap (ky, j') x folded channel r*cv_pad + c holds
    // T[cu][c][ky][kx], kx = s (j' + jmin) + r + pad_w (zero where kx falls outside the kernel).  kh, kw, cv, cv_pad above then
    // describe the folded image; fold_cv / fold_kw are the original channel count and kernel width.
    int fold_s, fold_pw, fold_jmin, fold_cvp, fold_cv, fold_kw;
};

struct PackArgs2 {  // both forms in one launch: blockIdx.y selects
    PackArgs f[2];
};

// One thread per (u, v) channel pair: its kh*kw taps are contiguous in the torch layout (one run of reads), and every tap
// image receives consecutive 2-byte elements from consecutive threads (v fastest in the gather images, u in the scatter ones).
__device__ __forceinline__ void pack_one(const PackArgs& a, int64_t first, int64_t stride) {
    const int taps = a.kh * a.kw;
    const int64_t pairs = (int64_t)a.cu_pad * a.cv_pad;
    for (int64_t pair = first; pair < pairs; pair += stride) {
        int u, v;
        if (!a.scatter) {  // [tap][cu_pad][cv_pad]
            u = (int)(pair / a.cv_pad);
            v = (int)(pair - (int64_t)u * a.cv_pad);
        } else {  // [phase-major tap][cv_pad][cu_pad]
            v = (int)(pair / a.cu_pad);
            u = (int)(pair - (int64_t)v * a.cu_pad);
        }
        if (a.fold_s) {  // (gather image only)
            const int r = v / a.fold_cvp, c = v - r * a.fold_cvp;
            const bool in = u < a.cu && c < a.fold_cv && r < a.fold_s;
            const float* src = a.T + ((int64_t)u * a.fold_cv + c) * (a.kh * a.fold_kw);
            for (int t = 0; t < taps; ++t) {
                const int ky = t / a.kw, jp = t - ky * a.kw;
                const int kx = a.fold_s * (jp + a.fold_jmin) + r + a.fold_pw;
                a.out[(int64_t)t * pairs + pair] = f2bf(in && kx >= 0 && kx < a.fold_kw ? src[ky * a.fold_kw + kx] : 0.f);
            }
            continue;
        }
        const bool in = u < a.cu && v < a.cv;
        const float* src = a.T + ((int64_t)u * a.cv + v) * taps;
        if (!a.scatter) {
            for (int t = 0; t < taps; ++t) a.out[(int64_t)t * pairs + pair] = f2bf(in ? src[t] : 0.f);
        } else {
            int img = 0;
            for (int r = 0; r < a.phases; ++r)
                for (int idx = 0; idx < a.tt.ntaps[r]; ++idx, ++img)
                    a.out[(int64_t)img * pairs + pair] = f2bf(in ? src[a.tt.ky[r][idx] * a.kw + a.tt.kx[r][idx]] : 0.f);
        }
    }
}

// Scatter images are [tap][cv_pad][cu_pad] (u fastest) while the torch layout has v next to the taps: one thread per pair with
// u fastest reads a 36-byte run from a line of its own (64 lines per wave-instruction, 32x read amplification on a 3x3 layer).
// Instead a 256-thread workgroup moves 32 (u) x 32 (v) tiles through LDS: read with v fastest (a wave covers 32 x taps
// contiguous floats per u), written with u fastest.
constexpr int kPackTile = 32;
__device__ __forceinline__ void pack_scatter_tiles(const PackArgs& a, float (*tile)[kPackTile][kPackTile + 1], int first_tile, int tile_stride) {
    const int taps = a.kh * a.kw;
    const int tu = a.cu_pad / kPackTile, tv = a.cv_pad / kPackTile;  // (padded channel counts are multiples of 32)
    const int64_t pairs = (int64_t)a.cu_pad * a.cv_pad;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;  // 32 x 8
    for (int tl = first_tile; tl < tu * tv; tl += tile_stride) {
        const int u0 = (tl / tv) * kPackTile, v0 = (tl % tv) * kPackTile;
        for (int t0 = 0; t0 < taps; t0 += 8) {  // (LDS holds eight tap planes at a time)
            const int nt = taps - t0 < 8 ? taps - t0 : 8;
            __syncthreads();
            for (int uu = ly; uu < kPackTile; uu += 8) {
                const int u = u0 + uu, v = v0 + lx;
                const bool in = u < a.cu && v < a.cv;
                const float* src = a.T + ((int64_t)u * a.cv + v) * taps;
                for (int t = 0; t < nt; ++t) {
                    // tap image t0 + t in phase-major order -> its (ky, kx)
                    int r = 0, idx = t0 + t;
                    while (r < a.phases - 1 && idx >= a.tt.ntaps[r]) idx -= a.tt.ntaps[r++];
                    tile[t][uu][lx] = in ? src[a.tt.ky[r][idx] * a.kw + a.tt.kx[r][idx]] : 0.f;
                }
            }
            __syncthreads();
            for (int vv = ly; vv < kPackTile; vv += 8)
                for (int t = 0; t < nt; ++t) a.out[(int64_t)(t0 + t) * pairs + (int64_t)(v0 + vv) * a.cu_pad + u0 + lx] = f2bf(tile[t][lx][vv]);
        }
    }
}

__global__ __launch_bounds__(256) void pack_weight_kernel(const PackArgs2 both) {
    __shared__ float tile[8][kPackTile][kPackTile + 1];
    const PackArgs& a = both.f[blockIdx.y];
    if (a.scatter && !a.fold_s) pack_scatter_tiles(a, tile, blockIdx.x, gridDim.x);
    else pack_one(a, blockIdx.x * (int64_t)blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

// every layer of a model in ONE launch (after an optimiser step all packed images are stale): blockIdx.y = table entry
__global__ __launch_bounds__(256) void pack_weight_batch_kernel(const PackArgs* items) {
    __shared__ float tile[8][kPackTile][kPackTile + 1];
    const PackArgs& a = items[blockIdx.y];
    if (a.scatter && !a.fold_s) pack_scatter_tiles(a, tile, blockIdx.x, gridDim.x);
    else pack_one(a, blockIdx.x * (int64_t)blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
}

__global__ void unpack_wgrad_kernel(const float* packed, float* dT, int cu, int cv, int cu_pad, int cv_pad, int kh,
                                    int kw, int accumulate) {
    const int64_t total = (int64_t)cu * cv * kh * kw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t r = i;
        const int kx = (int)(r % kw);
        r /= kw;
        const int ky = (int)(r % kh);
        r /= kh;
        const int v = (int)(r % cv);
        const int u = (int)(r / cv);
        const float g = packed[((int64_t)(ky * kw + kx) * cu_pad + u) * cv_pad + v];
        dT[i] = accumulate ? dT[i] + g : g;
    }
}

}  // namespace

extern "C" int64_t rv_packed_weight_bytes(const rvTapGeom* g) {
    return (int64_t)g->kh * g->kw * rv_pad32(g->cu) * rv_pad32(g->cv) * (int64_t)sizeof(bf16_t);
}

extern "C" int rv_pack_weight(const rvTapGeom* g, const float* T, void* gather_w, void* scatter_w, rvStream stream) {
    RV_REQUIRE(g && T, "rv_pack_weight: null argument");
    PackArgs2 both;
    memset(&both, 0, sizeof(both));
    int n = 0;
    for (int form = 0; form < 2; ++form) {
        void* out = form ? scatter_w : gather_w;
        if (!out) continue;
        PackArgs& a = both.f[n++];
        int step;
        if (rv_build_tap_table(g, form == 1, &a.tt, &a.phases, &step)) return 1;
        a.T = T;
        a.out = (bf16_t*)out;
        a.cu = g->cu;
        a.cv = g->cv;
        a.cu_pad = rv_pad32(g->cu);
        a.cv_pad = rv_pad32(g->cv);
        a.kh = g->kh;
        a.kw = g->kw;
        a.scatter = form;
    }
    if (n == 0) return 0;
    const int64_t total = (int64_t)rv_pad32(g->cu) * rv_pad32(g->cv);  // one thread per channel pair
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks, n), dim3(256), 0, (hipStream_t)stream, both);  // one launch for both forms
    RV_CHECK_LAUNCH("pack_weight_kernel");
    return 0;
}

static int fill_pack_args(const rvTapGeom* g, const float* T, void* out, int form, PackArgs* a) {
    int step;
    memset(a, 0, sizeof(*a));
    if (rv_build_tap_table(g, form == 1, &a->tt, &a->phases, &step)) return 1;
    a->T = T;
    a->out = (bf16_t*)out;
    a->cu = g->cu;
    a->cv = g->cv;
    a->cu_pad = rv_pad32(g->cu);
    a->cv_pad = rv_pad32(g->cv);
    a->kh = g->kh;
    a->kw = g->kw;
    a->scatter = form;
    return 0;
}

static int floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

extern "C" int rv_fold_geom(const rvTapGeom* g, rvTapGeom* gf) {
    RV_REQUIRE(g && gf && g->stride_w > 1, "rv_fold_geom: a strided geometry is required");
    const int s = g->stride_w, jmin = floor_div(-g->pad_w, s), jmax = floor_div(g->kw - 1 - g->pad_w, s);
    gf->kh = g->kh;
    gf->kw = jmax - jmin + 1;
    gf->stride_w = 1;
    gf->pad_h = g->pad_h;
    gf->pad_w = -jmin;
    gf->cu = g->cu;
    gf->cv = s * rv_pad32(g->cv);
    RV_REQUIRE(gf->kh * gf->kw <= kMaxTaps, "rv_fold_geom: folded kernel too large");
    return 0;
}

static int fill_pack_args_folded(const rvTapGeom* g, const float* T, void* out, PackArgs* a) {
    rvTapGeom gf;
    if (rv_fold_geom(g, &gf)) return 1;
    memset(a, 0, sizeof(*a));
    a->T = T;
    a->out = (bf16_t*)out;
    a->cu = gf.cu;
    a->cv = gf.cv;
    a->cu_pad = rv_pad32(gf.cu);
    a->cv_pad = gf.cv;  // already a multiple of 32
    a->kh = gf.kh;
    a->kw = gf.kw;
    a->scatter = 0;
    a->fold_s = g->stride_w;
    a->fold_pw = g->pad_w;
    a->fold_jmin = -gf.pad_w;
    a->fold_cvp = rv_pad32(g->cv);
    a->fold_cv = g->cv;
    a->fold_kw = g->kw;
    return 0;
}

extern "C" int rv_pack_weight_folded(const rvTapGeom* g, const float* T, void* gather_w_folded, rvStream stream) {
    RV_REQUIRE(g && T && gather_w_folded, "rv_pack_weight_folded: null argument");
    PackArgs2 both;
    memset(&both, 0, sizeof(both));
    if (fill_pack_args_folded(g, T, gather_w_folded, &both.f[0])) return 1;
    const int64_t total = (int64_t)both.f[0].cu_pad * both.f[0].cv_pad;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks, 1), dim3(256), 0, (hipStream_t)stream, both);
    RV_CHECK_LAUNCH("pack_weight_kernel");
    return 0;
}

extern "C" int rv_pack_batch_fill_folded(const rvTapGeom* g, const float* T, void* gather_w_folded, void* host_entry) {
    RV_REQUIRE(g && T && gather_w_folded && host_entry, "rv_pack_batch_fill_folded: null argument");
    return fill_pack_args_folded(g, T, gather_w_folded, (PackArgs*)host_entry);
}

namespace {
// dT[u][c][ky][kx] (+)= dTF[u][r*cvp + c][ky][j'],  kx = s (j' + jmin) + r + pad_w
__global__ void unfold_wgrad_kernel(const float* dTF, float* dT, int cu, int cv, int kh, int kw, int s, int pw, int jmin, int cvp, int kwf,
                                    int accumulate) {
    const int64_t total = (int64_t)cu * cv * kh * kw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t q = i;
        const int kx = (int)(q % kw);
        q /= kw;
        const int ky = (int)(q % kh);
        q /= kh;
        const int c = (int)(q % cv);
        const int u = (int)(q / cv);
        const int d = kx - pw;
        const int j = d >= 0 ? d / s : -((-d + s - 1) / s);
        const int r = d - j * s;
        const float g = dTF[(((int64_t)u * (s * cvp) + r * cvp + c) * kh + ky) * kwf + (j - jmin)];
        dT[i] = accumulate ? dT[i] + g : g;
    }
}
}  // namespace

extern "C" int rv_unfold_weight_grad(const rvTapGeom* g, const float* dT_folded, float* dT, int32_t accumulate, rvStream stream) {
    RV_REQUIRE(g && dT_folded && dT, "rv_unfold_weight_grad: null argument");
    rvTapGeom gf;
    if (rv_fold_geom(g, &gf)) return 1;
    const int64_t total = (int64_t)g->cu * g->cv * g->kh * g->kw;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(unfold_wgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dT_folded, dT, g->cu, g->cv, g->kh, g->kw,
                       g->stride_w, g->pad_w, -gf.pad_w, rv_pad32(g->cv), gf.kw, accumulate);
    RV_CHECK_LAUNCH("unfold_wgrad_kernel");
    return 0;
}

extern "C" int64_t rv_pack_batch_entry_bytes(void) { return (int64_t)sizeof(PackArgs); }

extern "C" int rv_pack_batch_fill(const rvTapGeom* g, const float* T, void* gather_w, void* scatter_w, void* host_entries) {
    RV_REQUIRE(g && T && gather_w && scatter_w && host_entries, "rv_pack_batch_fill: null argument");
    PackArgs* e = (PackArgs*)host_entries;
    return fill_pack_args(g, T, gather_w, 0, e) || fill_pack_args(g, T, scatter_w, 1, e + 1);
}

extern "C" int rv_pack_batch(const void* dev_table, int32_t n_entries, rvStream stream) {
    RV_REQUIRE(dev_table && n_entries > 0 && n_entries <= 65535, "rv_pack_batch: bad table");
    hipLaunchKernelGGL(pack_weight_batch_kernel, dim3(128, n_entries), dim3(256), 0, (hipStream_t)stream, (const PackArgs*)dev_table);
    RV_CHECK_LAUNCH("pack_weight_batch_kernel");
    return 0;
}

extern "C" int rv_unpack_weight_grad(const rvTapGeom* g, const float* packed, float* dT, int32_t accumulate,
                                     rvStream stream) {
    RV_REQUIRE(g && packed && dT, "rv_unpack_weight_grad: null argument");
    const int64_t total = (int64_t)g->cu * g->cv * g->kh * g->kw;
    const int blocks = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, packed, dT, g->cu, g->cv,
                       rv_pad32(g->cu), rv_pad32(g->cv), g->kh, g->kw, accumulate);
    RV_CHECK_LAUNCH("unpack_wgrad_kernel");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// column reduction of partial rows: in[rows][cols] (fp32) -> out[groups][cols] (fp64)
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int kRedGroups = 64;

__global__ __launch_bounds__(256) void col_reduce_kernel(const float* in, int rows, int cols, double* out) {
    // block: 32 columns x 8 row lanes; blockIdx.y = row group
    __shared__ double red[8][33];
    const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int col = blockIdx.x * 32 + cx;
    const int per = (rows + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
    double acc = 0.0;
    if (col < cols)
        for (int r = r0 + ry; r < r1; r += 8) acc += (double)in[(int64_t)r * cols + col];
    red[ry][cx] = acc;
    __syncthreads();
    if (ry == 0 && col < cols) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += red[k][cx];
        out[(int64_t)blockIdx.y * cols + col] = s;
    }
}

// 256 threads = 64 channels x 4 group lanes: the (at most 64) fp64 group rows of a channel are summed by four threads
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* red, int groups, int c, double inv_count, double unbias,
                                   const float* gamma, const float* beta, float eps, float momentum,
                                   float* running_mean, float* running_var, float* scale, float* shift, float* mean_out,
                                   float* invstd_out) {
    __shared__ double part[2][4][64];
    const int cl = threadIdx.x & 63, gl = threadIdx.x >> 6;
    const int ch = blockIdx.x * 64 + cl;
    double s = 0.0, q = 0.0;
    if (ch < c)
        for (int g = gl; g < groups; g += 4) {
            s += red[(int64_t)g * 2 * c + ch];
            q += red[(int64_t)g * 2 * c + c + ch];
        }
    part[0][gl][cl] = s;
    part[1][gl][cl] = q;
    __syncthreads();
    if (gl != 0 || ch >= c) return;
    s = (part[0][0][cl] + part[0][1][cl]) + (part[0][2][cl] + part[0][3][cl]);
    q = (part[1][0][cl] + part[1][1][cl]) + (part[1][2][cl] + part[1][3][cl]);
    const double mean = s * inv_count;
    double var = q * inv_count - mean * mean;
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

// One launch instead of col_reduce + finalize: a block of 256 threads = 16 channels x 16 row lanes (small blocks: a 1024-thread block cannot co-reside with a
// 512-thread MFMA block of the weight-gradient stream on one CU and waited ~90 us for a free CU) sums the partial rows
// of its channels in fp64 (64-byte row segments, 128 loads in flight per thread pair), then finalises them.
__global__ __launch_bounds__(256) void bn_reduce_finalize_kernel(const float* partial, int rows, int c, double inv_count, double unbias,
                                                                  const float* gamma, const float* beta, float eps, float momentum,
                                                                  float* running_mean, float* running_var, float* scale, float* shift,
                                                                  float* mean_out, float* invstd_out, const float* count_dev) {
    if (count_dev) {  // SyncBN: the global element count travelled with the all-reduced totals (a device scalar)
        const double n = (double)count_dev[0];
        inv_count = 1.0 / n;
        unbias = n > 1.0 ? n / (n - 1.0) : 1.0;
    }
    __shared__ double red[2][16][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int ch = blockIdx.x * 16 + cx;
    double s = 0.0, q = 0.0;
    if (ch < c) {
        const float* p = partial + ch;
        int r = ry;
        for (; r + 48 < rows; r += 64) {  // four independent row pairs per iteration
            const float a0 = p[(int64_t)r * 2 * c], b0 = p[(int64_t)r * 2 * c + c];
            const float a1 = p[(int64_t)(r + 16) * 2 * c], b1 = p[(int64_t)(r + 16) * 2 * c + c];
            const float a2 = p[(int64_t)(r + 32) * 2 * c], b2 = p[(int64_t)(r + 32) * 2 * c + c];
            const float a3 = p[(int64_t)(r + 48) * 2 * c], b3 = p[(int64_t)(r + 48) * 2 * c + c];
            s += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
            q += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
        }
        for (; r < rows; r += 16) {
            s += (double)p[(int64_t)r * 2 * c];
            q += (double)p[(int64_t)r * 2 * c + c];
        }
    }
    red[0][ry][cx] = s;
    red[1][ry][cx] = q;
    __syncthreads();
    for (int half = 8; half > 0; half >>= 1) {
        if (ry < half) {
            red[0][ry][cx] += red[0][ry + half][cx];
            red[1][ry][cx] += red[1][ry + half][cx];
        }
        __syncthreads();
    }
    if (ry != 0 || ch >= c) return;
    s = red[0][0][cx];
    q = red[1][0][cx];
    const double mean = s * inv_count;
    double var = q * inv_count - mean * mean;
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

__global__ void bn_fold_eval_kernel(int c, const float* gamma, const float* beta, const float* rm, const float* rv,
                                    float eps, float* scale, float* shift) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    const double sc = (double)gamma[ch] / sqrt((double)rv[ch] + (double)eps);
    scale[ch] = (float)sc;
    shift[ch] = (float)((double)beta[ch] - (double)rm[ch] * sc);
}

}  // namespace

// scratch: the caller's partial buffer must have room for kRedGroups extra rows of 2*c doubles
// == 2 * kRedGroups rows of 2*c floats after the `rows` partial rows (documented in rv3d.h).
int rv_col_reduce(const float* partial, int rows, int cols, double* scratch, int* groups, hipStream_t st) {
    int g = rows < kRedGroups ? rows : kRedGroups;
    if (g < 1) g = 1;
    *groups = g;
    hipLaunchKernelGGL(col_reduce_kernel, dim3(rv_ceil_div(cols, 32), g), dim3(256), 0, st, partial, rows, cols, scratch);
    RV_CHECK_LAUNCH("col_reduce_kernel");
    return 0;
}

// fp32 totals of `rows` partial rows ([rows][cols], fp64 accumulation) -> out[cols]; scratch = the rows behind `rows`
__global__ void sum_groups_f32_kernel(const double* red, int groups, int cols, float* out) {
    const int col = blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= cols) return;
    double s = 0.0;
    for (int g = 0; g < groups; ++g) s += red[(int64_t)g * cols + col];
    out[col] = (float)s;
}

// One launch: block = 16 columns x 16 row lanes, fp64 accumulation, four independent rows in flight per thread (the rows of a
// SyncBN statistic are on the critical path of the step: conv -> totals -> all-reduce -> finalize -> next conv).
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* partial, int rows, int cols, float* out, float* out_copy,
                                                          float count, int write_count) {
    __shared__ double red[16][17];
    const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
    const int col = blockIdx.x * 16 + cx;
    double s = 0.0;
    if (col < cols) {
        const float* p = partial + col;
        int r = ry;
        for (; r + 48 < rows; r += 64) {
            const float a0 = p[(int64_t)r * cols], a1 = p[(int64_t)(r + 16) * cols], a2 = p[(int64_t)(r + 32) * cols], a3 = p[(int64_t)(r + 48) * cols];
            s += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
        }
        for (; r < rows; r += 16) s += (double)p[(int64_t)r * cols];
    }
    red[ry][cx] = s;
    __syncthreads();
    for (int half = 8; half > 0; half >>= 1) {
        if (ry < half) red[ry][cx] += red[ry + half][cx];
        __syncthreads();
    }
    if (ry == 0 && col < cols) {
        const float v = (float)red[0][cx];
        out[col] = v;
        if (out_copy) out_copy[col] = v;
    }
    if (write_count && blockIdx.x == 0 && threadIdx.x == 0) out[cols] = count;
}

extern "C" int rv_reduce_rows(const float* partial, int32_t rows, int32_t cols, float* out, rvStream stream) {
    RV_REQUIRE(partial && out && rows > 0 && cols > 0, "rv_reduce_rows: bad argument");
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(rv_ceil_div(cols, 16)), dim3(256), 0, (hipStream_t)stream, partial, rows, cols, out, (float*)nullptr, 0.f, 0);
    RV_CHECK_LAUNCH("reduce_rows_kernel");
    return 0;
}

extern "C" int rv_reduce_rows_count(const float* partial, int32_t rows, int32_t cols, float count, float* out, float* out_copy,
                                    rvStream stream) {
    RV_REQUIRE(partial && out && rows > 0 && cols > 0, "rv_reduce_rows_count: bad argument");
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(rv_ceil_div(cols, 16)), dim3(256), 0, (hipStream_t)stream, partial, rows, cols, out, out_copy, count, 1);
    RV_CHECK_LAUNCH("reduce_rows_kernel");
    return 0;
}

extern "C" int rv_bn_finalize(const float* partial, int32_t rows, int32_t c, int64_t count, const float* gamma,
                              const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                              float* scale, float* shift, float* mean, float* invstd, rvStream stream) {
    RV_REQUIRE(partial && gamma && beta && scale && shift, "rv_bn_finalize: null argument");
    RV_REQUIRE(rows > 0 && c > 0 && count != 0, "rv_bn_finalize: empty reduction");
    RV_REQUIRE(count > 0 || rows == 1, "rv_bn_finalize: a device-side count (count < 0) needs the single row of all-reduced totals");
    const float* count_dev = count < 0 ? partial + 2 * c : nullptr;  // the slot right behind the (2, c) totals
    const double unbias = count > 1 ? (double)count / (double)(count - 1) : 1.0;
    // few partial rows: one launch reduces and finalises; many (4096 rows behind a 512-channel tapconv4 launch): the
    // 64-group column reduction spreads them over the chip first (29 us vs 14 us measured for the single launch)
    if (rows <= 2048) {
        hipLaunchKernelGGL(bn_reduce_finalize_kernel, dim3(rv_ceil_div(c, 16)), dim3(256), 0, (hipStream_t)stream, partial, rows, c,
                           1.0 / (double)count, unbias, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean, invstd,
                           count_dev);
        RV_CHECK_LAUNCH("bn_reduce_finalize_kernel");
        return 0;
    }
    RV_REQUIRE(count > 0, "rv_bn_finalize: the two-stage path (> 2048 rows) takes a host-side count only");
    double* scratch = (double*)(partial + (int64_t)rows * 2 * c);
    int groups;
    if (rv_col_reduce(partial, rows, 2 * c, scratch, &groups, (hipStream_t)stream)) return 1;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(rv_ceil_div(c, 64)), dim3(256), 0, (hipStream_t)stream, scratch, groups, c,
                       1.0 / (double)count, unbias, gamma, beta, eps, momentum, running_mean, running_var, scale, shift,
                       mean, invstd);
    RV_CHECK_LAUNCH("bn_finalize_kernel");
    return 0;
}

extern "C" int rv_bn_fold_eval(int32_t c, const float* gamma, const float* beta, const float* running_mean,
                               const float* running_var, float eps, float* scale, float* shift, rvStream stream) {
    RV_REQUIRE(gamma && beta && running_mean && running_var && scale && shift, "rv_bn_fold_eval: null argument");
    hipLaunchKernelGGL(bn_fold_eval_kernel, dim3(rv_ceil_div(c, 64)), dim3(64), 0, (hipStream_t)stream, c, gamma, beta,
                       running_mean, running_var, eps, scale, shift);
    RV_CHECK_LAUNCH("bn_fold_eval_kernel");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// element-wise combine: out = relu?( fa(a) + fb(b) )
// ---------------------------------------------------------------------------------------------
namespace {

struct EwArgs {
    const bf16_t* a;
    const bf16_t* b;
    bf16_t* out;
    const float *a_scale, *a_shift, *b_scale, *b_shift;
    int64_t pixels;
    int c8;  // channel octets
    int ld_a, ld_b, ld_out;
    int flags;
};

__device__ __forceinline__ void load8(const bf16_t* p, float* f) {
    const u32x4 v = *(const u32x4*)p;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f[2 * j] = bf_lo(v[j]);
        f[2 * j + 1] = bf_hi(v[j]);
    }
}
__device__ __forceinline__ void store8(bf16_t* p, const float* f) {
    u32x4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = pack_bf2(f[2 * j], f[2 * j + 1]);
    *(u32x4*)p = v;
}

__global__ __launch_bounds__(256) void ew_combine_kernel(const EwArgs e) {
    const int64_t total = e.pixels * e.c8;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t px = i / e.c8;
        const int c = (int)(i - px * e.c8) * 8;
        float va[8], vb[8];
        load8(e.a + px * e.ld_a + c, va);
        if (e.a_scale) {
#pragma unroll
            for (int j = 0; j < 8; ++j) va[j] = va[j] * e.a_scale[c + j] + e.a_shift[c + j];
        }
        if (e.flags & RV_EW_RELU_A) {
#pragma unroll
            for (int j = 0; j < 8; ++j) va[j] = fmaxf(va[j], 0.f);
        }
        if (e.b) {
            load8(e.b + px * e.ld_b + c, vb);
            if (e.b_scale) {
#pragma unroll
                for (int j = 0; j < 8; ++j) vb[j] = vb[j] * e.b_scale[c + j] + e.b_shift[c + j];
            }
            if (e.flags & RV_EW_RELU_B) {
#pragma unroll
                for (int j = 0; j < 8; ++j) vb[j] = fmaxf(vb[j], 0.f);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) va[j] += vb[j];
        }
        if (e.flags & RV_EW_RELU_OUT) {
#pragma unroll
            for (int j = 0; j < 8; ++j) va[j] = fmaxf(va[j], 0.f);
        }
        store8(e.out + px * e.ld_out + c, va);
    }
}

// The same map with thread = (pixel lane, channel octet): the folded-BatchNorm constants of the thread's octet stay in
// registers, and every workgroup walks one CONTIGUOUS pixel range with two pixels in flight (see bn_bwd_apply_kernel:
// contiguous ranges read + write ~15 % faster than a grid-stride comb).  c8 <= 256.
// HAS_B = false: the one-operand form (writing relu(bn(y)) out for a DMA kernel) in at most 64 VGPRs, so that a workgroup of it
// fits on a CU beside a resident tapconv6 workgroup (2 waves per SIMD x 224 VGPRs): the towers of a head run on two streams
// (program.dense_head_pair_program) and one tower's write-out pass streams while the other tower's conv computes.
template <bool NT, bool HAS_B>  // NT: non-temporal loads and stores (tensors beyond the Infinity Cache: nothing of them is found there again)
__global__ __launch_bounds__(256, HAS_B ? 1 : 8) void ew_combine_rows_kernel(const EwArgs e) {
    const int tid = threadIdx.x;
    const int lanes_px = 256 / e.c8;
    const int oct = tid % e.c8, pl = tid / e.c8;
    if (pl >= lanes_px) return;
    const int c = oct * 8;
    float as[8], ah[8], bs[8], bh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        as[j] = e.a_scale ? e.a_scale[c + j] : 1.f;
        ah[j] = e.a_scale ? e.a_shift[c + j] : 0.f;
        bs[j] = (HAS_B && e.b_scale) ? e.b_scale[c + j] : 1.f;
        bh[j] = (HAS_B && e.b_scale) ? e.b_shift[c + j] : 0.f;
    }
    const int64_t per = ((e.pixels + gridDim.x - 1) / gridDim.x + lanes_px - 1) / lanes_px * lanes_px;
    const int64_t end = (int64_t)(blockIdx.x + 1) * per < e.pixels ? (int64_t)(blockIdx.x + 1) * per : e.pixels;
    auto finish = [&](int64_t px, const u32x4 ra, const u32x4 rb) {
        float va[8], vb[8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            va[2 * j] = bf_lo(ra[j]);
            va[2 * j + 1] = bf_hi(ra[j]);
            vb[2 * j] = bf_lo(rb[j]);
            vb[2 * j + 1] = bf_hi(rb[j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float x = e.a_scale ? va[j] * as[j] + ah[j] : va[j];
            if (e.flags & RV_EW_RELU_A) x = fmaxf(x, 0.f);
            if (HAS_B && e.b) {
                float y = e.b_scale ? vb[j] * bs[j] + bh[j] : vb[j];
                if (e.flags & RV_EW_RELU_B) y = fmaxf(y, 0.f);
                x += y;
            }
            if (e.flags & RV_EW_RELU_OUT) x = fmaxf(x, 0.f);
            va[j] = x;
        }
        if (NT) {
            u32x4 v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = pack_bf2(va[2 * j], va[2 * j + 1]);
            __builtin_nontemporal_store(v, (u32x4*)(e.out + px * e.ld_out + c));
        } else {
            store8(e.out + px * e.ld_out + c, va);
        }
    };
    auto ld = [&](const bf16_t* p) -> u32x4 { return NT ? __builtin_nontemporal_load((const u32x4*)p) : *(const u32x4*)p; };
    int64_t px = (int64_t)blockIdx.x * per + pl;
    for (; px + lanes_px < end; px += 2 * lanes_px) {
        const u32x4 a0 = ld(e.a + px * e.ld_a + c), a1 = ld(e.a + (px + lanes_px) * e.ld_a + c);
        u32x4 b0 = {}, b1 = {};
        if (HAS_B && e.b) {
            b0 = ld(e.b + px * e.ld_b + c);
            b1 = ld(e.b + (px + lanes_px) * e.ld_b + c);
        }
        finish(px, a0, b0);
        finish(px + lanes_px, a1, b1);
    }
    if (px < end) {
        const u32x4 a0 = ld(e.a + px * e.ld_a + c);
        u32x4 b0 = {};
        if (HAS_B && e.b) b0 = ld(e.b + px * e.ld_b + c);
        finish(px, a0, b0);
    }
}

}  // namespace

static int ew_grid(int64_t work) {
    int64_t b = (work + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 256 * 8 ? 256 * 8 : b));
}

extern "C" int rv_ew_combine(int64_t pixels, int32_t c, const void* a, int32_t ld_a, const float* a_scale,
                             const float* a_shift, const void* b, int32_t ld_b, const float* b_scale,
                             const float* b_shift, void* out, int32_t ld_out, int32_t flags, rvStream stream) {
    RV_REQUIRE(a && out, "rv_ew_combine: null argument");
    RV_REQUIRE(c % 8 == 0 && ld_a % 8 == 0 && ld_out % 8 == 0 && (!b || ld_b % 8 == 0), "rv_ew_combine: channels / strides must be multiples of 8");
    RV_REQUIRE((a_scale == nullptr) == (a_shift == nullptr) && (b_scale == nullptr) == (b_shift == nullptr), "rv_ew_combine: scale and shift go together");
    EwArgs e{(const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, a_scale, a_shift, b_scale, b_shift, pixels, c / 8, ld_a, ld_b, ld_out, flags};
    // (measured, profiles/r02_hbm_kernels.md: the row-range kernel wins on tensors beyond the Infinity Cache, 4.9-5.2 vs 4.7-4.8 TB/s;
    //  on small ones its per-thread constant prologue costs more than the comb's address arithmetic, 2.9 vs 6.2 TB/s)
    if (e.c8 <= 256 && pixels * c * 2 >= ((int64_t)256 << 20)) {
        const int lanes_px = 256 / e.c8;
        int64_t blocks = (pixels + lanes_px - 1) / lanes_px;
        if (blocks > 4096) blocks = 4096;
        if (e.b) hipLaunchKernelGGL((ew_combine_rows_kernel<true, true>), dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, e);  // <non-temporal, two operands>
        else hipLaunchKernelGGL((ew_combine_rows_kernel<true, false>), dim3((int)blocks), dim3(256), 0, (hipStream_t)stream, e);
    } else {
        hipLaunchKernelGGL(ew_combine_kernel, dim3(ew_grid(pixels * (c / 8))), dim3(256), 0, (hipStream_t)stream, e);
    }
    RV_CHECK_LAUNCH("ew_combine_kernel");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// layout conversion at the module boundary
// ---------------------------------------------------------------------------------------------
namespace {

// one thread per pixel; reads are coalesced along W, the few channels are written as one run
template <typename DstT>
__global__ void nchw_to_nhwc_kernel(const float* src, int N, int C, int H, int W, DstT* dst, int ld, int c_off) {
    const int64_t hw = (int64_t)H * W, total = (int64_t)N * hw;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = p / hw, r = p - n * hw;
        for (int c = 0; c < C; ++c) {
            const float v = src[(n * C + c) * hw + r];
            if constexpr (sizeof(DstT) == 2)
                dst[p * ld + c_off + c] = f2bf(v);
            else
                dst[p * ld + c_off + c] = v;
        }
    }
}

template <typename SrcT>
__global__ void nhwc_to_nchw_kernel(const SrcT* src, int ld, int c_off, int N, int C, int H, int W, float* dst) {
    // tile: 64 pixels x 32 channels through LDS so that both sides are coalesced
    __shared__ float tile[32][65];
    const int64_t hw = (int64_t)H * W;
    const int64_t p0 = (int64_t)blockIdx.x * 64;  // pixel tile (within N*H*W)
    const int c0 = blockIdx.y * 32;
    const int64_t total = (int64_t)N * hw;
    for (int i = threadIdx.x; i < 64 * 32; i += blockDim.x) {
        const int px = i >> 5, c = i & 31;
        float v = 0.f;
        if (p0 + px < total && c0 + c < C) {
            if constexpr (sizeof(SrcT) == 2)
                v = bf2f(src[(p0 + px) * ld + c_off + c0 + c]);
            else
                v = src[(p0 + px) * ld + c_off + c0 + c];
        }
        tile[c][px] = v;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 32; i += blockDim.x) {
        const int c = i >> 6, px = i & 63;
        const int64_t p = p0 + px;
        if (p < total && c0 + c < C) {
            const int64_t n = p / hw, r = p - n * hw;
            dst[(n * C + c0 + c) * hw + r] = tile[c][px];
        }
    }
}

}  // namespace

extern "C" int rv_nchw_f32_to_nhwc_bf16(const float* src, int32_t N, int32_t C, int32_t H, int32_t W, void* dst,
                                        int32_t ld_dst, int32_t c_offset, rvStream stream) {
    RV_REQUIRE(src && dst && c_offset + C <= ld_dst, "rv_nchw_f32_to_nhwc_bf16: bad arguments");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3(ew_grid((int64_t)N * H * W)), dim3(256), 0, (hipStream_t)stream,
                       src, N, C, H, W, (bf16_t*)dst, ld_dst, c_offset);
    RV_CHECK_LAUNCH("nchw_to_nhwc_kernel");
    return 0;
}
// ---------------------------------------------------------------------------------------------------------------
// RangePartition stem operand (nn/stems/__init__.py:121-135): every input channel once per range band, zero outside the band and
// where the pixel holds no return, as the bf16 NHWC operand of the projecting BasicBlock -- channel index band * C + c (what
// `flatten(1, 2)` of (B, bands, C, H, W) gives).  d = ||cart|| in fp32 without contraction (torch's fp32 norm: sum of squares, then
// a correctly rounded square root); the bands are closed intervals compared in fp32.  One thread per pixel.
// ---------------------------------------------------------------------------------------------------------------
namespace {
struct BandArgs {
    float lower[8], upper[8];
};
__global__ void range_partition_kernel(const float* feat, const float* cart, const uint8_t* mask, int N, int C, int H, int W, BandArgs b, int bands,
                                       bf16_t* dst, int ld) {
#pragma clang fp contract(off)
    const int64_t hw = (int64_t)H * W, total = (int64_t)N * hw;
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < total; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t n = p / hw, r = p - n * hw;
        const float* cp = cart + n * 3 * hw + r;
        const float x = cp[0], y = cp[hw], z = cp[2 * hw];
        const float d = sqrtf(x * x + y * y + z * z);
        const bool valid = mask[p] != 0;
        bf16_t* o = dst + p * ld;
        for (int k = 0; k < bands; ++k) {
            const bool in = valid && d >= b.lower[k] && d <= b.upper[k];
            // (the reference multiplies: outside the band the value is 0 * f -- a zero with f's sign, NaN for a non-finite f)
            for (int c = 0; c < C; ++c) o[k * C + c] = f2bf(feat[(n * C + c) * hw + r] * (in ? 1.f : 0.f));
        }
        for (int c = bands * C; c < ld; ++c) o[c] = 0;
    }
}
}  // namespace

extern "C" int rv_range_partition(const float* features_nchw, const float* cart_nchw, const uint8_t* mask, int32_t N, int32_t C, int32_t H, int32_t W,
                                  const float* lower, const float* upper, int32_t bands, void* dst, int32_t ld_dst, rvStream stream) {
    RV_REQUIRE(features_nchw && cart_nchw && mask && lower && upper && dst, "rv_range_partition: null argument");
    RV_REQUIRE(bands >= 1 && bands <= 8 && C >= 1 && bands * C <= ld_dst, "rv_range_partition: %d bands x %d channels do not fit a row of %d", bands, C, ld_dst);
    BandArgs b;
    for (int k = 0; k < 8; ++k) {
        b.lower[k] = k < bands ? lower[k] : 0.f;
        b.upper[k] = k < bands ? upper[k] : 0.f;
    }
    hipLaunchKernelGGL(range_partition_kernel, dim3(ew_grid((int64_t)N * H * W)), dim3(256), 0, (hipStream_t)stream, features_nchw, cart_nchw, mask, N, C, H,
                       W, b, bands, (bf16_t*)dst, ld_dst);
    RV_CHECK_LAUNCH("range_partition_kernel");
    return 0;
}

extern "C" int rv_nchw_f32_to_nhwc_f32(const float* src, int32_t N, int32_t C, int32_t H, int32_t W, float* dst,
                                       int32_t ld_dst, rvStream stream) {
    RV_REQUIRE(src && dst && C <= ld_dst, "rv_nchw_f32_to_nhwc_f32: bad arguments");
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(ew_grid((int64_t)N * H * W)), dim3(256), 0, (hipStream_t)stream,
                       src, N, C, H, W, dst, ld_dst, 0);
    RV_CHECK_LAUNCH("nchw_to_nhwc_kernel");
    return 0;
}
extern "C" int rv_nhwc_bf16_to_nchw_f32(const void* src, int32_t ld_src, int32_t c_offset, int32_t N, int32_t C,
                                        int32_t H, int32_t W, float* dst, rvStream stream) {
    RV_REQUIRE(src && dst && c_offset + C <= ld_src, "rv_nhwc_bf16_to_nchw_f32: bad arguments");
    const int64_t total = (int64_t)N * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3((unsigned)((total + 63) / 64), rv_ceil_div(C, 32)), dim3(256), 0,
                       (hipStream_t)stream, (const bf16_t*)src, ld_src, c_offset, N, C, H, W, dst);
    RV_CHECK_LAUNCH("nhwc_to_nchw_kernel");
    return 0;
}
extern "C" int rv_nhwc_f32_to_nchw_f32(const float* src, int32_t ld_src, int32_t N, int32_t C, int32_t H, int32_t W,
                                       float* dst, rvStream stream) {
    RV_REQUIRE(src && dst && C <= ld_src, "rv_nhwc_f32_to_nchw_f32: bad arguments");
    const int64_t total = (int64_t)N * H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3((unsigned)((total + 63) / 64), rv_ceil_div(C, 32)), dim3(256), 0,
                       (hipStream_t)stream, src, ld_src, 0, N, C, H, W, dst);
    RV_CHECK_LAUNCH("nhwc_to_nchw_kernel");
    return 0;
}

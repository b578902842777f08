// project.hip -- range-image projection: spherical binning + z-buffer (integer/atomic work, HBM-bound).
//
// Reference: cart_to_sph + build_range_view_coordinates + z_buffer,
//   converters/av2/utils.py:108-208  (converter binning)  ==  math/numpy/conversions.py:9-128 (library binning).
//
// The reference z-buffer is a sequential scan: point i takes pixel p iff range_i (fp64) <
// buffer[p] (fp32 rounding of the current owner's range).  That scan has a closed form, which
// is what the three passes below evaluate with 64-bit atomics (result identical to the scan,
// including the fp64-vs-fp32 quirk):
//   Bmin(p)  = min_i fp32(range_i)                      (the buffer only ever decreases to this)
//   i0(p)    = smallest index with fp32(range_i) == Bmin (the point that first writes Bmin)
//   L(p)     = { i : fp32(range_i) == Bmin and range_i < (fp64)Bmin }   (they all come at or after i0
//              and each of them overwrites the pixel again)
//   owner(p) = max(L) if L is non-empty else i0.
// Pass 1: atomicMin on (float_bits(fp32 range) << 32 | index)  -> Bmin, i0.
// Pass 2: atomicMax(winner[p], i) over L.       Pass 3: gather the owner's features.
#include "common.h"

#include "atan_table.h"

namespace {

// ---------------------------------------------------------------------------------------------------------------
// Correctly rounded fp64 atan2.  The column of a point is W - rint((az + pi) * W / tau): integer work that must be
// bit-exact (BASELINE.json north_star), so the azimuth must be a function of (x, y) alone -- the device libm's atan2 is
// within 1-2 ulp, as is numpy's SIMD arctan2 (which differs from glibc's in ~1 % of inputs), and a last-bit difference
// moves points that sit on a half-bin boundary.  This one returns THE round-to-nearest-even value of atan2(y, x):
// 128-bit fixed-point CORDIC (Q.124, table from gen_atan_table.py; accumulated error < 2^-116 against an ulp of
// >= 2^-83 on this path), a two-term series in double-double for tiny positive-x angles, libm for the exact special
// cases (zeros, infinities, NaN).  Witness: oracle/project.py::atan2_cr (80-bit atan2l + exact decimal fallback).
// ---------------------------------------------------------------------------------------------------------------
typedef __int128 i128_t;
typedef unsigned __int128 u128_t;
struct Q128 { uint64_t hi, lo; };
__constant__ Q128 c_atan_q[RV_ATAN_N] = RV_ATAN_TABLE_INIT;

__device__ __forceinline__ void frexp_bits(double v, uint64_t* mant53, int* exp2) {  // v = mant53 * 2^exp2, mant53 in [2^52, 2^53)
    const uint64_t b = (uint64_t)__double_as_longlong(v) & 0x7fffffffffffffffull;
    int e = (int)(b >> 52);
    uint64_t m = b & 0xfffffffffffffull;
    if (e == 0) {  // subnormal: normalise
        const int sh = __clzll((long long)m) - 11;
        m <<= sh;
        e = 1 - sh;
    } else {
        m |= 1ull << 52;
    }
    *mant53 = m;
    *exp2 = e - 1075;
}

__device__ double round_q124(u128_t z) {  // non-negative Q.124 -> nearest-even double
    if (z == 0) return 0.0;
    const uint64_t hi = (uint64_t)(z >> 64), lo = (uint64_t)z;
    const int p = hi ? 127 - __clzll((long long)hi) : 63 - __clzll((long long)lo);
    if (p <= 52) return ldexp((double)lo, -RV_ATAN_FRAC);
    const int sh = p - 52;
    uint64_t mant = (uint64_t)(z >> sh);
    const u128_t rem = z & ((((u128_t)1) << sh) - 1), half = ((u128_t)1) << (sh - 1);
    if (rem > half || (rem == half && (mant & 1))) ++mant;
    return ldexp((double)mant, sh - RV_ATAN_FRAC);  // mant <= 2^53: exact conversion
}

__device__ double atan2_cr(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    if (ax == 0.0 || ay == 0.0 || !isfinite(ax) || !isfinite(ay)) return atan2(y, x);  // exact special values
    if (x > 0.0 && ay < ax * 0x1p-30) {
        // atan(q) = q - q^3/3 + O(q^5), q = ay/ax < 2^-30: the q^5 term is < 2^-122 relative
        const double qh = ay / ax;
        const double ql = fma(-qh, ax, ay) / ax;
        const double corr = -(qh * qh * qh) / 3.0;
        return copysign(qh + (ql + corr), y);
    }
    uint64_t mx, my;
    int ex, ey;
    frexp_bits(ax, &mx, &ex);
    frexp_bits(ay, &my, &ey);
    const int e = ex > ey ? ex : ey;
    i128_t X = (e - ex < 128) ? (i128_t)((((u128_t)mx) << 72) >> (e - ex)) : (i128_t)0;
    i128_t Y = (e - ey < 128) ? (i128_t)((((u128_t)my) << 72) >> (e - ey)) : (i128_t)0;
    i128_t z = 0;
    for (int i = 0; i < RV_ATAN_N; ++i) {
        const i128_t a = (i128_t)((((u128_t)c_atan_q[i].hi) << 64) | c_atan_q[i].lo);
        const i128_t xs = X >> i, ys = Y >> i;  // arithmetic shifts (floor)
        if (Y >= 0) {
            X += ys;
            Y -= xs;
            z += a;
        } else {
            X -= ys;
            Y += xs;
            z -= a;
        }
    }
    if (z < 0) z = 0;
    if (x < 0.0) z = (i128_t)((((u128_t)RV_PI_Q_HI) << 64) | RV_PI_Q_LO) - z;
    return copysign(round_q124((u128_t)z), y);
}

__global__ void atan2_cr_kernel(const double* y, const double* x, int64_t n, double* out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = atan2_cr(y[i], x[i]);
}

// ---------------------------------------------------------------------------------------------------------------
// fp64 hypot as the reference's converter gets it.  `np.hypot` (math/numpy/conversions.py:64-65, converters/av2/utils.py:
// 108-131) is the C library's `hypot`; on this image that is glibc 2.35 (sysdeps/ieee754/dbl-64/e_hypot.c), a third-party
// dependency of the reference that is not correctly rounded (0.5 % of random inputs differ from the rounded exact value) and
// that the device libm does not reproduce either (last-bit differences in ~1 point in 6).  The range feeds the z-buffer's
// comparisons (index work: bit-exact), so this restates glibc's published algorithm -- Borges, "An Improved Algorithm for
// hypot(a,b)" (arXiv:1904.09481), the variant without fused multiply-add that an x86-64 build takes: h = sqrt(ax^2+ay^2)
// followed by one correction step from the exactly computed residual -- with glibc's scaling for huge / tiny operands.
// Only IEEE +,-,*,/,sqrt in fp64 (all correctly rounded on gfx950) with contraction off, so the same operations give the same
// bits.  Checked against np.hypot on 1.2e6 points in tests/test_gpu_forward.py.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double hypot_kernel(double ax, double ay) {
#pragma clang fp contract(off)
    double h = sqrt(ax * ax + ay * ay);
    double t1, t2;
    if (h <= 2.0 * ay) {
        const double delta = h - ay;
        t1 = ax * (2.0 * delta - ax);
        t2 = (delta - 2.0 * (ax - ay)) * delta;
    } else {
        const double delta = h - ax;
        t1 = 2.0 * delta * (ax - 2.0 * ay);
        t2 = (4.0 * delta - ay) * ay + delta * delta;
    }
    h -= (t1 + t2) / (2.0 * h);
    return h;
}

__device__ double hypot_libc(double x, double y) {
#pragma clang fp contract(off)
    const double kScale = 0x1p-600, kLarge = 0x1p+511, kTiny = 0x1p-459, kEps = 0x1p-54;
    if (!isfinite(x) || !isfinite(y)) return (isinf(x) || isinf(y)) ? INFINITY : x + y;  // inf wins over NaN (C99 F.9.4.3)
    x = fabs(x);
    y = fabs(y);
    const double ax = x < y ? y : x, ay = x < y ? x : y;
    if (ax > kLarge) {
        if (ay <= ax * kEps) return ax + ay;
        return hypot_kernel(ax * kScale, ay * kScale) / kScale;
    }
    if (ay < kTiny) {
        if (ax >= ay / kEps) return ax + ay;
        return hypot_kernel(ax / kScale, ay / kScale) * kScale;
    }
    if (ax >= ay / kEps) return ax + ay;
    return hypot_kernel(ax, ay);
}

__global__ void hypot_libc_kernel(const double* x, const double* y, int64_t n, double* out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) out[i] = hypot_libc(x[i], y[i]);
}

__global__ void project_indices_kernel(const double* cart, const int32_t* laser, const int32_t* laser_mapping, int64_t n,
                                       int H, int W, int variant, int32_t* rows, int32_t* cols, double* range) {
    const double kPi = 3.141592653589793, kTau = 6.283185307179586;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double x = cart[3 * i], y = cart[3 * i + 1], z = cart[3 * i + 2];
        const double hyp = hypot_libc(x, y);
        const double r = hypot_libc(hyp, z);
        double az = atan2_cr(y, x);
        az += kPi;
        az *= (double)W / kTau;
        double col = variant == 0 ? (double)W - rint(az) : rint((double)W - az - 1.0);  // rint: round-half-even
        col = fmin(fmax(col, 0.0), (double)(W - 1));
        rows[i] = H - laser_mapping[laser[i]] - 1;
        cols[i] = (int32_t)col;
        range[i] = r;
    }
}

__global__ void zbuf_init_kernel(uint64_t* keys, int64_t* winner, int64_t n_pix) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n_pix; i += (int64_t)gridDim.x * blockDim.x) {
        keys[i] = ~0ull;
        winner[i] = -1;
    }
}

__device__ __forceinline__ bool zbuf_pixel(const int32_t* rows, const int32_t* cols, const double* range, int64_t i, int H,
                                           int W, double min_range, int64_t* pix) {
    const int r = rows[i], c = cols[i];
    if (r < 0 || r >= H || c < 0 || c >= W) return false;
    if (range[i] < min_range) return false;
    *pix = (int64_t)r * W + c;
    return true;
}

__global__ void zbuf_min_kernel(const int32_t* rows, const int32_t* cols, const double* range, int64_t n, int H, int W,
                                double min_range, unsigned long long* keys) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t p;
        if (!zbuf_pixel(rows, cols, range, i, H, W, min_range, &p)) continue;
        const float rf = (float)range[i];  // round-to-nearest-even, as the fp32 buffer store does
        const unsigned long long key = ((unsigned long long)__float_as_uint(rf) << 32) | (unsigned long long)(uint32_t)i;
        atomicMin(&keys[p], key);
    }
}

__global__ void zbuf_last_kernel(const int32_t* rows, const int32_t* cols, const double* range, int64_t n, int H, int W,
                                 double min_range, const unsigned long long* keys, long long* winner) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t p;
        if (!zbuf_pixel(rows, cols, range, i, H, W, min_range, &p)) continue;
        const float bmin = __uint_as_float((uint32_t)(keys[p] >> 32));
        if ((float)range[i] == bmin && range[i] < (double)bmin) atomicMax(&winner[p], (long long)i);
    }
}

__global__ void zbuf_gather_kernel(const unsigned long long* keys, long long* winner, const double* features, int64_t n,
                                   int c, int64_t n_pix, float* image) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n_pix; p += (int64_t)gridDim.x * blockDim.x) {
        long long w = winner[p];
        if (w < 0 && keys[p] != ~0ull) w = (long long)(uint32_t)(keys[p] & 0xffffffffull);
        winner[p] = w;
        for (int ch = 0; ch < c; ++ch) image[(int64_t)ch * n_pix + p] = w >= 0 ? (float)features[(int64_t)ch * n + w] : 0.f;
    }
}

int grid_for(int64_t work) {
    int64_t b = (work + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" int rv_atan2_cr(const double* y, const double* x, int64_t n, double* out, rvStream stream) {
    if (n == 0) return 0;
    RV_REQUIRE(y && x && out, "rv_atan2_cr: null argument");
    hipLaunchKernelGGL(atan2_cr_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, y, x, n, out);
    RV_CHECK_LAUNCH("atan2_cr_kernel");
    return 0;
}

extern "C" int rv_hypot_libc(const double* x, const double* y, int64_t n, double* out, rvStream stream) {
    if (n == 0) return 0;
    RV_REQUIRE(x && y && out, "rv_hypot_libc: null argument");
    hipLaunchKernelGGL(hypot_libc_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x, y, n, out);
    RV_CHECK_LAUNCH("hypot_libc_kernel");
    return 0;
}

extern "C" int rv_project_indices(const double* cart, const int32_t* laser, const int32_t* laser_mapping, int64_t n,
                                  int32_t H, int32_t W, int32_t variant, int32_t* rows, int32_t* cols, double* range,
                                  rvStream stream) {
    if (n == 0) return 0;
    RV_REQUIRE(cart && laser && laser_mapping && rows && cols && range, "rv_project_indices: null argument");
    RV_REQUIRE(variant == 0 || variant == 1, "rv_project_indices: variant must be 0 (converter) or 1 (library)");
    hipLaunchKernelGGL(project_indices_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, cart, laser,
                       laser_mapping, n, H, W, variant, rows, cols, range);
    RV_CHECK_LAUNCH("project_indices_kernel");
    return 0;
}

extern "C" int rv_z_buffer(const int32_t* rows, const int32_t* cols, const double* range, const double* features,
                           int64_t n, int32_t c, int32_t H, int32_t W, double min_range, uint64_t* keys, float* image,
                           int64_t* winner, rvStream stream) {
    RV_REQUIRE(keys && image && winner, "rv_z_buffer: null output");
    RV_REQUIRE(n == 0 || (rows && cols && range && features), "rv_z_buffer: null input");
    RV_REQUIRE(n < (1ll << 32), "rv_z_buffer: more than 2^32 points");
    hipStream_t st = (hipStream_t)stream;
    const int64_t n_pix = (int64_t)H * W;
    hipLaunchKernelGGL(zbuf_init_kernel, dim3(grid_for(n_pix)), dim3(256), 0, st, keys, winner, n_pix);
    if (n > 0) {
        hipLaunchKernelGGL(zbuf_min_kernel, dim3(grid_for(n)), dim3(256), 0, st, rows, cols, range, n, H, W, min_range,
                           (unsigned long long*)keys);
        hipLaunchKernelGGL(zbuf_last_kernel, dim3(grid_for(n)), dim3(256), 0, st, rows, cols, range, n, H, W, min_range,
                           (const unsigned long long*)keys, (long long*)winner);
    }
    hipLaunchKernelGGL(zbuf_gather_kernel, dim3(grid_for(n_pix)), dim3(256), 0, st, (const unsigned long long*)keys,
                       (long long*)winner, features, n, c, n_pix, image);
    RV_CHECK_LAUNCH("z_buffer kernels");
    return 0;
}

// ---------------------------------------------------------------------------------------------
// S1: spherical <-> Cartesian (math/conversions.py:28-81; numpy twins math/numpy/conversions.py:46-103) and
// R2: the loader's W padding (prototype/loader.py:792-815).  Element-wise, one point / pixel per thread.
// ---------------------------------------------------------------------------------------------
namespace {

template <typename T>
__global__ void cart_to_sph_kernel(const T* cart, int64_t n, T* sph) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T x = cart[3 * i], y = cart[3 * i + 1], z = cart[3 * i + 2];
        const T hyp = hypot(x, y);
        sph[3 * i] = atan2(y, x);
        sph[3 * i + 1] = atan2(z, hyp);
        sph[3 * i + 2] = hypot(hyp, z);
    }
}

template <typename T>
__global__ void sph_to_cart_kernel(const T* sph, int64_t n, T* cart) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T az = sph[3 * i], inc = sph[3 * i + 1], r = sph[3 * i + 2];
        const T rc = r * cos(inc);
        cart[3 * i] = rc * cos(az);
        cart[3 * i + 1] = rc * sin(az);
        cart[3 * i + 2] = r * sin(inc);
    }
}

// (C,H,W) -> (C,H,W+2*pad): zeros ("constant") or wrap-around in azimuth ("circular"); optional per-pixel mask multiply
__global__ void pad_width_kernel(const float* src, const float* mask, int C, int H, int W, int pad, int circular, float* dst) {
    const int Wp = W + 2 * pad;
    const int64_t total = (int64_t)C * H * Wp;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int w = (int)(i % Wp);
        const int64_t ch = i / Wp;  // c*H + h
        const int h = (int)(ch % H);
        int ws = w - pad;
        float v = 0.f;
        if (circular) ws = (ws % W + W) % W;
        if (ws >= 0 && ws < W) {
            v = src[ch * W + ws];
            if (mask) v *= mask[(int64_t)h * W + ws];
        }
        dst[i] = v;
    }
}

}  // namespace

extern "C" int rv_cart_to_sph(const void* cart, int64_t n, int32_t is_f64, void* sph, rvStream stream) {
    if (n == 0) return 0;
    RV_REQUIRE(cart && sph, "rv_cart_to_sph: null argument");
    if (is_f64)
        hipLaunchKernelGGL(cart_to_sph_kernel<double>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const double*)cart, n, (double*)sph);
    else
        hipLaunchKernelGGL(cart_to_sph_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)cart, n, (float*)sph);
    RV_CHECK_LAUNCH("cart_to_sph_kernel");
    return 0;
}

extern "C" int rv_sph_to_cart(const void* sph, int64_t n, int32_t is_f64, void* cart, rvStream stream) {
    if (n == 0) return 0;
    RV_REQUIRE(cart && sph, "rv_sph_to_cart: null argument");
    if (is_f64)
        hipLaunchKernelGGL(sph_to_cart_kernel<double>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const double*)sph, n, (double*)cart);
    else
        hipLaunchKernelGGL(sph_to_cart_kernel<float>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float*)sph, n, (float*)cart);
    RV_CHECK_LAUNCH("sph_to_cart_kernel");
    return 0;
}

extern "C" int rv_pad_range_view(const float* image, const float* mask, int32_t C, int32_t H, int32_t W, int32_t pad,
                                 int32_t circular, float* out, rvStream stream) {
    RV_REQUIRE(image && out && pad >= 0 && W > 0, "rv_pad_range_view: bad argument");
    hipLaunchKernelGGL(pad_width_kernel, dim3(grid_for((int64_t)C * H * (W + 2 * pad))), dim3(256), 0, (hipStream_t)stream, image, mask,
                       C, H, W, pad, circular, out);
    RV_CHECK_LAUNCH("pad_width_kernel");
    return 0;
}

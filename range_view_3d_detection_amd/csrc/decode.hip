// decode.hip -- per-pixel box decoding + range-stratified candidate sampling (HBM-bound).
//
// Reference: RangeDecoder.decode (nn/decoders/range_decoder.py:46-76), sample_by_range (:127-156),
// decode_range_view / egovehicle_from_azimuth (math/ops/coding.py:79-144), yaw_to_quat
// (math/linalg/lie/SO3.py:122-134).  One thread per output candidate; the class loop reads
// NCHW logits with the W axis on the lanes (coalesced for rate-1 bands).  The box arithmetic is
// evaluated in fp64 and rounded once to fp32, exactly like the reference (coding.py:126-128,144).
#include "common.h"

namespace {

struct DecodeArgs {
    const float* logits;
    const float* reg;
    const float* cart;
    const uint8_t* mask;
    int B, n_cls, H, W;
    int az_inv;
    int n_bands;
    float lower[4], upper[4];
    int rate[4];
    int64_t band_off[5];  // candidate offset of each band inside one sweep
    int64_t cat_offset;
    float* scores;
    int64_t* cats;
    float* boxes;
};

__device__ __forceinline__ void decode_box(const float* reg, const float* cart, int64_t hw, int64_t pix, int az_inv,
                                           float* out7) {
    const double ox = reg[0 * hw + pix], oy = reg[1 * hw + pix], oz = reg[2 * hw + pix];
    const double l = exp((double)reg[3 * hw + pix]), w = exp((double)reg[4 * hw + pix]), hgt = exp((double)reg[5 * hw + pix]);
    double yaw = atan2((double)reg[6 * hw + pix], (double)reg[7 * hw + pix]);
    const double px = cart[0 * hw + pix], py = cart[1 * hw + pix], pz = cart[2 * hw + pix];
    double dx = ox, dy = oy;
    if (az_inv) {
        const double az = atan2(py, px);
        const double s = sin(az), c = cos(az);
        dx = c * ox - s * oy;
        dy = s * ox + c * oy;
        yaw += az;
    }
    out7[0] = (float)(px + dx);
    out7[1] = (float)(py + dy);
    out7[2] = (float)(pz + oz);
    out7[3] = (float)l;
    out7[4] = (float)w;
    out7[5] = (float)hgt;
    out7[6] = (float)yaw;
}

__global__ __launch_bounds__(256) void decode_candidates_kernel(const DecodeArgs a) {
    const int64_t K = a.band_off[a.n_bands > 0 ? a.n_bands : 1];
    const int64_t hw = (int64_t)a.H * a.W;
    const int64_t total = (int64_t)a.B * K;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / K);
        const int64_t k = i - (int64_t)b * K;
        int band = 0, h, w;
        if (a.n_bands > 0) {
            while (band + 1 < a.n_bands && k >= a.band_off[band + 1]) ++band;
            const int64_t r = k - a.band_off[band];
            const int ncols = (a.W + a.rate[band] - 1) / a.rate[band];
            h = (int)(r / ncols);
            w = (int)(r - (int64_t)h * ncols) * a.rate[band];
        } else {
            h = (int)(k / a.W);
            w = (int)(k - (int64_t)h * a.W);
        }
        const int64_t pix = (int64_t)h * a.W + w;
        const float m = a.mask[(int64_t)b * hw + pix] ? 1.f : 0.f;
        const float* lg = a.logits + (int64_t)b * a.n_cls * hw + pix;
        float best = -1.f;
        int best_c = 0;
        for (int c = 0; c < a.n_cls; ++c) {
            const float s = (1.f / (1.f + expf(-lg[(int64_t)c * hw]))) * m;
            if (s > best) {  // strict: ties keep the lowest class index (torch CPU max semantics)
                best = s;
                best_c = c;
            }
        }
        const float* cart = a.cart + (int64_t)b * 3 * hw;
        if (a.n_bands > 0) {
            const float x = cart[pix], y = cart[hw + pix], z = cart[2 * hw + pix];
            const float dist = sqrtf(x * x + y * y + z * z);
            if (!(dist > a.lower[band] && dist <= a.upper[band])) best = best * 0.f;
        }
        float box[7];
        decode_box(a.reg + (int64_t)b * 8 * hw, cart, hw, pix, a.az_inv, box);
        a.scores[i] = best;
        a.cats[i] = (int64_t)best_c + a.cat_offset;
#pragma unroll
        for (int j = 0; j < 7; ++j) a.boxes[i * 7 + j] = box[j];
    }
}

__global__ void decode_range_view_kernel(const float* reg, const float* cart, int B, int H, int W, int az_inv, float* out) {
    const int64_t hw = (int64_t)H * W, total = (int64_t)B * hw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / hw, pix = i - b * hw;
        float box[7];
        decode_box(reg + b * 8 * hw, cart + b * 3 * hw, hw, pix, az_inv, box);
#pragma unroll
        for (int j = 0; j < 7; ++j) out[(b * 7 + j) * hw + pix] = box[j];
    }
}

__global__ void yaw_to_quat_kernel(const float* yaw, int64_t n, int64_t stride, float* quat) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float half = yaw[i * stride] * 0.5f;
        quat[i * 4 + 0] = cosf(half);
        quat[i * 4 + 1] = 0.f;
        quat[i * 4 + 2] = 0.f;
        quat[i * 4 + 3] = sinf(half);
    }
}

int grid_for(int64_t work) {
    int64_t b = (work + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

}  // namespace

extern "C" int64_t rv_decode_num_candidates(int32_t H, int32_t W, int32_t n_bands, const int32_t* host_rates) {
    if (n_bands <= 0) return (int64_t)H * W;
    int64_t k = 0;
    for (int i = 0; i < n_bands; ++i) k += (int64_t)H * ((W + host_rates[i] - 1) / host_rates[i]);
    return k;
}

extern "C" int rv_decode_candidates(const float* logits, const float* regressands, const float* cart, const uint8_t* mask,
                                    int32_t B, int32_t n_cls, int32_t H, int32_t W, int32_t azimuth_invariant,
                                    int32_t n_bands, const float* lower, const float* upper, const int32_t* rates,
                                    int64_t category_offset, float* scores, int64_t* categories, float* boxes,
                                    rvStream stream) {
    RV_REQUIRE(logits && regressands && cart && mask && scores && categories && boxes, "rv_decode_candidates: null argument");
    RV_REQUIRE(n_bands >= 0 && n_bands <= 4, "rv_decode_candidates: at most 4 range bands (got %d)", n_bands);
    RV_REQUIRE(n_bands == 0 || (lower && upper && rates), "rv_decode_candidates: band tables missing");
    RV_REQUIRE(B > 0 && n_cls > 0 && H > 0 && W > 0, "rv_decode_candidates: empty input");
    DecodeArgs a;
    memset(&a, 0, sizeof(a));
    a.logits = logits;
    a.reg = regressands;
    a.cart = cart;
    a.mask = mask;
    a.B = B;
    a.n_cls = n_cls;
    a.H = H;
    a.W = W;
    a.az_inv = azimuth_invariant;
    a.n_bands = n_bands;
    a.band_off[0] = 0;
    if (n_bands == 0) a.band_off[1] = (int64_t)H * W;
    for (int i = 0; i < n_bands; ++i) {  // host pointers: three tiny tables, read here
        RV_REQUIRE(rates[i] >= 1, "rv_decode_candidates: subsampling rate must be >= 1");
        a.lower[i] = lower[i];
        a.upper[i] = upper[i];
        a.rate[i] = rates[i];
        a.band_off[i + 1] = a.band_off[i] + (int64_t)H * ((W + rates[i] - 1) / rates[i]);
    }
    a.cat_offset = category_offset;
    a.scores = scores;
    a.cats = categories;
    a.boxes = boxes;
    const int64_t K = a.band_off[n_bands > 0 ? n_bands : 1];
    hipLaunchKernelGGL(decode_candidates_kernel, dim3(grid_for((int64_t)B * K)), dim3(256), 0, (hipStream_t)stream, a);
    RV_CHECK_LAUNCH("decode_candidates_kernel");
    return 0;
}

extern "C" int rv_decode_range_view(const float* regressands, const float* cart, int32_t B, int32_t H, int32_t W,
                                    int32_t azimuth_invariant, float* out, rvStream stream) {
    RV_REQUIRE(regressands && cart && out, "rv_decode_range_view: null argument");
    hipLaunchKernelGGL(decode_range_view_kernel, dim3(grid_for((int64_t)B * H * W)), dim3(256), 0, (hipStream_t)stream,
                       regressands, cart, B, H, W, azimuth_invariant, out);
    RV_CHECK_LAUNCH("decode_range_view_kernel");
    return 0;
}

extern "C" int rv_yaw_to_quat(const float* yaw, int64_t n, int64_t yaw_stride, float* quat, rvStream stream) {
    if (n == 0) return 0;
    RV_REQUIRE(yaw && quat, "rv_yaw_to_quat: null argument");
    hipLaunchKernelGGL(yaw_to_quat_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, yaw, n, yaw_stride, quat);
    RV_CHECK_LAUNCH("yaw_to_quat_kernel");
    return 0;
}

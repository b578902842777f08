// loss.hip -- fused detection loss (forward reductions + analytic gradients), one thread per pixel.
//
// Reference: compute_classification_targets with GAUSSIAN affinity, k = inf, normalize_affinities =
// false (math/ops/assignment.py:76-161 -- the per-instance python loop reduces to a per-pixel map),
// varifocal_loss (nn/functional/__init__.py:8-27), DetectionHead.compute_*_loss and
// reduce_multiscale_loss (nn/heads/detection_head.py:309-449).  HBM-bound element-wise work:
// reads (n_cls + 8) logits/regressands + 3 + 8 + a few labels per pixel, writes the same number of
// gradients.  Reductions: registers -> wave shuffles -> one fp64 atomic per wave and quantity.
//
//   sums[0] = sum w_cls * VFL * mask            sums[1] = ... * foreground      sums[2] = ... * background
//   sums[3] = number of foreground pixels       sums[4..11] = sum of the un-normalised regression terms
//   sums[12] = max(total_objects, 1)            sums[13] = sums[3] + smoothing (total_fg)
// loss = sums[0]/sums[13] + (sums[4]+..+sums[11])/sums[12]   (assembled by the host wrapper, on device).
#include "common.h"

namespace {

struct LossArgs {
    const float* logits;  // NHWC, ld_logits
    const float* reg;     // NHWC, ld_reg
    const float* cart;    // NCHW (B,3,H,W)
    const uint8_t* mask;  // (B,H,W)
    const int64_t* labels;
    const int64_t* panoptics;
    const float* reg_targets;  // NCHW (B,8,H,W)
    const int64_t* ppo;
    const int32_t* num_objects;
    int B, n_cls, H, W, ld_logits, ld_reg;
    float coding[8];
    float cls_w, reg_w, smoothing, sigma, alpha, gamma;
    int az_inv;
    double* sums;
    float* soft;  // optional NCHW (B,n_cls,H,W)
    float* fg;    // optional (B,H,W)
    float* d_logits;
    float* d_reg;
    float grad_scale;
};

// centre of decode_range_view: fp64 arithmetic rounded to fp32 (math/ops/coding.py:126-144)
__device__ __forceinline__ void decode_centre(const float* r, float px, float py, float pz, int az_inv, float* c) {
    double dx = r[0], dy = r[1];
    if (az_inv) {
        const double az = atan2((double)py, (double)px);
        const double s = sin(az), co = cos(az);
        const double x = co * dx - s * dy, y = s * dx + co * dy;
        dx = x;
        dy = y;
    }
    c[0] = (float)((double)px + dx);
    c[1] = (float)((double)py + dy);
    c[2] = (float)((double)pz + (double)r[2]);
}

__device__ __forceinline__ float softplus(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

template <bool BACKWARD>
__global__ __launch_bounds__(256) void loss_kernel(const LossArgs a) {
    const int64_t hw = (int64_t)a.H * a.W, total = (int64_t)a.B * hw;
    double acc[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) acc[j] = 0.0;
    const double total_fg = BACKWARD ? a.sums[13] : 1.0;
    const double total_obj = BACKWARD ? a.sums[12] : 1.0;
    const float gscale = BACKWARD ? a.grad_scale * (float)a.sums[15] : 1.f;  // host factor x device factor (sums[15]: 1 unless the caller wrote the incoming gradient there)
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / hw, pix = i - b * hw;
        const float m = a.mask[i] ? 1.f : 0.f;
        const float* cart = a.cart + b * 3 * hw;
        const float px = cart[pix], py = cart[hw + pix], pz = cart[2 * hw + pix];
        float r[8], tg[8];
        {  // (rows are 128-byte aligned: stored channel counts are multiples of 32 -- two 16-byte loads instead of eight 4-byte ones)
            const f32x4 r0 = *(const f32x4*)(a.reg + i * a.ld_reg), r1 = *(const f32x4*)(a.reg + i * a.ld_reg + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                r[j] = r0[j];
                r[4 + j] = r1[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) tg[j] = a.reg_targets[(b * 8 + j) * hw + pix];
        const int64_t label = a.labels[i];
        const bool inst = a.panoptics[i] > 0;
        float aff = 0.f;
        if (inst) {
            float cp[3], cg[3];
            decode_centre(r, px, py, pz, 1, cp);  // predictions are always decoded azimuth-invariantly (assignment.py:112)
            decode_centre(tg, px, py, pz, a.az_inv, cg);
            const float dx = cp[0] - cg[0], dy = cp[1] - cg[1], dz = cp[2] - cg[2];
            aff = expf(-sqrtf(dx * dx + dy * dy + dz * dz) / (a.sigma * a.sigma));
        }
        const bool fg = aff != 0.f;
        const bool bg = !fg && m != 0.f;
        if (!BACKWARD && a.fg) a.fg[i] = fg ? 1.f : 0.f;
        // ---- classification: varifocal loss over the classes ----
        float cls_sum = 0.f;
        // A thread reads its pixel's logits: with 4-byte loads that is n_cls instructions of 64 lanes x 4 bytes, every lane on a
        // line of its own (128-byte rows), and the same again for the gradient stores -- the rows thrash the 32 KB L1 and the
        // kernel spent its time re-fetching lines.  Rows of 32 floats (the rv-* recipes: 26 / 3 classes) go through eight
        // 16-byte loads / stores and a fully unrolled class loop instead.
        const bool row32 = a.ld_logits == 32;
        f32x4 lv[8], gv[8];
        if (row32) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                lv[q] = *(const f32x4*)(a.logits + i * 32 + q * 4);
                gv[q] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            if (c >= (row32 ? a.n_cls : 0)) break;
            const float x = lv[c >> 2][c & 3];
            const float t = (label == c) ? aff : 0.f;
            if (!BACKWARD && a.soft) a.soft[(b * a.n_cls + c) * hw + pix] = t;
            const float e = expf(-fabsf(x));
            const float p = x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
            const float sp = fmaxf(x, 0.f) + log1pf(e);
            const float pg = a.gamma == 2.f ? p * p : powf(p, a.gamma);
            if (!BACKWARD) {
                const float bce = sp - x * t;
                cls_sum += t > 0.f ? t * bce : a.alpha * pg * bce;
            } else {
                const float g = t > 0.f ? t * (p - t) : a.alpha * pg * (a.gamma * (1.f - p) * sp + p);
                gv[c >> 2][c & 3] = (float)((double)(g * a.cls_w * m) / total_fg) * gscale;
            }
        }
        if (BACKWARD && row32) {
#pragma unroll
            for (int q = 0; q < 8; ++q) *(f32x4*)(a.d_logits + i * 32 + q * 4) = gv[q];
        }
        for (int c = row32 ? a.n_cls : 0; c < a.n_cls; ++c) {  // (other row lengths: the scalar loop)
            const float x = a.logits[i * a.ld_logits + c];
            const float t = (label == c) ? aff : 0.f;
            if (!BACKWARD && a.soft) a.soft[(b * a.n_cls + c) * hw + pix] = t;
            // one exponential per class serves the sigmoid and the softplus; p^gamma is a product for the recipe's gamma = 2
            // (the transcendental functions, not the 34 floats per pixel, are what this kernel's time goes to)
            const float e = expf(-fabsf(x));
            const float p = x >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
            const float sp = fmaxf(x, 0.f) + log1pf(e);
            const float pg = a.gamma == 2.f ? p * p : powf(p, a.gamma);
            if (!BACKWARD) {
                const float bce = sp - x * t;  // BCE-with-logits
                const float l = t > 0.f ? t * bce : a.alpha * pg * bce;
                cls_sum += l;
            } else {
                float g;
                if (t > 0.f)
                    g = t * (p - t);
                else
                    g = a.alpha * pg * (a.gamma * (1.f - p) * sp + p);
                a.d_logits[i * a.ld_logits + c] = (float)((double)(g * a.cls_w * m) / total_fg) * gscale;
            }
        }
        // ---- regression: L1 with the per-object normaliser in fp64 ----
        const bool reg_on = label < a.n_cls;
        const double norm = 1.0 / ((double)a.ppo[i] + (double)a.smoothing);
        if (!BACKWARD) {
            const double v = (double)(cls_sum * a.cls_w * m);
            acc[0] += v;
            if (fg) acc[1] += v;
            if (bg) acc[2] += v;
            if (fg) acc[3] += 1.0;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (reg_on) acc[4 + j] += (double)(fabsf(r[j] - tg[j]) * a.reg_w) * norm * (double)m * (double)a.coding[j] / 8.0;
        } else {
            float dr[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float d = r[j] - tg[j];
                const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
                const double g = reg_on ? (double)(sgn * a.reg_w) * norm * (double)m * (double)a.coding[j] / 8.0 / total_obj : 0.0;
                dr[j] = (float)g * gscale;
            }
            *(f32x4*)(a.d_reg + i * a.ld_reg) = f32x4{dr[0], dr[1], dr[2], dr[3]};
            *(f32x4*)(a.d_reg + i * a.ld_reg + 4) = f32x4{dr[4], dr[5], dr[6], dr[7]};
        }
    }
    if (!BACKWARD) {
        // waves -> workgroup through LDS, then ONE fp64 atomic per workgroup and quantity (one per WAVE was 98 k atomics on
        // twelve addresses: the forward pass spent most of its time queueing at them)
        __shared__ double red[4][12];
#pragma unroll
        for (int j = 0; j < 12; ++j) {
            const double s = wave_sum_d(acc[j]);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][j] = s;
        }
        __syncthreads();
        if (threadIdx.x < 12) {
            const double s = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
            if (s != 0.0) atomicAdd(&a.sums[threadIdx.x], s);
        }
    }
}

__global__ void loss_finish_kernel(double* sums, const int32_t* num_objects, float smoothing) {
    const double n = (double)*num_objects;
    const double obj = n < 1.0 ? 1.0 : n, fg = sums[3] + (double)smoothing;
    sums[12] = obj;
    sums[13] = fg;
    sums[15] = 1.0;  // device-side factor of the backward pass (the caller copies the incoming gradient of the loss here: no host round trip)
    // the scalars detection_head.py:379-449 reports, formed here instead of by a dozen one-element launches on the host side
    const double coord = (sums[4] + sums[5] + sums[6]) / obj, dim = (sums[7] + sums[8] + sums[9]) / obj, rot = (sums[10] + sums[11]) / obj;
    const double cls = sums[0] / fg, reg = (((((((sums[4] + sums[5]) + sums[6]) + sums[7]) + sums[8]) + sums[9]) + sums[10]) + sums[11]) / obj;
    sums[16] = cls + reg;  // loss
    sums[17] = cls;
    sums[18] = sums[1] / fg;
    sums[19] = sums[2] / fg;
    sums[20] = coord;
    sums[21] = dim;
    sums[22] = rot;
    sums[23] = coord + dim + rot;
}

int fill(LossArgs* a, const float* logits, int32_t ld_logits, const float* reg, int32_t ld_reg, const float* cart,
         const uint8_t* mask, const int64_t* labels, const int64_t* panoptics, const float* reg_targets,
         const int64_t* ppo, const int32_t* num_objects, int32_t B, int32_t n_cls, int32_t H, int32_t W,
         const float* host_coding_weights, float cls_w, float reg_w, float smoothing, float sigma, float alpha, float gamma,
         int32_t az_inv, double* sums) {
    RV_REQUIRE(logits && reg && cart && mask && labels && panoptics && reg_targets && ppo && num_objects && sums && host_coding_weights,
               "rv_detection_loss: null argument");
    RV_REQUIRE(ld_logits >= n_cls && ld_reg >= 8 && ld_reg % 4 == 0, "rv_detection_loss: bad strides (rows of regressands must be 16-byte aligned)");
    memset(a, 0, sizeof(*a));
    a->logits = logits;
    a->reg = reg;
    a->cart = cart;
    a->mask = mask;
    a->labels = labels;
    a->panoptics = panoptics;
    a->reg_targets = reg_targets;
    a->ppo = ppo;
    a->num_objects = num_objects;
    a->B = B;
    a->n_cls = n_cls;
    a->H = H;
    a->W = W;
    a->ld_logits = ld_logits;
    a->ld_reg = ld_reg;
    for (int j = 0; j < 8; ++j) a->coding[j] = host_coding_weights[j];
    a->cls_w = cls_w;
    a->reg_w = reg_w;
    a->smoothing = smoothing;
    a->sigma = sigma;
    a->alpha = alpha;
    a->gamma = gamma;
    a->az_inv = az_inv;
    a->sums = sums;
    a->grad_scale = 1.f;
    return 0;
}

int grid_for(int64_t work) {
    int64_t b = (work + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" int rv_detection_loss_forward(const float* logits, int32_t ld_logits, const float* regressands, int32_t ld_reg,
                                         const float* cart, const uint8_t* mask, const int64_t* labels,
                                         const int64_t* panoptics, const float* reg_targets, const int64_t* points_per_obj,
                                         const int32_t* num_objects, int32_t B, int32_t n_cls, int32_t H, int32_t W,
                                         const float* host_coding_weights, float cls_weight, float reg_weight,
                                         float smoothing, float sigma, float alpha, float gamma, int32_t azimuth_invariant,
                                         double* sums, float* soft_targets, float* foreground, rvStream stream) {
    LossArgs a;
    if (fill(&a, logits, ld_logits, regressands, ld_reg, cart, mask, labels, panoptics, reg_targets, points_per_obj,
             num_objects, B, n_cls, H, W, host_coding_weights, cls_weight, reg_weight, smoothing, sigma, alpha, gamma,
             azimuth_invariant, sums))
        return 1;
    a.soft = soft_targets;
    a.fg = foreground;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(sums, 0, 24 * sizeof(double), st);
    if (e != hipSuccess) RV_FAIL("rv_detection_loss_forward: %s", hipGetErrorString(e));
    const int fwd_grid = grid_for((int64_t)B * H * W) < 512 ? grid_for((int64_t)B * H * W) : 512;  // (grid-stride: fewer, longer workgroups -> fewer atomics)
    hipLaunchKernelGGL(loss_kernel<false>, dim3(fwd_grid), dim3(256), 0, st, a);
    hipLaunchKernelGGL(loss_finish_kernel, dim3(1), dim3(1), 0, st, sums, num_objects, smoothing);
    RV_CHECK_LAUNCH("loss forward kernels");
    return 0;
}

extern "C" int rv_detection_loss_backward(const float* logits, int32_t ld_logits, const float* regressands, int32_t ld_reg,
                                          const float* cart, const uint8_t* mask, const int64_t* labels,
                                          const int64_t* panoptics, const float* reg_targets, const int64_t* points_per_obj,
                                          const int32_t* num_objects, int32_t B, int32_t n_cls, int32_t H, int32_t W,
                                          const float* host_coding_weights, float cls_weight, float reg_weight,
                                          float smoothing, float sigma, float alpha, float gamma, int32_t azimuth_invariant,
                                          const double* sums, float grad_scale, float* d_logits, float* d_regressands,
                                          rvStream stream) {
    LossArgs a;
    if (fill(&a, logits, ld_logits, regressands, ld_reg, cart, mask, labels, panoptics, reg_targets, points_per_obj,
             num_objects, B, n_cls, H, W, host_coding_weights, cls_weight, reg_weight, smoothing, sigma, alpha, gamma,
             azimuth_invariant, (double*)sums))
        return 1;
    RV_REQUIRE(d_logits && d_regressands, "rv_detection_loss_backward: null gradient buffers");
    a.d_logits = d_logits;
    a.d_reg = d_regressands;
    a.grad_scale = grad_scale;
    hipLaunchKernelGGL(loss_kernel<true>, dim3(grid_for((int64_t)B * H * W)), dim3(256), 0, (hipStream_t)stream, a);
    RV_CHECK_LAUNCH("loss backward kernel");
    return 0;
}

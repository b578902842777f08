// augment.hip -- the loader's range-view augmentations on device (HBM-bound, one read + one write per element).
//
// Reference: prototype/loader.py:825-990 (flip_azimuth, random_rotation, random_global_scale, random_global_translation).
// There every augmentation is a pass over a polars table on the host: all columns are re-ordered (np.flip / np.roll along
// the azimuth axis), then the x / y / z (and range) columns are rewritten.  A chain of them is
//   * a column map  w_src = (a * w + b) mod W,  a = +-1          (flips and rolls compose to this form: integer work, exact);
//   * an affine map of the Cartesian channels  xyz' = A xyz + t  (fp64 here, rounded once to fp32; the reference rounds
//     between steps, differences are a few fp32 ulps: tolerance 1e-6 of the channel maximum in the tests);
//   * range' = ||A_r xyz + t_r||  as of the LAST random_global_scale of the chain (the reference recomputes the range
//     there and only there -- a later translation leaves it stale), else the range passes through.
// The host composes (a, b, A, t, A_r, t_r) per sweep (prototype/loader.py of this package) and one launch applies them to a
// whole batch tensor: features (B,F,H,W) with the positions of its x / y / z / range channels, cart (B,3,H,W), mask (B,1,H,W).
#include "common.h"

namespace {

struct AugParams {  // per sweep, 32 doubles
    double a, b;    // column map
    double A[9];    // row-major
    double t[3];
    double Ar[9];
    double tr[3];
    double use_range;  // != 0: range channel := ||Ar xyz + tr||
    double pad[5];
};

__global__ void augment_kernel(const float* in, float* out, int B, int C, int H, int W, int ix, int iy, int iz, int ir,
                               const AugParams* params) {
    const int64_t hw = (int64_t)H * W, total = (int64_t)B * hw;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / hw);
        const int64_t p = i - b * hw;
        const int h = (int)(p / W), w = (int)(p - (int64_t)h * W);
        const AugParams& q = params[b];
        int ws = ((int)q.a * w + (int)q.b) % W;
        if (ws < 0) ws += W;
        const float* src = in + ((int64_t)b * C) * hw + (int64_t)h * W + ws;
        float* dst = out + ((int64_t)b * C) * hw + p;
        double x = 0.0, y = 0.0, z = 0.0;
        if (ix >= 0) {
            x = (double)src[(int64_t)ix * hw];
            y = (double)src[(int64_t)iy * hw];
            z = (double)src[(int64_t)iz * hw];
        }
        for (int c = 0; c < C; ++c) {
            float v = src[(int64_t)c * hw];
            if (c == ix) v = (float)(q.A[0] * x + q.A[1] * y + q.A[2] * z + q.t[0]);
            else if (c == iy) v = (float)(q.A[3] * x + q.A[4] * y + q.A[5] * z + q.t[1]);
            else if (c == iz) v = (float)(q.A[6] * x + q.A[7] * y + q.A[8] * z + q.t[2]);
            else if (c == ir && q.use_range != 0.0) {
                const double rx = q.Ar[0] * x + q.Ar[1] * y + q.Ar[2] * z + q.tr[0];
                const double ry = q.Ar[3] * x + q.Ar[4] * y + q.Ar[5] * z + q.tr[1];
                const double rz = q.Ar[6] * x + q.Ar[7] * y + q.Ar[8] * z + q.tr[2];
                v = (float)sqrt(rx * rx + ry * ry + rz * rz);
            }
            dst[(int64_t)c * hw] = v;
        }
    }
}

}  // namespace

extern "C" int rv_augment(const float* in, float* out, int32_t B, int32_t C, int32_t H, int32_t W, int32_t ix, int32_t iy,
                          int32_t iz, int32_t irange, const double* params, rvStream stream) {
    RV_REQUIRE(in && out && params && in != out, "rv_augment: null or aliased argument");
    RV_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "rv_augment: empty tensor");
    RV_REQUIRE((ix < 0 && iy < 0 && iz < 0) || (ix >= 0 && iy >= 0 && iz >= 0 && ix < C && iy < C && iz < C),
               "rv_augment: the x / y / z channel indices must be all given or all -1");
    RV_REQUIRE(irange < C && (irange < 0 || ix >= 0), "rv_augment: bad range channel");
    const int64_t total = (int64_t)B * H * W;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(augment_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, in, out, B, C, H, W, ix, iy, iz,
                       irange, (const AugParams*)params);
    RV_CHECK_LAUNCH("augment_kernel");
    return 0;
}

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

// ---- the loader's per-sweep contract (prototype/loader.py:568-705): table columns -> features / cart / mask ------------------
struct TableArgs {
    const float* table;  // [n_cols][hw] fp32, column-major like the feather table
    int64_t hw;
    int32_t n_feat;
    int32_t feat_col[16], feat_op[16];  // op: 0 copy, 1 tanh (Waymo intensity), 2 x 1e-9 (timedelta_ns)
    int32_t cart_col[3];
    int32_t range_col, roi_col;  // roi_col < 0: no ROI filter
    float* features;             // [n_feat][hw]
    float* cart;                 // [3][hw]
    uint8_t* mask;               // [hw], range > 0
};

__global__ void table_to_range_view_kernel(const TableArgs a) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < a.hw; i += (int64_t)gridDim.x * blockDim.x) {
        // `sweep * sweep["is_within_roi"]` (loader.py:599-601) multiplies EVERY column by the 0 / 1 flag
        const float roi = a.roi_col >= 0 ? (a.table[(int64_t)a.roi_col * a.hw + i] != 0.f ? 1.f : 0.f) : 1.f;
        for (int f = 0; f < a.n_feat; ++f) {
            float v = a.table[(int64_t)a.feat_col[f] * a.hw + i] * roi;
            if (a.feat_op[f] == 1) v = tanhf(v);
            else if (a.feat_op[f] == 2) v = (float)((double)v * 1e-9);
            a.features[(int64_t)f * a.hw + i] = v;
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) a.cart[(int64_t)c * a.hw + i] = a.table[(int64_t)a.cart_col[c] * a.hw + i] * roi;
        a.mask[i] = a.table[(int64_t)a.range_col * a.hw + i] * roi > 0.f ? 1 : 0;
    }
}

}  // namespace

extern "C" int rv_table_to_range_view(const float* table, int32_t n_cols, int64_t hw, int32_t n_feat, const int32_t* host_feat_col,
                                      const int32_t* host_feat_op, const int32_t* host_cart_col, int32_t range_col, int32_t roi_col,
                                      float* features, float* cart, uint8_t* mask, rvStream stream) {
    RV_REQUIRE(table && host_feat_col && host_feat_op && host_cart_col && features && cart && mask, "rv_table_to_range_view: null argument");
    RV_REQUIRE(n_feat >= 1 && n_feat <= 16, "rv_table_to_range_view: %d feature columns (1..16)", n_feat);
    RV_REQUIRE(hw > 0 && n_cols > 0 && range_col >= 0 && range_col < n_cols && roi_col < n_cols, "rv_table_to_range_view: bad table shape / column index");
    TableArgs a;
    a.table = table;
    a.hw = hw;
    a.n_feat = n_feat;
    for (int f = 0; f < n_feat; ++f) {
        RV_REQUIRE(host_feat_col[f] >= 0 && host_feat_col[f] < n_cols && host_feat_op[f] >= 0 && host_feat_op[f] <= 2, "rv_table_to_range_view: bad feature column %d", f);
        a.feat_col[f] = host_feat_col[f];
        a.feat_op[f] = host_feat_op[f];
    }
    for (int c = 0; c < 3; ++c) {
        RV_REQUIRE(host_cart_col[c] >= 0 && host_cart_col[c] < n_cols, "rv_table_to_range_view: bad Cartesian column %d", c);
        a.cart_col[c] = host_cart_col[c];
    }
    a.range_col = range_col;
    a.roi_col = roi_col;
    a.features = features;
    a.cart = cart;
    a.mask = mask;
    int64_t blocks = (hw + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(table_to_range_view_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a);
    RV_CHECK_LAUNCH("table_to_range_view_kernel");
    return 0;
}

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

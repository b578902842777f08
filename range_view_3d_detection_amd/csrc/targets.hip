// targets.hip -- dense target assignment on device (no per-instance host loop, no host syncs).
//
// Reference: compute_targets (nn/heads/detection_head.py:496-665) with cuboids_to_vertices /
// compute_interior_points_mask (math/polytope.py:14-107) and encode_regression_targets
// (detection_head.py:460-493), for the layout the rv-* configs use (one stride-1 FPN level, one
// task).  Numerics follow the reference: box vertices in fp32 through the yaw-only quaternion,
// the three slab tests in fp64, the centre offset / azimuth rotation in fp32, log / sin / cos of
// the box parameters in fp64 rounded to fp32.
//
//   pass 1  per (pixel, box of the pixel's sweep): inside test -> per-box interior-point counts
//           (wave ballot + one atomicAdd per wave and box)
//   pass 2  per sweep: rank boxes by (count ascending, original order) == the reference's stable sort
//   pass 3  per pixel: the containing box of smallest rank wins -> label, panoptic id (rank+1),
//           points_per_obj, regression targets; marks the box as owning a pixel
//   pass 4  num_objects = number of boxes that own >= 1 pixel (== sum over sweeps of the distinct
//           non-background panoptic ids, detection_head.py:379-390)
#include "common.h"

namespace {

struct BoxPlanes {
    double uvw[3][3];
    double lo[3], hi[3];  // the slab [min(d_ref,d_cor), max(d_ref,d_cor)] of each axis
};

__device__ void make_planes(const double* cub, BoxPlanes* bp) {
    const float cx = (float)cub[0], cy = (float)cub[1], cz = (float)cub[2];
    const float hl = (float)cub[3] / 2.0f, hw = (float)cub[4] / 2.0f, hh = (float)cub[5] / 2.0f;
    const float half = (float)cub[6] * 0.5f;
    float qw = cosf(half), qz = sinf(half);
    const float nrm = sqrtf(qw * qw + qz * qz);
    qw /= nrm;
    qz /= nrm;
    const float r00 = 1.f - 2.f * (qz * qz), r01 = 2.f * (0.f - qw * qz), r10 = 2.f * (qw * qz), r11 = r00;
    // unit vertices: 1 (+,-,+)  2 (+,-,-)  3 (+,+,-)  6 (-,-,-)   (math/polytope.py:79-91)
    const float ux[4] = {+1.f, +1.f, +1.f, -1.f};
    const float uy[4] = {-1.f, -1.f, +1.f, -1.f};
    const float uz[4] = {+1.f, -1.f, -1.f, -1.f};
    double v[4][3];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float ox = hl * ux[k], oy = hw * uy[k], oz = hh * uz[k];
        v[k][0] = (double)(ox * r00 + oy * r01 + cx);
        v[k][1] = (double)(ox * r10 + oy * r11 + cy);
        v[k][2] = (double)(oz + cz);
    }
    // reference vertex = 2; corners = vertices [6, 3, 1]
    const int corner[3] = {3, 2, 0};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double dref = 0.0, dcor = 0.0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double u = v[1][j] - v[corner[a]][j];
            bp->uvw[a][j] = u;
            dref += u * v[1][j];
            dcor += u * v[corner[a]][j];
        }
        bp->lo[a] = dref < dcor ? dref : dcor;
        bp->hi[a] = dref < dcor ? dcor : dref;
    }
}

__device__ __forceinline__ bool inside(const BoxPlanes& bp, double x, double y, double z) {
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double d = bp.uvw[a][0] * x + bp.uvw[a][1] * y + bp.uvw[a][2] * z;
        if (!(bp.lo[a] <= d && d <= bp.hi[a])) return false;
    }
    return true;
}

constexpr int kBoxTile = 64;

template <bool ASSIGN>
__global__ __launch_bounds__(256) void box_pixel_kernel(const double* cuboids, const int32_t* box_offsets, const float* cart,
                                                        int H, int W, int n_cls, int az_inv, int32_t* counts,
                                                        const int32_t* rank, int32_t* owned, int64_t* labels,
                                                        int64_t* panoptics, float* reg, int64_t* ppo) {
    __shared__ BoxPlanes planes[kBoxTile];
    const int b = blockIdx.y;
    const int64_t hw = (int64_t)H * W;
    const int64_t pix = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = pix < hw;
    const float* c0 = cart + (int64_t)b * 3 * hw;
    const float pxf = valid ? c0[pix] : 0.f, pyf = valid ? c0[hw + pix] : 0.f, pzf = valid ? c0[2 * hw + pix] : 0.f;
    const double px = pxf, py = pyf, pz = pzf;
    const int m0 = box_offsets[b], m1 = box_offsets[b + 1];
    int best_rank = 0x7fffffff, best_box = -1;
    for (int t0 = m0; t0 < m1; t0 += kBoxTile) {
        const int nt = (m1 - t0) < kBoxTile ? (m1 - t0) : kBoxTile;
        __syncthreads();
        if ((int)threadIdx.x < nt) make_planes(cuboids + (int64_t)(t0 + threadIdx.x) * 10, &planes[threadIdx.x]);
        __syncthreads();
        for (int k = 0; k < nt; ++k) {
            const bool in = valid && inside(planes[k], px, py, pz);
            if (!ASSIGN) {
                const unsigned long long bal = __ballot(in);
                if ((threadIdx.x & 63) == 0 && bal) atomicAdd(&counts[t0 + k], __popcll(bal));
            } else if (in) {
                const int r = rank[t0 + k];
                if (r < best_rank) {
                    best_rank = r;
                    best_box = t0 + k;
                }
            }
        }
    }
    if (!ASSIGN || !valid) return;
    const int64_t o = (int64_t)b * hw + pix;
    if (best_box < 0) {
        labels[o] = n_cls;
        panoptics[o] = 0;
        ppo[o] = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) reg[((int64_t)b * 8 + j) * hw + pix] = 0.f;
        return;
    }
    const double* cub = cuboids + (int64_t)best_box * 10;
    labels[o] = (int64_t)cub[8];
    panoptics[o] = best_rank + 1;
    ppo[o] = counts[best_box];
    owned[best_box] = 1;
    float ox = (float)cub[0] - pxf, oy = (float)cub[1] - pyf;
    const float oz = (float)cub[2] - pzf;
    double rots = cub[6];
    if (az_inv) {
        const float az = atan2f(pyf, pxf);
        rots -= (double)az;
        const float c = cosf(az), s = sinf(az);
        const float x1 = c * ox + s * oy, x2 = -s * ox + c * oy;
        ox = x1;
        oy = x2;
    }
    float* r = reg + (int64_t)b * 8 * hw + pix;
    r[0] = ox;
    r[hw] = oy;
    r[2 * hw] = oz;
    r[3 * hw] = (float)log(cub[3]);
    r[4 * hw] = (float)log(cub[4]);
    r[5 * hw] = (float)log(cub[5]);
    r[6 * hw] = (float)sin(rots);
    r[7 * hw] = (float)cos(rots);
}

__global__ void rank_kernel(const int32_t* box_offsets, const int32_t* counts, int32_t* rank) {
    const int b = blockIdx.x;
    const int m0 = box_offsets[b], m1 = box_offsets[b + 1];
    for (int i = m0 + threadIdx.x; i < m1; i += blockDim.x) {
        const int ci = counts[i];
        int r = 0;
        for (int j = m0; j < m1; ++j) {
            const int cj = counts[j];
            r += (cj < ci) || (cj == ci && j < i);
        }
        rank[i] = r;
    }
}

__global__ void count_owned_kernel(const int32_t* owned, int m, int32_t* num_objects) {
    int s = 0;
    for (int i = threadIdx.x; i < m; i += blockDim.x) s += owned[i] != 0;
    s = (int)wave_sum((float)s);
    __shared__ int part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *num_objects = part[0] + part[1] + part[2] + part[3];
}

}  // namespace

extern "C" int rv_assign_targets(const double* cuboids, int32_t m, const int32_t* box_offsets, const float* cart, int32_t B,
                                 int32_t H, int32_t W, int32_t n_cls, int32_t azimuth_invariant, int32_t* counts,
                                 int32_t* order, int32_t* owned, int64_t* labels, int64_t* panoptics, float* reg_targets,
                                 int64_t* points_per_obj, int32_t* num_objects, rvStream stream) {
    RV_REQUIRE(box_offsets && cart && labels && panoptics && reg_targets && points_per_obj && num_objects, "rv_assign_targets: null argument");
    RV_REQUIRE(m == 0 || (cuboids && counts && order && owned), "rv_assign_targets: null box buffers");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(num_objects, 0, sizeof(int32_t), st);
    if (m > 0 && e == hipSuccess) e = hipMemsetAsync(counts, 0, sizeof(int32_t) * m, st);
    if (m > 0 && e == hipSuccess) e = hipMemsetAsync(owned, 0, sizeof(int32_t) * m, st);
    if (e != hipSuccess) RV_FAIL("rv_assign_targets: %s", hipGetErrorString(e));
    const dim3 grid(rv_ceil_div((int64_t)H * W, 256), B);
    if (m > 0) {
        hipLaunchKernelGGL(box_pixel_kernel<false>, grid, dim3(256), 0, st, cuboids, box_offsets, cart, H, W, n_cls,
                           azimuth_invariant, counts, (const int32_t*)nullptr, owned, labels, panoptics, reg_targets,
                           points_per_obj);
        hipLaunchKernelGGL(rank_kernel, dim3(B), dim3(128), 0, st, box_offsets, counts, order);
    }
    hipLaunchKernelGGL(box_pixel_kernel<true>, grid, dim3(256), 0, st, cuboids, box_offsets, cart, H, W, n_cls,
                       azimuth_invariant, counts, (const int32_t*)order, owned, labels, panoptics, reg_targets, points_per_obj);
    if (m > 0) hipLaunchKernelGGL(count_owned_kernel, dim3(1), dim3(256), 0, st, owned, m, num_objects);
    RV_CHECK_LAUNCH("rv_assign_targets kernels");
    return 0;
}

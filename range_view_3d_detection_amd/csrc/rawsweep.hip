// rawsweep.hip -- from a raw (ego-frame, motion-compensated) sweep to the points the range-image projection bins:
// per-point un-motion-compensation, laser-row correction, ego -> sensor rigid transform.  fp64 / integer per-point work,
// HBM-bound (one read + one write per point).  Reference: converters/av2/utils.py:211-295 and :32-55.
//
//   unmotion_compensate: t_i = sweep timestamp + offset_ns[i]; points with t_i outside (min pose ts, max pose ts) are
//     dropped (kept[i] = 0); pose of the ego vehicle at t_i: rotation by Slerp between the bracketing poses
//     (j = searchsorted_left(pose_ts, t_i): poses j-1 and j, alpha = (t_i - ts[j-1]) / (ts[j] - ts[j-1]);
//     R = R_{j-1} exp(alpha log(R_{j-1}^-1 R_j))), translation t_{j-1} * alpha + (1 - alpha) * t_j -- the reference's weights,
//     swapped as they are (:275-276); the point is moved from the target pose's frame into the per-point frame:
//     p' = R_p^T (R_target p + t_target - t_p).
//   correct_laser_numbers: logs with the known laser-ordering defect remap ids through LASER_MAPPING (upper block first,
//     then every id < 32 -- in that order, :214-220), then the id -> image-row table (ROW_MAPPING_64 / _32).
//   ego -> sensor: p_sensor = R_ext^T (p - t_ext)   (sensor_SE3_egovehicle = inverse of egovehicle_SE3_sensor, :47-55).
#include "common.h"

namespace {

struct Q { double w, x, y, z; };

__device__ __forceinline__ Q qmul(const Q& a, const Q& b) {
    return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
            a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ Q qload(const double* p) {
    Q q = {p[0], p[1], p[2], p[3]};
    const double n = 1.0 / sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    q.w *= n; q.x *= n; q.y *= n; q.z *= n;
    return q;
}
// v' = R(q) v
__device__ __forceinline__ void qrot(const Q& q, const double v[3], double out[3]) {
    const double xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z, xy = q.x * q.y, xz = q.x * q.z, yz = q.y * q.z;
    const double wx = q.w * q.x, wy = q.w * q.y, wz = q.w * q.z;
    out[0] = (1 - 2 * (yy + zz)) * v[0] + 2 * (xy - wz) * v[1] + 2 * (xz + wy) * v[2];
    out[1] = 2 * (xy + wz) * v[0] + (1 - 2 * (xx + zz)) * v[1] + 2 * (yz - wx) * v[2];
    out[2] = 2 * (xz - wy) * v[0] + 2 * (yz + wx) * v[1] + (1 - 2 * (xx + yy)) * v[2];
}

__global__ void unmotion_kernel(const double* xyz, const int32_t* offset_ns, int64_t n, int64_t sweep_ts, const int64_t* pose_ts,
                                const double* pose_q, const double* pose_t, int P, int target, double* xyz_p, uint8_t* kept) {
    const int64_t ts_min = pose_ts[0], ts_max = pose_ts[P - 1];  // sorted
    const Q qt = qload(pose_q + 4 * target);
    const double tt[3] = {pose_t[3 * target], pose_t[3 * target + 1], pose_t[3 * target + 2]};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ts = sweep_ts + (int64_t)offset_ns[i];
        const bool ok = ts > ts_min && ts < ts_max;
        kept[i] = ok ? 1 : 0;
        if (!ok) {
            xyz_p[3 * i] = xyz_p[3 * i + 1] = xyz_p[3 * i + 2] = 0.0;
            continue;
        }
        int lo = 0, hi = P;  // searchsorted, side = left: first j with pose_ts[j] >= ts
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (pose_ts[mid] < ts) lo = mid + 1; else hi = mid;
        }
        const int j = lo;  // 1 <= j <= P-1 for kept points
        const double alpha = (double)(ts - pose_ts[j - 1]) / (double)(pose_ts[j] - pose_ts[j - 1]);
        const Q q0 = qload(pose_q + 4 * (j - 1)), q1 = qload(pose_q + 4 * j);
        Q rel = qmul(Q{q0.w, -q0.x, -q0.y, -q0.z}, q1);
        if (rel.w < 0) { rel.w = -rel.w; rel.x = -rel.x; rel.y = -rel.y; rel.z = -rel.z; }  // the shorter arc
        const double vn = sqrt(rel.x * rel.x + rel.y * rel.y + rel.z * rel.z);
        const double half = atan2(vn, rel.w) * alpha;  // (angle / 2) * alpha
        const double sc = vn > 0.0 ? sin(half) / vn : 0.0;
        const Q qp = qmul(q0, Q{cos(half), rel.x * sc, rel.y * sc, rel.z * sc});
        double tp[3];
        for (int k = 0; k < 3; ++k) tp[k] = pose_t[3 * (j - 1) + k] * alpha + (1.0 - alpha) * pose_t[3 * j + k];  // (sic)
        const double p[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
        double w[3], o[3];
        qrot(qt, p, w);  // into the city frame with the target pose
        for (int k = 0; k < 3; ++k) w[k] += tt[k] - tp[k];
        qrot(Q{qp.w, -qp.x, -qp.y, -qp.z}, w, o);  // back with the inverse of the per-point pose
        xyz_p[3 * i] = o[0];
        xyz_p[3 * i + 1] = o[1];
        xyz_p[3 * i + 2] = o[2];
    }
}

__global__ void laser_rows_kernel(const int32_t* laser, int64_t n, int affected, const int32_t* laser_mapping, const int32_t* row_mapping,
                                  int n_rows, int32_t* out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int l = laser[i];
        if (affected) {
            if (l >= 32) l = laser_mapping[l - 32] + 32;  // the upper block first ...
            if (l < 32) l = laser_mapping[l];             // ... then every id below 32 (ids moved by the first step stay >= 32)
        }
        out[i] = (l >= 0 && l < n_rows) ? row_mapping[l] : -1;
    }
}

__global__ void se3_inverse_kernel(const double* xyz, int64_t n, const double* q_wxyz, const double* t, const uint8_t* kept, double* out) {
    const Q q = qload(q_wxyz);
    const Q qi = {q.w, -q.x, -q.y, -q.z};
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double d[3] = {xyz[3 * i] - t[0], xyz[3 * i + 1] - t[1], xyz[3 * i + 2] - t[2]};
        double o[3];
        qrot(qi, d, o);
        if (kept && !kept[i]) o[0] = o[1] = o[2] = 0.0;  // dropped point: range 0, which the z-buffer skips (min range 1 m)
        out[3 * i] = o[0];
        out[3 * i + 1] = o[1];
        out[3 * i + 2] = o[2];
    }
}

int grid_for(int64_t work) {
    int64_t b = (work + 255) / 256;
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" int rv_unmotion_compensate(const double* xyz, const int32_t* offset_ns, int64_t n, int64_t sweep_timestamp_ns,
                                      const int64_t* pose_timestamps_ns, const double* pose_q_wxyz, const double* pose_t, int32_t n_poses,
                                      int32_t target_pose, double* xyz_p, uint8_t* kept, rvStream stream) {
    if (n == 0) return 0;
    RV_REQUIRE(xyz && offset_ns && pose_timestamps_ns && pose_q_wxyz && pose_t && xyz_p && kept, "rv_unmotion_compensate: null argument");
    RV_REQUIRE(n_poses >= 2 && target_pose >= 0 && target_pose < n_poses, "rv_unmotion_compensate: need >= 2 poses and a target pose among them");
    hipLaunchKernelGGL(unmotion_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, xyz, offset_ns, n, sweep_timestamp_ns,
                       pose_timestamps_ns, pose_q_wxyz, pose_t, n_poses, target_pose, xyz_p, kept);
    RV_CHECK_LAUNCH("unmotion_kernel");
    return 0;
}

extern "C" int rv_correct_laser_numbers(const int32_t* laser, int64_t n, int32_t affected, const int32_t* laser_mapping_32,
                                        const int32_t* row_mapping, int32_t n_rows, int32_t* out, rvStream stream) {
    if (n == 0) return 0;
    RV_REQUIRE(laser && row_mapping && out && (!affected || laser_mapping_32), "rv_correct_laser_numbers: null argument");
    hipLaunchKernelGGL(laser_rows_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, laser, n, affected, laser_mapping_32, row_mapping,
                       n_rows, out);
    RV_CHECK_LAUNCH("laser_rows_kernel");
    return 0;
}

extern "C" int rv_se3_inverse_apply(const double* xyz, int64_t n, const double* q_wxyz, const double* t, const uint8_t* kept, double* out,
                                    rvStream stream) {
    if (n == 0) return 0;
    RV_REQUIRE(xyz && q_wxyz && t && out, "rv_se3_inverse_apply: null argument");
    hipLaunchKernelGGL(se3_inverse_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, xyz, n, q_wxyz, t, kept, out);
    RV_CHECK_LAUNCH("se3_inverse_kernel");
    return 0;
}

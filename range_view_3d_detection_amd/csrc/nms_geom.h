// Rotated-BEV-IoU of two boxes [x1, y1, x2, y2, ry] (shared by nms.hip and nms2.hip; both are compiled with
// -ffp-contract=off so that the result matches oracle/c/oracle.c bit for bit).
#pragma once
#include "common.h"

namespace {

struct Pt {
    float x, y;
};

__device__ __forceinline__ float cross2(Pt a, Pt b, Pt p) { return (b.x - a.x) * (p.y - a.y) - (b.y - a.y) * (p.x - a.x); }

__device__ void corners(const float* b, float s, float c, Pt* out) {
    const float cx = (b[0] + b[2]) * 0.5f, cy = (b[1] + b[3]) * 0.5f;
    const float hx = (b[2] - b[0]) * 0.5f, hy = (b[3] - b[1]) * 0.5f;
    const float dx[4] = {hx, -hx, -hx, hx};
    const float dy[4] = {hy, hy, -hy, -hy};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        out[k].x = cx + (dx[k] * c - dy[k] * s);
        out[k].y = cy + (dx[k] * s + dy[k] * c);
    }
}

// rectangle A clipped by the four half-planes of rectangle B (Sutherland-Hodgman), shoelace area
__device__ float rotated_iou(const float* a, float sa, float ca, const float* b, float sb, float cb) {
    const float area_a = (a[2] - a[0]) * (a[3] - a[1]);
    const float area_b = (b[2] - b[0]) * (b[3] - b[1]);
    if (!(area_a > 0.0f) || !(area_b > 0.0f)) return 0.0f;
    Pt pa[4], pb[4], poly[16], tmp[16];
    corners(a, sa, ca, pa);
    corners(b, sb, cb, pb);
    int n = 4;
    for (int k = 0; k < 4; ++k) poly[k] = pa[k];
    for (int e = 0; e < 4 && n > 0; ++e) {
        const Pt e0 = pb[e], e1 = pb[(e + 1) & 3];
        int m = 0;
        for (int k = 0; k < n; ++k) {
            const Pt p = poly[k], q = poly[(k + 1 == n) ? 0 : k + 1];
            const float dp = cross2(e0, e1, p), dq = cross2(e0, e1, q);
            const bool in_p = dp >= 0.0f, in_q = dq >= 0.0f;
            if (in_p) tmp[m++] = p;
            if (in_p != in_q) {
                const float t = dp / (dp - dq);
                Pt r;
                r.x = p.x + t * (q.x - p.x);
                r.y = p.y + t * (q.y - p.y);
                tmp[m++] = r;
            }
        }
        n = m;
        for (int k = 0; k < n; ++k) poly[k] = tmp[k];
    }
    if (n < 3) return 0.0f;
    float twice = 0.0f;
    for (int k = 0; k < n; ++k) {
        const Pt p = poly[k], q = poly[(k + 1 == n) ? 0 : k + 1];
        twice += (p.x - poly[0].x) * (q.y - poly[0].y) - (p.y - poly[0].y) * (q.x - poly[0].x);
    }
    const float inter = 0.5f * fabsf(twice);
    const float uni = area_a + area_b - inter;
    if (!(uni > 0.0f)) return 0.0f;
    return inter / uni;
}

}  // namespace

/*
 * CPU oracle (plain C) for the integer / index / sequential parts of the range-view hot
 * path.  TEST INFRASTRUCTURE -- never linked into the shipped library (see
 * oracle/__init__.py).  Built by oracle/build.py into oracle/_build/liboracle.so with
 * -ffp-contract=off so that fp32 arithmetic is performed operation by operation (the HIP
 * kernels it checks are compiled the same way and must match bit for bit).
 *
 * Reference call sites followed (paths relative to /root/reference/):
 *   z_buffer            converters/av2/utils.py:186-208  ==  src/torchbox3d/math/numpy/conversions.py:106-128
 *   range-view binning  converters/av2/utils.py:108-153 (converter variant)
 *                       src/torchbox3d/math/numpy/conversions.py:9-43 (library variant)
 *   weighted NMS        src/torchbox3d/math/ops/nms.py:126-177 (wrapper contract only; the
 *                       arithmetic is in the absent third-party `weighted_nms_ext` --
 *                       PARITY UNPINNED, semantics declared in oracle/nms.py)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* z-buffer: sequential scan, point i replaces pixel p iff dist_i (f64) < buffer[p] (f32) */
/* ------------------------------------------------------------------------------------ */
void rvo_z_buffer(const int64_t* rows, const int64_t* cols, const double* dist, const double* feat /* (C,N) */,
                  int64_t n_points, int n_channels, int height, int width, double min_distance,
                  float* image /* (C,H*W) zero-initialised by caller */, float* buffer /* (H*W) +inf */,
                  int64_t* winner /* (H*W) -1 */) {
    const int64_t n_pix = (int64_t)height * width;
    for (int64_t i = 0; i < n_points; ++i) {
        const int64_t p = rows[i] * width + cols[i];
        if (p < 0 || p >= n_pix) continue; /* the reference would raise; callers clip first */
        if (dist[i] < min_distance) continue;
        if (dist[i] < (double)buffer[p]) {
            for (int c = 0; c < n_channels; ++c) image[(int64_t)c * n_pix + p] = (float)feat[(int64_t)c * n_points + i];
            buffer[p] = (float)dist[i];
            winner[p] = i;
        }
    }
}

/* ------------------------------------------------------------------------------------ */
/* rotated BEV IoU (fp32, Sutherland-Hodgman clipping of rectangle A by rectangle B)     */
/* box = [x1, y1, x2, y2, ry]: centre ((x1+x2)/2,(y1+y2)/2), extents (x2-x1, y2-y1)       */
/* sin/cos of ry are supplied by the caller as fp32 roundings of the fp64 values.        */
/* ------------------------------------------------------------------------------------ */
typedef struct { float x, y; } pt_t;

static void corners(const float* b, float s, float c, pt_t* out) {
    const float cx = (b[0] + b[2]) * 0.5f, cy = (b[1] + b[3]) * 0.5f;
    const float hx = (b[2] - b[0]) * 0.5f, hy = (b[3] - b[1]) * 0.5f;
    const float dx[4] = {hx, -hx, -hx, hx};
    const float dy[4] = {hy, hy, -hy, -hy};
    for (int k = 0; k < 4; ++k) { /* counter-clockwise */
        out[k].x = cx + (dx[k] * c - dy[k] * s);
        out[k].y = cy + (dx[k] * s + dy[k] * c);
    }
}

static float cross2(pt_t a, pt_t b, pt_t p) { /* (b-a) x (p-a) */
    return (b.x - a.x) * (p.y - a.y) - (b.y - a.y) * (p.x - a.x);
}

float rvo_rotated_iou(const float* a, float sa, float ca, const float* b, float sb, float cb) {
    const float area_a = (a[2] - a[0]) * (a[3] - a[1]);
    const float area_b = (b[2] - b[0]) * (b[3] - b[1]);
    if (!(area_a > 0.0f) || !(area_b > 0.0f)) return 0.0f;
    pt_t pa[4], pb[4], poly[16], tmp[16];
    corners(a, sa, ca, pa);
    corners(b, sb, cb, pb);
    int n = 4;
    for (int k = 0; k < 4; ++k) poly[k] = pa[k];
    for (int e = 0; e < 4 && n > 0; ++e) {
        const pt_t e0 = pb[e], e1 = pb[(e + 1) & 3];
        int m = 0;
        for (int k = 0; k < n; ++k) {
            const pt_t p = poly[k], q = poly[(k + 1 == n) ? 0 : k + 1];
            const float dp = cross2(e0, e1, p), dq = cross2(e0, e1, q);
            const int in_p = dp >= 0.0f, in_q = dq >= 0.0f;
            if (in_p) tmp[m++] = p;
            if (in_p != in_q) {
                const float t = dp / (dp - dq);
                pt_t r;
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
        const pt_t p = poly[k], q = poly[(k + 1 == n) ? 0 : k + 1];
        twice += (p.x - poly[0].x) * (q.y - poly[0].y) - (p.y - poly[0].y) * (q.x - poly[0].x);
    }
    float inter = 0.5f * fabsf(twice);
    const float uni = area_a + area_b - inter;
    if (!(uni > 0.0f)) return 0.0f;
    return inter / uni;
}

/* ------------------------------------------------------------------------------------ */
/* weighted NMS -- declared semantics (oracle/nms.py).  Inputs sorted by score desc.     */
/*   boxes (N,5) f32, data (N,D) f32 whose LAST column is the score == merge weight.      */
/*   Walk i = 0..N-1: if i is not suppressed it becomes output row `o`:                   */
/*     cluster = {i} U { j > i : not suppressed before i was visited, IoU(i,j) > merge }  */
/*     output[o,:] = sum_j w_j * data[j,:] / sum_j w_j (ascending j, fp32), count[o]=|cl| */
/*     every j > i with IoU(i,j) > nms_thresh becomes suppressed.                         */
/*   Returns the number of outputs; output rows >= num_out stay 0 (nms.py:173).           */
/* ------------------------------------------------------------------------------------ */
int64_t rvo_weighted_nms(const float* boxes, const float* data, int64_t n, int d, float nms_thresh, float merge_thresh,
                         float* output, int64_t* keep, int64_t* count) {
    uint8_t* dead = (uint8_t*)calloc((size_t)(n > 0 ? n : 1), 1);
    float* sn = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    float* cs = (float*)malloc(sizeof(float) * (size_t)(n > 0 ? n : 1));
    for (int64_t i = 0; i < n; ++i) {
        sn[i] = (float)sin((double)boxes[i * 5 + 4]);
        cs[i] = (float)cos((double)boxes[i * 5 + 4]);
    }
    int64_t num_out = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (dead[i]) continue;
        float* out = output + num_out * d;
        const float wi = data[i * d + d - 1];
        float wsum = wi;
        for (int c = 0; c < d; ++c) out[c] = wi * data[i * d + c];
        int64_t members = 1;
        for (int64_t j = i + 1; j < n; ++j) {
            if (dead[j]) continue;
            const float iou = rvo_rotated_iou(boxes + i * 5, sn[i], cs[i], boxes + j * 5, sn[j], cs[j]);
            if (iou > merge_thresh) {
                const float wj = data[j * d + d - 1];
                for (int c = 0; c < d; ++c) out[c] += wj * data[j * d + c];
                wsum += wj;
                ++members;
            }
            if (iou > nms_thresh) dead[j] = 1;
        }
        for (int c = 0; c < d; ++c) out[c] = out[c] / wsum;
        keep[num_out] = i;
        count[num_out] = members;
        ++num_out;
    }
    free(dead);
    free(sn);
    free(cs);
    return num_out;
}

/* pairwise IoU matrix (row-major N x M), for tests of the device IoU */
void rvo_pairwise_iou(const float* a, int64_t n, const float* b, int64_t m, float* out) {
    for (int64_t i = 0; i < n; ++i) {
        const float sa = (float)sin((double)a[i * 5 + 4]), ca = (float)cos((double)a[i * 5 + 4]);
        for (int64_t j = 0; j < m; ++j) {
            const float sb = (float)sin((double)b[j * 5 + 4]), cb = (float)cos((double)b[j * 5 + 4]);
            out[i * m + j] = rvo_rotated_iou(a + i * 5, sa, ca, b + j * 5, sb, cb);
        }
    }
}

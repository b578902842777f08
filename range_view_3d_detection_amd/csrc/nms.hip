// nms.hip -- weighted NMS on device; replaces `weighted_nms_ext.wnms_gpu` (math/ops/nms.py:161-170).
//
// The third-party kernel's arithmetic is not in the reference tree (parity unpinned); this
// file implements the semantics declared in oracle/nms.py and matches oracle/c/oracle.c bit for
// bit (compiled with -ffp-contract=off; IEEE division; sin/cos of the yaw are fp32 roundings of
// the fp64 values).  Three stages, all wavefront-level integer/bit work:
//   1. pairwise rotated-BEV-IoU bit masks (upper triangle): one 64-lane wave = 64 boxes x one
//      64-box column block staged in LDS; two u64 words per lane (IoU > nms, IoU > merge);
//   2. the inherently sequential scan over boxes in score order: one workgroup keeps the
//      suppressed-set bit vector in LDS; per kept box it ORs one mask row into it and masks the
//      merge row with the boxes still alive (=> cluster membership);
//   3. cluster merge: one wave per kept box, lanes = data columns, members visited in ascending
//      index order (fixed summation order => reproducible).
#include "common.h"
#include "nms_geom.h"

namespace {

__global__ void sincos_kernel(const float* boxes, int64_t n, float* sc) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double ry = (double)boxes[i * 5 + 4];
        sc[2 * i] = (float)sin(ry);
        sc[2 * i + 1] = (float)cos(ry);
    }
}

// grid (col_block, row_block); only col_block >= row_block does work
// cats (optional): class id per box -- boxes of different classes neither suppress nor merge (all classes of a sweep in
// one launch instead of the reference's per-class loop)
__global__ __launch_bounds__(64) void iou_mask_kernel(const float* boxes, const float* sc, const int32_t* cats, int64_t n, int cb,
                                                      float nms_t, float merge_t, unsigned long long* nms_mask,
                                                      unsigned long long* merge_mask) {
    const int col = blockIdx.x, row = blockIdx.y;
    if (col < row) return;
    __shared__ float cbox[64][7];
    __shared__ int32_t ccat[64];
    const int t = threadIdx.x;
    const int64_t j0 = (int64_t)col * 64;
    if (j0 + t < n) {
        ccat[t] = cats ? cats[j0 + t] : 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) cbox[t][k] = boxes[(j0 + t) * 5 + k];
        cbox[t][5] = sc[2 * (j0 + t)];
        cbox[t][6] = sc[2 * (j0 + t) + 1];
    }
    __syncthreads();
    const int64_t i = (int64_t)row * 64 + t;
    if (i >= n) return;
    float a[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) a[k] = boxes[i * 5 + k];
    const float sa = sc[2 * i], ca = sc[2 * i + 1];
    const int32_t cat_i = cats ? cats[i] : 0;
    // bounding circle of box i: boxes whose circles are apart cannot intersect -- their IoU is 0 in the clipping arithmetic
    // too, so skipping them changes no bit of the masks (the thresholds are positive)
    const float cxi = 0.5f * (a[0] + a[2]), cyi = 0.5f * (a[1] + a[3]);
    const float ri = 0.5f * sqrtf((a[2] - a[0]) * (a[2] - a[0]) + (a[3] - a[1]) * (a[3] - a[1]));
    const bool skip_far = nms_t >= 0.f && merge_t >= 0.f;
    unsigned long long bits_n = 0ull, bits_m = 0ull;
    const int jn = (int)((n - j0) < 64 ? (n - j0) : 64);
    for (int j = 0; j < jn; ++j) {
        if (j0 + j <= i || ccat[j] != cat_i) continue;
        const float dx = 0.5f * (cbox[j][0] + cbox[j][2]) - cxi, dy = 0.5f * (cbox[j][1] + cbox[j][3]) - cyi;
        const float rj = 0.5f * sqrtf((cbox[j][2] - cbox[j][0]) * (cbox[j][2] - cbox[j][0]) + (cbox[j][3] - cbox[j][1]) * (cbox[j][3] - cbox[j][1]));
        if (skip_far && dx * dx + dy * dy > (ri + rj) * (ri + rj) * 1.001f + 1e-4f) continue;
        const float iou = rotated_iou(a, sa, ca, cbox[j], cbox[j][5], cbox[j][6]);
        if (iou > nms_t) bits_n |= 1ull << j;
        if (iou > merge_t) bits_m |= 1ull << j;
    }
    nms_mask[i * cb + col] = bits_n;
    merge_mask[i * cb + col] = bits_m;
}

// One workgroup; remv (suppressed set) lives in LDS.  The scan walks the boxes in blocks of 64 (one mask word): inside a block
// the chain "is box b still alive?" is resolved by ONE wave from the 64 diagonal words held one per lane (64 register-only
// steps, no barrier); then every thread owning a later word w folds the rows of the block's kept boxes into remv[w] (and
// masks their merge rows with the boxes alive at their visit) -- one round of global loads and two barriers per 64 boxes
// instead of per kept box.  Same visiting order and the same sets as the box-by-box loop (oracle/nms.py).
__global__ __launch_bounds__(1024) void scan_kernel(int64_t n, int cb, const unsigned long long* nms_mask,
                                                    unsigned long long* merge_mask, long long* keep, long long* num_out) {
    extern __shared__ unsigned long long remv[];
    __shared__ unsigned long long kept_word;
    for (int w = threadIdx.x; w < cb; w += blockDim.x) remv[w] = 0ull;
    __syncthreads();
    long long kept_total = 0;
    for (int wi = 0; wi < cb; ++wi) {
        if (threadIdx.x < 64) {
            const int b = threadIdx.x;
            const int64_t i = (int64_t)wi * 64 + b;
            const bool in = i < n;
            const unsigned long long diag = in ? nms_mask[i * cb + wi] : 0ull;
            const unsigned long long in_bits = __ballot(in);
            const uint32_t dlo = (uint32_t)diag, dhi = (uint32_t)(diag >> 32);
            unsigned long long rem = remv[wi], kept = 0ull, alive_mine = 0ull;
            for (int q = 0; q < 64; ++q) {  // uniform loop; lane q's diagonal word through readlane
                if (!((in_bits >> q) & 1ull) || ((rem >> q) & 1ull)) continue;
                kept |= 1ull << q;
                if (b == q) alive_mine = ~rem;
                rem |= ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)dhi, q) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)dlo, q);
            }
            if ((kept >> b) & 1ull) {
                keep[kept_total + __popcll(kept & ((1ull << b) - 1ull))] = i;
                merge_mask[i * cb + wi] &= alive_mine;  // cluster = merge candidates not suppressed before i was visited
            }
            if (b == 0) {
                remv[wi] = rem;
                kept_word = kept;
            }
        }
        __syncthreads();
        const unsigned long long kept = kept_word;
        kept_total += __popcll(kept);
        for (int w = wi + 1 + threadIdx.x; w < cb; w += blockDim.x) {
            unsigned long long r = remv[w], bits = kept;
            while (bits) {
                const int q = __ffsll((long long)bits) - 1;
                bits &= bits - 1;
                const int64_t i = (int64_t)wi * 64 + q;
                merge_mask[i * cb + w] &= ~r;
                r |= nms_mask[i * cb + w];
            }
            remv[w] = r;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *num_out = kept_total;
}

// one wave per kept box; lane = data column
__global__ __launch_bounds__(64) void merge_kernel(const float* data, int d, int cb, const unsigned long long* merge_mask,
                                                   const long long* keep, const long long* num_out, float* output,
                                                   long long* count) {
    const long long o = blockIdx.x;
    if (o >= *num_out) return;
    const long long i = keep[o];
    const int c = threadIdx.x;
    const bool active = c < d;
    const float wi = data[i * d + d - 1];
    float acc = active ? wi * data[i * d + c] : 0.f;
    float wsum = wi;
    long long members = 1;
    for (int w = (int)(i >> 6); w < cb; ++w) {
        unsigned long long bits = merge_mask[i * cb + w];
        while (bits) {
            const int b = __ffsll((long long)bits) - 1;
            bits &= bits - 1;
            const long long j = (long long)w * 64 + b;
            const float wj = data[j * d + d - 1];
            if (active) acc += wj * data[j * d + c];
            wsum += wj;
            ++members;
        }
    }
    if (active) output[o * d + c] = acc / wsum;
    if (c == 0) count[o] = members;
}

__global__ void pairwise_iou_kernel(const float* a, int64_t n, const float* b, int64_t m, float* out) {
    const int64_t total = n * m;
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < total; k += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = k / m, j = k - i * m;
        const float sa = (float)sin((double)a[i * 5 + 4]), ca = (float)cos((double)a[i * 5 + 4]);
        const float sb = (float)sin((double)b[j * 5 + 4]), cb = (float)cos((double)b[j * 5 + 4]);
        out[k] = rotated_iou(a + i * 5, sa, ca, b + j * 5, sb, cb);
    }
}

}  // namespace

static int64_t align256(int64_t v) { return (v + 255) & ~255ll; }

extern "C" int64_t rv_wnms_workspace_bytes(int64_t n) {
    const int64_t cb = (n + 63) / 64;
    return 2 * align256(n * cb * 8) + align256(n * 2 * 4) + 256;
}

extern "C" int rv_wnms(const float* boxes, const float* data, int64_t n, int32_t d, float nms_thresh, float merge_thresh,
                       float* output, int64_t* keep, int64_t* count, void* workspace, int64_t* host_num_out,
                       rvStream stream) {
    return rv_wnms_classes(boxes, data, nullptr, n, d, nms_thresh, merge_thresh, output, keep, count, workspace, host_num_out, stream);
}

extern "C" int rv_wnms_classes(const float* boxes, const float* data, const int32_t* cats, int64_t n, int32_t d, float nms_thresh,
                               float merge_thresh, float* output, int64_t* keep, int64_t* count, void* workspace,
                               int64_t* host_num_out, rvStream stream) {
    RV_REQUIRE(host_num_out, "rv_wnms: null host_num_out");
    *host_num_out = 0;
    if (n == 0) return 0;
    RV_REQUIRE(boxes && data && output && keep && count && workspace, "rv_wnms: null argument");
    RV_REQUIRE(d >= 1 && d <= 64, "rv_wnms: data width %d unsupported (1..64)", d);
    const int64_t cb64 = (n + 63) / 64;
    RV_REQUIRE(cb64 * 8 <= 160 * 1024 - 256, "rv_wnms: too many boxes (%lld)", (long long)n);
    const int cb = (int)cb64;
    hipStream_t st = (hipStream_t)stream;
    uint8_t* ws = (uint8_t*)workspace;
    unsigned long long* nms_mask = (unsigned long long*)ws;
    unsigned long long* merge_mask = (unsigned long long*)(ws + align256(n * cb64 * 8));
    float* sc = (float*)(ws + 2 * align256(n * cb64 * 8));
    long long* num_out = (long long*)(ws + 2 * align256(n * cb64 * 8) + align256(n * 2 * 4));
    hipLaunchKernelGGL(sincos_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, boxes, n, sc);
    hipLaunchKernelGGL(iou_mask_kernel, dim3(cb, cb), dim3(64), 0, st, boxes, sc, cats, n, cb, nms_thresh, merge_thresh, nms_mask,
                       merge_mask);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)scan_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);  // + the static word
        attr = true;
    }
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(cb < 1024 ? ((cb + 63) / 64) * 64 : 1024), (size_t)cb * 8, st, n, cb,
                       nms_mask, merge_mask, (long long*)keep, num_out);
    hipLaunchKernelGGL(merge_kernel, dim3((unsigned)n), dim3(64), 0, st, data, d, cb, merge_mask, (const long long*)keep,
                       num_out, output, (long long*)count);
    RV_CHECK_LAUNCH("wnms kernels");
    long long host = 0;
    hipError_t e = hipMemcpyAsync(&host, num_out, sizeof(host), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) RV_FAIL("rv_wnms: %s", hipGetErrorString(e));
    *host_num_out = host;
    return 0;
}

extern "C" int rv_rotated_iou(const float* a, int64_t n, const float* b, int64_t m, float* out, rvStream stream) {
    if (n * m == 0) return 0;
    RV_REQUIRE(a && b && out, "rv_rotated_iou: null argument");
    const int64_t blocks = (n * m + 255) / 256;
    hipLaunchKernelGGL(pairwise_iou_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(256), 0, (hipStream_t)stream,
                       a, n, b, m, out);
    RV_CHECK_LAUNCH("pairwise_iou_kernel");
    return 0;
}

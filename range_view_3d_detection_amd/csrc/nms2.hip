// nms2.hip -- the whole post-decode path of a batch of sweeps on device, without a host round trip in the middle:
// confidence filter + compaction, segmented (class, score) sort, class-segmented weighted NMS, per-class top-k, final
// compaction.  Replaces, for the common case, the reference's per-sweep / per-class Python loop
// (math/ops/nms.py:64-123, 181-266: boolean masks, `unique`, `topk`, one `wnms_gpu` call and one device->host copy per
// class) and round 1's one-launch-per-sweep variant (ATen sorts + a serial scan over ALL boxes + a host sync per sweep).
//
// Stages (one launch each for the WHOLE batch; every kernel reads its sweep's candidate count from device memory):
//   1. compact     candidates with score >= min_confidence -> slot list (wave ballot + one atomic per wave);
//   2. rank        position of every candidate in (class ascending, score descending, candidate index ascending) order by
//                  COUNTING (rank = number of candidates with a smaller key: n^2 comparisons through LDS tiles --
//                  n is a few thousand, the IoU stage is n^2 anyway) -- a sort without a sort library and without a sync;
//   3. gather      boxes in that order as NMS rectangles [x1,y1,x2,y2,ry], merge rows [x,y,z,l,w,h,sin,cos,score];
//                  class segment offsets;
//   4. iou masks   rotated-BEV-IoU bit masks, only for 64x64 blocks whose class ranges meet (classes are contiguous now);
//   5. scan        the sequential suppression scan -- ONE WORKGROUP PER CLASS (classes do not interact), each over its
//                  own segment: the serial chain is the boxes of one class, not of the sweep;
//   6. keep        prefix-sum compaction of the kept flags;
//   7. merge       one wave per kept box (cluster = merge candidates still alive when the box was visited);
//   8. post        per class: rank by merged score (counting again), drop ranks >= num_post_nms, write the rows in the
//                  reference's output order (classes ascending, merged score descending) and the final count.
// Results are the rows of the per-class loop, bit for bit (same IoU arithmetic, same merge summation order: within a class
// the score order is the same as in a sweep-wide score order).  The host reads back ONE small array (counts of all
// sweeps) at the end.
// Round 3: the bit masks are CLASS-RELATIVE -- a row of class c holds only the words of c's own segment, so the two masks
// take sum_c n_c * ceil(n_c / 64) words instead of n * ceil(n / 64): the per-candidate arrays are sized for `cap` candidates
// (up to 262 144: the decoder emits 212 992 per sweep), the masks for a word BUDGET; a sweep whose classes need more words
// reports the number (counts[.][3]) and the caller re-runs the mask stages (`resume`) over a buffer of that size -- the
// ordering stages are not repeated.  The reference's pre-NMS cut (`scores.topk(num_pre_nms)` per class, nms.py:83-84) is
// the end of the class segment: boxes past position seg[c] + num_pre_nms of their class take no part.
#include "common.h"
#include "nms_geom.h"

namespace {

constexpr int kMaxClasses = 64;

struct SweepPtrs {  // workspace layout of one sweep (all device pointers)
    int32_t* n;           // [2]: candidates >= min_conf (may exceed cap), kept boxes
    int32_t* cand;        // [cap] slot list (indices into the K candidates)
    int32_t* order;       // [cap] sorted position -> slot-list position
    int32_t* cats;        // [cap] class of the sorted boxes
    int32_t* seg;         // [kMaxClasses + 1] class segment offsets
    int32_t* send;        // [kMaxClasses] end of the part of the segment that takes part (pre-NMS top-k cut)
    int64_t* mbase;       // [kMaxClasses + 1] first mask word of each class's rows; [kMaxClasses] = words needed
    float* rect;          // [cap][5]
    float* data;          // [cap][9]
    float* sc;            // [cap][2] fp32(sin(ry)), fp32(cos(ry)) for the IoU
    unsigned long long* nms_mask;    // class c: [send[c] - seg[c]][words of the segment], at mbase[c]
    unsigned long long* merge_mask;
    uint8_t* kept;        // [cap]
    int32_t* keep;        // [cap] kept boxes, ascending sorted position
    float* merged;        // [cap][9]
};

struct Args {
    const float* scores;    // [B][K]
    const int64_t* cats;    // [B][K]
    const float* cuboids;   // [B][K][7]
    uint8_t* ws;
    int64_t ws_stride;      // bytes per sweep
    int64_t K;
    unsigned long long* mask_ws;  // [B][2][mask_words]
    int64_t mask_words;           // word budget per sweep and mask
    int32_t B, cap, cb, n_classes, num_post, num_pre, out_cap;
    float min_conf, nms_t, merge_t;
    float* out_boxes;       // [B][cap][7]
    float* out_scores;      // [B][cap]
    int32_t* out_cats;      // [B][cap]
    int64_t* out_counts;    // [B][4]: final rows (-1: mask budget exceeded, -2: more candidates than cap), candidates >= min_conf, kept, mask words needed
};

__host__ __device__ inline int64_t al256(int64_t v) { return (v + 255) & ~255ll; }

__host__ __device__ inline SweepPtrs carve(uint8_t* base, int cap, int cb) {
    SweepPtrs p;
    int64_t o = 0;
    p.n = (int32_t*)(base + o); o += 256;
    p.cand = (int32_t*)(base + o); o += al256((int64_t)cap * 4);
    p.order = (int32_t*)(base + o); o += al256((int64_t)cap * 4);
    p.cats = (int32_t*)(base + o); o += al256((int64_t)cap * 4);
    p.seg = (int32_t*)(base + o); o += al256((kMaxClasses + 1) * 4);
    p.send = (int32_t*)(base + o); o += al256((kMaxClasses + 1) * 4);
    p.mbase = (int64_t*)(base + o); o += al256((kMaxClasses + 1) * 8);
    p.rect = (float*)(base + o); o += al256((int64_t)cap * 5 * 4);
    p.data = (float*)(base + o); o += al256((int64_t)cap * 9 * 4);
    p.sc = (float*)(base + o); o += al256((int64_t)cap * 2 * 4);
    p.nms_mask = p.merge_mask = nullptr;  // set by sweep(): the masks live in their own buffer
    p.kept = (uint8_t*)(base + o); o += al256(cap);
    p.keep = (int32_t*)(base + o); o += al256((int64_t)cap * 4);
    p.merged = (float*)(base + o); o += al256((int64_t)cap * 9 * 4);
    return p;
}
inline int64_t sweep_bytes(int cap, int cb) {
    return 256 + 4 * al256((int64_t)cap * 4) + 2 * al256((kMaxClasses + 1) * 4) + al256((kMaxClasses + 1) * 8) + al256((int64_t)cap * 20) +
           2 * al256((int64_t)cap * 36) + al256((int64_t)cap * 8) + al256(cap);
}

__device__ __forceinline__ SweepPtrs sweep(const Args& a, int b) {
    SweepPtrs p = carve(a.ws + (int64_t)b * a.ws_stride, a.cap, a.cb);
    p.nms_mask = a.mask_ws + (int64_t)b * 2 * a.mask_words;
    p.merge_mask = p.nms_mask + a.mask_words;
    return p;
}
// mask word (box i of class c, global word index w) -- valid for seg[c] <= i < send[c], w within the words of [seg[c], send[c])
__device__ __forceinline__ int64_t mword(const SweepPtrs& p, int c, int i, int w) {
    const int s0 = p.seg[c], w0 = s0 >> 6, nw = ((p.send[c] - 1) >> 6) - w0 + 1;
    return p.mbase[c] + (int64_t)(i - s0) * nw + (w - w0);
}
__device__ __forceinline__ bool masks_fit(const Args& a, const SweepPtrs& p) { return p.n[0] <= a.cap && p.mbase[kMaxClasses] <= a.mask_words; }
__device__ __forceinline__ int n_of(const SweepPtrs& p, int cap) { const int n = p.n[0]; return n < cap ? n : cap; }

__global__ void k_zero(const Args a) {
    const SweepPtrs p = sweep(a, blockIdx.x);
    if (threadIdx.x < 2) p.n[threadIdx.x] = 0;
}

// 1. compaction: blockIdx.y = sweep
__global__ __launch_bounds__(256) void k_compact(const Args a) {
    const int b = blockIdx.y;
    const SweepPtrs p = sweep(a, b);
    const float* s = a.scores + (int64_t)b * a.K;
    const int lane = threadIdx.x & 63;
    for (int64_t base = (int64_t)blockIdx.x * 256; base < a.K; base += (int64_t)gridDim.x * 256) {
        const int64_t i = base + threadIdx.x;
        const bool pick = i < a.K && s[i] >= a.min_conf;
        const unsigned long long m = __ballot(pick);
        int pos0 = 0;
        if (lane == 0 && m) pos0 = atomicAdd(p.n, __popcll(m));
        pos0 = __shfl(pos0, 0, 64);
        if (pick) {
            const int pos = pos0 + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < a.cap) p.cand[pos] = (int32_t)i;
        }
    }
}

// 2. rank by counting.  key(i) < key(j)  <=>  (cat_i, -score_i, slot_i) lexicographically smaller
__global__ __launch_bounds__(256) void k_rank(const Args a) {
    const int b = blockIdx.y;
    const SweepPtrs p = sweep(a, b);
    const int n = n_of(p, a.cap);
    if ((int)(blockIdx.x * 256) >= n) return;
    const float* s = a.scores + (int64_t)b * a.K;
    const int64_t* c = a.cats + (int64_t)b * a.K;
    __shared__ float ts[256];
    __shared__ int32_t tc[256], ti[256];
    const int i = blockIdx.x * 256 + threadIdx.x;
    float si = 0.f;
    int ci = 0, ii = 0;
    if (i < n) {
        ii = p.cand[i];
        si = s[ii];
        ci = (int)c[ii];
    }
    int rank = 0;
    for (int j0 = 0; j0 < n; j0 += 256) {
        const int j = j0 + threadIdx.x;
        if (j < n) {
            const int jj = p.cand[j];
            ts[threadIdx.x] = s[jj];
            tc[threadIdx.x] = (int)c[jj];
            ti[threadIdx.x] = jj;
        }
        __syncthreads();
        const int m = n - j0 < 256 ? n - j0 : 256;
        for (int k = 0; k < m; ++k) {
            const int cj = tc[k];
            const float sj = ts[k];
            rank += (cj < ci) || (cj == ci && (sj > si || (sj == si && ti[k] < ii)));
        }
        __syncthreads();
    }
    if (i < n) p.order[rank] = i;
}

// 3. gather in sorted order + class segments
__global__ __launch_bounds__(256) void k_gather(const Args a) {
    const int b = blockIdx.y;
    const SweepPtrs p = sweep(a, b);
    const int n = n_of(p, a.cap);
    const float* s = a.scores + (int64_t)b * a.K;
    const int64_t* c = a.cats + (int64_t)b * a.K;
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < n) {
        const int64_t src = p.cand[p.order[r]];
        const float* q = a.cuboids + ((int64_t)b * a.K + src) * 7;
        const float x = q[0], y = q[1], l = q[3], w = q[4], yaw = q[6];
        const float hl = l / 2, hw = w / 2;
        float* rc = p.rect + (int64_t)r * 5;
        rc[0] = x - hl; rc[1] = y - hw; rc[2] = x + hl; rc[3] = y + hw; rc[4] = yaw;
        float* d = p.data + (int64_t)r * 9;
        d[0] = x; d[1] = y; d[2] = q[2]; d[3] = l; d[4] = w; d[5] = q[5];
        d[6] = sinf(yaw); d[7] = cosf(yaw); d[8] = s[src];
        p.sc[2 * r] = (float)sin((double)yaw);
        p.sc[2 * r + 1] = (float)cos((double)yaw);
        p.cats[r] = (int32_t)c[src];
        p.kept[r] = 0;
    }
}

// class segment offsets: seg[c] = first sorted position with class >= c (binary search; one thread per class)
__global__ void k_seg(const Args a) {
    const SweepPtrs p = sweep(a, blockIdx.x);
    const int n = n_of(p, a.cap);
    const int c = threadIdx.x;
    if (c > kMaxClasses) return;
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (p.cats[mid] < c) lo = mid + 1; else hi = mid;
    }
    p.seg[c] = lo;
    __syncthreads();
    if (c == 0) {  // effective segment ends (pre-NMS top-k cut) and the class-relative mask layout
        int64_t acc = 0;
        for (int q = 0; q < kMaxClasses; ++q) {
            const int s0 = p.seg[q], s1 = p.seg[q + 1];
            const int e = s1 - s0 > a.num_pre ? s0 + a.num_pre : s1;
            p.send[q] = e;
            p.mbase[q] = acc;
            if (e > s0) acc += (int64_t)(e - s0) * (((e - 1) >> 6) - (s0 >> 6) + 1);
        }
        p.mbase[kMaxClasses] = acc;
    }
}

// resumed call: clear the kept flags (k_gather did it in the first call; an aborted mask pass wrote none, a completed one did)
__global__ __launch_bounds__(256) void k_unkeep(const Args a) {
    const SweepPtrs p = sweep(a, blockIdx.y);
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < n_of(p, a.cap)) p.kept[r] = 0;
}

__device__ void k_iou_block(const Args& a, const SweepPtrs& p, int n, int row, int col) {
    const int64_t j0 = (int64_t)col * 64, i0 = (int64_t)row * 64;
    __shared__ float cbox[64][7];
    __shared__ int32_t ccat[64];
    const int t = threadIdx.x;
    if (j0 + t < n) {
        ccat[t] = p.cats[j0 + t];
#pragma unroll
        for (int k = 0; k < 5; ++k) cbox[t][k] = p.rect[(j0 + t) * 5 + k];
        cbox[t][5] = p.sc[2 * (j0 + t)];
        cbox[t][6] = p.sc[2 * (j0 + t) + 1];
    }
    __syncthreads();
    const int64_t i = i0 + t;
    if (i >= n) { __syncthreads(); return; }
    const int32_t cat_i = p.cats[i];
    const int send_i = p.send[cat_i];
    if (i >= send_i || col > ((send_i - 1) >> 6)) { __syncthreads(); return; }  // cut by the pre-NMS top-k / column block past this row's class
    float bx[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) bx[k] = p.rect[i * 5 + k];
    const float sa = p.sc[2 * i], ca = p.sc[2 * i + 1];
    // bounding circle of box i: boxes whose circles are apart cannot intersect -- their IoU is 0 in the clipping arithmetic
    // too, so skipping them changes no bit of the masks (thresholds are positive) and skips ~all pairs of a spread-out scene
    const float cxi = 0.5f * (bx[0] + bx[2]), cyi = 0.5f * (bx[1] + bx[3]);
    const float ri = 0.5f * sqrtf((bx[2] - bx[0]) * (bx[2] - bx[0]) + (bx[3] - bx[1]) * (bx[3] - bx[1]));
    const bool skip_far = a.nms_t >= 0.f && a.merge_t >= 0.f;
    unsigned long long bits_n = 0ull, bits_m = 0ull;
    const int jn = (int)((n - j0) < 64 ? (n - j0) : 64);
    for (int j = 0; j < jn; ++j) {
        if (j0 + j <= i || ccat[j] != cat_i || j0 + j >= send_i) continue;
        const float dx = 0.5f * (cbox[j][0] + cbox[j][2]) - cxi, dy = 0.5f * (cbox[j][1] + cbox[j][3]) - cyi;
        const float rj = 0.5f * sqrtf((cbox[j][2] - cbox[j][0]) * (cbox[j][2] - cbox[j][0]) + (cbox[j][3] - cbox[j][1]) * (cbox[j][3] - cbox[j][1]));
        if (skip_far && dx * dx + dy * dy > (ri + rj) * (ri + rj) * 1.001f + 1e-4f) continue;
        const float iou = rotated_iou(bx, sa, ca, cbox[j], cbox[j][5], cbox[j][6]);
        if (iou > a.nms_t) bits_n |= 1ull << j;
        if (iou > a.merge_t) bits_m |= 1ull << j;
    }
    const int64_t mw = mword(p, cat_i, (int)i, col);
    p.nms_mask[mw] = bits_n;
    p.merge_mask[mw] = bits_m;
    __syncthreads();  // (the LDS tile is reused by the next column block)
}


// 4. IoU bit masks: grid (kIouCols column offsets, row block, sweep).  A row block only meets the column blocks from its own
// up to the last block of the class of its last box (classes are contiguous); workgroup (dc, row) takes columns
// row + dc, row + dc + kIouCols, ...
constexpr int kIouCols = 16;
constexpr int kIouRows = 512;    // row blocks per launch wave (32 768 boxes); larger sweeps loop
constexpr int kMergeGrid = 16384;
__global__ __launch_bounds__(64) void k_iou(const Args a) {
    const SweepPtrs p = sweep(a, blockIdx.z);
    const int n = n_of(p, a.cap);
    if (!masks_fit(a, p)) return;
    for (int row = blockIdx.y; (int64_t)row * 64 < n; row += gridDim.y) {  // (the grid covers at most kIouRows row blocks at a time)
        const int64_t i0 = (int64_t)row * 64;
        const int row_last = (int)(i0 + 63 < n ? i0 + 63 : n - 1);
        const int c_last = p.cats[row_last];
        // the classes that END inside this row block have their last word here (col == row); the class of the last row may go
        // on (or may have been cut by the pre-NMS top-k before this block: then only the diagonal block is left to do)
        const int w_last = (p.send[c_last] - 1) >> 6;
        const int col_last = w_last > row ? w_last : row;
        for (int col = row + blockIdx.x; col <= col_last; col += kIouCols) k_iou_block(a, p, n, row, col);
    }
}

// 5. suppression scan: grid (class, sweep).  Words of the masks outside [first, last] word of the segment are never read
// (k_iou may not have written them).
// The scan walks the segment in blocks of 64 boxes (one mask word).  Inside a block the chain "is box b still alive?" is
// resolved by ONE wave from the 64 diagonal words held one per lane (64 register-only steps); then every thread owning a later
// word w folds the rows of the block's kept boxes into remv[w] (and masks their merge rows with the boxes alive at their
// visit) -- one round of global loads per 64 boxes instead of one per kept box.
__global__ __launch_bounds__(256) void k_scan(const Args a) {
    const SweepPtrs p = sweep(a, blockIdx.y);
    const int cls = blockIdx.x;
    const int s0 = p.seg[cls], s1 = p.send[cls];
    if (s0 >= s1 || !masks_fit(a, p)) return;
    extern __shared__ unsigned long long remv[];  // words w0 .. w1
    __shared__ unsigned long long kept_word;
    const int w0 = s0 >> 6, w1 = (s1 - 1) >> 6;
    for (int w = threadIdx.x; w <= w1 - w0; w += blockDim.x) remv[w] = 0ull;
    __syncthreads();
    for (int wi = w0; wi <= w1; ++wi) {
        if (threadIdx.x < 64) {
            const int b = threadIdx.x;
            const int i = wi * 64 + b;
            const bool in_seg = i >= s0 && i < s1;
            const unsigned long long diag = in_seg ? p.nms_mask[mword(p, cls, i, wi)] : 0ull;
            const unsigned long long seg_bits = __ballot(in_seg);
            const uint32_t dlo = (uint32_t)diag, dhi = (uint32_t)(diag >> 32);
            unsigned long long rem = remv[wi - w0], kept = 0ull, alive_mine = 0ull;
            for (int q = 0; q < 64; ++q) {  // uniform loop; lane q's diagonal word through readlane
                if (!((seg_bits >> q) & 1ull) || ((rem >> q) & 1ull)) continue;
                kept |= 1ull << q;
                if (b == q) alive_mine = ~rem;
                rem |= ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)dhi, q) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)dlo, q);
            }
            if ((kept >> b) & 1ull) {
                p.kept[i] = 1;
                p.merge_mask[mword(p, cls, i, wi)] &= alive_mine;  // cluster = merge candidates not suppressed before i was visited
            }
            if (b == 0) {
                remv[wi - w0] = rem;
                kept_word = kept;
            }
        }
        __syncthreads();
        const unsigned long long kept = kept_word;
        for (int w = wi + 1 + threadIdx.x; w <= w1; w += blockDim.x) {
            // a word at the segment's edge may hold bits of the neighbouring class: k_iou never sets those (class check)
            unsigned long long r = remv[w - w0], bits = kept;
            while (bits) {
                const int q = __ffsll((long long)bits) - 1;
                bits &= bits - 1;
                const int64_t mw = mword(p, cls, wi * 64 + q, w);
                p.merge_mask[mw] &= ~r;
                r |= p.nms_mask[mw];
            }
            remv[w - w0] = r;
        }
        __syncthreads();
    }
}

// 6. compaction of the kept flags (ascending position): one workgroup per sweep
__global__ __launch_bounds__(1024) void k_keep(const Args a) {
    const SweepPtrs p = sweep(a, blockIdx.x);
    const int n = masks_fit(a, p) ? n_of(p, a.cap) : 0;
    __shared__ int part[1024];
    const int per = (n + 1023) / 1024;
    const int i0 = threadIdx.x * per, i1 = i0 + per < n ? i0 + per : n;
    int cnt = 0;
    for (int i = i0; i < i1; ++i) cnt += p.kept[i];
    part[threadIdx.x] = cnt;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {  // inclusive scan
        const int v = (int)threadIdx.x >= off ? part[threadIdx.x - off] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int pos = part[threadIdx.x] - cnt;
    for (int i = i0; i < i1; ++i)
        if (p.kept[i]) p.keep[pos++] = i;
    if (threadIdx.x == 1023) p.n[1] = part[1023];
}

// 7. cluster merge: one wave per kept box; lane = data column; members in ascending position (fixed summation order)
__global__ __launch_bounds__(64) void k_merge(const Args a) {
    const SweepPtrs p = sweep(a, blockIdx.y);
    for (int o = blockIdx.x; o < p.n[1]; o += gridDim.x) {
    const int i = p.keep[o];
    const int cls = p.cats[i];
    const int w1 = (p.send[cls] - 1) >> 6;  // last mask word of the box's class segment
    const int c = threadIdx.x;
    const bool active = c < 9;
    const float wi = p.data[(int64_t)i * 9 + 8];
    float acc = active ? wi * p.data[(int64_t)i * 9 + c] : 0.f;
    float wsum = wi;
    for (int w = i >> 6; w <= w1; ++w) {
        unsigned long long bits = p.merge_mask[mword(p, cls, i, w)];
        while (bits) {
            const int bb = __ffsll((long long)bits) - 1;
            bits &= bits - 1;
            const int64_t j = (int64_t)w * 64 + bb;
            const float wj = p.data[j * 9 + 8];
            if (active) acc += wj * p.data[j * 9 + c];
            wsum += wj;
        }
    }
    if (active) p.merged[(int64_t)o * 9 + c] = acc / wsum;
    }
}

// 8. per-class top-k by merged score and the final order: grid (blocks over kept boxes, sweep)
__global__ __launch_bounds__(256) void k_post(const Args a) {
    const int b = blockIdx.y;
    const SweepPtrs p = sweep(a, b);
    const int k = p.n[1];
    __shared__ int cls_cnt[kMaxClasses], cls_base[kMaxClasses + 1];
    if (threadIdx.x < kMaxClasses) cls_cnt[threadIdx.x] = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < k; j += 256) atomicAdd(&cls_cnt[p.cats[p.keep[j]]], 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        int acc = 0;
        for (int c = 0; c < kMaxClasses; ++c) {
            cls_base[c] = acc;
            acc += cls_cnt[c] < a.num_post ? cls_cnt[c] : a.num_post;
        }
        cls_base[kMaxClasses] = acc;
        if (blockIdx.x == 0) {
            // -2: more candidates than the capacity; -1: the masks of this sweep need more words than the budget (resume with
            // a buffer of out_counts[.][3] words); in both cases nothing usable was written
            a.out_counts[4 * b] = p.n[0] > a.cap ? -2 : (p.mbase[kMaxClasses] > a.mask_words ? -1 : acc);
            a.out_counts[4 * b + 1] = p.n[0];
            a.out_counts[4 * b + 2] = k;
            a.out_counts[4 * b + 3] = p.mbase[kMaxClasses];
        }
    }
    __syncthreads();
    if ((int)(blockIdx.x * 256) >= k) return;
    const int o = blockIdx.x * 256 + threadIdx.x;
    const bool live = o < k;
    const int co = live ? p.cats[p.keep[o]] : -1;
    const float so = live ? p.merged[(int64_t)o * 9 + 8] : 0.f;
    __shared__ float ts[256];
    __shared__ int32_t tc[256];
    int rank = 0;  // within the class: merged score descending, kept order ascending on ties
    for (int j0 = 0; j0 < k; j0 += 256) {
        const int j = j0 + threadIdx.x;
        if (j < k) {
            tc[threadIdx.x] = p.cats[p.keep[j]];
            ts[threadIdx.x] = p.merged[(int64_t)j * 9 + 8];
        }
        __syncthreads();
        const int m = k - j0 < 256 ? k - j0 : 256;
        for (int q = 0; q < m; ++q) rank += tc[q] == co && (ts[q] > so || (ts[q] == so && j0 + q < o));
        __syncthreads();
    }
    if (!live || rank >= a.num_post) return;
    const int pos = cls_base[co] + rank;
    if (pos >= a.out_cap) return;  // (cannot happen when out_cap >= min(cap, n_classes * num_post_nms): checked on the host)
    const float* m = p.merged + (int64_t)o * 9;
    float* ob = a.out_boxes + ((int64_t)b * a.out_cap + pos) * 7;
#pragma unroll
    for (int q = 0; q < 6; ++q) ob[q] = m[q];
    ob[6] = atan2f(m[6], m[7]);
    a.out_scores[(int64_t)b * a.out_cap + pos] = so;
    a.out_cats[(int64_t)b * a.out_cap + pos] = co;
}

}  // namespace

extern "C" int64_t rv_nms_sweeps_workspace_bytes(int32_t B, int32_t cap) {
    if (B <= 0 || cap <= 0) return 0;
    return (int64_t)B * sweep_bytes(cap, (cap + 63) / 64);
}

extern "C" int rv_nms_sweeps(const float* scores, const int64_t* cats, const float* cuboids, int32_t B, int64_t K, int32_t n_classes,
                             float min_confidence, float nms_thresh, float merge_thresh, int32_t num_pre_nms, int32_t num_post_nms,
                             int32_t cap, int32_t out_cap, float* out_boxes, float* out_scores, int32_t* out_cats, int64_t* out_counts,
                             void* workspace, void* mask_workspace, int64_t mask_words, int32_t resume, rvStream stream) {
    RV_REQUIRE(scores && cats && cuboids && out_boxes && out_scores && out_cats && out_counts && workspace && mask_workspace,
               "rv_nms_sweeps: null argument");
    RV_REQUIRE(B > 0 && K > 0 && cap >= 64 && cap % 64 == 0 && cap <= 262144, "rv_nms_sweeps: bad sizes (cap: multiple of 64, <= 262144)");
    RV_REQUIRE(n_classes >= 1 && n_classes <= kMaxClasses, "rv_nms_sweeps: 1..%d classes", kMaxClasses);
    RV_REQUIRE(num_post_nms >= 1 && num_pre_nms >= 1, "rv_nms_sweeps: num_pre_nms / num_post_nms must be positive");
    RV_REQUIRE(mask_words >= 1, "rv_nms_sweeps: mask_words must be positive");
    const int64_t rows_max = (int64_t)n_classes * num_post_nms < cap ? (int64_t)n_classes * num_post_nms : cap;
    RV_REQUIRE(out_cap >= rows_max, "rv_nms_sweeps: out_cap %d below min(cap, n_classes * num_post_nms) = %lld", out_cap, (long long)rows_max);
    Args a;
    a.scores = scores; a.cats = cats; a.cuboids = cuboids;
    a.ws = (uint8_t*)workspace;
    a.mask_ws = (unsigned long long*)mask_workspace;
    a.mask_words = mask_words;
    a.cb = (cap + 63) / 64;
    a.ws_stride = sweep_bytes(cap, a.cb);
    a.K = K; a.B = B; a.cap = cap; a.n_classes = n_classes; a.num_post = num_post_nms; a.num_pre = num_pre_nms; a.out_cap = out_cap;
    a.min_conf = min_confidence; a.nms_t = nms_thresh; a.merge_t = merge_thresh;
    a.out_boxes = out_boxes; a.out_scores = out_scores; a.out_cats = out_cats; a.out_counts = out_counts;
    hipStream_t st = (hipStream_t)stream;
    const int cblocks = (int)((K + 255) / 256 < 256 ? (K + 255) / 256 : 256);
    if (!resume) {  // ordering stages (their results stay in `workspace` for a resumed call)
        hipLaunchKernelGGL(k_zero, dim3(B), dim3(64), 0, st, a);
        hipLaunchKernelGGL(k_compact, dim3(cblocks, B), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_rank, dim3((cap + 255) / 256, B), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_gather, dim3((cap + 255) / 256, B), dim3(256), 0, st, a);
        hipLaunchKernelGGL(k_seg, dim3(B), dim3(128), 0, st, a);
    } else {
        hipLaunchKernelGGL(k_unkeep, dim3((cap + 255) / 256, B), dim3(256), 0, st, a);
    }
    hipLaunchKernelGGL(k_iou, dim3(kIouCols, a.cb < kIouRows ? a.cb : kIouRows, B), dim3(64), 0, st, a);
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute((const void*)k_scan, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
        attr = true;
    }
    const int seg_max = cap < num_pre_nms ? cap : num_pre_nms;  // longest class segment that takes part
    hipLaunchKernelGGL(k_scan, dim3(n_classes, B), dim3(256), (size_t)(seg_max / 64 + 3) * 8, st, a);
    hipLaunchKernelGGL(k_keep, dim3(B), dim3(1024), 0, st, a);
    hipLaunchKernelGGL(k_merge, dim3(cap < kMergeGrid ? cap : kMergeGrid, B), dim3(64), 0, st, a);
    hipLaunchKernelGGL(k_post, dim3((cap + 255) / 256, B), dim3(256), 0, st, a);
    RV_CHECK_LAUNCH("rv_nms_sweeps kernels");
    return 0;
}

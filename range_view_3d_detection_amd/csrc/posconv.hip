// posconv.hip -- the two positional layers of the MetaKernel stem (nn/stems/__init__.py:41-49, 80: two
// Conv2dNormActivation 1x1 blocks on the (B, 3, 9, H*W) relative-coordinate tensor) as purpose-built streaming GEMMs.
//
// Both layers live on the 9x neighbour grid: P = 9*N*H*W "pixels" (4.7 M at 4 x 64 x 2048), C = 256 channels, so every
// tensor of the pair is 2.4 GB and the layers are HBM-bound, not MFMA-bound (0.62 TFLOP for the C x C GEMM).  The generic
// 1x1 tap-conv re-fetches its 128 KB weight matrix for every 256-pixel tile, cannot fold the element-wise first layer into
// its operand staging (LDS-DMA bypasses the registers), and leaves 2/3 of a tile's life to fill and drain (four K tiles).
// Here a workgroup is PERSISTENT (one per CU, eight waves), keeps the C x C weights in REGISTERS for the whole launch
// (wave w owns output channels 32 w .. 32 w + 31: 16 fragments = 64 VGPRs) and streams 128-pixel steps through a
// double-buffered LDS image of the K operand:
//
//   rv_pos_forward   h1 = relu(s1 * (W1 rel) + t1) is GENERATED in the K-operand staging from 8 bytes of `rel` per pixel
//                    (3 -> C is an element-wise map), written to HBM once (the weight gradient of the second layer reads
//                    it) and multiplied by W2 from LDS: y2 = W2 h1 (raw bf16 + fp32 batch statistics of the accumulators).
//                    Traffic: 2 x 2.4 GB written, nothing of that size read (the unfused pair wrote h1, read it, wrote y2).
//
// MFMA orientation: D = W X^T (weights as the A operand, pixels as B), so that a lane ends up with 8 CONSECUTIVE output
// channels of one pixel (rows m = 4 lg + r of tile j map to channel 8 lg + 4 j + r of the wave's 32): one 16-byte store
// per pixel tile, the four lane groups of a pixel covering 64 contiguous bytes.
//
// LDS image of a step: [128 pixels][32 k-octets of 16 B], octet slot XOR-ed with (pixel & 15).  Writers (thread = pixel
// lane x octet, 8 consecutive lanes = 8 consecutive octets of a pixel: one aligned 128-byte run) and readers (the
// ds_read_b128 lane groups {0-3, 12-15, 20-27} ... of MI355X_MICROARCH.md: pixels p and octets c, c^1 -> 16 distinct slots
// of a 256-byte window) are both conflict-free.
#include <stdlib.h>

#include <type_traits>

#include "common.h"
#include "tapconv.h"

namespace {

// Geometry per channel count C (256: rv-av2's stem, 128: rv-waymo's).  An LDS image is 64 KB either way, so a step is 128 pixels at
// C = 256 and 256 pixels at C = 128; a wave owns 32 output channels, and at C = 128 the eight waves are 4 channel slices x 2 pixel halves.
template <int C>
struct Pos {
    static_assert(C == 256 || C == 128, "positional pair: 256 or 128 channels");
    static constexpr int kC = C;                 // channels of both layers
    static constexpr int kTM = 128 * 256 / C;    // pixels per step
    static constexpr int kRow = C * 2;           // bytes of one pixel of an image
    static constexpr int kBuf = kTM * kRow;      // 64 KB per LDS image
    static constexpr int kOct = C / 8;           // 16-byte octets per pixel
    static constexpr int kPass = 512 / kOct;     // pixels the workgroup generates (or a DMA round moves) per pass
    static constexpr int kSel = 64 / kOct;       // pixels of one wave per pass
    static constexpr int kKS = C / 32;           // MFMA K-steps
    static constexpr int kWC = C / 32;           // waves along the channels ...
    static constexpr int kWP = 8 / kWC;          // ... and along the pixels of a step
    static constexpr int kGroups = kTM / 32 / kWP;  // 32-pixel groups a wave multiplies per step (4)
};
constexpr int kTMmin = 128;  // the row count of the partial buffers is sized for 128-pixel steps (rv_pos_forward_rows): surplus rows are zeros

struct PosFwdArgs {
    const bf16_t* rel;  // [P][ld_rel] bf16, channels 0..cin-1 used
    const bf16_t* w1;   // packed gather image of the first layer [C][ld_w1]
    const bf16_t* w2;   // packed gather image of the second layer [C][C], K contiguous
    const float *scale1, *shift1;
    bf16_t* h1;         // [P][C]
    bf16_t* y2;         // [P][C]
    float* partial;     // [gridDim.x][2][C] fp32 (sum, sum of squares) or NULL
    int64_t P;
    int ld_rel, ld_w1, cin;
    // inference form (EVAL): y2 is not stored -- geo = relu(scale2 * y2 + shift2) * feat[neighbour] goes to `y2`'s place, h1 nowhere
    const float *scale2, *shift2;
    const bf16_t* feat;  // [N*H*W][ld_feat]
    int ld_feat, H, W;
};

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef rv_elem_t bf16x2_t __attribute__((ext_vector_type(2)));
// two floats -> one dword of bf16 (round to nearest even): ONE v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack2(const f32x2 v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t)); }
__device__ __forceinline__ f32x2 max0(const f32x2 v) { return f32x2{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)}; }

template <int C, bool EVAL>
__global__ __launch_bounds__(512, 1) void pos_fwd_kernel(const PosFwdArgs a) {
    using G = Pos<C>;
    constexpr int kC = G::kC, kTM = G::kTM, kBuf = G::kBuf, kRow = G::kRow, kKS = G::kKS;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wn % G::kWC, wp = wn / G::kWC;  // wave = 32-channel slice of the output x pixel half of a step (C = 256: all 128 pixels)
    const int l15_ = lane & 15, lg_ = lane >> 4;
    const int l15 = l15_, lg = lg_;

    // ---- second-layer weights of this wave: A operand, row m = l15 of tile j is channel 32 wn + 8 (m >> 2) + 4 j + (m & 3),
    // so that a lane of the result (rows 4 lg + r of both tiles) holds the 8 consecutive channels 32 wn + 8 lg + 0..7
    bf16x8 fw[2][kKS];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ch = wc * 32 + (l15 >> 2) * 8 + j * 4 + (l15 & 3);
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) fw[j][ks] = *(const bf16x8*)(a.w2 + (int64_t)ch * kC + ks * 32 + lg * 8);
    }
    // ---- first layer: this thread generates octet `oct` of pixels pxl, pxl + kPass, ... of a step
    const int oct_ = tid % G::kOct, pxl_ = tid / G::kOct;
    const int oct = oct_;
    // Per-lane address parts, formed ONCE and kept opaque (the step loop adds wave-uniform and immediate parts only: recomputing
    // them per position cost ~200 of the ~830 VALU instructions a wave issued per step, and this kernel is VALU-issue bound):
    //   wb       LDS byte offset of this thread's octet in row pxl of an image (the XOR uses pxl & 15 = pl & 15: kPass % 16 == 0)
    //   rb[x]    LDS byte offset of the fragment read of row l15, K-step ks with (ks & 3) == x: slot ((ks*4 + lg) ^ l15) --
    //            the XOR reaches the low four slot bits only, so (ks >> 2), the 16-row tile and the group are immediates
    //   hoff / yoff   element offsets of this thread's h1 octet / y2 octet inside a step
    uint32_t wb = pxl_ * kRow + ((oct_ ^ (pxl_ & 15)) * 16), rb[4];
    int hoff = pxl_ * kC + oct_ * 8, yoff = (wp * G::kGroups * 32 + l15_) * kC + wc * 32 + lg_ * 8;
#pragma unroll
    for (int x = 0; x < 4; ++x) rb[x] = (wp * G::kGroups * 32 + l15_) * kRow + (((((x << 2) | lg_) ^ l15_) & 15) * 16);
    asm volatile("" : "+v"(wb), "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(hoff), "+v"(yoff));
    // (channel PAIRS -> v_pk_fma_f32; the arithmetic order is rv_smallk_forward's: relu(s (w . rel) + t))
    f32x2 gw[4][3], gs[4], gh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int c = oct * 8 + 2 * j + q;
#pragma unroll
            for (int e = 0; e < 3; ++e) gw[j][e][q] = e < a.cin ? bf2f(a.w1[(int64_t)c * a.ld_w1 + e]) : 0.f;
            gs[j][q] = a.scale1[c];
            gh[j][q] = a.shift1[c];
        }
    }
    // `rel` reaches the lanes through SCALAR loads (s_load_dwordx2: a wave generates kSel pixels per pass -- C = 256: lanes 0-31 one,
    // lanes 32-63 the other; C = 128: four groups of 16 lanes).  The step loop then contains NO vector-memory load: on gfx9 loads and stores share vmcnt and
    // return out of order with respect to each other, so a wait for any load inside the loop drains every outstanding h1 / y2
    // store first (measured: waves 74 % of their cycles in s_waitcnt, 1.6 ms instead of 0.9).
    const uint64_t* relq = (const uint64_t*)a.rel;
    const int ldq = a.ld_rel / 4;  // row pitch in 8-byte units
    auto generate8 = [&](int64_t step, int buf) __attribute__((always_inline)) {  // this thread's 8 pixels of `step` -> LDS image `buf`, h1
        // (inline asm: hipcc turns these into VECTOR loads because the h1 / y2 stores might alias `rel`; indices are clamped
        //  instead of guarded -- pixels past the end are masked in the epilogue and never stored)
        constexpr int kSel = G::kSel, kIt = 16 / kSel;  // sixteen scalar loads in flight: 8 passes x 2 pixels, or 4 passes x 4 pixels twice
        bf16_t* const h1_step = a.h1 + step * (int64_t)(kTM * kC);  // wave-uniform base: the lane part stays a 32-bit offset
        const int64_t left64 = a.P - step * kTM;                    // pixels of this step that exist
        const int left = left64 < kTM ? (int)left64 : kTM;
        const uint32_t wbuf = wb + buf * kBuf;
#pragma unroll
        for (int it0 = 0; it0 < 8; it0 += kIt) {
        uint64_t qs[16];
#pragma unroll
        for (int it = 0; it < kIt; ++it)
#pragma unroll
            for (int e = 0; e < kSel; ++e) {
                const int64_t pu = step * kTM + (it0 + it) * G::kPass + kSel * wn + e;  // wave-uniform
                const int64_t p0 = pu < a.P ? pu : a.P - 1;
                asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=s"(qs[it * kSel + e]) : "s"(relq + p0 * ldq));
            }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+s"(qs[0]), "+s"(qs[1]), "+s"(qs[2]), "+s"(qs[3]), "+s"(qs[4]), "+s"(qs[5]), "+s"(qs[6]), "+s"(qs[7]), "+s"(qs[8]),
                       "+s"(qs[9]), "+s"(qs[10]), "+s"(qs[11]), "+s"(qs[12]), "+s"(qs[13]), "+s"(qs[14]), "+s"(qs[15]));
#pragma unroll
        for (int it = 0; it < kIt; ++it) {
            const int pl0 = (it0 + it) * G::kPass;  // (+ pxl_)
            uint64_t q = qs[it * kSel];
#pragma unroll
            for (int e = 1; e < kSel; ++e) q = lane >= e * G::kOct ? qs[it * kSel + e] : q;
            const uint32_t lo = (uint32_t)q, hi = (uint32_t)(q >> 32);
            const float x = bf_lo(lo), y = bf_hi(lo), z = bf_lo(hi);
            const f32x2 xx = {x, x}, yy = {y, y}, zz = {z, z};
            u32x4 hv;
#pragma unroll
            for (int j = 0; j < 4; ++j) hv[j] = pack2(max0((gw[j][0] * xx + gw[j][1] * yy + gw[j][2] * zz) * gs[j] + gh[j]));
            *(u32x4*)(smem + wbuf + pl0 * kRow) = hv;
            if (!EVAL && pl0 + pxl_ < left) *(u32x4*)(h1_step + pl0 * kC + hoff) = hv;
            __builtin_amdgcn_sched_barrier(0);  // one pixel at a time: eight interleaved would need 32 more registers
        }
        }
    };

    f32x2 ssum[4], ssq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ssum[q] = ssq[q] = f32x2{0.f, 0.f};

    // ---- inference: the modulation of MetaKernel.forward (nn/stems/__init__.py:80-83) in the epilogue.  Row pk = 9 p + k of the
    // 9x grid is neighbour k of pixel p; a lane owns 8 channels of 8 rows per step and fetches the neighbour's 8 feature
    // channels (16 bytes, zero outside the image) with ordinary vector loads issued a WHOLE STEP before their use, right after
    // the epilogue that consumed the previous contents of the same registers: the shared vmcnt (see the note on `rel` above) then
    // only asks for stores that are a step old when a wave waits for its features.  Measured at 4 x 64 x 2048 x 256: 1.34 ms (one
    // generate phase ahead: 1.40) against 1.36 + 1.04 for the two kernels; with neither loads nor stores the step loop itself takes
    // 1.21 ms (generate + multiply + barrier per 128 pixels), which is what bounds this form -- the write bound would be 0.7.
    f32x2 sc2[4], sh2[4];
    u32x4 fv[G::kGroups][2];
    if (EVAL) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sc2[q] = f32x2{a.scale2[wc * 32 + lg * 8 + 2 * q], a.scale2[wc * 32 + lg * 8 + 2 * q + 1]};
            sh2[q] = f32x2{a.shift2[wc * 32 + lg * 8 + 2 * q], a.shift2[wc * 32 + lg * 8 + 2 * q + 1]};
        }
    }
    auto load_feat = [&](int64_t step, int hq) __attribute__((always_inline)) {  // fv[hq][:] of `step`
        const int pk0 = (int)(step * kTM);  // (the launcher checks 9 N H W < 2^31 and W >= 32)
        const int p0 = pk0 / 9, k0 = pk0 - 9 * p0;
        const int r0 = p0 / a.W, w0 = p0 - r0 * a.W, h0 = r0 % a.H;
        int l15 = l15_, lg = lg_;
        asm volatile("" : "+v"(l15), "+v"(lg));
        {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pl = ((wp * G::kGroups + hq) * 2 + i) * 16 + l15;
                const int d = pl + k0, q9 = (d * 7282) >> 16, k = d - 9 * q9;  // d < 265: exact d / 9
                int w = w0 + q9, h = h0;
                if (w >= a.W) w -= a.W, h += 1;
                if (h >= a.H) h = 0;
                const int dy = ((k * 11) >> 5) - 1, dx = k - 3 * (dy + 1) - 1;  // k / 3 - 1, k % 3 - 1
                const int hn = h + dy, wn = w + dx;
                const bool in = pk0 + pl < a.P && hn >= 0 && hn < a.H && wn >= 0 && wn < a.W;
                const int64_t nbr = (int64_t)(p0 + q9) + dy * a.W + dx;
                fv[hq][i] = u32x4{0u, 0u, 0u, 0u};
                if (in) fv[hq][i] = *(const u32x4*)(a.feat + nbr * a.ld_feat + wc * 32 + lg * 8);
            }
        }
    };

    // multiply the image of step `s` (LDS image `cur`) by this wave's weights, store y2, accumulate the statistics
    auto multiply = [&](int64_t s, int cur, int64_t nxt, bool has_next) __attribute__((always_inline)) {
        const int64_t left64 = a.P - s * kTM;
        const int left = left64 < kTM ? (int)left64 : kTM;
        const bool full = left == kTM;  // every pixel of the step exists
        const uint32_t rbc[4] = {rb[0] + cur * kBuf, rb[1] + cur * kBuf, rb[2] + cur * kBuf, rb[3] + cur * kBuf};
        bf16_t* const y2_step = a.y2 + s * (int64_t)(kTM * kC);
#pragma unroll
        for (int hq = 0; hq < G::kGroups; ++hq) {  // 32 pixels at a time: 16 accumulator + 16 fragment registers
            f32x4 acc[2][2];
            bf16x8 fa[4][2];
            auto read_fa = [&](int ks, bf16x8 (&f)[2]) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 2; ++i) f[i] = *(const bf16x8*)(smem + rbc[ks & 3] + ((hq * 2 + i) * 16 * kRow + (ks >> 2) * 256));
            };
            read_fa(0, fa[0]);
            read_fa(1, fa[1]);
            read_fa(2, fa[2]);
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) {
                if (ks + 3 < kKS) read_fa(ks + 3, fa[(ks + 3) & 3]);  // three K-steps ahead (4 MFMAs = 64 cycles per K-step against ~200 of LDS latency), and no further (sched_barrier below):
#pragma unroll                                                       // hoisting all the fragment reads costs 128 registers -> spills,
                for (int i = 0; i < 2; ++i)                          // and a scratch reload is a VMEM load: vmcnt(0) in the loop
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x4 c0 = ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j];
                        acc[i][j] = RV_MFMA_16x16x32(fw[j][ks], fa[ks & 3][i], c0, 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            // lane (pixel l15 of tile i, lane group lg) holds channels 32 wc + 8 lg + 4 j + r
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pl0 = (hq * 2 + i) * 16;  // (+ wp's groups + l15: the row inside the step)
                const bool ok = full || (wp * G::kGroups * 32 + pl0 + l15_) < left;
                u32x4 out;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) {
                        f32x2 x = {acc[i][j][2 * r2], acc[i][j][2 * r2 + 1]};
                        if (EVAL) {
                            // y2 rounded to the storage type first (what the separate modulation pass read), then its arithmetic
                            const uint32_t y16 = pack2(x), f16 = fv[hq][i][j * 2 + r2];
                            const f32x2 y = {bf_lo(y16), bf_hi(y16)}, f = {bf_lo(f16), bf_hi(f16)};
                            out[j * 2 + r2] = pack2(max0(y * sc2[j * 2 + r2] + sh2[j * 2 + r2]) * f);
                            continue;
                        }
                        if (!full) x = ok ? x : f32x2{0.f, 0.f};
                        ssum[j * 2 + r2] += x;
                        ssq[j * 2 + r2] += x * x;
                        out[j * 2 + r2] = pack2(x);
                    }
                if (ok) *(u32x4*)(y2_step + pl0 * kC + yoff) = out;
            }
            if (EVAL && has_next) load_feat(nxt, hq);
        }
    };

    const int64_t steps = (a.P + kTM - 1) / kTM;
    const bool gen_first = wn < 4;
    if ((int64_t)blockIdx.x < steps) {
        if (EVAL) {
#pragma unroll
            for (int hq = 0; hq < G::kGroups; ++hq) load_feat(blockIdx.x, hq);
        }
        generate8(blockIdx.x, 0);
    }
    __syncthreads();
    // The two waves of a SIMD (w and w + 4) walk a step in OPPOSITE order -- generate-then-multiply against
    // multiply-then-generate (the image being generated and the one being multiplied are different buffers) -- so that one
    // issues VALU work while the other keeps the matrix pipe busy.
    int cur = 0;  // k & 1 of the step being multiplied
    for (int64_t s = blockIdx.x; s < steps; s += gridDim.x) {
        const int64_t nxt = s + gridDim.x;
        const bool has_next = nxt < steps;
        if (gen_first) {
            if (has_next) generate8(nxt, cur ^ 1);
            multiply(s, cur, nxt, has_next);
        } else {
            multiply(s, cur, nxt, has_next);
            if (has_next) generate8(nxt, cur ^ 1);
        }
        // image cur^1 is complete and everyone is done reading image cur: LDS traffic only -- NOT __syncthreads(), whose
        // vmcnt(0) would wait for this step's 128 KB of global stores
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
    }
    if (!EVAL && a.partial) {
        // per-channel totals of this workgroup: sum over the 16 pixel lanes (l15), then (C = 128) over the two waves of a channel slice
        // through LDS (the images are dead: the loop ends on a barrier), then one row per workgroup
        float* const red = (float*)smem;  // [2][C]
        float sv[8], qv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            sv[q] = ssum[q >> 1][q & 1], qv[q] = ssq[q >> 1][q & 1];
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                sv[q] += __shfl_xor(sv[q], d, 64);
                qv[q] += __shfl_xor(qv[q], d, 64);
            }
        }
        if (G::kWP > 1) {
            if (wp == 1 && l15 == 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    red[wc * 32 + lg * 8 + q] = sv[q];
                    red[kC + wc * 32 + lg * 8 + q] = qv[q];
                }
            }
            __syncthreads();
        }
        if (wp == 0 && l15 == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int ch = wc * 32 + lg * 8 + q;
                const float s2 = G::kWP > 1 ? red[ch] : 0.f, q2 = G::kWP > 1 ? red[kC + ch] : 0.f;
                a.partial[((int64_t)blockIdx.x * 2) * kC + ch] = sv[q] + s2;
                a.partial[((int64_t)blockIdx.x * 2 + 1) * kC + ch] = qv[q] + q2;
            }
        }
    }
}

// -----------------------------------------------------------------------------------------------------------------
// Backward: the second layer's backward-data GEMM fused with the first layer's small-K BatchNorm backward.
//   dh1[p][ci] = sum_co dy2[p][co] W2[co][ci]            (never written: 2.4 GB less to store AND to read back)
//   g = dh1 * [s1 y1 + t1 > 0],  y1 = W1 rel (recomputed),  xhat = (y1 - mean1) invstd1
//   planes per channel: S0 = sum g, S1 = sum g xhat, R[d] = sum g rel[d]     (what bn_bwd_smallk_reduce_kernel forms)
// Same persistent structure as pos_fwd_kernel (weights W2^T of the wave's 32 channels in registers, 128-pixel steps
// through a double-buffered LDS image), with three differences: the image is filled by LDS-DMA straight from dy2 (the
// swizzle applied to the per-lane SOURCE address); the loop has no store at all, so ordinary vector loads and their
// vmcnt waits are harmless; and the MFMA orientation is D = X W^T (pixels x channels): a lane owns TWO channels for the
// whole launch, its sums stay in 10 registers, and the eight waves own disjoint channels -- no cross-wave reduction.
// -----------------------------------------------------------------------------------------------------------------
struct PosBwdArgs {
    const bf16_t* dy2;  // [P][C]
    const bf16_t* w2s;  // packed SCATTER image of the second layer [ci][co], K = co contiguous
    const bf16_t* rel;  // [P][ld_rel]
    const bf16_t* w1;   // packed gather image of the first layer [C][ld_w1]
    const float *scale1, *shift1, *mean1, *invstd1;
    float* partial;     // [gridDim.x][planes][C]
    int64_t P;
    int ld_rel, ld_w1, cin, planes;
};

template <int C>
__global__ __launch_bounds__(512, 1) void pos_bwd_kernel(const PosBwdArgs a) {
    using G = Pos<C>;
    constexpr int kC = G::kC, kTM = G::kTM, kBuf = G::kBuf, kRow = G::kRow, kKS = G::kKS;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wn % G::kWC, wp = wn / G::kWC;
    const int l15_ = lane & 15, lg_ = lane >> 4;
    const int l15 = l15_, lg = lg_;
    uint8_t* const relbuf = smem + 2 * kBuf;  // [2][kTM pixels][8 bytes]

    // B operand: row n = l15 of tile j is input channel ci = 32 wc + 16 j + l15
    bf16x8 fw[2][kKS];
    f32x2 w1c[3], sc, sh, mu, is;  // the lane's two channels (j = 0, 1) side by side: v_pk_* arithmetic in the epilogue
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ci = wc * 32 + j * 16 + l15;
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) fw[j][ks] = *(const bf16x8*)(a.w2s + (int64_t)ci * kC + ks * 32 + lg * 8);
#pragma unroll
        for (int e = 0; e < 3; ++e) w1c[e][j] = e < a.cin ? bf2f(a.w1[(int64_t)ci * a.ld_w1 + e]) : 0.f;
        sc[j] = a.scale1[ci];
        sh[j] = a.shift1[ci];
        mu[j] = a.mean1[ci];
        is[j] = a.invstd1[ci];
    }
    f32x2 s0 = {0.f, 0.f}, s1 = {0.f, 0.f}, rx = {0.f, 0.f}, ry = {0.f, 0.f}, rz = {0.f, 0.f};

    const int64_t steps = (a.P + kTM - 1) / kTM;
    // image fill: DMA instruction q (0..63) of a step moves 1 KB = kSel pixels (lane = pixel x kOct slots); wave w issues
    // q = w, w + 8, ...  The LDS side is lane-linear, so slot s of pixel p receives source octet s ^ (p & 15).
    __device__ __attribute__((aligned(256))) static uint32_t zero_page[64];
    auto fill = [&](int64_t step, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = wn + 8 * i;
            const int pl = G::kSel * q + lane / G::kOct, slot = lane % G::kOct;
            const int64_t px = step * kTM + pl;
            const bf16_t* src = px < a.P ? a.dy2 + px * kC + ((slot ^ (pl & 15)) * 8) : (const bf16_t*)zero_page + (lane & 7) * 8;
            __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(smem + buf * kBuf + q * 1024), 16, 0, 0);
        }
    };
    auto load_rel = [&](int64_t step) __attribute__((always_inline)) {
        u32x2 rv = {0u, 0u};
        const int64_t px = step * kTM + tid;
        if (tid < kTM && px < a.P) rv = *(const u32x2*)(a.rel + px * a.ld_rel);
        return rv;
    };
    if ((int64_t)blockIdx.x < steps) {
        fill(blockIdx.x, 0);
        const u32x2 r0 = load_rel(blockIdx.x);
        if (tid < kTM) *(u32x2*)(relbuf + tid * 8) = r0;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int64_t s = blockIdx.x; s < steps; s += gridDim.x) {
        const int64_t nxt = s + gridDim.x;
        const bool has_next = nxt < steps;
        u32x2 rel_next = {0u, 0u};
        if (has_next) {
            fill(nxt, cur ^ 1);
            rel_next = load_rel(nxt);
        }
        const uint8_t* img = smem + cur * kBuf;
        const uint8_t* rel_s = relbuf + cur * (kTM * 8);
        const int64_t left = a.P - s * kTM;
        int l15 = l15_, lg = lg_;
        asm volatile("" : "+v"(l15), "+v"(lg));  // (no hoisting of the per-position LDS offsets out of the step loop: registers)
#pragma unroll 1  // (rolled: unrolled x4 the allocator spilled the weight fragments)
        for (int qq = 0; qq < G::kGroups; ++qq) {  // 32 pixels at a time
            const int quarter = wp * G::kGroups + qq;
            f32x4 acc[2][2];
            bf16x8 fa[4][2];
            auto read_fa = [&](int ks, bf16x8 (&f)[2]) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    f[i] = *(const bf16x8*)(img + ((quarter * 2 + i) * 16 + l15) * kRow + (((ks * 4 + lg) ^ l15) * 16));
            };
            read_fa(0, fa[0]);
            read_fa(1, fa[1]);
            read_fa(2, fa[2]);
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) {
                if (ks + 3 < kKS) read_fa(ks + 3, fa[(ks + 3) & 3]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x4 c0 = ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j];
                        acc[i][j] = RV_MFMA_16x16x32(fa[ks & 3][i], fw[j][ks], c0, 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            // lane: channel ci(j) = 32 wc + 16 j + l15, pixels (quarter*2 + i)*16 + 4 lg + r
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int pl = (quarter * 2 + i) * 16 + lg * 4 + r;
                    const u32x2 rv = *(const u32x2*)(rel_s + pl * 8);
                    const float x = bf_lo(rv[0]), y = bf_hi(rv[0]), z = bf_lo(rv[1]);
                    const bool ok = pl < left;
                    const f32x2 xx = {x, x}, yy = {y, y}, zz = {z, z};
                    const f32x2 y1 = w1c[0] * xx + w1c[1] * yy + w1c[2] * zz;
                    const f32x2 act = y1 * sc + sh;
                    f32x2 g = {acc[i][0][r], acc[i][1][r]};
                    g[0] = (ok && act[0] > 0.f) ? g[0] : 0.f;
                    g[1] = (ok && act[1] > 0.f) ? g[1] : 0.f;
                    s0 += g;
                    s1 += g * ((y1 - mu) * is);
                    rx += g * xx;
                    ry += g * yy;
                    rz += g * zz;
                    __builtin_amdgcn_sched_barrier(0);  // one pixel at a time
                }
        }
        if (has_next && tid < kTM) *(u32x2*)(relbuf + (cur ^ 1) * (kTM * 8) + tid * 8) = rel_next;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's share of the next image has landed (no stores in this loop)
        __syncthreads();
        cur ^= 1;
    }
    // lanes lg = 0..3 of a channel hold disjoint pixels: sum them (C = 128: and the two waves of a channel slice, through LDS -- the
    // images are dead, the loop ends on a barrier), then one row of `planes` x C per workgroup
    float* const red = (float*)smem;  // [5][C]
    float v[2][5];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        v[j][0] = s0[j], v[j][1] = s1[j], v[j][2] = rx[j], v[j][3] = ry[j], v[j][4] = rz[j];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            v[j][q] += __shfl_xor(v[j][q], 16, 64);
            v[j][q] += __shfl_xor(v[j][q], 32, 64);
        }
    }
    if (G::kWP > 1) {
        if (wp == 1 && lg == 0) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 5; ++q) red[q * kC + wc * 32 + j * 16 + l15] = v[j][q];
        }
        __syncthreads();
    }
    if (wp == 0 && lg == 0) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int ci = wc * 32 + j * 16 + l15;
            float* row = a.partial + (int64_t)blockIdx.x * a.planes * kC + ci;
#pragma unroll
            for (int q = 0; q < 5; ++q) row[(int64_t)q * kC] = v[j][q] + (G::kWP > 1 ? red[q * kC + ci] : 0.f);
            for (int q = 5; q < a.planes; ++q) row[(int64_t)q * kC] = 0.f;
        }
    }
}

// -----------------------------------------------------------------------------------------------------------------
// Pointwise (1x1, stride 1) C -> C layers on plain bf16 tensors as the same persistent streaming GEMM (round 6).
// conv2d with a 1x1 kernel (nn/modules/conv.py:47-54 with kernel_size 1; the projection convs of nn/blocks/__init__.py:58-66 and the
// stem's second fusion conv) is y[p] = W x[p] over P = N*H*W pixels: 268 MB in, 268 MB out and 69 GFLOP at 4 x 64 x 2048 x 256 -- bound by
// HBM, not by the matrix cores.  The fourth-generation tap-conv runs these launches at 3.4-3.6 TB/s of algorithmic traffic (a 256-pixel tile is
// four K tiles long: two thirds of its life are fill and drain, and every tile fetches the 128 KB weight matrix again).  Here: the image fill of
// pos_bwd_kernel (LDS-DMA, swizzle on the per-lane source address) in front of the multiply / store / statistics of pos_fwd_kernel, weights in
// registers for the whole launch.  Loads and stores share vmcnt in issue order: the NEXT step's eight DMA instructions are issued BEFORE this
// step's eight stores, so the wait for them is `vmcnt(8)` -- it leaves the stores in flight (a step that has a successor is never ragged:
// all eight store instructions are issued).
// -----------------------------------------------------------------------------------------------------------------
struct PwArgs {
    const bf16_t* x;  // [P][ld_x]
    const bf16_t* w;  // [C][C] packed 1x1 image, K contiguous (gather image: forward; scatter image: backward-data)
    bf16_t* y;        // [P][ld_y]
    float* partial;   // [gridDim.x][2][C] fp32 (sum, sum of squares of the fp32 accumulators) or NULL
    int64_t P;
    int ld_x, ld_y;
    int slices;  // C_out = slices x C: workgroup (group, t) multiplies the steps of its group by the t-th C x C block of the weight rows
    int pair;    // 128 -> 128 layers on DENSE rows (ld = 128): two adjacent pixels are one 256-channel "pixel" and the weights are diag(W, W),
                 // formed in the register prologue from the 128 x 128 image (the zero blocks cost MFMAs this bandwidth-bound kernel has to spare);
                 // P, ld_x, ld_y are then those of the paired view; statistics rows: two per workgroup (one per pixel parity), 128 channels wide
};

template <int C>
__global__ __launch_bounds__(512, 1) void pointwise_kernel(const PwArgs a) {
    using G = Pos<C>;
    constexpr int kC = G::kC, kTM = G::kTM, kBuf = G::kBuf, kRow = G::kRow, kKS = G::kKS;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wn % G::kWC, wp = wn / G::kWC;
    const int l15_ = lane & 15, lg_ = lane >> 4;
    // Workgroup -> (step group, output slice), XCD-aware: the `slices` workgroups that multiply the SAME pixels by different blocks of the
    // weight rows sit on one XCD (blockIdx & 7) and fetch the image through one L2; slots of an XCD that do not fill a whole group idle
    // (slices = 9 on 32 slots: 27 busy).  slices = 1: a permutation of the workgroups.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = (gridDim.x >> 3) / a.slices;  // (gridDim.x % 8 == 0: the launcher)
    if (slot >= per_xcd * a.slices) return;  // (before any barrier)
    const int tsl = slot % a.slices;
    const int64_t group = xcd * per_xcd + slot / a.slices, n_groups = 8 * per_xcd;
    // weights of this wave: A operand, row m = l15 of tile j is channel 32 wc + 8 (m >> 2) + 4 j + (m & 3) (see pos_fwd_kernel)
    bf16x8 fw[2][kKS];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ch = tsl * kC + wc * 32 + (l15_ >> 2) * 8 + j * 4 + (l15_ & 3);
#pragma unroll
        for (int ks = 0; ks < kKS; ++ks) {
            if (!a.pair) {
                fw[j][ks] = *(const bf16x8*)(a.w + (int64_t)ch * kC + ks * 32 + lg_ * 8);
            } else {  // diag(W, W): output half ch >> 7 against input half ks >> 2 (K-steps 0..3 = the first pixel's 128 channels)
                const bf16x8 wv = *(const bf16x8*)(a.w + (int64_t)(ch & 127) * 128 + (ks & 3) * 32 + lg_ * 8);
                bf16x8 z;
#pragma unroll
                for (int e = 0; e < 8; ++e) z[e] = (rv_elem_t)0.f;
                fw[j][ks] = ((ch >> 7) == (ks >> 2)) ? wv : z;
            }
        }
    }
    uint32_t rb[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) rb[x] = (wp * G::kGroups * 32 + l15_) * kRow + (((((x << 2) | lg_) ^ l15_) & 15) * 16);
    int yoff = (wp * G::kGroups * 32 + l15_) * a.ld_y + tsl * kC + wc * 32 + lg_ * 8;
    asm volatile("" : "+v"(rb[0]), "+v"(rb[1]), "+v"(rb[2]), "+v"(rb[3]), "+v"(yoff));
    // image fill: DMA instruction q (0..63) of a step moves 1 KB = kSel pixels (lane = pixel x kOct slots); wave w issues q = w, w + 8, ...
    // The LDS side is lane-linear, so slot s of pixel p receives source octet s ^ (p & 15).  Per-lane source offsets once (32-bit elements).
    __device__ __attribute__((aligned(256))) static uint32_t pw_zero_page[64];
    int foff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = wn + 8 * i;
        const int pl = G::kSel * q + lane / G::kOct, slot = lane % G::kOct;
        foff[i] = pl * a.ld_x + ((slot ^ (pl & 15)) * 8);
    }
    auto fill = [&](int64_t step, int buf) __attribute__((always_inline)) {
        const bf16_t* const x_step = a.x + step * (int64_t)kTM * a.ld_x;
        const int64_t left = a.P - step * kTM;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = wn + 8 * i;
            const int pl = G::kSel * q + lane / G::kOct;
            const bf16_t* src = pl < left ? x_step + foff[i] : (const bf16_t*)pw_zero_page + (lane & 7) * 8;
            __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(smem + buf * kBuf + q * 1024), 16, 0, 0);
        }
    };
    f32x2 ssum[4], ssq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ssum[q] = ssq[q] = f32x2{0.f, 0.f};
    auto multiply = [&](int64_t s, int cur) __attribute__((always_inline)) {
        const int64_t left64 = a.P - s * kTM;
        const int left = left64 < kTM ? (int)left64 : kTM;
        const bool full = left == kTM;
        const uint32_t rbc[4] = {rb[0] + cur * kBuf, rb[1] + cur * kBuf, rb[2] + cur * kBuf, rb[3] + cur * kBuf};
        bf16_t* const y_step = a.y + s * (int64_t)kTM * a.ld_y;
#pragma unroll
        for (int hq = 0; hq < G::kGroups; ++hq) {
            f32x4 acc[2][2];
            bf16x8 fa[4][2];
            auto read_fa = [&](int ks, bf16x8 (&f)[2]) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 2; ++i) f[i] = *(const bf16x8*)(smem + rbc[ks & 3] + ((hq * 2 + i) * 16 * kRow + (ks >> 2) * 256));
            };
            read_fa(0, fa[0]);
            read_fa(1, fa[1]);
            read_fa(2, fa[2]);
#pragma unroll
            for (int ks = 0; ks < kKS; ++ks) {
                if (ks + 3 < kKS) read_fa(ks + 3, fa[(ks + 3) & 3]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x4 c0 = ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j];
                        acc[i][j] = RV_MFMA_16x16x32(fw[j][ks], fa[ks & 3][i], c0, 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pl0 = (hq * 2 + i) * 16;
                const bool ok = full || (wp * G::kGroups * 32 + pl0 + l15_) < left;
                u32x4 out;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) {
                        f32x2 x = {acc[i][j][2 * r2], acc[i][j][2 * r2 + 1]};
                        if (!full) x = ok ? x : f32x2{0.f, 0.f};
                        ssum[j * 2 + r2] += x;
                        ssq[j * 2 + r2] += x * x;
                        out[j * 2 + r2] = pack2(x);
                    }
                if (ok) *(u32x4*)(y_step + pl0 * a.ld_y + yoff) = out;
            }
        }
    };
    const int64_t steps = (a.P + kTM - 1) / kTM;
    if (group < steps) fill(group, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int cur = 0;
    for (int64_t s = group; s < steps; s += n_groups) {
        const int64_t nxt = s + n_groups;
        const bool has_next = nxt < steps;
        if (has_next) fill(nxt, cur ^ 1);  // (in front of this step's stores: see the note on vmcnt above)
        multiply(s, cur);
        // this wave's share of the next image has landed (the eight stores behind it stay in flight); everyone is done reading image cur
        if (has_next) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no wave ends with memory operations in flight (the rule of csrc/wgrad.hip::wgrad3_body)
    if (a.partial) {
        float* const red = (float*)smem;  // [2][C]
        float sv[8], qv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            sv[q] = ssum[q >> 1][q & 1], qv[q] = ssq[q >> 1][q & 1];
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                sv[q] += __shfl_xor(sv[q], d, 64);
                qv[q] += __shfl_xor(qv[q], d, 64);
            }
        }
        if (G::kWP > 1) {
            if (wp == 1 && l15_ == 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    red[wc * 32 + lg_ * 8 + q] = sv[q];
                    red[kC + wc * 32 + lg_ * 8 + q] = qv[q];
                }
            }
            __syncthreads();
        }
        if (wp == 0 && l15_ == 0) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int ch = wc * 32 + lg_ * 8 + q;
                const float s2 = G::kWP > 1 ? red[ch] : 0.f, q2 = G::kWP > 1 ? red[kC + ch] : 0.f;
                if (a.pair) {  // rows of 128 channels: row 2 b + (pixel parity), the paired channel ch is logical channel ch & 127
                    float* row = a.partial + ((int64_t)(blockIdx.x * 2 + (ch >> 7)) * 2) * 128 + (ch & 127);
                    row[0] = sv[q] + s2;
                    row[128] = qv[q] + q2;
                } else {
                    a.partial[((int64_t)blockIdx.x * 2) * kC + ch] = sv[q] + s2;
                    a.partial[((int64_t)blockIdx.x * 2 + 1) * kC + ch] = qv[q] + q2;
                }
            }
        }
    }
}

}  // namespace

// Pointwise path of the tap-conv dispatcher (tapconv.hip::tap_launch): 1x1 stride-1 layers C -> C, C = 256 or 128, plain bf16 in and out,
// at most the batch-statistics epilogue, and enough pixels for one step per workgroup and round.
static bool pointwise_pair(const TapConvArgs* a) {  // 128 -> 128 on dense rows, an even number of pixels: the paired 256-channel view
    return a->C_src == 128 && a->C_dst == 128 && a->ld_src == 128 && a->ld_dst == 128 && (((int64_t)a->N * a->H * a->W_src) & 1) == 0;
}

bool rv_pointwise_plan(const TapConvArgs* a, bool scatter, int* grid, size_t* lds, int* stats_rows) {
    if (a->sel & RV_SEL_NO_POINTWISE) return false;
    if (scatter && (a->sel & RV_SEL_NO_POINTWISE_BWD)) return false;
    if (a->phases != 1 || a->step != 1 || a->tt.ntaps[0] != 1 || a->tt.dh[0][0] != 0 || a->tt.dw[0][0] != 0) return false;
    // 256 input channels, or 128 -> 128 on dense rows as PAIRS of pixels through the same 256-channel instance (a native 128-channel instance was
    // written too and is exact; it is not instantiated: one kernel to soak instead of two -- profiles/r06_ab_notes.md section 4)
    const bool pair = pointwise_pair(a);
    if (!pair && (a->C_src != 256 || a->C_dst % 256 != 0)) return false;
    const int slices = pair ? 1 : a->C_dst / 256;  // C_out = slices x 256: the backward-data of the stem's 9 C -> C fusion conv is nine 256-channel blocks of one input
    if (slices > 16) return false;
    if (a->flags & ~RV_OUT_STATS) return false;  // (no folded input, bias, fp32 output, accumulate, ReLU, BatchNorm-backward sums)
    if (slices > 1 && a->flags) return false;    // (statistics: one row per workgroup of a whole-width launch)
    if (a->W_src != a->W_dst) return false;
    const int64_t P = (int64_t)a->N * a->H * a->W_src;
    const int tm = Pos<256>::kTM;
    if (P * a->ld_src >= ((int64_t)1 << 31) || P * a->ld_dst >= ((int64_t)1 << 31)) return false;  // (32-bit per-lane offsets inside a step only, but keep the tensors addressable the same way)
    const int64_t steps = ((pair ? P / 2 : P) + tm - 1) / tm;
    const int cus = rv_persistent_grid();
    const int groups = 8 * ((cus / 8) / slices);
    if (groups < 8 || steps < ((a->sel & RV_SEL_SMALL_GRIDS) ? 1 : 2 * (int64_t)groups)) return false;  // fewer than two steps per group: the tiled kernels fill the chip better
    // one workgroup per CU; fewer steps than groups (crops in the tests): eight workgroups per row of `slices` x 8 ... keep the grid a multiple of 8 x slices
    int g = cus;
    if (steps < groups) g = 8 * slices * (int)((steps + 7) / 8);
    *grid = g;
    *stats_rows = pair ? 2 * g : g;
    *lds = (size_t)2 * Pos<256>::kBuf;
    return true;
}

int rv_pointwise_launch(const TapConvArgs& a, int grid, size_t lds, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)pointwise_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * Pos<256>::kBuf);
        attr_set = true;
    }
    PwArgs p{};
    p.x = a.src, p.w = a.w, p.y = (bf16_t*)a.dst, p.partial = (a.flags & RV_OUT_STATS) ? a.stats : nullptr;
    p.P = (int64_t)a.N * a.H * a.W_src, p.ld_x = a.ld_src, p.ld_y = a.ld_dst;
    p.pair = pointwise_pair(&a) ? 1 : 0;
    if (p.pair) p.P /= 2, p.ld_x = 256, p.ld_y = 256;
    p.slices = p.pair ? 1 : a.C_dst / 256;
    RV_REQUIRE((p.pair || a.C_src == 256) && grid % 8 == 0 && (grid / 8) / p.slices >= 1, "pointwise kernel: 256 input channels (or paired 128), grid a multiple of 8 with room for a group per XCD");
    hipLaunchKernelGGL(pointwise_kernel<256>, dim3(grid), dim3(512), lds, stream, p);
    RV_CHECK_LAUNCH("pointwise_kernel");
    return 0;
}

// launcher used by rv_pos_backward_sums (bnbwd.hip, which owns the small-K reduction workspace layout)
int rv_pos_bwd_launch(int64_t pixels, const void* dy2, const void* w2_scatter, const void* rel, int32_t ld_rel, int32_t cin,
                      const void* w1_packed, int32_t ld_w1, const float* scale1, const float* shift1, const float* mean1,
                      const float* invstd1, float* partial, int32_t planes, int32_t max_rows, int32_t* rows, int32_t c, hipStream_t stream) {
    RV_REQUIRE(c == 256 || c == 128, "rv_pos_backward_sums: built for 256 or 128 channels (got %d)", c);
    const int lds = c == 256 ? 2 * Pos<256>::kBuf + 2 * Pos<256>::kTM * 8 : 2 * Pos<128>::kBuf + 2 * Pos<128>::kTM * 8;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)pos_bwd_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * Pos<256>::kBuf + 2 * Pos<256>::kTM * 8);
        (void)hipFuncSetAttribute((const void*)pos_bwd_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * Pos<128>::kBuf + 2 * Pos<128>::kTM * 8);
        attr_set = true;
    }
    PosBwdArgs a{};
    a.dy2 = (const bf16_t*)dy2, a.w2s = (const bf16_t*)w2_scatter, a.rel = (const bf16_t*)rel, a.w1 = (const bf16_t*)w1_packed;
    a.scale1 = scale1, a.shift1 = shift1, a.mean1 = mean1, a.invstd1 = invstd1;
    a.partial = partial, a.P = pixels, a.ld_rel = ld_rel, a.ld_w1 = ld_w1, a.cin = cin, a.planes = planes;
    const int64_t steps = (pixels + kTMmin - 1) / kTMmin;
    int grid = (int)(steps < 256 ? steps : 256);
    if (grid > max_rows) grid = max_rows;  // (the caller's partial-row buffer; a persistent workgroup takes any number of steps)
    *rows = grid;
    if (c == 256) hipLaunchKernelGGL(pos_bwd_kernel<256>, dim3(*rows), dim3(512), lds, stream, a);
    else hipLaunchKernelGGL(pos_bwd_kernel<128>, dim3(*rows), dim3(512), lds, stream, a);
    RV_CHECK_LAUNCH("pos_bwd_kernel");
    return 0;
}

extern "C" int32_t rv_pos_forward_rows(int64_t pixels) {
    const int64_t steps = (pixels + kTMmin - 1) / kTMmin;
    return (int32_t)(steps < 256 ? steps : 256);
}

extern "C" int rv_pos_forward(const void* rel, int32_t ld_rel, int32_t cin, int64_t pixels, const void* w1_packed, int32_t ld_w1,
                              const float* scale1, const float* shift1, const void* w2_packed, int32_t c, void* h1, void* y2,
                              float* stats_partial, rvStream stream) {
    RV_REQUIRE(rel && w1_packed && scale1 && shift1 && w2_packed && h1 && y2, "rv_pos_forward: null argument");
    RV_REQUIRE(c == 256 || c == 128, "rv_pos_forward: built for 256 or 128 channels (got %d)", c);
    RV_REQUIRE(cin >= 1 && cin <= 3 && ld_rel >= 4 && ld_rel % 4 == 0 && pixels > 0, "rv_pos_forward: bad shape");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)pos_fwd_kernel<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * Pos<256>::kBuf);
        (void)hipFuncSetAttribute((const void*)pos_fwd_kernel<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * Pos<128>::kBuf);
        attr_set = true;
    }
    PosFwdArgs a{};
    a.rel = (const bf16_t*)rel, a.w1 = (const bf16_t*)w1_packed, a.w2 = (const bf16_t*)w2_packed;
    a.scale1 = scale1, a.shift1 = shift1;
    a.h1 = (bf16_t*)h1, a.y2 = (bf16_t*)y2, a.partial = stats_partial;
    a.P = pixels, a.ld_rel = ld_rel, a.ld_w1 = ld_w1, a.cin = cin;
    if (c == 256) hipLaunchKernelGGL((pos_fwd_kernel<256, false>), dim3(rv_pos_forward_rows(pixels)), dim3(512), 2 * Pos<256>::kBuf, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((pos_fwd_kernel<128, false>), dim3(rv_pos_forward_rows(pixels)), dim3(512), 2 * Pos<128>::kBuf, (hipStream_t)stream, a);
    RV_CHECK_LAUNCH("pos_fwd_kernel");
    return 0;
}

extern "C" int rv_pos_modulate_forward(const void* rel, int32_t ld_rel, int32_t cin, const void* w1_packed, int32_t ld_w1, const float* scale1,
                                       const float* shift1, const void* w2_packed, int32_t c, const float* scale2, const float* shift2,
                                       const void* feat, int32_t ld_feat, int32_t N, int32_t H, int32_t W, void* geo, rvStream stream) {
    RV_REQUIRE(rel && w1_packed && scale1 && shift1 && w2_packed && scale2 && shift2 && feat && geo, "rv_pos_modulate_forward: null argument");
    RV_REQUIRE(c == 256 || c == 128, "rv_pos_modulate_forward: built for 256 or 128 channels (got %d)", c);
    RV_REQUIRE(cin >= 1 && cin <= 3 && ld_rel >= 4 && ld_rel % 4 == 0, "rv_pos_modulate_forward: bad shape");
    RV_REQUIRE(N > 0 && H > 0 && W >= 32 && ld_feat >= c && ld_feat % 8 == 0, "rv_pos_modulate_forward: W >= 32, feature rows of at least c channels");
    const int64_t pixels = (int64_t)N * H * W * 9;
    RV_REQUIRE(pixels < ((int64_t)1 << 31) - 512, "rv_pos_modulate_forward: 9 N H W must stay below 2^31");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)pos_fwd_kernel<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * Pos<256>::kBuf);
        (void)hipFuncSetAttribute((const void*)pos_fwd_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * Pos<128>::kBuf);
        attr_set = true;
    }
    PosFwdArgs a{};
    a.rel = (const bf16_t*)rel, a.w1 = (const bf16_t*)w1_packed, a.w2 = (const bf16_t*)w2_packed;
    a.scale1 = scale1, a.shift1 = shift1, a.scale2 = scale2, a.shift2 = shift2;
    a.feat = (const bf16_t*)feat, a.ld_feat = ld_feat, a.H = H, a.W = W;
    a.h1 = nullptr, a.y2 = (bf16_t*)geo, a.partial = nullptr;
    a.P = pixels, a.ld_rel = ld_rel, a.ld_w1 = ld_w1, a.cin = cin;
    if (c == 256) hipLaunchKernelGGL((pos_fwd_kernel<256, true>), dim3(rv_pos_forward_rows(pixels)), dim3(512), 2 * Pos<256>::kBuf, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((pos_fwd_kernel<128, true>), dim3(rv_pos_forward_rows(pixels)), dim3(512), 2 * Pos<128>::kBuf, (hipStream_t)stream, a);
    RV_CHECK_LAUNCH("pos_fwd_kernel<eval>");
    return 0;
}

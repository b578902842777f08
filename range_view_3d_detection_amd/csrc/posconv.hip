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

namespace {

constexpr int kC = 256;                 // channels of both layers
constexpr int kTM = 128;                // pixels per step
constexpr int kBuf = kTM * kC * 2;      // 64 KB per LDS image

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
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef rv_elem_t bf16x2_t __attribute__((ext_vector_type(2)));
// two floats -> one dword of bf16 (round to nearest even): ONE v_cvt_pk_bf16_f32
__device__ __forceinline__ uint32_t pack2(const f32x2 v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t)); }
__device__ __forceinline__ f32x2 max0(const f32x2 v) { return f32x2{fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)}; }

__global__ __launch_bounds__(512, 1) void pos_fwd_kernel(const PosFwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);  // wave = 32-channel slice of the output, all 128 pixels of a step
    const int l15_ = lane & 15, lg_ = lane >> 4;
    const int l15 = l15_, lg = lg_;

    // ---- second-layer weights of this wave: A operand, row m = l15 of tile j is channel 32 wn + 8 (m >> 2) + 4 j + (m & 3),
    // so that a lane of the result (rows 4 lg + r of both tiles) holds the 8 consecutive channels 32 wn + 8 lg + 0..7
    bf16x8 fw[2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ch = wn * 32 + (l15 >> 2) * 8 + j * 4 + (l15 & 3);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) fw[j][ks] = *(const bf16x8*)(a.w2 + (int64_t)ch * kC + ks * 32 + lg * 8);
    }
    // ---- first layer: this thread generates octet `oct` of pixels pxl, pxl + 16, ... of a step
    const int oct_ = tid & 31, pxl_ = tid >> 5;
    const int oct = oct_;
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
    // `rel` reaches the lanes through SCALAR loads (s_load_dwordx2: a wave generates two pixels per call, lanes 0-31 one,
    // lanes 32-63 the other).  The step loop then contains NO vector-memory load: on gfx9 loads and stores share vmcnt and
    // return out of order with respect to each other, so a wait for any load inside the loop drains every outstanding h1 / y2
    // store first (measured: waves 74 % of their cycles in s_waitcnt, 1.6 ms instead of 0.9).
    const uint64_t* relq = (const uint64_t*)a.rel;
    const int ldq = a.ld_rel / 4;  // row pitch in 8-byte units
    auto generate8 = [&](int64_t step, int buf) __attribute__((always_inline)) {  // this thread's 8 pixels of `step` -> LDS image `buf`, h1
        // (inline asm: hipcc turns these into VECTOR loads because the h1 / y2 stores might alias `rel`; indices are clamped
        //  instead of guarded -- pixels past the end are masked in the epilogue and never stored)
        uint64_t q0[8], q1[8];
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int64_t pu = step * kTM + it * 16 + 2 * wn;  // wave-uniform
            const int64_t p0 = pu < a.P ? pu : a.P - 1, p1 = pu + 1 < a.P ? pu + 1 : a.P - 1;
            asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=s"(q0[it]) : "s"(relq + p0 * ldq));
            asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=s"(q1[it]) : "s"(relq + p1 * ldq));
        }
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+s"(q0[0]), "+s"(q0[1]), "+s"(q0[2]), "+s"(q0[3]), "+s"(q0[4]), "+s"(q0[5]), "+s"(q0[6]), "+s"(q0[7]), "+s"(q1[0]),
                       "+s"(q1[1]), "+s"(q1[2]), "+s"(q1[3]), "+s"(q1[4]), "+s"(q1[5]), "+s"(q1[6]), "+s"(q1[7]));
        int oct = oct_, pxl = pxl_;
        asm volatile("" : "+v"(oct), "+v"(pxl));  // (see multiply: no hoisting of the eight pixels' offsets)
        bf16_t* const h1_step = a.h1 + step * (int64_t)(kTM * kC);  // wave-uniform base: the lane part stays a 32-bit offset
        const int64_t left = a.P - step * kTM;                      // pixels of this step that exist
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int pl = it * 16 + pxl;
            const uint64_t q = lane < 32 ? q0[it] : q1[it];
            const uint32_t lo = (uint32_t)q, hi = (uint32_t)(q >> 32);
            const float x = bf_lo(lo), y = bf_hi(lo), z = bf_lo(hi);
            const f32x2 xx = {x, x}, yy = {y, y}, zz = {z, z};
            u32x4 hv;
#pragma unroll
            for (int j = 0; j < 4; ++j) hv[j] = pack2(max0((gw[j][0] * xx + gw[j][1] * yy + gw[j][2] * zz) * gs[j] + gh[j]));
            *(u32x4*)(smem + buf * kBuf + pl * 512 + ((oct ^ (pl & 15)) * 16)) = hv;
            if (pl < left) *(u32x4*)(h1_step + pl * kC + oct * 8) = hv;
            __builtin_amdgcn_sched_barrier(0);  // one pixel at a time: eight interleaved would need 32 more registers
        }
    };

    f32x2 ssum[4], ssq[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ssum[q] = ssq[q] = f32x2{0.f, 0.f};

    // multiply the image of step `s` (LDS image `cur`) by this wave's weights, store y2, accumulate the statistics
    auto multiply = [&](int64_t s, int cur) __attribute__((always_inline)) {
        const bool full = (s + 1) * kTM <= a.P;  // every pixel of the step exists
        // (opaque copies of the lane coordinates: derived LDS / store offsets are recomputed per step instead of being hoisted
        //  out of the step loop for every unrolled position, which cost ~40 registers and spilled)
        int l15 = l15_, lg = lg_;
        asm volatile("" : "+v"(l15), "+v"(lg));
        const uint8_t* img = smem + cur * kBuf;
        bf16_t* const y2_step = a.y2 + s * (int64_t)(kTM * kC);
#pragma unroll
        for (int half = 0; half < 4; ++half) {  // 32 pixels at a time: 16 accumulator + 16 fragment registers
            f32x4 acc[2][2];
            bf16x8 fa[4][2];
            auto read_fa = [&](int ks, bf16x8 (&f)[2]) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    f[i] = *(const bf16x8*)(img + ((half * 2 + i) * 16 + l15) * 512 + (((ks * 4 + lg) ^ l15) * 16));
            };
            read_fa(0, fa[0]);
            read_fa(1, fa[1]);
            read_fa(2, fa[2]);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 3 < 8) read_fa(ks + 3, fa[(ks + 3) & 3]);  // three K-steps ahead (4 MFMAs = 64 cycles per K-step against ~200 of LDS latency), and no further (sched_barrier below):
#pragma unroll                                                       // hoisting all the fragment reads costs 128 registers -> spills,
                for (int i = 0; i < 2; ++i)                          // and a scratch reload is a VMEM load: vmcnt(0) in the loop
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x4 c0 = ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j];
                        acc[i][j] = RV_MFMA_16x16x32(fw[j][ks], fa[ks & 3][i], c0, 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            // lane (pixel l15 of tile i, lane group lg) holds channels 32 wn + 8 lg + 4 j + r
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pl = (half * 2 + i) * 16 + l15;
                const bool ok = full || pl < a.P - s * kTM;
                u32x4 out;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r2 = 0; r2 < 2; ++r2) {
                        f32x2 x = {acc[i][j][2 * r2], acc[i][j][2 * r2 + 1]};
                        x = ok ? x : f32x2{0.f, 0.f};
                        ssum[j * 2 + r2] += x;
                        ssq[j * 2 + r2] += x * x;
                        out[j * 2 + r2] = pack2(x);
                    }
                if (ok) *(u32x4*)(y2_step + pl * kC + wn * 32 + lg * 8) = out;
            }
        }
    };

    const int64_t steps = (a.P + kTM - 1) / kTM;
    if ((int64_t)blockIdx.x < steps) generate8(blockIdx.x, 0);
    __syncthreads();
    // The two waves of a SIMD (w and w + 4) walk a step in OPPOSITE order -- generate-then-multiply against
    // multiply-then-generate (the image being generated and the one being multiplied are different buffers) -- so that one
    // issues VALU work while the other keeps the matrix pipe busy.
    const bool gen_first = wn < 4;
    int cur = 0;  // k & 1 of the step being multiplied
    for (int64_t s = blockIdx.x; s < steps; s += gridDim.x) {
        const int64_t nxt = s + gridDim.x;
        const bool has_next = nxt < steps;
        if (gen_first) {
            if (has_next) generate8(nxt, cur ^ 1);
            multiply(s, cur);
        } else {
            multiply(s, cur);
            if (has_next) generate8(nxt, cur ^ 1);
        }
        // image cur^1 is complete and everyone is done reading image cur: LDS traffic only -- NOT __syncthreads(), whose
        // vmcnt(0) would wait for this step's 128 KB of global stores
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        cur ^= 1;
    }
    if (a.partial) {
        // per-channel totals of this workgroup: sum over the 16 pixel lanes (l15), then one row per workgroup
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float sv = ssum[q >> 1][q & 1], qv = ssq[q >> 1][q & 1];
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                sv += __shfl_xor(sv, d, 64);
                qv += __shfl_xor(qv, d, 64);
            }
            if (l15 == 0) {
                const int ch = wn * 32 + lg * 8 + q;
                a.partial[((int64_t)blockIdx.x * 2) * kC + ch] = sv;
                a.partial[((int64_t)blockIdx.x * 2 + 1) * kC + ch] = qv;
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

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

__global__ __launch_bounds__(512, 1) void pos_bwd_kernel(const PosBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wn = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15_ = lane & 15, lg_ = lane >> 4;
    const int l15 = l15_, lg = lg_;
    uint8_t* const relbuf = smem + 2 * kBuf;  // [2][128 pixels][8 bytes]

    // B operand: row n = l15 of tile j is input channel ci = 32 wn + 16 j + l15
    bf16x8 fw[2][8];
    f32x2 w1c[3], sc, sh, mu, is;  // the lane's two channels (j = 0, 1) side by side: v_pk_* arithmetic in the epilogue
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int ci = wn * 32 + j * 16 + l15;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) fw[j][ks] = *(const bf16x8*)(a.w2s + (int64_t)ci * kC + ks * 32 + lg * 8);
#pragma unroll
        for (int e = 0; e < 3; ++e) w1c[e][j] = e < a.cin ? bf2f(a.w1[(int64_t)ci * a.ld_w1 + e]) : 0.f;
        sc[j] = a.scale1[ci];
        sh[j] = a.shift1[ci];
        mu[j] = a.mean1[ci];
        is[j] = a.invstd1[ci];
    }
    f32x2 s0 = {0.f, 0.f}, s1 = {0.f, 0.f}, rx = {0.f, 0.f}, ry = {0.f, 0.f}, rz = {0.f, 0.f};

    const int64_t steps = (a.P + kTM - 1) / kTM;
    // image fill: DMA instruction q (0..63) of a step moves pixels 2q, 2q+1 (lane = pixel-in-pair x 32 slots); wave w issues
    // q = w, w + 8, ...  The LDS side is lane-linear, so slot s of pixel p receives source octet s ^ (p & 15).
    __device__ __attribute__((aligned(256))) static uint32_t zero_page[64];
    auto fill = [&](int64_t step, int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int q = wn + 8 * i;
            const int pl = 2 * q + (lane >> 5), slot = lane & 31;
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
        for (int quarter = 0; quarter < 4; ++quarter) {  // 32 pixels at a time
            f32x4 acc[2][2];
            bf16x8 fa[4][2];
            auto read_fa = [&](int ks, bf16x8 (&f)[2]) __attribute__((always_inline)) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    f[i] = *(const bf16x8*)(img + ((quarter * 2 + i) * 16 + l15) * 512 + (((ks * 4 + lg) ^ l15) * 16));
            };
            read_fa(0, fa[0]);
            read_fa(1, fa[1]);
            read_fa(2, fa[2]);
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                if (ks + 3 < 8) read_fa(ks + 3, fa[(ks + 3) & 3]);
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) {
                        const f32x4 c0 = ks == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[i][j];
                        acc[i][j] = RV_MFMA_16x16x32(fa[ks & 3][i], fw[j][ks], c0, 0, 0, 0);
                    }
                __builtin_amdgcn_sched_barrier(0);
            }
            // lane: channel ci(j) = 32 wn + 16 j + l15, pixels (quarter*2 + i)*16 + 4 lg + r
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
    // lanes lg = 0..3 of a channel hold disjoint pixels: sum them, then one row of `planes` x C per workgroup
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float v[5] = {s0[j], s1[j], rx[j], ry[j], rz[j]};
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            v[q] += __shfl_xor(v[q], 16, 64);
            v[q] += __shfl_xor(v[q], 32, 64);
        }
        if (lg == 0) {
            const int ci = wn * 32 + j * 16 + l15;
            float* row = a.partial + (int64_t)blockIdx.x * a.planes * kC + ci;
#pragma unroll
            for (int q = 0; q < 5; ++q) row[(int64_t)q * kC] = v[q];
            for (int q = 5; q < a.planes; ++q) row[(int64_t)q * kC] = 0.f;
        }
    }
}

}  // namespace

// launcher used by rv_pos_backward_sums (bnbwd.hip, which owns the small-K reduction workspace layout)
int rv_pos_bwd_launch(int64_t pixels, const void* dy2, const void* w2_scatter, const void* rel, int32_t ld_rel, int32_t cin,
                      const void* w1_packed, int32_t ld_w1, const float* scale1, const float* shift1, const float* mean1,
                      const float* invstd1, float* partial, int32_t planes, int32_t max_rows, int32_t* rows, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)pos_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kBuf + 2 * kTM * 8);
        attr_set = true;
    }
    PosBwdArgs a{};
    a.dy2 = (const bf16_t*)dy2, a.w2s = (const bf16_t*)w2_scatter, a.rel = (const bf16_t*)rel, a.w1 = (const bf16_t*)w1_packed;
    a.scale1 = scale1, a.shift1 = shift1, a.mean1 = mean1, a.invstd1 = invstd1;
    a.partial = partial, a.P = pixels, a.ld_rel = ld_rel, a.ld_w1 = ld_w1, a.cin = cin, a.planes = planes;
    const int64_t steps = (pixels + kTM - 1) / kTM;
    int grid = (int)(steps < 256 ? steps : 256);
    if (grid > max_rows) grid = max_rows;  // (the caller's partial-row buffer; a persistent workgroup takes any number of steps)
    *rows = grid;
    hipLaunchKernelGGL(pos_bwd_kernel, dim3(*rows), dim3(512), 2 * kBuf + 2 * kTM * 8, stream, a);
    RV_CHECK_LAUNCH("pos_bwd_kernel");
    return 0;
}

extern "C" int32_t rv_pos_forward_rows(int64_t pixels) {
    const int64_t steps = (pixels + kTM - 1) / kTM;
    return (int32_t)(steps < 256 ? steps : 256);
}

extern "C" int rv_pos_forward(const void* rel, int32_t ld_rel, int32_t cin, int64_t pixels, const void* w1_packed, int32_t ld_w1,
                              const float* scale1, const float* shift1, const void* w2_packed, int32_t c, void* h1, void* y2,
                              float* stats_partial, rvStream stream) {
    RV_REQUIRE(rel && w1_packed && scale1 && shift1 && w2_packed && h1 && y2, "rv_pos_forward: null argument");
    RV_REQUIRE(c == kC, "rv_pos_forward: built for %d channels (got %d)", kC, c);
    RV_REQUIRE(cin >= 1 && cin <= 3 && ld_rel >= 4 && ld_rel % 4 == 0 && pixels > 0, "rv_pos_forward: bad shape");
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)pos_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * kBuf);
        attr_set = true;
    }
    PosFwdArgs a{};
    a.rel = (const bf16_t*)rel, a.w1 = (const bf16_t*)w1_packed, a.w2 = (const bf16_t*)w2_packed;
    a.scale1 = scale1, a.shift1 = shift1;
    a.h1 = (bf16_t*)h1, a.y2 = (bf16_t*)y2, a.partial = stats_partial;
    a.P = pixels, a.ld_rel = ld_rel, a.ld_w1 = ld_w1, a.cin = cin;
    hipLaunchKernelGGL(pos_fwd_kernel, dim3(rv_pos_forward_rows(pixels)), dim3(512), 2 * kBuf, (hipStream_t)stream, a);
    RV_CHECK_LAUNCH("pos_fwd_kernel");
    return 0;
}

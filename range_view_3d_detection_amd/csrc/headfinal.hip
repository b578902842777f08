// headfinal.hip -- backward of a tower's FINAL 1x1 conv (C -> n_out <= 32 channels, fp32 logits / regressands) fused with the
// BatchNorm(+ReLU) backward of the tower's last conv -> BatchNorm -> ReLU unit (nn/heads/dense_head.py:44-57, 74-76).
//
// The reference's autograd runs, per tower: conv2d backward-data (writes dA = W^T dY, C channels), native_batch_norm_backward
// (reads dA and y), threshold_backward.  Here the chain was four passes over C-channel tensors of the whole image -- backward-data
// (write dA), BatchNorm-backward reduce (read dA, y), apply (read dA, y, write dy) = 6 transfers of 537 MB at 4 x 64 x 2048 x 512.
// dA is a K = 32 GEMM of the tiny dY: it costs 17 GFLOP to form and 537 MB to store, so it is RECOMPUTED instead:
//   rv_head_final_bwd_sums   reads y and dY:  g = (W^T dY) * [scale*y+shift > 0],  partial rows of (sum g, sum g*xhat)
//   rv_head_final_bwd_apply  reads y and dY again, writes dy = coef0 * (g - coef1 - xhat * coef2)
// = 3 transfers (y twice, dy once) + the 32-channel dY twice.  dA never exists; it is also never rounded to bf16 (the stored
// gradient was), so the sums are formed from fp32 values.
//
// The sums pass also forms the final conv's own WEIGHT gradient dW[o][c] = sum_px dY[px][o] * a[px][c], a = relu(scale*y+shift) (ATen conv2d
// backward-weight; here it was a wgrad2 launch + split-K reduction reading y a third time): the activated tile it has in registers and
// the dY fragment go through a wave-private LDS image and come back TRANSPOSED (ds_read_b64_tr_b16: the reduction index, the pixel, must be
// contiguous per lane) as the operands of v_mfma_f32_16x16x16 (K = the step's 16 pixels); per-range fp32 partials, summed in range
// order by rv_reduce_rows (reproducible: no atomics).
//
// MFMA as D = W' dY^T (v_mfma_f32_16x16x32: M = 16 channels, N = 16 pixels, K = 32 output channels = ONE K step): a lane then
// holds pixel l15 and -- with the rows of the four channel tiles of a wave permuted (tile j, row 4g + r <-> channel 16g + 4j + r) --
// 4 NT consecutive channels of that pixel (NT tiles per wave): y is read and dy written as one 8 NT-byte access per lane, 32 NT
// contiguous bytes per pixel and wave, no LDS.  A workgroup covers 256 channels of one pixel range; the ranges are sized for at
// most 512 partial rows (rv_bn_bwd_finalize's one-launch form takes up to 1024).
#include "common.h"

namespace {

struct HeadFinalArgs {
    const bf16_t* y;    // raw output of the tower's last conv (pre-BatchNorm), [pixels][ld_y]
    const bf16_t* dY;   // gradient w.r.t. the final conv's output as bf16, [pixels][ld_dy], channels n_out .. 31 zero
    const bf16_t* w;    // the final conv's packed SCATTER image: [c][32] (k = output channel contiguous)
    const float *scale, *shift, *mean, *invstd;  // folded BatchNorm of the last unit (the ReLU gate) and its batch statistics
    const float* coef;  // apply: [3][c] from rv_bn_bwd_finalize
    float* partial;     // sums: [rows][2][c]
    float* dw_partial;  // sums: [rows][32][c] partial weight gradients of the final conv (NULL: not formed)
    bf16_t* dy;         // apply: gradient w.r.t. y, [pixels][ld_out]
    int64_t pixels;
    int32_t ld_y, ld_dy, ld_out, c, range, relu;
};

constexpr int kStepPx = 16;
constexpr int kHfTiles = 2;  // 16-channel tiles per wave (head_final_bwd_kernel's NT)

typedef __attribute__((ext_vector_type(4))) short s16x4;
#ifdef RV_OPERAND_F16
typedef __attribute__((ext_vector_type(4))) _Float16 rv_elem4_t;
#define RV_MFMA_16x16x16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_bit_cast(rv_elem4_t, A), __builtin_bit_cast(rv_elem4_t, B), C, 0, 0, 0)
#else
#define RV_MFMA_16x16x16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A, B, C, 0, 0, 0)
#endif
// transposing LDS read: within a 16-lane group the lanes address a 4-row x 16-column block of 2-byte elements (lane = row q, column
// quad p) and lane i of the group receives COLUMN i (4 rows).  Inline asm (the compiler does not track it: lds_tr_wait below).
__device__ __forceinline__ s16x4 lds_read_tr(uint32_t addr) {
    s16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
    return v;
}

// NT: 16-channel MFMA tiles per wave (a lane holds 4 NT consecutive channels of its pixel; a workgroup of 16 / NT waves covers 256
// channels).  NT = 2: ~110 / ~150 registers (apply / sums) against 164 / 236 at NT = 4 -- the passes are bandwidth-bound and want
// the extra waves in flight (measured, one-stream trace: sums + dW 205 us at NT = 4).
template <bool APPLY, int NT>
__global__ __launch_bounds__(1024 / NT) void head_final_bwd_kernel(const HeadFinalArgs a) {
    constexpr int CL = 4 * NT;            // channels per lane
    constexpr int CW = 16 * NT;           // channels per wave
    constexpr int NW = 16 / NT;           // waves per workgroup
    constexpr int kARow = CW * 2 + 16;    // wave-private LDS image of the activated tile: 16 pixel rows x CW channels (+16 B: bank spread)
    constexpr int kDRow = 32 * 2 + 16;    // ... and of the dY tile: 16 pixel rows x 32 output channels
    constexpr int kWaveLds = kStepPx * (kARow + kDRow);
    __shared__ __attribute__((aligned(16))) uint8_t hf_lds[APPLY ? 16 : NW * kWaveLds];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int cb = blockIdx.y * 256 + wave * CW;  // this wave's channels
    const int c0 = cb + CL * lg;                  // this lane's channels
    const int64_t p_begin = (int64_t)blockIdx.x * a.range;
    const int64_t p_end = p_begin + a.range < a.pixels ? p_begin + a.range : a.pixels;

    // A operand: row m of tile j <-> channel cb + CL (m >> 2) + 4 j + (m & 3); lane (m = l15, k = 8 lg .. 8 lg + 7)
    bf16x8 wf[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) wf[j] = *(const bf16x8*)(a.w + (int64_t)(cb + CL * (l15 >> 2) + 4 * j + (l15 & 3)) * 32 + 8 * lg);

    // per-channel constants of this lane's channels (index q = 4 j + r <-> channel c0 + q)
    float sc[CL], sh[CL];
#pragma unroll
    for (int q = 0; q < CL; ++q) {
        sc[q] = a.scale[c0 + q];
        sh[q] = a.shift[c0 + q];
    }
    // ReLU behind the BatchNorm or not, in one register: the gate is `t > thr` and the activated operand `max(t, thr)` -- with
    // thr = -inf (no ReLU) the gate is always open and the operand is scale * y + shift itself (round-5 advice: it used to be a constant 1)
    const float thr = a.relu ? 0.f : -__builtin_inff();
    // sums:  s0 = sum g, s1 = sum g * y (turned into sum g * xhat = invstd * (s1 - mean * s0) at the end: two constants fewer in the loop)
    // apply: dy = k0 * g + (ca + cb_ * y),  ca = k0 * (c2 * mean * invstd - c1),  cb_ = -k0 * c2 * invstd   [= k0 (g - c1 - xhat c2)]
    float s0[CL], s1[CL], k0v[APPLY ? CL : 1];
#pragma unroll
    for (int q = 0; q < CL; ++q) {
        if (APPLY) {
            const float k0 = a.coef[c0 + q], c1 = a.coef[a.c + c0 + q], c2 = a.coef[2 * a.c + c0 + q];
            const float is = a.invstd[c0 + q], mu = a.mean[c0 + q];
            s0[q] = k0 * (c2 * mu * is - c1);  // ca
            s1[q] = -k0 * c2 * is;             // cb_
            k0v[APPLY ? q : 0] = k0;
        } else {
            s0[q] = 0.f;
            s1[q] = 0.f;
        }
    }

    typedef __attribute__((ext_vector_type(2 * NT))) uint32_t ypack_t;  // 4 NT bf16 of one pixel
    auto load = [&](int64_t p0, ypack_t& yv, bf16x8& df) {
        int64_t p = p0 + l15;
        const bool ok = p < p_end;
        p = ok ? p : p_end - 1;  // (clamped: the loads stay inside the tensors; the fragment is zeroed instead)
        yv = *(const ypack_t*)(a.y + p * a.ld_y + c0);
        df = *(const bf16x8*)(a.dY + p * a.ld_dy + 8 * lg);
        if (!ok) {
#pragma unroll
            for (int i = 0; i < 8; ++i) df[i] = (rv_elem_t)0.f;
        }
    };

    // weight gradient of the final conv (sums pass): D[o][c] over 2 x NT tiles of 16 x 16; lane holds rows o = 4 lg + r, column c = l15
    const bool with_dw = !APPLY && a.dw_partial != nullptr;
    f32x4 dw[APPLY ? 1 : 2][APPLY ? 1 : NT];
    uint8_t* my_lds = hf_lds + (APPLY ? 0 : wave * kWaveLds);
    const uint32_t lds_a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)my_lds;
    const uint32_t lds_d = lds_a + kStepPx * kARow;
    const int tq = (lane >> 2) & 3, tp = lane & 3;  // transposed reads: row 4 lg + tq, column quad tp
    if (!APPLY) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) dw[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    ypack_t yv, nv;
    bf16x8 df, nf;
    if (p_begin < p_end) load(p_begin, yv, df);
    for (int64_t p0 = p_begin; p0 < p_end; p0 += kStepPx) {
        if (p0 + kStepPx < p_end) load(p0 + kStepPx, nv, nf);  // the next step's operands in flight under this step's arithmetic
        f32x4 acc[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = RV_MFMA_16x16x32(wf[j], df, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        ypack_t ov;
        if (!APPLY && with_dw) *(bf16x8*)(my_lds + kStepPx * kARow + l15 * kDRow + 16 * lg) = df;  // dY tile [pixel l15][o = 8 lg ..]
#pragma unroll
        for (int q = 0; q < CL; q += 2) {
            // channels c0 + q, c0 + q + 1: one packed word of y; accumulator registers (tile q / 4, row q % 4)
            const uint32_t yw = yv[q >> 1];
            const float y0 = bf_lo(yw), y1 = bf_hi(yw);
            float g0 = acc[q >> 2][q & 3], g1 = acc[(q + 1) >> 2][(q + 1) & 3];
            const float t0 = y0 * sc[q] + sh[q], t1 = y1 * sc[q + 1] + sh[q + 1];
            g0 = t0 > thr ? g0 : 0.f;
            g1 = t1 > thr ? g1 : 0.f;
            if (APPLY) {
                const float d0 = k0v[APPLY ? q : 0] * g0 + (s1[q] * y0 + s0[q]);
                const float d1 = k0v[APPLY ? q + 1 : 0] * g1 + (s1[q + 1] * y1 + s0[q + 1]);
                ov[q >> 1] = pack_bf2(d0, d1);
            } else {
                ov[q >> 1] = pack_bf2(fmaxf(t0, thr), fmaxf(t1, thr));  // the activated operand of the final conv (weight gradient)
                s0[q] += g0;
                s0[q + 1] += g1;
                s1[q] += g0 * y0;
                s1[q + 1] += g1 * y1;
            }
        }
        if (APPLY) {
            const int64_t p = p0 + l15;
            if (p < p_end) *(ypack_t*)(a.dy + p * a.ld_out + c0) = ov;
        } else if (with_dw) {
            // activated tile [pixel l15][channels CL lg .. CL lg + CL - 1] -> LDS, then both operands back transposed (K = pixel)
            *(ypack_t*)(my_lds + l15 * kARow + 2 * CL * lg) = ov;
            s16x4 fo[2], fc[NT];
#pragma unroll
            for (int i = 0; i < 2; ++i) fo[i] = lds_read_tr(lds_d + (4 * lg + tq) * kDRow + (16 * i + 4 * tp) * 2);
#pragma unroll
            for (int j = 0; j < NT; ++j) fc[j] = lds_read_tr(lds_a + (4 * lg + tq) * kARow + (16 * j + 4 * tp) * 2);
            if constexpr (NT == 4)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fo[0]), "+v"(fo[1]), "+v"(fc[0]), "+v"(fc[1]), "+v"(fc[NT - 2]), "+v"(fc[NT - 1])::"memory");
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fo[0]), "+v"(fo[1]), "+v"(fc[0]), "+v"(fc[NT - 1])::"memory");
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) dw[APPLY ? 0 : i][APPLY ? 0 : j] = RV_MFMA_16x16x16(fo[i], fc[j], dw[APPLY ? 0 : i][APPLY ? 0 : j]);
        }
        yv = nv;
        df = nf;
    }
    if (!APPLY && with_dw) {
        // dw[i][j][r]: output channel o = 16 i + 4 lg + r; column l15 of tile j = the tile's 16 consecutive channels cb + 16 j + l15
        float* dwr = a.dw_partial + (int64_t)blockIdx.x * 32 * a.c;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) dwr[(int64_t)(16 * i + 4 * lg + r) * a.c + cb + 16 * j + l15] = dw[APPLY ? 0 : i][APPLY ? 0 : j][r];
    }
    if (!APPLY) {
        // lanes with the same lg hold the same channels (sixteen pixels apart): sum over l15, then lane l15 == 0 of each group writes
#pragma unroll
        for (int q = 0; q < CL; ++q) {
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                s0[q] += __shfl_xor(s0[q], d, 64);
                s1[q] += __shfl_xor(s1[q], d, 64);
            }
        }
        if (l15 == 0) {
            float* row = a.partial + (int64_t)blockIdx.x * 2 * a.c;
#pragma unroll
            for (int q = 0; q < CL; ++q) {
                const float is = a.invstd[c0 + q], mu = a.mean[c0 + q];
                row[c0 + q] = s0[q];
                row[a.c + c0 + q] = is * (s1[q] - mu * s0[q]);
            }
        }
    }
}

// pixel ranges: one partial row of the BatchNorm sums AND one [32][c] partial weight gradient each -- as many as keep every CU busy
// with three workgroups (4 x 64 x 2048: 512 ranges of 1024 pixels, 2 x 512 workgroups; 32 MB of fp32 partials beside 537 MB of y)
int64_t head_final_range(int64_t pixels) {
    const int64_t per = (pixels + 511) / 512;
    return (per + kStepPx - 1) / kStepPx * kStepPx;
}

int head_final_fill(HeadFinalArgs* a, int64_t pixels, int32_t c, const void* y, int32_t ld_y, const void* dY, int32_t ld_dy, const void* w_scatter,
                    const float* scale, const float* shift, const float* mean, const float* invstd, int32_t relu) {
    RV_REQUIRE(y && dY && w_scatter && scale && shift && mean && invstd, "rv_head_final_bwd: null argument");
    RV_REQUIRE(pixels > 0 && c > 0 && c % 256 == 0, "rv_head_final_bwd: the tower width must be a multiple of 256 channels (got %d)", c);
    RV_REQUIRE(ld_y >= c && ld_y % 8 == 0 && ld_dy >= 32 && ld_dy % 8 == 0, "rv_head_final_bwd: bad channel strides (%d, %d)", ld_y, ld_dy);
    memset(a, 0, sizeof(*a));
    a->y = (const bf16_t*)y;
    a->dY = (const bf16_t*)dY;
    a->w = (const bf16_t*)w_scatter;
    a->scale = scale;
    a->shift = shift;
    a->mean = mean;
    a->invstd = invstd;
    a->pixels = pixels;
    a->ld_y = ld_y;
    a->ld_dy = ld_dy;
    a->c = c;
    a->relu = relu;
    a->range = (int32_t)head_final_range(pixels);
    return 0;
}

}  // namespace

extern "C" {

int32_t rv_head_final_bwd_rows(int64_t pixels) {
    if (pixels <= 0) return 0;
    const int64_t range = head_final_range(pixels);
    return (int32_t)((pixels + range - 1) / range);
}

int rv_head_final_bwd_sums(int64_t pixels, int32_t c, const void* y, int32_t ld_y, const void* dY, int32_t ld_dy, const void* w_scatter,
                           const float* scale, const float* shift, const float* mean, const float* invstd, int32_t relu, float* partial,
                           float* dw_partial, rvStream stream) {
    HeadFinalArgs a;
    if (head_final_fill(&a, pixels, c, y, ld_y, dY, ld_dy, w_scatter, scale, shift, mean, invstd, relu)) return 1;
    RV_REQUIRE(partial, "rv_head_final_bwd_sums: null partial buffer");
    a.partial = partial;
    a.dw_partial = dw_partial;
    hipLaunchKernelGGL((head_final_bwd_kernel<false, kHfTiles>), dim3(rv_head_final_bwd_rows(pixels), c / 256), dim3(1024 / kHfTiles), 0, (hipStream_t)stream, a);
    RV_CHECK_LAUNCH("head_final_bwd_kernel<sums>");
    return 0;
}

int rv_head_final_bwd_apply(int64_t pixels, int32_t c, const void* y, int32_t ld_y, const void* dY, int32_t ld_dy, const void* w_scatter,
                            const float* scale, const float* shift, const float* mean, const float* invstd, int32_t relu, const float* coef,
                            void* dy, int32_t ld_out, rvStream stream) {
    HeadFinalArgs a;
    if (head_final_fill(&a, pixels, c, y, ld_y, dY, ld_dy, w_scatter, scale, shift, mean, invstd, relu)) return 1;
    RV_REQUIRE(coef && dy && ld_out >= c && ld_out % 8 == 0, "rv_head_final_bwd_apply: null argument / bad stride");
    a.coef = coef;
    a.dy = (bf16_t*)dy;
    a.ld_out = ld_out;
    hipLaunchKernelGGL((head_final_bwd_kernel<true, kHfTiles>), dim3(rv_head_final_bwd_rows(pixels), c / 256), dim3(1024 / kHfTiles), 0, (hipStream_t)stream, a);
    RV_CHECK_LAUNCH("head_final_bwd_kernel<apply>");
    return 0;
}

}  // extern "C"

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
// SIXTEEN consecutive channels 16g .. 16g + 15 of that pixel: y is read and dy written as two 16-byte accesses per lane, 128
// contiguous bytes per pixel and wave, no LDS.  A workgroup is four waves = 256 channels of one pixel range; the ranges are sized
// for at most 1024 partial rows (rv_bn_bwd_finalize's one-launch form).
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
constexpr int kARow = 64 * 2 + 16;   // wave-private LDS image of the activated tile: 16 pixel rows x 64 channels (+16 B: bank spread)
constexpr int kDRow = 32 * 2 + 16;   // ... and of the dY tile: 16 pixel rows x 32 output channels
constexpr int kWaveLds = kStepPx * (kARow + kDRow);

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

template <bool APPLY>
__global__ __launch_bounds__(256) void head_final_bwd_kernel(const HeadFinalArgs a) {
    __shared__ __attribute__((aligned(16))) uint8_t hf_lds[APPLY ? 16 : 4 * kWaveLds];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, lg = lane >> 4;
    const int cb = blockIdx.y * 256 + wave * 64;  // this wave's 64 channels
    const int c0 = cb + 16 * lg;                  // this lane's 16 channels
    const int64_t p_begin = (int64_t)blockIdx.x * a.range;
    const int64_t p_end = p_begin + a.range < a.pixels ? p_begin + a.range : a.pixels;

    // A operand: row m of tile j <-> channel cb + 16 (m >> 2) + 4 j + (m & 3); lane (m = l15, k = 8 lg .. 8 lg + 7)
    bf16x8 wf[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) wf[j] = *(const bf16x8*)(a.w + (int64_t)(cb + 16 * (l15 >> 2) + 4 * j + (l15 & 3)) * 32 + 8 * lg);

    // per-channel constants of this lane's 16 channels (index q = 4 j + r <-> channel c0 + q)
    float sc[16], sh[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        sc[q] = a.scale[c0 + q];
        sh[q] = a.shift[c0 + q];
    }
    // sums:  s0 = sum g, s1 = sum g * y (turned into sum g * xhat = invstd * (s1 - mean * s0) at the end: two constants fewer in the loop)
    // apply: dy = k0 * g + (ca + cb_ * y),  ca = k0 * (c2 * mean * invstd - c1),  cb_ = -k0 * c2 * invstd   [= k0 (g - c1 - xhat c2)]
    float s0[16], s1[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        if (APPLY) {
            const float k0 = a.coef[c0 + q], c1 = a.coef[a.c + c0 + q], c2 = a.coef[2 * a.c + c0 + q];
            const float is = a.invstd[c0 + q], mu = a.mean[c0 + q];
            s0[q] = k0 * (c2 * mu * is - c1);  // ca
            s1[q] = -k0 * c2 * is;             // cb_
            sh[q] = a.relu ? sh[q] : 1.f;      // (no ReLU behind the BatchNorm: the gate is always open)
            sc[q] = a.relu ? sc[q] : 0.f;
        } else {
            s0[q] = 0.f;
            s1[q] = 0.f;
            sh[q] = a.relu ? sh[q] : 1.f;
            sc[q] = a.relu ? sc[q] : 0.f;
        }
    }
    float k0v[APPLY ? 16 : 1];
    if (APPLY) {
#pragma unroll
        for (int q = 0; q < 16; ++q) k0v[q] = a.coef[c0 + q];
    }

    auto load = [&](int64_t p0, u32x4& ya, u32x4& yb, bf16x8& df) {
        int64_t p = p0 + l15;
        const bool ok = p < p_end;
        p = ok ? p : p_end - 1;  // (clamped: the loads stay inside the tensors; the fragment is zeroed instead)
        const bf16_t* yp = a.y + p * a.ld_y + c0;
        ya = *(const u32x4*)yp;
        yb = *(const u32x4*)(yp + 8);
        df = *(const bf16x8*)(a.dY + p * a.ld_dy + 8 * lg);
        if (!ok) {
#pragma unroll
            for (int i = 0; i < 8; ++i) df[i] = (rv_elem_t)0.f;
        }
    };

    // weight gradient of the final conv (sums pass): D[o][c] over 2 x 4 tiles of 16 x 16; lane holds rows o = 4 lg + r, column c = l15
    const bool with_dw = !APPLY && a.dw_partial != nullptr;
    f32x4 dw[APPLY ? 1 : 2][APPLY ? 1 : 4];
    uint8_t* my_lds = hf_lds + (APPLY ? 0 : wave * kWaveLds);
    const uint32_t lds_a = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)my_lds;
    const uint32_t lds_d = lds_a + kStepPx * kARow;
    const int tq = (lane >> 2) & 3, tp = lane & 3;  // transposed reads: row 4 lg + tq, column quad tp
    if (!APPLY) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) dw[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    u32x4 ya, yb, na, nb;
    bf16x8 df, nf;
    if (p_begin < p_end) load(p_begin, ya, yb, df);
    for (int64_t p0 = p_begin; p0 < p_end; p0 += kStepPx) {
        if (p0 + kStepPx < p_end) load(p0 + kStepPx, na, nb, nf);  // the next step's operands in flight under this step's arithmetic
        f32x4 acc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = RV_MFMA_16x16x32(wf[j], df, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        u32x4 oa, ob;
        if (!APPLY && with_dw) *(bf16x8*)(my_lds + kStepPx * kARow + l15 * kDRow + 16 * lg) = df;  // dY tile [pixel l15][o = 8 lg ..]
#pragma unroll
        for (int q = 0; q < 16; q += 2) {
            // channels c0 + q, c0 + q + 1: one packed word of y; accumulator registers (tile q / 4, row q % 4)
            const uint32_t yw = q < 8 ? ya[q >> 1] : yb[(q - 8) >> 1];
            const float y0 = bf_lo(yw), y1 = bf_hi(yw);
            float g0 = acc[q >> 2][q & 3], g1 = acc[(q + 1) >> 2][(q + 1) & 3];
            const float t0 = y0 * sc[q] + sh[q], t1 = y1 * sc[q + 1] + sh[q + 1];
            g0 = t0 > 0.f ? g0 : 0.f;
            g1 = t1 > 0.f ? g1 : 0.f;
            if (!APPLY) {  // the activated operand of the final conv (relu: scale, shift as given; no relu: the gate constants above are not a)
                const uint32_t o = pack_bf2(fmaxf(t0, 0.f), fmaxf(t1, 0.f));
                if (q < 8) oa[q >> 1] = o;
                else ob[(q - 8) >> 1] = o;
            }
            if (APPLY) {
                const float d0 = k0v[APPLY ? q : 0] * g0 + (s1[q] * y0 + s0[q]);
                const float d1 = k0v[APPLY ? q + 1 : 0] * g1 + (s1[q + 1] * y1 + s0[q + 1]);
                const uint32_t o = pack_bf2(d0, d1);
                if (q < 8) oa[q >> 1] = o;
                else ob[(q - 8) >> 1] = o;
            } else {
                s0[q] += g0;
                s0[q + 1] += g1;
                s1[q] += g0 * y0;
                s1[q + 1] += g1 * y1;
            }
        }
        if (APPLY) {
            const int64_t p = p0 + l15;
            if (p < p_end) {
                bf16_t* op = a.dy + p * a.ld_out + c0;
                *(u32x4*)op = oa;
                *(u32x4*)(op + 8) = ob;
            }
        } else if (with_dw) {
            // activated tile [pixel l15][channels 16 lg .. 16 lg + 15] -> LDS, then both operands back transposed (K = pixel)
            *(u32x4*)(my_lds + l15 * kARow + 32 * lg) = oa;
            *(u32x4*)(my_lds + l15 * kARow + 32 * lg + 16) = ob;
            s16x4 fo[2], fc[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) fo[i] = lds_read_tr(lds_d + (4 * lg + tq) * kDRow + (16 * i + 4 * tp) * 2);
#pragma unroll
            for (int j = 0; j < 4; ++j) fc[j] = lds_read_tr(lds_a + (4 * lg + tq) * kARow + (16 * j + 4 * tp) * 2);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fo[0]), "+v"(fo[1]), "+v"(fc[0]), "+v"(fc[1]), "+v"(fc[2]), "+v"(fc[3])::"memory");
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) dw[APPLY ? 0 : i][APPLY ? 0 : j] = RV_MFMA_16x16x16(fo[i], fc[j], dw[APPLY ? 0 : i][APPLY ? 0 : j]);
        }
        ya = na;
        yb = nb;
        df = nf;
    }
    if (!APPLY && with_dw) {
        // dw[i][j][r]: output channel o = 16 i + 4 lg + r, tower channel cb + 16 j + l15
        float* dwr = a.dw_partial + (int64_t)blockIdx.x * 32 * a.c;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) dwr[(int64_t)(16 * i + 4 * lg + r) * a.c + cb + 16 * j + l15] = dw[APPLY ? 0 : i][APPLY ? 0 : j][r];
    }
    if (!APPLY) {
        // lanes with the same lg hold the same channels (sixteen pixels apart): sum over l15, then lane l15 == 0 of each group writes
#pragma unroll
        for (int q = 0; q < 16; ++q) {
#pragma unroll
            for (int d = 1; d < 16; d <<= 1) {
                s0[q] += __shfl_xor(s0[q], d, 64);
                s1[q] += __shfl_xor(s1[q], d, 64);
            }
        }
        if (l15 == 0) {
            float* row = a.partial + (int64_t)blockIdx.x * 2 * a.c;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
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
    hipLaunchKernelGGL(head_final_bwd_kernel<false>, dim3(rv_head_final_bwd_rows(pixels), c / 256), dim3(256), 0, (hipStream_t)stream, a);
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
    hipLaunchKernelGGL(head_final_bwd_kernel<true>, dim3(rv_head_final_bwd_rows(pixels), c / 256), dim3(256), 0, (hipStream_t)stream, a);
    RV_CHECK_LAUNCH("head_final_bwd_kernel<apply>");
    return 0;
}

}  // extern "C"

// tapconv.hip -- implicit-GEMM "tap convolution" on the gfx950 matrix cores.
//
// One kernel family serves every convolution-like layer of the range-view detector in both
// directions (include/rv3d.h: rv_tap_gather / rv_tap_scatter):
//   dst[n,h,w_dst, d] = sum_{tap, k} Wp[tap][d][k] * f(src[n, h+dh(tap), m*S + dw(tap), k]),  w_dst = P*m + r
// with f = folded BatchNorm (+ReLU) of the producing layer applied while the operand is
// staged (the reference's separate F.pad / BatchNorm2d / ReLU passes never touch HBM).
//
// Mapping to CDNA4:
//   * 256-thread workgroup = 4 wavefronts (2x2); block tile BM x BN = (32*MT) x (32*NT) output
//     pixels x channels of ONE image row and ONE output phase; each wave owns a (16*MT)x(16*NT)
//     sub-tile as MT x NT accumulators of v_mfma_f32_16x16x32_bf16 (fp32 accumulate).
//   * K loop = input-channel chunks of 32 (one MFMA K-step) x taps.  The input halo tile
//     (rows dh_min..dh_max, columns covering the tile for every tap) of a chunk is staged ONCE
//     in LDS and re-read at shifted addresses for every tap: a 3x3 layer reads its input once.
//   * LDS pixel stride is 96 B (64 B of data + 32 B pad): for ds_read_b128 fragment reads the
//     16-byte slot index is (6*pixel + kgroup) mod 16, which is distinct inside each of the
//     hardware's 16-lane service groups at any column shift => conflict-free for S == 1.
//   * global->LDS goes through registers (the operand transform needs the VALU anyway); loads for
//     the next chunk / next tap are issued before the MFMAs of the current step and written
//     after them (issue-early / write-late).
//   * epilogue: optional bias, per-channel sum / sum-of-squares of the fp32 accumulators
//     (BatchNorm batch statistics; wave shuffles, one partial row per wave, no atomics =>
//     bitwise reproducible), then the tile is transposed through LDS and stored as whole
//     16-byte channel runs (coalesced NHWC rows).
#include <stdlib.h>

#include "common.h"
#include "tapconv.h"

namespace {

constexpr int kPix = 48;     // LDS elements per pixel row: 32 data + 16 pad (96 B)
constexpr int kNaMax = 7;    // max 16-byte chunks of the A halo per thread

template <int MT, int NT>
__global__ __launch_bounds__(256, 2) void tapconv_kernel(const TapConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int BM = 32 * MT, BN = 32 * NT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;

    int bx = blockIdx.x;
    const int mt = bx % a.m_tiles;
    bx /= a.m_tiles;
    const int h = bx % a.H;
    bx /= a.H;
    const int n = bx % a.N;
    const int ph = bx / a.N;
    const int n0 = blockIdx.y * BN;
    const int m0 = mt * BM;
    const int T = a.tt.ntaps[ph];
    const int Wt = a.tt.w_tile[ph];
    const int R = a.tt.rows;
    const int S = a.step;

    bf16_t* ldsA = (bf16_t*)smem;
    bf16_t* ldsB = ldsA + a.lds_a_elems;

    const bf16_t* src_row = a.src + ((int64_t)(n * a.H + h) * a.W_src) * a.ld_src;
    // per-tap LDS offsets from an LDS table (a runtime-indexed kernarg lookup would be a vector load whose vmcnt(0)
    // wait drains the prefetch loads -- see tapconv3.hip)
    int* tap_tab = (int*)(smem + a.lds_tab_offset);
    if (tid < T) tap_tab[tid] = ((a.tt.dh[ph][tid] - a.tt.dh_min) * Wt + (a.tt.dw[ph][tid] - a.tt.dw_min[ph])) * kPix;


    // ---- per-thread A-halo slots (fixed for the whole K loop) ------------------------------
    int src_off[kNaMax];
    int lds_off[kNaMax];
    const int oct = tid & 3;
    const int n_slots = R * Wt;
#pragma unroll
    for (int i = 0; i < kNaMax; ++i) {
        const int slot = (tid >> 2) + 64 * i;
        src_off[i] = INT32_MIN;
        lds_off[i] = -1;
        if (slot < n_slots) {
            const int rr = slot / Wt, cc = slot - rr * Wt;
            const int hs = h + a.tt.dh_min + rr;
            const int ws = m0 * S + a.tt.dw_min[ph] + cc;
            lds_off[i] = slot * kPix + oct * 8;
            if (hs >= 0 && hs < a.H && ws >= 0 && ws < a.W_src)
                src_off[i] = ((a.tt.dh_min + rr) * a.W_src + ws) * a.ld_src + oct * 8;
        }
    }
    // ---- per-thread B slots -----------------------------------------------------------------
    const int b_row = tid >> 1, b_half = tid & 1;
    const bool b_active = (b_row < BN);
    const bool b_valid = b_active && (n0 + b_row < a.C_dst);
    const int64_t w_img = (int64_t)a.C_dst * a.C_src;  // elements per tap image
    const int w_base_idx = a.tt.w_first[ph];

    u32x4 a_reg[kNaMax];
    u32x4 b_reg[2];

    auto load_a = [&](int kc) {
#pragma unroll
        for (int i = 0; i < kNaMax; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (src_off[i] != INT32_MIN) v = *(const u32x4*)(src_row + src_off[i] + kc * 32);
            a_reg[i] = v;
        }
    };
    auto write_a = [&](int kc) {
        const bool affine = a.flags & RV_IN_AFFINE, relu = a.flags & RV_IN_RELU;
        float sc[8], sh[8];
        if (affine) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sc[j] = a.in_scale[kc * 32 + oct * 8 + j];
                sh[j] = a.in_shift[kc * 32 + oct * 8 + j];
            }
        }
#pragma unroll
        for (int i = 0; i < kNaMax; ++i) {
            if (lds_off[i] < 0) continue;
            pin_here(a_reg[i]);
            u32x4 v = a_reg[i];
            if ((affine || relu) && src_off[i] != INT32_MIN) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float lo = bf_lo(v[j]), hi = bf_hi(v[j]);
                    if (affine) {
                        lo = lo * sc[2 * j] + sh[2 * j];
                        hi = hi * sc[2 * j + 1] + sh[2 * j + 1];
                    }
                    if (relu) {
                        lo = fmaxf(lo, 0.f);
                        hi = fmaxf(hi, 0.f);
                    }
                    v[j] = pack_bf2(lo, hi);
                }
            }
            *(u32x4*)(ldsA + lds_off[i]) = v;
        }
    };
    auto load_b = [&](int kc, int t) {
        b_reg[0] = u32x4{0u, 0u, 0u, 0u};
        b_reg[1] = u32x4{0u, 0u, 0u, 0u};
        if (b_valid) {
            const bf16_t* p = a.w + (int64_t)(w_base_idx + t) * w_img + (int64_t)(n0 + b_row) * a.C_src + kc * 32 +
                              b_half * 16;
            b_reg[0] = *(const u32x4*)p;
            b_reg[1] = *(const u32x4*)(p + 8);
        }
    };
    auto write_b = [&](int buf) {
        if (b_active) {
            bf16_t* p = ldsB + buf * (BN * kPix) + b_row * kPix + b_half * 16;
            pin_here(b_reg[0]);
            pin_here(b_reg[1]);
            *(u32x4*)p = b_reg[0];
            *(u32x4*)(p + 8) = b_reg[1];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int a_lane = (wm * MT * 16 + l15) * S * kPix + lg * 8;
    const int b_lane = (wn * NT * 16 + l15) * kPix + lg * 8;

    // a phase of a strided scatter can have no tap at all (1x1 stride-2 projection: odd columns receive nothing)
    const int nk = (T > 0) ? a.C_src / 32 : 0;
    if (nk > 0) {
        load_a(0);
        load_b(0, 0);
        write_a(0);
        write_b(0);
    }
    __syncthreads();
    int buf = 0;
    for (int kc = 0; kc < nk; ++kc) {
        const bool next_chunk = kc + 1 < nk;
        load_a(next_chunk ? kc + 1 : kc);  // unconditional (see tapconv3.hip)
        for (int t = 0; t < T; ++t) {
            const bool last_tap = (t == T - 1);
            const bool has_next = !last_tap || next_chunk;
            load_b(has_next ? (last_tap ? kc + 1 : kc) : kc, has_next ? (last_tap ? 0 : t + 1) : t);
            // ---- MFMAs of tap t on chunk kc ----
            {
                const int tap_off = tap_tab[t];
                const bf16_t* pa = ldsA + tap_off + a_lane;
                const bf16_t* pb = ldsB + buf * (BN * kPix) + b_lane;
                bf16x8 fb[NT];
#pragma unroll
                for (int j = 0; j < NT; ++j) fb[j] = *(const bf16x8*)(pb + j * 16 * kPix);
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    const bf16x8 fa = *(const bf16x8*)(pa + i * 16 * S * kPix);
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = RV_MFMA_16x16x32(fa, fb[j], acc[i][j], 0, 0, 0);
                }
            }
            write_b(buf ^ 1);
            if (last_tap && next_chunk) {
                __syncthreads();
                write_a(kc + 1);
            }
            __syncthreads();
            buf ^= 1;
        }
    }

    // ------------------------------------ epilogue --------------------------------------------
    const int Wm = a.W_dst / a.phases;  // valid m per phase
    // rows (pixels) this lane holds: m = m0 + wm*MT*16 + i*16 + lg*4 + reg ; column n = n0 + wn*NT*16 + j*16 + l15
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm * MT * 16 + i * 16 + lg * 4 + r;
            if (m >= Wm) {
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j][r] = 0.f;
            }
        }
    }
    if (a.flags & RV_OUT_STATS) {
        // BatchNorm statistics of the bias-free fp32 result; one partial row per (block, wave-row)
        float* prow = a.stats + ((int64_t)(blockIdx.x * 2 + wm) * 2) * a.C_dst;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[i][j][r];
                    s += v;
                    q += v * v;
                }
            s += __shfl_xor(s, 16, 64);
            q += __shfl_xor(q, 16, 64);
            s += __shfl_xor(s, 32, 64);
            q += __shfl_xor(q, 32, 64);
            const int c = n0 + wn * NT * 16 + j * 16 + l15;
            if (lg == 0 && c < a.C_dst) {
                prow[c] = s;
                prow[a.C_dst + c] = q;
            }
        }
    }
    if (a.flags & RV_OUT_BIAS) {
        const bool relu_out = (a.flags & RV_OUT_RELU) != 0;  // eval: BatchNorm folded into weights + bias, ReLU on the way out
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int c = n0 + wn * NT * 16 + j * 16 + l15;
            const float b = (c < a.C_dst) ? a.bias[c] : 0.f;
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = relu_out ? fmaxf(acc[i][j][r] + b, 0.f) : acc[i][j][r] + b;
        }
    }
    const int64_t dst_row = ((int64_t)(n * a.H + h) * a.W_dst) * a.ld_dst;
    if (a.flags & RV_OUT_F32) {
        float* dst = (float*)a.dst + dst_row;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * MT * 16 + i * 16 + lg * 4 + r;
                if (m >= Wm) continue;
                const int64_t px = (int64_t)(a.phases * m + ph) * a.ld_dst;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const int c = n0 + wn * NT * 16 + j * 16 + l15;
                    if (c < a.C_dst) dst[px + c] = acc[i][j][r];
                }
            }
        return;
    }
    // bf16: transpose through LDS, then whole 16-byte channel runs per pixel
    constexpr int kEpi = BN + 8;  // row stride (elements): 16-byte aligned, breaks the power of two
    bf16_t* epi = (bf16_t*)smem;
    // (the K loop ended with a barrier: every wave is done with ldsA / ldsB)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pm = wm * MT * 16 + i * 16 + lg * 4 + r;
                const int pc = wn * NT * 16 + j * 16 + l15;
                epi[pm * kEpi + pc] = f2bf(acc[i][j][r]);
            }
    __syncthreads();
    bf16_t* dst = (bf16_t*)a.dst + dst_row;
    const int64_t res_row = ((int64_t)(n * a.H + h) * a.W_dst) * a.ld_res;
    constexpr int kChunks = BN / 8;
    const bool accum = a.flags & RV_OUT_ACCUM;
    for (int q = tid; q < BM * kChunks; q += 256) {
        const int pm = q / kChunks, c8 = q - pm * kChunks;
        const int m = m0 + pm, c = n0 + c8 * 8;
        if (m >= Wm || c >= a.C_dst) continue;
        u32x4 v = *(const u32x4*)(epi + pm * kEpi + c8 * 8);
        bf16_t* p = dst + (int64_t)(a.phases * m + ph) * a.ld_dst + c;
        if (accum) {
            const u32x4 o = *(const u32x4*)(a.res + res_row + (int64_t)(a.phases * m + ph) * a.ld_res + c);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = pack_bf2(bf_lo(v[j]) + bf_lo(o[j]), bf_hi(v[j]) + bf_hi(o[j]));
            if (a.flags & RV_OUT_RES_RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = pack_bf2(fmaxf(bf_lo(v[j]), 0.f), fmaxf(bf_hi(v[j]), 0.f));
            }
        }
        *(u32x4*)p = v;
    }
}

template <int MT, int NT>
int launch(const TapConvArgs& a, int grid_x, int grid_y, size_t lds, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)tapconv_kernel<MT, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    hipLaunchKernelGGL((tapconv_kernel<MT, NT>), dim3(grid_x, grid_y), dim3(256), lds, stream, a);
    RV_CHECK_LAUNCH("tapconv_kernel");
    return 0;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// host side: tap tables + launch configuration
// ---------------------------------------------------------------------------------------------
int rv_build_tap_table(const rvTapGeom* g, bool scatter, TapTable* tt, int* phases, int* step) {
    memset(tt, 0, sizeof(*tt));
    RV_REQUIRE(g->kh >= 1 && g->kw >= 1 && g->kh * g->kw <= kMaxTaps, "tap geometry: kernel %dx%d unsupported", g->kh, g->kw);
    RV_REQUIRE(g->stride_w == 1 || g->stride_w == 2 || g->stride_w == 4, "tap geometry: stride_w %d unsupported", g->stride_w);
    int dh_lo = 127, dh_hi = -127;
    if (!scatter) {
        // U[h, wu] <- V[h + ky - pad_h, wu*s + kx - pad_w]
        *phases = 1;
        *step = g->stride_w;
        tt->ntaps[0] = g->kh * g->kw;
        tt->w_first[0] = 0;
        int lo = 127, hi = -127;
        for (int ky = 0; ky < g->kh; ++ky)
            for (int kx = 0; kx < g->kw; ++kx) {
                const int t = ky * g->kw + kx;
                tt->ky[0][t] = (int8_t)ky;
                tt->kx[0][t] = (int8_t)kx;
                tt->dh[0][t] = (int8_t)(ky - g->pad_h);
                tt->dw[0][t] = (int8_t)(kx - g->pad_w);
                lo = lo < kx - g->pad_w ? lo : kx - g->pad_w;
                hi = hi > kx - g->pad_w ? hi : kx - g->pad_w;
                dh_lo = dh_lo < ky - g->pad_h ? dh_lo : ky - g->pad_h;
                dh_hi = dh_hi > ky - g->pad_h ? dh_hi : ky - g->pad_h;
            }
        tt->dw_min[0] = lo;
        tt->dw_max[0] = hi;
    } else {
        // V[h, s*m + r] <- U[h - ky + pad_h, m + floor((r+pad_w)/s) - j],  kx = (r+pad_w)%s + s*j
        const int s = g->stride_w;
        *phases = s;
        *step = 1;
        int first = 0;
        for (int r = 0; r < s; ++r) {
            int cnt = 0, lo = 127, hi = -127;
            const int kx0 = (r + g->pad_w) % s, base = (r + g->pad_w) / s;
            for (int ky = 0; ky < g->kh; ++ky)
                for (int kx = kx0, j = 0; kx < g->kw; kx += s, ++j) {
                    tt->dh[r][cnt] = (int8_t)(g->pad_h - ky);
                    tt->dw[r][cnt] = (int8_t)(base - j);
                    tt->ky[r][cnt] = (int8_t)ky;
                    tt->kx[r][cnt] = (int8_t)kx;
                    lo = lo < base - j ? lo : base - j;
                    hi = hi > base - j ? hi : base - j;
                    dh_lo = dh_lo < g->pad_h - ky ? dh_lo : g->pad_h - ky;
                    dh_hi = dh_hi > g->pad_h - ky ? dh_hi : g->pad_h - ky;
                    ++cnt;
                }
            if (cnt == 0) { lo = 0; hi = 0; }
            tt->ntaps[r] = cnt;
            tt->w_first[r] = first;
            tt->dw_min[r] = lo;
            tt->dw_max[r] = hi;
            first += cnt;
        }
    }
    tt->dh_min = dh_lo;
    tt->rows = dh_hi - dh_lo + 1;
    return 0;
}

static int tap_launch(const rvTapGeom* g, const rvTapShape* s, bool scatter, const void* src, const float* in_scale,
                      const float* in_shift, const void* w, const float* bias, void* dst, float* stats,
                      rvStream stream, bool dry_run, int* stats_rows, int* info = nullptr, const rvBnbEpilogue* bnb = nullptr,
                      int* bnb_rows = nullptr, const void* residual = nullptr, int32_t ld_res = 0) {
    TapConvArgs a;
    memset(&a, 0, sizeof(a));
    int phases, step;
    if (rv_build_tap_table(g, scatter, &a.tt, &phases, &step)) return 1;
    RV_REQUIRE(s->Wv == s->Wu * g->stride_w, "tap shape: Wv (%d) must equal Wu (%d) * stride_w (%d)", s->Wv, s->Wu, g->stride_w);
    RV_REQUIRE(s->N > 0 && s->H > 0 && s->Wu > 0, "tap shape: empty tensor");
    const int cu = rv_pad32(g->cu), cv = rv_pad32(g->cv);
    a.N = s->N;
    a.H = s->H;
    a.W_src = scatter ? s->Wu : s->Wv;
    a.W_dst = scatter ? s->Wv : s->Wu;
    a.C_src = scatter ? cu : cv;
    a.C_dst = scatter ? cv : cu;
    a.ld_src = s->ld_src;
    a.ld_dst = s->ld_dst;
    RV_REQUIRE(a.ld_src >= a.C_src && a.ld_dst >= a.C_dst, "tap shape: channel strides (%d,%d) smaller than padded channels (%d,%d)",
               a.ld_src, a.ld_dst, a.C_src, a.C_dst);
    RV_REQUIRE(a.ld_src % 8 == 0 && a.ld_dst % 8 == 0, "tap shape: channel strides must be multiples of 8");
    a.phases = phases;
    a.step = step;
    a.flags = s->flags & ~RV_SEL_MASK;
    a.sel = s->flags & RV_SEL_MASK;  // per-call kernel-selection hints (tests pin a generation / lift the tile-count heuristics)
    a.res = (const bf16_t*)dst;  // RV_OUT_ACCUM adds into dst ...
    a.ld_res = a.ld_dst;
    if (residual) {  // ... rv_tap_residual adds another tensor of the same pixels
        RV_REQUIRE(!(a.flags & (RV_OUT_ACCUM | RV_OUT_STATS)), "rv_tap_residual: not together with RV_OUT_ACCUM / RV_OUT_STATS");
        RV_REQUIRE(ld_res >= a.C_dst && ld_res % 8 == 0, "rv_tap_residual: bad channel stride of the residual (%d)", ld_res);
        a.flags |= RV_OUT_ACCUM;
        a.res = (const bf16_t*)residual;
        a.ld_res = ld_res;
    } else {
        RV_REQUIRE(dry_run || !(a.flags & RV_OUT_RES_RELU), "RV_OUT_RES_RELU belongs to rv_tap_residual");
    }
    RV_REQUIRE(!((a.flags & RV_OUT_F32) && (a.flags & RV_OUT_ACCUM)), "RV_OUT_ACCUM needs a bf16 destination");
    RV_REQUIRE(dry_run || !(a.flags & RV_IN_AFFINE) || (in_scale && in_shift), "RV_IN_AFFINE without scale/shift");
    RV_REQUIRE(dry_run || !(a.flags & RV_OUT_BIAS) || bias, "RV_OUT_BIAS without bias");
    RV_REQUIRE(!(a.flags & RV_OUT_STATS) || stats || dry_run, "RV_OUT_STATS without a partial buffer");
    a.src = (const bf16_t*)src;
    a.dst = dst;
    a.w = (const bf16_t*)w;
    a.in_scale = in_scale;
    a.in_shift = in_shift;
    a.bias = bias;
    a.stats = stats;
    if (bnb_rows) *bnb_rows = 0;
    if (bnb || bnb_rows) {  // backward-data launch that also forms the BatchNorm-backward sums of its destination layer
        a.flags |= RV_OUT_BNB;
        if (bnb) {
            RV_REQUIRE(bnb->y && bnb->scale && bnb->shift && bnb->mean && bnb->invstd && bnb->partial, "rv_tap_data_grad_bnb: null epilogue pointer");
            RV_REQUIRE(bnb->ld_y >= a.C_dst && bnb->ld_y % 8 == 0, "rv_tap_data_grad_bnb: bad channel stride of y (%d)", bnb->ld_y);
            a.bnb_y = (const bf16_t*)bnb->y;
            a.ld_bnb_y = bnb->ld_y;
            a.bnb_flags = bnb->flags;
            a.bnb_scale = bnb->scale;
            a.bnb_shift = bnb->shift;
            a.bnb_mean = bnb->mean;
            a.bnb_invstd = bnb->invstd;
            a.bnb_partial = bnb->partial;
        }
    }

    // multi-tap layers with at least one round of 512-pixel x 128-channel tiles (tapconv6.hip)
    if (!(a.sel & (RV_SEL_NO_GEN6 | RV_SEL_NO_GEN5))) {
        int tiles, srows, brows;
        size_t lds6;
        TapConvArgs a6 = a;
        if (rv_tapconv6_plan(&a6, &tiles, &lds6, &srows, &brows)) {
            if (stats_rows) *stats_rows = srows;
            if (bnb_rows) *bnb_rows = brows;
            if (info) {
                info[0] = 6;
                info[1] = 128;
                info[2] = tiles;
                info[3] = a6.n_tiles;
            }
            if (dry_run) return 0;
            return rv_tapconv6_launch(a6, lds6, (hipStream_t)stream);
        }
    }
    // multi-tap layers with 256-channel output tiles: input halo resident in LDS across the taps (tapconv5.hip)
    if (!(a.sel & RV_SEL_NO_GEN5)) {
        int tiles, bn5;
        size_t lds5;
        TapConvArgs a5 = a;
        if (rv_tapconv5_plan(&a5, &tiles, &lds5, &bn5)) {
            if (stats_rows) *stats_rows = tiles * 2;
            if (bnb_rows) *bnb_rows = tiles;
            if (info) {
                info[0] = 5;
                info[1] = bn5;
                info[2] = tiles;
                info[3] = a5.n_tiles;
            }
            if (dry_run) return 0;
            return rv_tapconv5_launch(a5, lds5, bn5, (hipStream_t)stream);
        }
    }
    if (a.flags & RV_OUT_BNB) {  // only the fifth-generation kernel carries that epilogue
        if (dry_run) return 0;   // (rows = 0: the caller takes the separate reduce pass)
        RV_FAIL("rv_tap_data_grad_bnb: this launch has no fused BatchNorm-backward sums (rv_tap_bnb_rows returned 0)");
    }
    // 1x1 stride-1 C -> C layers on plain tensors: the persistent streaming GEMM with the weights in registers (posconv.hip, round 6)
    {
        int grid7, rows7;
        size_t lds7;
        if (rv_pointwise_plan(&a, scatter, &grid7, &lds7, &rows7)) {
            if (stats_rows) *stats_rows = rows7;
            if (info) {
                info[0] = 7;
                info[1] = a.C_dst;
                info[2] = grid7;
                info[3] = a.C_dst / a.C_src;  // 256-channel output slices per step group (1: also the paired 128 -> 128 form)
            }
            if (dry_run) return 0;
            return rv_pointwise_launch(a, grid7, lds7, (hipStream_t)stream);
        }
    }
    // 256 x 256 (or x 128) tiles streamed by LDS-DMA, counted waits (tapconv4.hip); plain bf16 inputs only
    {
        int tiles, bn, srows;
        size_t lds4;
        TapConvArgs a4 = a;
        if (rv_tapconv4_plan(&a4, &tiles, &lds4, &bn, &srows)) {
            if (stats_rows) *stats_rows = srows;
            if (info) {
                info[0] = 4;
                info[1] = bn;
                info[2] = tiles;
                info[3] = a4.n_tiles;
            }
            if (dry_run) return 0;
            return rv_tapconv4_launch(a4, lds4, bn, (hipStream_t)stream);
        }
    }
    // fast path: 2-row x 64-column tiles, 64-channel chunks (tapconv2.hip)
    {
        int gx, gy, ks;
        size_t lds2;
        if (rv_tapconv2_plan(&a, &gx, &gy, &lds2, &ks)) {
            if (stats_rows) *stats_rows = gx * 2;
            if (info) {
                info[0] = 2;
                info[1] = ks;
                info[2] = gx;
                info[3] = gy;
            }
            if (dry_run) return 0;
            return rv_tapconv2_launch(a, gx, gy, lds2, ks, (hipStream_t)stream);
        }
    }
    // generic path -- tile selection: keep the LDS halo small for strided gathers; narrow N tile for thin outputs
    int mt = (step == 1) ? 4 : (step == 2 ? 2 : 1);
    const int wm_total = a.W_dst / phases;
    while (mt > 1 && 32 * mt / 2 >= wm_total) mt >>= 1;  // tiny images (tests)
    int nt = a.C_dst >= 128 ? 4 : (a.C_dst >= 64 ? 2 : 1);
    const int BM = 32 * mt, BN = 32 * nt;
    int a_elems = 0;
    for (int r = 0; r < phases; ++r) {
        a.tt.w_tile[r] = (BM - 1) * step + (a.tt.dw_max[r] - a.tt.dw_min[r]) + 1;
        const int e = a.tt.rows * a.tt.w_tile[r] * kPix;
        a_elems = a_elems > e ? a_elems : e;
        RV_REQUIRE(a.tt.rows * a.tt.w_tile[r] * 4 <= 256 * kNaMax, "tap conv: halo tile too large (%d x %d)", a.tt.rows, a.tt.w_tile[r]);
    }
    a.lds_a_elems = a_elems;
    size_t lds = (size_t)(a_elems + 2 * BN * kPix) * sizeof(bf16_t);
    const size_t epi = (size_t)BM * (BN + 8) * sizeof(bf16_t);
    if (lds < epi) lds = epi;
    lds = (lds + 15) & ~(size_t)15;
    a.lds_tab_offset = (int32_t)lds;
    lds += 128;
    a.m_tiles = rv_ceil_div(wm_total, BM);
    const int grid_x = a.m_tiles * a.H * a.N * phases;
    const int grid_y = rv_ceil_div(a.C_dst, BN);
    if (stats_rows) *stats_rows = grid_x * 2;
    if (info) {
        info[0] = 1;
        info[1] = mt * 16 + nt;
        info[2] = grid_x;
        info[3] = grid_y;
    }
    if (dry_run) return 0;
    hipStream_t st = (hipStream_t)stream;
#define RV_CASE(M, Nn) \
    if (mt == M && nt == Nn) return launch<M, Nn>(a, grid_x, grid_y, lds, st);
    RV_CASE(4, 4) RV_CASE(4, 2) RV_CASE(4, 1) RV_CASE(2, 4) RV_CASE(2, 2) RV_CASE(2, 1) RV_CASE(1, 4) RV_CASE(1, 2) RV_CASE(1, 1)
#undef RV_CASE
    RV_FAIL("tap conv: no kernel for tile %dx%d", mt, nt);
}

extern "C" {

int32_t rv_tap_stats_rows(const rvTapGeom* g, const rvTapShape* s, int32_t scatter) {
    int rows = 0;
    if (tap_launch(g, s, scatter != 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, true, &rows)) return -1;
    return rows;
}

int rv_tap_launch_info(const rvTapGeom* g, const rvTapShape* s, int32_t scatter, int32_t* host_info) {
    RV_REQUIRE(g && s && host_info, "rv_tap_launch_info: null argument");
    return tap_launch(g, s, scatter != 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, true, nullptr, host_info);
}

int32_t rv_tap_bnb_rows(const rvTapGeom* g, const rvTapShape* s, int32_t scatter) {
    int rows = 0;
    if (!g || !s) return 0;
    if (tap_launch(g, s, scatter != 0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, true, nullptr, nullptr, nullptr, &rows))
        return -1;
    return rows;
}

int rv_tap_data_grad_bnb(const rvTapGeom* g, const rvTapShape* s, int32_t scatter, const void* dout, const void* w, void* dx,
                         const rvBnbEpilogue* e, rvStream stream) {
    RV_REQUIRE(g && s && dout && w && dx && e, "rv_tap_data_grad_bnb: null argument");
    return tap_launch(g, s, scatter != 0, dout, nullptr, nullptr, w, nullptr, dx, nullptr, stream, false, nullptr, nullptr, e);
}

int rv_tap_residual(const rvTapGeom* g, const rvTapShape* s, int32_t scatter, const void* src, const void* w, const float* bias,
                    const void* res, int32_t ld_res, void* dst, rvStream stream) {
    RV_REQUIRE(g && s && src && w && res && dst, "rv_tap_residual: null argument");
    return tap_launch(g, s, scatter != 0, src, nullptr, nullptr, w, bias, dst, nullptr, stream, false, nullptr, nullptr, nullptr, nullptr, res,
                      ld_res);
}

int rv_tap_gather(const rvTapGeom* g, const rvTapShape* s, const void* V, const float* in_scale, const float* in_shift,
                  const void* gather_w, const float* bias, void* U, float* stats_partial, rvStream stream) {
    RV_REQUIRE(g && s && V && gather_w && U, "rv_tap_gather: null argument");
    return tap_launch(g, s, false, V, in_scale, in_shift, gather_w, bias, U, stats_partial, stream, false, nullptr);
}

int rv_tap_scatter(const rvTapGeom* g, const rvTapShape* s, const void* U, const float* in_scale, const float* in_shift,
                   const void* scatter_w, const float* bias, void* V, float* stats_partial, rvStream stream) {
    RV_REQUIRE(g && s && U && scatter_w && V, "rv_tap_scatter: null argument");
    return tap_launch(g, s, true, U, in_scale, in_shift, scatter_w, bias, V, stats_partial, stream, false, nullptr);
}

}  // extern "C"

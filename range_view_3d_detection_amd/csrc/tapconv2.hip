// tapconv2.hip -- second-generation tap-conv kernel for the layers that dominate the step
// (stride-1 gathers and every scatter form on images at least 64 columns wide, >= 64 output channels).
//
// Differences to tapconv_kernel (tapconv.hip), each aimed at a measured bottleneck of round 1
// (539 TFLOP/s, tap steps of only 16 MFMAs between barriers, 50 % LDS padding):
//   * output tile = 2 image rows x 64 columns (x 128 channels): the 3x3 halo is 4 rows x 66 px for
//     128 outputs (2.06x) instead of 3 x 130 (3.05x) => fewer global loads and 41 KB instead of 37 KB of
//     LDS for TWICE the channel depth per chunk;
//   * K chunk = 64 channels (two MFMA K-steps per tap): 32 MFMAs per wave between barriers, half the
//     barriers, prefetch distance of a tap step doubled (covers L2 latency with 2 blocks per CU);
//   * dense LDS images made of 16-byte chunks: A[halo row][k-group][column (padded to 16)][8], B[k-group][128 n][8].
//     A fragment read (16 pixels x 4 k-groups) touches slots (column + const) mod 16 within each hardware
//     service group => conflict-free without padding every pixel row by 50 %;
//   * staging threads are mapped 8 pixels x 8 k-groups per wave: global reads cover whole 128-byte pixel rows
//     (coalesced) and every 8-lane ds_write_b128 group writes 128 contiguous bytes (conflict-free);
//   * 1x1 layers double-buffer the (small) A tile: one barrier per 32-MFMA step instead of two.
#include <stdlib.h>

#include "common.h"
#include "tapconv.h"

namespace {

constexpr int kTC = 64;   // tile columns (m positions)
constexpr int kTR = 2;    // tile rows
constexpr int kBN = 128;  // tile channels
constexpr int kNa2 = 9;   // max A chunks per thread: 4 rows x 66 px x 8 groups / 256 threads

template <int KS, bool BDMA>  // KS: MFMA K-steps per chunk (chunk = 32*KS channels); BDMA: weights by LDS-DMA
__global__ __launch_bounds__(256, 2) void tapconv2_kernel(const TapConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int G = 4 * KS;        // 16-byte k-groups per chunk
    constexpr int BK = 32 * KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;  // wm = tile row, wn = channel half
    const int l15 = lane & 15, lg = lane >> 4;

    int bx = blockIdx.x;
    const int tc = bx % a.m_tiles;
    bx /= a.m_tiles;
    const int th = bx % a.h_tiles;
    bx /= a.h_tiles;
    const int n = bx % a.N;
    const int ph = bx / a.N;
    const int n0 = blockIdx.y * kBN;
    const int m0 = tc * kTC, h0 = th * kTR;
    const int T = a.tt.ntaps[ph];
    const int Wt = a.tt.w_tile[ph];           // halo columns actually staged
    const int Wtp = (Wt + 15) & ~15;          // padded column count of the LDS image
    const int HR = kTR + a.tt.rows - 1;       // halo rows
    const int a_buf_elems = HR * G * Wtp * 8;
    const bool a_double = (T == 1);

    bf16_t* ldsA = (bf16_t*)smem;
    bf16_t* ldsB = ldsA + a.lds_a_elems;  // host: lds_a_elems = (a_double ? 2 : 1) * max a_buf_elems

    const bf16_t* src_img = a.src + ((int64_t)n * a.H * a.W_src) * a.ld_src;

    // Per-tap (halo row, halo column) offsets live in LDS.  Indexing the kernel-argument tap table with the runtime
    // tap index makes hipcc emit a VECTOR load (global_load_sbyte) inside the K loop, and the s_waitcnt vmcnt(0) for
    // its result drains every prefetch load issued just before it (vmcnt counts in order) -- in round 1 that
    // serialised all global loads with the MFMAs.  An LDS lookup waits on lgkmcnt only.
    int* tap_tab = (int*)(smem + a.lds_tab_offset);
    if (tid < T) tap_tab[tid] = ((a.tt.dh[ph][tid] - a.tt.dh_min) << 16) | (a.tt.dw[ph][tid] - a.tt.dw_min[ph]);


    // ---- staging map -----------------------------------------------------------------------
    const int s_px = tid & 7, s_g = (tid >> 3) & (G - 1), s_hi = tid >> (3 + (KS == 2 ? 3 : 2));
    const int px_per_iter = 256 / G;  // 32 (KS=2) or 64 (KS=1)
    const int n_px = HR * Wt;
    int src_off[kNa2];
    int lds_off[kNa2];
#pragma unroll
    for (int i = 0; i < kNa2; ++i) {
        const int p = i * px_per_iter + s_hi * 8 + s_px;
        src_off[i] = INT32_MIN;
        lds_off[i] = -1;
        if (p < n_px) {
            const int hr = p / Wt, cc = p - hr * Wt;
            const int hs = h0 + a.tt.dh_min + hr;
            const int ws = m0 + a.tt.dw_min[ph] + cc;
            lds_off[i] = ((hr * G + s_g) * Wtp + cc) * 8;
            if (hs >= 0 && hs < a.H && ws >= 0 && ws < a.W_src) src_off[i] = (hs * a.W_src + ws) * a.ld_src + s_g * 8;
        }
    }
    // B: 128 n x G groups = 128*G chunks; thread handles n = (q>>... ) with the same 8x8 wave map
    constexpr int kNb = (kBN * G) / 256;  // 4 (KS=2) or 2 (KS=1)
    const int64_t w_img = (int64_t)a.C_dst * a.C_src;
    const int w_base_idx = a.tt.w_first[ph];

    u32x4 a_reg[kNa2];
    u32x4 b_reg[kNb];
    const bool affine = a.flags & RV_IN_AFFINE, relu = a.flags & RV_IN_RELU;

    auto load_a = [&](int kc) {
#pragma unroll
        for (int i = 0; i < kNa2; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (src_off[i] != INT32_MIN) v = *(const u32x4*)(src_img + src_off[i] + kc * BK);
            a_reg[i] = v;
        }
    };
    auto write_a = [&](int abuf, int kc) {
        bf16_t* base = ldsA + abuf * a_buf_elems;
        float sc[8], sh[8];
        if (affine) {  // loaded here, after the MFMA section: a wait for these would drain the prefetch loads too
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sc[j] = a.in_scale[kc * BK + s_g * 8 + j];
                sh[j] = a.in_shift[kc * BK + s_g * 8 + j];
            }
        }
#pragma unroll
        for (int i = 0; i < kNa2; ++i) {
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
            *(u32x4*)(base + lds_off[i]) = v;
        }
    };
    auto load_b = [&](int kc, int t, int buf_next) {
        const bf16_t* wt = a.w + (int64_t)(w_base_idx + t) * w_img + kc * BK;
        if constexpr (BDMA) {
            // LDS-DMA (global_load_lds_dwordx4): one wave-instruction fills 64 consecutive 16-byte chunks = one
            // k-group x 64 output channels of the dense B image; no VGPR round trip, no ds_write.
            constexpr int kInstr = (G * kBN) / 64 / 4;  // instructions per wave (4 or 2)
#pragma unroll
            for (int i = 0; i < kInstr; ++i) {
                const int idx = wave * kInstr + i;
                const int gg = idx >> 1, nh = idx & 1;
                const int nn = nh * 64 + lane;
                bf16_t* dst = ldsB + buf_next * (G * kBN * 8) + (gg * kBN + nh * 64) * 8;  // wave-uniform base
                if (n0 + nn < a.C_dst)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(wt + (int64_t)(n0 + nn) * a.C_src + gg * 8),
                                                     (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
            }
        } else {
#pragma unroll
            for (int i = 0; i < kNb; ++i) {
                const int nn = i * px_per_iter + s_hi * 8 + s_px;  // 0..127
                u32x4 v = {0u, 0u, 0u, 0u};
                if (n0 + nn < a.C_dst) v = *(const u32x4*)(wt + s_g * 8 + (int64_t)(n0 + nn) * a.C_src);
                b_reg[i] = v;
            }
        }
    };
    auto write_b = [&](int buf) {
        if constexpr (BDMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's DMA pieces have landed; the barrier publishes them
        } else {
            bf16_t* base = ldsB + buf * (G * kBN * 8);
#pragma unroll
            for (int i = 0; i < kNb; ++i) {
                const int nn = i * px_per_iter + s_hi * 8 + s_px;
                pin_here(b_reg[i]);
                *(u32x4*)(base + (s_g * kBN + nn) * 8) = b_reg[i];
            }
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (T > 0) ? a.C_src / BK : 0;
    if (nk > 0) {
        load_a(0);
        load_b(0, 0, 0);
        write_a(0, 0);
        write_b(0);
    }
    __syncthreads();
    int buf = 0, abuf = 0;
    for (int kc = 0; kc < nk; ++kc) {
        const bool next_chunk = kc + 1 < nk;
        load_a(next_chunk ? kc + 1 : kc);  // unconditional (see tapconv3.hip): no register merge => no early vmcnt(0)
        for (int t = 0; t < T; ++t) {
            const bool last_tap = (t == T - 1);
            const bool has_next = !last_tap || next_chunk;
            load_b(has_next ? (last_tap ? kc + 1 : kc) : kc, has_next ? (last_tap ? 0 : t + 1) : t, buf ^ 1);
            {
                const int ti = tap_tab[t];
                const int hr = wm + (ti >> 16), c0 = ti & 0xffff;
                const bf16_t* pa = ldsA + abuf * a_buf_elems + ((hr * G + lg) * Wtp + c0 + l15) * 8;
                const bf16_t* pb = ldsB + buf * (G * kBN * 8) + (lg * kBN + wn * 64 + l15) * 8;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    // all eight fragment reads of the K-step are issued before its 16 MFMAs: one exposed LDS latency per
                    // K-step (covered by the partner wave on the same SIMD) instead of one per group of four MFMAs
                    bf16x8 fb[4], fa[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[j] = *(const bf16x8*)(pb + (ks * 4 * kBN + j * 16) * 8);
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[i] = *(const bf16x8*)(pa + (ks * 4 * Wtp + i * 16) * 8);
                    // all 8 reads are issued first (sched_barrier pins that); hipcc's own counted lgkmcnt waits then let the
                    // first MFMA group start as soon as its operands (5 of the 8 reads) have landed
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = RV_MFMA_16x16x32(fa[i], fb[j], acc[i][j], 0, 0, 0);
                    __builtin_amdgcn_s_setprio(0);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            write_b(buf ^ 1);
            if (last_tap && next_chunk) {
                if (a_double) {
                    write_a(abuf ^ 1, kc + 1);
                    abuf ^= 1;
                } else {
                    __syncthreads();
                    write_a(0, kc + 1);
                }
            }
            __syncthreads();
            buf ^= 1;
        }
    }

    // ------------------------------------ epilogue --------------------------------------------
    const int Wm = a.W_dst / a.phases;
    const int h = h0 + wm;
    const bool row_ok = h < a.H;
    // this lane's pixels: m = m0 + i*16 + lg*4 + r (row h), channel n0 + wn*64 + j*16 + l15
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + i * 16 + lg * 4 + r;
            if (m >= Wm || !row_ok) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j][r] = 0.f;
            }
        }
    if (a.flags & RV_OUT_STATS) {
        float* prow = a.stats + ((int64_t)(blockIdx.x * 2 + wm) * 2) * a.C_dst;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
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
            const int c = n0 + wn * 64 + j * 16 + l15;
            if (lg == 0 && c < a.C_dst) {
                prow[c] = s;
                prow[a.C_dst + c] = q;
            }
        }
    }
    if (a.flags & RV_OUT_BIAS) {
        const bool relu_out = (a.flags & RV_OUT_RELU) != 0;  // eval: BatchNorm folded into weights + bias, ReLU on the way out
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = n0 + wn * 64 + j * 16 + l15;
            const float b = (c < a.C_dst) ? a.bias[c] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = relu_out ? fmaxf(acc[i][j][r] + b, 0.f) : acc[i][j][r] + b;
        }
    }
    if (a.flags & RV_OUT_F32) {
        if (!row_ok) return;
        float* dst = (float*)a.dst + ((int64_t)(n * a.H + h) * a.W_dst) * a.ld_dst;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + i * 16 + lg * 4 + r;
                if (m >= Wm) continue;
                const int64_t px = (int64_t)(a.phases * m + ph) * a.ld_dst;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c = n0 + wn * 64 + j * 16 + l15;
                    if (c < a.C_dst) dst[px + c] = acc[i][j][r];
                }
            }
        return;
    }
    constexpr int kEpi = kBN + 8;
    bf16_t* epi = (bf16_t*)smem;  // [2 rows * 64 cols][kEpi]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pm = wm * kTC + i * 16 + lg * 4 + r;
                const int pc = wn * 64 + j * 16 + l15;
                epi[pm * kEpi + pc] = f2bf(acc[i][j][r]);
            }
    __syncthreads();
    constexpr int kChunks = kBN / 8;
    const bool accum = a.flags & RV_OUT_ACCUM;
    for (int q = tid; q < kTR * kTC * kChunks; q += 256) {
        const int pm = q / kChunks, c8 = q - pm * kChunks;
        const int rr = pm / kTC, mm = pm - rr * kTC;
        const int m = m0 + mm, c = n0 + c8 * 8, hh = h0 + rr;
        if (m >= Wm || c >= a.C_dst || hh >= a.H) continue;
        u32x4 v = *(const u32x4*)(epi + pm * kEpi + c8 * 8);
        bf16_t* p = (bf16_t*)a.dst + (((int64_t)(n * a.H + hh) * a.W_dst) + (a.phases * m + ph)) * a.ld_dst + c;
        if (accum) {
            const u32x4 o = *(const u32x4*)(a.res + (((int64_t)(n * a.H + hh) * a.W_dst) + (a.phases * m + ph)) * a.ld_res + c);
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

}  // namespace

// returns false when the layer is not eligible (caller falls back to tapconv_kernel)
bool rv_tapconv2_plan(TapConvArgs* a, int* grid_x, int* grid_y, size_t* lds, int* ks) {
    if (a->step != 1) return false;
    const int wm_total = a->W_dst / a->phases;
    if (wm_total < kTC || a->C_dst < 64 || a->H < 2) return false;
    int a_max = 0;
    bool any_single_tap = true;
    const int KS = (a->C_src % 64 == 0) ? 2 : 1;
    const int G = 4 * KS;
    const int HR = kTR + a->tt.rows - 1;
    for (int r = 0; r < a->phases; ++r) {
        a->tt.w_tile[r] = (kTC - 1) + (a->tt.dw_max[r] - a->tt.dw_min[r]) + 1;
        const int wtp = (a->tt.w_tile[r] + 15) & ~15;
        const int e = HR * G * wtp * 8;
        a_max = a_max > e ? a_max : e;
        if (HR * a->tt.w_tile[r] > kNa2 * (256 / G)) return false;
        if (a->tt.ntaps[r] != 1) any_single_tap = false;
    }
    // the A tile is double-buffered by phases with a single tap; size the region for it when any phase needs it
    bool some_single = false;
    for (int r = 0; r < a->phases; ++r) some_single |= (a->tt.ntaps[r] == 1);
    (void)any_single_tap;
    a->lds_a_elems = a_max * (some_single ? 2 : 1);
    size_t bytes = (size_t)(a->lds_a_elems + 2 * G * kBN * 8) * sizeof(bf16_t);
    const size_t epi = (size_t)kTR * kTC * (kBN + 8) * sizeof(bf16_t);
    if (bytes < epi) bytes = epi;
    bytes = (bytes + 15) & ~(size_t)15;
    a->lds_tab_offset = (int32_t)bytes;
    bytes += 128;  // per-tap offset table (<= 24 ints)
    if (bytes > 80 * 1024) return false;  // keep two workgroups per CU
    a->m_tiles = rv_ceil_div(wm_total, kTC);
    a->h_tiles = rv_ceil_div(a->H, kTR);
    *grid_x = a->m_tiles * a->h_tiles * a->N * a->phases;
    *grid_y = rv_ceil_div(a->C_dst, kBN);
    *lds = bytes;
    *ks = KS;
    return true;
}

int rv_tapconv2_launch(const TapConvArgs& a, int grid_x, int grid_y, size_t lds, int ks, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)tapconv2_kernel<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)tapconv2_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (ks == 2) hipLaunchKernelGGL((tapconv2_kernel<2, true>), dim3(grid_x, grid_y), dim3(256), lds, stream, a);  // (weights by LDS-DMA)
    else hipLaunchKernelGGL((tapconv2_kernel<1, true>), dim3(grid_x, grid_y), dim3(256), lds, stream, a);
    RV_CHECK_LAUNCH("tapconv2_kernel");
    return 0;
}

// metachain.hip -- backward of the MetaKernel modulation CHAINED with the backward-data GEMM of the fusion conv in front of it
// and the BatchNorm(+ReLU) backward of the positional layer behind it (nn/stems/__init__.py:64-85; round-4 review item 5).
//
//   forward:   P = relu(bn(y)),  geo[p][t][c] = P[p][t][c] * feat[p + off(t)][c],  z1 = Wf geo   (1x1 conv, 9C -> K channels)
//   backward:  dgeo[p][t][c] = sum_k dz[p][k] Wf[k][t][c]                 <- conv2d backward-data: a (pixels x K) x (K x 9C) GEMM
//              g[p][t][c]    = dgeo * feat[p + off(t)] * [P > 0]          gradient w.r.t. the BatchNorm output
//              dfeat[q][c]   = sum_t dgeo[q - off(t)][t][c] * P[q - off(t)][t][c]
//              dy[p][t][c]   = k0 (g - S0/n - xhat S1/n),  S0 = sum g, S1 = sum g * xhat
// Until round 5 dgeo was a tensor: written by the GEMM (2.4 GB at 4 x 64 x 2048 x 9 x 256), read by the sums pass and again by
// the apply pass (csrc/meta.hip).  Here it lives in MFMA accumulators only: BOTH passes recompute it (0.62 TFLOP each) and consume
// it in the tile epilogue, one tap at a time --
//   rv_meta_chain_bwd_sums   TARGET order: the tile is 64 target pixels q of one image row x all C channels; for tap t the GEMM's
//       rows are dz at the SOURCE pixels q - off(t) (a shifted view, out-of-image rows from a zero page), so the nine tap results of
//       one register slot all belong to the same dfeat[q][c] -- a second accumulator set takes sum_t D_t * P -- and to the target's
//       own feat[q][c] (no neighbour gather); every (p, t) whose neighbour lies in the image is the t-th source of exactly one q.
//   rv_meta_chain_bwd_apply  SOURCE order: rows are dz at the tile's own pixels for all nine taps, feat comes from p + off(t);
//       writes dy, the only 9x-grid tensor this chain writes.
// HBM per step: y twice + dy once (7.2 GB) where the unchained passes moved 14.5 GB; dgeo is never rounded to bf16 either.
//
// MFMA as D^T = Wf_t dz^T (v_mfma_f32_16x16x32: M = 16 channels, N = 16 pixels), rows permuted as in headfinal.hip so that a lane
// holds pixel l15 and SIXTEEN consecutive channels of it over the wave's four channel tiles: y / feat / dy / dfeat move as 32
// contiguous bytes per lane and pixel, 128 per wave, a full 2C-byte row per workgroup -- no LDS on the way out.
// Workgroup = C / 64 waves (wave = 64 channels x 64 pixels: 64 accumulator registers per set), TWO workgroups per CU: one streams y
// in its epilogue while the other issues MFMAs.  Operands reach LDS by LDS-DMA in 32-k stages (weights [C][32 k] + pixels [64][32 k],
// 64-byte rows, tapconv6's one-bit swizzle on the per-lane source address), ring of three stages, one barrier and one counted
// wait per stage.  Workgroups are persistent over a range of tiles walked in 16-row column strips (the dz rows of h +- 1 are the
// next tile's own rows: they stay in the XCD's L2); the BatchNorm sums stay in registers over the whole range: one partial row per
// workgroup (<= 2 x CUs rows: rv_bn_bwd_finalize's one-launch form).
#include "common.h"

#ifndef RV_MC_ABL
#define RV_MC_ABL 0  // ablation builds (profiles/tools/mb_metachain.py): 1 no tap epilogue (nor its loads), 2 no fragment reads / MFMAs, 4 no operand DMA
#endif

namespace {

constexpr int kPx = 64;    // pixels per tile (one image row segment)
constexpr int kBK = 32;    // K per stage = one MFMA K step

typedef __attribute__((address_space(3))) void lds_void_t;

// sum over the 16 lanes of a DPP row (every lane of the row ends with the total)
__device__ __forceinline__ float row_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
    return v;
}

struct MetaChainArgs {
    const bf16_t* dz;    // gradient w.r.t. the fusion conv's output, [pixels][ld_dz], K channels
    const bf16_t* w;     // the fusion conv's packed scatter image [9 * C][K] (row t * C + c, k contiguous)
    const bf16_t* y;     // raw output of the last positional conv, [pixels * 9][C]
    const bf16_t* feat;  // projection output, [pixels][ld_feat]
    const float *scale, *shift, *mean, *invstd;  // folded BatchNorm of the positional layer and its batch statistics
    const float* coef;   // apply: [3][C] from rv_bn_bwd_finalize
    bf16_t* dfeat;       // sums: [pixels][ld_dfeat]
    float* partial;      // sums: [workgroups][2][C]
    bf16_t* dy;          // apply: [pixels * 9][C]
    int32_t N, H, W, K, ld_dz, ld_feat, ld_dfeat;
    int32_t col_blocks, strip_rows, tiles, tiles_per_wg;
};

// raw buffer resource over a tensor (stride 0, 32-bit byte offsets: one VGPR per address instead of a 64-bit pointer, and the uniform
// part of an address in an SGPR); the tensors of this path are below 4 GB (checked by the host)
typedef __attribute__((ext_vector_type(4))) uint32_t rv_u32x4;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
__device__ __forceinline__ u32x4 buf_load16(__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
    return __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// opaque copy of a small integer: LDS reads indexed through it are re-issued where they are written instead of being hoisted and
// kept in registers over the whole epilogue (the per-channel constants: 32..80 registers this kernel does not have)
__device__ __forceinline__ int opaque(int v) {
    asm volatile("" : "+v"(v));
    return v;
}

template <bool APPLY, int NW>
__global__ __launch_bounds__(NW * 64, 2) void meta_chain_bwd_kernel(const MetaChainArgs a) {
    constexpr int C = NW * 64;
    constexpr int kRing = APPLY ? 3 : 2;       // stages in LDS (the sums pass keeps the target tile's feat rows there instead of a third stage)
    constexpr int kSlot = C * 64 + kPx * 64;   // one stage: weights [C][32 k] then pixels [64][32 k]
    constexpr int kConst = kRing * kSlot;      // float table [5][C] behind the ring: scale, shift, (apply) k0, ca, cb
    constexpr int kSums = kConst + 5 * C * 4;  // sums pass: float [2][C], the workgroup's (sum g, sum g * y) -- see the tap epilogue
    constexpr int kFeat = kSums + 2 * C * 4;   // sums pass: the tile's feat rows [64 pixels][C] (16-byte chunk ^ (pixel & 15))
    constexpr int nA = C / 16, nB = kPx / 16;  // DMA instructions (16 rows x 64 B each) per stage
    constexpr int nAw = nA / NW, nBw = nB / NW, nD = nAw + nBw;  // ... per wave
    constexpr int nFw = kPx * C * 2 / 1024 / NW;                 // feat tile: 1 KB instructions per wave (8)
    constexpr int kSpr = C / 8;                                  // 16-byte chunks per feat row
    static_assert(nB % NW == 0 && nA % NW == 0, "DMA rows split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, lg = lane >> 4;
    const int c0 = wave * 64 + lg * 8;  // this lane's sixteen channels: c0 .. c0 + 7 (accumulator index q = 0..7) and c0 + 32 .. c0 + 39 (q = 8..15)
    const int H = a.H, W = a.W, K = a.K;
    const int KS = K / kBK;

    float* ctab = (float*)(smem + kConst);
    for (int c = tid; c < C; c += NW * 64) {
        ctab[c] = a.scale[c];
        ctab[C + c] = a.shift[c];
        if (APPLY) {
            const float k0 = a.coef[c], c1 = a.coef[C + c], c2 = a.coef[2 * C + c], is = a.invstd[c], mu = a.mean[c];
            ctab[2 * C + c] = k0;
            ctab[3 * C + c] = k0 * (c2 * mu * is - c1);  // dy = k0 g + (ca + cb y)  [= k0 (g - c1 - xhat c2)]
            ctab[4 * C + c] = -k0 * c2 * is;
        } else {
            ((float*)(smem + kSums))[c] = 0.f;
            ((float*)(smem + kSums))[C + c] = 0.f;
        }
    }

    // ---- DMA maps: lane = (row-in-16 r16 = lane >> 2, 16-byte slot = lane & 3); the slot holds logical chunk slot ^ 2 * ((row >> 2) & 1) ----
    const int r16 = lane >> 2;
    const int chunk = (lane & 3) ^ (2 * ((lane >> 4) & 1));
    // weights: LDS rows 16 d .. 16 d + 15 of instruction d = wave + NW * u hold channel tile j = d & 3 of wave block d >> 2, MFMA row m = r16
    // <-> channel 64 (d >> 2) + 32 (j >> 1) + 8 (r16 >> 2) + 4 (j & 1) + (r16 & 3): a lane's accumulator rows 4 lg + r of tiles j = 0..3 are
    // then channels 8 lg + (0..7) and 32 + 8 lg + (0..7) of the wave's 64 -- the four lanes of a pixel read / write 64 CONTIGUOUS bytes
    // per instruction.  The d part is uniform (scalar offset), the lane keeps ONE byte offset.
    const uint32_t a_lane = (__umul24((uint32_t)(8 * (r16 >> 2) + (r16 & 3)), (uint32_t)K) + chunk * 8u) * 2u;
    const uint32_t px_total = (uint32_t)a.N * H * W;
    const __amdgpu_buffer_rsrc_t r_w = make_rsrc(a.w, 9u * C * K * 2u), r_dz = make_rsrc(a.dz, px_total * a.ld_dz * 2u),
                                 r_y = make_rsrc(a.y, px_total * 9u * C * 2u), r_f = make_rsrc(a.feat, px_total * a.ld_feat * 2u);
    const int swz = (lg ^ (2 * ((l15 >> 2) & 1))) * 16;
    const int a_rd = (wave * 64 + l15) * 64 + swz;  // + j * 1024
    const int b_rd = C * 64 + l15 * 64 + swz;       // + i * 1024
    __syncthreads();

    const int t_begin = blockIdx.x * a.tiles_per_wg;
    const int t_end = t_begin + a.tiles_per_wg < a.tiles ? t_begin + a.tiles_per_wg : a.tiles;
    for (int tile = t_begin; tile < t_end; ++tile) {
        // strip order: row within the strip fastest, then column block, then strip, then sweep
        int ti = tile;
        const int rin = ti % a.strip_rows;
        ti /= a.strip_rows;
        const int cb = ti % a.col_blocks;
        ti /= a.col_blocks;
        const int strips = H / a.strip_rows;
        const int h = (ti % strips) * a.strip_rows + rin;
        const int n = ti / strips;
        const int x0 = cb * kPx;
        const int64_t row_px = ((int64_t)n * H + h) * W;  // pixel index of (n, h, 0)
        const uint32_t row_px32 = (uint32_t)row_px;

        // the pixel rows of the DMA: column x0 + 16 e + r16 of the tile, e = wave + NW * v
        int bx[nBw];
#pragma unroll
        for (int v = 0; v < nBw; ++v) bx[v] = x0 + 16 * (wave + NW * v) + r16;

        auto issue = [&](int t, int ks, int slot_byte) {
            const int dh = t / 3 - 1, dw = t - (t / 3) * 3 - 1;
            const uint32_t w_soff = ((uint32_t)t * C * K + ks * kBK) * 2u;
#pragma unroll
            for (int u = 0; u < nAw; ++u) {
                const int d = wave + NW * u;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (lds_void_t*)(smem + slot_byte + d * 1024), 16, a_lane, w_soff + (uint32_t)(64 * (d >> 2) + 32 * ((d & 3) >> 1) + 4 * (d & 1)) * K * 2u, 0, 0);
            }
#pragma unroll
            for (int v = 0; v < nBw; ++v) {
                // sums: dz at the SOURCE pixel (h - dh, x - dw) of the target x; apply: at the tile's own pixel.  Rows outside the image
                // read pixel 0 instead: their products are masked in the epilogue (no reliance on what an out-of-range DMA lane writes)
                const int hs = APPLY ? h : h - dh, xs = APPLY ? bx[v] : bx[v] - dw;
                const bool ok = hs >= 0 && hs < H && xs >= 0 && xs < W && bx[v] < W;
                const uint32_t voff = ok ? (__umul24(row_px32 + (uint32_t)((hs - h) * W + xs), (uint32_t)a.ld_dz) + chunk * 8u) * 2u : (uint32_t)(chunk * 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_dz, (lds_void_t*)(smem + slot_byte + C * 64 + (wave + NW * v) * 1024), 16, voff, ks * kBK * 2, 0, 0);
            }
        };

        f32x4 acc[4][4], dfa[APPLY ? 1 : 4][APPLY ? 1 : 4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (!APPLY) dfa[APPLY ? 0 : j][APPLY ? 0 : i] = f32x4{0.f, 0.f, 0.f, 0.f};
            }

        if (!APPLY) {
            // the target tile's own feat rows, once per tile: instruction f covers 1 KB = 1024 / (2 C) pixel rows, lane-linear; the slot
            // of lane L in its row holds the logical chunk slot ^ (pixel & 15) (conflict-free ds_read_b128 of the lane layout below)
            const int ln = opaque(lane);  // (recomputed per tile: hoisted out of the tile loop these eight offsets were eight spilled register pairs)
#pragma unroll
            for (int f = 0; f < nFw; ++f) {
                const int inst = wave + NW * f;
                const int px = inst * (1024 / (2 * C)) + ln / kSpr, slot = ln % kSpr;
                const int x = x0 + px;
                const uint32_t voff = x < W ? (__umul24(row_px32 + x, (uint32_t)a.ld_feat) + (slot ^ (px & 15)) * 8u) * 2u : 0u;  // (columns past W: masked below)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(r_f, (lds_void_t*)(smem + kFeat + inst * 1024), 16, voff, 0, 0, 0);
            }
        }

        const int S = 9 * KS;
        // prefetch distance kRing - 1 stages
        if (!(RV_MC_ABL & 4)) {
            issue(0, 0, 0);
            if (kRing == 3) issue(0, 1, kSlot);
        }
        int t = 0, ks = 0;                 // the stage being computed
        int t2 = 0, ks2 = kRing - 1;       // the stage being issued
        int slot = 0, slot2 = (kRing - 1) * kSlot;
        u32x4 yv[4][2], fv[APPLY ? 4 : 1][2];  // y (and, apply, feat) of the tap: loaded two stages before its epilogue
        uint32_t okm = 0;                      // bit i: the pair of pixel fragment i exists (sums) / bit 4 + i: its neighbour does (apply)
        for (int s = 0; s < S; ++s) {
            // my pieces of stage s have landed (younger loads may stay in flight: the kRing - 2 stages issued since, and the epilogue
            // operands issued in the stage before this tap's last); after the barrier everybody's have, and everybody is done reading
            // stage s - 1, whose slot the stage issued below takes.  Stores among the younger operations only make this wait longer.
            if (kRing == 3 && s + 1 < S) {
                // (issue order around an epilogue: DMA(s+1) | DMA(s+2), 16 epilogue loads | DMA(s+3), 8 stores | ...: the counter retires in
                //  issue order, loads, stores and LDS-DMA alike, so everything younger than the stage waited for may stay in flight --
                //  the stores of the previous tap in particular: draining them here costs a write round trip per tap)
                if (ks == KS - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(nD + 16) : "memory");
                else if (ks == 0 && s > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(nD + 24) : "memory");
                else if (ks == 1 && s > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(nD + 8) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(nD) : "memory");
            } else if (kRing == 2 && ks == KS - 1) {
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (s + kRing - 1 < S && !(RV_MC_ABL & 4)) issue(t2, ks2, slot2);
            if (ks == KS - 2 && !(RV_MC_ABL & 1)) {
                // epilogue operands of this tap: in flight over the last two stages
                const int dh = t / 3 - 1, dw = t - (t / 3) * 3 - 1;
                okm = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int x = x0 + 16 * i + l15;
                    // y of the (source pixel, tap) pair; apply: feat of the pair's neighbour
                    const int hy = APPLY ? h : h - dh, xy = APPLY ? x : x - dw;
                    const bool oky = x < W && hy >= 0 && hy < H && xy >= 0 && xy < W;
                    okm |= (oky ? 1u : 0u) << i;
                    const uint32_t yoff = oky ? ((__umul24(row_px32 + (uint32_t)((hy - h) * W + xy), 9u) + t) * C + c0) * 2u : 0u;
                    yv[i][0] = buf_load16(r_y, yoff, 0);
                    yv[i][1] = buf_load16(r_y, yoff, 64);
                    if (APPLY) {
                        const int hf = h + dh, xf = x + dw;
                        const bool okf = x < W && hf >= 0 && hf < H && xf >= 0 && xf < W;
                        okm |= (okf ? 1u : 0u) << (4 + i);
                        const uint32_t foff = okf ? (__umul24(row_px32 + (uint32_t)(dh * W + xf), (uint32_t)a.ld_feat) + c0) * 2u : 0u;
                        fv[APPLY ? i : 0][0] = buf_load16(r_f, foff, 0);
                        fv[APPLY ? i : 0][1] = buf_load16(r_f, foff, 64);
                    }
                }
            }
            bf16x8 fa[4], fb[4];
            if (!(RV_MC_ABL & 2)) {
#pragma unroll
            for (int j = 0; j < 4; ++j) fa[j] = *(const bf16x8*)(smem + slot + j * 1024 + a_rd);
#pragma unroll
            for (int i = 0; i < 4; ++i) fb[i] = *(const bf16x8*)(smem + slot + i * 1024 + b_rd);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = RV_MFMA_16x16x32(fa[j], fb[i], acc[j][i], 0, 0, 0);
            }
            slot = slot + kSlot == kRing * kSlot ? 0 : slot + kSlot;
            slot2 = slot2 + kSlot == kRing * kSlot ? 0 : slot2 + kSlot;
            if (++ks2 == KS) {
                ks2 = 0;
                ++t2;
            }
            if (++ks < KS) continue;
            ks = 0;
            if (RV_MC_ABL & 1) {
                ++t;
                continue;
            }
            // ---- tap epilogue: acc[j][i][r] = dgeo of pixel column x0 + 16 i + l15, channel c0 + 4 j + r, tap t ----------------
            // (pairs outside the image are masked: their accumulator rows and y / feat values come from pixel 0)
            if constexpr (APPLY) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int o = opaque(c0);
                    u32x4 ov[2];
#pragma unroll
                    for (int qq = 0; qq < 16; qq += 2) {
                        const int q = (qq & 7) + 32 * (qq >> 3);  // channel offset of accumulator index qq
                        const uint32_t yw = yv[i][qq >> 3][(qq >> 1) & 3], fw = fv[i][qq >> 3][(qq >> 1) & 3];
                        const float y0 = bf_lo(yw), y1 = bf_hi(yw), f0 = bf_lo(fw), f1 = bf_hi(fw);
                        const float t0 = y0 * ctab[o + q] + ctab[C + o + q], t1 = y1 * ctab[o + q + 1] + ctab[C + o + q + 1];
                        float g0 = acc[qq >> 2][i][qq & 3], g1 = acc[(qq + 1) >> 2][i][(qq + 1) & 3];
                        const bool nb = (okm >> (4 + i)) & 1u;  // the neighbour exists (outside the image the modulation was with zero)
                        g0 = (nb && t0 > 0.f) ? g0 : 0.f;  // dgeo where the ReLU passed
                        g1 = (nb && t1 > 0.f) ? g1 : 0.f;
                        const float d0 = ctab[2 * C + o + q] * (g0 * f0) + (ctab[4 * C + o + q] * y0 + ctab[3 * C + o + q]);
                        const float d1 = ctab[2 * C + o + q + 1] * (g1 * f1) + (ctab[4 * C + o + q + 1] * y1 + ctab[3 * C + o + q + 1]);
                        ov[qq >> 3][(qq >> 1) & 3] = pack_bf2(d0, d1);
                    }
                    const int x = x0 + 16 * i + l15;
                    if (x < W) {
                        bf16_t* po = a.dy + ((row_px + x) * 9 + t) * C + c0;
                        *(u32x4*)po = ov[0];
                        *(u32x4*)(po + 32) = ov[1];
                    }
                }
            } else {
                // channel pair outermost, the four pixel fragments inside: the pair's constants, its (sum g, sum g * y) and one feat word
                // at a time are all that lives beside the two accumulator sets and y -- the sums leave the registers pair by pair
                // (row sum over the sixteen lanes that hold the same channels by DPP; lane l15 == 0 adds to the workgroup's table:
                // its channels are nobody else's, plain LDS adds in a fixed order -> reproducible)
#pragma unroll
                for (int q = 0; q < 16; q += 2) {
                    const int o = opaque(c0 + (q & 7) + 32 * (q >> 3));  // the pair's first channel
                    const float sc0 = ctab[o], sc1 = ctab[o + 1], sh0 = ctab[C + o], sh1 = ctab[C + o + 1];
                    float s00 = 0.f, s01 = 0.f, s10 = 0.f, s11 = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const uint32_t yw = yv[i][q >> 3][(q >> 1) & 3];
                        const uint32_t fw = *(const uint32_t*)(smem + kFeat + (16 * i + l15) * 2 * C + ((o >> 3) ^ l15) * 16 + ((q >> 1) & 3) * 4);
                        const float y0 = bf_lo(yw), y1 = bf_hi(yw), f0 = bf_lo(fw), f1 = bf_hi(fw);
                        const float t0 = y0 * sc0 + sh0, t1 = y1 * sc1 + sh1;
                        float g0 = acc[q >> 2][i][q & 3], g1 = acc[(q + 1) >> 2][i][(q + 1) & 3];
                        const bool pair = (okm >> i) & 1u;  // the (source pixel, tap) pair exists
                        g0 = (pair && t0 > 0.f) ? g0 : 0.f;  // dgeo where the ReLU passed
                        g1 = (pair && t1 > 0.f) ? g1 : 0.f;
                        dfa[APPLY ? 0 : q >> 2][APPLY ? 0 : i][q & 3] += g0 * t0;
                        dfa[APPLY ? 0 : (q + 1) >> 2][APPLY ? 0 : i][(q + 1) & 3] += g1 * t1;
                        const float z0 = g0 * f0, z1 = g1 * f1;
                        s00 += z0;
                        s01 += z1;
                        s10 += z0 * y0;
                        s11 += z1 * y1;
                    }
                    s00 = row_sum16(s00);
                    s01 = row_sum16(s01);
                    s10 = row_sum16(s10);
                    s11 = row_sum16(s11);
                    if (l15 == 0) {
                        float* st = (float*)(smem + kSums) + o;
                        st[0] += s00;
                        st[1] += s01;
                        st[C] += s10;
                        st[C + 1] += s11;
                    }
                    // (one pair at a time: left alone, the compiler sinks all 64 accumulator updates below the last pair and keeps their
                    //  64 operands alive until then)
#pragma unroll
                    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(dfa[APPLY ? 0 : q >> 2][APPLY ? 0 : i]));
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};
            ++t;
        }
        if (!APPLY) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int x = x0 + 16 * i + l15;
                if (x >= W) continue;  // (index q of the accumulators <-> channel c0 + (q & 7) + 32 (q >> 3): two 16-byte pieces 64 bytes apart)
                u32x4 ov[2];
#pragma unroll
                for (int q = 0; q < 16; q += 2)
                    ov[q >> 3][(q >> 1) & 3] = pack_bf2(dfa[APPLY ? 0 : q >> 2][APPLY ? 0 : i][q & 3], dfa[APPLY ? 0 : (q + 1) >> 2][APPLY ? 0 : i][(q + 1) & 3]);
                bf16_t* po = a.dfeat + (row_px + x) * a.ld_dfeat + c0;
                *(u32x4*)po = ov[0];
                *(u32x4*)(po + 32) = ov[1];
            }
        }
        // the ring (and the feat rows) are rewritten by the next tile's first loads: nobody may still be reading this tile's
        __builtin_amdgcn_s_barrier();
    }
    if (!APPLY) {
        __syncthreads();
        const float* st = (const float*)(smem + kSums);
        float* row = a.partial + (int64_t)blockIdx.x * 2 * C;
        for (int c = tid; c < C; c += NW * 64) {
            const float is = a.invstd[c], mu = a.mean[c];
            row[c] = st[c];
            row[C + c] = is * (st[C + c] - mu * st[c]);  // sum g * xhat from (sum g * y, sum g)
        }
    }
}

int chain_strip_rows(int H) {
    for (int r = 16; r > 1; --r)
        if (H % r == 0) return r;
    return 1;
}

// persistent grid: two workgroups per CU, every workgroup a contiguous range of tiles
void chain_plan(int32_t N, int32_t H, int32_t W, int32_t* tiles, int32_t* per_wg, int32_t* wgs) {
    const int64_t cb = (W + kPx - 1) / kPx;
    const int64_t n_tiles = (int64_t)N * H * cb;
    const int64_t want = 2 * (int64_t)rv_cu_count();
    const int64_t per = (n_tiles + want - 1) / want;
    *tiles = (int32_t)n_tiles;
    *per_wg = (int32_t)(per < 1 ? 1 : per);
    *wgs = (int32_t)((n_tiles + *per_wg - 1) / *per_wg);
}

int chain_fill(MetaChainArgs* a, const void* dz, int32_t ld_dz, int32_t K, const void* w_scatter, const void* pos_raw, const float* scale,
               const float* shift, const float* mean, const float* invstd, const void* feat, int32_t ld_feat, int32_t N, int32_t H, int32_t W,
               int32_t C) {
    RV_REQUIRE(dz && w_scatter && pos_raw && scale && shift && mean && invstd && feat, "rv_meta_chain_bwd: null argument");
    RV_REQUIRE(N > 0 && H > 0 && W > 0, "rv_meta_chain_bwd: empty image");
    RV_REQUIRE(C == 256 || C == 128, "rv_meta_chain_bwd: the stem width must be 128 or 256 channels (got %d)", C);
    RV_REQUIRE(K >= 128 && K % 32 == 0, "rv_meta_chain_bwd: the fusion conv's output width must be a multiple of 32, at least 128 (got %d)", K);
    RV_REQUIRE(ld_dz >= K && ld_dz % 8 == 0 && ld_feat >= C && ld_feat % 8 == 0, "rv_meta_chain_bwd: bad channel strides (%d, %d)", ld_dz, ld_feat);
    RV_REQUIRE((int64_t)N * H * W * 9 * C * 2 < 0xffff0000ll && (int64_t)N * H * W * ld_dz * 2 < 0xffff0000ll && (int64_t)N * H * W * ld_feat * 2 < 0xffff0000ll,
               "rv_meta_chain_bwd: tensors of 4 GB and more are not addressed by this kernel (32-bit buffer offsets)");
    memset(a, 0, sizeof(*a));
    a->dz = (const bf16_t*)dz, a->w = (const bf16_t*)w_scatter, a->y = (const bf16_t*)pos_raw, a->feat = (const bf16_t*)feat;
    a->scale = scale, a->shift = shift, a->mean = mean, a->invstd = invstd;
    a->N = N, a->H = H, a->W = W, a->K = K, a->ld_dz = ld_dz, a->ld_feat = ld_feat;
    a->col_blocks = (W + kPx - 1) / kPx;
    a->strip_rows = chain_strip_rows(H);
    int32_t wgs;
    chain_plan(N, H, W, &a->tiles, &a->tiles_per_wg, &wgs);
    return 0;
}

template <bool APPLY>
int chain_launch(const MetaChainArgs& a, int32_t C, hipStream_t st) {
    int32_t tiles, per, wgs;
    chain_plan(a.N, a.H, a.W, &tiles, &per, &wgs);
    const int lds = APPLY ? 3 * (C * 64 + kPx * 64) + 5 * C * 4 : 2 * (C * 64 + kPx * 64) + 7 * C * 4 + kPx * C * 2;
    if (C == 256) {
        static bool once = false;
        if (!once) {
            once = true;
            (void)hipFuncSetAttribute((const void*)meta_chain_bwd_kernel<APPLY, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        }
        hipLaunchKernelGGL((meta_chain_bwd_kernel<APPLY, 4>), dim3(wgs), dim3(256), lds, st, a);
    } else {
        hipLaunchKernelGGL((meta_chain_bwd_kernel<APPLY, 2>), dim3(wgs), dim3(128), lds, st, a);
    }
    return 0;
}

}  // namespace

extern "C" {

int32_t rv_meta_chain_rows(int32_t N, int32_t H, int32_t W) {
    if (N <= 0 || H <= 0 || W <= 0) return 0;
    int32_t tiles, per, wgs;
    chain_plan(N, H, W, &tiles, &per, &wgs);
    return wgs;
}

int rv_meta_chain_bwd_sums(const void* dz, int32_t ld_dz, int32_t K, const void* w_scatter, const void* pos_raw, const float* scale,
                           const float* shift, const float* mean, const float* invstd, const void* feat, int32_t ld_feat, int32_t N, int32_t H,
                           int32_t W, int32_t C, void* dfeat, int32_t ld_dfeat, float* partial, rvStream stream) {
    MetaChainArgs a;
    if (chain_fill(&a, dz, ld_dz, K, w_scatter, pos_raw, scale, shift, mean, invstd, feat, ld_feat, N, H, W, C)) return 1;
    RV_REQUIRE(dfeat && partial && ld_dfeat >= C && ld_dfeat % 8 == 0, "rv_meta_chain_bwd_sums: null argument / bad stride");
    a.dfeat = (bf16_t*)dfeat, a.ld_dfeat = ld_dfeat, a.partial = partial;
    chain_launch<false>(a, C, (hipStream_t)stream);
    RV_CHECK_LAUNCH("meta_chain_bwd_kernel<sums>");
    return 0;
}

int rv_meta_chain_bwd_apply(const void* dz, int32_t ld_dz, int32_t K, const void* w_scatter, const void* pos_raw, const float* scale,
                            const float* shift, const float* mean, const float* invstd, const float* coef, const void* feat, int32_t ld_feat,
                            int32_t N, int32_t H, int32_t W, int32_t C, void* dy, rvStream stream) {
    MetaChainArgs a;
    if (chain_fill(&a, dz, ld_dz, K, w_scatter, pos_raw, scale, shift, mean, invstd, feat, ld_feat, N, H, W, C)) return 1;
    RV_REQUIRE(coef && dy, "rv_meta_chain_bwd_apply: null argument");
    a.coef = coef, a.dy = (bf16_t*)dy;
    chain_launch<true>(a, C, (hipStream_t)stream);
    RV_CHECK_LAUNCH("meta_chain_bwd_kernel<apply>");
    return 0;
}

}  // extern "C"

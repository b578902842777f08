// tapconv6.hip -- sixth-generation tap-conv kernel: 512 pixels x 128 channels per workgroup, 32-channel chunks.
//
// Why (round-3 review): tapconv5's 256 x 256 tile pulls 45 KB of halo + 9 x 32 KB of weights from L2 per 64-channel
// chunk of a 3x3 layer -- 333 KB per 75.5 MFLOP -- and that fill stream, not the MFMA schedule, is what the loop waits for
// (profiles/r02_tapconv_ablation.md).  The weights are 86 % of those bytes, so the cut is MORE PIXELS PER WEIGHT BYTE: the same
// 64 K accumulators laid out as 16 rows x 32 columns x 128 channels need 83 KB of halo + 9 x 16 KB of weights per 64 input
// channels = 227 KB for the same FLOPs (-32 %), and the 128-channel layers (rv-waymo, the DLA stages) get 128 accumulator
// registers per lane where tapconv5<128> has 64 (twice the MFMAs per weight byte and per fixed tile cost).
// An 18-row halo of 64 channels is 81 KB -- two of them do not fit 160 KB of LDS -- hence 32-channel chunks: LDS rows are 64 B
// (one MFMA K step), a DMA instruction moves 16 rows x 64 B, and the XOR swizzle is one bit (below).
//
// Tile: 16 image rows x 32 columns (M = 512; halo 18 x 34, 1.20x) x 128 channels; K tile = one tap x 32 channels.
// 8 waves as 4 (M) x 2 (N): 128 x 64 outputs per wave (128 accumulator VGPRs, the wave tile of tapconv5<256>), TWO phases of
// 16 MFMAs per K tile (pixel halves against one set of four weight fragments); the two halves of the workgroup (waves 0-3,
// 4-7: one of each per SIMD) run one barrier apart, so that one half issues loads and fragment reads while the other issues
// its MFMA cluster.  Loads, the same branch-free code on all eight waves:
//   weights: one 8 KB piece ([128 channels][32 k]) per K tile = ONE instruction per wave, into a ring of eight pieces,
//            issued kAhead K tiles before its first read;
//   halo:    the NEXT chunk's 41 instructions, six per wave (seven dummies), two per K tile during the first three K tiles
//            of the current chunk.
//   vmcnt retires in order: the K-tile body is instantiated per position in the chunk with the exact count for that position.
// Swizzle: ds_read_b128 serves 16 lanes per LDS cycle; with 64-byte rows and the MFMA fragment layout (lane = row l15, 16-byte
// chunk lg) sixteen lanes of a group cover all 64 banks iff chunk' = lg ^ 2*((row >> 2) & 1) -- applied, as in tapconv4/5, to
// the per-lane SOURCE address of the DMA (which writes LDS lane-linearly).  Row pitch 36 = 4 * 9: one image row further the
// swizzle bit flips, so ONE address computation per K tile serves all four rows a wave reads (a_b1 = a_b0 ^ 32).
//
// Eligible: stride-1 phases with 6..16 taps on a rows x columns grid whose halo fits (<= 18 rows x 34 columns), C_src % 32 == 0,
// C_dst % 128 == 0, plain bf16 input, bf16 output (+ statistics / bias / accumulate / BatchNorm-backward sums), H >= 16 and
// at least one round of workgroups.  Everything else stays on tapconv5 / tapconv4.
#include <stdlib.h>

#include "common.h"
#include "tapconv.h"

namespace {

constexpr int kTC = 32;                  // tile columns
constexpr int kTR = 16;                  // tile rows
constexpr int kBK = 32;                  // K tile (channels of one tap) = one MFMA K step
constexpr int kBN = 128;                 // channels per workgroup
constexpr int kPitch = 36;               // halo row pitch in pixel slots (>= 34 used columns); 36/4 odd: see the swizzle note
constexpr int kHaloRows = 18;
constexpr int kHaloInstr = (kHaloRows * kPitch + 15) / 16;  // 41 DMA instructions (16 pixel slots x 64 B each) per halo
constexpr int kHaloBytes = kHaloInstr * 1024;               // 41984
constexpr int kPiece = kBN * kBK * 2;                       // one weight piece: 128 rows x 32 k = 8 KB
constexpr int kNRing = 8;
#ifndef RV_T6_AHEAD
#define RV_T6_AHEAD 4
#endif
constexpr int kAhead = RV_T6_AHEAD;                         // a weight piece is issued this many K tiles before its K tile (<= 6)
constexpr int kRing = 2 * kHaloBytes;                       // weight ring behind the two halo buffers
constexpr int kScratch = kRing + kNRing * kPiece;           // 1 KB target of the dummy halo instructions
constexpr int kTab = kScratch + 1024;                       // tap table (prologue hand-off)
constexpr int kStatAcc = kTab + 32 * 4;                     // per-workgroup statistic accumulators: float [8 waves][4 j][16 channels][2]
constexpr int kBnbAcc = kStatAcc + 8 * 4 * 16 * 2 * 4;      // per-workgroup BatchNorm-backward sums: float [16 chunks][16] (thread tid < 256 owns one)
constexpr int kLds = kBnbAcc + 256 * 4;
constexpr int kMinTaps = kAhead + 2 > 6 ? kAhead + 2 : 6;  // (the K-tile bodies 0..5 are unconditional)

__device__ __attribute__((aligned(256))) uint32_t g_zero_page6[64];

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// counted wait of the K tile at position U of its chunk: the kAhead - 1 younger weight instructions plus the halo instructions
// (two per K tile at positions 0..2) issued since the piece of the NEXT K tile went out, kAhead - 1 K tiles ago; from position
// kAhead + 1 on the window holds no halo instruction (and the last K tile of a chunk, U >= kMinTaps - 1, thereby retires the
// whole halo of the next chunk)
constexpr int wait_count(int U) {
    if (U > kAhead) return kAhead - 1;
    int n = 0;
    for (int u = U - kAhead + 1; u <= U; ++u) n += (u >= 0 && u < 3) ? 2 : 0;
    return kAhead - 1 + n;
}

// EPI: 0 plain store (+ statistics / bias), 1 BatchNorm-sum epilogue (RV_OUT_BNB), 2 accumulate (RV_OUT_ACCUM) -- template parameter
// for the reason given in tapconv5.hip (spills); the masked last-writer form (both) stays on tapconv5
// (Measured and dropped in round 4, profiles/r04_tapconv6_ablation.md: a role split -- four waves issue every LDS-DMA and never store, four
//  store and never load -- 0 % / -5 % / -9 % on the 512- / 256- / 128-channel layers; a pipelined tile boundary with the next tile's
//  prologue issued before the stores, 2 ms per step slower; non-temporal epilogue stores, neutral.  Neither is in the library.)
template <int EPI>
__global__ __launch_bounds__(512, 2) void tapconv6_kernel(const TapConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, lg = lane >> 4;
    constexpr int kHaloPer = 6;                 // halo instruction slots per wave (41 real ones over 8 waves, the rest dummies)
    constexpr int kHaloPerTile = kHaloPer / 3;  // ... issued per K tile at positions 0..2 of a chunk

    // XCD-aware persistent block order as in tapconv5: the channel tiles of one pixel tile sit on neighbouring workgroups of
    // one XCD (they read the same halo through that XCD's L2)
    const int gy = a.n_tiles;
    const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, nslots = gridDim.x >> 3;  // (gridDim.x % 8 == 0)
    // RV_OUT_STATS of a persistent launch: the (sum, sum of squares) rows of a workgroup's tiles are accumulated here and written
    // once at the end -- 4 rows per group of gy workgroups instead of 4 per tile (4096 rows behind a 4 x 64 x 2048 launch: the
    // finalize then needs its two-stage column reduction; <= 1024 rows take the one-launch form).  Slots are private to one lane.
    auto stat_slot = [&]() { return (float*)(smem + kStatAcc) + (wave * 4 * 16 + l15) * 2; };  // + j * 32 (recomputed where used: no register held over the K loop)
    if (a.stats_per_wg && lg == 0) {
        float* stat_acc = stat_slot();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            stat_acc[j * 32] = 0.f;
            stat_acc[j * 32 + 1] = 0.f;
        }
    }
    if ((EPI == 1) && a.stats_per_wg && tid < 256)
        ((float*)(smem + kBnbAcc))[tid] = 0.f;  // (same rows rule for the BatchNorm-backward sums; slot owner = the thread that adds to it)
    for (int k = 0;; ++k) {
    const int xslot = wslot + nslots * k;
    if (xslot >= a.tiles_per_xcd * gy) break;
    const int tile = xcd * a.tiles_per_xcd + xslot / gy;
    if (tile >= a.total_tiles) continue;
    const int n0 = (xslot % gy) * kBN;
    int bx = tile;
    const int tc = bx % a.m_tiles;
    bx /= a.m_tiles;
    const int th = bx % a.h_tiles;
    bx /= a.h_tiles;
    const int n = bx % a.N;
    const int ph = bx / a.N;
    const int m0 = tc * kTC, h0 = th * kTR;
    const int T = a.tt.ntaps[ph];
    const int nkc = a.C_src / kBK;
    const int nkt = T * nkc;
    const int HW = kTC + a.tt.dw_max[ph] - a.tt.dw_min[ph];  // halo columns actually used (<= 34)
    const int HR = kTR + a.tt.rows - 1;                      // halo rows (<= 18)

    const bf16_t* src_img = a.src + ((int64_t)n * a.H * a.W_src) * a.ld_src;
    // tap shifts as three scalars (the taps of a phase form a grid; checked by the host), handed over through LDS once per tile
    int* tap_tab = (int*)(smem + kTab);
    if (tid < 16) tap_tab[tid] = tid < T ? (a.tt.dh[ph][tid] - a.tt.dh_min) * kPitch + (a.tt.dw[ph][tid] - a.tt.dw_min[ph]) : 0;
    __syncthreads();
    const int ncol = T / a.tt.rows;  // taps per kernel row
    const int sh0 = __builtin_amdgcn_readfirstlane(tap_tab[0]);
    const int sh_dcol = ncol > 1 ? __builtin_amdgcn_readfirstlane(tap_tab[1]) - sh0 : 0;
    const int sh_drow = ncol < T ? __builtin_amdgcn_readfirstlane(tap_tab[ncol]) - __builtin_amdgcn_readfirstlane(tap_tab[ncol - 1]) : 0;
    const int row_base = h0 + a.tt.dh_min, col_base = m0 + a.tt.dw_min[ph];

    // ---- DMA maps ---------------------------------------------------------------------------------------------------
    // Every DMA instruction moves 16 LDS rows x 64 B; lane = (row-in-16 = lane >> 2, 16-byte slot = lane & 3) and the slot holds
    // logical k-chunk slot ^ 2*((row >> 2) & 1): row = 16 * instruction + lane >> 2, so the same constant for every
    // instruction of a lane.
    const int s_row = lane >> 2, s_slot = lane & 3;
    const int kq8 = (s_slot ^ (((s_row >> 2) & 1) << 1)) * 8;
    const bf16_t* zero = (const bf16_t*)g_zero_page6 + s_slot * 8;
    // Weights: piece j = K tile j = [128 channels][32 k] of the row-major image: wave w's instruction covers channels 16 w + s_row.
    // (Measured and dropped: a tiled copy of the image with every 8 KB piece contiguous and pre-swizzled, i.e. eight whole lines
    //  per instruction instead of sixteen half-lines -- the same kernel time, profiles/r04_tapconv6_ablation.md: all CUs of an XCD
    //  then pull the same 8 KB from one or two L2 channels at the same moment.)
    const int64_t w_img = (int64_t)a.C_dst * a.C_src;
    const bf16_t* w_ph = a.w + (int64_t)a.tt.w_first[ph] * w_img;
    const int b_voff = (n0 + wave * 16 + s_row) * a.C_src + kq8;
    int bq = 0, bt = 0;  // K tile / tap of the piece being issued
    int b_so = 0;        // element offset of its (tap image, chunk) in the packed weight -- kept scalar
    auto stage_b = [&](int j) {
        const int so = __builtin_amdgcn_readfirstlane(b_so);
        const bf16_t* p = w_ph + so + b_voff;
        __builtin_amdgcn_global_load_lds((glb_void_t*)p, (lds_void_t*)(smem + kRing + (j & (kNRing - 1)) * kPiece + wave * 1024), 16, 0, 0);
        const bool go = bq + 1 < nkt;  // pieces past the last K tile re-fetch it (never read; keeps the wait counts uniform)
        const bool wrap = bt + 1 == T;
        bq += go ? 1 : 0;
        b_so += go ? (wrap ? kBK - (T - 1) * (int)w_img : (int)w_img) : 0;
        bt = go ? (wrap ? 0 : bt + 1) : bt;
    };
    // Halo: instruction q (0..40) covers pixel slots 16 q + s_row of the [18 rows][36 slots] image; wave w issues q = w, w + 8,
    // ..., w + 40 (q >= 41: dummies, zero page -> a scratch KB).  Source offsets once per tile (-1 = zero page).
    int hoff[kHaloPer];
#pragma unroll
    for (int i = 0; i < kHaloPer; ++i) {
        const int q = wave + 8 * i;
        const int p = q * 16 + s_row;
        const int hr = (p * 1821) >> 16, hc = p - hr * kPitch;  // 1821 = ceil(65536 / 36): exact for p < 2^12
        const int row = row_base + hr, col = col_base + hc;
        const int bad = (q >= kHaloInstr) | (hc >= HW) | (hr >= HR) | (row < 0) | (row >= a.H) | (col < 0) | (col >= a.W_src);
        hoff[i] = ((row * a.W_src + col) * a.ld_src + kq8) | -bad;
    }
    auto stage_halo = [&](int buf, int i, int kc) {  // i: compile-time index 0 .. kHaloPer - 1
        const int q = wave + 8 * i;
        const bf16_t* src = hoff[i] >= 0 ? src_img + (hoff[i] + kc * kBK) : zero;
        const int dst = q < kHaloInstr ? buf * kHaloBytes + q * 1024 : kScratch;
        __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(smem + dst), 16, 0, 0);
    };

    // ---- fragment reads ---------------------------------------------------------------------------------------------
    // A: output pixel (row r, column c) under a tap = halo slot 36 r + c + shift(tap).  Wave row wr owns rows 4 wr .. 4 wr + 3;
    // fragment i (0..7): row 4 wr + i/2, columns 16 (i&1) + l15.  B: piece row = channel (wc * 64 + 16 j + l15).
    const int p_lane = (4 * wr) * kPitch + l15;
    const int b_rd = (wc * 64 + l15) * 64 + ((lg ^ (((l15 >> 2) & 1) << 1)) * 16);
    bf16x8 fa[4], fb[4];
    int a_b0 = 0, a_b1 = 0;
    auto addr_a = [&](int halo_byte, int shift) {
        const int p = p_lane + shift;
        a_b0 = halo_byte + p * 64 + ((lg ^ (((p >> 2) & 1) << 1)) * 16);
        a_b1 = a_b0 ^ 32;
    };
    auto read_a = [&](int mq) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
                fa[rr * 2 + cc] = *(const bf16x8*)(smem + (((2 * mq + rr) & 1) ? a_b1 : a_b0) + (2 * mq + rr) * (kPitch * 64) + cc * 1024);
    };
    auto read_b = [&](int j) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) fb[jj] = *(const bf16x8*)(smem + kRing + (j & (kNRing - 1)) * kPiece + jj * 1024 + b_rd);
    };

    f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

// One phase: barrier | 16 MFMAs, nothing else | barrier.
#define RV_PHASE_COMPUTE(MQ)                                                                            \
    __builtin_amdgcn_s_barrier();                                                                       \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    __builtin_amdgcn_s_setprio(1);                                                                      \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int jj = 0; jj < 4; ++jj)      \
        acc[(MQ) * 4 + i][jj] = RV_MFMA_16x16x32(fa[i], fb[jj], acc[(MQ) * 4 + i][jj], 0, 0, 0);        \
    __builtin_amdgcn_s_setprio(0);                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    __builtin_amdgcn_s_barrier();

    // ---- prologue: halo of chunk 0, weight pieces 0 .. kAhead - 1 -------------------------------------------------------
#pragma unroll
    for (int i = 0; i < kHaloPer; ++i) stage_halo(0, i, 0);
#pragma unroll
    for (int j = 0; j < kAhead; ++j) stage_b(j);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wave >= 4) __builtin_amdgcn_s_barrier();  // the second half of the workgroup runs one barrier behind the first

    int kt = 0, kc = 0;    // K tile, chunk
    int sh = sh0, ix = 0;  // slot shift / grid column of the tap of K tile kt
    // One K tile: [weight instruction, halo instruction, fragment reads] barrier | 16 MFMAs | barrier [halo instruction,
    // fragment reads, counted wait] barrier | 16 MFMAs | barrier.  HALO: this chunk has a successor whose halo is loaded
    // during positions U = 0, 1, 2.  The wait (with the barrier behind it) makes piece kt + 1 visible to the next K tile.
#define RV_KTILE(U, HALO)                                                                                         \
    {                                                                                                              \
        constexpr int W = (HALO) ? wait_count(U) : kAhead - 1;                                                     \
        const int hbuf = (kc + 1) & 1;                                                                             \
        stage_b(kt + kAhead);                                                                                      \
        if constexpr ((HALO) && (U) <= 2) stage_halo(hbuf, kHaloPerTile * (U), kc + 1);                            \
        addr_a((kc & 1) * kHaloBytes, sh);                                                                         \
        read_b(kt);                                                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        read_a(0);                                                                                                 \
        RV_PHASE_COMPUTE(0);                                                                                       \
        read_a(1);                                                                                                 \
        if constexpr ((HALO) && (U) <= 2) stage_halo(hbuf, kHaloPerTile * (U) + 1, kc + 1);                        \
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W) : "memory");                                                   \
        RV_PHASE_COMPUTE(1);                                                                                       \
        ++kt;                                                                                                      \
        const bool wrap = ix + 1 == ncol;                                                                          \
        sh += wrap ? sh_drow : sh_dcol;                                                                            \
        ix = wrap ? 0 : ix + 1;                                                                                    \
    }
    for (; kc + 1 < nkc; ++kc) {
        RV_KTILE(0, true)
        RV_KTILE(1, true)
        RV_KTILE(2, true)
        RV_KTILE(3, true)
        RV_KTILE(4, true)
        RV_KTILE(5, true)
        if (T > 6) RV_KTILE(6, true)
        if (T > 7) RV_KTILE(7, true)
        if (T > 8) RV_KTILE(8, true)
        for (int t = 9; t < T; ++t) RV_KTILE(9, true)
        sh = sh0;
        ix = 0;
    }
    for (int t = 0; t < T; ++t) RV_KTILE(9, false)  // last chunk: no halo to load
#undef RV_KTILE
#undef RV_PHASE_COMPUTE
    if (wave < 4) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ------------------------------------ epilogue --------------------------------------------
    // acc[i][j][r]: tile row 4*wr + i/2, column (i&1)*16 + lg*4 + r, channel n0 + wc*64 + j*16 + l15
    const int Wm = a.W_dst / a.phases;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool row_ok = h0 + 4 * wr + (i >> 1) < a.H;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + (i & 1) * 16 + lg * 4 + r;
            if (m >= Wm || !row_ok) {
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j][r] = 0.f;
            }
        }
    }
    if (a.flags & RV_OUT_STATS) {
        float* prow = a.stats + ((int64_t)(tile * 4 + wr) * 2) * a.C_dst;
        float* stat_acc = stat_slot();
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i)
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
            const int c = n0 + wc * 64 + j * 16 + l15;
            if (lg == 0) {
                if (a.stats_per_wg) {
                    stat_acc[j * 32] += s;
                    stat_acc[j * 32 + 1] += q;
                } else {
                    prow[c] = s;
                    prow[a.C_dst + c] = q;
                }
            }
        }
    }
    if (a.flags & RV_OUT_BIAS) {
        const bool relu_out = (a.flags & RV_OUT_RELU) != 0;  // eval: BatchNorm folded into weights + bias, ReLU on the way out
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float b = a.bias[n0 + wc * 64 + j * 16 + l15];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = relu_out ? fmaxf(acc[i][j][r] + b, 0.f) : acc[i][j][r] + b;
        }
    }
    constexpr int kEpi = kBN + 8;
    bf16_t* epi = (bf16_t*)smem;  // [16 rows * 32 cols][kEpi]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pm = (4 * wr + (i >> 1)) * kTC + (i & 1) * 16 + lg * 4 + r;
                const int pc = wc * 64 + j * 16 + l15;
                epi[pm * kEpi + pc] = f2bf(acc[i][j][r]);
            }
    constexpr int kChunks = kBN / 8;  // 16
    constexpr bool accum = EPI == 2;
    constexpr bool bnb = EPI == 1;  // BatchNorm-backward sums of the layer whose output gradient is being written
    static_assert(EPI >= 0 && EPI <= 2, "the masked last-writer epilogue (three prefetches per pass) stays on tapconv5");
    // the store loop: thread st of the storing waves keeps ONE 8-channel chunk (st % 16) through all passes
    constexpr int kStoreThreads = 512;
    constexpr int kPasses = kTR * kTC * kChunks / kStoreThreads;  // 16
    int st = tid;
    asm volatile("" : "+v"(st));  // opaque HERE: the store loop's per-pass offsets are formed after the K loop, not hoisted above it (spills)
    float bsc[8], bsh[8], bmu[8], bis[8], s0[8], s1[8];
    if (bnb) {
        const int c = n0 + (st & (kChunks - 1)) * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bsc[j] = a.bnb_scale[c + j];
            bsh[j] = a.bnb_shift[c + j];
            bmu[j] = a.bnb_mean[c + j];
            bis[j] = a.bnb_invstd[c + j];
            s0[j] = 0.f;
            s1[j] = 0.f;
        }
    }
    // Passes in ROUNDS of eight with the next round's global reads (y of the BatchNorm sums / the old values of an accumulating
    // launch) in flight while the current one is stored: 2 x 8 prefetch registers per array instead of one per pass.  The reads are unconditional (clamped coordinates): straight-line code, so the compiler counts its vmcnt waits exactly
    // instead of draining the queue at the first use.
    constexpr int kRound = 8, kRounds = kPasses / kRound;
    constexpr bool pre = bnb || accum;
    u32x4 yv[pre ? 2 : 1][pre ? kRound : 1];
    auto prefetch = [&](int r, int buf) {
#pragma unroll
        for (int i = 0; i < kRound; ++i) {
            const int q = st + (r * kRound + i) * kStoreThreads, pm = q / kChunks, c8 = q - pm * kChunks;
            const int rr = pm / kTC, mm = pm - rr * kTC;
            const int m = min(m0 + mm, Wm - 1), hh = min(h0 + rr, a.H - 1), c = n0 + c8 * 8;
            const int64_t px = ((int64_t)(n * a.H + hh) * a.W_dst) + (a.phases * m + ph);
            yv[pre ? buf : 0][pre ? i : 0] = bnb ? *(const u32x4*)(a.bnb_y + px * a.ld_bnb_y + c) : *(const u32x4*)(a.res + px * a.ld_res + c);
        }
    };
    if (pre) prefetch(0, 0);  // (before the barrier that publishes the staged tile)
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        if (pre && r + 1 < kRounds) prefetch(r + 1, (r + 1) & 1);
#pragma unroll
        for (int i = 0; i < kRound; ++i) {
            const int q = st + (r * kRound + i) * kStoreThreads;
            const int pm = q / kChunks, c8 = q - pm * kChunks;
            const int rr = pm / kTC, mm = pm - rr * kTC;
            const int m = m0 + mm, c = n0 + c8 * 8, hh = h0 + rr;
            if (m >= Wm || hh >= a.H) continue;
            u32x4 v = *(const u32x4*)(epi + pm * kEpi + c8 * 8);
            const int64_t px = ((int64_t)(n * a.H + hh) * a.W_dst) + (a.phases * m + ph);
            bf16_t* p = (bf16_t*)a.dst + px * a.ld_dst + c;
            const u32x4 pv = yv[pre ? (r & 1) : 0][pre ? i : 0];
            if (accum) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = pack_bf2(bf_lo(v[j]) + bf_lo(pv[j]), bf_hi(v[j]) + bf_hi(pv[j]));
                if (a.flags & RV_OUT_RES_RELU) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = pack_bf2(fmaxf(bf_lo(v[j]), 0.f), fmaxf(bf_hi(v[j]), 0.f));
                }
            }
            *(u32x4*)p = v;
            if (bnb) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float y0 = bf_lo(pv[j]), y1 = bf_hi(pv[j]);
                    float g0 = bf_lo(v[j]), g1 = bf_hi(v[j]);
                    if (a.bnb_flags & 1) {  // RV_BNB_RELU_Z
                        g0 = y0 * bsc[2 * j] + bsh[2 * j] > 0.f ? g0 : 0.f;
                        g1 = y1 * bsc[2 * j + 1] + bsh[2 * j + 1] > 0.f ? g1 : 0.f;
                    }
                    s0[2 * j] += g0;
                    s0[2 * j + 1] += g1;
                    s1[2 * j] += g0 * ((y0 - bmu[2 * j]) * bis[2 * j]);
                    s1[2 * j + 1] += g1 * ((y1 - bmu[2 * j + 1]) * bis[2 * j + 1]);
                }
            }
        }
    }
    if (bnb) {
        // lanes kChunks apart hold the same chunk; then the storing waves through LDS (the staged tile is dead now)
        __syncthreads();
        float* red = (float*)smem;  // [storing waves][kChunks][16]
        constexpr int kSW = kStoreThreads / 64;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int d = kChunks; d < 64; d <<= 1) {
                s0[j] += __shfl_xor(s0[j], d, 64);
                s1[j] += __shfl_xor(s1[j], d, 64);
            }
        }
        if (lane < kChunks) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                red[((st >> 6) * kChunks + lane) * 16 + j] = s0[j];
                red[((st >> 6) * kChunks + lane) * 16 + 8 + j] = s1[j];
            }
        }
        __syncthreads();
        const int chunk = st >> 4, jj = st & 15;  // kChunks x 16 values, summed by the first 256 storing threads
        if (chunk < kChunks) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < kSW; ++w) sum += red[(w * kChunks + chunk) * 16 + jj];
            if (a.stats_per_wg) ((float*)(smem + kBnbAcc))[st] += sum;  // (st = chunk * 16 + jj: a private slot)
            else a.bnb_partial[((int64_t)tile * 2 + (jj >> 3)) * a.C_dst + n0 + chunk * 8 + (jj & 7)] = sum;
        }
    }
    __syncthreads();  // the staged output / BatchNorm sums of this tile are dead before the next tile's loads land in LDS
    }  // persistent tile loop
    if ((EPI == 1) && a.stats_per_wg && tid < 256) {
        const int so = tid, chunk = so >> 4, jj = so & 15;
        a.bnb_partial[((int64_t)(xcd * (nslots / gy) + wslot / gy) * 2 + (jj >> 3)) * a.C_dst + (wslot % gy) * kBN + chunk * 8 + (jj & 7)] =
            ((const float*)(smem + kBnbAcc))[so];
    }
    if ((a.flags & RV_OUT_STATS) && a.stats_per_wg && lg == 0) {
        // (nslots % gy == 0, checked by the host: the channel tile of a workgroup is the same for all of its tiles, and the gy
        //  workgroups wslot / gy == const of one XCD fill one row group between them -- workgroups without a tile write zeros)
        float* prow = a.stats + ((int64_t)((xcd * (nslots / gy) + wslot / gy) * 4 + wr) * 2) * a.C_dst;
        const float* stat_acc = stat_slot();
        const int c0 = (wslot % gy) * kBN + wc * 64 + l15;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            prow[c0 + j * 16] = stat_acc[j * 32];
            prow[a.C_dst + c0 + j * 16] = stat_acc[j * 32 + 1];
        }
    }
}


}  // namespace


// returns false when the layer is not eligible (caller falls back to tapconv5 / tapconv4 / ...)
static int tapconv6_grid(const TapConvArgs& a) {
    int grid = 8 * a.tiles_per_xcd * a.n_tiles;
    if (grid > rv_cu_count()) grid = rv_persistent_grid();  // one workgroup per CU
    return grid;
}

bool rv_tapconv6_plan(TapConvArgs* a, int* tiles, size_t* lds, int* stats_rows, int* bnb_rows) {
    if (a->step != 1) return false;
    if (a->flags & (RV_IN_AFFINE | RV_IN_RELU | RV_OUT_F32)) return false;  // the DMA path has no register prologue
    // BatchNorm-backward sums over an ACCUMULATED gradient (the round-3/4 "masked last-writer" form: three 16-byte prefetches per
    // pass, 192 registers over this tile's sixteen passes, 177 of them spilled, 16 % matrix-pipe occupancy -- profiles/r04_mfma_counters.json)
    // exist in NO generation since round 5 (tapconv5 rejects the combination as well, rv_tap_bnb_rows returns 0): the caller takes the
    // separate reduce pass over the complete gradient
    if ((a->flags & RV_OUT_BNB) && (a->flags & RV_OUT_ACCUM)) return false;
    if (a->C_src % kBK != 0 || a->C_dst % kBN != 0) return false;
    const int wm_total = a->W_dst / a->phases;
    if (wm_total < kTC || a->H < kTR) return false;
    if (kTR + a->tt.rows - 1 > kHaloRows) return false;
    for (int r = 0; r < a->phases; ++r) {
        if (a->tt.ntaps[r] < kMinTaps || a->tt.ntaps[r] > 16) return false;
        const int hw = kTC + a->tt.dw_max[r] - a->tt.dw_min[r];
        if (hw > 34) return false;
        // the kernel steps through the taps as a (rows x columns) grid: check that this phase's table is one
        const int T = a->tt.ntaps[r], nrow = a->tt.rows;
        if (T % nrow != 0) return false;
        const int ncol = T / nrow;
        auto shift = [&](int t) { return (a->tt.dh[r][t] - a->tt.dh_min) * kPitch + (a->tt.dw[r][t] - a->tt.dw_min[r]); };
        const int dcol = ncol > 1 ? shift(1) - shift(0) : 0, drow = ncol < T ? shift(ncol) - shift(ncol - 1) : 0;
        int sh = shift(0), ix = 0;
        for (int t = 1; t < T; ++t) {
            if (ix + 1 == ncol) {
                sh += drow;
                ix = 0;
            } else {
                sh += dcol;
                ++ix;
            }
            if (sh != shift(t)) return false;
        }
    }
    if ((int64_t)a->H * a->W_src * a->ld_src >= (1ll << 31) || (int64_t)kMaxTaps * a->C_dst * a->C_src >= (1ll << 31)) return false;
    a->m_tiles = rv_ceil_div(wm_total, kTC);
    a->h_tiles = rv_ceil_div(a->H, kTR);
    a->total_tiles = a->m_tiles * a->h_tiles * a->N * a->phases;
    a->n_tiles = a->C_dst / kBN;
    a->tiles_per_xcd = rv_ceil_div(a->total_tiles, 8);
    if ((int64_t)a->total_tiles * a->n_tiles < ((a->sel & RV_SEL_SMALL_GRIDS6) ? 1 : rv_cu_count())) return false;  // fewer tiles than CUs: the 256-pixel tiles fill the chip better
    *tiles = a->total_tiles;
    const int grid = tapconv6_grid(*a), nslots = grid / 8;
    a->stats_per_wg = (nslots % a->n_tiles == 0) ? 1 : 0;
    *stats_rows = a->stats_per_wg ? (grid / a->n_tiles) * 4 : a->total_tiles * 4;
    *bnb_rows = a->stats_per_wg ? grid / a->n_tiles : a->total_tiles;
    *lds = (size_t)kLds;
    const size_t epi = (size_t)kTR * kTC * (kBN + 8) * sizeof(bf16_t);
    if (*lds < epi) *lds = epi;
    return true;
}

int rv_tapconv6_launch(const TapConvArgs& a, size_t lds, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)tapconv6_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)tapconv6_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)tapconv6_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int grid = tapconv6_grid(a);
    const int epi = (a.flags & RV_OUT_BNB) ? 1 : ((a.flags & RV_OUT_ACCUM) ? 2 : 0);
    if (epi == 1) hipLaunchKernelGGL((tapconv6_kernel<1>), dim3(grid), dim3(512), lds, stream, a);
    else if (epi == 2) hipLaunchKernelGGL((tapconv6_kernel<2>), dim3(grid), dim3(512), lds, stream, a);
    else hipLaunchKernelGGL((tapconv6_kernel<0>), dim3(grid), dim3(512), lds, stream, a);
    RV_CHECK_LAUNCH("tapconv6_kernel");
    return 0;
}

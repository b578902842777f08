// tapconv4.hip -- fourth-generation tap-conv kernel: 256 pixels x 256 channels per workgroup, 64-channel K tiles,
// operands streamed global -> LDS by LDS-DMA (global_load_lds_dwordx4) that stays in flight across barriers.
//
// Why: tapconv3 (register staging, 128-channel tiles, ds_write pass, __syncthreads) tops out at ~900-970 TFLOP/s on the
// 512->512 3x3 layers; every K step waits for its staging loads at the barrier.  Here nothing on the K loop passes
// through VGPRs on its way to LDS, the only waits are COUNTED (s_waitcnt vmcnt(8): four 16 KB pieces stay in flight
// over every barrier), and the two halves of the workgroup run one barrier apart, so one wave of each SIMD issues
// MFMAs while the other reads its fragments.
//
// Tile: 4 image rows x 64 columns (M = 256) x 256 output channels; 8 waves as 2 (M) x 4 (N), 128 x 64 outputs per wave
// (acc = 128 VGPRs).  One K tile = one tap x 64 input channels = four 16 KB pieces in LDS (two buffers, 128 KiB):
//     X0  input rows {0, 2} of the tile   (first 64 M rows of each wave pair)      read in phase 0
//     X1  weights, first 32 channels of each wave's 64                              read in phase 0
//     X2  weights, last 32 channels                                                 read in phase 1
//     X3  input rows {1, 3}                                                         read in phase 2
// A K tile is four phases of 16 MFMAs per wave (one 64 x 32 quadrant x K = 64); each phase issues one piece:
//     phase 0: X2(kt+1)   phase 1: X3(kt+1)   phase 2: X0(kt+2)   phase 3: X1(kt+2)
// i.e. a piece is re-staged two or three phases after its last read and lands five phases before its first.
// Each piece is a [128 rows][64 k] bf16 image (128-byte rows) written lane-linearly by the DMA; the XOR swizzle that
// makes the ds_read_b128 fragment reads conflict-free (16-byte slot = k-chunk ^ (row & 7)) is applied to the per-lane
// SOURCE address.  Zero padding: lanes whose pixel falls outside the image read a zero page instead.
//
// Eligible layers: stride-1 phases (1x1 included: 1.8-2.1x tapconv2 on the 256-channel pointwise convs), C_src % 64 == 0, C_dst % 256 == 0, plain bf16 input (no folded
// BatchNorm on the way in -- the DMA bypasses the registers), bf16 output (+ stats / bias / accumulate).
#include <stdlib.h>

#include "common.h"
#include "tapconv.h"

namespace {

constexpr int kTC = 64;    // tile columns
constexpr int kTR = 4;     // tile rows
constexpr int kBK = 64;    // K tile (channels of one tap)
constexpr int kPiece = 128 * 128;    // bytes of one piece
constexpr int kBuf = 4 * kPiece;     // one K tile
constexpr int kTabOffset = 2 * kBuf; // tap table behind the two buffers

__device__ __attribute__((aligned(256))) uint32_t g_zero_page[64];

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

struct Staged {   // the K tile a piece belongs to
    int aoff;     // element offset of (tap, channel chunk) in the source image
    int woff;     // element offset of (tap image, channel chunk) in the packed weight
    uint32_t vb;  // validity of the thread's four tile rows under this tap
};

// byte offset of the per-workgroup statistic accumulators in dynamic LDS (see the kernel)
constexpr int stat_acc_offset(int bn) {
    const int tab_end = (bn == 256 ? kTabOffset : 9 * kPiece) + 256, epi_end = kTR * kTC * (bn + 8) * 2;
    return ((tab_end > epi_end ? tab_end : epi_end) + 15) & ~15;
}

// BN: channels per workgroup tile, 256 (two 16 KB weight pieces per K tile) or 128 (one).  ACC: the launch accumulates into dst
// (RV_OUT_ACCUM) -- a template parameter so that only that instance carries the prefetch registers of the old values.
template <int BN, bool ACC>
__global__ __launch_bounds__(512, 2) void tapconv4_kernel(const TapConvArgs a) {
    constexpr int NJ = BN / 64;  // 16-channel accumulator fragments per wave (wave tile = 128 pixels x BN/4 channels)
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, lg = lane >> 4;

    // XCD-aware block order (see tapconv3.hip); persistent workgroups as in tapconv5.hip (a 1x1 layer has only four K tiles
    // per tile: the per-workgroup dispatch is a large share of a tile's life here)
    const int gy = a.n_tiles;
    const int vtotal = 8 * a.tiles_per_xcd * gy;
    // RV_OUT_STATS of a persistent launch (stats_per_wg, as in tapconv6.hip): the (sum, sum of squares) rows of a workgroup's tiles are
    // accumulated in LDS and written once at the end -- 2 rows per group of gy workgroups instead of 2 per tile (4096 rows behind a 1x1
    // layer at 4 x 64 x 2048: the finalize then needs its two-stage column reduction; <= 2048 rows take the one-launch form).
    // behind the tap table AND behind the staged output tile of the epilogue (which overwrites the buffers from offset 0): float [8 waves][NJ][16 channels][2]
    constexpr int kStatAcc = stat_acc_offset(BN);
    auto stat_slot = [&]() { return (float*)(smem + kStatAcc) + (wave * NJ * 16 + l15) * 2; };  // + j * 32
    if (a.stats_per_wg && lg == 0) {
        float* stat_acc = stat_slot();
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            stat_acc[j * 32] = 0.f;
            stat_acc[j * 32 + 1] = 0.f;
        }
    }
    for (int vb = blockIdx.x; vb < vtotal; vb += gridDim.x) {
    const int xcd = vb & 7, xslot = vb >> 3;
    const int tile = xcd * a.tiles_per_xcd + xslot / gy;
    if (tile >= a.total_tiles) continue;
    const int n0 = (xslot % gy) * BN;
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

    const bf16_t* src_img = a.src + ((int64_t)n * a.H * a.W_src) * a.ld_src;
    // per-tap tables in LDS: element offset of the tap, and its (dh, dw).  Each of the first T threads fetches its own
    // tap from the kernel arguments (one load latency for the whole table); nothing below indexes the arguments with a
    // runtime tap number, which would cost a dependent global load per tap and thread.
    constexpr int kTab = (BN == 256) ? kTabOffset : 9 * kPiece;  // behind two 64 KB buffers / three 48 KB buffers
    int* tap_tab = (int*)(smem + kTab);
    int* tap_dhw = tap_tab + 32;
    if (tid < T) {
        const int dh = a.tt.dh[ph][tid], dw = a.tt.dw[ph][tid];
        tap_tab[tid] = (dh * a.W_src + dw) * a.ld_src;
        tap_dhw[tid] = (dh << 16) | (dw & 0xffff);
    }
    __syncthreads();

    // ---- staging map: thread = (tile column, 16-byte slot); two rows per piece ------------------------------
    const int s_c = tid >> 3, s_slot = tid & 7;
    const int kq = s_slot ^ (s_c & 7);  // logical k-chunk held by LDS slot s_slot of rows == s_c (mod 8)
    uint64_t vmask = 0;
    for (int t = 0; t < T; ++t) {
        const int e = tap_dhw[t];
        const int dh = e >> 16, dw = (int)(int16_t)(e & 0xffff);
        const int ws = m0 + s_c + dw;
        const bool cok = ws >= 0 && ws < a.W_src;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
            const int hs = h0 + rr + dh;
            if (cok && hs >= 0 && hs < a.H) vmask |= 1ull << (t * 4 + rr);
        }
    }
    const int row_stride = a.W_src * a.ld_src;
    const int a_base = (h0 * a.W_src + m0 + s_c) * a.ld_src + kq * 8;
    // weights: piece row = (wave column wcc, channel jl of the 32 it reads per piece); rows tid>>3 and 64 + tid>>3
    const int b_base = (n0 + ((s_c >> 5) * (BN / 4)) + (s_c & 31)) * a.C_src + kq * 8;
    const int64_t w_img = (int64_t)a.C_dst * a.C_src;
    const bf16_t* w_ph = a.w + (int64_t)a.tt.w_first[ph] * w_img;
    const bf16_t* zero = (const bf16_t*)g_zero_page + s_slot * 8;
    const int lds_lane_base = wave * 1024;  // this wave's 1 KiB of each 8 KiB half piece

    auto make_staged = [&](int q) {
        q = q < nkt ? q : nkt - 1;  // pieces past the last K tile re-fetch it (never read; keeps the wait counts uniform)
        const int kc = q / T, t = q - kc * T;
        Staged s;
        s.aoff = tap_tab[t] + kc * kBK;
        s.woff = t * (int)w_img + kc * kBK;
        s.vb = (uint32_t)(vmask >> (t * 4)) & 15u;
        return s;
    };
    auto stage_a = [&](int piece_byte, int mq, const Staged& s) {  // rows {mq, mq + 2} of the tile
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rr = 2 * i + mq;
            const bf16_t* p = ((s.vb >> rr) & 1u) ? src_img + (a_base + rr * row_stride + s.aoff) : zero;
            __builtin_amdgcn_global_load_lds((glb_void_t*)p, (lds_void_t*)(smem + piece_byte + i * 8192 + lds_lane_base), 16, 0, 0);
        }
    };
    auto stage_b = [&](int piece_byte, int nq, const Staged& s) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const bf16_t* p = w_ph + (s.woff + b_base + (i * (BN / 2) + nq * 32) * a.C_src);
            __builtin_amdgcn_global_load_lds((glb_void_t*)p, (lds_void_t*)(smem + piece_byte + i * 8192 + lds_lane_base), 16, 0, 0);
        }
    };

    // ---- fragment reads -------------------------------------------------------------------------------------
    const int swz = (lg ^ (l15 & 7)) * 16;                       // k-chunk lg of the first 32 k; ^64 for the second
    const int a_rd = (wr * 64 + l15) * 128 + swz;                // + i * 2048 per m-frag
    const int b_rd = (wc * 32 + l15) * 128 + swz;                // + jj * 2048 per n-frag
    bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
    auto read_a = [&](int piece_byte) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fa[i][ks] = *(const bf16x8*)(smem + piece_byte + i * 2048 + (a_rd ^ (ks * 64)));
    };
    auto read_b = [&](bf16x8 (&fb)[2][2], int piece_byte) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fb[jj][ks] = *(const bf16x8*)(smem + piece_byte + jj * 2048 + (b_rd ^ (ks * 64)));
    };

    f32x4 acc[8][NJ];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

// One phase: [fragment reads, this phase's piece (two DMA instructions), counted wait] barrier | 16 MFMAs | barrier.
#define RV_MFMA_HALF(MQ, NQ, FB, KS)                                                                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                 \
        acc[(MQ) * 4 + i][(NQ) * 2 + jj] =                                                                         \
            RV_MFMA_16x16x32(fa[i][KS], FB[jj][KS], acc[(MQ) * 4 + i][(NQ) * 2 + jj], 0, 0, 0);
#define RV_PHASE_COMPUTE(MQ, NQ, FB, STAGE)                \
    __builtin_amdgcn_s_barrier();                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_setprio(1);                         \
    RV_MFMA_HALF(MQ, NQ, FB, 0)                            \
    __builtin_amdgcn_sched_barrier(0);                     \
    STAGE;                                                 \
    __builtin_amdgcn_sched_barrier(0);                     \
    RV_MFMA_HALF(MQ, NQ, FB, 1)                            \
    __builtin_amdgcn_s_setprio(0);                         \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_barrier();
#define RV_WAIT_PIECES(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

    if constexpr (BN == 256) {
        // ---- prologue: K tile 0 complete, X0 / X1 of K tile 1 ---------------------------------------------------
        Staged scur = make_staged(0);
        stage_a(0 * kPiece, 0, scur);
        stage_b(1 * kPiece, 0, scur);
        stage_b(2 * kPiece, 1, scur);
        stage_a(3 * kPiece, 1, scur);
        scur = make_staged(1);
        stage_a(kBuf + 0 * kPiece, 0, scur);
        stage_b(kBuf + 1 * kPiece, 0, scur);
        RV_WAIT_PIECES(8);  // X0(0), X1(0) landed; four pieces in flight
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();  // the second half of the workgroup runs one barrier behind the first

        for (int kt = 0; kt < nkt; ++kt) {
            const int rb = (kt & 1) * kBuf, ob = kBuf - rb;
            const Staged snext = make_staged(kt + 2);  // its tap-table read lands long before phase 2 needs it
            // The piece of each phase is issued in its read section, before the counted wait: four pieces stay in flight.
            // Measured and dropped (same process, same device): issuing the DMA from inside the MFMA cluster (-5 %); a 4-byte
            // DMA per K tile that warms L2 with the next channel chunk's input lines (-3..5 %).
            read_b(fb0, rb + 1 * kPiece);
            __builtin_amdgcn_sched_barrier(0);
            read_a(rb + 0 * kPiece);
            stage_b(ob + 2 * kPiece, 1, scur);
            RV_WAIT_PIECES(8);
            RV_PHASE_COMPUTE(0, 0, fb0, (void)0);
            read_b(fb1, rb + 2 * kPiece);
            stage_a(ob + 3 * kPiece, 1, scur);
            RV_WAIT_PIECES(8);
            RV_PHASE_COMPUTE(0, 1, fb1, (void)0);
            scur = snext;
            read_a(rb + 3 * kPiece);
            stage_a(rb + 0 * kPiece, 0, scur);
            RV_PHASE_COMPUTE(1, 1, fb1, (void)0);
            stage_b(rb + 1 * kPiece, 0, scur);
            RV_WAIT_PIECES(8);
            RV_PHASE_COMPUTE(1, 0, fb0, (void)0);
        }
    } else {
        // BN = 128: a K tile is THREE pieces (input rows {0,2}, weights, input rows {1,3}; 48 KB) and two phases; the
        // ring holds three K tiles.  Phase 0 of K tile kt issues A0 and B of kt+2 (their regions of that buffer were last
        // read two phases earlier), phase 1 issues A1 of kt+2; the waits leave 10 / 8 DMA instructions in flight.
        constexpr int kBuf3 = 3 * kPiece;
        Staged scur = make_staged(0);
        stage_a(0 * kPiece, 0, scur);
        stage_b(1 * kPiece, 0, scur);
        stage_a(2 * kPiece, 1, scur);
        scur = make_staged(1);
        stage_a(kBuf3 + 0 * kPiece, 0, scur);
        stage_b(kBuf3 + 1 * kPiece, 0, scur);
        stage_a(kBuf3 + 2 * kPiece, 1, scur);
        RV_WAIT_PIECES(8);  // A0(0), B(0) landed
        __builtin_amdgcn_s_barrier();
        if (wr == 1) __builtin_amdgcn_s_barrier();
        int rb = 0, sb = 2 * kBuf3;  // buffers of K tile kt and kt+2
        for (int kt = 0; kt < nkt; ++kt) {
            scur = make_staged(kt + 2);
            read_b(fb0, rb + 1 * kPiece);
            __builtin_amdgcn_sched_barrier(0);
            read_a(rb + 0 * kPiece);
            stage_a(sb + 0 * kPiece, 0, scur);
            stage_b(sb + 1 * kPiece, 0, scur);
            RV_WAIT_PIECES(10);  // A1(kt) landed
            RV_PHASE_COMPUTE(0, 0, fb0, (void)0);
            read_a(rb + 2 * kPiece);
            stage_a(sb + 2 * kPiece, 1, scur);
            RV_WAIT_PIECES(8);  // A0(kt+1), B(kt+1) landed
            RV_PHASE_COMPUTE(1, 0, fb0, (void)0);
            rb = (rb == 2 * kBuf3) ? 0 : rb + kBuf3;
            sb = (sb == 2 * kBuf3) ? 0 : sb + kBuf3;
        }
    }
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#undef RV_MFMA_HALF
#undef RV_PHASE_COMPUTE
#undef RV_WAIT_PIECES

    // ------------------------------------ epilogue --------------------------------------------
    // acc[mq*4+i][j][r]: tile row 2*wr + mq, column i*16 + lg*4 + r, channel n0 + wc*(BN/4) + j*16 + l15
    constexpr int WN = BN / 4;
    const int Wm = a.W_dst / a.phases;
#pragma unroll
    for (int mq = 0; mq < 2; ++mq) {
        const bool row_ok = h0 + 2 * wr + mq < a.H;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + i * 16 + lg * 4 + r;
                if (m >= Wm || !row_ok) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) acc[mq * 4 + i][j][r] = 0.f;
                }
            }
    }
    if (a.flags & RV_OUT_STATS) {
        float* prow = a.stats + ((int64_t)(tile * 2 + wr) * 2) * a.C_dst;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
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
            const int c = n0 + wc * WN + j * 16 + l15;
            if (lg == 0) {
                if (a.stats_per_wg) {
                    float* stat_acc = stat_slot();
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
        for (int j = 0; j < NJ; ++j) {
            const float b = a.bias[n0 + wc * WN + j * 16 + l15];
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = relu_out ? fmaxf(acc[i][j][r] + b, 0.f) : acc[i][j][r] + b;
        }
    }
    constexpr int kEpi = BN + 8;
    bf16_t* epi = (bf16_t*)smem;  // [4 rows * 64 cols][kEpi]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pm = (2 * wr + (i >> 2)) * kTC + (i & 3) * 16 + lg * 4 + r;
                const int pc = wc * WN + j * 16 + l15;
                epi[pm * kEpi + pc] = f2bf(acc[i][j][r]);
            }
    constexpr int kChunks = BN / 8;
    constexpr int kPasses = kTR * kTC * kChunks / 512;
    constexpr bool accum = ACC;
    // accumulate: the old values of this thread's chunks all in flight before the barrier that publishes the staged tile (read
    // inside the store loop each one waited out its own round trip between two stores)
    u32x4 ov[accum ? kPasses : 1];
    if (accum) {
#pragma unroll
        for (int it = 0; it < kPasses; ++it) {
            const int q = tid + it * 512, pm = q / kChunks, c8 = q - pm * kChunks;
            const int rr = pm / kTC, mm = pm - rr * kTC;
            const int m = m0 + mm, c = n0 + c8 * 8, hh = h0 + rr;
            ov[accum ? it : 0] = u32x4{0u, 0u, 0u, 0u};
            if (m < Wm && hh < a.H)
                ov[accum ? it : 0] = *(const u32x4*)(a.res + (((int64_t)(n * a.H + hh) * a.W_dst) + (a.phases * m + ph)) * a.ld_res + c);
        }
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < kPasses; ++it) {
        const int q = tid + it * 512;
        const int pm = q / kChunks, c8 = q - pm * kChunks;
        const int rr = pm / kTC, mm = pm - rr * kTC;
        const int m = m0 + mm, c = n0 + c8 * 8, hh = h0 + rr;
        if (m >= Wm || hh >= a.H) continue;
        u32x4 v = *(const u32x4*)(epi + pm * kEpi + c8 * 8);
        bf16_t* p = (bf16_t*)a.dst + (((int64_t)(n * a.H + hh) * a.W_dst) + (a.phases * m + ph)) * a.ld_dst + c;
        if (accum) {
            const u32x4 o = ov[accum ? it : 0];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = pack_bf2(bf_lo(v[j]) + bf_lo(o[j]), bf_hi(v[j]) + bf_hi(o[j]));
            if (a.flags & RV_OUT_RES_RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = pack_bf2(fmaxf(bf_lo(v[j]), 0.f), fmaxf(bf_hi(v[j]), 0.f));
            }
        }
        *(u32x4*)p = v;
    }
    __syncthreads();  // the staged output of this tile is dead before the next tile's loads land in LDS
    }  // persistent tile loop
    if ((a.flags & RV_OUT_STATS) && a.stats_per_wg && lg == 0) {
        // (gridDim.x / 8 is a multiple of gy, checked by the host: the channel tile of a workgroup is the same for all of its tiles, and the
        //  gy workgroups with the same wslot / gy of one XCD fill one row group between them -- workgroups without a tile write zeros)
        const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
        float* prow = a.stats + ((int64_t)((xcd * (nslots / gy) + wslot / gy) * 2 + wr) * 2) * a.C_dst;
        const float* stat_acc = stat_slot();
        const int c0 = (wslot % gy) * BN + wc * (BN / 4) + l15;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            prow[c0 + j * 16] = stat_acc[j * 32];
            prow[a.C_dst + c0 + j * 16] = stat_acc[j * 32 + 1];
        }
    }
}

}  // namespace


// Smallest grid (workgroups) the DMA kernel is chosen for: below one round of 256 CUs the register-staged kernels with
// their smaller tiles fill the chip better.  A speed heuristic only -- the per-call hint RV_SEL_SMALL_GRIDS (rvTapShape.flags) lifts
// it so that the parity tests run the production kernels on crops the CPU oracle can afford.

// returns false when the layer is not eligible (caller falls back to tapconv2 / the generic kernel)
static int tapconv4_grid(const TapConvArgs& a) {
    int grid = 8 * a.tiles_per_xcd * a.n_tiles;
    if (grid > rv_cu_count()) grid = rv_persistent_grid();  // persistent: one workgroup per CU
    return grid;
}

bool rv_tapconv4_plan(TapConvArgs* a, int* tiles, size_t* lds, int* bn, int* stats_rows) {
    if (a->step != 1) return false;
    if (a->flags & (RV_IN_AFFINE | RV_IN_RELU | RV_OUT_F32)) return false;  // the DMA path has no register prologue
    if (a->C_src % kBK != 0 || a->C_dst % 128 != 0) return false;
    const int BN = (a->C_dst % 256 == 0) ? 256 : 128;
    const int wm_total = a->W_dst / a->phases;
    if (wm_total < kTC || a->H < kTR) return false;
    for (int r = 0; r < a->phases; ++r)
        if (a->tt.ntaps[r] < 1 || a->tt.ntaps[r] > 16) return false;
    if ((int64_t)a->H * a->W_src * a->ld_src >= (1ll << 31) || (int64_t)kMaxTaps * a->C_dst * a->C_src >= (1ll << 31)) return false;
    a->m_tiles = rv_ceil_div(wm_total, kTC);
    a->h_tiles = rv_ceil_div(a->H, kTR);
    a->total_tiles = a->m_tiles * a->h_tiles * a->N * a->phases;
    a->n_tiles = a->C_dst / BN;
    a->tiles_per_xcd = rv_ceil_div(a->total_tiles, 8);
    if ((int64_t)a->total_tiles * a->n_tiles < ((a->sel & RV_SEL_SMALL_GRIDS) ? 1 : rv_cu_count())) return false;  // too few tiles to fill the chip
    *tiles = a->total_tiles;
    *bn = BN;
    const int grid = tapconv4_grid(*a), nslots = grid / 8;
    a->stats_per_wg = (grid < 8 * a->tiles_per_xcd * a->n_tiles && nslots % a->n_tiles == 0) ? 1 : 0;  // persistent, channel tile fixed per workgroup
    *stats_rows = a->stats_per_wg ? (grid / a->n_tiles) * 2 : a->total_tiles * 2;
    *lds = (size_t)stat_acc_offset(BN) + 8 * 4 * 16 * 2 * sizeof(float);
    const size_t epi = (size_t)kTR * kTC * (BN + 8) * sizeof(bf16_t);
    if (*lds < epi) *lds = epi;
    return true;
}

int rv_tapconv4_launch(const TapConvArgs& a, size_t lds, int bn, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)tapconv4_kernel<256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)tapconv4_kernel<128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)tapconv4_kernel<256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)tapconv4_kernel<128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const int grid = tapconv4_grid(a);
    const bool acc = (a.flags & RV_OUT_ACCUM) != 0;
    if (bn == 256) {
        if (acc) hipLaunchKernelGGL((tapconv4_kernel<256, true>), dim3(grid), dim3(512), lds, stream, a);
        else hipLaunchKernelGGL((tapconv4_kernel<256, false>), dim3(grid), dim3(512), lds, stream, a);
    } else {
        if (acc) hipLaunchKernelGGL((tapconv4_kernel<128, true>), dim3(grid), dim3(512), lds, stream, a);
        else hipLaunchKernelGGL((tapconv4_kernel<128, false>), dim3(grid), dim3(512), lds, stream, a);
    }
    RV_CHECK_LAUNCH("tapconv4_kernel");
    return 0;
}

// tapconv5.hip -- fifth-generation tap-conv kernel for multi-tap layers: 256 pixels x 256 channels per workgroup like
// tapconv4, but the INPUT of a channel chunk stays resident in LDS for all of its taps.
//
// Why (round-2 ablations of tapconv4 on the 512->512 3x3 layer, profiles/r02_tapconv_ablation.md): with every DMA removed
// the loop runs at 1950 TFLOP/s, with the MFMAs removed the loads alone take as long as the MFMAs alone, and together they
// take the SUM -- each CU pulls 64 KB per K tile (one tap x 64 channels: 32 KB of input + 32 KB of weights) from L2, which
// is 90 % of the ~70 GB/s per CU an LDS fill stream reaches on this chip, so the loads cannot hide behind the MFMAs.
// The lever is bytes per FLOP.  A 3x3 layer re-stages the same input pixels nine times (once per tap); here the halo of a
// 64-channel chunk (10 rows x 34 columns for an 8 x 32 tile, 45 KB with the pad columns) is loaded ONCE, double-buffered
// against the next chunk, and the nine taps read it at shifted LDS rows: 45 + 9 x 32 = 333 KB per chunk instead of 576 KB.
//
// Tile: 8 image rows x 32 columns (M = 256; halo overhead 1.33x, against 1.55x for 4 x 64) x 256 channels.
// 8 waves as 2 (M) x 4 (N), 128 x 64 outputs per wave (128 accumulator VGPRs), four phases of 16 MFMAs per K tile,
// the two halves of the workgroup one barrier apart -- the MFMA/barrier skeleton of tapconv4 (which alone sustains
// 2100 TFLOP/s).  Loads, the same branch-free code on all eight waves:
//   weights: one 16 KB piece ([4 wave columns x 32 channels][64 k]) per two phases into a ring of four pieces, one
//            instruction per wave and phase, issued one and a half K tiles ahead of its first read;
//   halo:    the NEXT chunk's 45 instructions, six per wave, one per phase during the first one and a half K tiles of the
//            current chunk (K-tile bodies U = 0, 1 below); landed by the end of its third K tile.
//   vmcnt retires in order, so a counted wait also waits for every halo load issued before the instructions it leaves
//   in flight: the K-tile body is instantiated per position in the chunk with the exact counts for that position.
// LDS images are [rows][64 k] bf16 with 128-byte rows, written lane-linearly by the DMA; the XOR swizzle that makes the
// ds_read_b128 fragment reads conflict-free (16-byte slot = k-chunk ^ (row & 7)) is applied to the per-lane SOURCE
// address.  For the halo the "row" is the halo pixel slot (row pitch 36), so a tap is a constant shift of the row -- and of
// its swizzle.  Zero padding lives entirely in the halo load (pixels outside the image read a zero page): no per-tap masks.
//
// Eligible: stride-1 phases with 3..16 taps forming a rows x columns grid whose halo fits (<= 10 rows x 34 columns),
// C_src % 64 == 0, C_dst % 256 == 0, plain bf16 input, bf16 output (+ stats / bias / accumulate).  1x1 layers stay on
// tapconv4 (nothing to reuse).  Measured: +3..8 % over tapconv4 on the 3x3 layers of the rv-av2 model.
#include <stdlib.h>

#include "common.h"
#include "tapconv.h"

namespace {

constexpr int kTC = 32;                  // tile columns
constexpr int kTR = 8;                   // tile rows
constexpr int kBK = 64;                  // K tile (channels of one tap)
constexpr int kPitch = 36;               // halo row pitch in pixel slots (>= 34 used columns).  36 = 4 (mod 8): the swizzle of
                                         // row r+1 is the swizzle of row r with the two 64-byte halves of the line swapped,
                                         // so ONE address computation per K tile serves all four rows a wave reads
constexpr int kHaloRows = 10;
constexpr int kHaloInstr = kHaloRows * kPitch / 8;  // 45 DMA instructions (8 pixel slots x 128 B each) per halo
constexpr int kHaloBytes = kHaloInstr * 1024;       // 46080
constexpr int kPiece = 128 * 128;                   // one weight piece: 128 rows x 64 k
constexpr int kRing = 2 * kHaloBytes;               // weight ring behind the two halo buffers
constexpr int kScratch = kRing + 4 * kPiece;        // 1 KB target of the dummy halo instructions
constexpr int kTab = kScratch + 1024;               // tap table (prologue hand-off)
constexpr int kLds = kTab + 32 * 4;

__device__ __attribute__((aligned(256))) uint32_t g_zero_page5[64];

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// kBN: channels per workgroup, 256 or 128 (narrow layers: one weight piece per K tile, two phases).  EPI: which epilogue the
// launch carries -- template parameters because the BatchNorm-sum epilogue's 16 prefetch registers per pass and per-channel
// constants, merely PRESENT in the code behind a run-time flag, cost every launch 154 spilled registers per tile (11 without:
// forward launches and plain backward-data launches run the lean instance); the accumulate instance can then afford the
// same prefetch for the old values it adds to.
template <int kBN, int EPI>  // EPI: 0 plain store (+ statistics / bias), 1 BatchNorm-sum epilogue (RV_OUT_BNB), 2 accumulate (RV_OUT_ACCUM)
__global__ __launch_bounds__(512, 2) void tapconv5_kernel(const TapConvArgs a) {
    constexpr int NJ = kBN / 64;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int l15 = lane & 15, lg = lane >> 4;

    // XCD-aware block order (see tapconv3.hip).  PERSISTENT workgroups: the launch has (a multiple of 8, at most the tile
    // count) workgroups, each walks the virtual block indices vb = blockIdx.x, blockIdx.x + gridDim.x, ... -- same XCD every
    // time (gridDim.x % 8 == 0) -- so a CU pays the workgroup dispatch (LDS allocation, wave start, kernel-argument loads)
    // once per launch instead of once per tile (16 tiles per CU on the 512-channel layers).
    const int gy = a.n_tiles;
    // (Measured and dropped: letting one workgroup take the two channel tiles of a pixel tile back to back, and dealing the
    //  channel tiles to different XCDs -- neither moved the step time.  The doubled HBM-side traffic that prompted both turned
    //  out to be scratch traffic of spilled registers, see the BNB template parameter.)
    const int xcd = blockIdx.x & 7, wslot = blockIdx.x >> 3, nslots = gridDim.x >> 3;  // (gridDim.x % 8 == 0)
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
    const int HR = kTR + a.tt.rows - 1;                      // halo rows (<= 10)

    const bf16_t* src_img = a.src + ((int64_t)n * a.H * a.W_src) * a.ld_src;
    // Tap shifts (halo pixel-slot shift of each tap): the taps of a phase form a grid (rows x columns, enumerated row-major,
    // ascending for the gather form, descending for the scatter form -- checked by the host), so the shift of the next tap
    // is the current one plus a column or a row step: three scalars, no table look-up (and no compiler-placed lgkmcnt wait)
    // inside the loop.  The first T threads fetch their tap from the kernel arguments once and hand it over through LDS.
    int* tap_tab = (int*)(smem + kTab);
    if (tid < 16) tap_tab[tid] = tid < T ? (a.tt.dh[ph][tid] - a.tt.dh_min) * kPitch + (a.tt.dw[ph][tid] - a.tt.dw_min[ph]) : 0;
    __syncthreads();
    const int ncol = T / a.tt.rows;  // taps per kernel row
    const int sh0 = __builtin_amdgcn_readfirstlane(tap_tab[0]);
    const int sh_dcol = ncol > 1 ? __builtin_amdgcn_readfirstlane(tap_tab[1]) - sh0 : 0;
    const int sh_drow = ncol < T ? __builtin_amdgcn_readfirstlane(tap_tab[ncol]) - __builtin_amdgcn_readfirstlane(tap_tab[ncol - 1]) : 0;
    const int row_base = h0 + a.tt.dh_min, col_base = m0 + a.tt.dw_min[ph];

    // ---- DMA maps ---------------------------------------------------------------------------------------------------
    // Every DMA instruction moves 8 image rows x 128 B; lane = (row-in-8 = lane >> 3, 16-byte slot = lane & 7) and the slot
    // holds logical k-chunk slot ^ (row & 7): the same constant for every instruction of a lane.
    const int s_row = lane >> 3, s_slot = lane & 7;
    const int kq8 = (s_slot ^ s_row) * 8;
    const bf16_t* zero = (const bf16_t*)g_zero_page5 + s_slot * 8;
    // Weights: a piece is 128 rows (rho = wave column * 32 + channel) = 16 DMA instructions; ALL eight waves issue one per
    // phase: instruction e = half * 8 + wave of the piece covers rho = e*8 + s_row, i.e. wave column e >> 2, channels
    // (e & 3) * 8 + s_row.  The per-lane part of the address is one constant; tap image, chunk, channel half and `half`
    // are wave-uniform.
    const int64_t w_img = (int64_t)a.C_dst * a.C_src;
    const bf16_t* w_ph = a.w + (int64_t)a.tt.w_first[ph] * w_img;
    // (kBN = 128: a piece is the whole K tile -- wave column e >> 2 owns 32 channels, no channel halves)
    const int b_voff = (n0 + (wave >> 2) * (kBN / 4) + (wave & 3) * 8 + s_row) * a.C_src + kq8;
    const int b_halfstep = (kBN / 2) * a.C_src, b_nqstep = kBN == 256 ? 32 * a.C_src : 0;

    // B piece number j (0, 1, 2, ...) = (K tile j >> 1, channel half j & 1) lives in ring slot j & 3.
    int bq = 0, bt = 0;  // K tile / tap of the piece pair being issued
    int b_so = 0;        // element offset of its (tap image, chunk) in the packed weight -- kept scalar
    auto stage_b = [&](int j, int half) {  // this wave's instruction of one half of piece j
        const int so = __builtin_amdgcn_readfirstlane(b_so + (kBN == 256 ? (j & 1) * b_nqstep : 0) + half * b_halfstep);
        const bf16_t* p = w_ph + so + b_voff;
        __builtin_amdgcn_global_load_lds((glb_void_t*)p, (lds_void_t*)(smem + kRing + (j & 3) * kPiece + (half * 8 + wave) * 1024), 16, 0, 0);
    };
    auto advance_b = [&]() {  // after both pieces of a K tile have been issued
        const bool go = bq + 1 < nkt;  // pieces past the last K tile re-fetch it (never read; keeps the wait counts uniform)
        const bool wrap = bt + 1 == T;
        bq += go ? 1 : 0;
        b_so += go ? (wrap ? kBK - (T - 1) * (int)w_img : (int)w_img) : 0;
        bt = go ? (wrap ? 0 : bt + 1) : bt;
    };
    // Halo: instruction q (0..44) covers pixel slots q*8 + s_row of the [10 rows][36 slots] image; wave w issues
    // q = w, w + 8, ..., w + 40: six instructions per chunk (those with q >= 45 are dummies: zero page -> a scratch KB, so that
    // every wave issues the same number and the counted waits are the same code for all).  The source offset of a slot is
    // computed ONCE per tile (`hoff[i]`, -1 = zero page: outside the image, pad column, dummy); per chunk only the channel
    // offset is added.
    int hoff[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int q = wave + 8 * i;
        const int p = q * 8 + s_row;
        const int hr = (p * 1821) >> 16, hc = p - hr * kPitch;  // 1821 = ceil(65536 / 36): exact for p < 2^12
        const int row = row_base + hr, col = col_base + hc;
        const int bad = (q >= kHaloInstr) | (hc >= HW) | (hr >= HR) | (row < 0) | (row >= a.H) | (col < 0) | (col >= a.W_src);
        hoff[i] = ((row * a.W_src + col) * a.ld_src + kq8) | -bad;
    }
    auto stage_halo = [&](int buf, int i, int kc) {  // i: compile-time index 0..5
        const int q = wave + 8 * i;
        const bf16_t* src = hoff[i] >= 0 ? src_img + (hoff[i] + kc * kBK) : zero;
        const int dst = q < kHaloInstr ? buf * kHaloBytes + q * 1024 : kScratch;
        __builtin_amdgcn_global_load_lds((glb_void_t*)src, (lds_void_t*)(smem + dst), 16, 0, 0);
    };

    // ---- fragment reads ---------------------------------------------------------------------------------------------
    // A: output pixel (row r, column c) under a tap = halo slot r*36 + c + shift(tap), stored with 16-byte chunks XOR-ed
    // by (slot & 7).  Wave wr owns rows 4*wr .. 4*wr+3; fragment i (0..7): row 4*wr + i/2, columns (i&1)*16 + l15.
    // Row r+1 sits 36 slots = 4608 bytes further, and its swizzle differs by exactly the 64-byte half: its first k-half is
    // read at (base ^ 64) + 4608.  So a K tile needs ONE per-lane base (`a_b0`; a_b1 = a_b0 ^ 64) and immediates.
    // B: as in tapconv4 (piece row = wc*32 + channel).
    const int p_lane = (4 * wr) * kPitch + l15;
    const int b_rd = (wc * 32 + l15) * 128 + ((lg ^ (l15 & 7)) * 16);
    bf16x8 fa[4][2], fb0[2][2], fb1[2][2];
    int a_b0 = 0, a_b1 = 0;
    auto addr_a = [&](int halo_byte, int shift) {
        const int p = p_lane + shift;
        a_b0 = halo_byte + p * 128 + (((lg ^ p) & 7) * 16);
        a_b1 = a_b0 ^ 64;
    };
    auto read_a = [&](int mq) {
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int cc = 0; cc < 2; ++cc)
#pragma unroll
                for (int ks = 0; ks < 2; ++ks)
                    fa[rr * 2 + cc][ks] = *(const bf16x8*)(smem + ((((2 * mq + rr + ks) & 1) ? a_b1 : a_b0) + (2 * mq + rr) * (kPitch * 128) + cc * 2048));
    };
    auto read_b = [&](bf16x8 (&fb)[2][2], int j) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) fb[jj][ks] = *(const bf16x8*)(smem + kRing + (j & 3) * kPiece + jj * 2048 + (b_rd ^ (ks * 64)));
    };

    f32x4 acc[8][NJ];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#define RV_MFMA_HALF(MQ, NQ, FB, KS)                                                                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                 \
        acc[(MQ) * 4 + i][(NQ) * 2 + jj] =                                                                         \
            RV_MFMA_16x16x32(fa[i][KS], FB[jj][KS], acc[(MQ) * 4 + i][(NQ) * 2 + jj], 0, 0, 0);
// One phase: barrier | 16 MFMAs, nothing else | barrier.
#define RV_PHASE_COMPUTE(MQ, NQ, FB)                       \
    __builtin_amdgcn_s_barrier();                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_setprio(1);                         \
    RV_MFMA_HALF(MQ, NQ, FB, 0)                            \
    RV_MFMA_HALF(MQ, NQ, FB, 1)                            \
    __builtin_amdgcn_s_setprio(0);                         \
    __builtin_amdgcn_sched_barrier(0);                     \
    __builtin_amdgcn_s_barrier();
#define RV_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory")

    if constexpr (kBN == 256) {
    // ---- prologue: halo of chunk 0, weight pieces 0, 1, 2 and the first half of 3 --------------------------------------
#pragma unroll
    for (int i = 0; i < 6; ++i) stage_halo(0, i, 0);
    stage_b(0, 0);
    stage_b(0, 1);
    stage_b(1, 0);
    stage_b(1, 1);
    advance_b();
    stage_b(2, 0);
    stage_b(2, 1);
    stage_b(3, 0);
    RV_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // the second half of the workgroup runs one barrier behind the first

    int kt = 0, kc = 0;     // K tile, chunk
    int sh = sh0, ix = 0;   // slot shift / grid column of the tap of K tile kt
    // One K tile: four phases of [DMA instructions, fragment reads, counted wait] barrier | 16 MFMAs | barrier, the same
    // branch-free code for all eight waves.  Weight piece j0+4 is issued in phases 1-2, piece j0+5 in phases 3 and 0 (of the
    // next K tile): one instruction per wave and phase; a piece is first read six to seven phases after its first
    // instruction.  U = position of the K tile in its chunk: U = 0 carries halo instructions 0-3 of the NEXT chunk (one per
    // phase), U = 1 numbers 4, 5 (phases 0, 1), U >= 2 none.  W0 / W3 = the counted waits: weight instructions younger than
    // the piece that must have landed (4 at phase 0, 5 at phase 3) plus the halo instructions issued among them -- vmcnt
    // retires in order, so the count has to be exact for each position (a smaller one is safe but waits for more).
#define RV_KTILE(U, W0, W3)                                                                                        \
    {                                                                                                              \
        const int j0 = 2 * kt;                                                                                     \
        const int hbuf = (kc + 1) & 1;                                                                             \
        stage_b(j0 + 3, 1);                                                                                        \
        advance_b();                                                                                               \
        if constexpr ((U) == 0) stage_halo(hbuf, 0, kc + 1);                                                       \
        if constexpr ((U) == 1) stage_halo(hbuf, 4, kc + 1);                                                       \
        addr_a((kc & 1) * kHaloBytes, sh);                                                                         \
        read_b(fb0, j0);                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        read_a(0);                                                                                                 \
        RV_WAIT_VM(W0);                                                                                            \
        RV_PHASE_COMPUTE(0, 0, fb0);                                                                               \
        stage_b(j0 + 4, 0);                                                                                        \
        if constexpr ((U) == 0) stage_halo(hbuf, 1, kc + 1);                                                       \
        if constexpr ((U) == 1) stage_halo(hbuf, 5, kc + 1);                                                       \
        read_b(fb1, j0 + 1);                                                                                       \
        RV_PHASE_COMPUTE(0, 1, fb1);                                                                               \
        stage_b(j0 + 4, 1);                                                                                        \
        if constexpr ((U) == 0) stage_halo(hbuf, 2, kc + 1);                                                       \
        read_a(1);                                                                                                 \
        RV_PHASE_COMPUTE(1, 1, fb1);                                                                               \
        stage_b(j0 + 5, 0);                                                                                        \
        if constexpr ((U) == 0) stage_halo(hbuf, 3, kc + 1);                                                       \
        RV_WAIT_VM(W3);                                                                                            \
        RV_PHASE_COMPUTE(1, 0, fb0);                                                                               \
        ++kt;                                                                                                      \
        const bool wrap = ix + 1 == ncol;                                                                          \
        sh += wrap ? sh_drow : sh_dcol;                                                                            \
        ix = wrap ? 0 : ix + 1;                                                                                    \
    }
    // chunks with a successor: K tiles 0 and 1 carry the successor's halo; it has landed by the end of K tile 2 (the wait of
    // its phase 3 leaves only the five youngest instructions in flight) -- hence T >= 3.
    for (; kc + 1 < nkc; ++kc) {
        RV_KTILE(0, 5, 9)
        RV_KTILE(1, 8, 8)
        RV_KTILE(2, 5, 5)
        for (int t = 3; t < T; ++t) RV_KTILE(3, 4, 5)
        sh = sh0;
        ix = 0;
    }
    // last chunk: no halo to load
    for (int t = 0; t < T; ++t) RV_KTILE(3, 4, 5)
#undef RV_KTILE
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    } else {
    // ---- kBN = 128: piece j IS K tile j.  Two phases per K tile (pixel halves MQ = 0, 1 against the one weight fragment
    // set), one weight instruction per wave and phase as above: phase 0 issues the second half of piece kt+3, phase 1 the first
    // half of piece kt+4 (into the slot of piece kt, read in phase 0).  The next chunk's halo goes out one instruction per
    // phase during K tiles U = 0, 1, 2 of the current chunk.  One counted wait per K tile, in phase 1 (a wave's wait + the
    // barrier that follows make piece kt+1 visible to the reads of the next K tile's phase 0): five younger weight instructions
    // plus the halo instructions issued among them -- those of positions U-2, U-1, U: 2, 4, 6, 4, 2 at U = 0..4, none after
    // (hence T >= 6: the last K tile of a chunk waits with 5, which also retires the next chunk's halo).
#pragma unroll
    for (int i = 0; i < 6; ++i) stage_halo(0, i, 0);
    stage_b(0, 0);
    stage_b(0, 1);
    advance_b();
    stage_b(1, 0);
    stage_b(1, 1);
    advance_b();
    stage_b(2, 0);
    stage_b(2, 1);
    advance_b();
    stage_b(3, 0);
    RV_WAIT_VM(0);
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();  // the second half of the workgroup runs one barrier behind the first

    int kt = 0, kc = 0;
    int sh = sh0, ix = 0;
#define RV_KTILE128(U, W)                                                                                          \
    {                                                                                                              \
        const int hbuf = (kc + 1) & 1;                                                                             \
        stage_b(kt + 3, 1);                                                                                        \
        advance_b();                                                                                               \
        if constexpr ((U) <= 2) stage_halo(hbuf, 2 * (U), kc + 1);                                                 \
        addr_a((kc & 1) * kHaloBytes, sh);                                                                         \
        read_b(fb0, kt);                                                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                                         \
        read_a(0);                                                                                                 \
        RV_PHASE_COMPUTE(0, 0, fb0);                                                                               \
        stage_b(kt + 4, 0);                                                                                        \
        if constexpr ((U) <= 2) stage_halo(hbuf, 2 * (U) + 1, kc + 1);                                             \
        read_a(1);                                                                                                 \
        RV_WAIT_VM(W);                                                                                             \
        RV_PHASE_COMPUTE(1, 0, fb0);                                                                               \
        ++kt;                                                                                                      \
        const bool wrap = ix + 1 == ncol;                                                                          \
        sh += wrap ? sh_drow : sh_dcol;                                                                            \
        ix = wrap ? 0 : ix + 1;                                                                                    \
    }
    for (; kc + 1 < nkc; ++kc) {
        RV_KTILE128(0, 7)
        RV_KTILE128(1, 9)
        RV_KTILE128(2, 11)
        RV_KTILE128(3, 9)
        RV_KTILE128(4, 7)
        for (int t = 5; t < T; ++t) RV_KTILE128(5, 5)
        sh = sh0;
        ix = 0;
    }
    for (int t = 0; t < T; ++t) RV_KTILE128(5, 5)  // last chunk: no halo to load
#undef RV_KTILE128
    if (wr == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    }
#undef RV_MFMA_HALF
#undef RV_PHASE_COMPUTE
#undef RV_WAIT_VM

    // ------------------------------------ epilogue --------------------------------------------
    // acc[i][j][r]: tile row 4*wr + i/2, column (i&1)*16 + lg*4 + r, channel n0 + wc*64 + j*16 + l15
    constexpr int WN = kBN / 4;
    const int Wm = a.W_dst / a.phases;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const bool row_ok = h0 + 4 * wr + (i >> 1) < a.H;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + (i & 1) * 16 + lg * 4 + r;
            if (m >= Wm || !row_ok) {
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j][r] = 0.f;
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
                prow[c] = s;
                prow[a.C_dst + c] = q;
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
    constexpr int kEpi = kBN + 8;
    bf16_t* epi = (bf16_t*)smem;  // [8 rows * 32 cols][kEpi]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int pm = (4 * wr + (i >> 1)) * kTC + (i & 1) * 16 + lg * 4 + r;
                const int pc = wc * WN + j * 16 + l15;
                epi[pm * kEpi + pc] = f2bf(acc[i][j][r]);
            }
    constexpr int kChunks = kBN / 8;
    constexpr bool accum = EPI == 2;
    // RV_OUT_BNB: this launch writes dOut of a BatchNorm(+ReLU) layer -- its backward needs sum(g) and sum(g * xhat) per channel
    // with g = dOut * [scale*y+shift > 0], xhat = (y - mean) * invstd.  A thread keeps ONE 8-channel chunk through the store loop
    // (512 threads = 16 pixels x 32 chunks per pass), so the sums are formed here, from the bf16 values being stored, with one
    // extra 16-byte read of y per chunk: the separate reduce pass over (dOut, y) disappears (bnbwd.hip: bn_bwd_reduce_kernel).
    constexpr bool bnb = EPI == 1;
    static_assert(EPI >= 0 && EPI <= 2, "epilogue kinds: plain, BatchNorm sums, accumulate");
    float bsc[8], bsh[8], bmu[8], bis[8], s0[8], s1[8];
    if (bnb) {
        const int c = n0 + (tid & (kChunks - 1)) * 8;  // (512 % kChunks == 0: the chunk of a thread is the same in every pass)
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
    constexpr int kPasses = kTR * kTC * kChunks / 512;  // 16
    u32x4 yv[kPasses];
    if (bnb) {  // (before the barrier that publishes the staged tile) all of this thread's y chunks in flight at once (the staged tile is being read meanwhile): one HBM latency, not sixteen
#pragma unroll
        for (int it = 0; it < kPasses; ++it) {
            const int q = tid + it * 512, pm = q / kChunks, c8 = q - pm * kChunks;
            const int rr = pm / kTC, mm = pm - rr * kTC;
            const int m = m0 + mm, c = n0 + c8 * 8, hh = h0 + rr;
            yv[it] = u32x4{0u, 0u, 0u, 0u};
            if (m < Wm && hh < a.H) yv[it] = *(const u32x4*)(a.bnb_y + (((int64_t)(n * a.H + hh) * a.W_dst) + (a.phases * m + ph)) * a.ld_bnb_y + c);
        }
    }
    if (accum) {  // the old values of this thread's sixteen chunks likewise: read inside the store loop each one waited out its
                  // own round trip between two stores (+17 us per tile on the 256-channel layers)
#pragma unroll
        for (int it = 0; it < kPasses; ++it) {
            const int q = tid + it * 512, pm = q / kChunks, c8 = q - pm * kChunks;
            const int rr = pm / kTC, mm = pm - rr * kTC;
            const int m = m0 + mm, c = n0 + c8 * 8, hh = h0 + rr;
            u32x4 o = u32x4{0u, 0u, 0u, 0u};
            const int64_t px = ((int64_t)(n * a.H + hh) * a.W_dst) + (a.phases * m + ph);
            if (m < Wm && hh < a.H) o = *(const u32x4*)(a.res + px * a.ld_res + c);
            yv[it] = o;
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
        const int64_t px = ((int64_t)(n * a.H + hh) * a.W_dst) + (a.phases * m + ph);
        bf16_t* p = (bf16_t*)a.dst + px * a.ld_dst + c;
        if (accum) {
            const u32x4 o = yv[it];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = pack_bf2(bf_lo(v[j]) + bf_lo(o[j]), bf_hi(v[j]) + bf_hi(o[j]));
            if (a.flags & RV_OUT_RES_RELU) {
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = pack_bf2(fmaxf(bf_lo(v[j]), 0.f), fmaxf(bf_hi(v[j]), 0.f));
            }
        }
        *(u32x4*)p = v;
        if (bnb) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float y0 = bf_lo(yv[it][j]), y1 = bf_hi(yv[it][j]);
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
    if (bnb) {
        // lanes kChunks apart hold the same chunk; then the eight waves through LDS (the staged tile is dead now)
        __syncthreads();
        float* red = (float*)smem;  // [8 waves][kChunks][16]
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
                red[(wave * kChunks + lane) * 16 + j] = s0[j];
                red[(wave * kChunks + lane) * 16 + 8 + j] = s1[j];
            }
        }
        __syncthreads();
        const int chunk = tid >> 4, jj = tid & 15;  // kChunks x 16 values
        if (chunk < kChunks) {
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) sum += red[(w * kChunks + chunk) * 16 + jj];
            a.bnb_partial[((int64_t)tile * 2 + (jj >> 3)) * a.C_dst + n0 + chunk * 8 + (jj & 7)] = sum;
        }
    }
    __syncthreads();  // the staged output / BatchNorm sums of this tile are dead before the next tile's loads land in LDS
    }  // persistent tile loop
}

}  // namespace


// returns false when the layer is not eligible (caller falls back to tapconv4 / tapconv3 / ...)
bool rv_tapconv5_plan(TapConvArgs* a, int* tiles, size_t* lds, int* bn) {
    if (a->step != 1) return false;
    if (a->flags & (RV_IN_AFFINE | RV_IN_RELU | RV_OUT_F32)) return false;  // the DMA path has no register prologue
    if ((a->flags & RV_OUT_BNB) && (a->flags & RV_OUT_ACCUM)) return false;  // (sums over an ACCUMULATED gradient: not formed in an epilogue)
    if (a->C_src % kBK != 0 || a->C_dst % 128 != 0) return false;
    const int kBN = a->C_dst % 256 == 0 ? 256 : 128;  // narrow layers: 128-channel tiles, one weight piece per K tile
    const int min_taps = kBN == 256 ? 3 : 6;          // (K tiles the next chunk's halo needs to land, see the kernel)
    *bn = kBN;
    const int wm_total = a->W_dst / a->phases;
    if (wm_total < kTC || a->H < kTR) return false;
    if (kTR + a->tt.rows - 1 > kHaloRows) return false;
    for (int r = 0; r < a->phases; ++r) {
        if (a->tt.ntaps[r] < min_taps || a->tt.ntaps[r] > 16) return false;
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
    if ((int64_t)a->total_tiles * a->n_tiles < ((a->sel & RV_SEL_SMALL_GRIDS) ? 1 : rv_cu_count())) return false;  // too few tiles to fill the chip
    *tiles = a->total_tiles;  // stats rows = 2 * tiles
    *lds = (size_t)kLds;
    const size_t epi = (size_t)kTR * kTC * (kBN + 8) * sizeof(bf16_t);
    if (*lds < epi) *lds = epi;
    return true;
}

int rv_tapconv5_launch(const TapConvArgs& a, size_t lds, int bn, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)tapconv5_kernel<256, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)tapconv5_kernel<128, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)tapconv5_kernel<256, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)tapconv5_kernel<128, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)tapconv5_kernel<256, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)tapconv5_kernel<128, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    int grid = 8 * a.tiles_per_xcd * a.n_tiles;
    if (grid > rv_cu_count()) grid = rv_persistent_grid();  // one workgroup per CU (154 KB of LDS each)
    const int epi = (a.flags & RV_OUT_BNB) ? 1 : ((a.flags & RV_OUT_ACCUM) ? 2 : 0);
#define RV_T5_LAUNCH(BN_, EPI_) hipLaunchKernelGGL((tapconv5_kernel<BN_, EPI_>), dim3(grid), dim3(512), lds, stream, a)
    if (bn == 256) {
        if (epi == 1) RV_T5_LAUNCH(256, 1);
        else if (epi == 2) RV_T5_LAUNCH(256, 2);
        else RV_T5_LAUNCH(256, 0);
    } else {
        if (epi == 1) RV_T5_LAUNCH(128, 1);
        else if (epi == 2) RV_T5_LAUNCH(128, 2);
        else RV_T5_LAUNCH(128, 0);
    }
#undef RV_T5_LAUNCH
    RV_CHECK_LAUNCH("tapconv5_kernel");
    return 0;
}

// tapconv3.hip -- third-generation tap-conv kernel: 8 wavefronts, 4-row x 64-column x 128-channel tiles,
// three taps per weight stage.
//
// Round-1 experiments on tapconv2 (512->512 3x3, 4x64x2048): 673 TFLOP/s; 910 with the weight loads
// removed, 1040 with all global loads removed => the per-CU load path (weights re-fetched for every 128
// output pixels) and the short 32-MFMA barrier interval are the limiters.  This kernel
//   * doubles the pixels per block (256 = 4 rows x 64 columns, halo 6 x 66 = 1.55x) so every weight byte
//     fetched from L2 feeds twice the MFMAs;
//   * stages the weights of up to THREE taps (one kernel row) at once, double-buffered in LDS (2 x 48 KB next to
//     the 60 KB halo tile): 96 MFMAs per wave between barriers, a whole group of compute to cover the loads;
//   * keeps tapconv2's dense, conflict-free LDS images and 8-pixel x 8-k-group staging map.
// One workgroup (512 threads, ~157 KB LDS) per CU, two waves per SIMD.
#include "common.h"
#include "tapconv.h"

namespace {

constexpr int kTC = 64;   // tile columns (m positions)
constexpr int kTR = 4;    // tile rows
constexpr int kBN = 128;  // tile channels
constexpr int kNa2 = 7;   // max A chunks per thread: 6 rows x 66 px x 8 groups / 512 threads
constexpr int kTG = 3;    // taps per weight stage

template <int KS>  // MFMA K-steps per chunk: chunk = 32*KS channels
__global__ __launch_bounds__(512, 2) void tapconv3_kernel(const TapConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    constexpr int G = 4 * KS;        // 16-byte k-groups per chunk
    constexpr int BK = 32 * KS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;  // wm = tile row (0..3), wn = channel half
    const int l15 = lane & 15, lg = lane >> 4;

    // XCD-aware block order: workgroups are dealt round-robin over the 8 XCDs (private L2 each), so block b runs on
    // XCD b % 8.  Each XCD gets a contiguous range of pixel tiles and walks the channel tiles of one pixel tile
    // back to back: the gy blocks that read the same input halo run at the same time on the same L2.
    const int gy = a.n_tiles;
    const int xcd = blockIdx.x & 7, xslot = blockIdx.x >> 3;
    const int tile = xcd * a.tiles_per_xcd + xslot / gy;
    if (tile >= a.total_tiles) return;
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
    const int px_per_iter = 512 / G;  // 64 (KS=2) or 128 (KS=1)
    const int n_px = HR * Wt;
    // per-slot halo coordinates packed as (valid << 31) | (in_tile << 30) | (hr << 8) | cc  -- offsets are recomputed
    // from them at load / write time (two multiply-adds) instead of living in 2 x 7 registers across the K loop
    uint32_t slot[kNa2];
    const int row_base = h0 + a.tt.dh_min, col_base = m0 + a.tt.dw_min[ph];
#pragma unroll
    for (int i = 0; i < kNa2; ++i) {
        const int p = i * px_per_iter + s_hi * 8 + s_px;
        slot[i] = 0u;
        if (p < n_px) {
            const int hr = p / Wt, cc = p - hr * Wt;
            const int hs = row_base + hr, ws = col_base + cc;
            const bool ok = hs >= 0 && hs < a.H && ws >= 0 && ws < a.W_src;
            slot[i] = (ok ? 0x80000000u : 0u) | 0x40000000u | ((uint32_t)hr << 8) | (uint32_t)cc;
        }
    }
    auto slot_src = [&](uint32_t sl) { return ((row_base + (int)((sl >> 8) & 0xff)) * a.W_src + col_base + (int)(sl & 0xff)) * a.ld_src + s_g * 8; };
    auto slot_lds = [&](uint32_t sl) { return (((int)((sl >> 8) & 0xff) * G + s_g) * Wtp + (int)(sl & 0xff)) * 8; };
    // B: 128 n x G groups = 128*G chunks; thread handles n = (q>>... ) with the same 8x8 wave map
    constexpr int kBTap = G * kBN * 8;          // elements of one tap image in LDS
    const int64_t w_img = (int64_t)a.C_dst * a.C_src;
    const int w_base_idx = a.tt.w_first[ph];

    u32x4 a_reg[kNa2];
    const bool affine = a.flags & RV_IN_AFFINE, relu = a.flags & RV_IN_RELU;

    auto load_a = [&](int kc) {
#pragma unroll
        for (int i = 0; i < kNa2; ++i) {
            u32x4 v = {0u, 0u, 0u, 0u};
            if (slot[i] & 0x80000000u) v = *(const u32x4*)(src_img + slot_src(slot[i]) + kc * BK);
            a_reg[i] = v;
        }
    };
    auto write_a = [&](int abuf, int kc) {
        bf16_t* base = ldsA + abuf * a_buf_elems;
        float sc[8], sh[8];
        if (affine) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                sc[j] = a.in_scale[kc * BK + s_g * 8 + j];
                sh[j] = a.in_shift[kc * BK + s_g * 8 + j];
            }
        }
#pragma unroll
        for (int i = 0; i < kNa2; ++i) {
            if (!(slot[i] & 0x40000000u)) continue;
            pin_here(a_reg[i]);
            u32x4 v = a_reg[i];
            if ((affine || relu) && (slot[i] & 0x80000000u)) {
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
            *(u32x4*)(base + slot_lds(slot[i])) = v;
        }
    };
    constexpr int kNb = (kTG * kBN * G) / 512;  // 6 (KS=2) or 3 (KS=1): all taps of a stage
    u32x4 b_reg[kNb];
    auto load_b = [&](int kc, int t0, int) {  // taps t0 .. t0+kTG-1 (those < T); 8 rows x 8 k-groups per wave-instruction
#pragma unroll
        for (int i = 0; i < kNb; ++i) {
            const int q = i * px_per_iter + s_hi * 8 + s_px;  // 0 .. kTG*128-1
            const int tt = q >> 7, nn = q & 127;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (t0 + tt < T && n0 + nn < a.C_dst)
                v = *(const u32x4*)(a.w + (int64_t)(w_base_idx + t0 + tt) * w_img + (int64_t)(n0 + nn) * a.C_src + kc * BK + s_g * 8);
            b_reg[i] = v;
        }
    };
    auto write_b = [&](int buf) {
        bf16_t* base = ldsB + buf * (kTG * kBTap);
#pragma unroll
        for (int i = 0; i < kNb; ++i) {
            const int q = i * px_per_iter + s_hi * 8 + s_px;
            const int tt = q >> 7, nn = q & 127;
            pin_here(b_reg[i]);
            *(u32x4*)(base + tt * kBTap + (s_g * kBN + nn) * 8) = b_reg[i];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = (T > 0) ? a.C_src / BK : 0;
    const int ngroups = (T + kTG - 1) / kTG;
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
        // Prefetch loads are issued UNCONDITIONALLY (the final ones re-fetch the current tile, harmlessly): a conditional
        // load leaves hipcc with a register merge of "loaded" and "not loaded" values at the join, and the v_mov it emits
        // there needs the data => s_waitcnt vmcnt(0) right after the loads, in front of the MFMAs they should overlap.
        load_a(next_chunk ? kc + 1 : kc);
        for (int gi = 0; gi < ngroups; ++gi) {
            const bool last_g = (gi == ngroups - 1);
            const bool has_next = !last_g || next_chunk;
            load_b(has_next ? (last_g ? kc + 1 : kc) : kc, has_next ? (last_g ? 0 : (gi + 1) * kTG) : gi * kTG, buf ^ 1);
            const int t0 = gi * kTG;
            const int tn = (T - t0) < kTG ? (T - t0) : kTG;
            for (int tt = 0; tt < tn; ++tt) {
                const int t = t0 + tt;
                const int ti = tap_tab[t];
                const int hr = wm + (ti >> 16), c0 = ti & 0xffff;
                const bf16_t* pa = ldsA + abuf * a_buf_elems + ((hr * G + lg) * Wtp + c0 + l15) * 8;
                const bf16_t* pb = ldsB + buf * (kTG * kBTap) + tt * kBTap + (lg * kBN + wn * 64 + l15) * 8;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
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
            if (last_g && next_chunk) {
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
        float* prow = a.stats + ((int64_t)(tile * kTR + wm) * 2) * a.C_dst;
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
    for (int q = tid; q < kTR * kTC * kChunks; q += 512) {
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
bool rv_tapconv3_plan(TapConvArgs* a, int* grid_x, int* grid_y, size_t* lds, int* ks) {
    if (a->step != 1) return false;
    const int wm_total = a->W_dst / a->phases;
    if (wm_total < kTC || a->C_dst < 64 || a->H < kTR) return false;
    for (int r = 0; r < a->phases; ++r)
        if (a->tt.ntaps[r] < 2) return false;  // single-tap (1x1) phases: the smaller tapconv2 tile measured faster
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
        if (HR * a->tt.w_tile[r] > kNa2 * (512 / G)) return false;
        if (a->tt.ntaps[r] != 1) any_single_tap = false;
    }
    // the A tile is double-buffered by phases with a single tap; size the region for it when any phase needs it
    bool some_single = false;
    for (int r = 0; r < a->phases; ++r) some_single |= (a->tt.ntaps[r] == 1);
    (void)any_single_tap;
    a->lds_a_elems = a_max * (some_single ? 2 : 1);
    size_t bytes = (size_t)(a->lds_a_elems + 2 * kTG * G * kBN * 8) * sizeof(bf16_t);
    const size_t epi = (size_t)kTR * kTC * (kBN + 8) * sizeof(bf16_t);
    if (bytes < epi) bytes = epi;
    bytes = (bytes + 15) & ~(size_t)15;
    a->lds_tab_offset = (int32_t)bytes;
    bytes += 128;  // per-tap offset table (<= 24 ints)
    if (bytes > 160 * 1024) return false;  // one workgroup per CU
    a->m_tiles = rv_ceil_div(wm_total, kTC);
    a->h_tiles = rv_ceil_div(a->H, kTR);
    a->total_tiles = a->m_tiles * a->h_tiles * a->N * a->phases;
    a->n_tiles = rv_ceil_div(a->C_dst, kBN);
    a->tiles_per_xcd = rv_ceil_div(a->total_tiles, 8);
    *grid_x = a->total_tiles;  // (stats rows = 4 * grid_x); the launch uses 8 * tiles_per_xcd * n_tiles blocks
    *grid_y = a->n_tiles;
    if ((int64_t)*grid_x * *grid_y < 512) return false;  // too few 256-pixel tiles to fill 256 CUs: use tapconv2
    *lds = bytes;
    *ks = KS;
    return true;
}

int rv_tapconv3_launch(const TapConvArgs& a, int grid_x, int grid_y, size_t lds, int ks, hipStream_t stream) {
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)tapconv3_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute((const void*)tapconv3_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    if (ks == 2)
        hipLaunchKernelGGL((tapconv3_kernel<2>), dim3(8 * a.tiles_per_xcd * a.n_tiles), dim3(512), lds, stream, a);
    else
        hipLaunchKernelGGL((tapconv3_kernel<1>), dim3(8 * a.tiles_per_xcd * a.n_tiles), dim3(512), lds, stream, a);
    RV_CHECK_LAUNCH("tapconv3_kernel");
    return 0;
}

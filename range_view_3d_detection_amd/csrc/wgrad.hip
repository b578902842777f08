// wgrad.hip -- weight gradient of a tap layer as split-K "TN" GEMMs on the matrix cores.
//
//   dT[tap][cu][cv] = sum_{n,h,wu} U[n,h,wu,cu] * f(V[n, h+dh(tap), wu*s+dw(tap), cv])
//
// Both operands are NHWC, i.e. the reduction index (the pixel) is the ROW index of both tiles,
// while the MFMA wants it contiguous per lane.  gfx950's transposed LDS read
// (ds_read_b64_tr_b16: a 16-lane group reads a 4-row x 16-column block and each lane receives
// one column) does that transpose for free, so the tiles are staged exactly as they sit in HBM
// ([32 pixels][128 channels] bf16, whole 256-byte pixel rows => coalesced) and both MFMA
// operands are read with it.  Rows are 256 B apart (all 64 banks), so the 32-byte column
// granule a 4-lane quad reads is XOR-swizzled with sigma(k) = (k&3) | ((k>>3)&1)<<2: the eight
// rows a half-wave touches in one instruction ({0-3, 8-11} + 16g) land on eight distinct
// 8-bank groups => conflict-free.
//
// Block = 4 waves (2x2), tile 128(cu) x 128(cv) of ONE tap, one K slice; fp32 partial slabs are
// summed by a second kernel in a fixed order (bitwise reproducible, no atomics).
#include <stdlib.h>

#include <type_traits>
#include <utility>

#include "common.h"
#include "tapconv.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) rv_elem_t bf16x4;

struct WgradArgs {
    const bf16_t* U;
    const bf16_t* V;
    const float* scale;
    const float* shift;
    float* slabs;  // [ksplit][taps][cu_pad][cv_pad]
    int32_t N, H, Wu, Wv;
    int32_t cu_pad, cv_pad, ld_u, ld_v;
    int32_t stride_w;
    int32_t taps;
    int32_t ksplit, k_per_split;  // pixels per slice (multiple of 32)
    int32_t tiles_u, tiles_v;
    int32_t flags, v_affine;
    int8_t dh[kMaxTaps], dw[kMaxTaps];
};

__device__ __forceinline__ int sigma(int k) { return (k & 3) | (((k >> 3) & 1) << 2); }

__device__ __forceinline__ u32x4 transform8(u32x4 v, const float* sc, const float* sh, bool affine, bool relu) {
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
    return v;
}

__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WgradArgs a) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[2][2][32 * 128];  // [buffer][U/V][k][128]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // block decode: tiles fastest, then tap, then K slice (blocks of one slice run together => L2 reuse)
    int bx = blockIdx.x;
    const int tv = bx % a.tiles_v;
    bx /= a.tiles_v;
    const int tu = bx % a.tiles_u;
    bx /= a.tiles_u;
    const int tap = bx % a.taps;
    const int ks = bx / a.taps;
    const int u0 = tu * 128, v0 = tv * 128;
    const int dh = a.dh[tap], dw = a.dw[tap];

    const int64_t K = (int64_t)a.N * a.H * a.Wu;
    const int64_t k_begin = (int64_t)ks * a.k_per_split;
    const int64_t k_end = (k_begin + a.k_per_split < K) ? k_begin + a.k_per_split : K;

    // staging role: 2 pixels x one 16-byte chunk per operand per K step
    const int chunk = tid & 15;      // 8 channels
    const int prow = tid >> 4;       // pixel rows prow and prow + 16
    const bool u_ok = (u0 + chunk * 8) < a.cu_pad, v_ok = (v0 + chunk * 8) < a.cv_pad;
    const bool affine = a.flags & RV_IN_AFFINE, relu = a.flags & RV_IN_RELU;
    float sc[8], sh[8];
    if (affine) {
        const int c0 = (a.v_affine ? v0 : u0) + chunk * 8;
        const bool ok = a.v_affine ? v_ok : u_ok;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc[j] = ok ? a.scale[c0 + j] : 0.f;
            sh[j] = ok ? a.shift[c0 + j] : 0.f;
        }
    }
    // pixel coordinates of this thread's two rows, advanced incrementally
    int pn[2], phh[2], pw[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int64_t k = k_begin + prow + 16 * j;
        const int64_t hw = (int64_t)a.H * a.Wu;
        pn[j] = (int)(k / hw);
        const int64_t r = k - (int64_t)pn[j] * hw;
        phh[j] = (int)(r / a.Wu);
        pw[j] = (int)(r - (int64_t)phh[j] * a.Wu);
    }
    u32x4 ru[2], rv[2];
    bool ru_ok[2], rv_ok[2];
    auto load = [&](int64_t kbase) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t k = kbase + prow + 16 * j;
            ru[j] = u32x4{0u, 0u, 0u, 0u};
            rv[j] = u32x4{0u, 0u, 0u, 0u};
            ru_ok[j] = false;
            rv_ok[j] = false;
            if (k < k_end) {
                if (u_ok) {
                    ru_ok[j] = true;
                    ru[j] = *(const u32x4*)(a.U + ((int64_t)(pn[j] * a.H + phh[j]) * a.Wu + pw[j]) * a.ld_u + u0 + chunk * 8);
                }
                const int hv = phh[j] + dh, wv = pw[j] * a.stride_w + dw;
                if (v_ok && hv >= 0 && hv < a.H && wv >= 0 && wv < a.Wv) {
                    rv_ok[j] = true;
                    rv[j] = *(const u32x4*)(a.V + ((int64_t)(pn[j] * a.H + hv) * a.Wv + wv) * a.ld_v + v0 + chunk * 8);
                }
            }
            // advance this row by 32 pixels
            pw[j] += 32;
            while (pw[j] >= a.Wu) {
                pw[j] -= a.Wu;
                if (++phh[j] == a.H) {
                    phh[j] = 0;
                    ++pn[j];
                }
            }
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = prow + 16 * j;
            const int off = k * 128 + (((chunk >> 1) ^ sigma(k)) << 4) + (chunk & 1) * 8;
            pin_here(ru[j]);
            pin_here(rv[j]);
            // zero-filled (out-of-range) slots hold exact zeros; relu(shift) must not leak into them
            if ((affine || relu) && !a.v_affine && ru_ok[j]) ru[j] = transform8(ru[j], sc, sh, affine, relu);
            if ((affine || relu) && a.v_affine && rv_ok[j]) rv[j] = transform8(rv[j], sc, sh, affine, relu);
            *(u32x4*)(&lds[buf][0][off]) = ru[j];
            *(u32x4*)(&lds[buf][1][off]) = rv[j];
        }
    };

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // transposed-read addressing: lane = 16*g + 4*q + p supplies row (8g + q [+4]), columns 4p..4p+3 of its tile
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int row_lo = 8 * g + q, row_hi = row_lo + 4;
    auto frag = [&](const bf16_t* tile, int col0) -> bf16x8 {
        // col0: first channel of the 16-wide fragment inside the 128-wide tile
        const int gran = col0 >> 4;
        const bf16_t* p_lo = tile + row_lo * 128 + ((gran ^ sigma(row_lo)) << 4) + 4 * p;
        const bf16_t* p_hi = tile + row_hi * 128 + ((gran ^ sigma(row_hi)) << 4) + 4 * p;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p_lo);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p_hi);
        typedef __attribute__((ext_vector_type(8))) short s16x8;
        const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, both);
    };

    if (k_begin < k_end) {
        load(k_begin);
        store(0);
        __syncthreads();
        int buf = 0;
        for (int64_t kb = k_begin; kb < k_end; kb += 32) {
            const bool has_next = kb + 32 < k_end;
            if (has_next) load(kb + 32);
            bf16x8 fb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = frag(&lds[buf][1][0], wn * 64 + j * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16x8 fa = frag(&lds[buf][0][0], wm * 64 + i * 16);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = RV_MFMA_16x16x32(fa, fb[j], acc[i][j], 0, 0, 0);
            }
            if (has_next) store(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
    // slab store: D[row = cu][col = cv]: col = lane&15, row = (lane>>4)*4 + r
    float* slab = a.slabs + ((int64_t)ks * a.taps + tap) * a.cu_pad * a.cv_pad;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int cu = u0 + wm * 64 + i * 16 + (lane >> 4) * 4 + r;
            if (cu >= a.cu_pad) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int cv = v0 + wn * 64 + j * 16 + (lane & 15);
                if (cv < a.cv_pad) slab[(int64_t)cu * a.cv_pad + cv] = acc[i][j][r];
            }
        }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (no wave ends with slab stores in flight: see wgrad3_body)
}

// torch layout of the result (RV_WGRAD_TORCH_LAYOUT): dT[cu][cv][kh][kw] instead of the packed [tap][cu_pad][cv_pad]
struct UnpackTo {
    int32_t on, cu, cv, cu_pad, cv_pad, taps;
};

// sum of the split-K slabs in slab order (bitwise reproducible); elems is a multiple of 1024 (padded channel counts)
// 16-byte slab loads at AGENT scope (sc1): served by the memory side, never by a line of the recycled workspace that an earlier
// reduction left in this XCD's L2.  Inline asm (the builtin loads carry no scope), so the wait for the data is part of the same
// statement: the compiler does not track vmcnt for these.
__device__ __forceinline__ void slab_load4(const f32x4* p0, const f32x4* p1, const f32x4* p2, const f32x4* p3, f32x4& a, f32x4& b, f32x4& c,
                                           f32x4& d) {
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc1\n\t"
        "global_load_dwordx4 %1, %5, off sc1\n\t"
        "global_load_dwordx4 %2, %6, off sc1\n\t"
        "global_load_dwordx4 %3, %7, off sc1\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
        : "memory");
}
__device__ __forceinline__ f32x4 slab_load1(const f32x4* p) {
    f32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
    return v;
}

__device__ __forceinline__ void wgrad_reduce_body(const float* slabs, int ksplit, int64_t elems, float* out, const UnpackTo& up, int64_t first,
                                                  int64_t stride) {
    const int64_t n4 = elems >> 2;
    const f32x4* in = (const f32x4*)slabs;
    for (int64_t i = first; i < n4; i += stride) {
        // the slabs in slab order, left to right (bitwise reproducible, the order of every earlier form of this loop)
        f32x4 s;
        int k = 0;
        if (ksplit >= 4) {
            f32x4 a, b, c, d;
            slab_load4(in + i, in + n4 + i, in + 2 * n4 + i, in + 3 * n4 + i, a, b, c, d);
            s = ((a + b) + c) + d;
            k = 4;
        } else {
            s = slab_load1(in + i);
            k = 1;
        }
        for (; k + 3 < ksplit; k += 4) {
            f32x4 a, b, c, d;
            slab_load4(in + (int64_t)k * n4 + i, in + (int64_t)(k + 1) * n4 + i, in + (int64_t)(k + 2) * n4 + i, in + (int64_t)(k + 3) * n4 + i, a, b, c, d);
            s = (((s + a) + b) + c) + d;
        }
        for (; k < ksplit; ++k) s += slab_load1(in + (int64_t)k * n4 + i);
        if (!up.on) {
            ((f32x4*)out)[i] = s;
            continue;
        }
        // packed index (tap, u, v0..v0+3) -> dT[(u*cv + v)*taps + tap]; the padding rows / columns are dropped
        const int64_t e = i * 4;
        const int v0 = (int)(e % up.cv_pad);
        const int64_t r = e / up.cv_pad;
        const int u = (int)(r % up.cu_pad), tap = (int)(r / up.cu_pad);
        if (u >= up.cu) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (v0 + j < up.cv) out[((int64_t)u * up.cv + v0 + j) * up.taps + tap] = s[j];
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* slabs, int ksplit, int64_t elems, float* out, const UnpackTo up) {
    wgrad_reduce_body(slabs, ksplit, elems, out, up, blockIdx.x * (int64_t)blockDim.x + threadIdx.x, (int64_t)gridDim.x * blockDim.x);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (no wave ends with stores in flight: see wgrad3_body; rv_unfold_weight_grad may follow)
}


// ---------------------------------------------------------------------------------------------
// wgrad2: stride-1 layers whose width is a multiple of 64.  One block = 8 waves = up to THREE taps of one
// kernel row x 128(cu) x 128(cv) x one K slice.  A K step is 64 consecutive pixels of one image row: the U tile
// [64 px][128 cu] and the V halo [64+2 px][128 cv] are staged once and serve all taps of the group (tap j reads
// the V rows shifted by j), so each staged byte feeds 3x the MFMAs of wgrad_kernel and a step holds 48 MFMAs
// per wave between barriers instead of 16.
// ---------------------------------------------------------------------------------------------
struct Wgrad2Args {
    const bf16_t* U;
    const bf16_t* V;
    const float* scale;
    const float* shift;
    float* slabs;
    int32_t N, H, Wu;
    int32_t cu_pad, cv_pad, ld_u, ld_v;
    int32_t taps, groups;
    int32_t ksplit, chunks_per_split, chunks;  // 64-pixel chunks
    int32_t main_blocks, left_m;               // wgrad3's balanced split (plan()): blocks >= main_blocks take the K remainder of left_m tiles each
    int32_t tiles_u, tiles_v;
    int32_t flags, v_affine;
    int32_t xcd_remap;
    int8_t g_first[kMaxTaps], g_count[kMaxTaps];  // tap group -> first tap index / number of taps (<= 3)
    int8_t dh[kMaxTaps], dw[kMaxTaps];
};

template <int TG>
__device__ __forceinline__ void wgrad2_body(const Wgrad2Args& a, bf16_t (*lds)[2][66 * 128]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    int bx = blockIdx.x;
    const int tv = bx % a.tiles_v;
    bx /= a.tiles_v;
    const int tu = bx % a.tiles_u;
    bx /= a.tiles_u;
    const int grp = bx % a.groups;
    const int ks = bx / a.groups;
    const int u0 = tu * 128, v0 = tv * 128;
    const int tap0 = a.g_first[grp];
    const int dh = a.dh[tap0], dw0 = a.dw[tap0];
    const int wchunks = (a.Wu + 63) / 64;  // the last chunk of an image row may be partial (W = 1808, 2656 ...): zero-filled

    const int c_begin = ks * a.chunks_per_split;
    const int c_end = (c_begin + a.chunks_per_split < a.chunks) ? c_begin + a.chunks_per_split : a.chunks;

    const int chunk = tid & 15, prow = tid >> 4;  // prow 0..31: rows prow, prow+32 (U) / prow, +32, +64 (V halo)
    const bool u_ok = (u0 + chunk * 8) < a.cu_pad, v_ok = (v0 + chunk * 8) < a.cv_pad;
    const bool affine = a.flags & RV_IN_AFFINE, relu = a.flags & RV_IN_RELU;
    float sc[8], sh[8];
    if (affine) {
        const int c0 = (a.v_affine ? v0 : u0) + chunk * 8;
        const bool ok = a.v_affine ? v_ok : u_ok;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            sc[j] = ok ? a.scale[c0 + j] : 0.f;
            sh[j] = ok ? a.shift[c0 + j] : 0.f;
        }
    }
    u32x4 ru[2], rv[3];
    bool rv_ok[3];
    auto load = [&](int c) {
        const int row = c / wchunks, w0 = (c - row * wchunks) * 64;  // row = n*H + h
        const int h = row % a.H;
        const bf16_t* urow = a.U + ((int64_t)row * a.Wu) * a.ld_u + u0 + chunk * 8;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            ru[j] = u32x4{0u, 0u, 0u, 0u};
            if (u_ok && w0 + prow + 32 * j < a.Wu) ru[j] = *(const u32x4*)(urow + (int64_t)(w0 + prow + 32 * j) * a.ld_u);
        }
        const int hv = h + dh;
        const bool row_ok = hv >= 0 && hv < a.H;
        const bf16_t* vrow = a.V + ((int64_t)(row + dh) * a.Wu) * a.ld_v + v0 + chunk * 8;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            rv[j] = u32x4{0u, 0u, 0u, 0u};
            const int r = prow + 32 * j;  // halo row 0..65(+)
            const int wv = w0 + r + dw0;
            rv_ok[j] = v_ok && row_ok && r < 64 + TG - 1 && wv >= 0 && wv < a.Wu;
            if (rv_ok[j]) rv[j] = *(const u32x4*)(vrow + (int64_t)wv * a.ld_v);
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = prow + 32 * j;
            pin_here(ru[j]);
            if ((affine || relu) && !a.v_affine && u_ok) ru[j] = transform8(ru[j], sc, sh, affine, relu);
            *(u32x4*)(&lds[buf][0][k * 128 + (((chunk >> 1) ^ sigma(k)) << 4) + (chunk & 1) * 8]) = ru[j];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int k = prow + 32 * j;
            pin_here(rv[j]);
            if ((affine || relu) && a.v_affine && rv_ok[j]) rv[j] = transform8(rv[j], sc, sh, affine, relu);
            if (k < 66) *(u32x4*)(&lds[buf][1][k * 128 + (((chunk >> 1) ^ sigma(k)) << 4) + (chunk & 1) * 8]) = rv[j];
        }
    };

    f32x4 acc[TG][4][2];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    auto frag = [&](const bf16_t* tile, int row0, int col0) -> bf16x8 {
        // rows row0 + 8g + q (+4), columns col0 + 4p .. +3 (before the granule swizzle)
        const int r_lo = row0 + 8 * g + q, r_hi = r_lo + 4, gran = col0 >> 4;
        const bf16_t* p_lo = tile + r_lo * 128 + ((gran ^ sigma(r_lo)) << 4) + 4 * p;
        const bf16_t* p_hi = tile + r_hi * 128 + ((gran ^ sigma(r_hi)) << 4) + 4 * p;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p_lo);
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p_hi);
        const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, both);
    };

    if (c_begin < c_end) {
        load(c_begin);
        store(0);
        __syncthreads();
        int buf = 0;
        for (int c = c_begin; c < c_end; ++c) {
            const bool has_next = c + 1 < c_end;
            if (has_next) load(c + 1);
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {  // two MFMA K-steps of 32 pixels
                bf16x8 fa[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = frag(&lds[buf][0][0], kk * 32, wm * 64 + i * 16);
#pragma unroll
                for (int t = 0; t < TG; ++t) {
                    bf16x8 fb[2];
#pragma unroll
                    for (int j = 0; j < 2; ++j) fb[j] = frag(&lds[buf][1][0], kk * 32 + t, wn * 32 + j * 16);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            acc[t][i][j] = RV_MFMA_16x16x32(fb[j], fa[i], acc[t][i][j], 0, 0, 0);  // (transposed: as wgrad3)
                }
            }
            if (has_next) store(buf ^ 1);
            __syncthreads();
            buf ^= 1;
        }
    }
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        float* slab = a.slabs + ((int64_t)ks * a.taps + tap0 + t) * a.cu_pad * a.cv_pad;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cu = u0 + wm * 64 + i * 16 + (lane & 15);
            if (cu >= a.cu_pad) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cv = v0 + wn * 32 + j * 16 + (lane >> 4) * 4;  // (cv_pad % 32 == 0: a quad is inside or outside as a whole)
                if (cv < a.cv_pad) *(f32x4*)(slab + (int64_t)cu * a.cv_pad + cv) = acc[t][i][j];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (no wave ends with slab stores in flight: see wgrad3_body)
}

__global__ __launch_bounds__(512, 2) void wgrad2_kernel(const Wgrad2Args a) {
    __shared__ __attribute__((aligned(16))) bf16_t lds[2][2][66 * 128];
    int bx = blockIdx.x / (a.tiles_v * a.tiles_u);
    const int cnt = a.g_count[bx % a.groups];  // uniform per block
    if (cnt == 3)
        wgrad2_body<3>(a, lds);
    else if (cnt == 2)
        wgrad2_body<2>(a, lds);
    else
        wgrad2_body<1>(a, lds);
}

// ------------------------------------------------------------------------------------------------------------------
// wgrad3: wgrad2's tiling (128 cu x 128 cv x up to 3 taps of one kernel row, 64-pixel chunks, 8 waves) with the operands
// streamed global -> LDS by LDS-DMA into a ring of FOUR chunk slots: the chunk three ahead is issued during each chunk, the
// only memory waits are counted (two chunks stay in flight across the barrier) and there is one raw barrier per chunk.
// Plain bf16 operands only (the DMA bypasses the registers, so no folded BatchNorm on the way in) -- which is what the
// engine provides on the wide layers (engine.py, MATERIALIZE_FOR_DMA).  The 32-byte granule swizzle sigma() is applied to
// the per-lane SOURCE address (the DMA writes LDS lane-linearly).
// What the loop is built around (profiles/r02_wgrad_ablation.md): DMA issue, fragment reads and MFMAs of the first version
// ADDED UP instead of overlapping -- ~120 address instructions per wave and chunk, all eight waves in the same phase.  Here
// (a) no address arithmetic is left in the loop (incremental source pointers, immediate LDS offsets), (b) the fragments of
// tap-step u+1 are requested before the MFMAs of tap-step u, (c) the two waves of a SIMD issue their DMA half a chunk apart.
// ------------------------------------------------------------------------------------------------------------------

__device__ __attribute__((aligned(256))) uint32_t g_wgrad_zero_page[64];

// LDS image of the ring: [group of 4 pixel rows][slot][4 rows x 256 B] -- a DMA instruction still writes 1 KiB contiguous
// (4 rows of one slot), while the slot and the K-step enter every fragment read as an IMMEDIATE offset (slot * 1 KiB,
// K-step * 32 KiB): the per-lane read addresses are loop-invariant registers and the main loop has no address arithmetic.
constexpr int kW3VBase = 16 * 4096;            // U: 16 groups, V: 17 groups (rows 64..67 = the tap halo)
constexpr int kW3Lds = kW3VBase + 17 * 4096;   // 135 168 B: one workgroup per CU

template <int I>
using w3_int = std::integral_constant<int, I>;
template <class F, int... I>
__device__ __forceinline__ void w3_for(F&& f, std::integer_sequence<int, I...>) {
    (f(w3_int<I>{}), ...);
}

struct W3Frag {  // the two halves of an MFMA operand as the transposed reads deliver them; joined only at the use, after the wait
    s16x4 lo, hi;
};
template <int OFF_LO, int OFF_HI>
__device__ __forceinline__ void w3_read(W3Frag& f, uint32_t a_lo, uint32_t a_hi) {
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.lo) : "v"(a_lo), "i"(OFF_LO));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(f.hi) : "v"(a_hi), "i"(OFF_HI));
}
__device__ __forceinline__ bf16x8 w3_join(const W3Frag& f) {
    typedef __attribute__((ext_vector_type(8))) short s16x8;
    const s16x8 both = __builtin_shufflevector(f.lo, f.hi, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf16x8, both);
}

// One (tile, K range) of the launch: tile (tu, tv, tap group grp), chunks [c_begin, c_end), partial sums into slab `ks`.
template <int TG>
__device__ __forceinline__ void wgrad3_body(const Wgrad2Args& a, uint8_t* smem, const int tu, const int tv, const int grp, const int ks,
                                            const int c_begin, const int c_end) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));  // (per-lane addresses are formed per segment: hoisted out of the remainder blocks' loop they stay live through the epilogue)
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int u0 = tu * 128, v0 = tv * 128;
    const int tap0 = a.g_first[grp];
    const int dh = a.dh[tap0], dw0 = a.dw[tap0];
    const int wchunks = (a.Wu + 63) / 64;  // the last chunk of an image row may be partial (W = 1808, 2656 ...): zero-filled
    if (c_begin >= c_end) return;  // (the plan never makes an empty slice; its slab would be left unwritten)

    // ---- operand stream.  One wave-instruction = 4 pixel rows x 16 chunks of 16 bytes; wave w owns rows 8w .. 8w+7 of both
    // tiles, wave 0 also the halo rows 64..67.  Per lane five NOMINAL source pointers (valid or not) that advance by 64 pixels
    // per chunk; a chunk that touches an image edge (first / last of a row, a row whose tap row is outside the image) takes the
    // slow path, which recomputes them and swaps the out-of-image lanes' pointers for the zero page.
    const int d_row = lane >> 4, d_c16 = lane & 15;
    typedef const __attribute__((address_space(1))) void glb_t;
    typedef __attribute__((address_space(3))) void lds_t;
    const uint8_t* zero = (const uint8_t*)g_wgrad_zero_page + (d_c16 & 7) * 16;
    const int sg = d_row | ((wave & 1) << 2);  // sigma(k) of this lane's rows 8w + 4j + d_row; rows 64.. : d_row
    const int swz_own = ((((d_c16 >> 1) ^ sg) << 1) | (d_c16 & 1)) * 16, swz_halo = ((((d_c16 >> 1) ^ d_row) << 1) | (d_c16 & 1)) * 16;
    const bool halo_wave = wave == 0;
    const bool halo_lane = d_row < TG - 1;  // halo rows 64 .. 64 + TG - 2 exist
    const int64_t step_u = (int64_t)64 * a.ld_u * 2, step_v = (int64_t)64 * a.ld_v * 2;
    const uint8_t *pu0 = zero, *pu1 = zero, *pv0 = zero, *pv1 = zero, *pv2 = zero;
    bool fresh = true;  // the pointers hold nothing yet
    int cur = c_begin, cur_row = c_begin / wchunks, cur_w0 = (c_begin - cur_row * wchunks) * 64, cur_h = cur_row % a.H;
    // (by-value helper: `c ? x : y` on two pointer LVALUES is itself an lvalue, which clang lowers to a select of stack addresses)
    auto pick = [](bool c, const uint8_t* x, const uint8_t* y) -> const uint8_t* { return c ? x : y; };
    auto dma = [&](const uint8_t* p, int lds_byte) { __builtin_amdgcn_global_load_lds((glb_t*)p, (lds_t*)(smem + lds_byte), 16, 0, 0); };
    const uint8_t *su0 = zero, *su1 = zero, *sv0 = zero, *sv1 = zero, *sv2 = zero;  // what the next chunk's instructions read
    auto prepare = [&]() {  // sources of the chunk at the cursor; then the cursor moves on
        su0 = zero, su1 = zero, sv0 = zero, sv1 = zero, sv2 = zero;
        if (cur < c_end) {  // (past the slice: same instruction count -- the waits are counted -- from the zero page)
            const int hv = cur_h + dh;
            const bool row_ok = hv >= 0 && hv < a.H;
            const bool slow = fresh || cur_w0 == 0 || cur_w0 + 64 + dw0 + TG - 1 > a.Wu || cur_w0 + 64 > a.Wu || cur_w0 + dw0 < 0 || !row_ok;
            if (!slow) {
                pu0 += step_u, pu1 += step_u, pv0 += step_v, pv1 += step_v, pv2 += step_v;
                su0 = pu0, su1 = pu1, sv0 = pv0, sv1 = pv1, sv2 = pick(halo_lane, pv2, zero);
            } else {
                const uint8_t* urow = (const uint8_t*)(a.U + ((int64_t)cur_row * a.Wu + cur_w0) * a.ld_u + u0);
                const uint8_t* vrow = (const uint8_t*)(a.V + ((int64_t)(cur_row + dh) * a.Wu + cur_w0 + dw0) * a.ld_v + v0);
                const int k0 = wave * 8 + d_row, k1 = k0 + 4, wv = cur_w0 + dw0;
                pu0 = urow + (int64_t)k0 * a.ld_u * 2 + swz_own;
                pu1 = urow + (int64_t)k1 * a.ld_u * 2 + swz_own;
                pv0 = vrow + (int64_t)k0 * a.ld_v * 2 + swz_own;
                pv1 = vrow + (int64_t)k1 * a.ld_v * 2 + swz_own;
                pv2 = vrow + (int64_t)(64 + d_row) * a.ld_v * 2 + swz_halo;
                su0 = pick(cur_w0 + k0 < a.Wu, pu0, zero);
                su1 = pick(cur_w0 + k1 < a.Wu, pu1, zero);
                sv0 = pick(row_ok && wv + k0 >= 0 && wv + k0 < a.Wu, pv0, zero);
                sv1 = pick(row_ok && wv + k1 >= 0 && wv + k1 < a.Wu, pv1, zero);
                sv2 = pick(row_ok && halo_lane && wv + 64 + d_row < a.Wu, pv2, zero);
            }
            fresh = false;
            ++cur;
            cur_w0 += 64;
            if (cur_w0 >= a.Wu) {
                cur_w0 = 0;
                ++cur_row;
                cur_h = cur_h + 1 == a.H ? 0 : cur_h + 1;
            }
        }
    };
    auto piece = [&](int k, int slot) {  // instruction k of the prepared chunk -> ring slot
        const int dst = slot * 1024 + wave * 8192;  // rows 8w.. = groups 2w, 2w+1
        if (k == 0) dma(su0, dst);
        if (k == 1) dma(su1, dst + 4096);
        if (k == 2) dma(sv0, kW3VBase + dst);
        if (k == 3) dma(sv1, kW3VBase + dst + 4096);
        if (k == 4 && halo_wave) dma(sv2, kW3VBase + 16 * 4096 + slot * 1024);
    };
    auto issue = [&](int slot) {
        prepare();
#pragma unroll
        for (int k = 0; k < 5; ++k) piece(k, slot);
    };

    f32x4 acc[TG][4][2];
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[t][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- fragment read addresses (transposed reads as inline asm: hipcc puts an s_waitcnt vmcnt(0) in front of the
    // ds_read_tr BUILTIN whenever an LDS-DMA is in flight, which would drain the ring every chunk; the price is that the compiler
    // does not see these reads, so their lgkmcnt waits are placed by hand and pinned with sched_barrier, rule 18).
    // Row r of a tile sits at (r >> 2) * 4096 + slot * 1024 + (r & 3) * 256; a lane reads rows r = row0 + 8g + q and r + 4.
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem;
    uint32_t bu[4], bv[TG][2][2];
    auto row_byte = [](int r) { return (uint32_t)((r >> 2) * 4096 + (r & 3) * 256); };
#pragma unroll
    for (int i = 0; i < 4; ++i)  // U rows 8g + q (+4: the next group, same sigma -> immediate +4096)
        bu[i] = lds0 + row_byte(8 * g + q) + (uint32_t)((((wm * 4 + i) ^ sigma(8 * g + q)) << 4) + 4 * p) * 2u;
#pragma unroll
    for (int t = 0; t < TG; ++t)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int hi = 0; hi < 2; ++hi) {
                const int r = t + 8 * g + q + 4 * hi;  // tap t reads the V rows shifted by t pixels
                bv[t][j][hi] = lds0 + kW3VBase + row_byte(r) + (uint32_t)((((wn * 2 + j) ^ sigma(r)) << 4) + 4 * p) * 2u;
            }

    W3Frag fa[2][4], fb[2][2];
    // one UNIT = one tap of one 32-pixel K-step = 8 MFMAs; units of a chunk: u = kk * TG + t
    auto read_unit = [&](auto S_, auto U_) {  // fragments of unit U of the chunk in slot S
        constexpr int S = decltype(S_)::value, U = decltype(U_)::value, kk = U / TG, t = U % TG;
        constexpr int off = kk * 8 * 4096 + S * 1024;
        if constexpr (t == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) w3_read<off, off + 4096>(fa[kk & 1][i], bu[i], bu[i]);
        }
        w3_read<off, off>(fb[U & 1][0], bv[t][0][0], bv[t][0][1]);
        w3_read<off, off>(fb[U & 1][1], bv[t][1][0], bv[t][1][1]);
    };
    auto mfma_unit = [&](auto U_) {
        constexpr int U = decltype(U_)::value, kk = U / TG, t = U % TG;
        const bf16x8 b0 = w3_join(fb[U & 1][0]), b1 = w3_join(fb[U & 1][1]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bf16x8 ai = w3_join(fa[kk & 1][i]);
            // (operands swapped: D = V^T U, so that a lane holds FOUR CONSECUTIVE cv of one cu -- 16-byte slab stores, see the epilogue)
            acc[t][i][0] = RV_MFMA_16x16x32(b0, ai, acc[t][i][0], 0, 0, 0);
            acc[t][i][1] = RV_MFMA_16x16x32(b1, ai, acc[t][i][1], 0, 0, 0);
        }
    };
    // The two halves of the workgroup (waves w and w + 4 share a SIMD) issue their DMA at different points of the chunk -- half A
    // before its first unit, half B between the two K-steps -- so that one half's address / issue work runs under the other
    // half's MFMAs; both meet at the one barrier per chunk.
    const bool half_b = wave >= 4;
    auto step = [&](auto S_) {  // chunk in slot S: its unit-0 fragments were requested before the barrier that opened it
        constexpr int S = decltype(S_)::value, NU = 2 * TG;
        // the five (four) instructions of chunk +3 go into the slot the chunk before this one was read from (closed by the last
        // barrier), one after each tap-step: half A after units 0.., half B one unit later, all before the last unit's wait
        prepare();
        w3_for([&](auto U_) {
            constexpr int U = decltype(U_)::value;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // unit U's fragments are in registers
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (U + 1 < NU) {
                read_unit(S_, w3_int<U + 1>{});  // lands under unit U's MFMAs
            } else {
                // every fragment of this chunk has been read by this wave and the next chunk's DMA has landed: close the chunk
                if (halo_wave)
                    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");  // chunks +2 and +3 stay in flight (5 instructions each)
                else
                    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");  // (4 each)
                __builtin_amdgcn_s_barrier();
                read_unit(w3_int<(S + 1) & 3>{}, w3_int<0>{});
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_unit(U_);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (U < NU - 1) {
#pragma unroll
                for (int k = 0; k < 5; ++k) {
                    const int ua = k < NU - 2 ? k : NU - 2, ub = k + 1 < NU - 2 ? k + 1 : NU - 2;
                    if (ua == U && !half_b) piece(k, (S + 3) & 3);
                    if (ub == U && half_b) piece(k, (S + 3) & 3);
                }
            }
        }, std::make_integer_sequence<int, NU>{});
    };

    issue(0);
    issue(1);
    issue(2);
    if (halo_wave)
        asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_unit(w3_int<0>{}, w3_int<0>{});
    for (int c = c_begin;;) {
        step(w3_int<0>{});
        if (++c >= c_end) break;
        step(w3_int<1>{});
        if (++c >= c_end) break;
        step(w3_int<2>{});
        if (++c >= c_end) break;
        step(w3_int<3>{});
        if (++c >= c_end) break;
    }
    // The last step's read_unit -- unit 0 of a chunk that does not exist -- is still IN FLIGHT here, and its twelve destination register pairs are dead
    // as far as the compiler knows (inline asm: it sees them written at the asm statement).  Untied, it hoisted the epilogue's address arithmetic above
    // this wait and formed the first slab address in two of those registers; the hardware does not order a VALU write against the return of an older LDS
    // read, so a read that came back late (a neighbour on the CU keeping the LDS busy) overwrote the address -- with the zeros of the past-the-slice
    // slot: sixteen-byte stores to address 0, "memory access fault on address (nil)", free-running rv-waymo steps of round 6 (profiles/r06_ab_notes.md
    // section 4: found with the debug agent's register dump); with other contents, stores to a wrong place.  The registers are operands of the wait.
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)"
                 : "+v"(fa[0][0].lo), "+v"(fa[0][0].hi), "+v"(fa[0][1].lo), "+v"(fa[0][1].hi), "+v"(fa[0][2].lo), "+v"(fa[0][2].hi), "+v"(fa[0][3].lo),
                   "+v"(fa[0][3].hi), "+v"(fb[0][0].lo), "+v"(fb[0][0].hi), "+v"(fb[0][1].lo), "+v"(fb[0][1].hi)
                 :
                 : "memory");
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < TG; ++t) {
        float* slab = a.slabs + ((int64_t)ks * a.taps + tap0 + t) * a.cu_pad * a.cv_pad;
        // acc[t][i][j][r] = dT[cu = 16 i + l15][cv = 16 j + 4 lg + r] of this wave's 64 x 32 tile (transposed accumulation, above): one
        // 16-byte store per accumulator register quad -- 24 store instructions per thread instead of 96 four-byte ones (a CU's store
        // path takes ~150 cycles per instruction whatever its width: -5 us per block, a quarter of a small layer's launch)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int cu = u0 + wm * 64 + i * 16 + (lane & 15);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int cv = v0 + wn * 32 + j * 16 + (lane >> 4) * 4;
                *(f32x4*)(slab + (int64_t)cu * a.cv_pad + cv) = acc[t][i][j];
            }
        }
    }
    // A wave does not END with its slab stores in flight, and the reduction reads the slabs with agent-scope loads (slab_load):
    // once in ~2500 two-stream training steps of round 4 the split-K reduction that follows on the SAME stream summed slab lines that
    // held the previous tenant of the recycled workspace -- always the 64-workgroup launch of one 1x1 layer, one wrong weight gradient,
    // nothing else (profiles/r04_ab_notes.md, "One wrong weight gradient"; the round-5 soak: profiles/r05_race_soak_*.txt).  Two readings
    // fit that symptom -- stores still in flight at the end of the kernel with another queue busy, or a line of the recycled workspace
    // that the PREVIOUS reduction left in the reading XCD's L2 -- and both are closed: this wait, and loads that do not hit in a
    // non-coherent L2 line.  (Round 6: a third reading, and the likeliest -- the address race fixed above the stores misplaces a wave's slab lines.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__device__ __forceinline__ void wgrad3_segment(const Wgrad2Args& a, uint8_t* smem, int tile, int ks, int c_begin, int c_end) {
    const int tv = tile % a.tiles_v;
    tile /= a.tiles_v;
    const int tu = tile % a.tiles_u;
    const int grp = tile / a.tiles_u;
    const int cnt = a.g_count[grp];  // uniform per block
    if (cnt == 3)
        wgrad3_body<3>(a, smem, tu, tv, grp, ks, c_begin, c_end);
    else if (cnt == 2)
        wgrad3_body<2>(a, smem, tu, tv, grp, ks, c_begin, c_end);
    else
        wgrad3_body<1>(a, smem, tu, tv, grp, ks, c_begin, c_end);
}

__global__ __launch_bounds__(512, 2) void wgrad3_kernel(const Wgrad2Args a) {
    extern __shared__ __attribute__((aligned(16))) uint8_t w3_smem[];
    // XCD-aware block order: workgroups are dealt round-robin over the 8 XCDs (one L2 each).  Blocks that share a K slice
    // read the same pixels of U and V, so each XCD gets a CONTIGUOUS range of the (K-slice-major) block list: a slice is
    // then fetched into one or two L2s instead of all eight (bijective for any grid size).
    int bx = blockIdx.x;
    if (a.xcd_remap) {
        const int total = gridDim.x, q = total >> 3, r = total & 7, xcd = bx & 7;
        bx = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bx >> 3);
    }
    const int n_tiles = a.tiles_v * a.tiles_u * a.groups;
    // regular block: one tile, one K slice of chunks_per_split chunks.  Remainder block of the balanced split (plan()): the K remainder
    // [ks_main * chunks_per_split, chunks) of left_m tiles, one after the other, into the last slab.
    const bool left = bx >= a.main_blocks;
    const int ks = left ? a.main_blocks / n_tiles : bx / n_tiles;
    const int first = left ? (bx - a.main_blocks) * a.left_m : bx - ks * n_tiles;
    const int last = left ? (first + a.left_m < n_tiles ? first + a.left_m : n_tiles) : first + 1;
    const int c_begin = ks * a.chunks_per_split;
    const int c_end = (left || c_begin + a.chunks_per_split > a.chunks) ? a.chunks : c_begin + a.chunks_per_split;
    for (int t = first; t < last; ++t) {
        if (t != first) __syncthreads();  // (the ring is rewritten: every wave has read its last fragments)
        wgrad3_segment(a, w3_smem, __builtin_amdgcn_readfirstlane(t), __builtin_amdgcn_readfirstlane(ks), __builtin_amdgcn_readfirstlane(c_begin),
                       __builtin_amdgcn_readfirstlane(c_end));
    }
}

struct WgradPlan {
    int taps, tiles_u, tiles_v, ksplit, k_per_split;
    int64_t elems;
    bool v2;
    int groups, chunks, chunks_per_split;
    int main_blocks, left_m, left_blocks;  // wgrad3's balanced split
};

bool wgrad_dma_eligible(const rvTapGeom* g, const rvTapShape* s);

int plan(const rvTapGeom* g, const rvTapShape* s, WgradPlan* p) {
    p->taps = g->kh * g->kw;
    const int cu = rv_pad32(g->cu), cv = rv_pad32(g->cv);
    p->tiles_u = rv_ceil_div(cu, 128);
    p->tiles_v = rv_ceil_div(cv, 128);
    const int64_t K = (int64_t)s->N * s->H * s->Wu;
    const int64_t chunks = (K + 31) / 32;
    const int base = p->taps * p->tiles_u * p->tiles_v;
    int64_t ks = (1024 + base - 1) / base;  // ~4 blocks per CU
    if (ks > chunks) ks = chunks;
    if (ks < 1) ks = 1;
    const int64_t per = (chunks + ks - 1) / ks;
    p->k_per_split = (int)(per * 32);
    p->ksplit = (int)((chunks + per - 1) / per);
    p->elems = (int64_t)p->taps * cu * cv;
    // wgrad2: stride 1, whole 64-pixel chunks per image row
    p->v2 = (g->stride_w == 1) && (s->Wu >= 64);
    if (p->v2) {
        p->groups = g->kh * ((g->kw + 2) / 3);
        p->chunks = s->N * s->H * ((s->Wu + 63) / 64);
        const int base2 = p->groups * p->tiles_u * p->tiles_v;
        // split-K factor: ONE round of workgroups, as close to 256 as the tile count allows (round-4 sweep over the factor,
        // profiles/r04_wgrad_ksplit.txt: 128 <-> 128 3x3 at W = 2656 takes 185 us with 255 workgroups and 220-290 us with 504 / 768
        // / 1008; 256 <-> 256 3x3 504 us with 252 against 528 with 768; the 1x1 256 <-> 256 107 us with 256 against 155 with 1024;
        // 512 <-> 512 the same with 240 and 768 -- every extra round pays the per-workgroup prologue, the 196 KB slab and its
        // share of the reduction again), at least 32 K chunks per block so that the fp32 slab traffic stays small next to the MFMA
        // work; layers with more tiles than CUs take no split at all
        int64_t ks_max = p->chunks / 32;
        if (ks_max < 1) ks_max = 1;
        const int64_t cus = rv_cu_count();
        int64_t ks2 = cus / base2;
        if (ks2 < 1) ks2 = 1;
        if (ks2 > ks_max) ks2 = ks_max;
        p->chunks_per_split = (int)((p->chunks + ks2 - 1) / ks2);
        p->ksplit = (p->chunks + p->chunks_per_split - 1) / p->chunks_per_split;
        p->main_blocks = base2 * p->ksplit;
        p->left_m = 0;
        p->left_blocks = 0;
        // Balanced split (wgrad3): 48 tiles (512 <-> 512, 3x3) take 5 slices each = 240 workgroups on 256 CUs.  The 16 idle CUs get
        // work: every tile keeps 5 slices of L chunks and leaves a remainder R = C - 5 L that a "remainder" workgroup sums for m = 3
        // tiles in turn (one more slab per tile, three epilogues for that workgroup), with m R + (m - 1) overhead = L so that all 256
        // workgroups finish together.  The regular slices of all tiles still cover the same pixels at the same time (L2 sharing), and
        // so do the remainders.
        const int64_t free_cus = cus - (int64_t)base2 * ks2;
        if (wgrad_dma_eligible(g, s) && ks2 >= 2 && p->ksplit == ks2 && free_cus >= 8 &&
            (int64_t)base2 * ks2 * 100 < cus * 96) {
            const int64_t m = (base2 + free_cus - 1) / free_cus;
            const int64_t ovh = 24;  // chunks one more prologue + epilogue is worth
            const int64_t len = (m * p->chunks + (m - 1) * ovh + m * ks2) / (m * ks2 + 1);  // (rounded up)
            const int64_t rem = p->chunks - ks2 * len;
            if (m <= 4 && rem >= 32 && len >= 32) {
                p->chunks_per_split = (int)len;
                p->ksplit = (int)ks2 + 1;
                p->main_blocks = base2 * (int)ks2;
                p->left_m = (int)m;
                p->left_blocks = (int)((base2 + m - 1) / m);
            }
        }
    }
    return 0;
}

}  // namespace

extern "C" int64_t rv_tap_wgrad_workspace_bytes(const rvTapGeom* g, const rvTapShape* s) {
    WgradPlan p;
    plan(g, s, &p);
    return p.elems * p.ksplit * (int64_t)sizeof(float);
}

namespace {
bool wgrad_dma_eligible(const rvTapGeom* g, const rvTapShape* s) {
    return !(s->flags & (RV_IN_AFFINE | RV_IN_RELU)) && rv_pad32(g->cu) % 128 == 0 && rv_pad32(g->cv) % 128 == 0;
}
}  // namespace

extern "C" int rv_tap_wgrad_info(const rvTapGeom* g, const rvTapShape* s, int32_t* host_info) {
    RV_REQUIRE(g && s && host_info, "rv_tap_wgrad_info: null argument");
    WgradPlan p;
    plan(g, s, &p);
    host_info[0] = p.v2 ? (wgrad_dma_eligible(g, s) ? 3 : 2) : 1;
    host_info[1] = p.ksplit;
    host_info[2] = p.v2 ? p.main_blocks + p.left_blocks : p.tiles_v * p.tiles_u * p.taps * p.ksplit;
    return 0;
}

extern "C" int rv_tap_wgrad(const rvTapGeom* g, const rvTapShape* s, const void* U, int32_t ld_u, const void* V,
                            int32_t ld_v, const float* in_scale, const float* in_shift, int32_t v_affine,
                            float* dT_packed, void* workspace, rvStream stream) {
    RV_REQUIRE(g && s && U && V && dT_packed && workspace, "rv_tap_wgrad: null argument");
    RV_REQUIRE(s->Wv == s->Wu * g->stride_w, "rv_tap_wgrad: Wv (%d) must equal Wu (%d) * stride_w (%d)", s->Wv, s->Wu, g->stride_w);
    RV_REQUIRE(!(s->flags & RV_IN_AFFINE) || (in_scale && in_shift), "rv_tap_wgrad: RV_IN_AFFINE without scale/shift");
    RV_REQUIRE(g->kh * g->kw <= kMaxTaps, "rv_tap_wgrad: kernel %dx%d unsupported", g->kh, g->kw);
    WgradPlan p;
    plan(g, s, &p);
    UnpackTo up;
    up.on = (s->flags & RV_WGRAD_TORCH_LAYOUT) ? 1 : 0;
    up.cu = g->cu;
    up.cv = g->cv;
    up.cu_pad = rv_pad32(g->cu);
    up.cv_pad = rv_pad32(g->cv);
    up.taps = g->kh * g->kw;
    if (p.v2) {
        Wgrad2Args b;
        memset(&b, 0, sizeof(b));
        b.U = (const bf16_t*)U;
        b.V = (const bf16_t*)V;
        b.scale = in_scale;
        b.shift = in_shift;
        b.slabs = (float*)workspace;
        b.N = s->N;
        b.H = s->H;
        b.Wu = s->Wu;
        b.cu_pad = rv_pad32(g->cu);
        b.cv_pad = rv_pad32(g->cv);
        b.ld_u = ld_u;
        b.ld_v = ld_v;
        RV_REQUIRE(ld_u >= b.cu_pad && ld_v >= b.cv_pad && ld_u % 8 == 0 && ld_v % 8 == 0, "rv_tap_wgrad: bad channel strides");
        b.taps = p.taps;
        b.groups = p.groups;
        b.ksplit = p.ksplit;
        b.chunks_per_split = p.chunks_per_split;
        b.chunks = p.chunks;
        b.main_blocks = p.main_blocks;
        b.left_m = p.left_m;
        b.tiles_u = p.tiles_u;
        b.tiles_v = p.tiles_v;
        b.flags = s->flags & ~(RV_WGRAD_TORCH_LAYOUT | RV_SEL_MASK);
        b.v_affine = v_affine;
#ifdef RV_DIAG_NO_XCD_REMAP  // (diagnostic build only, profiles/tools/diag_wgrad.sh: every K slice spread over all eight L2s -- the slope of time against L2-miss traffic)
        b.xcd_remap = 0;
#else
        b.xcd_remap = 1;
#endif
        int gi = 0;
        for (int ky = 0; ky < g->kh; ++ky)
            for (int kx = 0; kx < g->kw; ++kx) {
                b.dh[ky * g->kw + kx] = (int8_t)(ky - g->pad_h);
                b.dw[ky * g->kw + kx] = (int8_t)(kx - g->pad_w);
                if (kx % 3 == 0) {
                    b.g_first[gi] = (int8_t)(ky * g->kw + kx);
                    b.g_count[gi] = (int8_t)((g->kw - kx) < 3 ? (g->kw - kx) : 3);
                    ++gi;
                }
            }
        hipStream_t st2 = (hipStream_t)stream;
        const int grid2 = p.main_blocks + p.left_blocks;
        const bool dma = wgrad_dma_eligible(g, s);
        if (dma) {
            static bool attr_set = false;
            if (!attr_set) {
                (void)hipFuncSetAttribute((const void*)wgrad3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                attr_set = true;
            }
            hipLaunchKernelGGL(wgrad3_kernel, dim3(grid2), dim3(512), kW3Lds, st2, b);
            RV_CHECK_LAUNCH("wgrad3_kernel");
        } else {
            hipLaunchKernelGGL(wgrad2_kernel, dim3(grid2), dim3(512), 0, st2, b);
            RV_CHECK_LAUNCH("wgrad2_kernel");
        }
#ifndef RV_DIAG_SKIP_REDUCE  // (diagnostic build only: WRONG gradients -- an upper bound on what ANY way of folding the split-K reduction away can return)
        const int rb2 = (int)((p.elems / 4 + 255) / 256 < 4096 ? (p.elems / 4 + 255) / 256 : 4096);
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb2), dim3(256), 0, st2, (const float*)workspace, p.ksplit, p.elems, dT_packed, up);
        RV_CHECK_LAUNCH("wgrad_reduce_kernel");
#endif
        return 0;
    }
    WgradArgs a;
    memset(&a, 0, sizeof(a));
    a.U = (const bf16_t*)U;
    a.V = (const bf16_t*)V;
    a.scale = in_scale;
    a.shift = in_shift;
    a.slabs = (float*)workspace;
    a.N = s->N;
    a.H = s->H;
    a.Wu = s->Wu;
    a.Wv = s->Wv;
    a.cu_pad = rv_pad32(g->cu);
    a.cv_pad = rv_pad32(g->cv);
    a.ld_u = ld_u;
    a.ld_v = ld_v;
    RV_REQUIRE(ld_u >= a.cu_pad && ld_v >= a.cv_pad && ld_u % 8 == 0 && ld_v % 8 == 0, "rv_tap_wgrad: bad channel strides");
    a.stride_w = g->stride_w;
    a.taps = p.taps;
    a.ksplit = p.ksplit;
    a.k_per_split = p.k_per_split;
    a.tiles_u = p.tiles_u;
    a.tiles_v = p.tiles_v;
    a.flags = s->flags & ~(RV_WGRAD_TORCH_LAYOUT | RV_SEL_MASK);
    a.v_affine = v_affine;
    for (int ky = 0; ky < g->kh; ++ky)
        for (int kx = 0; kx < g->kw; ++kx) {
            a.dh[ky * g->kw + kx] = (int8_t)(ky - g->pad_h);
            a.dw[ky * g->kw + kx] = (int8_t)(kx - g->pad_w);
        }
    hipStream_t st = (hipStream_t)stream;
    const int grid = p.tiles_v * p.tiles_u * p.taps * p.ksplit;
    hipLaunchKernelGGL(wgrad_kernel, dim3(grid), dim3(256), 0, st, a);
    RV_CHECK_LAUNCH("wgrad_kernel");
    const int rb = (int)((p.elems / 4 + 255) / 256 < 4096 ? (p.elems / 4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rb), dim3(256), 0, st, (const float*)workspace, p.ksplit, p.elems, dT_packed, up);
    RV_CHECK_LAUNCH("wgrad_reduce_kernel");
    return 0;
}

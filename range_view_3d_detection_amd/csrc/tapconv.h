// Tap tables shared by the tap-conv kernels, the weight packers and the weight-gradient kernel.
#pragma once
#include "common.h"

constexpr int kMaxTaps = 24;   // 3x8 (ConvTranspose2d k=(3,8)) is the largest kernel on the path
constexpr int kMaxPhases = 4;  // stride_w <= 4
constexpr int RV_OUT_BNB = 1 << 16;  // internal launch flag: see TapConvArgs::bnb_*

struct TapTable {
    int32_t ntaps[kMaxPhases];    // taps of each output phase
    int32_t w_first[kMaxPhases];  // index of the phase's first tap image in the packed weight
    int32_t dw_min[kMaxPhases], dw_max[kMaxPhases];
    int32_t w_tile[kMaxPhases];   // halo width (pixels) of the LDS input tile, set per launch
    int32_t dh_min, rows;         // halo rows: dh_min .. dh_min + rows - 1
    int8_t dh[kMaxPhases][kMaxTaps];
    int8_t dw[kMaxPhases][kMaxTaps];
    int8_t ky[kMaxPhases][kMaxTaps];  // kernel coordinates of the tap (weight packing)
    int8_t kx[kMaxPhases][kMaxTaps];
};

struct TapConvArgs {
    const bf16_t* src;
    void* dst;
    const bf16_t* w;
    const float* in_scale;
    const float* in_shift;
    const float* bias;
    float* stats;
    int32_t N, H, W_src, W_dst;
    int32_t C_src, C_dst;  // padded channel counts (K and N of the implicit GEMM)
    int32_t ld_src, ld_dst;
    int32_t phases, step;
    int32_t m_tiles;
    int32_t h_tiles;  // tapconv2/3: tiles of two / four image rows
    int32_t total_tiles, n_tiles, tiles_per_xcd;  // tapconv3: XCD-aware 1-D grid
    int32_t lds_a_elems;
    int32_t lds_tab_offset;  // byte offset of the per-tap offset table in dynamic LDS
    int32_t flags;
    int32_t sel;  // RV_SEL_* kernel-selection hints of the call (rvTapShape.flags & RV_SEL_MASK)
    // RV_OUT_BNB (backward-data launches): BatchNorm-backward sums of the layer whose output gradient this launch writes
    const bf16_t* bnb_y;
    const float *bnb_scale, *bnb_shift, *bnb_mean, *bnb_invstd;
    float* bnb_partial;  // [tiles][2][C_dst]
    int32_t ld_bnb_y, bnb_flags;
    // RV_OUT_ACCUM: the tensor added to the result -- dst itself (gradient fan-in) or a residual (rv_tap_residual), same pixels as dst
    const bf16_t* res;
    int32_t ld_res;
    int32_t stats_per_wg;  // tapconv6: RV_OUT_STATS / RV_OUT_BNB rows per WORKGROUP (accumulated over its tiles in LDS) instead of per tile
    TapTable tt;
};

// fills tt (except w_tile), *phases and *step; scatter == false: GATHER form, true: SCATTER form
int rv_build_tap_table(const rvTapGeom* g, bool scatter, TapTable* tt, int* phases, int* step);

// second-generation kernel (tapconv2.hip): plan returns false when the layer is not eligible
bool rv_tapconv2_plan(TapConvArgs* a, int* grid_x, int* grid_y, size_t* lds, int* ks);
int rv_tapconv2_launch(const TapConvArgs& a, int grid_x, int grid_y, size_t lds, int ks, hipStream_t stream);

// fourth-generation kernel (tapconv4.hip): 256 x 256 tiles, LDS-DMA staging, counted waits; *stats_rows = rows of the partial-statistics
// buffer (2 per group of workgroups that share a pixel tile when the launch is persistent, else 2 per tile)
bool rv_tapconv4_plan(TapConvArgs* a, int* tiles, size_t* lds, int* bn, int* stats_rows);
int rv_tapconv4_launch(const TapConvArgs& a, size_t lds, int bn, hipStream_t stream);

// fifth-generation kernel (tapconv5.hip): 256 x 256 tiles with the input halo of a channel chunk resident in LDS for all
// taps (multi-tap layers); stats rows = 2 * tiles
bool rv_tapconv5_plan(TapConvArgs* a, int* tiles, size_t* lds, int* bn);
int rv_tapconv5_launch(const TapConvArgs& a, size_t lds, int bn, hipStream_t stream);

// sixth-generation kernel (tapconv6.hip): 512-pixel x 128-channel tiles, 32-channel chunks, input halo resident in LDS across
// the taps; *stats_rows = rows of the partial-statistics buffer (4 per group of workgroups that share a pixel tile when the
// launch is persistent, else 4 per tile), *bnb_rows = rows of the BatchNorm-backward partial sums (one per such group, else per tile)
bool rv_tapconv6_plan(TapConvArgs* a, int* tiles, size_t* lds, int* stats_rows, int* bnb_rows);
int rv_tapconv6_launch(const TapConvArgs& a, size_t lds, hipStream_t stream);

// pointwise streaming GEMM (posconv.hip): 1x1 stride-1 layers C -> C (C = 256 / 128) on plain bf16 tensors, weights in registers,
// pixels streamed through LDS; stats rows = one (sum, sum of squares) pair per workgroup (two for the paired 128 -> 128 form)
bool rv_pointwise_plan(const TapConvArgs* a, bool scatter, int* grid, size_t* lds, int* stats_rows);
int rv_pointwise_launch(const TapConvArgs& a, int grid, size_t lds, hipStream_t stream);

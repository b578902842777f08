/*
 * rv3d.h -- C ABI of librv3d_hip.so: the MI355X (gfx950) implementation of the range-view
 * detector's data-parallel hot path.
 *
 * Drop-in boundary.  The reference (benjaminrwilson/range-view-3d-detection, `torchbox3d`)
 * is pure Python; the native code it reaches on this path is (a) ATen/cuDNN kernels behind
 * torch.nn modules and (b) ONE explicit op-level FFI, `weighted_nms_ext.wnms_gpu`
 * (src/torchbox3d/math/ops/nms.py:161-170).  Every entry point below names the reference
 * interface it replaces (paths relative to the reference root, `src/torchbox3d/` elided
 * where unambiguous).  The Python host (`range_view_3d_detection_amd`) binds these symbols
 * with ctypes; INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - Plain C: pointers + sizes.  All pointers are DEVICE pointers unless named `host_*`.
 *   - The caller (PyTorch's caching allocator in practice) owns every buffer; the library
 *     never allocates, frees or retains a pointer beyond the call (workspaces are passed in).
 *   - Every launch is enqueued on `stream` (a hipStream_t passed as void*); calls are
 *     asynchronous unless documented otherwise (only rv_wnms returns a host count).
 *   - Return value: 0 on success, non-zero on error; rv_last_error() returns a thread-local
 *     description.  Shapes are validated on the host before any launch.
 *   - Activations are NHWC ("pixel-major"): pixel (n,h,w) of a tensor with channel stride
 *     `ld` starts at element ((n*H + h)*W + w)*ld.  Activations / activation gradients are
 *     bf16 (uint16 storage); statistics, parameters, parameter gradients and the final
 *     head outputs are fp32.  Stored channel counts are multiples of 32, zero padded.
 *   - No global mutable state except a read-only device-property cache; safe to use from
 *     one process per GPU (the reference's Lightning-DDP process model).
 */
#ifndef RV3D_H_
#define RV3D_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* rvStream; /* hipStream_t */

int rv_version(void);
const char* rv_last_error(void);

/* ---------------------------------------------------------------------------------------
 * Tap geometry of one convolution-like layer.
 *
 * A layer owns a torch-layout weight T[cu][cv][kh][kw] and relates a "coarse" tensor U
 * (N,H,Wu,cu) to a "fine" tensor V (N,H,Wv,cv) through
 *       (hu, wu, ky, kx)  <->  (hv, wv) = (hu + ky - pad_h, wu*stride_w + kx - pad_w).
 *   nn.Conv2d (via Conv2dSame, nn/modules/conv.py:25-80; torchvision Conv2dNormActivation in
 *   nn/heads/dense_head.py:32-57, nn/stems/__init__.py:41-62):
 *       forward  y = GATHER(x)   (U = y, V = x, T = weight[co][ci][kh][kw])
 *       d/dx     dx = SCATTER(dy)
 *   nn.ConvTranspose2d (nn/blocks/__init__.py:149-156):
 *       forward  y = SCATTER(x)  (U = x, V = y, T = weight[ci][co][kh][kw])
 *       d/dx     dx = GATHER(dy)
 *   weight gradient (both): dT[cu][cv][ky][kx] = sum_{n,hu,wu} U[n,hu,wu,cu] * V[n,hv,wv,cv].
 * Vertical stride is 1 everywhere in this model (nn/backbones/dla.py:37-108).
 * ------------------------------------------------------------------------------------- */
typedef struct {
    int32_t kh, kw;       /* kernel extent */
    int32_t stride_w;     /* 1, 2 or 4 */
    int32_t pad_h, pad_w; /* zero padding (top/left); Conv2dSame: (k-1)/2 */
    int32_t cu, cv;       /* logical channel counts of U and V */
} rvTapGeom;

/* padded channel count used for every stored tensor / packed weight: round up to 32 */
int32_t rv_pad_channels(int32_t c);

/* Bytes of the packed bf16 weight images (gather / scatter form) for a geometry. */
int64_t rv_packed_weight_bytes(const rvTapGeom* g);

/* T (fp32, torch layout [cu][cv][kh][kw]) -> bf16 tap-major images with zero channel padding:
 *   gather_w  [tap = ky*kw+kx][cu_pad][cv_pad]               (K dim = cv contiguous)
 *   scatter_w [phase][tap-in-phase][cv_pad][cu_pad]          (K dim = cu contiguous)
 * Either output may be NULL.  Replaces nothing in the reference (cuDNN re-lays weights
 * internally); it is what keeps `state_dict()` in the reference's OIHW layout. */
int rv_pack_weight(const rvTapGeom* g, const float* T, void* gather_w, void* scatter_w, rvStream stream);

/* All layers of a model in ONE launch (every packed image is stale after an optimiser step; ~160 layers on the rv-* models).
 * The caller keeps a table of 2 entries per layer (gather image, scatter image; rv_pack_batch_entry_bytes() each): filled on
 * the HOST by rv_pack_batch_fill (pointers are device pointers), copied to the device once, and re-used every step for as
 * long as the parameter and image buffers stay where they are. */
int64_t rv_pack_batch_entry_bytes(void);
int rv_pack_batch_fill(const rvTapGeom* g, const float* T, void* gather_w, void* scatter_w, void* host_entries);
int rv_pack_batch(const void* dev_table, int32_t n_entries, rvStream stream);

/* FOLDED form of a stride-s layer (stride_w 2 or 4): the s fine pixels of a coarse pixel are contiguous in NHWC, so the fine
 * tensor V (N,H,s*Wu,cv_pad) read as (N,H,Wu,s*cv_pad) turns U = GATHER_s(V) -- a strided Conv2d's forward, a
 * ConvTranspose2d's backward-data -- and the layer's weight gradient into STRIDE-1 operations with kw' <= 3 column taps over
 * s*cv_pad channels, which the LDS-DMA kernels take (the strided forms run on the generic kernel at a third of the rate; the
 * zero entries of the folded weight cost 1.3-1.5x the FLOPs).  rv_fold_geom gives the folded geometry (use it with
 * rv_tap_gather / rv_tap_wgrad, Wv = Wu, ld of V multiplied by s; requires ld(V) == cv_pad), rv_pack_weight_folded its packed
 * gather image (rv_packed_weight_bytes(gf) bytes), rv_unfold_weight_grad maps a folded weight gradient (torch layout
 * [cu][s*cv_pad][kh][kw']) back to dT[cu][cv][kh][kw]. */
int rv_fold_geom(const rvTapGeom* g, rvTapGeom* folded);
int rv_pack_weight_folded(const rvTapGeom* g, const float* T, void* gather_w_folded, rvStream stream);
int rv_pack_batch_fill_folded(const rvTapGeom* g, const float* T, void* gather_w_folded, void* host_entry);
int rv_unfold_weight_grad(const rvTapGeom* g, const float* dT_folded, float* dT, int32_t accumulate, rvStream stream);

/* fp32 packed weight gradient [kh*kw][cu_pad][cv_pad] -> accumulate/store into torch layout
 * dT[cu][cv][kh][kw] (fp32).  accumulate != 0: dT += value. */
int rv_unpack_weight_grad(const rvTapGeom* g, const float* packed, float* dT, int32_t accumulate, rvStream stream);

/* flags for rv_tap_gather / rv_tap_scatter */
#define RV_IN_AFFINE 1   /* operand = in_scale[c]*x + in_shift[c] (folded BatchNorm) */
#define RV_IN_RELU 2     /* ... followed by ReLU; padding positions stay exactly 0 */
#define RV_OUT_F32 4     /* dst is fp32 (final head convs); default bf16 */
#define RV_OUT_BIAS 8    /* dst += bias[c] */
#define RV_OUT_STATS 16  /* write per-block partial sum / sum-of-squares of the fp32 result */
#define RV_OUT_ACCUM 32  /* dst += result (gradient fan-in); bf16 dst only */
#define RV_OUT_RELU 64   /* with RV_OUT_BIAS: dst = max(result + bias[c], 0) -- inference: an eval-mode BatchNorm folded into the
                          * weights (w * gamma / sqrt(var + eps)) and the bias (beta - mean * scale), ReLU in the epilogue, so that
                          * conv -> BatchNorm -> ReLU is ONE launch and one write (cuDNN conv + batch_norm + relu_ in the reference) */
#define RV_OUT_RES_RELU 256 /* rv_tap_residual only: ReLU AFTER the residual has been added (RV_OUT_RELU: before) */
#define RV_WGRAD_TORCH_LAYOUT 128 /* rv_tap_wgrad only: dT_packed receives the torch layout dT[cu][cv][kh][kw] (cu*cv*kh*kw fp32,
                                  * no padding) straight from the split-K reduction -- no rv_unpack_weight_grad pass */
/* Kernel-selection hints, also in rvTapShape.flags: speed heuristics only, never results.  PER CALL -- the library keeps no
 * mutable state; the parity tests use them so that crops the CPU oracle can afford run the kernels of the
 * full-size sweeps, and to pin a kernel generation. */
#define RV_SEL_SMALL_GRIDS (1 << 20)  /* the LDS-DMA tap-convs of generations 4 and 5 also take layers with fewer tiles than CUs */
#define RV_SEL_SMALL_GRIDS6 (1 << 21) /* ... and generation 6 */
#define RV_SEL_NO_GEN6 (1 << 22)      /* do not select generation 6 */
#define RV_SEL_NO_GEN5 (1 << 23)      /* do not select generations 5 and 6 (multi-tap layers stay on generation 4) */
#define RV_SEL_NO_POINTWISE (1 << 24) /* 1x1 C -> C layers stay on the tiled kernels (generation 7 = the pointwise streaming GEMM, round 6) */
#define RV_SEL_NO_POINTWISE_BWD (1 << 25) /* ... only its SCATTER-form (backward-data) launches stay on the tiled kernels (diagnostics: profiles/r06_ab_notes.md section 4) */
#define RV_SEL_MASK (63 << 20)

typedef struct {
    int32_t N, H, Wu, Wv; /* U is (N,H,Wu), V is (N,H,Wv) */
    int32_t ld_src, ld_dst; /* channel strides (elements) of the source / destination tensors */
    int32_t flags;
} rvTapShape;

/* Rows of the partial-statistics buffer ([rows][2][c_pad] fp32) a launch with RV_OUT_STATS
 * writes; `scatter` selects the SCATTER form.
 * The planning entry points (this one, rv_tap_launch_info, rv_tap_wgrad_info, rv_tap_wgrad_workspace_bytes, rv_tap_bnb_rows) size
 * persistent grids by the compute-unit count of the CURRENT device (hipGetDevice + one cached attribute query per device): they
 * launch nothing but they do initialise the HIP runtime -- call them after any fork and after GPU_MAX_HW_QUEUES is in the
 * environment; without a visible device they plan for 256 compute units. */
int32_t rv_tap_stats_rows(const rvTapGeom* g, const rvTapShape* s, int32_t scatter);
/* Launch plan the library picks for a tap op: info = {kernel generation, variant, grid.x, grid.y}:
 * generation 1 = tapconv_kernel<MT,NT> (variant = 16*MT + NT, block tile 32*MT pixels x 32*NT channels),
 * generation 2 = tapconv2_kernel<KS> (variant = KS, block tile 2 rows x 64 columns x 128 channels, 32*KS-channel
 * chunks), generations 4 / 5 / 6 = the LDS-DMA kernels (variant = channels per workgroup, grid.x = pixel tiles, grid.y = channel
 * tiles), generation 7 = the pointwise streaming GEMM (1x1 stride-1 C -> C layers, C = 256 / 128: variant = C, grid.x = persistent
 * workgroups).  Used by bench.py to label per-kernel timings. */
int rv_tap_launch_info(const rvTapGeom* g, const rvTapShape* s, int32_t scatter, int32_t* host_info);
/* Every partial-statistics buffer handed to rv_bn_finalize / rv_bn_bwd_finalize must have room
 * for this many extra rows after its `rows` partial rows (second-stage reduction scratch). */
#define RV_STATS_SCRATCH_ROWS 128

/* U = GATHER(V):  U[n,h,wu,cu] = sum_{ky,kx,cv} T[cu][cv][ky][kx] * f(V[n, h+ky-pad_h, wu*s+kx-pad_w, cv]).
 * Replaces nn.Conv2d forward (cuDNN/ATen conv2d; nn/modules/conv.py:80) including the
 * F.pad copy of Conv2dSame (:79), the preceding BatchNorm2d+ReLU when RV_IN_AFFINE|RV_IN_RELU
 * (nn/blocks/__init__.py:41-42), and ATen's conv_transpose2d backward-data. */
int rv_tap_gather(const rvTapGeom* g, const rvTapShape* s, const void* V, const float* in_scale,
                  const float* in_shift, const void* gather_w, const float* bias, void* U,
                  float* stats_partial, rvStream stream);

/* V = SCATTER(U): V[n,h,wv,cv] = sum over (ky,kx,wu) with wu*s+kx-pad_w == wv of
 *                 T[cu][cv][ky][kx] * f(U[n, h-ky+pad_h, wu, cu]).
 * Replaces nn.ConvTranspose2d forward (ATen conv_transpose2d; nn/blocks/__init__.py:176)
 * and cuDNN's conv2d backward-data. */
int rv_tap_scatter(const rvTapGeom* g, const rvTapShape* s, const void* U, const float* in_scale,
                   const float* in_shift, const void* scatter_w, const float* bias, void* V,
                   float* stats_partial, rvStream stream);

/* Inference: a tap op (scatter != 0: the SCATTER form) whose epilogue adds a residual tensor of the output's pixels,
 *   dst = [relu]( [relu]( op(src) + bias ) + res ),   s->flags: RV_OUT_BIAS, RV_OUT_RELU (inner), RV_OUT_RES_RELU (outer),
 * with the eval-mode BatchNorm folded into w / bias: the block outputs relu_(net(x) + proj(x)) of BasicBlock.forward
 * (nn/blocks/__init__.py:68-81) and x1 + relu(bn(convT(x2))) of AggregationBlock.forward (:165-182) leave the conv's own
 * launch -- no separate element-wise pass (ATen add + relu_ in the reference).  The result is the one the separate pass
 * produces, bit for bit: the conv result is rounded to the storage type before the residual is added, as a stored tensor
 * would have been.  res: bf16/fp16 [pixels of dst][ld_res]; plain operands only (no RV_IN_*, RV_OUT_STATS, RV_OUT_F32). */
int rv_tap_residual(const rvTapGeom* g, const rvTapShape* s, int32_t scatter, const void* src, const void* w, const float* bias,
                    const void* res, int32_t ld_res, void* dst, rvStream stream);

/* Backward-data launch that ALSO forms the BatchNorm-backward sums of the layer whose output gradient it writes
 * (conv -> BatchNorm(+ReLU) -> THIS conv: autograd's native_batch_norm_backward reduce over (dOut, y), fused into the
 * epilogue of conv2d backward-data).  dx = the gradient w.r.t. the (activated) BatchNorm output, written as by
 * rv_tap_gather / rv_tap_scatter (scatter != 0: the SCATTER form) without RV_OUT_ACCUM; from the bf16 values it stores,
 *   g = dx * [scale*y+shift > 0 if flags & RV_BNB_RELU_Z],   partial[row][0][c] = sum g,  partial[row][1][c] = sum g * (y-mean)*invstd
 * over the row's pixels -- the layout rv_bn_bwd_finalize takes (rv_bn_bwd_reduce then is not needed).
 * rv_tap_bnb_rows: partial rows such a launch writes; 0 = the kernel (g, s) selects has no such epilogue (generations 5 and 6
 * have it, for non-accumulating launches): use rv_tap_gather/scatter + rv_bn_bwd_reduce. */
typedef struct rvBnbEpilogue {
    const void* y;      /* bf16 NHWC pre-BatchNorm conv output of the destination layer, same pixels as dx */
    int32_t ld_y;       /* its channel stride (elements) */
    int32_t flags;      /* RV_BNB_RELU_Z */
    const float* scale; /* folded BatchNorm: gamma*invstd, beta - mean*scale (the ReLU mask) */
    const float* shift;
    const float* mean;
    const float* invstd;
    float* partial;     /* [rows + RV_STATS_SCRATCH_ROWS][2][c_pad] fp32 */
} rvBnbEpilogue;
int32_t rv_tap_bnb_rows(const rvTapGeom* g, const rvTapShape* s, int32_t scatter);
int rv_tap_data_grad_bnb(const rvTapGeom* g, const rvTapShape* s, int32_t scatter, const void* dout, const void* w, void* dx,
                         const rvBnbEpilogue* e, rvStream stream);

/* Weight gradient.  dT_packed[tap][cu_pad][cv_pad] (fp32) = sum_{n,h,wu} U * f(V) (shifted).
 * `workspace` holds split-K partial slabs; rv_tap_wgrad_workspace_bytes() sizes it.
 * V may carry the same folded BN+ReLU as in the forward (RV_IN_AFFINE|RV_IN_RELU in
 * s->flags applies to V when v_affine != 0, else to U).
 * Replaces cuDNN conv2d backward-weight / ATen conv_transpose2d backward-weight. */
int64_t rv_tap_wgrad_workspace_bytes(const rvTapGeom* g, const rvTapShape* s);
/* host-side introspection (bench / tests): info[0] = kernel generation that rv_tap_wgrad will launch for (g, s)
 * (1 generic, 2 register-staged 3-tap groups, 3 LDS-DMA ring), info[1] = split-K factor, info[2] = workgroups */
int rv_tap_wgrad_info(const rvTapGeom* g, const rvTapShape* s, int32_t* host_info);
int rv_tap_wgrad(const rvTapGeom* g, const rvTapShape* s, const void* U, int32_t ld_u, const void* V, int32_t ld_v,
                 const float* in_scale, const float* in_shift, int32_t v_affine, float* dT_packed,
                 void* workspace, rvStream stream);

/* ---------------------------------------------------------------------------------------
 * BatchNorm2d (nn.BatchNorm2d train/eval; nn/blocks/__init__.py:41,51,63,158; torchvision
 * Conv2dNormActivation norm layer).  eps / momentum are torch defaults passed by the host.
 * ------------------------------------------------------------------------------------- */
/* partial[rows][2][c] (sum, sum of squares) -> batch mean / biased var -> folded affine
 * scale = gamma*invstd, shift = beta - mean*scale; saves mean/invstd for backward and
 * updates running_mean / running_var (unbiased) in place when they are non-NULL. */
/* Column sums of `rows` partial rows ([rows][cols] fp32, accumulated in fp64) -> out[cols]; `partial` must have the
 * scratch rows of the convention above behind it.  SyncBN: totals written straight into the all-reduce buffer. */
int rv_reduce_rows(const float* partial, int32_t rows, int32_t cols, float* out, rvStream stream);
/* The same in the layout of one SyncBN collective: out[0 : cols] = the totals, out[cols] = `count` (this rank's element count,
 * summed by the all-reduce with the totals); out_copy (may be NULL) receives the LOCAL totals too -- in the backward pass
 * they are this rank's (dbeta, dgamma), which stay local (DDP averages parameter gradients).  One launch. */
int rv_reduce_rows_count(const float* partial, int32_t rows, int32_t cols, float count, float* out, float* out_copy, rvStream stream);
/* count < 0 (SyncBN, rows == 1): the element count is read from the device, partial[2*c] (fp32), where it travelled with
 * the all-reduced totals -- ranks may hold different numbers of pixels.  Same convention in rv_bn_bwd_finalize. */
int rv_bn_finalize(const float* partial, int32_t rows, int32_t c, int64_t count, const float* gamma,
                   const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                   float* scale, float* shift, float* mean, float* invstd, rvStream stream);
/* eval mode: scale/shift from the running statistics */
int rv_bn_fold_eval(int32_t c, const float* gamma, const float* beta, const float* running_mean,
                    const float* running_var, float eps, float* scale, float* shift, rvStream stream);

/* out = relu?( fa(a) + fb(b) ), f*(x) = relu?(scale*x + shift) per channel when the scale
 * pointer is non-NULL, identity otherwise; b may be NULL.  bf16 in / bf16 out.
 * Replaces `F.relu_(self.net(x) + residual)` (nn/blocks/__init__.py:81), `x_1 + x_2` after
 * BN+ReLU (:177-180) and standalone BatchNorm2d+ReLU applications. */
#define RV_EW_RELU_A 1
#define RV_EW_RELU_B 2
#define RV_EW_RELU_OUT 4
int rv_ew_combine(int64_t pixels, int32_t c, const void* a, int32_t ld_a, const float* a_scale,
                  const float* a_shift, const void* b, int32_t ld_b, const float* b_scale,
                  const float* b_shift, void* out, int32_t ld_out, int32_t flags, rvStream stream);

/* BatchNorm backward, fused with the ReLU masks around it.
 *   g  = dOut * [OUT > 0 if out != NULL] * [scale*y+shift > 0 if RV_BNB_RELU_Z]
 *   pass 1 (reduce): partial[rows][2][c] = (sum g, sum g*xhat), xhat = (y-mean)*invstd
 *   finalize       : dgamma = sum g*xhat, dbeta = sum g (accumulated into the fp32 grads),
 *                    coef[0][c] = gamma*invstd, coef[1][c] = mean(g), coef[2][c] = mean(g*xhat)
 *   pass 2 (apply) : dY = coef0 * (g - coef1 - xhat*coef2)  (bf16), and optionally
 *                    dRes (+)= g  (identity residual branch).
 * Replaces cuDNN BatchNorm backward + ReLU backward + the add's gradient fan-out. */
#define RV_BNB_RELU_Z 1
#define RV_BNB_RES_ACCUM 2
#define RV_BNB_Y_FROM_INPUT 4 /* rv_bn_bwd_smallk*: y (may be NULL) is recomputed as W v from the conv input and w_packed */
int32_t rv_bn_bwd_rows(int64_t pixels);
int rv_bn_bwd_reduce(int64_t pixels, int32_t c, const void* dout, int32_t ld_dout, const void* out, int32_t ld_out,
                     const void* y, int32_t ld_y, const float* scale, const float* shift, const float* mean,
                     const float* invstd, int32_t flags, float* partial, rvStream stream);
/* The reduce pass for the TWO BatchNorms under one block sum out = relu(bn_a(ya) + bn_b(yb)) -- a BasicBlock with a projection
 * (nn/blocks/__init__.py:68-81: net(x) + projection_block(x)) -- in one launch: both take g = dOut * [out > 0]; partial_a /
 * partial_b receive (sum g, sum g * xhat_a) and (sum g, sum g * xhat_b) in rv_bn_bwd_reduce's layout ([rv_bn_bwd_rows(pixels) +
 * RV_STATS_SCRATCH_ROWS][2][c]).  Four tensor reads where two rv_bn_bwd_reduce launches take six. */
int rv_bn_bwd_reduce_pair(int64_t pixels, int32_t c, const void* dout, int32_t ld_dout, const void* out, int32_t ld_out,
                          const void* ya, int32_t ld_ya, const float* mean_a, const float* invstd_a, const void* yb, int32_t ld_yb,
                          const float* mean_b, const float* invstd_b, float* partial_a, float* partial_b, rvStream stream);
/* ... and their apply pass: dYa = coef_a0 * (g - coef_a1 - xhat_a * coef_a2), dYb likewise, g formed once (coef_* from
 * rv_bn_bwd_finalize over partial_a / partial_b).  Four reads and two writes where two rv_bn_bwd_apply launches take six and two. */
int rv_bn_bwd_apply_pair(int64_t pixels, int32_t c, const void* dout, int32_t ld_dout, const void* out, int32_t ld_out,
                         const void* ya, int32_t ld_ya, const float* mean_a, const float* invstd_a, const float* coef_a, void* dya,
                         int32_t ld_dya, const void* yb, int32_t ld_yb, const float* mean_b, const float* invstd_b,
                         const float* coef_b, void* dyb, int32_t ld_dyb, rvStream stream);
int rv_bn_bwd_finalize(const float* partial, int32_t rows, int32_t c, int64_t count, const float* gamma,
                       const float* invstd, float* dgamma, float* dbeta, int32_t accumulate, float* coef,
                       rvStream stream);
int rv_bn_bwd_apply(int64_t pixels, int32_t c, const void* dout, int32_t ld_dout, const void* out, int32_t ld_out,
                    const void* y, int32_t ld_y, const float* scale, const float* shift, const float* mean,
                    const float* invstd, const float* coef, int32_t flags, void* dy, int32_t ld_dy, void* dres,
                    int32_t ld_dres, rvStream stream);
/* Backward of a tower's FINAL 1x1 conv (c -> n_out <= 32 channels; nn/heads/dense_head.py:44-57, 74-76) fused with the BatchNorm
 * (+ReLU) backward of the conv -> BatchNorm -> ReLU unit in front of it: the input gradient dA = W^T dY (a K = 32 GEMM of the tiny
 * output gradient) is RECOMPUTED in both passes instead of being stored and read back twice --
 *   _sums : g = dA * [scale*y+shift > 0 if relu], partial[row][0][c] = sum g, partial[row][1][c] = sum g * (y-mean)*invstd over
 *           rv_head_final_bwd_rows(pixels) <= 512 pixel ranges (the rows rv_bn_bwd_finalize takes; + RV_STATS_SCRATCH_ROWS);
 *           dw_partial (may be NULL): [rows][32][c] fp32 partial WEIGHT gradients of the final conv, dW[o][c] = sum_px dY[px][o] *
 *           relu?(scale*y+shift)[px][c] over the row's pixels -- sum the rows in order (rv_reduce_rows with cols = 32 c) for the
 *           gradient in the parameter's own layout [n_out][c][1][1] (rows o >= n_out are zero)
 *   _apply: dy = coef0 * (g - coef1 - xhat * coef2)  (bf16; coef from rv_bn_bwd_finalize)
 * y: raw bf16 output of the unit's conv [pixels][ld_y]; dY: the final conv's output gradient as bf16 [pixels][ld_dy >= 32] with
 * channels n_out..31 zero; w_scatter: the final conv's packed scatter image ([c][32]); c % 256 == 0.
 * Replaces ATen conv2d backward-data + conv2d backward-weight + native_batch_norm_backward + threshold_backward for that pair of
 * layers (three transfers of a c-channel tensor instead of seven). */
int32_t rv_head_final_bwd_rows(int64_t pixels);
int rv_head_final_bwd_sums(int64_t pixels, int32_t c, const void* y, int32_t ld_y, const void* dY, int32_t ld_dy, const void* w_scatter,
                           const float* scale, const float* shift, const float* mean, const float* invstd, int32_t relu, float* partial,
                           float* dw_partial, rvStream stream);
int rv_head_final_bwd_apply(int64_t pixels, int32_t c, const void* y, int32_t ld_y, const void* dY, int32_t ld_dy, const void* w_scatter,
                            const float* scale, const float* shift, const float* mean, const float* invstd, int32_t relu, const float* coef,
                            void* dy, int32_t ld_out, rvStream stream);
/* Small-K layers whose input needs no gradient (1x1 conv with cin <= 8 followed by BatchNorm: the stem's 3 -> C positional
 * conv, the 5/6 -> C feature projections): BatchNorm backward AND the conv's weight gradient from one pass over
 * (dOut, y, v) -- dy is never written.  v = the conv input (bf16 NHWC, >= 8 stored channels), w_packed = the layer's packed
 * gather image (bf16 [c][ld_w]); dW is fp32 [c][cin].  Replaces cuDNN BatchNorm backward + conv2d backward-weight. */
int64_t rv_bn_bwd_smallk_workspace_bytes(int64_t pixels, int32_t c, int32_t cin);
int rv_bn_bwd_smallk(int64_t pixels, int32_t c, const void* dout, int32_t ld_dout, const void* out, int32_t ld_out,
                     const void* y, int32_t ld_y, const float* scale, const float* shift, const float* mean,
                     const float* invstd, int32_t flags, const void* v, int32_t ld_v, int32_t cin, const void* w_packed,
                     int32_t ld_w, const float* gamma, const float* stat_mean, const float* stat_invstd, int64_t count,
                     float* dgamma, float* dbeta, float* dW, void* workspace, rvStream stream);
/* The same in two phases, for SyncBN: (A) this rank's sums -- sums[(2 + cin_pad) * c] = planes (sum g, sum g*xhat,
 * sum g*v_d), moms[cin_pad + cin_pad^2] = (sum v, sum v v^T) -- then the caller all-reduces sums[0 : 2c] into global_s01
 * and (B) forms the gradients (global_s01 == NULL: single rank). */
int rv_bn_bwd_smallk_sums(int64_t pixels, int32_t c, const void* dout, int32_t ld_dout, const void* out, int32_t ld_out,
                          const void* y, int32_t ld_y, const float* scale, const float* shift, const float* mean,
                          const float* invstd, int32_t flags, const void* v, int32_t ld_v, int32_t cin, const void* w_packed,
                          int32_t ld_w, double* sums, double* moms, void* workspace, rvStream stream);
int rv_bn_bwd_smallk_from_sums(int32_t c, int32_t cin, const double* sums, const double* moms, const double* global_s01,
                               const void* w_packed, int32_t ld_w, const float* gamma, const float* stat_mean,
                               const float* stat_invstd, int64_t count, float* dgamma, float* dbeta, float* dW,
                               rvStream stream);
/* Forward of the same layers: h = relu?(BatchNorm(W v)) as ONE element-wise pass; in training the batch statistics come
 * in closed form from the data moments (m1 = sum v, M2 = sum v v^T: rv_smallk_moments writes cin_pad + cin_pad^2 doubles,
 * cin_pad = 4 or 8; the caller all-reduces them under SyncBN), so neither the raw conv output nor a statistics pass over
 * it exists.  moments == NULL: eval, scale/shift are inputs (rv_bn_fold_eval).  h == NULL: statistics only (the caller
 * applies them itself: rv_pos_forward).  For the backward of such a layer call
 * rv_bn_bwd_smallk with RV_BNB_Y_FROM_INPUT (y = NULL: the raw output is recomputed from v, 16 bytes per pixel instead of
 * 2 C) and the layer's own scale / shift / mean / invstd; or, without the flag, with y = h, scale = 1, shift = 0,
 * mean = beta, invstd = 1/gamma, stat_* = the batch statistics (xhat rebuilt from the activated output).
 * Replaces cuDNN conv2d + BatchNorm + ReLU (nn/stems/__init__.py:40-57, nn/blocks/__init__.py:38-51 on 5/6-channel input). */
int64_t rv_smallk_forward_workspace_bytes(int32_t cin);
int rv_smallk_moments(const void* v, int32_t ld_v, int64_t pixels, int32_t cin, double* moments, void* workspace,
                      rvStream stream);
int rv_smallk_forward(const void* v, int32_t ld_v, int64_t pixels, int32_t cin, const void* w_packed, int32_t ld_w,
                      int32_t c, const double* moments, int64_t count, const float* gamma, const float* beta, float eps,
                      float momentum, float* running_mean, float* running_var, float* scale, float* shift, float* mean,
                      float* invstd, int32_t relu, void* h, int32_t ld_h, rvStream stream);
/* gradient of rv_ew_combine's plain (non-BN) inputs: d (+)= dOut * [OUT > 0 if out != NULL] */
int rv_ew_mask_grad(int64_t pixels, int32_t c, const void* dout, int32_t ld_dout, const void* out, int32_t ld_out,
                    void* d, int32_t ld_d, int32_t accumulate, rvStream stream);

/* RangePartition stem (nn/stems/__init__.py:88-135, the stem RangeNet builds for stem_type RANGE_PARTITION, nn/backbones/dla.py:164-171):
 * `features = (partitions[:, :, None] * features[:, None]).flatten(1, 2) * mask` with partitions = (||cart|| >= lower) & (||cart|| <= upper)
 * as the 16-bit NHWC operand of the projecting BasicBlock -- channel band * C + c, zeros in the padding channels.  features / cart fp32
 * NCHW, mask one byte per pixel; `lower` / `upper`: `bands` host floats (the module's lower_bounds / upper_bounds parameters). */
int rv_range_partition(const float* features_nchw, const float* cart_nchw, const uint8_t* mask, int32_t N, int32_t C, int32_t H, int32_t W,
                       const float* lower, const float* upper, int32_t bands, void* dst, int32_t ld_dst, rvStream stream);

/* ---------------------------------------------------------------------------------------
 * Layout conversion at the module boundary (the reference's tensors are NCHW fp32).
 * ------------------------------------------------------------------------------------- */
int rv_nchw_f32_to_nhwc_bf16(const float* src, int32_t N, int32_t C, int32_t H, int32_t W, void* dst, int32_t ld_dst,
                             int32_t c_offset, rvStream stream);
int rv_nhwc_bf16_to_nchw_f32(const void* src, int32_t ld_src, int32_t c_offset, int32_t N, int32_t C, int32_t H,
                             int32_t W, float* dst, rvStream stream);
int rv_nhwc_f32_to_nchw_f32(const float* src, int32_t ld_src, int32_t N, int32_t C, int32_t H, int32_t W, float* dst,
                            rvStream stream);
int rv_nchw_f32_to_nhwc_f32(const float* src, int32_t N, int32_t C, int32_t H, int32_t W, float* dst, int32_t ld_dst,
                            rvStream stream);

/* ---------------------------------------------------------------------------------------
 * MetaKernel stem (nn/stems/__init__.py:64-85): F.unfold of features and of `cart`,
 * relative coordinates, positional MLP, element-wise product.  The 9x unfolded tensors of
 * the reference are never written to memory.
 * ------------------------------------------------------------------------------------- */
/* rel[n,h,w,tap,0:3] = cart[n,h+dy,w+dx,:] (0 outside) - cart[n,h,w,:], written as a bf16
 * NHWC tensor (N, H, W*9, 32) whose channels 3..31 are zero (operand of the 3->C 1x1 conv).
 * `cart` is NCHW fp32 (B,3,H,W) exactly as the reference's batch dict holds it. */
int rv_meta_relative(const float* cart_nchw, int32_t N, int32_t H, int32_t W, void* rel, rvStream stream);
/* geo[n,h,w, tap*C + c] = relu(scale[c]*pos[n,h,w,tap,c] + shift[c]) * feat[n,h+dy,w+dx,c]
 * (zero outside the image).  The reference's channel order c*9+tap (F.unfold) is absorbed
 * into the packed weight of the 9C->C fusion conv. */
int rv_meta_modulate(const void* pos_raw, const float* scale, const float* shift, const void* feat, int32_t ld_feat,
                     int32_t N, int32_t H, int32_t W, int32_t C, void* geo, rvStream stream);
/* backward of rv_meta_modulate: dpos_act = dgeo * feat_nbr (then ReLU/BN backward via
 * rv_bn_bwd_*), dfeat[n,h',w',c] += sum_tap dgeo * pos_act. */
int rv_meta_modulate_bwd(const void* dgeo, const void* pos_raw, const float* scale, const float* shift,
                         const void* feat, int32_t ld_feat, int32_t N, int32_t H, int32_t W, int32_t C,
                         void* dpos_act, void* dfeat, int32_t ld_dfeat, rvStream stream);
/* The same backward FUSED with the BatchNorm(+ReLU) backward of the positional layer whose output `pos_raw` is
 * (training path; nn/stems/__init__.py:80-84 differentiated through Conv2dNormActivation's BatchNorm2d + ReLU):
 *   z = dgeo * feat_nbr * [relu passed]  (never written),  S0 = sum z, S1 = sum z*xhat,  dfeat as above.
 * _sums: one pass over (dgeo, pos_raw): dfeat and rv_meta_bwd_rows(N,H,W) partial rows [row][2][C] of (S0, S1) in the
 *        layout rv_bn_bwd_finalize reads (allocate rows + RV_STATS_SCRATCH_ROWS).
 * _apply: dy[n,h,w,tap,c] = coef0 (z - coef1 - xhat coef2) with `coef` from rv_bn_bwd_finalize.
 * Two passes over the 9x-grid tensors instead of four, and z stays in registers. */
int32_t rv_meta_bwd_rows(int32_t N, int32_t H, int32_t W);
int rv_meta_modulate_bwd_sums(const void* dgeo, const void* pos_raw, const float* scale, const float* shift,
                              const float* mean, const float* invstd, const void* feat, int32_t ld_feat, int32_t N,
                              int32_t H, int32_t W, int32_t C, void* dfeat, int32_t ld_dfeat, float* partial,
                              rvStream stream);
int rv_meta_modulate_bwd_apply(const void* dgeo, const void* pos_raw, const float* scale, const float* shift,
                               const float* mean, const float* invstd, const float* coef, const void* feat,
                               int32_t ld_feat, int32_t N, int32_t H, int32_t W, int32_t C, void* dy, rvStream stream);

/* The two positional layers of the MetaKernel stem (nn/stems/__init__.py:41-49, 80: Conv2dNormActivation(3, C, 1) ->
 * Conv2dNormActivation(C, C, 1) on the 9x neighbour grid) as ONE persistent streaming GEMM, C = 256 (rv-av2) or 128 (rv-waymo):
 *   h1 = relu(scale1 * (W1 rel) + shift1)   generated in the K-operand staging from `rel` (bf16 [pixels][ld_rel], cin <= 3
 *                                           channels used) and written once (the second layer's weight gradient reads it),
 *   y2 = W2 h1                              raw bf16, with fp32 (sum, sum of squares) rows of the accumulators in
 *                                           stats_partial [rv_pos_forward_rows(pixels) + RV_STATS_SCRATCH_ROWS][2][C] (NULL: eval).
 * scale1 / shift1: the first layer's folded BatchNorm (rv_smallk_forward with h == NULL forms them in closed form).
 * Replaces rv_smallk_forward's apply pass + rv_tap_gather of the second layer (cuDNN conv2d x2 + BatchNorm + ReLU). */
int32_t rv_pos_forward_rows(int64_t pixels);
int rv_pos_forward(const void* rel, int32_t ld_rel, int32_t cin, int64_t pixels, const void* w1_packed, int32_t ld_w1,
                   const float* scale1, const float* shift1, const void* w2_packed, int32_t c, void* h1, void* y2,
                   float* stats_partial, rvStream stream);

/* Inference form of the same kernel with the modulation of MetaKernel.forward (nn/stems/__init__.py:80-83: positional
 * weights times the unfolded neighbour features) in its epilogue: row 9 p + k of the 9x grid is neighbour k of pixel p, and
 *   geo[p][k*c + ch] = relu(scale2[ch] * y2[9 p + k][ch] + shift2[ch]) * feat[neighbour k of p][ch]   (0 outside the image)
 * is what the kernel stores -- neither h1 nor y2 reaches memory (2 x 2.4 GB at 4 x 64 x 2048 x 256), and the separate
 * rv_meta_modulate pass (read 2.4 GB, write 2.4 GB) disappears.  y2 is rounded to the storage type before the BatchNorm, as the
 * stored tensor was: the result equals rv_pos_forward + rv_meta_modulate bit for bit.  scale2 / shift2: the second layer's
 * eval-mode BatchNorm (rv_bn_fold_eval); feat: bf16/fp16 [N*H*W][ld_feat]; geo: [N*H*W][9*c].  W >= 32, 9 N H W < 2^31. */
int rv_pos_modulate_forward(const void* rel, int32_t ld_rel, int32_t cin, const void* w1_packed, int32_t ld_w1, const float* scale1,
                            const float* shift1, const void* w2_packed, int32_t c, const float* scale2, const float* shift2,
                            const void* feat, int32_t ld_feat, int32_t N, int32_t H, int32_t W, void* geo, rvStream stream);

/* Backward of the same pair: the second layer's backward-data GEMM dh1 = dy2 W2 fused with phase (A) of the first layer's
 * small-K BatchNorm backward (rv_bn_bwd_smallk_sums with RV_BNB_Y_FROM_INPUT): dh1 is consumed in registers and never
 * written -- 2.4 GB less to store and 2.4 GB less to read back at 4 x 64 x 2048.  w2_scatter = the second layer's packed
 * scatter image; sums / moms / workspace as rv_bn_bwd_smallk_sums (cin_pad = 4); continue with rv_bn_bwd_smallk_from_sums.
 * Replaces rv_tap_scatter of the second layer + rv_bn_bwd_smallk_sums of the first (ATen conv backward-data + BatchNorm
 * backward + conv backward-weight). */
int rv_pos_backward_sums(int64_t pixels, int32_t c, const void* dy2, const void* w2_scatter, const void* rel, int32_t ld_rel,
                         int32_t cin, const void* w1_packed, int32_t ld_w1, const float* scale1, const float* shift1,
                         const float* mean1, const float* invstd1, double* sums, double* moms, void* workspace, rvStream stream);

/* ---------------------------------------------------------------------------------------
 * Optimiser step of the recipe (nn/meta/arch.py:57 -> torch.optim.AdamW via conf/model/range_view.yaml:52-55; gradient
 * clipping = Lightning's gradient_clip_val 35.0, conf/trainer/train.yaml) for ALL parameters in two launches.
 * tensors: n_tensors x {float* p; const float* g; float* m; float* v; int64 n} (device table), chunks: n_chunks x
 * {int32 tensor; int32 chunk} with chunk < ceil(n / rv_optim_chunk_elems()), partial: n_chunks floats of scratch.
 * Arithmetic = torch/optim/adamw.py (foreach, non-capturable) in fp32, `step` = the 1-based step count of the bias
 * corrections; max_norm > 0: g is scaled by min(1, max_norm / (||g||_2 + 1e-6)) on the fly (torch.nn.utils.clip_grad_norm_;
 * the stored gradients stay untouched), total_norm (optional) receives ||g||_2.
 * Replaces ATen's foreach norm / mul / lerp / addcmul / sqrt / div / addcdiv launches.
 * ------------------------------------------------------------------------------------- */
int32_t rv_optim_chunk_elems(void);
int rv_adamw_step(const void* tensors, const void* chunks, int32_t n_chunks, float* partial, double lr, double beta1,
                  double beta2, double eps, double weight_decay, int64_t step, double max_norm, float* total_norm,
                  rvStream stream);

/* ---------------------------------------------------------------------------------------
 * Decoder (nn/decoders/range_decoder.py:29-156, math/ops/coding.py:79-144,
 * math/linalg/lie/SO3.py:122-134).
 * ------------------------------------------------------------------------------------- */
/* Per pixel: score = max_c sigmoid(logit_c)*mask (ties -> lowest class), category = argmax,
 * box = decode_range_view(regressands, cart) evaluated in fp64 and rounded to fp32.
 * Inputs are NCHW fp32 (the module boundary layout).  With `n_bands` > 0 the outputs are
 * written in sample_by_range order: K = H * sum_i ceil(W / rate_i) candidates per sweep,
 * band-major; scores are zeroed outside the band (lower, upper], boxes/categories are not.
 * With n_bands == 0 the outputs are the dense H*W grid.
 *   scores (B,K) f32, categories (B,K) i64, boxes (B,K,7) f32. */
int rv_decode_candidates(const float* logits, const float* regressands, const float* cart, const uint8_t* mask,
                         int32_t B, int32_t n_cls, int32_t H, int32_t W, int32_t azimuth_invariant, int32_t n_bands,
                         const float* host_lower, const float* host_upper, const int32_t* host_rates,
                         int64_t category_offset,
                         float* scores, int64_t* categories, float* boxes, rvStream stream);
int64_t rv_decode_num_candidates(int32_t H, int32_t W, int32_t n_bands, const int32_t* host_rates);
/* decode_range_view alone on NCHW fp32 (B,8,H,W)+(B,3,H,W) -> (B,7,H,W) (math/ops/coding.py:110-144) */
int rv_decode_range_view(const float* regressands, const float* cart, int32_t B, int32_t H, int32_t W,
                         int32_t azimuth_invariant, float* out, rvStream stream);
/* yaw (n,) -> wxyz quaternions (n,4)  (SO3.py:122-134) */
int rv_yaw_to_quat(const float* yaw, int64_t n, int64_t yaw_stride, float* quat, rvStream stream);

/* ---------------------------------------------------------------------------------------
 * Weighted NMS -- replaces `weighted_nms_ext.wnms_gpu(boxes, data2merge_score, output, keep,
 * count, nms_thresh, merge_thresh, device_index) -> int` (math/ops/nms.py:161-170).
 * Same contract: inputs sorted by score descending; boxes (n,5) = [x1,y1,x2,y2,ry] f32;
 * data (n,d) f32 with the score in the last column; caller-allocated zero-initialised
 * `output` (n,d), `keep` (n,) i64, `count` (n,) i64 -- all DEVICE buffers here (the
 * reference keeps `keep` on the host); returns the number of kept boxes through
 * `host_num_out` after synchronising `stream` (the reference call is synchronous too).
 * `workspace`: rv_wnms_workspace_bytes(n) bytes.  Semantics: see oracle/nms.py (the
 * third-party kernel's arithmetic is not in the reference tree -- parity unpinned).
 * ------------------------------------------------------------------------------------- */
int64_t rv_wnms_workspace_bytes(int64_t n);
int rv_wnms(const float* boxes, const float* data, int64_t n, int32_t d, float nms_thresh, float merge_thresh,
            float* output, int64_t* keep, int64_t* count, void* workspace, int64_t* host_num_out, rvStream stream);
/* The same with a class id per box (cats, i32, may be NULL): boxes of different classes neither suppress nor merge, so ONE
 * call does what the reference's per-class loop does (weighted_multiclass_nms, math/ops/nms.py:64-123) -- identical rows
 * per class, since classes do not interact and the score order within a class is preserved. */
int rv_wnms_classes(const float* boxes, const float* data, const int32_t* cats, int64_t n, int32_t d, float nms_thresh,
                    float merge_thresh, float* output, int64_t* keep, int64_t* count, void* workspace,
                    int64_t* host_num_out, rvStream stream);
/* pairwise rotated BEV IoU (n x m), exposed for tests */
/* The whole post-decode path of a BATCH of sweeps, device-resident from end to end (csrc/nms2.hip): confidence filter +
 * compaction, (class, score) ordering, class-segmented weighted NMS (one scan workgroup per class), per-class top-k by
 * merged score, final compaction in the reference's output order (math/ops/nms.py:64-123, 181-266: sweeps in order, classes
 * ascending, merged score descending).  scores (B,K) f32, cats (B,K) i64 in [0, n_classes), cuboids (B,K,7) f32
 * [x,y,z,l,w,h,yaw]; `cap` = candidate capacity per sweep (multiple of 64, <= 262144: the decoder emits 212 992 per
 * 64 x 2048 sweep; per-candidate workspace: rv_nms_sweeps_workspace_bytes, ~117 bytes per candidate).  The pair masks are
 * class-relative and live in `mask_workspace`: B x 2 x `mask_words` 64-bit words; a sweep whose classes need more than
 * `mask_words` (= sum over classes of n_c * ceil(n_c / 64), n_c cut at num_pre_nms: the reference's per-class pre-NMS top-k,
 * nms.py:83-84) reports the number and is redone by a call with `resume` != 0 over a buffer of that size (the ordering
 * stages are not repeated; `workspace` must be untouched in between).  Outputs (device): out_boxes (B,out_cap,7),
 * out_scores (B,out_cap), out_cats (B,out_cap), out_cap >= min(cap, n_classes * num_post_nms);
 * out_counts (B,4) i64 = {rows written, or -1: mask budget exceeded, -2: more than `cap` candidates;
 * candidates >= min_confidence; boxes kept by the NMS; mask words the sweep needs}.
 * Asynchronous: the caller reads out_counts back once for the whole batch. */
int64_t rv_nms_sweeps_workspace_bytes(int32_t B, int32_t cap);
int rv_nms_sweeps(const float* scores, const int64_t* cats, const float* cuboids, int32_t B, int64_t K, int32_t n_classes,
                  float min_confidence, float nms_thresh, float merge_thresh, int32_t num_pre_nms, int32_t num_post_nms,
                  int32_t cap, int32_t out_cap, float* out_boxes, float* out_scores, int32_t* out_cats, int64_t* out_counts,
                  void* workspace, void* mask_workspace, int64_t mask_words, int32_t resume, rvStream stream);
int rv_rotated_iou(const float* a, int64_t n, const float* b, int64_t m, float* out, rvStream stream);

/* ---------------------------------------------------------------------------------------
 * Range-image projection (converters/av2/utils.py:108-208 == math/numpy/conversions.py:9-128).
 * ------------------------------------------------------------------------------------- */
/* Raw sweep -> the points the projection bins (converters/av2/utils.py:211-295, :32-55).  All device pointers.
 * rv_unmotion_compensate: xyz (n,3) fp64 ego-frame points of the sweep stamped `sweep_timestamp_ns` with per-point
 *   `offset_ns`; the pose track (sorted timestamps, wxyz quaternions, translations; `target_pose` = index of the pose at the
 *   sweep timestamp).  kept[i] = 0 for points outside (first, last) pose time (the reference drops them; their xyz_p is 0);
 *   xyz_p = the point in the ego frame AT ITS OWN capture time (scipy-Slerp rotation, the reference's translation weights).
 * rv_correct_laser_numbers: laser ids -> image rows through LASER_MAPPING (only for `affected` logs) and the row table
 *   (ROW_MAPPING_64 / _32 of datasets/argoverse/constants.py, passed in as data); ids outside the table give -1.
 * rv_se3_inverse_apply: out = R(q)^T (xyz - t): ego -> sensor with the sensor's extrinsics (egovehicle_SE3_sensor); points with
 *   kept[i] == 0 (kept may be NULL) come out as the origin: range 0, which rv_z_buffer skips -- the point order is preserved. */
int rv_unmotion_compensate(const double* xyz, const int32_t* offset_ns, int64_t n, int64_t sweep_timestamp_ns,
                           const int64_t* pose_timestamps_ns, const double* pose_q_wxyz, const double* pose_t, int32_t n_poses,
                           int32_t target_pose, double* xyz_p, uint8_t* kept, rvStream stream);
int rv_correct_laser_numbers(const int32_t* laser, int64_t n, int32_t affected, const int32_t* laser_mapping_32,
                             const int32_t* row_mapping, int32_t n_rows, int32_t* out, rvStream stream);
int rv_se3_inverse_apply(const double* xyz, int64_t n, const double* q_wxyz, const double* t, const uint8_t* kept, double* out,
                         rvStream stream);
/* Correctly rounded (round-to-nearest-even) fp64 atan2, elementwise -- the azimuth rv_project_indices bins with
 * (np.arctan2 at converters/av2/utils.py:172; see csrc/project.hip for why the device value must be THE rounded one). */
int rv_atan2_cr(const double* y, const double* x, int64_t n, double* out, rvStream stream);
/* fp64 hypot with the bits of the C library the reference runs on (np.hypot at math/numpy/conversions.py:64-65 ==
 * glibc 2.35 hypot: Borges' corrected sqrt, not correctly rounded) -- the range rv_project_indices returns and the
 * z-buffer compares (csrc/project.hip). */
int rv_hypot_libc(const double* x, const double* y, int64_t n, double* out, rvStream stream);
/* cart (n,3) f64 -> rows/cols (i32) + range (f64); variant 0 = converter binning
 * (col = W - round((az+pi)*W/tau)), 1 = library binning (col = round(W - (az+pi)*W/tau - 1));
 * round-half-to-even, clip to [0, W-1] before the integer cast; row = H - laser_mapping[laser] - 1. */
int rv_project_indices(const double* cart, const int32_t* laser, const int32_t* laser_mapping, int64_t n,
                       int32_t H, int32_t W, int32_t variant, int32_t* rows, int32_t* cols, double* range,
                       rvStream stream);
/* z-buffer with the reference's sequential semantics: skip range < min_range; pixel owner =
 * the point with the smallest float(range) ... see DESIGN.md §z-buffer for the exact rule
 * (fp64-vs-fp32 comparison, earliest index on ties).  features (c,n) f64 -> image (c,H,W) f32
 * (zeros where empty); winner (H,W) i64 (-1 where empty); `keys` is an (H*W) u64 scratch. */
int rv_z_buffer(const int32_t* rows, const int32_t* cols, const double* range, const double* features, int64_t n,
                int32_t c, int32_t H, int32_t W, double min_range, uint64_t* keys, float* image, int64_t* winner,
                rvStream stream);

/* Spherical <-> Cartesian (math/conversions.py:28-81, math/numpy/conversions.py:46-103): (n,3) arrays,
 * [azimuth, inclination, radius] <-> [x, y, z]; is_f64 selects double (numpy twins) or float (torch versions). */
int rv_cart_to_sph(const void* cart, int64_t n, int32_t is_f64, void* sph, rvStream stream);
int rv_sph_to_cart(const void* sph, int64_t n, int32_t is_f64, void* cart, rvStream stream);
/* The loader's per-sweep contract (prototype/loader.py:568-705, DataLoader.__getitem__): a range-view table with H*W rows
 * and named fp32 columns (`table`: [n_cols][hw], one column after the other, as Arrow stores it) -> features [n_feat][hw]
 * (= (F,H,W)), cart [3][hw], mask [hw] (range > 0).  roi_col >= 0: every column is first multiplied by that 0/1 column
 * (`filter_roi`, loader.py:599-601).  host_feat_op[f]: 0 copy, 1 tanh (Waymo intensity, :627), 2 times 1e-9 (timedelta_ns, :633).
 * The index arrays are HOST arrays (<= 16 features).  Replaces the polars select / to_numpy / transpose / reshape chain. */
int rv_table_to_range_view(const float* table, int32_t n_cols, int64_t hw, int32_t n_feat, const int32_t* host_feat_col,
                           const int32_t* host_feat_op, const int32_t* host_cart_col, int32_t range_col, int32_t roi_col,
                           float* features, float* cart, uint8_t* mask, rvStream stream);

/* Loader augmentations on device (prototype/loader.py:825-990: flip_azimuth, random_rotation, random_global_scale,
 * random_global_translation, and chains of them).  in / out: (B, C, H, W) fp32, distinct buffers.  params: B x 32 doubles
 * on the DEVICE: {a, b} column map w_src = (a*w + b) mod W with a = +-1; A[9], t[3] affine map of the channels ix / iy / iz
 * (xyz' = A xyz + t, fp64, rounded once); Ar[9], tr[3], use_range: the channel `irange` becomes ||Ar xyz + tr|| when
 * use_range != 0 (the reference recomputes the range only in random_global_scale); 5 pad doubles.  ix = iy = iz = -1: only the
 * column map is applied (mask, extra feature maps).  Every other channel is copied through the column map bit for bit. */
int rv_augment(const float* in, float* out, int32_t B, int32_t C, int32_t H, int32_t W, int32_t ix, int32_t iy, int32_t iz,
               int32_t irange, const double* params, rvStream stream);
/* subsample_range_view's W padding at x_stride 1 (prototype/loader.py:792-815): out (C,H,W+2*pad) = pad(image * mask);
 * mask (H,W) may be NULL; circular != 0 wraps around in azimuth, else zeros.  AV2 pad 4 (1800 -> 1808), Waymo 3. */
int rv_pad_range_view(const float* image, const float* mask, int32_t C, int32_t H, int32_t W, int32_t pad,
                      int32_t circular, float* out, rvStream stream);

/* ---------------------------------------------------------------------------------------
 * Targets + losses on device (nn/heads/detection_head.py:496-665, math/ops/assignment.py:76-161,
 * nn/functional/__init__.py:8-27, detection_head.py:202-449) -- see rv3d.h section in DESIGN.md.
 * ------------------------------------------------------------------------------------- */
/* cuboids (m,10) f64 [x,y,z,l,w,h,yaw,task,category,batch] grouped by sweep; `box_offsets` (B+1) i32 is the
 * CSR of the grouping (device); cart NCHW f32.  Outputs: labels (B,H,W) i64 (background = n_cls),
 * panoptics (B,H,W) i64 (0 = background, else 1-based rank of the owning box by interior-point
 * count ascending, ties in input order == the reference's stable sort), regression targets
 * (B,8,H,W) f32, points_per_obj (B,H,W) i64, num_objects (1) i32 = boxes owning >= 1 pixel.
 * counts / order / owned: (m) i32 scratch.  No host synchronisation (the reference's loop calls
 * .unique()/.tolist() per sweep, task and instance). */
int rv_assign_targets(const double* cuboids, int32_t m, const int32_t* box_offsets, const float* cart, int32_t B,
                      int32_t H, int32_t W, int32_t n_cls, int32_t azimuth_invariant, int32_t* counts, int32_t* order,
                      int32_t* owned, int64_t* labels, int64_t* panoptics, float* reg_targets,
                      int64_t* points_per_obj, int32_t* num_objects, rvStream stream);

/* Fused detection loss.  logits / regressands are NHWC fp32 with channel strides ld_* (the layout
 * the head kernels write); cart / reg_targets NCHW fp32; mask (B,H,W) u8.
 * forward : sums[24] (f64, device; 16 before round 6) -- [0] sum w*VFL*mask, [1] foreground part, [2] background part,
 *           [3] #foreground, [4..11] un-normalised regression sums per regressand, [12] max(objects,1),
 *           [13] #foreground + smoothing, [15] = 1.0 (the backward pass's device-side factor, below), [16..23] the scalars
 *           detection_head.py:379-449 reports: loss = sums[0]/sums[13] + (sums[4]+...+sums[11])/sums[12], classification, foreground,
 *           background, coordinate, dimension, rotation, regression loss; optional soft targets (B,n_cls,H,W) and foreground map.
 * backward: d loss / d logits, d loss / d regressands (same NHWC strides), scaled by grad_scale * sums[15] -- the caller may copy
 *           the incoming gradient of the loss (a device scalar) into sums[15] instead of multiplying both tensors afterwards;
 *           reads sums[12], sums[13], sums[15] on device (no host round trip).  Padding channels of the gradient buffers are not written. */
int rv_detection_loss_forward(const float* logits, int32_t ld_logits, const float* regressands, int32_t ld_reg,
                              const float* cart, const uint8_t* mask, const int64_t* labels, const int64_t* panoptics,
                              const float* reg_targets, const int64_t* points_per_obj, const int32_t* num_objects,
                              int32_t B, int32_t n_cls, int32_t H, int32_t W, const float* host_coding_weights,
                              float cls_weight, float reg_weight, float smoothing, float sigma, float alpha, float gamma,
                              int32_t azimuth_invariant, double* sums, float* soft_targets, float* foreground,
                              rvStream stream);
int rv_detection_loss_backward(const float* logits, int32_t ld_logits, const float* regressands, int32_t ld_reg,
                               const float* cart, const uint8_t* mask, const int64_t* labels, const int64_t* panoptics,
                               const float* reg_targets, const int64_t* points_per_obj, const int32_t* num_objects,
                               int32_t B, int32_t n_cls, int32_t H, int32_t W, const float* host_coding_weights,
                               float cls_weight, float reg_weight, float smoothing, float sigma, float alpha, float gamma,
                               int32_t azimuth_invariant, const double* sums, float grad_scale, float* d_logits,
                               float* d_regressands, rvStream stream);

#ifdef __cplusplus
}
#endif
#endif /* RV3D_H_ */

// optim.hip -- the optimiser step of the training recipe (nn/meta/arch.py:48-75: torch.optim.AdamW; gradient clipping at
// 35.0 = conf/trainer/train.yaml gradient_clip_val) for ALL parameters in two launches: (1) per-chunk sums of squares of the
// gradients, (2) total norm -> clip coefficient -> decoupled weight decay + Adam update of every chunk.  Replaces ATen's
// clip_grad_norm_ (foreach norm, stack, norm, clamp, foreach mul) + foreach AdamW (~10 multi-tensor launches that move
// each of p, g, m, v several times).  HBM-bound: g is read twice, p / m / v once and written once, 16-byte accesses.
//
// The arithmetic follows torch/optim/adamw.py (_multi_tensor_adamw, non-capturable) operation by operation in fp32:
//   p *= 1 - lr*wd;  m = m + (g - m)(1 - b1);  v = v*b2 + g*g*(1 - b2);
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// with g = grad * min(1, max_norm / (||grad||_2 + 1e-6)) (torch.nn.utils.clip_grad_norm_).
#include "common.h"

namespace {

constexpr int kChunk = 65536;  // elements per workgroup

struct OptTensor {
    float* p;
    const float* g;
    float* m;
    float* v;
    int64_t n;
};
struct OptChunk {
    int32_t tensor, chunk;
};

__global__ __launch_bounds__(256) void optim_sqnorm_kernel(const OptTensor* tensors, const OptChunk* chunks, float* partial) {
    const OptChunk c = chunks[blockIdx.x];
    const OptTensor t = tensors[c.tensor];
    const int64_t lo = (int64_t)c.chunk * kChunk, hi = lo + kChunk < t.n ? lo + kChunk : t.n;
    const float* g = t.g + lo;
    const int64_t n = hi - lo;
    float s = 0.f;
    if ((((uintptr_t)g) & 15) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = threadIdx.x; i < n4; i += 256) {
            const f32x4 x = ((const f32x4*)g)[i];
            s += x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
        }
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
    } else {
        for (int64_t i = threadIdx.x; i < n; i += 256) s += g[i] * g[i];
    }
    __shared__ float red[4];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

struct OptHyper {  // the scalars torch forms in double on the host, each rounded to fp32 once (as the foreach ops take them)
    float decay, one_minus_beta1, beta2, one_minus_beta2, step_size, bc2_sqrt, eps, max_norm;
};

__device__ __forceinline__ void adam_elem(float& p, const float g, float& m, float& v, const OptHyper& h, const float clip) {
    const float gc = g * clip;
    p *= h.decay;
    m = m + (gc - m) * h.one_minus_beta1;
    v = v * h.beta2 + gc * gc * h.one_minus_beta2;
    const float denom = sqrtf(v) / h.bc2_sqrt + h.eps;
    p -= h.step_size * (m / denom);
}

__global__ __launch_bounds__(256) void optim_adamw_kernel(const OptTensor* tensors, const OptChunk* chunks, const float* partial,
                                                          int n_partial, const OptHyper h, float* total_norm_out) {
    // every workgroup forms the same total (fixed order: reproducible) from the per-chunk sums
    __shared__ double red[4];
    __shared__ float clip_s;
    double s = 0.0;
    for (int i = threadIdx.x; i < n_partial; i += 256) s += (double)partial[i];
    s = wave_sum_d(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
        float clip = 1.f;
        if (h.max_norm > 0.f) {
            clip = h.max_norm / (norm + 1e-6f);
            clip = clip < 1.f ? clip : 1.f;
        }
        clip_s = clip;
        if (blockIdx.x == 0 && total_norm_out) total_norm_out[0] = norm;
    }
    __syncthreads();
    const float clip = clip_s;
    const OptChunk c = chunks[blockIdx.x];
    const OptTensor t = tensors[c.tensor];
    const int64_t lo = (int64_t)c.chunk * kChunk, hi = lo + kChunk < t.n ? lo + kChunk : t.n;
    const int64_t n = hi - lo;
    float* p = t.p + lo;
    const float* g = t.g + lo;
    float* m = t.m + lo;
    float* v = t.v + lo;
    int64_t done = 0;
    if (((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0) {
        const int64_t n4 = n >> 2;
        for (int64_t i = threadIdx.x; i < n4; i += 256) {
            const f32x4 p4 = ((f32x4*)p)[i], m4 = ((f32x4*)m)[i], v4 = ((f32x4*)v)[i], g4 = ((const f32x4*)g)[i];
            float pp[4] = {p4[0], p4[1], p4[2], p4[3]}, mm[4] = {m4[0], m4[1], m4[2], m4[3]}, vv[4] = {v4[0], v4[1], v4[2], v4[3]};
            const float gg[4] = {g4[0], g4[1], g4[2], g4[3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) adam_elem(pp[j], gg[j], mm[j], vv[j], h, clip);
            ((f32x4*)p)[i] = f32x4{pp[0], pp[1], pp[2], pp[3]};
            ((f32x4*)m)[i] = f32x4{mm[0], mm[1], mm[2], mm[3]};
            ((f32x4*)v)[i] = f32x4{vv[0], vv[1], vv[2], vv[3]};
        }
        done = n4 << 2;
    }
    for (int64_t i = done + threadIdx.x; i < n; i += 256) adam_elem(p[i], g[i], m[i], v[i], h, clip);
}

}  // namespace

extern "C" int32_t rv_optim_chunk_elems(void) { return kChunk; }

extern "C" int rv_adamw_step(const void* tensors, const void* chunks, int32_t n_chunks, float* partial, double lr, double beta1,
                             double beta2, double eps, double weight_decay, int64_t step, double max_norm, float* total_norm,
                             rvStream stream) {
    RV_REQUIRE(tensors && chunks && partial && n_chunks > 0, "rv_adamw_step: null argument");
    RV_REQUIRE(step >= 1 && beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0, "rv_adamw_step: bad hyper-parameters");
    OptHyper h;
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    h.decay = (float)(1.0 - lr * weight_decay);
    h.one_minus_beta1 = (float)(1.0 - beta1);
    h.beta2 = (float)beta2;
    h.one_minus_beta2 = (float)(1.0 - beta2);
    h.step_size = (float)(lr / bc1);
    h.bc2_sqrt = (float)sqrt(bc2);
    h.eps = (float)eps;
    h.max_norm = (float)max_norm;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(optim_sqnorm_kernel, dim3(n_chunks), dim3(256), 0, st, (const OptTensor*)tensors, (const OptChunk*)chunks, partial);
    hipLaunchKernelGGL(optim_adamw_kernel, dim3(n_chunks), dim3(256), 0, st, (const OptTensor*)tensors, (const OptChunk*)chunks, partial,
                       n_chunks, h, total_norm);
    RV_CHECK_LAUNCH("optim_adamw_kernel");
    return 0;
}

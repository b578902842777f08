// What an LDS-DMA instruction costs the wave that issues it, among MFMAs (gfx950).  Evidence for DESIGN 3.1 / r04_tapconv6_ablation.md
// section 3: would a one-wave-per-SIMD (512-register) tap-conv pay for issuing its own DMA?
//
// Each wave loops over blocks of 64 x v_mfma_f32_16x16x32_bf16 (independent accumulators, operands in registers) with D LDS-DMA
// instructions (global_load_lds_dwordx4, 1 KB each, from an L2-resident 1 MB buffer, into a private LDS slot) spread evenly through
// the block and one `s_waitcnt vmcnt(D)` per block (the loads of the previous block must have landed).  Workgroups of 256 threads
// (one wave per SIMD) and 512 threads (two per SIMD, no barriers: the partner's MFMAs fill the issue gaps), one workgroup per CU.
// Prints s_memtime ticks per block (median over workgroups; NOT shader cycles under load: compare within a row group only) and the
// chip-wide TFLOP/s, which is the figure to read.
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb_dma_issue profiles/tools/mb_dma_issue.hip && /tmp/mb_dma_issue
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

template <int D, int THREADS>
__global__ __launch_bounds__(THREADS, 1) void kernel(const uint8_t* src, float* sink, uint64_t* cycles, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 8; ++j) {
            a[i][j] = (__bf16)(float)((tid * 7 + i * 3 + j) % 13 - 6);
            b[i][j] = (__bf16)(float)((tid * 5 + i + j * 2) % 11 - 5);
        }
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint8_t* p = src + ((blockIdx.x * 8 + wave) * 4096 % (1 << 20)) + lane * 16;
    uint8_t* slot = smem + wave * (D > 0 ? D : 1) * 1024;
    __syncthreads();
    const uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {  // 4 x 16 MFMAs
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
            // D instructions per block of 64 MFMAs, D / 4 after each group of 16 (D = 1, 2: after the first group(s))
#pragma unroll
            for (int d = 0; d < (D + 3 - q) / 4; ++d) {
                const int n = q + 4 * d;
                __builtin_amdgcn_global_load_lds((glb_void_t*)(p + (n & 3) * 1024), (lds_void_t*)(slot + n * 1024), 16, 0, 0);
            }
        }
        if (D > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint64_t t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][3];
    sink[blockIdx.x * THREADS + tid] = s + (float)smem[tid];
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int D, int THREADS>
void run(const uint8_t* src, float* sink, uint64_t* cyc_d, int iters) {
    const int grid = 256;
    const size_t lds = (size_t)(THREADS / 64) * (D > 0 ? D : 1) * 1024 + 1024;
    hipFuncSetAttribute((const void*)kernel<D, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((kernel<D, THREADS>), dim3(grid), dim3(THREADS), lds, 0, src, sink, cyc_d, iters);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((kernel<D, THREADS>), dim3(grid), dim3(THREADS), lds, 0, src, sink, cyc_d, iters);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> c(grid);
    hipMemcpy(c.data(), cyc_d, grid * sizeof(uint64_t), hipMemcpyDeviceToHost);
    std::sort(c.begin(), c.end());
    const double per_block = (double)c[grid / 2] / iters;
    const double flops = 2.0 * 16 * 16 * 32 * 64 * (THREADS / 64) * grid * (double)iters;
    printf("waves/SIMD %d  DMA per 64 MFMAs %2d : %7.1f cycles per block per wave (MFMA-bound: %d), %6.0f TFLOP/s, %5.1f GB/s of fill per CU\n",
           THREADS / 256, D, per_block, 1024 * (THREADS / 256), flops / (ms * 1e-3) / 1e12,
           (double)D * 1024 * (THREADS / 64) * iters / (ms * 1e-3) / 1e9);
}

int main() {
    uint8_t* src;
    float* sink;
    uint64_t* cyc;
    hipMalloc(&src, 1 << 21);
    hipMemset(src, 1, 1 << 21);
    hipMalloc(&sink, 256 * 512 * sizeof(float));
    hipMalloc(&cyc, 256 * sizeof(uint64_t));
    const int iters = 4000;
    run<0, 256>(src, sink, cyc, iters);
    run<1, 256>(src, sink, cyc, iters);
    run<2, 256>(src, sink, cyc, iters);
    run<3, 256>(src, sink, cyc, iters);
    run<4, 256>(src, sink, cyc, iters);
    run<8, 256>(src, sink, cyc, iters);
    run<0, 512>(src, sink, cyc, iters / 2);
    run<1, 512>(src, sink, cyc, iters / 2);
    run<2, 512>(src, sink, cyc, iters / 2);
    run<4, 512>(src, sink, cyc, iters / 2);
    run<8, 512>(src, sink, cyc, iters / 2);
    return 0;
}

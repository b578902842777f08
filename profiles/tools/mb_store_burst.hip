// Microbenchmark: does a 128 KB store burst per CU cost less when the CUs do NOT burst in the same microsecond?
//   hipcc --offload-arch=gfx950 -O3 -o build_mb/mb_store_burst profiles/tools/mb_store_burst.hip && build_mb/mb_store_burst
// 256 workgroups of 512 threads (one per CU), each: `tiles` times { 128 KB of global_store_dwordx4 to fresh memory in the tapconv6
// epilogue's pattern (a wave instruction = 4 pixels x 256 B at a pixel pitch of `ld` bytes), wait for them (vmcnt(0)), idle `gap`
// us }.  The burst time (first store issued -> all acknowledged) is stamped with s_memrealtime (100 MHz).  Phase: all workgroups in
// step (what persistent workgroups with equal tiles do), or each workgroup delayed by a pseudo-random part of one period.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void idle_us(int us) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (uint64_t)us * 100) __builtin_amdgcn_s_sleep(32);
}

// rows > 0: the tile is 16 rows x 32 pixels of an N x 64 x 2048-pixel image (row pitch 2048 ld), tiles walked as tapconv6 walks them
__global__ __launch_bounds__(512) void burst_kernel(char* base, int tiles, int gap_us, int ld, int desync_us, uint32_t* out, int rows = 0) {
    const int tid = threadIdx.x;
    const int wg = blockIdx.x;
    if (desync_us > 0) idle_us((int)(((uint32_t)wg * 2654435761u >> 16) % (uint32_t)desync_us));
    const u32x4 v = {(uint32_t)tid, 1u, 2u, 3u};
    uint64_t acc = 0;
    for (int t = 0; t < tiles; ++t) {
        // tile t of this workgroup: 512 pixels x 256 B, pixel pitch ld: fresh memory every time
        char* tile = base + ((size_t)(t * gridDim.x + wg) * 512) * ld;
        if (rows) {
            const int idx = t * gridDim.x + wg;           // 64 tile columns, 4 tile rows per image
            const int tc = idx % 64, th = (idx / 64) % 4, n = idx / 256;
            tile = base + (((size_t)n * 64 + th * 16) * 2048 + tc * 32) * ld;
        }
        __syncthreads();
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int px = it * 32 + (tid >> 4);
            if (rows) *(u32x4*)(tile + ((size_t)it * 2048 + (tid >> 4)) * ld + (tid & 15) * 16) = v;
            else *(u32x4*)(tile + (size_t)px * ld + (tid & 15) * 16) = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        acc += __builtin_amdgcn_s_memrealtime() - t0;
        idle_us(gap_us);
    }
    if (tid == 0) out[wg] = (uint32_t)acc;
}

int main() {
    const int grid = 256, tiles = 64;
    char* buf;
    uint32_t* out;
    const size_t bytes = (size_t)tiles * grid * 512 * 1024 + ((size_t)64 << 20);  // ld up to 1024
    hipMalloc(&buf, bytes);
    hipMalloc(&out, grid * 4);
    std::vector<uint32_t> h(grid);
    for (int ld : {256, 1024}) {
        for (int gap : {0, 20, 50}) {
            for (int desync : {0, 10, 60}) {
                for (int rep = 0; rep < 2; ++rep) {
                    burst_kernel<<<grid, 512>>>(buf, tiles, gap, ld, desync, out);
                    hipDeviceSynchronize();
                }
                hipMemcpy(h.data(), out, grid * 4, hipMemcpyDeviceToHost);
                std::sort(h.begin(), h.end());
                printf("pitch %4d B, idle %2d us between bursts, phase spread %2d us: burst of 128 KB takes %6.2f us median, %6.2f max (per CU %5.1f GB/s)\n",
                       ld, gap, desync, h[grid / 2] / 100.0 / tiles, h[grid - 1] / 100.0 / tiles, 131072.0 / (h[grid / 2] / 100.0 / tiles * 1e-6) / 1e9);
            }
        }
    }
    for (int ld : {256, 512, 1024}) {  // the image layout: 64 tiles x 256 workgroups = 16384 tiles = 16 images of 64 x 2048
        for (int gap : {0, 20}) {
            for (int rep = 0; rep < 2; ++rep) {
                burst_kernel<<<grid, 512>>>(buf, tiles, gap, ld, 0, out, 1);
                hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), out, grid * 4, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            printf("IMAGE ROWS (16 x 32 px, row pitch %4d KB), pixel pitch %4d B, idle %2d us: burst of 128 KB takes %6.2f us median, %6.2f max\n", 2048 * ld / 1024, ld, gap,
                   h[grid / 2] / 100.0 / tiles, h[grid - 1] / 100.0 / tiles);
        }
    }
    return 0;
}

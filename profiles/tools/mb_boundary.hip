// Microbenchmark of a tapconv6 tile boundary as the memory system sees it: every CU stores a 128 KB output tile and then loads the
// 73 KB prologue of its next tile (fresh HBM addresses), all CUs in step, then computes (idles) for `gap` us.
//   hipcc --offload-arch=gfx950 -O3 -o build_mb/mb_boundary profiles/tools/mb_boundary.hip && build_mb/mb_boundary
// mode 0: stores only   1: loads only   2: stores, then loads (the kernel's order)   3: loads, then stores (pipelined boundary)
// Times (s_memrealtime, 100 MHz): until the loads have landed (`ld`), until everything is acknowledged (`all`).
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

template <int MODE>
__global__ __launch_bounds__(512) void boundary_kernel(char* wbase, const char* rbase, int tiles, int gap_us, uint32_t* out) {
    const int tid = threadIdx.x, wg = blockIdx.x;
    const u32x4 v = {(uint32_t)tid, 1u, 2u, 3u};
    uint64_t t_ld = 0, t_all = 0;
    u32x4 sink = {0, 0, 0, 0};
    for (int t = 0; t < tiles; ++t) {
        char* wt = wbase + (size_t)(t * gridDim.x + wg) * 131072;
        const char* rt = rbase + (size_t)(t * gridDim.x + wg) * 81920;
        __syncthreads();
        const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
        u32x4 r[9];
        auto loads = [&]() {
#pragma unroll
            for (int i = 0; i < 9; ++i) r[i] = __builtin_nontemporal_load((const u32x4*)(rt + (size_t)(i * 512 + tid) * 16));  // 72 KB
        };
        auto stores = [&]() {
#pragma unroll
            for (int it = 0; it < 16; ++it) *(u32x4*)(wt + (size_t)(it * 512 + tid) * 16) = v;
        };
        if (MODE == 0) stores();
        if (MODE == 1) loads();
        if (MODE == 2) { stores(); loads(); }
        if (MODE == 3) { loads(); stores(); }
        if (MODE == 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (MODE != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (MODE != 0) {
#pragma unroll
            for (int i = 0; i < 9; ++i) sink += r[i];
        }
        __syncthreads();
        t_ld += __builtin_amdgcn_s_memrealtime() - t0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        t_all += __builtin_amdgcn_s_memrealtime() - t0;
        idle_us(gap_us);
    }
    if (sink[0] == 0x12345678u) out[1000] = sink[1];
    if (tid == 0) {
        out[wg] = (uint32_t)t_ld;
        out[256 + wg] = (uint32_t)t_all;
    }
}

int main() {
    const int grid = 256, tiles = 64;
    char *wbuf, *rbuf;
    uint32_t* out;
    hipMalloc(&wbuf, (size_t)tiles * grid * 131072);
    hipMalloc(&rbuf, (size_t)tiles * grid * 81920);
    hipMemset(rbuf, 1, (size_t)tiles * grid * 81920);
    hipMalloc(&out, 2048 * 4);
    std::vector<uint32_t> h(512);
    const char* names[4] = {"stores only", "loads only", "stores, then loads", "loads, then stores"};
    for (int gap : {20, 50}) {
        for (int mode = 0; mode < 4; ++mode) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) boundary_kernel<0><<<grid, 512>>>(wbuf, rbuf, tiles, gap, out);
                if (mode == 1) boundary_kernel<1><<<grid, 512>>>(wbuf, rbuf, tiles, gap, out);
                if (mode == 2) boundary_kernel<2><<<grid, 512>>>(wbuf, rbuf, tiles, gap, out);
                if (mode == 3) boundary_kernel<3><<<grid, 512>>>(wbuf, rbuf, tiles, gap, out);
                hipDeviceSynchronize();
            }
            hipMemcpy(h.data(), out, 512 * 4, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.begin() + 256);
            std::sort(h.begin() + 256, h.end());
            printf("idle %2d us, %-20s: loads landed after %6.2f us (median), all acknowledged after %6.2f us\n", gap, names[mode], h[128] / 100.0 / tiles,
                   h[256 + 128] / 100.0 / tiles);
        }
    }
    return 0;
}

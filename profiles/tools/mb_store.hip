// Microbenchmark: what ONE CU's store path sustains, by store width and by how many CUs store at once.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mb_store profiles/tools/mb_store.hip && /tmp/mb_store
// One workgroup of 512 threads per CU; each wave writes fresh, contiguous memory (values from registers) with
// global_store_dword / dwordx2 / dwordx4, default or non-temporal; 8 stores in flight per wave before the next address bump.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

template <int W, bool NT>
__global__ __launch_bounds__(512) void store_kernel(char* base, size_t bytes_per_wg, int waves_storing) {
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (wave >= waves_storing) return;
    char* p = base + (size_t)blockIdx.x * bytes_per_wg;
    const size_t per_wave = bytes_per_wg / waves_storing;
    char* q = p + (size_t)wave * per_wave + lane * W;
    const size_t n = per_wave / (64 * W);
    const u32x4 v4 = {(uint32_t)tid, 1u, 2u, 3u};
    for (size_t i = 0; i < n; ++i) {
        if (W == 16) { if (NT) __builtin_nontemporal_store(v4, (u32x4*)q); else *(u32x4*)q = v4; }
        if (W == 8) { const u32x2 v = {v4[0], v4[1]}; if (NT) __builtin_nontemporal_store(v, (u32x2*)q); else *(u32x2*)q = v; }
        if (W == 4) { if (NT) __builtin_nontemporal_store(v4[0], (uint32_t*)q); else *(uint32_t*)q = v4[0]; }
        q += 64 * W;
    }
}

template <int W, bool NT>
void run(char* buf, int grid, int waves, const char* name) {
    const size_t per_wg = (size_t)16 << 20;  // 16 MB per workgroup
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    store_kernel<W, NT><<<grid, 512>>>(buf, per_wg, waves);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    store_kernel<W, NT><<<grid, 512>>>(buf, per_wg, waves);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double gbs = (double)grid * per_wg / (ms * 1e-3) / 1e9;
    printf("%-14s grid %3d waves %d: %8.3f ms  %8.1f GB/s total  %6.1f GB/s per workgroup\n", name, grid, waves, ms, gbs, gbs / grid);
}

int main() {
    char* buf;
    hipMalloc(&buf, (size_t)256 * (16 << 20));
    for (int grid : {8, 32, 64, 128, 256}) {
        for (int waves : {8, 4, 1}) {
            run<16, false>(buf, grid, waves, "dwordx4");
            run<16, true>(buf, grid, waves, "dwordx4 nt");
            run<8, false>(buf, grid, waves, "dwordx2");
            run<4, false>(buf, grid, waves, "dword");
        }
    }
    return 0;
}

// How fast does a 4096 x 4096 f16 output (32 MiB) leave the chip when 256 workgroups of 8 waves each write one 256 x 256
// tile, as the big GEMM's epilogue does?  Pattern A: a wave owns 32 channels (64 B of a row) and one store instruction
// covers 16 rows x 64 B (the 1 x 8 wave layout's natural stores).  Pattern B: a wave owns 32 rows and one store instruction
// covers 2 rows x 512 B (what an LDS transpose of the tile would allow).  Both with non-temporal and plain 16-byte stores.
// Build: hipcc --offload-arch=gfx950 -O2 store_pattern.hip -o store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int N = 4096, M = 4096;

template <int PATTERN, bool NT>
__global__ void __launch_bounds__(512) kern(uint16_t* __restrict__ y, uint32_t seed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tile = blockIdx.x, m0 = (tile / 16) * 256, n0 = (tile % 16) * 256;
    const u32x4 v = {seed + lane, seed, seed ^ 0x55u, seed + wave};
    if constexpr (PATTERN == 0) {
        // lane (c = lane & 15, g = lane >> 4): row 16 i + c, 8 channels at 32 wave + 8 g
        const int c = lane & 15, g = lane >> 4;
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            u32x4* p = reinterpret_cast<u32x4*>(y + (size_t)(m0 + 16 * i + c) * N + n0 + 32 * wave + 8 * g);
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
    } else {
        // lane: row 32 wave + 2 i + (lane >> 5), 8 channels at 8 (lane & 31)
#pragma unroll 4
        for (int i = 0; i < 16; ++i) {
            u32x4* p = reinterpret_cast<u32x4*>(y + (size_t)(m0 + 32 * wave + 2 * i + (lane >> 5)) * N + n0 + 8 * (lane & 31));
            if (NT) __builtin_nontemporal_store(v, p); else *p = v;
        }
    }
}

template <int PATTERN, bool NT>
static void run(uint16_t** bufs, int nbuf, const char* name) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < nbuf; ++i) hipLaunchKernelGGL((kern<PATTERN, NT>), dim3(256), dim3(512), 0, 0, bufs[i], i);
    hipDeviceSynchronize();
    float best = 1e9f, sum = 0;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(a);
        for (int i = 0; i < nbuf; ++i) hipLaunchKernelGGL((kern<PATTERN, NT>), dim3(256), dim3(512), 0, 0, bufs[i], r * 16 + i);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        ms /= nbuf; sum += ms; if (ms < best) best = ms;
    }
    printf("%-44s avg %6.2f us  best %6.2f us  -> %6.0f GB/s\n", name, sum / 5 * 1e3, best * 1e3, (double)M * N * 2 / (sum / 5) / 1e6);
}

int main() {
    const int nbuf = 12;   // 12 x 32 MiB > 256 MiB Infinity Cache
    uint16_t* bufs[nbuf];
    for (int i = 0; i < nbuf; ++i) hipMalloc(&bufs[i], (size_t)M * N * 2);
    run<0, false>(bufs, nbuf, "A: 16 rows x 64 B per store, plain");
    run<0, true>(bufs, nbuf, "A: 16 rows x 64 B per store, non-temporal");
    run<1, false>(bufs, nbuf, "B: 2 rows x 512 B per store, plain");
    run<1, true>(bufs, nbuf, "B: 2 rows x 512 B per store, non-temporal");
    return 0;
}

// How fast can a [N][RB]-byte row-major matrix be streamed HBM -> LDS (LDS-DMA, no compute) when one
// wave-instruction (64 lanes x 16 B) covers R rows x SPAN contiguous bytes?  SPAN = 64, 128, 256, 512, 1024.
// Each wave owns 16 rows x 1024 bytes (16 pieces, all issued up front, then vmcnt(0)); 4 waves per workgroup.
// Build: hipcc --offload-arch=gfx950 -O2 stream_pattern.hip -o stream_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef __attribute__((address_space(3))) void* lptr_t;

template <int SPAN>
__global__ void __launch_bounds__(256, 2) stream_kernel(const uint8_t* __restrict__ w, int64_t rb, uint32_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];  // 4 waves x 16 KiB
    constexpr int L = SPAN / 16, R = 64 / L, RBLK = 16 / R;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t row0 = ((int64_t)blockIdx.x * 4 + wid) * 16;
    const int64_t col0 = (int64_t)blockIdx.y * 1024;
    const uint32_t base = (uint32_t)(uintptr_t)(lptr_t)lds + wid * 16384;
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const int row_block = p % RBLK, span_idx = p / RBLK;
        const int64_t row = row0 + row_block * R + lane / L;
        const uint8_t* src = w + row * rb + col0 + span_idx * SPAN + (lane % L) * 16;
        const uint32_t dst = __builtin_amdgcn_readfirstlane(base + p * 1024);
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
    // touch the data so nothing is optimised away
    uint32_t acc = 0;
#pragma unroll
    for (int p = 0; p < 16; ++p) acc ^= *reinterpret_cast<const uint32_t*>(lds + wid * 16384 + p * 1024 + lane * 16);
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

template <int SPAN>
int run(const std::vector<uint8_t*>& bufs, int64_t n, int64_t rb, uint32_t* out) {
    auto kern = stream_kernel<SPAN>;
    CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    dim3 grid((unsigned)(n / 64), (unsigned)(rb / 1024)), block(256);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 6; ++i) hipLaunchKernelGGL(kern, grid, block, 65536, 0, bufs[i % bufs.size()], rb, out);
    CK(hipDeviceSynchronize());
    std::vector<float> ts;
    for (int rep = 0; rep < 9; ++rep) {
        CK(hipEventRecord(a));
        for (size_t i = 0; i < bufs.size(); ++i) hipLaunchKernelGGL(kern, grid, block, 65536, 0, bufs[i], rb, out);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        ts.push_back(ms / bufs.size());
    }
    std::sort(ts.begin(), ts.end());
    const double bytes = (double)n * rb;
    printf("span %4d B (%2d rows per wave-instruction): %7.2f us per launch -> %6.0f GB/s\n", SPAN, 64 / (SPAN / 16), ts[4] * 1e3, bytes / ts[4] / 1e6);
    return 0;
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 16384, rb = argc > 2 ? atoll(argv[2]) : 4096;
    const int nb = (int)std::max<int64_t>(2, (768ll << 20) / (n * rb));
    std::vector<uint8_t*> bufs(nb);
    for (auto& p : bufs) { CK(hipMalloc(&p, n * rb)); CK(hipMemset(p, 0x55, n * rb)); }
    uint32_t* out; CK(hipMalloc(&out, 4096));
    printf("matrix %lld rows x %lld bytes, %d buffers\n", (long long)n, (long long)rb, nb);
    if (run<64>(bufs, n, rb, out)) return 1;
    if (run<128>(bufs, n, rb, out)) return 1;
    if (run<256>(bufs, n, rb, out)) return 1;
    if (run<512>(bufs, n, rb, out)) return 1;
    if (run<1024>(bufs, n, rb, out)) return 1;
    return 0;
}

// How fast can a few-row product stream packed weights when the ONLY work per weight is the 2-bit decode and the matrix
// instruction?  The weight stream of the few-row kernels (16 channels x 64 bytes per wave instruction, non-temporal), the in-register
// decode of gemm_mfma_kernels.h and v_mfma_f32_16x16x32_f16 against activation fragments that sit in registers (no LDS, no activation
// loads, no barrier): an upper bound for any kernel of that family, as a function of the waves per SIMD, the loads in flight per wave
// and the channel tiles per wave.  Prints microseconds and TB/s of packed weights for n x k = 16384 x 16384 (64 MiB, rotating copies).
//   usage: decode_rate [n k]
#include "../../ternary-vllm_amd/csrc/gemm_mfma_kernels.h"

#include <stdlib.h>
#include <vector>

namespace tvl { void set_error(const char*, ...) {} }
using namespace tvl;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// WG = 4 waves; wave w of workgroup (bx, by): channel tiles (4 bx + w) CT .. + CT - 1, k steps [by * per, (by + 1) * per) of 256.
// MODE 0: load only (sum the dwords)  1: load + decode (xor the fragments)  2: load + decode + MFMA
template <int WPS, int CT, int DEPTH, int MODE>
__global__ void __launch_bounds__(256, WPS)
decode_rate_kernel(const uint8_t* __restrict__ pw, int n, int64_t ld_pw, int per, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int tile0 = ((int)blockIdx.x * 4 + wave) * CT;
    const uint4* wp[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) wp[t] = reinterpret_cast<const uint4*>(pw + (int64_t)((tile0 + t) * 16 + c) * ld_pw + (int64_t)blockIdx.y * per * 64 + 16 * g);
    const DecodeConsts dc = make_decode_consts();
    floatx4 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t) acc[t] = (floatx4){0.f, 0.f, 0.f, 0.f};
    uint4 xf[4];   // stand-ins for activation fragments
#pragma unroll
    for (int j = 0; j < 4; ++j) xf[j] = make_uint4(0x3C003C00u + lane + j, 0x3C003800u + j, 0x38003C00u, 0x3C003C00u ^ (lane << 3));
    uint4 w[DEPTH][CT];
    auto load = [&](int s, uint4 (&dst)[CT]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            const nts_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nts_u32x4*>(wp[t] + (int64_t)s * 4));
            dst[t] = make_uint4(v[0], v[1], v[2], v[3]);
        }
    };
    uint32_t sink = 0;
    auto compute = [&](const uint4 (&ww)[CT]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < CT; ++t) {
            if constexpr (MODE == 0) { asm volatile("" :: "v"(ww[t].x), "v"(ww[t].y), "v"(ww[t].z), "v"(ww[t].w)); continue; }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t d = (j >> 1) == 0 ? ww[t].x : (j >> 1) == 1 ? ww[t].y : (j >> 1) == 2 ? ww[t].z : ww[t].w;
                const uint4 a = (j & 1) ? decode8<f16, 1>(d, dc) : decode8<f16, 0>(d, dc);
                if constexpr (MODE == 1) asm volatile("" :: "v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w));
                else acc[t] = mfma16<f16>(a, xf[j & 3], acc[t]);
            }
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) load(d < per ? d : per - 1, w[d]);
    for (int s = 0; s < per; s += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            if (s + d < per) compute(w[d]);
            load(s + d + DEPTH < per ? s + d + DEPTH : per - 1, w[d]);
        }
    }
    float r = (float)sink;
#pragma unroll
    for (int t = 0; t < CT; ++t) r += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    if (r == 12345.678f) out[0] = r;   // keeps everything alive, never true
}

template <int WPS, int CT, int DEPTH, int MODE>
static void run(const std::vector<uint8_t*>& bufs, int n, int k, int ksplit, float* out) {
    const int per = k / 256 / ksplit;
    const dim3 grid(n / (64 * CT), ksplit), block(256);
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float best = 1e9f;
    for (int round = 0; round < 5; ++round) {
        CHECK(hipEventRecord(a, 0));
        for (size_t i = 0; i < bufs.size(); ++i)
            hipLaunchKernelGGL((decode_rate_kernel<WPS, CT, DEPTH, MODE>), grid, block, 0, 0, bufs[i], n, (int64_t)(k / 4), per, out);
        CHECK(hipEventRecord(b, 0)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b));
        best = ms / bufs.size() < best ? ms / bufs.size() : best;
    }
    CHECK(hipGetLastError());
    const double bytes = (double)n * k / 4;
    const int waves_per_simd = (int)((size_t)grid.x * grid.y * 4 / 1024);
    printf("%-22s launch_bounds(256,%d) tiles/wave %d loads in flight %d k ranges %2d (grid waves per SIMD %2d): %7.2f us  %5.2f TB/s\n",
           MODE == 0 ? "load only" : MODE == 1 ? "load + decode" : "load + decode + mfma", WPS, CT, DEPTH, ksplit, waves_per_simd, best * 1e3, bytes / (best * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const int n = argc > 2 ? atoi(argv[1]) : 16384, k = argc > 2 ? atoi(argv[2]) : 16384;
    const size_t bytes = (size_t)n * k / 4;
    const int nb = (int)((640u << 20) / bytes) < 2 ? 2 : (int)((640u << 20) / bytes);
    std::vector<uint8_t> h(bytes);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    for (auto& v : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; v = (uint8_t)((s % 3) | (((s >> 8) % 3) << 2) | (((s >> 16) % 3) << 4) | (((s >> 24) % 3) << 6)); }
    std::vector<uint8_t*> bufs(nb);
    for (auto& p : bufs) { CHECK(hipMalloc(&p, bytes)); CHECK(hipMemcpy(p, h.data(), bytes, hipMemcpyHostToDevice)); }
    float* out; CHECK(hipMalloc(&out, 64));
    printf("%d x %d packed weights (%zu MiB), %d rotating copies\n", n, k, bytes >> 20, nb);
    run<2, 1, 4, 0>(bufs, n, k, 2, out);
    run<4, 1, 4, 0>(bufs, n, k, 4, out);
    run<8, 1, 2, 0>(bufs, n, k, 8, out);
    run<8, 1, 4, 0>(bufs, n, k, 8, out);
    run<2, 1, 4, 1>(bufs, n, k, 2, out);
    run<4, 1, 4, 1>(bufs, n, k, 4, out);
    run<8, 1, 2, 1>(bufs, n, k, 8, out);
    run<2, 1, 4, 2>(bufs, n, k, 2, out);
    run<4, 1, 4, 2>(bufs, n, k, 4, out);
    run<4, 2, 2, 2>(bufs, n, k, 8, out);
    run<8, 1, 2, 2>(bufs, n, k, 8, out);
    run<8, 1, 4, 2>(bufs, n, k, 8, out);
    run<8, 1, 2, 2>(bufs, n, k, 16, out);
    run<6, 1, 4, 2>(bufs, n, k, 8, out);
    return 0;
}

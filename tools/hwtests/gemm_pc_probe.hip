// The loader-wave / matrix-wave kernel (ternary-vllm_amd/csrc/gemm_pc_kernels.h) against the shipped symmetric 256 x 256 tile, same
// inputs, one process: outputs compared bit for bit, event-pair time per launch after ~1 s of back-to-back launches on random
// data, and in-kernel stamps (s_memtime / s_memrealtime) of the LAST launch as medians over workgroups: clock held over the k
// loop, prologue / k loop / epilogue, cycles per 64-deep step.
//   usage: gemm_pc_probe [M N K [reps]]
#define TVL_CLOCK_PROBE 1
#include "../../ternary-vllm_amd/csrc/gemm_pc_kernels.h"

#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include <vector>

namespace tvl { void set_error(const char*, ...) {} }
using namespace tvl;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

struct Args { const f16* x; const uint8_t* pw; const float *sc, *bi; f16* y; int m, n, k; };

static void report(const char* name, float ms_per, int nwg, int steps_per_wg, double mfma_cycles_per_step) {
    std::vector<unsigned long long> h(8 * 4096);
    CHECK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(tvl_clock_probe), h.size() * 8));
    std::vector<double> ghz, pro_us, loop_us, epi_us, all_us, loopc;
    unsigned long long r_first = ~0ull, r_last = 0;
    for (int bl = 0; bl < std::min(nwg, 4096); ++bl) {
        const unsigned long long* c = &h[8 * bl]; const unsigned long long* r = c + 4;
        ghz.push_back((double)(c[2] - c[1]) / (double)(r[2] - r[1]) * 0.1); loopc.push_back((double)(c[2] - c[1]));
        pro_us.push_back((r[1] - r[0]) * 0.01); loop_us.push_back((r[2] - r[1]) * 0.01); epi_us.push_back((r[3] - r[2]) * 0.01);
        all_us.push_back((r[3] - r[0]) * 0.01);
        r_first = std::min(r_first, r[0]); r_last = std::max(r_last, r[3]);
    }
    const double cps = med(loopc) / steps_per_wg;
    printf("%-44s %8.2f us per launch | clock %.3f GHz | prologue %.2f us, k loop %.2f us (%.0f cyc per step, matrix pipe %.1f %%), tail %.2f us | first entry to last exit %.2f us\n",
           name, ms_per * 1e3, med(ghz), med(pro_us), med(loop_us), cps, 100.0 * mfma_cycles_per_step / cps, med(epi_us), (r_last - r_first) * 0.01);
}

template <typename F>
static float time_launches(F&& launch, int reps, int rounds) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float ms = 0;
    for (int r = 0; r < rounds; ++r) {
        CHECK(hipEventRecord(a, 0));
        for (int i = 0; i < reps; ++i) launch();
        CHECK(hipEventRecord(b, 0)); CHECK(hipEventSynchronize(b));
        CHECK(hipGetLastError());
        CHECK(hipEventElapsedTime(&ms, a, b));
    }
    return ms / reps;
}

static float run_shipped(const Args& a, int reps, int rounds) {
    constexpr int NB = 4;
    auto kern = gemm_mfma16_kernel<f16, f16, true, TVL_ACT_NONE, 1, false, NB, 256, 8, false, 2, 32>;
    const size_t lds = (size_t)NB * 256 * ROW_BYTES;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int tiles_n = (a.n + 255) / 256, nblk = ((a.m + 255) / 256) * tiles_n;
    const float ms = time_launches([&] {
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), lds, 0, a.x, a.m, a.k, (int64_t)a.k, a.pw, a.n, (int64_t)(a.k / 4), a.sc, a.bi, 100.0f, 0, a.y,
                           (int64_t)a.n, tiles_n, nblk, a.k / 64, 0);
    }, reps, rounds);
    report("shipped 256 x 256, 8 symmetric waves", ms, nblk, a.k / 64, 2048.0);
    return ms;
}

template <int BM, int NPROD, bool PERSIST>
static float run_pc(const Args& a, int reps, int rounds, const char* name) {
    auto kern = gemm_pc_kernel<f16, f16, true, TVL_ACT_NONE, BM, NPROD, PERSIST, false>;
    const size_t lds = pc_lds_bytes(BM);
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int tiles_n = (a.n + PC_BN - 1) / PC_BN, nblk = ((a.m + BM - 1) / BM) * tiles_n;
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int grid = PERSIST ? std::min(nblk, prop.multiProcessorCount) : nblk;
    const float ms = time_launches([&] {
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * (4 + NPROD)), lds, 0, a.x, a.m, a.k, (int64_t)a.k, a.pw, a.n, (int64_t)(a.k / 4), a.sc, a.bi, 100.0f, 0,
                           a.y, (int64_t)a.n, tiles_n, nblk, a.k / 64, 0);
    }, reps, rounds);
    const int tiles_per_wg = (nblk + grid - 1) / grid;
    report(name, ms, grid, tiles_per_wg * (a.k / 64), BM == 256 ? 1024.0 : 512.0);
    return ms;
}

int main(int argc, char** argv) {
    const int m = argc > 3 ? atoi(argv[1]) : 4096, n = argc > 3 ? atoi(argv[2]) : 4096, k = argc > 3 ? atoi(argv[3]) : 4096;
    const int reps = argc > 4 ? atoi(argv[4]) : 1000;
    std::vector<uint16_t> hx((size_t)m * k);
    std::vector<uint8_t> hw((size_t)n * k / 4);
    std::vector<float> hs(n), hb(n);
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (auto& v : hx) {   // roughly normal f16: sum of four uniforms, centred
        const uint64_t r = rnd();
        const float f = ((r & 0xFFFF) + ((r >> 16) & 0xFFFF) + ((r >> 32) & 0xFFFF) + ((r >> 48) & 0xFFFF)) / 65536.0f - 2.0f;
        const _Float16 hf = (_Float16)(f * 1.7f);
        v = *reinterpret_cast<const uint16_t*>(&hf);
    }
    for (auto& v : hw) { const uint64_t r = rnd(); v = (uint8_t)((r % 3) | (((r >> 8) % 3) << 2) | (((r >> 16) % 3) << 4) | (((r >> 24) % 3) << 6)); }
    for (int i = 0; i < n; ++i) { hs[i] = 0.5f + (rnd() & 0xFFFF) / 65536.0f; hb[i] = (float)(rnd() & 0xFF) / 64.0f - 2.0f; }
    f16 *x, *y0, *y1; uint8_t* pw; float *sc, *bi;
    const size_t ybytes = (size_t)m * n * 2;
    CHECK(hipMalloc(&x, hx.size() * 2)); CHECK(hipMalloc(&y0, ybytes)); CHECK(hipMalloc(&y1, ybytes)); CHECK(hipMalloc(&pw, hw.size()));
    CHECK(hipMalloc(&sc, n * 4)); CHECK(hipMalloc(&bi, n * 4));
    CHECK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(pw, hw.data(), hw.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(sc, hs.data(), n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(bi, hb.data(), n * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(y0, 0xFF, ybytes));
    std::vector<uint16_t> ref(ybytes / 2), got(ybytes / 2);
    Args a{x, pw, sc, bi, y0, m, n, k};
    auto check = [&](const char* name) {
        CHECK(hipMemcpy(got.data(), y1, ybytes, hipMemcpyDeviceToHost));
        size_t bad = 0, first = 0;
        for (size_t i = 0; i < got.size(); ++i) if (got[i] != ref[i]) { if (!bad) first = i; ++bad; }
        printf("  %s vs shipped: %zu of %zu outputs differ%s\n", name, bad, got.size(), bad ? "" : " (bit-equal)");
        if (bad) printf("    first at row %zu col %zu: got %04x want %04x\n", first / n, first % n, got[first], ref[first]);
        CHECK(hipMemset(y1, 0xFF, ybytes));
    };
    // correctness first, one launch each
    run_shipped(a, 1, 1);
    CHECK(hipMemcpy(ref.data(), y0, ybytes, hipMemcpyDeviceToHost));
    a.y = y1;
    CHECK(hipMemset(y1, 0xFF, ybytes));
    run_pc<256, 4, false>(a, 1, 1, "roles 256 x 128, 4 + 4 waves");            check("256 x 128, 4 + 4");
    run_pc<256, 4, true>(a, 1, 1, "roles 256 x 128, 4 + 4 waves, persistent"); check("256 x 128, 4 + 4, persistent");
    run_pc<128, 4, false>(a, 1, 1, "roles 128 x 128, 4 + 4 waves");            check("128 x 128, 4 + 4");
    run_pc<128, 8, false>(a, 1, 1, "roles 128 x 128, 4 + 8 waves");            check("128 x 128, 4 + 8");
    run_pc<128, 4, true>(a, 1, 1, "roles 128 x 128, 4 + 4 waves, persistent"); check("128 x 128, 4 + 4, persistent");
    if (reps <= 1) return 0;
    // timing: interleaved rounds in one process
    printf("--- timing, %d launches per sample ---\n", reps);
    for (int round = 0; round < 2; ++round) {
        a.y = y0; run_shipped(a, reps, 4);
        a.y = y1;
        run_pc<256, 4, false>(a, reps, 4, "roles 256 x 128, 4 + 4 waves");
        run_pc<256, 4, true>(a, reps, 4, "roles 256 x 128, 4 + 4 waves, persistent");
        run_pc<128, 4, true>(a, reps, 4, "roles 128 x 128, 4 + 4 waves, persistent");
        run_pc<128, 8, false>(a, reps, 4, "roles 128 x 128, 4 + 8 waves");
    }
    return 0;
}

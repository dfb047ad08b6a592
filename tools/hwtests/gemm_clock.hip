// In-kernel clock of the 256 x 256 packed-ternary GEMM tile under sustained load: builds the product's own kernel template with
// shader-clock (s_memtime) and 100 MHz (s_memrealtime) stamps at entry, around the k loop and after the epilogue, runs back-to-back
// launches on random data for ~2 s and prints, as medians over workgroups of the LAST launch: the clock held over the k loop,
// cycles and microseconds of prologue / loop / epilogue, cycles per 64-deep k step.
//   usage: gemm_clock [form=0|1|2] [M N K [splits [lds_tiles]]]   (0: the shipped 256 x 256 tile, 1: four waves x 64 channels, 2: 128 x 128 split-K, 3: the persistent 256 x 256 form, 4: the same with three LDS tiles and four phases, 5: form 0 in eight phases)
#define TVL_CLOCK_PROBE 1
#include "../../ternary-vllm_amd/csrc/gemm_mfma_kernels.h"

#include <algorithm>
#include <stdlib.h>
#include <vector>

namespace tvl { void set_error(const char*, ...) {} }
using namespace tvl;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

static double med(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

// the mid-size path's split-K form: 128 x 128 tiles of four waves, `splits` k ranges per tile, raw float32 partial sums to slabs
template <int NB, int WAVES_M = 1, int BM = 128, int WAVES = 4 * WAVES_M>
static void run_mid_split(int m, int n, int k, int splits, const f16* x, const uint8_t* pw, const float* sc, const float* bi, float* slab) {
    // WAVES_M = 2: eight waves of 64 rows x 32 channels on the same 128 x 128 tile (two waves per SIMD; each weight decoded twice)
    // BM = 64, WAVES = 8: a 64 x 256 tile (12 KiB of loads per step instead of 18: packed weights are cheap to load, activations are not)
    constexpr int BN = 32 * WAVES / WAVES_M;
    auto kern = gemm_mfma16_kernel<f16, float, false, TVL_ACT_NONE, WAVES_M, false, NB, BM, WAVES, true, (WAVES == 8 ? 2 : 3)>;
    const size_t lds = (size_t)NB * BM * ROW_BYTES + 2048;   // + the step stamps
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int tiles_n = (n + BN - 1) / BN, nblk = ((m + BM - 1) / BM) * tiles_n, per = (k / 64 + splits - 1) / splits;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float ms = 0;
    for (int round = 0; round < 8; ++round) {
        CHECK(hipEventRecord(a, 0));
        for (int i = 0; i < 5000; ++i)
            hipLaunchKernelGGL(kern, dim3(nblk, splits), dim3(64 * WAVES), lds, 0, x, m, k, (int64_t)k, pw, n, (int64_t)(k / 4), sc, bi, 100.0f, 0, slab, (int64_t)n,
                               tiles_n, nblk, k / 64, per);
        CHECK(hipEventRecord(b, 0)); CHECK(hipEventSynchronize(b));
        CHECK(hipEventElapsedTime(&ms, a, b));
    }
    std::vector<unsigned long long> h(8 * 4096);
    CHECK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(tvl_clock_probe), h.size() * 8));
    std::vector<double> ghz, pro_us, loop_us, epi_us, all_us, loopc;
    unsigned long long r_first = ~0ull, r_last = 0;
    for (int bl = 0; bl < std::min(nblk, 4096); ++bl) {   // the stamps of the blockIdx.y == 0 workgroups
        const unsigned long long* c = &h[8 * bl]; const unsigned long long* r = c + 4;
        ghz.push_back((double)(c[2] - c[1]) / (double)(r[2] - r[1]) * 0.1); loopc.push_back((double)(c[2] - c[1]));
        pro_us.push_back((r[1] - r[0]) * 0.01); loop_us.push_back((r[2] - r[1]) * 0.01); epi_us.push_back((r[3] - r[2]) * 0.01);
        all_us.push_back((r[3] - r[0]) * 0.01);
        r_first = std::min(r_first, r[0]); r_last = std::max(r_last, r[3]);
    }
    printf("%d x %d tiles, %d waves, %d LDS tiles, %d tiles x %d k ranges of %d steps: %.2f us per launch (event pair)\n", BM, BN, WAVES, NB, nblk, splits, per, ms / 5000 * 1e3);
    printf("  clock over the k loop %.3f GHz   prologue %.2f us   k loop %.2f us (%.0f cyc per step)   epilogue (slab stores) %.2f us   workgroup lifetime %.2f us\n",
           med(ghz), med(pro_us), med(loop_us), med(loopc) / per, med(epi_us), med(all_us));
    printf("  first workgroup entry to last workgroup exit (k range 0 only): %.2f us\n", (r_last - r_first) * 0.01);
}

template <int WAVES, int WN, int MINW, int PARTS = 2>
static void run(int m, int n, int k, const f16* x, const uint8_t* pw, const float* sc, const float* bi, f16* y) {
    constexpr int NB = 4;
    auto kern = gemm_mfma16_kernel<f16, f16, true, TVL_ACT_NONE, 1, false, NB, 256, WAVES, false, MINW, WN, PARTS>;
    const size_t lds = (size_t)NB * 256 * ROW_BYTES + 2048;   // + the step stamps
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int tiles_n = (n + 255) / 256, nblk = ((m + 255) / 256) * tiles_n;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float ms = 0;
    int launches = 0;
    for (int round = 0; round < 8; ++round) {   // ~8 x 2500 x 0.1 ms = 2 s
        CHECK(hipEventRecord(a, 0));
        for (int i = 0; i < 2500; ++i)
            hipLaunchKernelGGL(kern, dim3(nblk), dim3(64 * WAVES), lds, 0, x, m, k, (int64_t)k, pw, n, (int64_t)(k / 4), sc, bi, 100.0f, 0, y, (int64_t)n,
                               tiles_n, nblk, k / 64, 0);
        CHECK(hipEventRecord(b, 0)); CHECK(hipEventSynchronize(b));
        CHECK(hipEventElapsedTime(&ms, a, b));
        launches += 2500;
    }
    std::vector<unsigned long long> h(8 * 4096);
    CHECK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(tvl_clock_probe), h.size() * 8));
    std::vector<double> ghz, pro, loop, epi, pro_us, loop_us, epi_us, all_us;
    unsigned long long r_first = ~0ull, r_last = 0;
    for (int bl = 0; bl < std::min(nblk, 4096); ++bl) {
        const unsigned long long* c = &h[8 * bl]; const unsigned long long* r = c + 4;
        ghz.push_back((double)(c[2] - c[1]) / (double)(r[2] - r[1]) * 0.1);
        pro.push_back((double)(c[1] - c[0])); loop.push_back((double)(c[2] - c[1])); epi.push_back((double)(c[3] - c[2]));
        pro_us.push_back((r[1] - r[0]) * 0.01); loop_us.push_back((r[2] - r[1]) * 0.01); epi_us.push_back((r[3] - r[2]) * 0.01);
        all_us.push_back((r[3] - r[0]) * 0.01);
        r_first = std::min(r_first, r[0]); r_last = std::max(r_last, r[3]);
    }
    printf("waves %d x %d channels: %.2f us per launch (event pair, last round of %d launches)\n", WAVES, WN, ms / 2500 * 1e3, launches);
    printf("  clock over the k loop %.3f GHz (median of %d workgroups)\n", med(ghz), (int)ghz.size());
    printf("  prologue %.0f cyc / %.2f us   k loop %.0f cyc / %.2f us (%.0f cyc per step; matrix pipe %.0f = %.1f %%)   epilogue %.0f cyc / %.2f us   workgroup lifetime %.2f us\n",
           med(pro), med(pro_us), med(loop), med(loop_us), med(loop) / (k / 64), 2048.0, 100.0 * 2048.0 * (k / 64) / med(loop), med(epi), med(epi_us), med(all_us));
    printf("  first workgroup entry to last workgroup exit: %.2f us\n", (r_last - r_first) * 0.01);
}

// the persistent form (one workgroup per CU walks its tiles as one stream of k steps): per-launch time and the shader clocks wave 0
// spends in a tile switch (epilogue arithmetic, store issue, zeroing the accumulators, the next tile's origin)
template <int NB, int PARTS>
static void run_persistent(int m, int n, int k, const f16* x, const uint8_t* pw, const float* sc, const float* bi, f16* y) {
    auto kern = gemm_mfma16_kernel<f16, f16, true, TVL_ACT_NONE, 1, true, NB, 256, 8, false, 2, 32, PARTS>;
    const size_t lds = (size_t)NB * 256 * ROW_BYTES + persist_param_lds_bytes(8) + 2048;   // + the step stamps
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int tiles_n = (n + 255) / 256, nblk = ((m + 255) / 256) * tiles_n;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    float ms = 0;
    const int per = 500;
    for (int round = 0; round < 8; ++round) {
        CHECK(hipEventRecord(a, 0));
        for (int i = 0; i < per; ++i)
            hipLaunchKernelGGL(kern, dim3(256), dim3(512), lds, 0, x, m, k, (int64_t)k, pw, n, (int64_t)(k / 4), sc, bi, 100.0f, 0, y, (int64_t)n,
                               tiles_n, nblk, k / 64, 0);
        CHECK(hipEventRecord(b, 0)); CHECK(hipEventSynchronize(b));
        CHECK(hipEventElapsedTime(&ms, a, b));
    }
    std::vector<unsigned long long> h(8 * 4096), hs(2 * 4096);
    CHECK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(tvl_clock_probe), h.size() * 8));
    CHECK(hipMemcpyFromSymbol(hs.data(), HIP_SYMBOL(tvl_clock_probe_switch), hs.size() * 8));
    std::vector<double> ghz, loop, loop_us, sw, pro_us;
    for (int bl = 0; bl < 256; ++bl) {
        const unsigned long long* c = &h[8 * bl]; const unsigned long long* r = c + 4;
        ghz.push_back((double)(c[2] - c[1]) / (double)(r[2] - r[1]) * 0.1);
        loop.push_back((double)(c[2] - c[1])); loop_us.push_back((r[2] - r[1]) * 0.01); pro_us.push_back((r[1] - r[0]) * 0.01);
        if (hs[2 * bl + 1]) sw.push_back((double)hs[2 * bl] / (double)hs[2 * bl + 1]);
    }
    std::vector<unsigned long long> st(4 * 256);
    CHECK(hipMemcpyFromSymbol(st.data(), HIP_SYMBOL(tvl_clock_probe_steps), st.size() * 8));
    printf("  shader clocks from step to step, workgroup 0 (a tile switch follows every %d-th step):", k / 64);
    for (int i = 1; i < 256 && st[i]; ++i) printf("%s%llu", (i - 1) % (k / 64) == 0 ? " | " : " ", st[i] - st[i - 1]);
    printf("\n");
    const double tiles_per_wg = (double)nblk / 256, steps = k / 64;
    printf("persistent 256 x 256 (%d LDS tiles, %d phases per step), %d tiles of %d steps on 256 workgroups: %.2f us per launch (event pair)\n", NB, 2 * PARTS, nblk, k / 64, ms / per * 1e3);
    printf("  clock %.3f GHz   prologue %.2f us   step loop %.0f cyc / %.2f us per workgroup = %.0f cyc per tile, of which the tile switch %.0f cyc (%.2f us), so %.0f cyc per step\n",
           med(ghz), med(pro_us), med(loop), med(loop_us), med(loop) / tiles_per_wg, med(sw), med(sw) / med(ghz) * 1e-3,
           (med(loop) / tiles_per_wg - med(sw)) / steps);
}

int main(int argc, char** argv) {
    const int wide = argc > 1 ? atoi(argv[1]) : 0;
    const int m = argc > 4 ? atoi(argv[2]) : 4096, n = argc > 4 ? atoi(argv[3]) : 4096, k = argc > 4 ? atoi(argv[4]) : 4096;
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
    f16 *x, *y; uint8_t* pw; float *sc, *bi;
    CHECK(hipMalloc(&x, hx.size() * 2)); CHECK(hipMalloc(&y, (size_t)m * n * 2)); CHECK(hipMalloc(&pw, hw.size()));
    CHECK(hipMalloc(&sc, n * 4)); CHECK(hipMalloc(&bi, n * 4));
    CHECK(hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice)); CHECK(hipMemcpy(pw, hw.data(), hw.size(), hipMemcpyHostToDevice));
    CHECK(hipMemcpy(sc, hs.data(), n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(bi, hb.data(), n * 4, hipMemcpyHostToDevice));
    if (wide == 2) {   // gemm_clock 2 M N K splits
        const int splits = argc > 5 ? atoi(argv[5]) : 4;
        float* slab; CHECK(hipMalloc(&slab, (size_t)splits * m * n * 4));
        const int nb = argc > 6 ? atoi(argv[6]) : 4;   // LDS tiles: a tile is requested nb - 2 steps before its use
        if (nb == 24) run_mid_split<4, 1, 64, 8>(m, n, k, splits, x, pw, sc, bi, slab);   // lds_tiles = 24 / 26: 64 x 256 tiles of eight waves, four / six LDS tiles
        else if (nb == 26) run_mid_split<6, 1, 64, 8>(m, n, k, splits, x, pw, sc, bi, slab);
        else if (nb == 14) run_mid_split<4, 2>(m, n, k, splits, x, pw, sc, bi, slab);   // lds_tiles = 14: four tiles, eight waves
        else if (nb == 16) run_mid_split<6, 2>(m, n, k, splits, x, pw, sc, bi, slab);
        else if (nb == 6) run_mid_split<6>(m, n, k, splits, x, pw, sc, bi, slab);
        else if (nb == 5) run_mid_split<5>(m, n, k, splits, x, pw, sc, bi, slab);
        else if (nb == 3) run_mid_split<3>(m, n, k, splits, x, pw, sc, bi, slab);
        else run_mid_split<4>(m, n, k, splits, x, pw, sc, bi, slab);
        return 0;
    }
    if (wide == 3) { run_persistent<4, 4>(m, n, k, x, pw, sc, bi, y); return 0; }
    if (wide == 4) { run_persistent<3, 2>(m, n, k, x, pw, sc, bi, y); return 0; }
    if (wide == 5) { run<8, 32, 2, 4>(m, n, k, x, pw, sc, bi, y); return 0; }   // the one-tile-per-workgroup form in eight phases
    if (wide) run<4, 64, 1>(m, n, k, x, pw, sc, bi, y);
    else run<8, 32, 2>(m, n, k, x, pw, sc, bi, y);
    return 0;
}

// The C ABI of libternary_hip.so used from plain C++ with the HIP runtime only (no Python, no torch): latent float32
// weights -> tvl_quantize_pack -> tvl_ternary_matmul on f16 activations, checked against a scalar loop on the host.
// Build:  make -C examples        Run (needs an MI355X):  examples/cabi_demo
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "tvl.h"

#define HIP_OK(call)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 2; } \
    } while (0)
#define TVL_CALL(call)                                                                           \
    do {                                                                                         \
        if ((call) != TVL_OK) { std::fprintf(stderr, "%s: %s\n", #call, tvl_last_error()); return 3; } \
    } while (0)

int main() {
    const int64_t shapes[][3] = {{1, 512, 1024}, {24, 384, 512}, {300, 640, 256}, {37, 50, 101}};   // m, n, k
    std::printf("%s (version %d)\n", tvl_build_info(), tvl_version());
    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    for (const auto& s : shapes) {
        const int64_t m = s[0], n = s[1], k = s[2], kb = (k + 3) / 4;
        // deterministic inputs: activations on the grid j/64 (every partial sum exact in float32), latent weights in (-1, 1)
        std::vector<_Float16> x(m * k);
        std::vector<float> w(n * k), scale(n), bias(n);
        uint32_t lcg = 12345u + (uint32_t)(m * 31 + n * 7 + k);
        auto next = [&]() { lcg = lcg * 1664525u + 1013904223u; return lcg >> 8; };
        for (auto& v : x) v = (_Float16)(((int)(next() % 129) - 64) / 64.0f);
        for (auto& v : w) v = ((int)(next() % 2001) - 1000) / 1000.0f;
        for (auto& v : scale) v = 0.5f + (next() % 9) / 8.0f;
        for (auto& v : bias) v = ((int)(next() % 33) - 16) / 8.0f;

        _Float16 *dx, *dy;
        float *dw, *dscale, *dbias;
        uint8_t* dpw;
        void* dws = nullptr;
        HIP_OK(hipMalloc(&dx, x.size() * 2));
        HIP_OK(hipMalloc(&dy, m * n * 2));
        HIP_OK(hipMalloc(&dw, w.size() * 4));
        HIP_OK(hipMalloc(&dscale, n * 4));
        HIP_OK(hipMalloc(&dbias, n * 4));
        HIP_OK(hipMalloc(&dpw, n * kb));
        const size_t ws_bytes = tvl_ternary_matmul_workspace_bytes_for(TVL_F16, m, n, k);
        if (ws_bytes) HIP_OK(hipMalloc(&dws, ws_bytes));
        HIP_OK(hipMemcpyAsync(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice, stream));
        HIP_OK(hipMemcpyAsync(dw, w.data(), w.size() * 4, hipMemcpyHostToDevice, stream));
        HIP_OK(hipMemcpyAsync(dscale, scale.data(), n * 4, hipMemcpyHostToDevice, stream));
        HIP_OK(hipMemcpyAsync(dbias, bias.data(), n * 4, hipMemcpyHostToDevice, stream));

        TVL_CALL(tvl_quantize_pack(dw, TVL_F32, n, k, 0, TVL_Q_TERNARY, 0.5f, dpw, 0, stream));
        TVL_CALL(tvl_ternary_matmul(dx, TVL_F16, m, k, 0, dpw, n, 0, dscale, dbias, 100.0f, TVL_ACT_NONE, dy, TVL_F16, 0, dws, ws_bytes, stream));

        std::vector<_Float16> y(m * n);
        std::vector<uint8_t> pw(n * kb);
        HIP_OK(hipMemcpyAsync(y.data(), dy, y.size() * 2, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(pw.data(), dpw, pw.size(), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));

        // host check: ternarize at +-0.5 (inclusive), pack, select-and-sum, scale, bias, clip, one rounding to f16
        int64_t bad = 0;
        for (int64_t r = 0; r < n; ++r)
            for (int64_t j = 0; j < kb; ++j) {
                unsigned b = 0;
                for (int i = 0; i < 4; ++i) {
                    const int64_t kk = 4 * j + i;
                    const int t = kk < k ? (w[r * k + kk] >= 0.5f) - (w[r * k + kk] <= -0.5f) : 0;
                    b |= (unsigned)(t + 1) << (2 * i);
                }
                bad += pw[r * kb + j] != b;
            }
        for (int64_t i = 0; i < m; ++i)
            for (int64_t r = 0; r < n; ++r) {
                float acc = 0.0f;
                for (int64_t kk = 0; kk < k; ++kk) {
                    const float wv = w[r * k + kk];
                    if (wv >= 0.5f) acc += (float)x[i * k + kk];
                    else if (wv <= -0.5f) acc -= (float)x[i * k + kk];
                }
                float v = acc * scale[r];
                v = v + bias[r];
                v = std::fmin(std::fmax(v, -100.0f), 100.0f);
                bad += (float)y[i * n + r] != (float)(_Float16)v;
            }
        std::printf("m=%lld n=%lld k=%lld workspace=%zu B: %s\n", (long long)m, (long long)n, (long long)k, ws_bytes, bad ? "MISMATCH" : "bit-exact");
        for (void* p : {(void*)dx, (void*)dy, (void*)dw, (void*)dscale, (void*)dbias, (void*)dpw, dws})
            if (p) HIP_OK(hipFree(p));
        if (bad) return 1;
    }
    // a bad argument is reported, not fatal
    if (tvl_ternary_matmul(nullptr, TVL_F16, 4, 64, 0, nullptr, 8, 0, nullptr, nullptr, 1.0f, 7, nullptr, TVL_F16, 0, nullptr, 0, stream) != TVL_ERR_INVALID) return 4;
    std::printf("invalid call rejected: %s\n", tvl_last_error());
    HIP_OK(hipStreamDestroy(stream));
    std::printf("cabi_demo ok\n");
    return 0;
}

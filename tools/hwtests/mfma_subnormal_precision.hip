// Precision of v_mfma_f32_16x16x32_f16 when the A operand holds f16 subnormals: A = c << 2s (value c*4^s*2^-24)
// in every slot, B = N(0,1)-like f16 values, chained over STEPS MFMAs (K = 32*STEPS).  Prints, per bit position s,
// the worst |result * 2^24 * 4^-s - exact| over 16 columns x TRIALS, next to the same with A = 1.0 (normal).
// Build: hipcc --offload-arch=gfx950 -O2 mfma_subnormal_precision.hip -o mfma_subnormal_precision
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <random>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int STEPS = 128;  // K = 4096

__global__ void kern(const _Float16* x, uint16_t abits, float* out) {
    // x: [16 columns][K]; lane (c, g): column c, slots 8g..8g+7 of each step
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    half8 a;
    for (int j = 0; j < 8; ++j) a[j] = __builtin_bit_cast(_Float16, abits);
    floatx4 acc = {0, 0, 0, 0};
    for (int s = 0; s < STEPS; ++s) {
        half8 b;
        for (int j = 0; j < 8; ++j) b[j] = x[(size_t)c * (32 * STEPS) + 32 * s + 8 * g + j];
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    }
    if (g == 0) out[c] = acc[0];
}

int main() {
    const int K = 32 * STEPS, TRIALS = 8;
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    _Float16* dx; float* dout;
    hipMalloc(&dx, 16 * K * sizeof(_Float16)); hipMalloc(&dout, 64);
    double worst[7] = {0, 0, 0, 0, 0, 0, 0}, sum_mag = 0;
    for (int t = 0; t < TRIALS; ++t) {
        std::vector<_Float16> hx(16 * K);
        for (auto& v : hx) v = (_Float16)nd(rng);
        hipMemcpy(dx, hx.data(), hx.size() * sizeof(_Float16), hipMemcpyHostToDevice);
        double exact[16];
        for (int c = 0; c < 16; ++c) { exact[c] = 0; for (int k = 0; k < K; ++k) exact[c] += (double)hx[c * K + k]; sum_mag += fabs(exact[c]); }
        for (int v = 0; v < 7; ++v) {   // v = 0..5: subnormal with the code at bits 2v (v = 5: 0x0400, first normal); v = 6: A = 1.0
            const uint16_t abits = v < 6 ? (uint16_t)(1u << (2 * v)) : 0x3C00;
            hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, dx, abits, dout);
            float ho[16];
            hipMemcpy(ho, dout, 64, hipMemcpyDeviceToHost);
            const double scale = v < 6 ? ldexp(1.0, 24 - 2 * v) : 1.0;
            for (int c = 0; c < 16; ++c) worst[v] = fmax(worst[v], fabs((double)ho[c] * scale - exact[c]));
        }
    }
    printf("K = %d, mean |sum x| = %.2f\n", K, sum_mag / (16 * TRIALS));
    for (int v = 0; v < 6; ++v) printf("A = 1 << %2d (value 2^%d): worst |err| = %.3e\n", 2 * v, 2 * v - 24, worst[v]);
    printf("A = 1.0 (normal)        : worst |err| = %.3e\n", worst[6]);
    return 0;
}

// Does v_dot2_f32_f16 treat f16 subnormal inputs exactly?  A = pairs of 2-bit codes left in place at bits 2s..2s+1 of each
// half (value c * 4^s * 2^-24, subnormal for s <= 4), B = N(0,1)-like f16 pairs, chained over K/2 dot instructions per lane.
// Prints, per bit position s, the worst |result * 2^24 * 4^-s - exact| next to the same chain with A = (f16)c (normal).
// Build: hipcc --offload-arch=gfx950 -O2 dot2_subnormal.hip -o dot2_subnormal
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <random>
#include <vector>

typedef _Float16 half2v __attribute__((ext_vector_type(2)));
constexpr int K = 4096;

__global__ void kern(const uint32_t* a, const uint32_t* x, float* out) {
    const int lane = threadIdx.x;
    float acc = 0.0f;
    for (int i = 0; i < K / 2; ++i)
        acc = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, a[lane * (K / 2) + i]), __builtin_bit_cast(half2v, x[lane * (K / 2) + i]), acc, false);
    out[lane] = acc;
}

int main() {
    std::mt19937 rng(99);
    std::normal_distribution<float> nd(0.f, 1.f);
    uint32_t *da, *dx; float* dout;
    hipMalloc(&da, 64 * K * 2); hipMalloc(&dx, 64 * K * 2); hipMalloc(&dout, 256);
    std::vector<_Float16> hx(64 * K);
    std::vector<int> code(64 * K);
    for (auto& v : hx) v = (_Float16)nd(rng);
    for (auto& c : code) c = rng() & 3;
    hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    double exact[64], mag = 0;
    for (int l = 0; l < 64; ++l) { exact[l] = 0; for (int k = 0; k < K; ++k) exact[l] += code[l * K + k] * (double)hx[l * K + k]; mag += fabs(exact[l]) / 64; }
    printf("K = %d, mean |sum c x| = %.2f\n", K, mag);
    for (int s = 0; s <= 5; ++s) {   // s = 5: bits 10..11 are exponent bits - not linear, expected to fail
        std::vector<uint16_t> ha(64 * K);
        for (size_t i = 0; i < ha.size(); ++i) ha[i] = s < 5 ? (uint16_t)(code[i] << (2 * s)) : __builtin_bit_cast(uint16_t, (_Float16)(float)code[i]);
        hipMemcpy(da, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, da, dx, dout);
        float ho[64];
        hipMemcpy(ho, dout, 256, hipMemcpyDeviceToHost);
        const double scale = s < 5 ? ldexp(1.0, 24 - 2 * s) : 1.0;
        double worst = 0;
        for (int l = 0; l < 64; ++l) worst = fmax(worst, fabs((double)ho[l] * scale - exact[l]));
        if (s < 5) printf("codes at bits %d..%d (subnormal): worst |err| = %.3e   (lane 0: %.6f vs %.6f)\n", 2 * s, 2 * s + 1, worst, ho[0] * scale, exact[0]);
        else printf("A = (f16)code (normal)        : worst |err| = %.3e\n", worst);
    }
    return 0;
}

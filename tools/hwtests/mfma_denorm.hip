// Does v_mfma_f32_16x16x32_f16 / 32x32x16_f16 treat f16 subnormal inputs exactly (no flush to zero)?
// Prints PASS/FAIL per instruction.  Build: hipcc --offload-arch=gfx950 -O2 mfma_denorm.hip -o mfma_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx16 __attribute__((ext_vector_type(16)));

// A[row][k] bit pattern = pat[(row + k) % NP]; B = x[k] (same for all columns)
__constant__ uint16_t pats[8] = {0x0001, 0x0002, 0x0003, 0x000C, 0x0300, 0x0400, 0x0800, 0x0030};

__global__ void k16(const _Float16* x, float* out) {
    const int lane = threadIdx.x, c = lane & 15, g = lane >> 4;
    half8 a, b;
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * g + j;
        uint16_t bits = pats[(c + k) & 7];
        a[j] = __builtin_bit_cast(_Float16, bits);
        b[j] = x[k];
    }
    floatx4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    // D: col = lane & 15, row = 4 g + e
    for (int e = 0; e < 4; ++e) out[(4 * g + e) * 16 + c] = acc[e];
}

__global__ void k32(const _Float16* x, float* out) {
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    half8 a, b;
    for (int j = 0; j < 8; ++j) {
        const int k = 8 * h + j;
        uint16_t bits = pats[(r + k) & 7];
        a[j] = __builtin_bit_cast(_Float16, bits);
        b[j] = x[k];
    }
    floatx16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    for (int e = 0; e < 16; ++e) out[((e & 3) + 8 * (e >> 2) + 4 * h) * 32 + r] = acc[e];
}

int main() {
    const uint16_t hp[8] = {0x0001, 0x0002, 0x0003, 0x000C, 0x0300, 0x0400, 0x0800, 0x0030};
    _Float16 hx[32];
    for (int k = 0; k < 32; ++k) hx[k] = (_Float16)((k % 7 - 3) / 4.0f + 0.125f);
    _Float16* dx; float* dout;
    hipMalloc(&dx, sizeof(hx)); hipMalloc(&dout, 32 * 32 * 4);
    hipMemcpy(dx, hx, sizeof(hx), hipMemcpyHostToDevice);
    float ho[32 * 32];
    int bad;

    hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dx, dout);
    hipMemcpy(ho, dout, 16 * 16 * 4, hipMemcpyDeviceToHost);
    bad = 0;
    for (int row = 0; row < 16; ++row) {
        double ref = 0;
        for (int k = 0; k < 32; ++k) ref += (double)hp[(row + k) & 7] * ldexp(1.0, -24) * (double)hx[k];
        for (int c = 0; c < 16; ++c) if ((double)ho[row * 16 + c] != ref) { if (bad < 4) printf("k16 row %d col %d got %.10e want %.10e\n", row, c, ho[row * 16 + c], ref); ++bad; }
    }
    printf("mfma_f32_16x16x32_f16 subnormal inputs: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);

    hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, dx, dout);
    hipMemcpy(ho, dout, 32 * 32 * 4, hipMemcpyDeviceToHost);
    bad = 0;
    for (int row = 0; row < 32; ++row) {
        double ref = 0;
        for (int k = 0; k < 16; ++k) ref += (double)hp[(row + k) & 7] * ldexp(1.0, -24) * (double)hx[k];
        for (int c = 0; c < 32; ++c) if ((double)ho[row * 32 + c] != ref) { if (bad < 4) printf("k32 row %d col %d got %.10e want %.10e\n", row, c, ho[row * 32 + c], ref); ++bad; }
    }
    printf("mfma_f32_32x32x16_f16 subnormal inputs: %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
    return 0;
}

// Issue rate of the vector instructions a few-row packed product is made of: cycles per wave-instruction with W waves per SIMD,
// each wave running a long chain-free stream (8 independent accumulators).  Prints shader cycles per instruction per SIMD.
//   usage: valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) rate(const uint32_t* in, float* out, unsigned long long* cyc, int iters) {
    float acc[8];
    uint32_t a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { acc[i] = (float)i; a[i] = in[threadIdx.x + 64 * i]; b[i] = in[threadIdx.x + 64 * i + 512]; }
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (OP == 0) acc[i] = __builtin_fmaf(__builtin_bit_cast(float, a[i]), __builtin_bit_cast(float, b[i]), acc[i]);
                else if constexpr (OP == 1) acc[i] = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, a[i]), __builtin_bit_cast(half2v, b[i]), acc[i], false);
                else if constexpr (OP == 2) acc[i] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a[i]), __builtin_bit_cast(bf16x2, b[i]), acc[i], false);
                else if constexpr (OP == 3) { half2v r = __builtin_elementwise_fma(__builtin_bit_cast(half2v, a[i]), __builtin_bit_cast(half2v, b[i]), __builtin_bit_cast(half2v, acc[i])); acc[i] = __builtin_bit_cast(float, r); }
                else if constexpr (OP == 4) { uint32_t r = __builtin_amdgcn_perm(a[i], __builtin_bit_cast(uint32_t, acc[i]), b[i]); acc[i] = __builtin_bit_cast(float, r); }
                else if constexpr (OP == 5) { uint32_t r = (__builtin_bit_cast(uint32_t, acc[i]) & a[i]) | b[i]; acc[i] = __builtin_bit_cast(float, r); }
                else if constexpr (OP == 6) { float r; asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,1,0]" : "=v"(r) : "v"(a[i]), "v"(b[i]), "v"(acc[i])); acc[i] = r; }
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(const char* name, const uint32_t* in, float* out, unsigned long long* cyc) {
    for (int wps : {1, 2, 4, 8}) {   // waves per SIMD = blocks of 256 threads per CU
        const int blocks = 256 * wps, iters = 2000;
        hipLaunchKernelGGL(rate<OP>, dim3(blocks), dim3(256), 0, 0, in, out, cyc, iters);
        CHECK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(blocks);
        CHECK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
        double s = 0; for (auto v : h) s += (double)v;
        const double per_wave = s / blocks / (iters * 32.0);
        printf("%-22s %d waves/SIMD: %5.2f cycles per instruction per wave -> %5.2f per SIMD\n", name, wps, per_wave, per_wave / wps);
    }
}

int main() {
    uint32_t* in; float* out; unsigned long long* cyc;
    CHECK(hipMalloc(&in, 4096 * 4)); CHECK(hipMalloc(&out, 256 * 8 * 256 * 4)); CHECK(hipMalloc(&cyc, 256 * 8 * 8));
    std::vector<uint32_t> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = 0x3C003800u + (i * 2654435761u & 0x03FF03FFu);
    CHECK(hipMemcpy(in, h.data(), 4096 * 4, hipMemcpyHostToDevice));
    run<0>("v_fma_f32", in, out, cyc);
    run<1>("v_dot2(c)_f32_f16", in, out, cyc);
    run<2>("v_dot2(c)_f32_bf16", in, out, cyc);
    run<3>("v_pk_fma_f16", in, out, cyc);
    run<4>("v_perm_b32", in, out, cyc);
    run<5>("v_and_or_b32", in, out, cyc);
    run<6>("v_fma_mix_f32", in, out, cyc);
    return 0;
}

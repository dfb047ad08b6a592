// Issue rate of v_mfma_f32_16x16x32_f16 in the access pattern of the GEMM kernels' matrix waves: one wave per SIMD (or two), 64 matrix
// instructions per "step" on 8 x 4 accumulators (operands in registers), optionally one workgroup barrier per step and idle partner
// waves that only take part in the barrier.  Prints shader cycles per step (1 024 = 16 cycles per instruction) as the median over
// workgroups, stamped inside the kernel.
//   usage: mfma_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

// MW matrix waves (waves 0 .. MW-1) + IW idle waves; BAR: barriers per step; ORDER 0: acc[i][t] with t fastest (the kernels' order), 1: i fastest
// DATA 0: small integers / a coarse ramp (few bits toggle); 1: weights in {-1, 0, +1} (25 / 50 / 25 %), activations ~N(0,1) as f16 - the
// operand statistics of the packed-ternary GEMM; 2: both operands ~N(0,1)
template <int MW, int IW, int BAR, int ORDER, int ACCS, int DATA = 0>
__global__ void __launch_bounds__(64 * (MW + IW)) mfma_rate_kernel(int steps, unsigned long long* __restrict__ out, float* __restrict__ sink) {
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (wid >= MW) {
        for (int s = 0; s < steps * BAR; ++s) __builtin_amdgcn_s_barrier();
        return;
    }
    half8 a[4], b[8];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) a[t][e] = (_Float16)(((lane * 7 + t * 3 + e) % 5) - 2);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) b[i][e] = (_Float16)(0.01f * ((lane * 13 + i * 5 + e * 3) % 97) - 0.5f);
    if constexpr (DATA != 0) {
        uint64_t st = 0x9E3779B97F4A7C15ull * (threadIdx.x + 1 + 1024 * blockIdx.x);
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
        auto gauss = [&]() { const uint64_t r = rnd(); return (((r & 0xFFFF) + ((r >> 16) & 0xFFFF) + ((r >> 32) & 0xFFFF) + ((r >> 48) & 0xFFFF)) / 65536.0f - 2.0f) * 1.7f; };
        for (int t = 0; t < 4; ++t)
            for (int e = 0; e < 8; ++e) a[t][e] = DATA == 1 ? (_Float16)(float)((int)(rnd() % 4 == 0) - (int)(rnd() % 4 == 1)) : (_Float16)gauss();
        for (int i = 0; i < 8; ++i)
            for (int e = 0; e < 8; ++e) b[i][e] = (_Float16)gauss();
    }
    floatx4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[i][t] = (floatx4){0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int s = 0; s < steps; ++s) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {   // 2 x 32 = 64 matrix instructions
            if constexpr (ORDER == 0) {
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[i % ACCS][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t], b[i], acc[i % ACCS][t], 0, 0, 0);
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i % ACCS][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t], b[i], acc[i % ACCS][t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (BAR == 2 || (BAR == 1 && rep == 1)) __builtin_amdgcn_s_barrier();
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) r += acc[i][t][0] + acc[i][t][3];
    if (r == 1234.5f) sink[0] = r;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

// Additive study of the 256 x 256 tile's step (8 waves, each 16 row tiles x 2 channel tiles, one barrier per step, four phases of 16 matrix
// instructions): ADD bit 0: the activation fragments come from LDS (32 ds_read_b128 per wave and step, the kernel's swizzled conflict-free
// pattern, each phase's reads issued one phase ahead); bit 1: the weight fragments come from the kernel's 2-bit decode (40 vector
// instructions per wave and step) of a register that changes every step; bit 2: four LDS-DMA pieces per wave and step from an
// L2-resident buffer into the tile that was read two steps ago (counted wait, as in the kernel).
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int ADD, bool WIDE = false, int SCHED = 0, bool FAT = false>
__global__ void __launch_bounds__(FAT ? 256 : 512, FAT ? 1 : 2) step_study_kernel(int steps, const uint4* __restrict__ src, unsigned long long* __restrict__ out, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];   // 4 tiles of 32 KiB
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = lane & 15, g = lane >> 4, q = (0x2130 >> (4 * g)) & 3, sw = (c >> 1) & 7;
    uint32_t wreg[4] = {0x1A2B4C90u ^ (lane * 0x01010101u), 0x24681248u + lane, 0x12489A60u ^ lane, 0x0A185264u + 3 * lane};
    static_assert(!(WIDE && FAT), "one layout at a time");
    constexpr int RT = WIDE ? 4 : 8, CTL = (WIDE || FAT) ? 4 : 2;   // row tiles per phase, channel tiles per wave (FAT: 4 waves x (256 rows x 64 channels))
    constexpr int NTH = FAT ? 256 : 512, PIECES = FAT ? 8 : 4;
    half8 bfrag[2][CTL];   // [MFMA group s][channel tile]
    for (int i = tid; i < 4 * 32768 / 16; i += NTH) reinterpret_cast<uint4*>(lds)[i] = src[i & 4095];
    __syncthreads();
    floatx4 acc[2 * RT][CTL];
#pragma unroll
    for (int i = 0; i < 2 * RT; ++i)
#pragma unroll
        for (int t = 0; t < CTL; ++t) acc[i][t] = (floatx4){0.f, 0.f, 0.f, 0.f};
    half8 afA[RT], afB[RT];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) { afA[i][e] = (_Float16)(0.01f * ((lane * 13 + i * 5 + e * 3) % 97) - 0.5f); afB[i][e] = afA[i][e] * (_Float16)0.5f; }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < CTL; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) bfrag[s][t][e] = (_Float16)(((lane * 7 + t * 3 + e + s) % 3) - 1);
    auto read_a = [&](int buf, int s, int half, half8 (&af)[RT]) __attribute__((always_inline)) {
        if constexpr (ADD & 1) {
            const unsigned char* base = lds + buf * 32768 + (WIDE ? (wid >> 2) * 128 * 128 : 0) + c * 128 + (((2 * q + s) ^ sw) * 16) + half * RT * 16 * 128;
#pragma unroll
            for (int i = 0; i < RT; ++i) af[i] = *reinterpret_cast<const half8*>(base + i * 16 * 128);
        }
    };
    auto decode = [&](int s, int t) __attribute__((always_inline)) {
        if constexpr (ADD & 2) {
            const uint32_t d = wreg[t];
            uint32_t r[4];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const uint32_t rep = __builtin_amdgcn_perm(d, d, 0x0C000C00u | (2 * s + h) | ((2 * s + h) << 16));
                typedef _Float16 half2v __attribute__((ext_vector_type(2)));
                const uint32_t v01 = (rep & 0x000C0003u) | 0x64006400u, v23 = (rep & 0x00C00030u) | 0x64006400u;
                const half2v m01 = {(_Float16)1.0f, (_Float16)0.25f}, a01 = {(_Float16)-1025.0f, (_Float16)-257.0f};
                const half2v m23 = {(_Float16)0.0625f, (_Float16)0.015625f}, a23 = {(_Float16)-65.0f, (_Float16)-17.0f};
                r[2 * h] = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(__builtin_bit_cast(half2v, v01), m01, a01));
                r[2 * h + 1] = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(__builtin_bit_cast(half2v, v23), m23, a23));
            }
            uint4 u = make_uint4(r[0], r[1], r[2], r[3]);
            bfrag[s][t] = __builtin_bit_cast(half8, u);
        }
    };
    auto mma = [&](const half8 (&af)[RT], int half, int s) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
            for (int t = 0; t < CTL; ++t) acc[RT * half + i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bfrag[s][t], af[i], acc[RT * half + i][t], 0, 0, 0);
    };
    auto decode_phase = [&](int s, int ph) __attribute__((always_inline)) {   // the decodes of one phase: one fragment (narrow) or two (wide)
        if constexpr (WIDE || FAT) { decode(s, 2 * ph); decode(s, 2 * ph + 1); } else decode(s, ph);
    };
    const uint32_t lds_base = (uint32_t)(uintptr_t)(lptr_t)lds;
    const uint32_t goff = (uint32_t)(tid * 16);
    auto stage = [&](int buf) __attribute__((always_inline)) {
        if constexpr (ADD & 4) {
#pragma unroll
            for (int i = 0; i < PIECES; ++i) {
                const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_base + buf * 32768 + (i * NTH + wid * 64) * 16);
                const uint4* sbase = src + i * NTH;
                uint32_t keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(goff), "s"(dst), "s"(sbase) : "memory");
            }
        }
    };
    stage(1); stage(2);
    read_a(0, 0, 0, afA);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int st = 0; st < steps; ++st) {
        const int buf = st & 3, buf1 = (st + 1) & 3;
        // SCHED 0: a phase's fragment reads first (pinned), then its 16 matrix instructions with the decode among them (the kernels' form);
        // SCHED 1: reads, decode and matrix instructions of a phase interleaved by the scheduler hint: 1 read, 2 matrix instructions, some vector work
        auto pin = [&]() __attribute__((always_inline)) { if constexpr (SCHED == 0) __builtin_amdgcn_sched_barrier(0); };
        auto hint = [&]() __attribute__((always_inline)) {
            if constexpr (SCHED == 1) {
                constexpr int GROUPS = WIDE ? 4 : 8, MF = RT * CTL / GROUPS, VA = (ADD & 2) ? (WIDE ? 10 : (FAT ? 5 : 3)) : 0, DS = (ADD & 1) ? 1 : 0;
#pragma unroll
                for (int j = 0; j < GROUPS; ++j) {
                    __builtin_amdgcn_sched_group_barrier(0x100, DS, 0);   // DS read
                    __builtin_amdgcn_sched_group_barrier(0x008, MF, 0);   // matrix instructions
                    __builtin_amdgcn_sched_group_barrier(0x002, VA, 0);   // decode VALU
                }
            }
        };
        read_a(buf, 0, 1, afB);
        pin();
        decode_phase(1, 0); mma(afA, 0, 0);
        hint();
        __builtin_amdgcn_sched_barrier(0);
        read_a(buf, 1, 0, afA);
        pin();
        decode_phase(1, 1); mma(afB, 1, 0);
        hint();
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (ADD & 4) { if constexpr (FAT) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        read_a(buf, 1, 1, afB);
        stage((st + 3) & 3);
        wreg[0] = wreg[0] * 1664525u + 1013904223u; wreg[1] ^= wreg[0] >> 3; wreg[2] += wreg[1]; wreg[3] ^= wreg[2] << 1;
        pin();
        decode_phase(0, 0); mma(afA, 0, 1);
        hint();
        __builtin_amdgcn_sched_barrier(0);
        read_a(buf1, 0, 0, afA);
        pin();
        decode_phase(0, 1); mma(afB, 1, 1);
        hint();
        __builtin_amdgcn_sched_barrier(0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if constexpr (ADD & 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float r = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * RT; ++i)
#pragma unroll
        for (int t = 0; t < CTL; ++t) r += acc[i][t][0] + acc[i][t][3];
    if (r == 1234.5f) sink[0] = r;
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

template <int ADD, bool WIDE = false, int SCHED = 0, bool FAT = false>
static void run_study(const char* name, const uint4* src, unsigned long long* d_out, float* sink) {
    const int steps = 8000, wgs = 256;
    auto kern = step_study_kernel<ADD, WIDE, SCHED, FAT>;
    const int nth = FAT ? 256 : 512, wv = FAT ? 4 : 8, per_wave = FAT ? 128 : 64;
    CHECK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 32768));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(wgs), dim3(nth), 4 * 32768, 0, steps, src, d_out, sink);
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kern, dim3(wgs), dim3(nth), 4 * 32768, 0, steps, src, d_out, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double us_per_step = ms * 1e3 / 2.0 / steps;
    std::vector<unsigned long long> h(wgs);
    CHECK(hipMemcpy(h.data(), d_out, wgs * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    printf("%-86s %6.3f us per step  %7.1f TFLOP/s  %6.0f cycles per step (%.2f GHz)\n", name, us_per_step, 256.0 * wv * per_wave * 16384.0 / (us_per_step * 1e-6) / 1e12,
           (double)h[wgs / 2] / steps, (double)h[wgs / 2] / steps / us_per_step * 1e-3);
}

template <int MW, int IW, int BAR, int ORDER, int ACCS, int DATA = 0>
static void run(const char* name, unsigned long long* d_out, float* sink) {
    const int steps = 20000, wgs = 256;   // ~10 ms per launch: long enough for the clock to settle
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((mfma_rate_kernel<MW, IW, BAR, ORDER, ACCS, DATA>), dim3(wgs), dim3(64 * (MW + IW)), 0, 0, steps, d_out, sink);
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((mfma_rate_kernel<MW, IW, BAR, ORDER, ACCS, DATA>), dim3(wgs), dim3(64 * (MW + IW)), 0, 0, steps, d_out, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipDeviceSynchronize());
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double tflops = 3.0 * wgs * MW * 64.0 * steps * 16384.0 / (ms * 1e-3) / 1e12;
    std::vector<unsigned long long> h(wgs);
    CHECK(hipMemcpy(h.data(), d_out, wgs * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    printf("%-78s %7.1f cycles per 64 matrix instructions per wave (%.2f per instruction)  %7.1f TFLOP/s by the event clock\n", name,
           (double)h[wgs / 2] / steps, (double)h[wgs / 2] / steps / 64, tflops);
}

int main() {
    unsigned long long* d_out; float* sink;
    CHECK(hipMalloc(&d_out, 256 * 8)); CHECK(hipMalloc(&sink, 64));
    run<4, 0, 0, 0, 8>("4 waves (1 per SIMD), no barrier, channel tile fastest", d_out, sink);
    run<4, 0, 0, 1, 8>("4 waves (1 per SIMD), no barrier, row tile fastest", d_out, sink);
    run<4, 0, 0, 0, 1>("4 waves, no barrier, 4 accumulators only (dependent every 4th)", d_out, sink);
    run<4, 0, 1, 0, 8>("4 waves, one barrier per step", d_out, sink);
    run<4, 0, 2, 0, 8>("4 waves, two barriers per step", d_out, sink);
    run<4, 4, 1, 0, 8>("4 matrix waves + 4 idle waves, one barrier per step", d_out, sink);
    run<8, 0, 0, 0, 8>("8 waves (2 per SIMD), no barrier", d_out, sink);
    run<8, 0, 1, 0, 8>("8 waves (2 per SIMD), one barrier per step", d_out, sink);
    run<8, 0, 1, 0, 8, 1>("8 waves, one barrier per step, ternary weights x N(0,1) activations", d_out, sink);
    run<8, 0, 1, 0, 8, 2>("8 waves, one barrier per step, N(0,1) x N(0,1)", d_out, sink);
    run<4, 4, 1, 0, 8, 1>("4 matrix + 4 idle waves, one barrier per step, ternary x N(0,1)", d_out, sink);
    std::vector<uint16_t> hs(4096 * 8);
    uint64_t st = 88172645463325252ull;
    for (auto& v : hs) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; const _Float16 f = (_Float16)((((st & 0xFFFF) + ((st >> 16) & 0xFFFF) + ((st >> 32) & 0xFFFF) + ((st >> 48) & 0xFFFF)) / 65536.0f - 2.0f) * 1.7f); v = *reinterpret_cast<const uint16_t*>(&f); }
    uint4* src; CHECK(hipMalloc(&src, 65536)); CHECK(hipMemcpy(src, hs.data(), 65536, hipMemcpyHostToDevice));
    printf("--- the 256 x 256 tile's step, built up (8 waves, 64 matrix instructions per wave and step, one barrier per step; 1.31 us per step in the shipped kernel)\n");
    run_study<0>("matrix instructions + barrier only", src, d_out, sink);
    run_study<1>("+ activation fragments from LDS (32 ds_read_b128 per wave and step)", src, d_out, sink);
    run_study<2>("+ 2-bit decode (40 vector instructions per wave and step), fragments in registers", src, d_out, sink);
    run_study<3>("+ LDS fragments + decode", src, d_out, sink);
    run_study<4>("+ LDS-DMA pieces only (4 per wave and step)", src, d_out, sink);
    run_study<5>("+ LDS fragments + LDS-DMA", src, d_out, sink);
    run_study<7>("+ LDS fragments + decode + LDS-DMA (the kernel's step without its weight loads)", src, d_out, sink);
    printf("--- the same with waves of 128 rows x 64 channels (2 x 4 waves: half the fragment reads, twice the decode)\n");
    run_study<1, true>("+ activation fragments from LDS (16 ds_read_b128 per wave and step)", src, d_out, sink);
    run_study<3, true>("+ LDS fragments + decode (80 vector instructions per wave and step)", src, d_out, sink);
    run_study<7, true>("+ LDS fragments + decode + LDS-DMA", src, d_out, sink);
    printf("--- fragment reads, decode and matrix instructions of a phase interleaved by sched_group_barrier instead of reads first\n");
    run_study<1, false, 1>("+ LDS fragments", src, d_out, sink);
    run_study<3, false, 1>("+ LDS fragments + decode", src, d_out, sink);
    run_study<7, false, 1>("+ LDS fragments + decode + LDS-DMA", src, d_out, sink);
    run_study<7, true, 1>("128 x 64 waves: + LDS fragments + decode + LDS-DMA", src, d_out, sink);
    printf("--- four waves of 256 rows x 64 channels (one per SIMD, 256 accumulator registers: half the fragment reads, the same decode)\n");
    run_study<0, false, 0, true>("matrix instructions + barrier only", src, d_out, sink);
    run_study<1, false, 0, true>("+ LDS fragments (32 per wave and step), reads first", src, d_out, sink);
    run_study<7, false, 0, true>("+ LDS fragments + decode + LDS-DMA, reads first", src, d_out, sink);
    run_study<1, false, 1, true>("+ LDS fragments, interleaved", src, d_out, sink);
    run_study<3, false, 1, true>("+ LDS fragments + decode, interleaved", src, d_out, sink);
    run_study<7, false, 1, true>("+ LDS fragments + decode + LDS-DMA, interleaved", src, d_out, sink);
    return 0;
}

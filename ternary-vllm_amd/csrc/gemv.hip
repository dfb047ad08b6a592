// Packed-ternary x activation product for very few activation rows (decode: m <= 8), one launch, no split-K.
//
// One wave owns CH output channels and walks the whole k range: per iteration of 2048 k every lane loads the 8 packed
// bytes (32 weights) of each of its channels - a wave-instruction reads 512 contiguous bytes of one row - and the
// matching 32 activations of every row (the activations are a few KiB: they come from L1/L2 after the first wave).
// The 2-bit codes are decoded in registers to f16 pairs (v_perm / v_and_or / v_pk_fma, 1.25 VALU per weight, shared
// by the rows) and multiplied with v_dot2c_f32_f16 / _bf16 (float32 accumulation; float32 activations: v_fma_f32).
// The lanes' partial sums are added over the wave with DPP row shifts in a fixed order, then the fused epilogue.
// No LDS, no barrier, no workspace: many small waves keep enough weight bytes in flight (32 waves per CU), and a
// 4096 x 4096 layer is ONE ~3 us launch where the split-K kernels need two launches (~8 us).
// The price is VALU work proportional to the row count, so gemv_supported() hands larger m x n x k to the
// matrix-core streaming kernel.  Requires k % 32 == 0, 16-byte aligned activation rows, 8-byte aligned packed rows.
#include "tvl_common.h"
#include "tvl_kernels.h"

#include <algorithm>

namespace tvl {

typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// 4 codes of byte BYTE of `d` -> two registers of f16 pairs (w0, w1), (w2, w3); see gemm_mfma.hip for the arithmetic
template <int BYTE>
__device__ __forceinline__ void gv_decode4(uint32_t d, uint32_t& r01, uint32_t& r23) {
    const uint32_t rep = __builtin_amdgcn_perm(d, d, 0x0C000C00u | BYTE | (BYTE << 16));
    const uint32_t v01 = (rep & 0x000C0003u) | 0x64006400u;
    const uint32_t v23 = (rep & 0x00C00030u) | 0x64006400u;
    const half2v m01 = {(_Float16)1.0f, (_Float16)0.25f}, a01 = {(_Float16)-1025.0f, (_Float16)-257.0f};
    const half2v m23 = {(_Float16)0.0625f, (_Float16)0.015625f}, a23 = {(_Float16)-65.0f, (_Float16)-17.0f};
    r01 = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(__builtin_bit_cast(half2v, v01), m01, a01));
    r23 = __builtin_bit_cast(uint32_t, __builtin_elementwise_fma(__builtin_bit_cast(half2v, v23), m23, a23));
}

template <int v>
__device__ __forceinline__ float gv_dpp_add(float x) {
    return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), v, 0xF, 0xF, true));
}
// sum over the 64 lanes in a fixed order (row-wise shifts, then the four row totals)
__device__ __forceinline__ float gv_wave_sum(float x) {
    x = gv_dpp_add<0x111>(x);
    x = gv_dpp_add<0x112>(x);
    x = gv_dpp_add<0x114>(x);
    x = gv_dpp_add<0x118>(x);
    const int xi = __builtin_bit_cast(int, x);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 15)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 31));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 47)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(xi, 63));
    return (r0 + r1) + (r2 + r3);
}

// 32 activations of one row -> 16 registers of pairs in the multiply type (f16 / bf16 pairs as loaded; f32: 32 floats)
template <typename TX> struct XVec { uint32_t r[16]; };
template <> struct XVec<float> { float r[32]; };

template <typename TX>
__device__ __forceinline__ void load_x32(const TX* p, XVec<TX>& v) {
    if constexpr (sizeof(TX) == 2) {
        const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
        for (int i = 0; i < 4; ++i) { const uint4 t = q[i]; v.r[4 * i] = t.x; v.r[4 * i + 1] = t.y; v.r[4 * i + 2] = t.z; v.r[4 * i + 3] = t.w; }
    } else {
        const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float4 t = q[i]; v.r[4 * i] = t.x; v.r[4 * i + 1] = t.y; v.r[4 * i + 2] = t.z; v.r[4 * i + 3] = t.w; }
    }
}

template <typename TX, typename TY, int CH, int ROWS>
__global__ void __launch_bounds__(256) gemv_kernel(const TX* __restrict__ x, int m, int k, int64_t ld_x, const uint8_t* __restrict__ pw, int n,
                                                   int64_t ld_pw, const float* __restrict__ scale, const float* __restrict__ bias, float clip,
                                                   int act, TY* __restrict__ y, int64_t ld_y) {
    const int lane = threadIdx.x & 63;
    const int ch0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * CH;   // wave-uniform
    if (ch0 >= n) return;
    const uint8_t* wrow[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int ch = ch0 + c < n ? ch0 + c : n - 1;
        wrow[c] = pw + (int64_t)ch * ld_pw + 8 * lane;
    }
    float acc[ROWS][CH];
#pragma unroll
    for (int r = 0; r < ROWS; ++r)
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[r][c] = 0.0f;

    for (int k0 = 32 * lane; k0 < k; k0 += 2048) {
        uint2 wq[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) wq[c] = *reinterpret_cast<const uint2*>(wrow[c] + (k0 - 32 * lane) / 4);
        uint32_t wd[CH][16];   // 32 decoded weights per channel as f16 (or bf16) pairs
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            gv_decode4<0>(wq[c].x, wd[c][0], wd[c][1]);   gv_decode4<1>(wq[c].x, wd[c][2], wd[c][3]);
            gv_decode4<2>(wq[c].x, wd[c][4], wd[c][5]);   gv_decode4<3>(wq[c].x, wd[c][6], wd[c][7]);
            gv_decode4<0>(wq[c].y, wd[c][8], wd[c][9]);   gv_decode4<1>(wq[c].y, wd[c][10], wd[c][11]);
            gv_decode4<2>(wq[c].y, wd[c][12], wd[c][13]); gv_decode4<3>(wq[c].y, wd[c][14], wd[c][15]);
            if constexpr (__is_same(TX, bf16_t)) {
#pragma unroll
                for (int i = 0; i < 16; ++i) wd[c][i] |= (wd[c][i] >> 3) & 0x03800380u;   // f16 patterns of {-1,0,1,2} -> bf16
            }
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            if (r < m) {
                XVec<TX> xv;
                load_x32<TX>(x + (int64_t)r * ld_x + k0, xv);
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    float a = acc[r][c];
                    if constexpr (__is_same(TX, f16)) {
#pragma unroll
                        for (int i = 0; i < 16; ++i)
                            a = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, wd[c][i]), __builtin_bit_cast(half2v, xv.r[i]), a, false);
                    } else if constexpr (__is_same(TX, bf16_t)) {
#pragma unroll
                        for (int i = 0; i < 16; ++i)
                            a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, wd[c][i]), __builtin_bit_cast(bf16x2, xv.r[i]), a, false);
                    } else {
#pragma unroll
                        for (int i = 0; i < 16; ++i) {
                            const half2v w2 = __builtin_bit_cast(half2v, wd[c][i]);
                            a = __builtin_fmaf((float)w2[0], xv.r[2 * i], a);
                            a = __builtin_fmaf((float)w2[1], xv.r[2 * i + 1], a);
                        }
                    }
                    acc[r][c] = a;
                }
            }
        }
    }

    const float cl = fabsf(clip);
    const bool has_bias = bias != nullptr;
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
        if (r < m) {
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const float s = gv_wave_sum(acc[r][c]);
                const int ch = ch0 + c;
                if (lane == 0 && ch < n)
                    y[(int64_t)r * ld_y + ch] = from_float<TY>(epilogue_rt(s, scale[ch], has_bias ? bias[ch] : 0.0f, has_bias, act, cl));
            }
        }
    }
}

// This kernel's time is ~ launch + max(bytes / stream rate, weights * (1.25 + a * m) VALU lane-operations / chip rate);
// the split-K matrix-core kernels cost two launches plus the stream.  Few rows or a small layer: this kernel.
bool gemv_supported(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw) {
    if (x_dtype != TVL_F16 && x_dtype != TVL_BF16 && x_dtype != TVL_F32) return false;
    if (m < 1 || m > 8 || k <= 0 || k % 32 != 0) return false;
    const int64_t esz = x_dtype == TVL_F32 ? 4 : 2;
    if (!aligned(x, 16) || (ld_x * esz) % 16 != 0) return false;
    if (!aligned(pw, 8) || ld_pw % 8 != 0) return false;
    if (n > INT32_MAX / 2 || k > INT32_MAX) return false;
    const double per_weight = (x_dtype == TVL_F32 ? 2.25 + 2.0 * m : (x_dtype == TVL_BF16 ? 2.25 : 1.25) + 0.5 * m);
    const double weights = (double)n * (double)k;
    const double t_gemv = 2.5 + std::max(weights / 4.0 / 3.8e6, weights * per_weight / 39.0e6);       // microseconds
    const double t_other = x_dtype == TVL_F32 ? 25.0 + weights * 3.0 * 2.0 * 32.0 / 1.0e9               // split + bf16 GEMM over 3K (32-row tiles)
                                               : 6.0 + weights / 4.0 / 3.4e6;                            // streaming kernel + reduce launch
    return t_gemv <= t_other;
}

template <typename TX, typename TY>
static int launch_gemv_xy(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                          const float* bias, float clip, int act, TY* y, int64_t ld_y, hipStream_t st) {
    const dim3 block(256);
    if (m <= 2) {
        constexpr int CH = 2;
        hipLaunchKernelGGL((gemv_kernel<TX, TY, CH, 2>), dim3((unsigned)ceil_div(n, 4 * CH)), block, 0, st, x, (int)m, (int)k, ld_x, pw, (int)n, ld_pw,
                           scale, bias, clip, act, y, ld_y);
    } else if (m <= 4) {
        constexpr int CH = 2;
        hipLaunchKernelGGL((gemv_kernel<TX, TY, CH, 4>), dim3((unsigned)ceil_div(n, 4 * CH)), block, 0, st, x, (int)m, (int)k, ld_x, pw, (int)n, ld_pw,
                           scale, bias, clip, act, y, ld_y);
    } else {
        constexpr int CH = 2;
        hipLaunchKernelGGL((gemv_kernel<TX, TY, CH, 8>), dim3((unsigned)ceil_div(n, 4 * CH)), block, 0, st, x, (int)m, (int)k, ld_x, pw, (int)n, ld_pw,
                           scale, bias, clip, act, y, ld_y);
    }
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

template <typename TX>
static int launch_gemv_x(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                         const float* bias, float clip, int act, void* y, int y_dtype, int64_t ld_y, hipStream_t st) {
    switch (y_dtype) {
        case TVL_F32: return launch_gemv_xy(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, (float*)y, ld_y, st);
        case TVL_F16: return launch_gemv_xy(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, (f16*)y, ld_y, st);
        case TVL_BF16: return launch_gemv_xy(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, (bf16_t*)y, ld_y, st);
        default: set_error("ternary_matmul: unsupported output dtype %d", y_dtype); return TVL_ERR_INVALID;
    }
}

int launch_gemv(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                const float* bias, float clip, int act, void* y, int y_dtype, int64_t ld_y, hipStream_t st) {
    if (x_dtype == TVL_F16) return launch_gemv_x((const f16*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, y, y_dtype, ld_y, st);
    if (x_dtype == TVL_BF16) return launch_gemv_x((const bf16_t*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, y, y_dtype, ld_y, st);
    return launch_gemv_x((const float*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, y, y_dtype, ld_y, st);
}

}  // namespace tvl

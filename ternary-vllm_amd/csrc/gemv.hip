// Packed-ternary x activation product for very few activation rows (decode: m <= 4), one launch, no split-K.
//
// One wave owns CH output channels and walks the whole k range: per iteration of 2048 k every lane loads the 8 packed
// bytes (32 weights) of each of its channels - a wave-instruction reads 512 contiguous bytes of one row - and ONCE
// the matching 32 activations of every row, which then serve all CH channels from registers.
// Decode is a table lookup: a packed byte indexes a 256-entry LDS table of its four weights, ready in the multiply
// type (two f16 / bf16 pairs, or four floats) - one shift-by-SDWA and one ds_read per 4 weights instead of 5 VALU
// instructions, which is what bounds a one-row product on the vector ALUs.  Two copies of the table (odd / even
// lanes) halve the bank conflicts of the data-dependent addresses.  The products go through v_dot2c_f32_f16 / _bf16
// (float32 accumulation; float32 activations: v_fma_f32), the lanes' partial sums are added over the wave with DPP
// row shifts in a fixed order, then the fused epilogue.  No barrier in the loop, no workspace: a 4096 x 4096 layer
// is ONE short launch where the split-K kernels need two.
// The VALU work still grows with the row count, so gemv_supported() hands larger m x n x k to the matrix-core
// streaming kernel.  Requires k % 32 == 0, 16-byte aligned activation rows, 8-byte aligned packed rows.
#include "tvl_common.h"
#include "tvl_kernels.h"

#include <stdlib.h>

namespace tvl {

typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t nt_u32x2 __attribute__((ext_vector_type(2)));

// table entry of one packed byte: its four weights (code - 1; code 3 -> +2) in the multiply type
template <typename TX> struct LutEntry { uint32_t w01, w23; };          // two pairs of f16 / bf16
template <> struct LutEntry<float> { float w[4]; };

template <typename TX>
__device__ __forceinline__ LutEntry<TX> make_entry(uint32_t byte) {
    LutEntry<TX> e;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)((int)((byte >> (2 * i)) & 3u) - 1);
    if constexpr (sizeof(TX) == 4) {
#pragma unroll
        for (int i = 0; i < 4; ++i) e.w[i] = v[i];
    } else {
        const TX h[4] = {from_float<TX>(v[0]), from_float<TX>(v[1]), from_float<TX>(v[2]), from_float<TX>(v[3])};
        e.w01 = (uint32_t)__builtin_bit_cast(uint16_t, h[0]) | ((uint32_t)__builtin_bit_cast(uint16_t, h[1]) << 16);
        e.w23 = (uint32_t)__builtin_bit_cast(uint16_t, h[2]) | ((uint32_t)__builtin_bit_cast(uint16_t, h[3]) << 16);
    }
    return e;
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

template <typename TX, typename TY, int CH, int ROWS, int WAVES, bool KSPLIT>
__global__ void __launch_bounds__(64 * WAVES) gemv_kernel(const TX* __restrict__ x, int m, int k, int64_t ld_x, const uint8_t* __restrict__ pw,
                                                          int n, int64_t ld_pw, const float* __restrict__ scale, const float* __restrict__ bias,
                                                          float clip, int act, int act_begin, TY* __restrict__ y, int64_t ld_y) {
    __shared__ LutEntry<TX> lut[2][256];
    const int lane = threadIdx.x & 63;

    // KSPLIT: the block's waves share CH channels and take k ranges 2048 * (wave + WAVES * i); otherwise each wave has
    // its own CH channels and the whole k range
    const int wave = threadIdx.x >> 6;
    const int ch0 = KSPLIT ? blockIdx.x * CH : (blockIdx.x * WAVES + wave) * CH;   // wave-uniform
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

    const int kstart = (KSPLIT ? 2048 * wave : 0) + 32 * lane, kstride = KSPLIT ? 2048 * WAVES : 2048;
    uint2 wq[CH];
    XVec<TX> xv[ROWS];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {   // once-read weight stream: non-temporal
            const nt_u32x2 t = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x2*>(wrow[c] + (k0 - 32 * lane) / 4));
            wq[c] = make_uint2(t[0], t[1]);
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r) load_x32<TX>(x + (int64_t)(r < m ? r : 0) * ld_x + k0, xv[r]);
    };
    // the first step's loads are in flight while the block fills its tables
    int k0 = kstart;
    if (k0 < k) fetch(k0);
    for (int i = threadIdx.x; i < 512; i += 64 * WAVES) lut[i >> 8][i & 255] = make_entry<TX>(i & 255);
    __syncthreads();
    const LutEntry<TX>* my_lut = lut[lane & 1];
    if (!KSPLIT && ch0 >= n) return;

    while (k0 < k) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            LutEntry<TX> e[8];   // 32 weights of channel c
#pragma unroll
            for (int b = 0; b < 8; ++b) e[b] = my_lut[((b < 4 ? wq[c].x : wq[c].y) >> (8 * (b & 3))) & 0xFFu];
#pragma unroll
            for (int r = 0; r < ROWS; ++r) {
                // two accumulation chains per output keep dependent dot instructions apart
                float a0 = acc[r][c], a1 = 0.0f;
                if constexpr (__is_same(TX, f16)) {
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        a0 = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, e[b].w01), __builtin_bit_cast(half2v, xv[r].r[2 * b]), a0, false);
                        a1 = __builtin_amdgcn_fdot2(__builtin_bit_cast(half2v, e[b].w23), __builtin_bit_cast(half2v, xv[r].r[2 * b + 1]), a1, false);
                    }
                } else if constexpr (__is_same(TX, bf16_t)) {
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        a0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, e[b].w01), __builtin_bit_cast(bf16x2, xv[r].r[2 * b]), a0, false);
                        a1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, e[b].w23), __builtin_bit_cast(bf16x2, xv[r].r[2 * b + 1]), a1, false);
                    }
                } else {
#pragma unroll
                    for (int b = 0; b < 8; ++b) {
                        a0 = __builtin_fmaf(e[b].w[0], xv[r].r[4 * b], a0);
                        a1 = __builtin_fmaf(e[b].w[1], xv[r].r[4 * b + 1], a1);
                        a0 = __builtin_fmaf(e[b].w[2], xv[r].r[4 * b + 2], a0);
                        a1 = __builtin_fmaf(e[b].w[3], xv[r].r[4 * b + 3], a1);
                    }
                }
                acc[r][c] = a0 + a1;
            }
        }
        k0 += kstride;
        if (k0 < k) fetch(k0);
    }

    const float cl = fabsf(clip);
    const bool has_bias = bias != nullptr;
    __shared__ float part[KSPLIT ? WAVES : 1][ROWS * CH];
#pragma unroll
    for (int r = 0; r < ROWS; ++r) {
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            const float s = gv_wave_sum(acc[r][c]);
            const int ch = ch0 + c;
            if constexpr (KSPLIT) {
                if (lane == 0) part[wave][r * CH + c] = s;
            } else {
                if (lane == 0 && r < m && ch < n)
                    y[(int64_t)r * ld_y + ch] = from_float<TY>(epilogue_rt(s, scale[ch], has_bias ? bias[ch] : 0.0f, has_bias, act, cl, ch, act_begin));
            }
        }
    }
    if constexpr (KSPLIT) {
        __syncthreads();
        if (threadIdx.x < ROWS * CH) {   // the waves' partial sums, added in wave order
            const int r = threadIdx.x / CH, ch = ch0 + threadIdx.x % CH;
            float s = part[0][threadIdx.x];
#pragma unroll
            for (int w = 1; w < WAVES; ++w) s += part[w][threadIdx.x];
            if (r < m && ch < n)
                y[(int64_t)r * ld_y + ch] = from_float<TY>(epilogue_rt(s, scale[ch], has_bias ? bias[ch] : 0.0f, has_bias, act, cl, ch, act_begin));
        }
    }
}

// Chosen against the two-launch split-K kernels by measured cost (microseconds; W = n * k / 1e6; tools/gemv_audit.sh times the library's
// choice beside this kernel switched off / forced): this kernel ~ 2.5 + W * (0.045 + 0.0255 m) * g with g = 1.4 / 1.3 / 0.92 for its three
// launch geometries, the alternatives as fitted below.  One row: always this kernel; two rows: except on layers of 16 384 channels and
// more with a short k (2 x 28672 x 4096: 16.2 against 14.5 us).  TVL_GEMV=0 / 2 switches it off / on wherever it can run (tuning).
bool gemv_supported(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw) {
    if (x_dtype != TVL_F16 && x_dtype != TVL_BF16 && x_dtype != TVL_F32) return false;
    if (m < 1 || m > 4 || k <= 0 || k % 32 != 0) return false;
    const int64_t esz = x_dtype == TVL_F32 ? 4 : 2;
    if (!aligned(x, 16) || (ld_x * esz) % 16 != 0) return false;
    if (!aligned(pw, 8) || ld_pw % 8 != 0) return false;
    if (n > INT32_MAX / 2 || k > INT32_MAX) return false;
    static const int mode = [] { const char* e = getenv("TVL_GEMV"); return e ? atoi(e) : 1; }();   // tuning: 0 = off, 2 = wherever it can run
    if (mode == 0) return false;
    if (mode == 2) return true;
    const double w = (double)n * (double)k / 1.0e6;
    const double per = x_dtype == TVL_F32 ? 0.045 + 0.05 * m : 0.045 + 0.0255 * m;
    // by launch geometry (launch_gemv_xy): 2 channels per wave on small layers, one-wave blocks of 8 channels, 4-wave blocks that split k
    const double geom = n < 16384 ? 1.4 : (k >= 8192 ? 0.92 : 1.3);
    const double t_gemv = 2.5 + w * per * geom;
    // the alternatives for one or two rows, fitted to tools/gemv_audit.sh: float32 = plane split + bf16 GEMM over 3K on 32-row tiles; bf16 = the
    // register-streaming kernel + reduce launch; f16 = the LDS-ring streaming kernel + reduce launch
    const double t_other = x_dtype == TVL_F32 ? 25.0 + 0.2 * w
                         : x_dtype == TVL_BF16 ? 9.7 + 0.09 * w
                         : (w <= 70.0 ? 8.5 + 0.085 * w : (6.2 + 0.071 * w > 14.3 ? 6.2 + 0.071 * w : 14.3));
    return t_gemv <= t_other;
}

template <typename TX, typename TY, int CH, int WAVES, bool KSPLIT>
static void launch_gemv_rows(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                             const float* bias, float clip, int act, int act_begin, TY* y, int64_t ld_y, hipStream_t st) {
    const dim3 grid((unsigned)ceil_div(n, KSPLIT ? CH : CH * WAVES)), block(64 * WAVES);
    if (m == 1)
        hipLaunchKernelGGL((gemv_kernel<TX, TY, CH, 1, WAVES, KSPLIT>), grid, block, 0, st, x, (int)m, (int)k, ld_x, pw, (int)n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y);
    else if (m == 2)
        hipLaunchKernelGGL((gemv_kernel<TX, TY, CH, 2, WAVES, KSPLIT>), grid, block, 0, st, x, (int)m, (int)k, ld_x, pw, (int)n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y);
    else
        hipLaunchKernelGGL((gemv_kernel<TX, TY, CH, 4, WAVES, KSPLIT>), grid, block, 0, st, x, (int)m, (int)k, ld_x, pw, (int)n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y);
}

template <typename TX, typename TY>
static int launch_gemv_xy(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                          const float* bias, float clip, int act, int act_begin, TY* y, int64_t ld_y, hipStream_t st) {
    // few channels: 2 per wave so that the layer still makes >= 2 waves per SIMD, and with k >= 4096 on layers of up to 6144 channels
    // the block's 4 waves split k over 2 or 4 shared channels (1 x 2048 x 8192: 5.0 -> 3.8 us, 2 x 2048 x 8192: 6.2 -> 4.7, 1 x 4096 x
    // 14336: 7.9 -> 7.1); many channels: 8 per wave (activations reused), and with a long k the waves split it as well
    if (n <= 6144 && k >= 4096) {
        if (n <= 2048 && m == 1) launch_gemv_rows<TX, TY, 2, 4, true>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, st);
        else launch_gemv_rows<TX, TY, 4, 4, true>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, st);
    } else if (n < 16384) launch_gemv_rows<TX, TY, 2, 4, false>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, st);
    else if (k >= 8192) launch_gemv_rows<TX, TY, 8, 4, true>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, st);
    else launch_gemv_rows<TX, TY, 8, 1, false>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, st);
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

template <typename TX>
static int launch_gemv_x(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                         const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y, hipStream_t st) {
    switch (y_dtype) {
        case TVL_F32: return launch_gemv_xy(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, (float*)y, ld_y, st);
        case TVL_F16: return launch_gemv_xy(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, (f16*)y, ld_y, st);
        case TVL_BF16: return launch_gemv_xy(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, (bf16_t*)y, ld_y, st);
        default: set_error("ternary_matmul: unsupported output dtype %d", y_dtype); return TVL_ERR_INVALID;
    }
}

int launch_gemv(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y, hipStream_t st) {
    if (x_dtype == TVL_F16) return launch_gemv_x((const f16*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, st);
    if (x_dtype == TVL_BF16) return launch_gemv_x((const bf16_t*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, st);
    return launch_gemv_x((const float*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, st);
}

}  // namespace tvl

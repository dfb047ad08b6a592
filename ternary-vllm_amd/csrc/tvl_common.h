// Shared helpers of libternary_hip.so: error reporting, dtype conversion, 2-bit code decode.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/tvl.h"

namespace tvl {

// ---- host-side error state (thread local) --------------------------------------------------
void set_error(const char* fmt, ...);

#define TVL_REQUIRE(cond, ...)                 \
    do {                                       \
        if (!(cond)) {                         \
            ::tvl::set_error(__VA_ARGS__);     \
            return TVL_ERR_INVALID;            \
        }                                      \
    } while (0)

#define TVL_HIP_CHECK(expr)                                                              \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess) {                                                         \
            ::tvl::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, __LINE__); \
            return TVL_ERR_HIP;                                                          \
        }                                                                                \
    } while (0)

// after a kernel launch
#define TVL_LAUNCH_CHECK() TVL_HIP_CHECK(hipGetLastError())

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline bool aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
static inline size_t dtype_size(int dt) {
    switch (dt) {
        case TVL_F32: return 4;
        case TVL_F16: return 2;
        case TVL_BF16: return 2;
        case TVL_I8: return 1;
        case TVL_I32: return 4;
        default: return 0;
    }
}

// ---- device-side scalar conversion ---------------------------------------------------------
using f16 = _Float16;
struct bf16_t { uint16_t v; };

__device__ __forceinline__ float to_float(float x) { return x; }
__device__ __forceinline__ float to_float(f16 x) { return (float)x; }
__device__ __forceinline__ float to_float(bf16_t x) { return __uint_as_float(((uint32_t)x.v) << 16); }

template <typename T> __device__ __forceinline__ T from_float(float x);
template <> __device__ __forceinline__ float from_float<float>(float x) { return x; }
template <> __device__ __forceinline__ f16 from_float<f16>(float x) { return (f16)x; }  // round to nearest even
template <> __device__ __forceinline__ bf16_t from_float<bf16_t>(float x) {
    // round to nearest even, NaN stays NaN (the plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
    __hip_bfloat16 h = __float2bfloat16(x);
    bf16_t r;
    r.v = *reinterpret_cast<uint16_t*>(&h);
    return r;
}
template <> __device__ __forceinline__ int8_t from_float<int8_t>(float x) { return (int8_t)x; }

// value of a 2-bit code: code - 1  (0 -> -1, 1 -> 0, 2 -> +1, 3 -> +2)
__device__ __forceinline__ int code_value(uint32_t byte, int i) { return (int)((byte >> (2 * i)) & 3u) - 1; }

// quantizers on a float value -> 2-bit code (value + 1)
template <int MODE> __device__ __forceinline__ uint32_t quantize_code(float w, float thr) {
    if constexpr (MODE == TVL_Q_TERNARY) {
        // (w >= thr) - (w <= -thr) + 1 ; NaN compares false twice -> code 1 (value 0)
        return 1u + (uint32_t)(w >= thr) - (uint32_t)(w <= -thr);
    } else if constexpr (MODE == TVL_Q_SIGN) {
        return (w >= 0.0f) ? 2u : 0u;  // +1 / -1
    } else {
        return (w >= 0.0f) ? 2u : 1u;  // +1 / 0
    }
}

// epilogue shared by every matmul kernel: two roundings (multiply, then add), clip, activation
// tanh-GELU, 0.5 x (1 + tanh(u)) with u = sqrt(2/pi) (x + 0.044715 x^3).  0.5 (1 + tanh(u)) = 1 - 1 / (exp(2u) + 1), so one hardware
// exponential and one hardware reciprocal (v_exp_f32, v_rcp_f32: 1 ulp each) replace the library tanhf (~3x the instructions; the
// epilogue of a 25216 x 3072 x 768 product spent a third of the launch there).  Absolute error <= |x| * 2e-7 against the exact form
// (tests: <= 1e-5 against a float64 evaluation); x = +-inf and NaN behave like the exact form (exp -> inf / 0, NaN propagates).
__device__ __forceinline__ float gelu_tanh_f32(float x) {
    // 2 u log2(e) = x (B + A x^2) with B = 2 log2(e) sqrt(2/pi), A = 0.044715 B: three instructions in front of the exponential
    const float B = 2.302208198144325f, A = 0.1029432395800235f;
    const float e = __builtin_amdgcn_exp2f(x * __builtin_fmaf(x * x, A, B));   // exp(2 u) = 2^(2 u log2(e))
    const float r = __builtin_amdgcn_rcpf(e + 1.0f);
    return x * (1.0f - r);   // (not x - x r: +inf must stay +inf)
}

// the first two steps of the epilogue alone (for callers that clip after their own rounding)
template <bool HAS_BIAS>
__device__ __forceinline__ float scale_bias(float acc, float scale, float bias) {
#pragma clang fp contract(off)
    float y = acc * scale;
    if constexpr (HAS_BIAS) y = y + bias;
    return y;
}

template <bool HAS_BIAS, int ACT>
__device__ __forceinline__ float epilogue(float acc, float scale, float bias, float clip) {
    // the reference does `x *= scale; x += bias` as two float32 operations: keep the compiler from
    // contracting them into one fma (the build uses -ffp-contract=fast elsewhere)
#pragma clang fp contract(off)
    float y = acc * scale;
    if constexpr (HAS_BIAS) y = y + bias;
    // IEEE 754-2019 minimum / maximum (v_minimum3_f32 / v_maximum3_f32 on gfx950) propagate NaN, like torch.clip:
    // two instructions where compare + select needs four
    y = __builtin_elementwise_maximum(__builtin_elementwise_minimum(y, clip), -clip);
    if constexpr (ACT == TVL_ACT_GELU_TANH) y = gelu_tanh_f32(y);
    return y;
}

// the activation only on channels >= act_begin: a layer whose leading channels are linear and whose trailing channels
// carry the activation (two Dense layers that share their input, run as one product)
template <bool HAS_BIAS, int ACT>
__device__ __forceinline__ float epilogue_ch(float acc, float scale, float bias, float clip, int ch, int act_begin) {
    float y = epilogue<HAS_BIAS, TVL_ACT_NONE>(acc, scale, bias, clip);
    if constexpr (ACT == TVL_ACT_GELU_TANH) {
        if (ch >= act_begin) y = gelu_tanh_f32(y);
    }
    return y;
}

// the same with the bias / activation choice made at run time (kernels where the epilogue is a vanishing share)
__device__ __forceinline__ float epilogue_rt(float acc, float scale, float bias, bool has_bias, int act, float clip, int ch = 0, int act_begin = 0) {
    const bool gelu = act == TVL_ACT_GELU_TANH && ch >= act_begin;
    if (has_bias) return gelu ? epilogue<true, TVL_ACT_GELU_TANH>(acc, scale, bias, clip) : epilogue<true, TVL_ACT_NONE>(acc, scale, bias, clip);
    return gelu ? epilogue<false, TVL_ACT_GELU_TANH>(acc, scale, bias, clip) : epilogue<false, TVL_ACT_NONE>(acc, scale, bias, clip);
}

}  // namespace tvl

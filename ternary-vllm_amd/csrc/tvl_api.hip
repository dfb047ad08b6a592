// C ABI of libternary_hip.so: argument validation and kernel selection.  See include/tvl.h.
#include "tvl_common.h"
#include "tvl_kernels.h"

#include <atomic>
#include <mutex>
#include <string>

namespace tvl {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

static inline bool float_dtype(int dt) { return dt == TVL_F32 || dt == TVL_F16 || dt == TVL_BF16; }

thread_local ResidualArgs* tls_residual = nullptr;

// y = (y + add1) + add2 over an [m, n] matrix, each sum rounded to y's type: four elements per thread where rows allow 16-byte /
// 8-byte accesses, else one
template <typename TY, int VEC>
__global__ void __launch_bounds__(256) add_rows_kernel(TY* __restrict__ y, int64_t m, int64_t n, int64_t ld_y, const TY* __restrict__ add1, int64_t ld1,
                                                       const TY* __restrict__ add2, int64_t ld2) {
    const int64_t per_row = n / VEC, total = m * per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / per_row, col = (i - r * per_row) * VEC;
        TY v[VEC], a[VEC], b[VEC];
        if constexpr (VEC == 4) {
            typedef uint32_t raw_t __attribute__((ext_vector_type(sizeof(TY))));   // 4 elements: 16 bytes (f32) or 8 bytes (16-bit types)
            *reinterpret_cast<raw_t*>(v) = *reinterpret_cast<const raw_t*>(y + r * ld_y + col);
            *reinterpret_cast<raw_t*>(a) = *reinterpret_cast<const raw_t*>(add1 + r * ld1 + col);
            if (add2) *reinterpret_cast<raw_t*>(b) = *reinterpret_cast<const raw_t*>(add2 + r * ld2 + col);
        } else {
            v[0] = y[r * ld_y + col]; a[0] = add1[r * ld1 + col];
            if (add2) b[0] = add2[r * ld2 + col];
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            v[e] = from_float<TY>(to_float(v[e]) + to_float(a[e]));
            if (add2) v[e] = from_float<TY>(to_float(v[e]) + to_float(b[e]));
        }
        if constexpr (VEC == 4) {
            typedef uint32_t raw_t __attribute__((ext_vector_type(sizeof(TY))));
            *reinterpret_cast<raw_t*>(y + r * ld_y + col) = *reinterpret_cast<const raw_t*>(v);
        } else {
            y[r * ld_y + col] = v[0];
        }
    }
}

template <typename TY>
static int launch_add_rows_t(TY* y, int64_t m, int64_t n, int64_t ld_y, const TY* add1, int64_t ld1, const TY* add2, int64_t ld2, hipStream_t st) {
    const size_t va = 4 * sizeof(TY);
    const bool vec = n % 4 == 0 && aligned(y, va) && aligned(add1, va) && (add2 == nullptr || aligned(add2, va)) && (ld_y * sizeof(TY)) % va == 0 &&
                     (ld1 * sizeof(TY)) % va == 0 && (ld2 * sizeof(TY)) % va == 0;
    int64_t g = ceil_div(vec ? m * (n / 4) : m * n, 256);
    g = g < 1 ? 1 : (g > 8192 ? 8192 : g);
    if (vec) hipLaunchKernelGGL((add_rows_kernel<TY, 4>), dim3((unsigned)g), dim3(256), 0, st, y, m, n, ld_y, add1, ld1, add2, ld2);
    else hipLaunchKernelGGL((add_rows_kernel<TY, 1>), dim3((unsigned)g), dim3(256), 0, st, y, m, n, ld_y, add1, ld1, add2, ld2);
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

int launch_add_rows(void* y, int y_dtype, int64_t m, int64_t n, int64_t ld_y, const void* add1, int64_t ld1, const void* add2, int64_t ld2, hipStream_t st) {
    switch (y_dtype) {
        case TVL_F32: return launch_add_rows_t((float*)y, m, n, ld_y, (const float*)add1, ld1, (const float*)add2, ld2, st);
        case TVL_F16: return launch_add_rows_t((f16*)y, m, n, ld_y, (const f16*)add1, ld1, (const f16*)add2, ld2, st);
        case TVL_BF16: return launch_add_rows_t((bf16_t*)y, m, n, ld_y, (const bf16_t*)add1, ld1, (const bf16_t*)add2, ld2, st);
        default: set_error("ternary_matmul_residual: unsupported output dtype %d", y_dtype); return TVL_ERR_INVALID;
    }
}

}  // namespace tvl

using namespace tvl;

extern "C" {

int tvl_version(void) { return TVL_VERSION; }

const char* tvl_last_error(void) { return g_last_error.c_str(); }

const char* tvl_build_info(void) {
    return "libternary_hip: gfx950 (CDNA4) HIP kernels; packed ternary GEMM on v_mfma_f32_16x16x32 / 32x32x16 {f16,bf16}, "
           "table-lookup GEMV, AND+popcount path";
}

int tvl_quantize_pack(const void* w, int w_dtype, int64_t n, int64_t k, int64_t ld_w, int mode, float threshold, uint8_t* pw,
                      int64_t ld_pw, void* stream) {
    TVL_REQUIRE(n >= 0 && k >= 0, "quantize_pack: negative shape n=%lld k=%lld", (long long)n, (long long)k);
    TVL_REQUIRE(float_dtype(w_dtype), "quantize_pack: weight dtype must be f32/f16/bf16, got %d", w_dtype);
    TVL_REQUIRE(mode == TVL_Q_TERNARY || mode == TVL_Q_SIGN || mode == TVL_Q_STEP, "quantize_pack: unknown quantizer %d", mode);
    if (ld_w == 0) ld_w = k;
    if (ld_pw == 0) ld_pw = ceil_div(k, 4);
    TVL_REQUIRE(ld_w >= k, "quantize_pack: ld_w=%lld < k=%lld", (long long)ld_w, (long long)k);
    TVL_REQUIRE(ld_pw >= ceil_div(k, 4), "quantize_pack: ld_pw=%lld < ceil(k/4)=%lld", (long long)ld_pw, (long long)ceil_div(k, 4));
    if (n == 0 || k == 0) return TVL_OK;
    TVL_REQUIRE(w && pw, "quantize_pack: NULL pointer");
    TVL_REQUIRE(aligned(w, dtype_size(w_dtype)), "quantize_pack: misaligned weight pointer");
    return launch_quantize_pack(w, w_dtype, n, k, ld_w, mode, threshold, pw, ld_pw, (hipStream_t)stream);
}

int tvl_quantize_pack_batch(const tvl_pack_item* items, int64_t count, int w_dtype, int mode, float threshold, void* stream) {
    TVL_REQUIRE(count >= 0, "quantize_pack_batch: negative count");
    TVL_REQUIRE(float_dtype(w_dtype), "quantize_pack_batch: weight dtype must be f32/f16/bf16, got %d", w_dtype);
    TVL_REQUIRE(mode == TVL_Q_TERNARY || mode == TVL_Q_SIGN || mode == TVL_Q_STEP, "quantize_pack_batch: unknown quantizer %d", mode);
    if (count == 0) return TVL_OK;
    TVL_REQUIRE(items != nullptr, "quantize_pack_batch: NULL item array");
    for (int64_t i = 0; i < count; ++i) {
        const tvl_pack_item& it = items[i];
        TVL_REQUIRE(it.n >= 0 && it.k >= 0, "quantize_pack_batch: item %lld has a negative shape", (long long)i);
        TVL_REQUIRE(it.ld_w == 0 || it.ld_w >= it.k, "quantize_pack_batch: item %lld: ld_w=%lld < k=%lld", (long long)i, (long long)it.ld_w, (long long)it.k);
        TVL_REQUIRE(it.ld_pw == 0 || it.ld_pw >= ceil_div(it.k, 4), "quantize_pack_batch: item %lld: ld_pw=%lld < ceil(k/4)=%lld", (long long)i,
                    (long long)it.ld_pw, (long long)ceil_div(it.k, 4));
        if (it.n == 0 || it.k == 0) continue;
        TVL_REQUIRE(it.w && it.pw, "quantize_pack_batch: item %lld has a NULL pointer", (long long)i);
        TVL_REQUIRE(aligned(it.w, dtype_size(w_dtype)), "quantize_pack_batch: item %lld: misaligned weight pointer", (long long)i);
    }
    return launch_quantize_pack_batch(items, count, w_dtype, mode, threshold, (hipStream_t)stream);
}

int tvl_pack_i8(const int8_t* t, int64_t n, int64_t k, int64_t ld_t, uint8_t* pw, int64_t ld_pw, void* stream) {
    TVL_REQUIRE(n >= 0 && k >= 0, "pack: negative shape");
    if (ld_t == 0) ld_t = k;
    if (ld_pw == 0) ld_pw = ceil_div(k, 4);
    TVL_REQUIRE(ld_t >= k && ld_pw >= ceil_div(k, 4), "pack: leading dimension too small");
    if (n == 0 || k == 0) return TVL_OK;
    TVL_REQUIRE(t && pw, "pack: NULL pointer");
    return launch_pack_i8(t, n, k, ld_t, pw, ld_pw, (hipStream_t)stream);
}

int tvl_unpack(const uint8_t* pw, int64_t n, int64_t k, int64_t ld_pw, void* out, int out_dtype, int64_t ld_out, void* stream) {
    TVL_REQUIRE(n >= 0 && k >= 0, "unpack: negative shape");
    TVL_REQUIRE(float_dtype(out_dtype) || out_dtype == TVL_I8, "unpack: output dtype must be f32/f16/bf16/i8, got %d", out_dtype);
    if (ld_pw == 0) ld_pw = ceil_div(k, 4);
    if (ld_out == 0) ld_out = k;
    TVL_REQUIRE(ld_pw >= ceil_div(k, 4), "unpack: orig_dim=%lld needs %lld packed bytes per row, rows have %lld", (long long)k,
                (long long)ceil_div(k, 4), (long long)ld_pw);
    TVL_REQUIRE(ld_out >= k, "unpack: ld_out too small");
    if (n == 0 || k == 0) return TVL_OK;
    TVL_REQUIRE(pw && out, "unpack: NULL pointer");
    TVL_REQUIRE(aligned(out, dtype_size(out_dtype)), "unpack: misaligned output pointer");
    return launch_unpack(pw, n, k, ld_pw, out, out_dtype, ld_out, (hipStream_t)stream);
}

int tvl_pack_planar_i8(const int8_t* t, int64_t len, int format, uint8_t* pw, void* stream) {
    TVL_REQUIRE(len >= 0, "pack_planar: negative length");
    TVL_REQUIRE(format == TVL_PLANAR_A || format == TVL_PLANAR_B, "pack_planar: unknown format %d", format);
    if (len == 0) return TVL_OK;
    TVL_REQUIRE(t && pw, "pack_planar: NULL pointer");
    return launch_pack_planar(t, len, format, pw, (hipStream_t)stream);
}

int tvl_unpack_planar(const uint8_t* pw, int64_t packed_len, int format, void* out, int out_dtype, void* stream) {
    TVL_REQUIRE(packed_len >= 0, "unpack_planar: negative length");
    TVL_REQUIRE(format == TVL_PLANAR_A || format == TVL_PLANAR_B, "unpack_planar: unknown format %d", format);
    TVL_REQUIRE(float_dtype(out_dtype) || out_dtype == TVL_I8, "unpack_planar: output dtype must be f32/f16/bf16/i8");
    if (packed_len == 0) return TVL_OK;
    TVL_REQUIRE(pw && out, "unpack_planar: NULL pointer");
    return launch_unpack_planar(pw, packed_len, format, out, out_dtype, (hipStream_t)stream);
}

int tvl_quantize(const void* w, int w_dtype, int64_t numel, int mode, float threshold, void* out, int out_dtype, void* stream) {
    TVL_REQUIRE(numel >= 0, "quantize: negative size");
    TVL_REQUIRE(float_dtype(w_dtype), "quantize: input dtype must be f32/f16/bf16");
    TVL_REQUIRE(mode == TVL_Q_TERNARY || mode == TVL_Q_SIGN || mode == TVL_Q_STEP, "quantize: unknown quantizer %d", mode);
    if (numel == 0) return TVL_OK;
    TVL_REQUIRE(w && out, "quantize: NULL pointer");
    return launch_quantize_dense(w, w_dtype, numel, mode, threshold, out, out_dtype, (hipStream_t)stream);
}

// Shapes the matrix-core kernels cannot take as they are (k not a multiple of 64, or rows they cannot load 16 bytes at a
// time): with enough workspace, x and the packed rows are first copied there zero-padded along k to the next multiple of 64
// (pad weights are code 1 = 0, pad activations 0, so the sums are unchanged) and the fast kernels run on the copies - two
// short HBM-bound launches instead of a vector-ALU product that is 20-50x slower on large shapes.
constexpr double PAD_ROUTE_MIN_WORK = 4.0e6;   // m * n * k below which the generic kernel's single launch is as fast

static inline size_t align256(size_t v) { return (v + 255) & ~size_t(255); }
static inline int64_t round_up64(int64_t k) { return (k + 63) / 64 * 64; }

static size_t aligned_workspace_bytes(int x_dtype, int64_t m, int64_t n, int64_t k) {   // k % 64 == 0
    if (x_dtype == TVL_F32) return gemm_mfma_workspace_bytes(TVL_F32, m, n, k);
    // split-K partial sums for shapes that underfill the chip (the largest of the kernels' wishes)
    const size_t a = gemm_mfma_workspace_bytes(TVL_F16, m, n, k);
    const size_t b = (m <= 32 && k % 256 == 0) ? gemm_skinny_workspace_bytes(m, n, k) : 0;
    const size_t c = (m <= 16 && k % 256 == 0) ? gemm_stream_workspace_bytes(m, n, k) : 0;
    const size_t d = (m <= 64 && k % 256 == 0) ? gemm_rows_shared_workspace_bytes(m, n, k) : 0;
    const size_t ab = a > b ? a : b, cd = c > d ? c : d;
    return ab > cd ? ab : cd;
}

static size_t pad_route_bytes(int x_dtype, int64_t m, int64_t n, int64_t k) {
    const int64_t kp = round_up64(k);
    return align256((size_t)m * kp * dtype_size(x_dtype)) + align256((size_t)n * (kp / 4)) + aligned_workspace_bytes(x_dtype, m, n, kp);
}

size_t tvl_ternary_matmul_workspace_bytes_for(int x_dtype, int64_t m, int64_t n, int64_t k) {
    if (m <= 0 || n <= 0 || k <= 0 || !float_dtype(x_dtype)) return 0;
    if (k % 64 != 0) return (double)m * (double)n * (double)k >= PAD_ROUTE_MIN_WORK ? pad_route_bytes(x_dtype, m, n, k) : 0;
    return aligned_workspace_bytes(x_dtype, m, n, k);
}

size_t tvl_ternary_matmul_workspace_bytes(int64_t m, int64_t n, int64_t k) { return tvl_ternary_matmul_workspace_bytes_for(TVL_F16, m, n, k); }

static int dispatch_matmul(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                           const float* scale, const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y,
                           void* workspace, size_t workspace_bytes, hipStream_t st, bool allow_pad) {
    if (gemm_rows_shared_supported(x, x_dtype, m, k, ld_x, pw, n, ld_pw, workspace, workspace_bytes))
        return launch_gemm_rows_shared(x, x_dtype, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, workspace, workspace_bytes, st);
    if (gemm_rows_supported(x, x_dtype, m, k, ld_x, pw, n, ld_pw, workspace, workspace_bytes))
        return launch_gemm_rows(x, x_dtype, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, workspace, workspace_bytes, st);
    if (gemv_supported(x, x_dtype, m, k, ld_x, pw, n, ld_pw))
        return launch_gemv(x, x_dtype, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, st);
    if (gemm_stream_supported(x, x_dtype, m, k, ld_x, pw, n, ld_pw, workspace, workspace_bytes))
        return launch_gemm_stream(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, workspace, workspace_bytes, st);
    if (gemm_skinny_supported(x, x_dtype, m, k, ld_x, pw, n, ld_pw))
        return launch_gemm_skinny(x, x_dtype, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, workspace, workspace_bytes, st);
    if (gemm_mfma_supported(x, x_dtype, m, k, ld_x, pw, n, ld_pw, workspace, workspace_bytes))
        return launch_gemm_mfma(x, x_dtype, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, workspace, workspace_bytes, st);
    if (allow_pad && k > 0 && (double)m * (double)n * (double)k >= PAD_ROUTE_MIN_WORK && workspace && aligned(workspace, 256) &&
        workspace_bytes >= pad_route_bytes(x_dtype, m, n, k)) {
        const int64_t kp = round_up64(k);
        unsigned char* base = static_cast<unsigned char*>(workspace);
        const size_t xb = align256((size_t)m * kp * dtype_size(x_dtype)), wb = align256((size_t)n * (kp / 4));
        void* xp = base;
        uint8_t* pwp = base + xb;
        int rc = launch_pad_rows(x, x_dtype, m, k, ld_x, xp, kp, st);
        if (rc != TVL_OK) return rc;
        rc = launch_pad_packed(pw, n, k, ld_pw, pwp, kp / 4, st);
        if (rc != TVL_OK) return rc;
        return dispatch_matmul(xp, x_dtype, m, kp, kp, pwp, n, kp / 4, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, base + xb + wb,
                               workspace_bytes - xb - wb, st, false);
    }
    return launch_gemm_generic(x, x_dtype, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, st);
}

int tvl_ternary_matmul(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                       const float* scale, const float* bias, float clip, int act, void* y, int y_dtype, int64_t ld_y,
                       void* workspace, size_t workspace_bytes, void* stream) {
    return tvl_ternary_matmul_act_from(x, x_dtype, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, 0, y, y_dtype, ld_y, workspace,
                                       workspace_bytes, stream);
}

int tvl_ternary_matmul_act_from(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n,
                                int64_t ld_pw, const float* scale, const float* bias, float clip, int act, int64_t act_begin_channel,
                                void* y, int y_dtype, int64_t ld_y, void* workspace, size_t workspace_bytes, void* stream) {
    TVL_REQUIRE(act_begin_channel >= 0 && act_begin_channel <= (n > 0 ? n : 0), "ternary_matmul: act_begin_channel=%lld outside [0, n=%lld]",
                (long long)act_begin_channel, (long long)n);
    TVL_REQUIRE(n <= INT32_MAX, "ternary_matmul: n=%lld too large", (long long)n);
    if (act_begin_channel >= n) act = act == TVL_ACT_GELU_TANH ? TVL_ACT_NONE : act;   // no channel carries the activation
    const int act_begin = act == TVL_ACT_NONE ? 0 : (int)act_begin_channel;
    TVL_REQUIRE(m >= 0 && n >= 0 && k >= 0, "ternary_matmul: negative shape m=%lld n=%lld k=%lld", (long long)m, (long long)n, (long long)k);
    TVL_REQUIRE(float_dtype(x_dtype), "ternary_matmul: activation dtype must be f32/f16/bf16, got %d", x_dtype);
    TVL_REQUIRE(float_dtype(y_dtype), "ternary_matmul: output dtype must be f32/f16/bf16, got %d", y_dtype);
    TVL_REQUIRE(act == TVL_ACT_NONE || act == TVL_ACT_GELU_TANH, "ternary_matmul: unknown activation %d", act);
    if (ld_x == 0) ld_x = k;
    if (ld_pw == 0) ld_pw = ceil_div(k, 4);
    if (ld_y == 0) ld_y = n;
    TVL_REQUIRE(ld_x >= k, "ternary_matmul: ld_x=%lld < k=%lld", (long long)ld_x, (long long)k);
    TVL_REQUIRE(ld_pw >= ceil_div(k, 4),
                "ternary_matmul: packed rows hold %lld bytes but k=%lld needs %lld (x and pw disagree on the input dimension)",
                (long long)ld_pw, (long long)k, (long long)ceil_div(k, 4));
    TVL_REQUIRE(ld_y >= n, "ternary_matmul: ld_y=%lld < n=%lld", (long long)ld_y, (long long)n);
    TVL_REQUIRE(!(clip != clip), "ternary_matmul: clip is NaN");
    if (m == 0 || n == 0) return TVL_OK;
    TVL_REQUIRE(y && scale, "ternary_matmul: NULL output or scale pointer");
    TVL_REQUIRE(k == 0 || (x && pw), "ternary_matmul: NULL input pointer");
    TVL_REQUIRE(aligned(x, dtype_size(x_dtype)) && aligned(y, dtype_size(y_dtype)) && aligned(scale, 4) && aligned(bias, 4),
                "ternary_matmul: misaligned pointer");
    return dispatch_matmul(x, x_dtype, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, workspace, workspace_bytes,
                           (hipStream_t)stream, true);
}

int tvl_ternary_matmul_residual(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                                const float* scale, const float* bias, float clip, int act, int64_t act_begin_channel, const void* add1,
                                int64_t ld_add1, const void* add2, int64_t ld_add2, void* y, int y_dtype, int64_t ld_y, void* workspace,
                                size_t workspace_bytes, void* stream) {
    if (add1 == nullptr && add2 != nullptr) { add1 = add2; ld_add1 = ld_add2; add2 = nullptr; ld_add2 = 0; }
    if (add1 == nullptr)
        return tvl_ternary_matmul_act_from(x, x_dtype, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin_channel, y, y_dtype, ld_y, workspace,
                                           workspace_bytes, stream);
    TVL_REQUIRE(n >= 0 && m >= 0, "ternary_matmul_residual: negative shape");
    if (ld_add1 == 0) ld_add1 = n;
    if (ld_add2 == 0) ld_add2 = n;
    if (ld_y == 0) ld_y = n;
    TVL_REQUIRE(ld_add1 >= n && ld_add2 >= n, "ternary_matmul_residual: addend row stride %lld / %lld < n=%lld", (long long)ld_add1, (long long)ld_add2, (long long)n);
    TVL_REQUIRE(float_dtype(y_dtype), "ternary_matmul_residual: output dtype must be f32/f16/bf16, got %d", y_dtype);
    TVL_REQUIRE(aligned(add1, dtype_size(y_dtype)) && aligned(add2, dtype_size(y_dtype)), "ternary_matmul_residual: misaligned addend pointer");
    // the product is written to y before the addends are read (unless the split-K reduction takes them): y must not be an addend
    TVL_REQUIRE(add1 != y && add2 != y, "ternary_matmul_residual: an addend is the output buffer itself (the product is written before the addends are read)");
    ResidualArgs res{add1, ld_add1, add2, ld_add2, y, m, false};
    tls_residual = &res;
    const int rc = tvl_ternary_matmul_act_from(x, x_dtype, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin_channel, y, y_dtype, ld_y, workspace,
                                               workspace_bytes, stream);
    tls_residual = nullptr;
    if (rc != TVL_OK || res.consumed || m == 0 || n == 0) return rc;
    return launch_add_rows(y, y_dtype, m, n, ld_y, add1, ld_add1, add2, ld_add2, (hipStream_t)stream);
}

int tvl_ternary_matmul_route(int x_dtype, int64_t m, int64_t n, int64_t k, size_t workspace_bytes, char* out, size_t out_len) {
    TVL_REQUIRE(out != nullptr && out_len > 0, "ternary_matmul_route: no output buffer");
    TVL_REQUIRE(float_dtype(x_dtype) && m > 0 && n > 0 && k > 0, "ternary_matmul_route: bad dtype or shape");
    // dense, well-aligned operands: pointers that pass every alignment test, never dereferenced
    const void* x = reinterpret_cast<const void*>(uintptr_t(1) << 20);
    const uint8_t* pw = reinterpret_cast<const uint8_t*>(uintptr_t(2) << 20);
    const void* ws = workspace_bytes ? reinterpret_cast<const void*>(uintptr_t(3) << 20) : nullptr;
    const int64_t ld_x = k, ld_pw = ceil_div(k, 4);
    const char* name = nullptr;
    if (gemm_rows_shared_supported(x, x_dtype, m, k, ld_x, pw, n, ld_pw, ws, workspace_bytes)) name = "gemm_rows_shared_kernel (+ splitk_reduce_kernel when k is split)";
    else if (gemm_rows_supported(x, x_dtype, m, k, ld_x, pw, n, ld_pw, ws, workspace_bytes)) name = "gemm_rows_staged_kernel / gemm_rows_kernel (one launch)";
    else if (gemv_supported(x, x_dtype, m, k, ld_x, pw, n, ld_pw)) name = "gemv_kernel";
    else if (gemm_stream_supported(x, x_dtype, m, k, ld_x, pw, n, ld_pw, ws, workspace_bytes)) name = "gemm_stream_kernel + splitk_reduce_kernel";
    else if (gemm_skinny_supported(x, x_dtype, m, k, ld_x, pw, n, ld_pw)) name = "gemm_skinny_kernel + splitk_reduce_kernel";
    if (name != nullptr) { snprintf(out, out_len, "%s", name); return TVL_OK; }
    if (gemm_mfma_supported(x, x_dtype, m, k, ld_x, pw, n, ld_pw, ws, workspace_bytes)) {
        gemm_mfma_describe(x_dtype, m, n, k, workspace_bytes, out, out_len);
        return TVL_OK;
    }
    if ((double)m * (double)n * (double)k >= PAD_ROUTE_MIN_WORK && ws && workspace_bytes >= pad_route_bytes(x_dtype, m, n, k)) {
        const int w = snprintf(out, out_len, "pad_rows_kernel + pad_packed_kernel + ");
        if (w > 0 && (size_t)w < out_len) {
            const int64_t kp = round_up64(k);
            const size_t used = align256((size_t)m * kp * dtype_size(x_dtype)) + align256((size_t)n * (kp / 4));
            return tvl_ternary_matmul_route(x_dtype, m, n, kp, workspace_bytes - used, out + w, out_len - w);
        }
        return TVL_OK;
    }
    snprintf(out, out_len, "gemm_generic_kernel");
    return TVL_OK;
}

}  // extern "C"

// ---- the matmul for callers that cannot allocate ------------------------------------------------------------------
// Scratch memory of tvl_ternary_matmul_noalloc: a file-scope device array, one copy per GPU, held from the moment the
// code object is loaded on that GPU until the process ends.  In host code the array's name is only a handle for
// hipGetSymbolAddress: the address the kernels get is looked up once per device and cached here.
static __device__ __align__(256) unsigned char scratch[TVL_NOALLOC_SCRATCH_BYTES];
constexpr int MAX_DEVICES = 64;
static std::atomic<void*> g_scratch_ptr[MAX_DEVICES];
static std::mutex g_scratch_mutex;

// device address of `scratch` on the CURRENT device (first call on a device: one symbol lookup, no allocation)
static int current_device_scratch(void** out) {
    int dev = -1;
    TVL_HIP_CHECK(hipGetDevice(&dev));
    TVL_REQUIRE(dev >= 0 && dev < MAX_DEVICES, "ternary_matmul_noalloc: device index %d outside [0, %d)", dev, MAX_DEVICES);
    void* p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (p == nullptr) {
        std::lock_guard<std::mutex> lock(g_scratch_mutex);
        p = g_scratch_ptr[dev].load(std::memory_order_relaxed);
        if (p == nullptr) {
            TVL_HIP_CHECK(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
            TVL_REQUIRE(p != nullptr && aligned(p, 256), "ternary_matmul_noalloc: bad scratch address %p on device %d", p, dev);
            g_scratch_ptr[dev].store(p, std::memory_order_release);
        }
    }
    *out = p;
    return TVL_OK;
}

extern "C" {

size_t tvl_noalloc_scratch_bytes(void) { return TVL_NOALLOC_SCRATCH_BYTES; }

int tvl_ternary_matmul_noalloc(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                               const float* scale, const float* bias, float clip, int act, int64_t act_begin_channel, void* y, int y_dtype,
                               int64_t ld_y, void* stream) {
    const size_t need = tvl_ternary_matmul_workspace_bytes_for(x_dtype, m, n, k);
    TVL_REQUIRE(need <= TVL_NOALLOC_SCRATCH_BYTES,
                "ternary_matmul_noalloc: m=%lld n=%lld k=%lld (dtype %d) needs %zu bytes of scratch memory, the built-in buffer holds %zu: "
                "use tvl_ternary_matmul with a caller-owned workspace, or split the rows over several calls",
                (long long)m, (long long)n, (long long)k, x_dtype, need, (size_t)TVL_NOALLOC_SCRATCH_BYTES);
    // Looked up on every call, whatever the size (one atomic load after the first): a call with m = 0 before a stream capture starts is
    // enough to keep the one-time hipGetSymbolAddress (a runtime call that is not a stream operation) out of the capture.
    void* ws = nullptr;
    if (const int rc = current_device_scratch(&ws); rc != TVL_OK) return rc;
    // `need` bytes, not the whole array: the planners then see exactly what tvl_ternary_matmul_act_from sees with a workspace of
    // tvl_ternary_matmul_workspace_bytes_for() bytes, so both entry points choose the same kernels and k splits
    return tvl_ternary_matmul_act_from(x, x_dtype, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin_channel, y, y_dtype, ld_y,
                                       need > 0 ? ws : nullptr, need, stream);
}

int tvl_sign_pack_bits(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, uint32_t* xb, int64_t ld_xb, void* stream) {
    TVL_REQUIRE(m >= 0 && k >= 0, "sign_pack_bits: negative shape");
    TVL_REQUIRE(float_dtype(x_dtype), "sign_pack_bits: dtype must be f32/f16/bf16, got %d", x_dtype);
    if (ld_x == 0) ld_x = k;
    if (ld_xb == 0) ld_xb = ceil_div(k, 32);
    TVL_REQUIRE(ld_x >= k && ld_xb >= ceil_div(k, 32), "sign_pack_bits: row stride too small");
    if (m == 0 || k == 0) return TVL_OK;
    TVL_REQUIRE(x && xb, "sign_pack_bits: NULL pointer");
    TVL_REQUIRE(aligned(x, dtype_size(x_dtype)) && aligned(xb, 4), "sign_pack_bits: misaligned pointer");
    return launch_sign_bits(x, x_dtype, m, k, ld_x, xb, ld_xb, (hipStream_t)stream);
}

int tvl_ternary_bitplanes(const uint8_t* pw, int64_t n, int64_t k, int64_t ld_pw, uint32_t* pos, uint32_t* neg, int64_t ld_planes,
                          void* stream) {
    TVL_REQUIRE(n >= 0 && k >= 0, "ternary_bitplanes: negative shape");
    if (ld_pw == 0) ld_pw = ceil_div(k, 4);
    if (ld_planes == 0) ld_planes = ceil_div(k, 32);
    TVL_REQUIRE(ld_pw >= ceil_div(k, 4) && ld_planes >= ceil_div(k, 32), "ternary_bitplanes: row stride too small");
    if (n == 0 || k == 0) return TVL_OK;
    TVL_REQUIRE(pw && pos && neg, "ternary_bitplanes: NULL pointer");
    TVL_REQUIRE(aligned(pos, 4) && aligned(neg, 4), "ternary_bitplanes: misaligned pointer");
    return launch_bitplanes(pw, n, k, ld_pw, pos, neg, ld_planes, (hipStream_t)stream);
}

int tvl_popcount_matmul(const uint32_t* xb, int64_t m, int64_t ld_xb, const uint32_t* pos, const uint32_t* neg, int64_t n, int64_t ld_planes,
                        int64_t k, const float* scale, const float* bias, float clip, int act, void* y, int y_dtype, int64_t ld_y, void* stream) {
    TVL_REQUIRE(m >= 0 && n >= 0 && k >= 0, "popcount_matmul: negative shape");
    TVL_REQUIRE(float_dtype(y_dtype), "popcount_matmul: output dtype must be f32/f16/bf16, got %d", y_dtype);
    TVL_REQUIRE(act == TVL_ACT_NONE || act == TVL_ACT_GELU_TANH, "popcount_matmul: unknown activation %d", act);
    TVL_REQUIRE(k <= (int64_t)1 << 30, "popcount_matmul: k too large for 32-bit counts");
    if (ld_xb == 0) ld_xb = ceil_div(k, 32);
    if (ld_planes == 0) ld_planes = ceil_div(k, 32);
    if (ld_y == 0) ld_y = n;
    TVL_REQUIRE(ld_xb >= ceil_div(k, 32) && ld_planes >= ceil_div(k, 32) && ld_y >= n, "popcount_matmul: row stride too small");
    TVL_REQUIRE(!(clip != clip), "popcount_matmul: clip is NaN");
    if (m == 0 || n == 0) return TVL_OK;
    TVL_REQUIRE(y && scale, "popcount_matmul: NULL output or scale pointer");
    TVL_REQUIRE(k == 0 || (xb && pos && neg), "popcount_matmul: NULL input pointer");
    TVL_REQUIRE(aligned(xb, 4) && aligned(pos, 4) && aligned(neg, 4) && aligned(y, dtype_size(y_dtype)) && aligned(scale, 4) && aligned(bias, 4),
                "popcount_matmul: misaligned pointer");
    return launch_popcount_matmul(xb, m, ld_xb, pos, neg, n, ld_planes, k, scale, bias, clip, act, 0, y, y_dtype, ld_y, (hipStream_t)stream);
}

int tvl_ternary_mm_f32(const float* x, int64_t m, int64_t k, int64_t ld_x, const float* tm, int64_t n, int64_t ld_tm, float* res,
                       int64_t ld_res, void* stream) {
    TVL_REQUIRE(m >= 0 && n >= 0 && k >= 0, "ternary_mm: negative shape");
    if (ld_x == 0) ld_x = k;
    if (ld_tm == 0) ld_tm = k;
    if (ld_res == 0) ld_res = n;
    TVL_REQUIRE(ld_x >= k && ld_tm >= k && ld_res >= n, "ternary_mm: leading dimension too small");
    if (m == 0 || n == 0) return TVL_OK;
    TVL_REQUIRE(res && (k == 0 || (x && tm)), "ternary_mm: NULL pointer");
    return launch_ternary_mm(x, m, k, ld_x, tm, n, ld_tm, res, ld_res, (hipStream_t)stream);
}

size_t tvl_fit_scale_workspace_bytes(int64_t n, int64_t k) {
    (void)n; (void)k;
    return fit_workspace_bytes();
}

int tvl_fit_ternary_scale_f32(const float* w, int64_t n, int64_t k, int64_t ld_w, int uniform, int iters, float* scale, float* latent,
                              int64_t ld_latent, void* workspace, size_t workspace_bytes, void* stream) {
    TVL_REQUIRE(n > 0 && k > 0, "fit_ternary_scale: empty matrix");
    TVL_REQUIRE(iters >= 0, "fit_ternary_scale: negative iteration count");
    if (ld_w == 0) ld_w = k;
    if (ld_latent == 0) ld_latent = k;
    TVL_REQUIRE(ld_w >= k && ld_latent >= k, "fit_ternary_scale: leading dimension too small");
    TVL_REQUIRE(w && scale, "fit_ternary_scale: NULL pointer");
    TVL_REQUIRE(n <= INT32_MAX, "fit_ternary_scale: too many rows");
    if (uniform) {
        TVL_REQUIRE(workspace && workspace_bytes >= fit_workspace_bytes() && aligned(workspace, 8),
                    "fit_ternary_scale: the whole-matrix fit needs %zu bytes of 8-byte aligned workspace", fit_workspace_bytes());
    }
    return launch_fit_ternary(w, n, k, ld_w, uniform, iters, scale, latent, ld_latent, workspace, (hipStream_t)stream);
}

int tvl_fit_binary_scale_f32(const float* w, int64_t n, int64_t k, int64_t ld_w, int mode, int uniform, float prescale, int natural,
                             float* scale, float* latent, int64_t ld_latent, void* workspace, size_t workspace_bytes, void* stream) {
    TVL_REQUIRE(n > 0 && k > 0, "fit_binary_scale: empty matrix");
    TVL_REQUIRE(mode == TVL_Q_SIGN || mode == TVL_Q_STEP, "fit_binary_scale: mode must be TVL_Q_SIGN or TVL_Q_STEP");
    if (ld_w == 0) ld_w = k;
    if (ld_latent == 0) ld_latent = k;
    TVL_REQUIRE(ld_w >= k && ld_latent >= k, "fit_binary_scale: leading dimension too small");
    TVL_REQUIRE(w && scale, "fit_binary_scale: NULL pointer");
    TVL_REQUIRE(n <= INT32_MAX, "fit_binary_scale: too many rows");
    if (uniform) {
        TVL_REQUIRE(workspace && workspace_bytes >= fit_workspace_bytes() && aligned(workspace, 8),
                    "fit_binary_scale: the whole-matrix fit needs %zu bytes of 8-byte aligned workspace", fit_workspace_bytes());
    }
    return launch_fit_binary(w, n, k, ld_w, mode, uniform, prescale, natural, scale, latent, ld_latent, workspace, (hipStream_t)stream);
}

int tvl_zero_out_i32(const int32_t* in, int64_t numel, int32_t* out, void* stream) {
    TVL_REQUIRE(numel >= 0, "zero_out: negative size");
    if (numel == 0) return TVL_OK;
    TVL_REQUIRE(in && out, "zero_out: NULL pointer");
    return launch_zero_out(in, numel, out, (hipStream_t)stream);
}

int tvl_prevent_folding_u8(const uint8_t* constant, int64_t numel, const void* witness, uint8_t* out, void* stream) {
    (void)witness;
    TVL_REQUIRE(numel >= 0, "prevent_folding: negative size");
    if (numel == 0) return TVL_OK;
    TVL_REQUIRE(constant && out, "prevent_folding: NULL pointer");
    return launch_copy_u8(constant, numel, out, (hipStream_t)stream);
}

}  // extern "C"

// Mid-sized products (about 100 .. 2000 activation rows): 128-row workgroup tiles of the 16x16x32 matrix-core kernel.
//
// Same device code as the 256 x 256 tile (gemm_mfma_kernels.h: LDS-DMA staged activations two steps ahead, packed weights
// straight to registers, scalar base + lane offset addressing, 16-byte non-temporal stores); a workgroup here is 4 or 8 waves on
// a 128 x 128 or 128 x 256 tile, every wave all 128 rows x 32 channels, so a decoded weight fragment feeds 8 row tiles (the older
// 32x32x16 small-tile kernel: 2) and 64 KiB of LDS lets two workgroups share a compute unit.  With split-K the partial sums go
// to float32 slabs and splitk_reduce_kernel (gemm_mfma.hip) finishes.
#include "gemm_mfma_kernels.h"

#include <stdlib.h>

namespace tvl {

constexpr int MID_NB = 4;   // activation tiles in LDS: a tile is requested two steps before its use

// The 128 x 256 tile in its low-register form (four row parts per MFMA group: <= 128 registers), so that TWO workgroups share a
// compute unit (4 waves per SIMD, 2 x 64 KiB of LDS): their prologues, k loops and epilogues overlap each other.
template <typename TX, typename TY>
static int launch_mid_dual(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                           const float* bias, float clip, int act, int act_begin, TY* y, int64_t ld_y, int w_steps, hipStream_t st) {
    constexpr int BM = 128, BN = 256;
    const int tiles_m = (int)ceil_div(m, BM), tiles_n = (int)ceil_div(n, BN);
    const int nblk = tiles_m * tiles_n;
    const size_t lds_bytes = (size_t)MID_NB * BM * ROW_BYTES;
    int rc;
#define TVL_DUAL(HB, ACT)                                                                                                   \
    do {                                                                                                                    \
        auto kern = gemm_mfma16_kernel<TX, TY, HB, ACT, 1, false, MID_NB, BM, 8, false, 4, 32, 4>;                          \
        if ((rc = set_dynamic_lds_limit((const void*)kern, lds_bytes)) != TVL_OK) return rc;                                \
        hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), lds_bytes, st, x, (int)m, (int)k, ld_x, pw, (int)n, ld_pw, scale, bias, clip, \
                           act_begin, y, ld_y, tiles_n, nblk, w_steps, 0);                                                  \
    } while (0)
    if (bias) {
        if (act == TVL_ACT_GELU_TANH) TVL_DUAL(true, TVL_ACT_GELU_TANH); else TVL_DUAL(true, TVL_ACT_NONE);
    } else {
        if (act == TVL_ACT_GELU_TANH) TVL_DUAL(false, TVL_ACT_GELU_TANH); else TVL_DUAL(false, TVL_ACT_NONE);
    }
#undef TVL_DUAL
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

template <typename TX, typename TY, int BM, int WAVES>
static int launch_mid_cfg(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                          const float* bias, float clip, int act, int act_begin, TY* y, int64_t ld_y, int splits, float* slab, int w_steps,
                          hipStream_t st) {
    constexpr int BN = 32 * WAVES;
    const int tiles_m = (int)ceil_div(m, BM), tiles_n = (int)ceil_div(n, BN);
    const int nblk = tiles_m * tiles_n;
    const size_t lds_bytes = (size_t)MID_NB * BM * ROW_BYTES;
    const dim3 block(64 * WAVES);
    int rc;
    if (splits > 1) {
        const int total_steps = (int)(k / BK);
        const int steps_per_split = (int)ceil_div(total_steps, splits);
        splits = (int)ceil_div(total_steps, steps_per_split);  // no empty split
        auto kern = gemm_mfma16_kernel<TX, float, false, TVL_ACT_NONE, 1, false, MID_NB, BM, WAVES, true, (WAVES == 4 ? 3 : 2)>;
        if ((rc = set_dynamic_lds_limit((const void*)kern, lds_bytes)) != TVL_OK) return rc;
        hipLaunchKernelGGL(kern, dim3(nblk, splits), block, lds_bytes, st, x, (int)m, (int)k, ld_x, pw, (int)n, ld_pw, scale, bias, clip, 0,
                           slab, n, tiles_n, nblk, w_steps, steps_per_split);
        TVL_LAUNCH_CHECK();
        return launch_splitk_reduce(slab, splits, m, n, scale, bias, clip, act, act_begin, y,
                                    std::is_same<TY, float>::value ? TVL_F32 : (std::is_same<TY, f16>::value ? TVL_F16 : TVL_BF16), ld_y, st);
    }
    // Many more tiles than compute units (tall activations, short k: the ViT shapes): one workgroup per CU walks its tiles as one
    // stream of k steps, the next tile's loads running under the current tile's epilogue (TVL_MID_PERSIST=0/1 pins it: tuning).
    static const int force_persist = [] { const char* e = getenv("TVL_MID_PERSIST"); return e ? atoi(e) : -1; }();
    const int cus = device_cu_count();
    const bool persist = WAVES == 8 && nblk > cus && force_persist != 0;
#define TVL_MID(HB, ACT)                                                                                                    \
    do {                                                                                                                    \
        if constexpr (WAVES == 8) {                                                                                         \
            if (persist) {                                                                                                  \
                auto kp = gemm_mfma16_kernel<TX, TY, HB, ACT, 1, true, MID_NB, BM, WAVES, false, 2>;                        \
                const size_t lds_p = lds_bytes + persist_param_lds_bytes(WAVES);   /* + the staged scale / bias */           \
                if ((rc = set_dynamic_lds_limit((const void*)kp, lds_p)) != TVL_OK) return rc;                              \
                hipLaunchKernelGGL(kp, dim3(cus), block, lds_p, st, x, (int)m, (int)k, ld_x, pw, (int)n, ld_pw, scale, bias, clip, \
                                   act_begin, y, ld_y, tiles_n, nblk, w_steps, 0);                                          \
                break;                                                                                                      \
            }                                                                                                               \
        }                                                                                                                   \
        auto kern = gemm_mfma16_kernel<TX, TY, HB, ACT, 1, false, MID_NB, BM, WAVES, false, (WAVES == 4 ? 3 : 2)>;          \
        if ((rc = set_dynamic_lds_limit((const void*)kern, lds_bytes)) != TVL_OK) return rc;                                \
        hipLaunchKernelGGL(kern, dim3(nblk), block, lds_bytes, st, x, (int)m, (int)k, ld_x, pw, (int)n, ld_pw, scale, bias, clip, \
                           act_begin, y, ld_y, tiles_n, nblk, w_steps, 0);                                                  \
    } while (0)
    if (bias) {
        if (act == TVL_ACT_GELU_TANH) TVL_MID(true, TVL_ACT_GELU_TANH); else TVL_MID(true, TVL_ACT_NONE);
    } else {
        if (act == TVL_ACT_GELU_TANH) TVL_MID(false, TVL_ACT_GELU_TANH); else TVL_MID(false, TVL_ACT_NONE);
    }
#undef TVL_MID
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

template <typename TX, typename TY>
static int launch_mid_xy(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                         const float* bias, float clip, int act, int act_begin, TY* y, int64_t ld_y, int bm, int bn, int splits, float* slab,
                         int w_steps, hipStream_t st) {
    if (bm == 128 && bn == 256 && splits < 0)   // (splits = -1: the two-workgroups-per-CU form)
        return launch_mid_dual<TX, TY>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, w_steps, st);
    if (bm == 128 && bn == 256)
        return launch_mid_cfg<TX, TY, 128, 8>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, splits, slab, w_steps, st);
    if (bm == 128 && bn == 128)
        return launch_mid_cfg<TX, TY, 128, 4>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, splits, slab, w_steps, st);
    if (bm == 64 && bn == 128)
        return launch_mid_cfg<TX, TY, 64, 4>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, splits, slab, w_steps, st);
    set_error("ternary_matmul: no %d x %d tile in the mid-size kernel", bm, bn);
    return TVL_ERR_INVALID;
}

template <typename TX>
static int launch_mid_x(const TX* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                        const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y, int bm, int bn, int splits,
                        float* slab, int w_steps, hipStream_t st) {
    switch (y_dtype) {
        case TVL_F32: return launch_mid_xy(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, (float*)y, ld_y, bm, bn, splits, slab, w_steps, st);
        case TVL_F16: return launch_mid_xy(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, (f16*)y, ld_y, bm, bn, splits, slab, w_steps, st);
        case TVL_BF16: return launch_mid_xy(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, (bf16_t*)y, ld_y, bm, bn, splits, slab, w_steps, st);
        default: set_error("ternary_matmul: unsupported output dtype %d", y_dtype); return TVL_ERR_INVALID;
    }
}

int launch_gemm_mid(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                    const float* scale, const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y,
                    int bm, int bn, int splits, float* slab, int w_steps, hipStream_t st) {
    if (x_dtype == TVL_F16)
        return launch_mid_x((const f16*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, bm, bn, splits, slab, w_steps, st);
    if (x_dtype == TVL_BF16)
        return launch_mid_x((const bf16_t*)x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, y_dtype, ld_y, bm, bn, splits, slab, w_steps, st);
    set_error("ternary_matmul: the mid-size kernel takes f16 / bf16 activations, got %d", x_dtype);
    return TVL_ERR_INVALID;
}

}  // namespace tvl

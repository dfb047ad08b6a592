// Internal launcher interface between the C ABI (tvl_api.hip) and the kernel files.
// Arguments were validated by the caller; every launcher returns a TVL_* status.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/tvl.h"

namespace tvl {

// pack_kernels.hip
int launch_quantize_pack(const void* w, int w_dtype, int64_t n, int64_t k, int64_t ld_w, int mode, float thr, uint8_t* pw,
                         int64_t ld_pw, hipStream_t st);
int launch_quantize_pack_batch(const tvl_pack_item* items, int64_t count, int w_dtype, int mode, float thr, hipStream_t st);
int launch_pack_i8(const int8_t* t, int64_t n, int64_t k, int64_t ld_t, uint8_t* pw, int64_t ld_pw, hipStream_t st);
int launch_unpack(const uint8_t* pw, int64_t n, int64_t k, int64_t ld_pw, void* out, int out_dtype, int64_t ld_out, hipStream_t st);
int launch_pack_planar(const int8_t* t, int64_t len, int format, uint8_t* pw, hipStream_t st);
int launch_unpack_planar(const uint8_t* pw, int64_t s, int format, void* out, int out_dtype, hipStream_t st);
int launch_quantize_dense(const void* w, int w_dtype, int64_t numel, int mode, float thr, void* out, int out_dtype, hipStream_t st);
int launch_zero_out(const int32_t* in, int64_t numel, int32_t* out, hipStream_t st);
int launch_copy_u8(const uint8_t* in, int64_t numel, uint8_t* out, hipStream_t st);
// zero-padded copies along k: x rows to kp elements (kp % 8 == 0), packed rows to bp bytes (bp % 4 == 0) with pad codes 1 (weight 0)
int launch_pad_rows(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, void* out, int64_t kp, hipStream_t st);
int launch_pad_packed(const uint8_t* pw, int64_t n, int64_t k, int64_t ld_pw, uint8_t* out, int64_t bp, hipStream_t st);

// gemm_generic.hip
int launch_gemm_generic(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n,
                        int64_t ld_pw, const float* scale, const float* bias, float clip, int act, int act_begin, void* y, int y_dtype,
                        int64_t ld_y, hipStream_t st);
int launch_ternary_mm(const float* x, int64_t m, int64_t k, int64_t ld_x, const float* tm, int64_t n, int64_t ld_tm, float* res,
                      int64_t ld_res, hipStream_t st);

// gemm_mfma.hip
bool gemm_mfma_supported(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n,
                         int64_t ld_pw, const void* workspace, size_t workspace_bytes);
int launch_gemm_mfma(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n,
                     int64_t ld_pw, const float* scale, const float* bias, float clip, int act, int act_begin, void* y, int y_dtype,
                     int64_t ld_y, void* workspace, size_t workspace_bytes, hipStream_t st);
size_t gemm_mfma_workspace_bytes(int x_dtype, int64_t m, int64_t n, int64_t k);
int launch_split_f32_rows(const float* x, int64_t m, int64_t k, int64_t ld_x, uint16_t* xs, hipStream_t st);   // float32 -> three bf16 planes, [3m][k]
int launch_splitk_reduce(const float* slab, int splits, int64_t m, int64_t n, const float* scale, const float* bias, float clip,
                         int act, int act_begin, void* y, int y_dtype, int64_t ld_y, hipStream_t st);

// Addends of tvl_ternary_matmul_residual, y = (y + add1) + add2 in y's type, for the duration of one call on the calling thread:
// launch_splitk_reduce takes them into its epilogue and marks them consumed when it writes the call's WHOLE output (y, m as
// recorded here: a product that runs as two row ranges is not); if no kernel of the route did, the entry point finishes with
// launch_add_rows.  (A thread-local hand-over instead of two more arguments on every launcher.)
struct ResidualArgs { const void* add1; int64_t ld1; const void* add2; int64_t ld2; const void* y; int64_t m; bool consumed; };
extern thread_local ResidualArgs* tls_residual;   // tvl_api.hip; nullptr outside such a call
// y[m, n] = (y + add1) + add2, each sum rounded to y's type (add2 may be NULL)
int launch_add_rows(void* y, int y_dtype, int64_t m, int64_t n, int64_t ld_y, const void* add1, int64_t ld1, const void* add2, int64_t ld2, hipStream_t st);
// host-side description of the tile plan launch_gemm_mfma would choose (tvl_ternary_matmul_route)
void gemm_mfma_describe(int x_dtype, int64_t m, int64_t n, int64_t k, size_t workspace_bytes, char* out, size_t out_len);

// gemm_mid.hip: 128-row tiles of the 16x16x32 kernel (with or without split-K into float32 slabs at `slab`)
int launch_gemm_mid(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                    const float* scale, const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y,
                    int bm, int bn, int splits, float* slab, int w_steps, hipStream_t st);

// device_info.hip: per-device facts looked up once
int device_cu_count();                                            // compute units of the current device (256 on MI355X)
int set_dynamic_lds_limit(const void* kernel, size_t lds_bytes);  // hipFuncSetAttribute(MaxDynamicSharedMemorySize), once per kernel and device

// gemm_rows.hip: few activation rows (3 .. 32 on layers where it wins), one launch, no workspace
bool gemm_rows_supported(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                         const void* workspace, size_t workspace_bytes);
int launch_gemm_rows(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                     const float* scale, const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y,
                     void* workspace, size_t workspace_bytes, hipStream_t st);

// gemm_rows.hip, larger layers: 3 .. 64 rows, the activations shared through LDS by 8 waves, k split into float32 slabs at `workspace`
bool gemm_rows_shared_supported(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                                const void* workspace, size_t workspace_bytes);
int launch_gemm_rows_shared(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                            const float* scale, const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y,
                            void* workspace, size_t workspace_bytes, hipStream_t st);
size_t gemm_rows_shared_workspace_bytes(int64_t m, int64_t n, int64_t k);

// gemm_skinny.hip
bool gemm_skinny_supported(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw);
int launch_gemm_skinny(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                       const float* scale, const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y, void* workspace,
                       size_t workspace_bytes, hipStream_t st);
size_t gemm_skinny_workspace_bytes(int64_t m, int64_t n, int64_t k);

// gemm_stream.hip
bool gemm_stream_supported(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                           const void* workspace, size_t workspace_bytes);
int launch_gemm_stream(const void* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                       const float* scale, const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y, void* workspace,
                       size_t workspace_bytes, hipStream_t st);
size_t gemm_stream_workspace_bytes(int64_t m, int64_t n, int64_t k);

// gemv.hip
bool gemv_supported(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw);
int launch_gemv(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y, hipStream_t st);

// popcount.hip
int launch_sign_bits(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, uint32_t* xb, int64_t ld_xb, hipStream_t st);
int launch_bitplanes(const uint8_t* pw, int64_t n, int64_t k, int64_t ld_pw, uint32_t* pos, uint32_t* neg, int64_t ld_planes, hipStream_t st);
int launch_popcount_matmul(const uint32_t* xb, int64_t m, int64_t ld_xb, const uint32_t* pos, const uint32_t* neg, int64_t n, int64_t ld_planes,
                           int64_t k, const float* scale, const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y, hipStream_t st);

// fit_scale.hip
size_t fit_workspace_bytes();
int launch_fit_ternary(const float* w, int64_t n, int64_t k, int64_t ld_w, int uniform, int iters, float* scale, float* latent,
                       int64_t ld_latent, void* workspace, hipStream_t st);
int launch_fit_binary(const float* w, int64_t n, int64_t k, int64_t ld_w, int mode, int uniform, float prescale, int natural,
                      float* scale, float* latent, int64_t ld_latent, void* workspace, hipStream_t st);

}  // namespace tvl

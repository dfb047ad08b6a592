/*
 * tvl.h - C ABI of libternary_hip.so: ternary / binary quantized matmul for AMD MI355X (gfx950).
 *
 * This is the drop-in boundary.  Each entry point names the reference interface it stands in
 * for (paths relative to the reference repository).  The reference's native operator sources
 * (native/tensorflow_custom_ops/cc) are not in its tree, and its packed ops come from an external
 * larq_compute_engine fork, so "replaces" means: same operator name, same argument meaning, the
 * arithmetic of the reference's Python layers.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the comment says otherwise; the caller owns all
 *     buffers (inputs and outputs); the library allocates nothing and keeps no pointer.
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream).
 *   - return value 0 = success.  TVL_ERR_INVALID = a bad argument (shape, dtype, alignment,
 *     NULL pointer): nothing was launched.  TVL_ERR_HIP = the HIP runtime reported an error.
 *     tvl_last_error() returns a description for the calling thread's last failure.
 *   - matrices are row-major; `ld_*` is the distance between rows IN ELEMENTS of that matrix
 *     (bytes for packed weights).  ld = 0 means "dense" (the natural row length).
 *   - packed weights ("format C"): uint8 [N, ld_pw >= ceil(K/4)], byte [n, j] holds weights
 *     k = 4j..4j+3 of output channel n, weight i in bits [2i+1:2i], code = value + 1
 *     (-1 -> 0, 0 -> 1, +1 -> 2; code 3 decodes to +2 exactly like the reference's
 *     `((b >> 2i) & 3) - 1`).  impl/tf/moondream/ternary.py:374-398 (pack), :507-520 (decode).
 *
 * Hardware assumptions
 *   - built for gfx950 (MI355X) only.  The tile order of the matrix-core kernels assumes 8 XCDs (workgroups with equal
 *     blockIdx % 8 share an L2): on a part with another count results stay correct, only the L2 reuse of the order is lost.
 *     The compute-unit count is read from the device, the kernel-choice cost tables are calibrated on MI355X.
 *
 * Tuning switches (environment variables, each read ONCE per process at the first matmul; results never depend on them,
 * only which kernel computes them; unset = the library's own choice; meant for tuning runs and the test suite, leave them
 * unset in production)
 *   TVL_FORCE_PLAN=kind:bm:bn:splits   pin the matrix-core tile and k split (kind 0 = 32x32x16 small tiles, 1 = 128-row tiles,
 *                                      2 = the 256 x 256 tile; splits -1 with 1:128:256 = two workgroups per compute unit)
 *   TVL_TAIL_SPLIT=0                   never split a ragged last round of big tiles into a second launch
 *   TVL_MID_PERSIST=0|1                pin the persistent form of the 128 x 256 tile off / on
 *   TVL_ROWS_STAGED=0|2                few-row kernel with x staged once in LDS: off / wherever it fits
 *   TVL_ROWS_SHARED=<splits>|-1        few-row kernel for larger layers: pin its k split / off;  TVL_ROWS_SHARED_ANY=1 lifts its size rules
 *   TVL_ROWS_BLOCKS=0|2                the same kernel on blocks of 64 rows (65 .. 512 rows, small layers): off / wherever it fits
 *   TVL_GEMV=0|2                       one- and two-row kernel: off / wherever it can run
 */
#ifndef TVL_H_
#define TVL_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TVL_VERSION 100 /* 0.1.0 */

enum { TVL_OK = 0, TVL_ERR_INVALID = 1, TVL_ERR_HIP = 2 };

/* element types of activations / outputs */
enum { TVL_F32 = 0, TVL_F16 = 1, TVL_BF16 = 2, TVL_I8 = 3, TVL_I32 = 4 };

/* weight quantizers.  binary/modules.py:21-26 (ternary), :5-10 (sign), :13-18 (step) */
enum { TVL_Q_TERNARY = 0, TVL_Q_SIGN = 1, TVL_Q_STEP = 2 };

/* activation fused after the clip.  GELU-tanh is what the reference attaches to fc1
 * (utils/conversion.py:185-187); it is applied after the op at impl/tf/moondream/ternary.py:298-299. */
enum { TVL_ACT_NONE = 0, TVL_ACT_GELU_TANH = 1 };

/* planar 1-D layouts of the pure-TF layer.  impl/tf/moondream/ternary.py:57-78 (A), :114-133 (B) */
enum { TVL_PLANAR_A = 0, TVL_PLANAR_B = 1 };

int tvl_version(void);
/* Host string, valid until the calling thread's next failing call. */
const char* tvl_last_error(void);
/* Host string describing the build (target arch, kernels). */
const char* tvl_build_info(void);

/* ---- quantize + pack (one-time, HBM-bound) ------------------------------------------------- */

/* Quantize latent weights w [n,k] (f32 / f16 / bf16) and pack them 2 bits each into pw.
 * mode TVL_Q_TERNARY: (w >= thr) - (w <= -thr), thresholds inclusive.
 * Replaces `ops.pack_fn(tf.cast(ternary_act(w), tf.int8))` at utils/conversion.py:71-72 and the
 * per-forward `tri_step_ste(self.weights)` at binary/modules.py:102.  Pad bits (k..4*ceil(k/4))
 * are written as code 1 (value 0); bytes beyond ceil(k/4) in a row are left untouched. */
int tvl_quantize_pack(const void* w, int w_dtype, int64_t n, int64_t k, int64_t ld_w, int mode,
                      float threshold, uint8_t* pw, int64_t ld_pw, void* stream);

/* tvl_quantize_pack for `count` matrices in as few launches as possible (a converted checkpoint: every selected layer of a
 * model into one packed arena - the loop of transfer_dense_params_torch_to_tf calls at utils/conversion.py:126-137, one
 * `ternary_act -> int8 -> ops.pack_fn` per Dense, :71-72).  `items` is a HOST array, read before the call returns; the
 * pointers inside it are device pointers; ld = 0 means dense; all matrices share w_dtype, mode and threshold.  Same bytes
 * as `count` single calls. */
typedef struct tvl_pack_item {
    const void* w;   /* latent weights [n, ld_w] */
    int64_t n, k, ld_w;
    uint8_t* pw;     /* packed rows [n, ld_pw] */
    int64_t ld_pw;
} tvl_pack_item;
int tvl_quantize_pack_batch(const tvl_pack_item* items, int64_t count, int w_dtype, int mode, float threshold, void* stream);

/* Pack already-quantized int8 values in {-1,0,+1} (code = value + 1, low two bits).
 * Replaces `ops.pack_fn(int8 [N,K])` (utils/conversion.py:72) / ScaledTernaryPackedMM.pack_fn
 * (impl/tf/moondream/ternary.py:374-398). */
int tvl_pack_i8(const int8_t* t, int64_t n, int64_t k, int64_t ld_t, uint8_t* pw, int64_t ld_pw,
                void* stream);

/* Unpack pw [n, ld_pw] to out [n, k] of out_dtype (f32 / f16 / bf16 / i8) holding {-1,0,+1,(+2)}.
 * Replaces `ops.unpack_ternary(pw, orig_dim)` at impl/tf/moondream/ternary.py:334. */
int tvl_unpack(const uint8_t* pw, int64_t n, int64_t k, int64_t ld_pw, void* out, int out_dtype,
               int64_t ld_out, void* stream);

/* Planar 1-D layouts A and B of ScaledTernaryTf: `len` int8 values -> ceil(len/4) bytes, and back
 * (4*packed_len values; the caller slices the first `len`).
 * Replaces pack_ternary / pack_ternary2 / unpack_ternary_iterative / unpack_ternary_iterative2
 * at impl/tf/moondream/ternary.py:57-78, :114-133, :101-110, :136-156. */
int tvl_pack_planar_i8(const int8_t* t, int64_t len, int format, uint8_t* pw, void* stream);
int tvl_unpack_planar(const uint8_t* pw, int64_t packed_len, int format, void* out, int out_dtype,
                      void* stream);

/* Dense quantizer: out[i] = q(w[i]) in the dtype of w (f32 / f16 / bf16) or int8 (out_dtype).
 * Replaces tri_step_ste / sign_ste / step_ste (binary/modules.py:5-26) as a forward value. */
int tvl_quantize(const void* w, int w_dtype, int64_t numel, int mode, float threshold, void* out,
                 int out_dtype, void* stream);

/* ---- the fused op ---------------------------------------------------------------------------- */

/* y[m,n] = act(clip((x[m,k] @ unpack(pw)[n,k]^T) * scale[n] + bias[n], -|clip|, +|clip|))
 * x: f16 / bf16 / f32 [m, ld_x]; y: f16 / bf16 / f32 [m, ld_y]; scale, bias: f32 [n] (bias may be
 * NULL).  Accumulation in float32; the epilogue is one f32 multiply, one f32 add (not fused), the
 * clip, the optional activation, then one rounding to y's type.
 * Replaces `ops.ternary_matmul(x, pw, scale, bias, clip)` at impl/tf/moondream/ternary.py:281 and
 * ScaledTernaryLinear.forward at binary/modules.py:101-109.
 * `workspace` (device, may be NULL, 256-byte aligned) lets the library pick kernels that need scratch
 * memory: split-K partial sums, and for float32 activations the three bf16 planes x = hi + mid + lo
 * (exact) that put a float32 product on the matrix cores, and - for a k that is not a multiple of 64 or rows
 * the fast kernels cannot load 16 bytes at a time - zero-padded copies of x and of the packed rows (pad
 * weights are code 1 = 0), so that those shapes run on the matrix cores as well.  Without it the same
 * results come from slower kernels.  tvl_ternary_matmul_workspace_bytes_for(x_dtype, m, n, k) gives the size that enables
 * everything; tvl_ternary_matmul_workspace_bytes(m, n, k) is the f16 / bf16 figure.  Contents need no
 * initialisation and are not preserved; calls that share a workspace must be ordered on one stream. */
int tvl_ternary_matmul(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x,
                       const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                       const float* bias, float clip, int act, void* y, int y_dtype, int64_t ld_y,
                       void* workspace, size_t workspace_bytes, void* stream);
/* The same product for two Dense layers that read the same input and are stored as ONE packed matrix (their
 * rows, scales and biases concatenated): channels [0, act_begin_channel) get no activation, channels
 * [act_begin_channel, n) get `act`.  The reference's Phi block feeds one LayerNorm output to both Wqkv (linear) and
 * fc1 (tanh-GELU) - impl/tf/moondream/layers.py:334-347 with the GELU folded into fc1 at utils/conversion.py:185-187 -
 * so the two run as one launch over n = 6144 + 8192 channels.  act_begin_channel = 0 is tvl_ternary_matmul. */
int tvl_ternary_matmul_act_from(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x,
                                const uint8_t* pw, int64_t n, int64_t ld_pw, const float* scale,
                                const float* bias, float clip, int act, int64_t act_begin_channel, void* y,
                                int y_dtype, int64_t ld_y, void* workspace, size_t workspace_bytes, void* stream);
size_t tvl_ternary_matmul_workspace_bytes(int64_t m, int64_t n, int64_t k);
size_t tvl_ternary_matmul_workspace_bytes_for(int x_dtype, int64_t m, int64_t n, int64_t k);
/* tvl_ternary_matmul_act_from followed by  y = (y + add1) + add2  in y's type, as ONE call: the closing line of the
 * reference's decoder block, `hidden_states = attn_outputs + feed_forward_hidden_states + residual`
 * (impl/tf/moondream/layers.py:345-347), rides on fc2's product with add1 = out_proj's output and add2 = the block's input.
 * add1 / add2: [m, ld_add* >= n] of y's type (ld 0 = dense), not overlapping y; add2, or both, may be NULL.  Every sum is rounded to y's type, so
 * the result has the bits of the unfused sequence (product, then two element-wise additions) - for float32 outputs the
 * reference's own value.  Where the chosen kernels end in the split-K reduction the addends ride in its epilogue; otherwise the
 * library appends one element-wise launch (still one call, no extra pass through the host). */
int tvl_ternary_matmul_residual(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n,
                                int64_t ld_pw, const float* scale, const float* bias, float clip, int act,
                                int64_t act_begin_channel, const void* add1, int64_t ld_add1, const void* add2,
                                int64_t ld_add2, void* y, int y_dtype, int64_t ld_y, void* workspace, size_t workspace_bytes,
                                void* stream);
/* Host-side description of the kernels tvl_ternary_matmul would launch for dense, aligned operands of this shape with a
 * workspace of `workspace_bytes` (names of the kernel templates in ternary-vllm_amd/csrc; no counterpart in the reference,
 * which never says which kernel ran).  For benchmark records and tests; `out` is a host buffer. */
int tvl_ternary_matmul_route(int x_dtype, int64_t m, int64_t n, int64_t k, size_t workspace_bytes, char* out, size_t out_len);

/* tvl_ternary_matmul_act_from for callers that cannot allocate device memory (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  The scratch memory the other
 * entry points take as `workspace` is a buffer built into the library instead: TVL_NOALLOC_SCRATCH_BYTES of device
 * memory per GPU, held until the process ends.  The call allocates and frees nothing.  Every call looks the buffer's
 * address up; the FIRST call on a GPU does so through hipGetSymbolAddress, a runtime call that is not a stream operation:
 * make one call (m = 0 is enough, it launches nothing) on each GPU before starting a stream capture there.  Same results, bit for bit, as tvl_ternary_matmul_act_from given a workspace of
 * tvl_ternary_matmul_workspace_bytes_for(x_dtype, m, n, k) bytes; a shape that needs more than the built-in buffer is
 * rejected with TVL_ERR_INVALID and nothing is launched (tvl_noalloc_scratch_bytes() returns the limit).  The device the
 * pointers live on must be current.  All calls on one GPU share that GPU's buffer: calls that overlap in time on one GPU
 * (different streams, or threads without ordering) race on it - order them on one stream.  The buffer's contents need
 * no initialisation and mean nothing between calls. */
#define TVL_NOALLOC_SCRATCH_BYTES (32u << 20)
int tvl_ternary_matmul_noalloc(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n,
                               int64_t ld_pw, const float* scale, const float* bias, float clip, int act,
                               int64_t act_begin_channel, void* y, int y_dtype, int64_t ld_y, void* stream);
size_t tvl_noalloc_scratch_bytes(void);

/* ---- binarized activations (larq `input_quantizer="ste_sign"`, impl/tf/moondream/ternary.py:36) ---- */

/* xb[m, w] bit i = (x[m, 32w+i] >= 0)  (sign_ste, binary/modules.py:5-10: +1 where x >= 0, else -1); pad bits 0.
 * xb: uint32 [m, ld_xb >= ceil(k/32)] (ld_xb == 0 means ceil(k/32)). */
int tvl_sign_pack_bits(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, uint32_t* xb,
                       int64_t ld_xb, void* stream);
/* Packed format-C rows -> two bit planes: pos bit = (code >= 2) i.e. weight +1, neg bit = (code == 0) i.e.
 * weight -1 (the (pos, neg) pair of the reference's pack_ternary, ternary.py:72-75).  Code 3 counts as +1.
 * pos, neg: uint32 [n, ld_planes >= ceil(k/32)] (0 means ceil(k/32)). */
int tvl_ternary_bitplanes(const uint8_t* pw, int64_t n, int64_t k, int64_t ld_pw, uint32_t* pos, uint32_t* neg,
                          int64_t ld_planes, void* stream);
/* y[m,n] = act(clip((sum_k s[m,k] t[n,k]) * scale[n] + bias[n])) with s = +-1 from xb and t from (pos, neg):
 * 2 (popc(xb & pos) - popc(xb & neg)) - (popc(pos) - popc(neg)), integer-exact, then the matmul's epilogue.
 * Equals tvl_ternary_matmul on the activations sign(x) bit for bit. */
int tvl_popcount_matmul(const uint32_t* xb, int64_t m, int64_t ld_xb, const uint32_t* pos, const uint32_t* neg,
                        int64_t n, int64_t ld_planes, int64_t k, const float* scale, const float* bias,
                        float clip, int act, void* y, int y_dtype, int64_t ld_y, void* stream);

/* Multiplication-free select-and-sum with DENSE float weights: res[m, i] = sum_{tm[i,k]>0} x[m,k]
 * - sum_{tm[i,k]<0} x[m,k].  Replaces ternary_mm at matrix_mul_tests.py:29-37 (f32 only). */
int tvl_ternary_mm_f32(const float* x, int64_t m, int64_t k, int64_t ld_x, const float* tm,
                       int64_t n, int64_t ld_tm, float* res, int64_t ld_res, void* stream);

/* ---- scale fits (one-time) -------------------------------------------------------------------- */

/* Ternary scale by the reference's masked-mean fixed point (binary/modules.py:280-293):
 * m0 = mean|w|; `iters` times: mask = |w| > m/2, m = sum(|w| mask)/count(mask).
 * uniform = 0: one m per row; uniform != 0: one m for the whole matrix (neuron_scale="uniform").
 * Writes scale[n] = m and, if latent != NULL, latent[n,k] = clip(w * (1/m), -1, 1).
 * workspace: tvl_fit_scale_workspace_bytes(n, k) bytes of device memory. */
int tvl_fit_ternary_scale_f32(const float* w, int64_t n, int64_t k, int64_t ld_w, int uniform,
                              int iters, float* scale, float* latent, int64_t ld_latent,
                              void* workspace, size_t workspace_bytes, void* stream);

/* Binary scales (binary/modules.py:197-210, :236-245): mode TVL_Q_SIGN: (sum_{w>0} w - sum_{w<0} w)
 * / count; TVL_Q_STEP: sum_{w>0} w / count; count = n*k (uniform) or k (per row).
 * latent (optional) = clip(w * prescale, -1, 1) with prescale = `prescale` when natural == 0, or
 * 1/scale[row] when natural != 0 (scaling="natural", :221-222). */
int tvl_fit_binary_scale_f32(const float* w, int64_t n, int64_t k, int64_t ld_w, int mode,
                             int uniform, float prescale, int natural, float* scale, float* latent,
                             int64_t ld_latent, void* workspace, size_t workspace_bytes,
                             void* stream);
size_t tvl_fit_scale_workspace_bytes(int64_t n, int64_t k);

/* ---- plumbing ops of the reference's op library ------------------------------------------------ */

/* out[0] = in[0], out[1..numel) = 0.  Replaces zero_out (custom_ops_mod.py:29). */
int tvl_zero_out_i32(const int32_t* in, int64_t numel, int32_t* out, void* stream);
/* out = constant (uint8 copy); `witness` is accepted and ignored: on this stack nothing folds
 * constants, the op exists so callers keep working.  Replaces prevent_folding
 * (custom_ops_mod.py:30; Python twin impl/tf/moondream/ternary.py:28-30). */
int tvl_prevent_folding_u8(const uint8_t* constant, int64_t numel, const void* witness,
                           uint8_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TVL_H_ */

// Packed-ternary x f16 matmul for few activation rows (m <= 16), weight-stream bound.
//
// The 2-bit codes are fed to the matrix cores WITHOUT arithmetic decode.  A code c in {0,1,2,3} sitting at
// bits [2s+1:2s] of an otherwise zero 16-bit field IS the f16 number c * 4^s * 2^-24 (bit patterns below
// 0x0800 are linear in their integer value: subnormals, then the first binade), and
// v_mfma_f32_16x16x32_f16 multiplies f16 subnormals exactly (tools/hwtests/mfma_denorm.hip).  Its adder
// tree, however, aligns the products by the NOMINAL exponent of a subnormal (2^-14), so the leading zeros of the
// mantissa cost accumulation precision: codes at bits 0-1 / 2-3 lose 7 / 4 bits against a normal operand, codes
// at bits 6-7 and 8-9 lose nothing (tools/hwtests/mfma_subnormal_precision.hip).  Only those two positions are
// used; the other codes are shifted there first:
//     code 3, 4 of each halfword:  d        & 0x00C000C0 ,  d        & 0x03000300
//     code 5, 6                 : (d >> 4)  & 0x00C000C0 , (d >> 4)  & 0x03000300
//     code 1, 2                 : (d << 4)  & 0x00C000C0 , (d << 4)  & 0x03000300
//     code 7, 0                 : (d >> 8)  & 0x00C000C0 , (d << 8)  & 0x03000300
// (a shift moves both halfwords of the dword at once; whatever crosses the halfword boundary is masked away):
// 12 VALU instructions per 16 weights instead of 20 for an arithmetic decode.  The price is bookkeeping:
//   * the two positions carry different powers (2^-18 and 2^-16), so there are two float32 accumulators per
//     output; the epilogue adds  acc3 * 2^18 + acc4 * 2^16  (exact scalings);
//   * the codes are value + 1, so  sum_k (c_k - 1) x_k = sum_k c_k x_k - sum_k x_k ; the second sum comes from
//     MFMAs whose weight operand is all ones (shared between the four waves);
//   * the two codes of one register sit 8 weights apart, and the four registers of an operand come from the
//     lane's four dwords, so MFMA number j of a 64-weight group contracts k = 16 i + 8 e + j (i < 4, e < 2):
//     the activations of the workgroup's whole k range are transposed to that pattern ONCE, in registers
//     (v_perm), when they are staged into LDS at the start of the kernel.
// Layout of the work: a workgroup = 4 waves, each wave owns TILES x 16 output channels (accumulator rows) and
// all activation rows (accumulator columns, padded to 16), over a k range of at most 64/RA steps of 256.
// The weight stream never waits for anything but itself: each wave keeps D steps of its own channels' bytes
// in flight as LDS-DMA pieces (1 KiB = 16 channels x 64 bytes) into a private LDS ring and retires them with
// counted s_waitcnt vmcnt - no barrier and no register is held by a load in flight.  k is split over
// blockIdx.y (float32 slabs + the deterministic reduce of gemm_mfma.hip).
// Requires f16 activations, m <= 16, k % 256 == 0, 16-byte aligned activation rows and packed rows.
#include "tvl_common.h"
#include "tvl_kernels.h"

#include <type_traits>

namespace tvl {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int STREAM_IMAGE_BYTES = 32 * 1024;   // activation image of one workgroup: RA rows x at most 64 / RA steps
constexpr int STREAM_DEPTH = 4;                 // steps of weights in flight per wave (deeper or "nt" loads: no gain measured)

// SPLIT: blockIdx.y selects a k range; the raw float32 partial results go to slab blockIdx.y (float [splits][m][n])
// and splitk_reduce_kernel (gemm_mfma.hip) sums the slabs in a fixed order and applies the epilogue.  (Folding that
// reduction into this launch - last-arriving workgroup per channel block, write-through slab stores, one ticket
// atomic - was measured and is not faster: the serial tail of the last arriver costs what the second launch costs.)
template <typename TY, int TILES, int RA, int D, bool SPLIT>
__global__ void __launch_bounds__(256, 2)
gemm_stream_kernel(const f16* __restrict__ x, int m, int k, int64_t ld_x, const uint8_t* __restrict__ pw, int n, int64_t ld_pw,
                   const float* __restrict__ scale, const float* __restrict__ bias, float clip, int act, int act_begin, TY* __restrict__ y, int64_t ld_y,
                   int steps_per_split, float* __restrict__ slab) {
    constexpr int RING = D + 1;                       // ring slots (steps) per wave
    constexpr int WRING = RING * TILES * 1024;        // bytes of one wave's weight ring
    constexpr int IMG_OFF = 4 * WRING;                // activation image [step][g][j][row][slot] behind the rings
    constexpr int STEP_IMG = 4 * 8 * RA * 16;         // image bytes per step
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int c = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * (64 * TILES) + wid * (16 * TILES);
    const int total_steps = k / 256;
    const int s0 = SPLIT ? (int)blockIdx.y * steps_per_split : 0;
    const int ns = SPLIT ? (total_steps - s0 < steps_per_split ? total_steps - s0 : steps_per_split) : total_steps;
    // ---- weight stream: lane (c, g), tile t fetches bytes [64 step + 16 g, +16) of channel n0 + 16 t + c -----
    const uint8_t* b_row[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        int gn = n0 + 16 * t + c;
        gn = gn < n ? gn : n - 1;
        b_row[t] = pw + (int64_t)gn * ld_pw + (int64_t)s0 * 64 + g * 16;
    }
    const uint32_t ring_base = (uint32_t)(uintptr_t)(lptr_t)lds + wid * WRING;
    auto issue = [&](int step, int slot) {
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const uint8_t* src = b_row[t] + (int64_t)step * 64;
            const uint32_t dst = __builtin_amdgcn_readfirstlane(ring_base + (slot * TILES + t) * 1024);
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        }
    };
    // ---- activations: unit = (row, 64-k group); 8 chunks of 8 -> 8 fragments of 8 (an 8x8 transpose) ---------
    // The loads of every thread's first unit are issued BEFORE the weight stream (memory returns loads in issue
    // order: behind D steps of HBM traffic they would arrive last) and waited for with a count that leaves
    // exactly the weight pieces in flight.
    const int groups = ns * 4;
    const int units = m * groups;
    auto transpose_store = [&](const u32x4 (&ch)[8], int row, int grp) {
        unsigned char* dst = lds + IMG_OFF + (grp * 8 * RA + row) * 16;   // + j * RA * 16
        const uint32_t* w0 = reinterpret_cast<const uint32_t*>(&ch[0]);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;
            uint4 o;   // register i of fragment j = (chunk 2i [j], chunk 2i+1 [j])
            o.x = __builtin_amdgcn_perm(w0[4 * 1 + (j >> 1)], w0[4 * 0 + (j >> 1)], sel);
            o.y = __builtin_amdgcn_perm(w0[4 * 3 + (j >> 1)], w0[4 * 2 + (j >> 1)], sel);
            o.z = __builtin_amdgcn_perm(w0[4 * 5 + (j >> 1)], w0[4 * 4 + (j >> 1)], sel);
            o.w = __builtin_amdgcn_perm(w0[4 * 7 + (j >> 1)], w0[4 * 6 + (j >> 1)], sel);
            *reinterpret_cast<uint4*>(dst + j * RA * 16) = o;
        }
    };
    {
        const int u = tid < units ? tid : 0;   // threads without a unit load unit 0 (never stored): same load count for every wave
        const int row = u % m, grp = u / m;
        const f16* src = x + (int64_t)row * ld_x + (int64_t)s0 * 256 + grp * 64;
        u32x4 ch[8];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(ch[q]) : "v"(src), "n"(q * 16) : "memory");
        // always D steps (a k range shorter than D re-fetches its last step into slots nobody reads), so that the
        // count below is a constant: one wait statement, no branch between the loads and their wait
#pragma unroll
        for (int d = 0; d < D; ++d) issue(d < ns ? d : ns - 1, d);
        // wait for the 8 activation loads only: D * TILES weight pieces were issued after them
        asm volatile("s_waitcnt vmcnt(%8)" : "+v"(ch[0]), "+v"(ch[1]), "+v"(ch[2]), "+v"(ch[3]), "+v"(ch[4]), "+v"(ch[5]), "+v"(ch[6]), "+v"(ch[7])
                     : "n"(D * TILES) : "memory");
        if (tid < units) transpose_store(ch, row, grp);
    }
    for (int u = tid + 256; u < units; u += 256) {
        const int row = u % m, grp = u / m;
        const u32x4* src = reinterpret_cast<const u32x4*>(x + (int64_t)row * ld_x + (int64_t)s0 * 256 + grp * 64);
        u32x4 ch[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) ch[q] = src[q];
        transpose_store(ch, row, grp);
    }
    __syncthreads();   // image visible to all waves

    floatx4 acc[2][TILES], accx = {0.f, 0.f, 0.f, 0.f};   // acc[0]: codes at bits 6-7 (2^-18), acc[1]: bits 8-9 (2^-16)
    const int wid_s = __builtin_amdgcn_readfirstlane(wid);
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int t = 0; t < TILES; ++t) acc[s][t] = (floatx4){0.f, 0.f, 0.f, 0.f};
    const u32x4 ones = {0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u};  // f16 1.0 x 8

    const unsigned char* my_ring = lds + wid * WRING + lane * 16;
    const unsigned char* my_img = lds + IMG_OFF + (g * 8 * RA + c % RA) * 16;   // lanes c >= RA alias a valid row: never stored
    auto compute = [&](int i, int slot) {
        u32x4 w[TILES];
#pragma unroll
        for (int t = 0; t < TILES; ++t) w[t] = *reinterpret_cast<const u32x4*>(my_ring + (slot * TILES + t) * 1024);
        const unsigned char* img = my_img + i * STEP_IMG;
        // MFMA j contracts the codes number j of every halfword: which shift brings them to bits 6-7 / 8-9
        //   j:      0    1    2    3    4    5    6    7
        //   shift: <<8  <<4  <<4   0    0   >>4  >>4  >>8
        //   bits:  8-9  6-7  8-9  6-7  8-9  6-7  8-9  6-7
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int cls = (j == 0 || j == 2 || j == 4 || j == 6) ? 1 : 0;
            const uint32_t mask = cls ? 0x03000300u : 0x00C000C0u;
            const half8 xf = *reinterpret_cast<const half8*>(img + j * RA * 16);
            // sum of x: the four waves need the same number, so each adds up a quarter of the fragments
            if ((j & 3) == wid_s) accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, ones), xf, accx, 0, 0, 0);
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                u32x4 a;
                if (j == 0) a = w[t] << 8;
                else if (j == 1 || j == 2) a = w[t] << 4;
                else if (j == 3 || j == 4) a = w[t];
                else if (j == 5 || j == 6) a = w[t] >> 4;
                else a = w[t] >> 8;
                a &= mask;
                acc[cls][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(half8, a), xf, acc[cls][t], 0, 0, 0);
            }
        }
    };
    // Steady state: D steps in flight behind the one being consumed; the counted wait retires the oldest.
    int slot = 0, i = 0;
    for (; i + D < ns; ++i) {
        int islot = slot + D;
        islot = islot >= RING ? islot - RING : islot;
        issue(i + D, islot);
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(D * TILES) : "memory");   // step i has landed
        compute(i, slot);
        slot = slot + 1 == RING ? 0 : slot + 1;
    }
    // Tail: nothing left to issue; r steps remain in flight behind the one consumed next.
#pragma unroll
    for (int r = D - 1; r >= 0; --r) {
        if (ns > r) {
            asm volatile("s_waitcnt vmcnt(%0)" : : "n"(r * TILES) : "memory");
            compute(ns - 1 - r, slot);
            slot = slot + 1 == RING ? 0 : slot + 1;
        }
    }

    // ---- sum of x over the k range: the waves' quarter sums, added in wave order (every row of the ones-tile is equal)
    __shared__ float s_sumx[4][16];
    if (g == 0) s_sumx[wid][c] = accx[0];
    __syncthreads();
    const float sumx = ((s_sumx[0][c] + s_sumx[1][c]) + s_sumx[2][c]) + s_sumx[3][c];

    // ---- epilogue: lane (c, g) holds activation row c, channels n0 + 16 t + 4 g + e ------------------------------
    const float cl = fabsf(clip);
    const bool has_bias = bias != nullptr;
    float v[TILES][4];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float sum = acc[0][t][e] * 262144.0f + acc[1][t][e] * 65536.0f;   // 2^18, 2^16: exact scalings
            v[t][e] = sum - sumx;
        }
    // A non-finite activation breaks the identity sum (c - 1) x = sum c x - sum x: code 0 times infinity is NaN inside the
    // matrix instruction and the other codes leave infinity - infinity, where the product with the decoded weight is
    // +-infinity (weight +-1) or NaN (weight 0).  Such a row shows in its sum of x; its outputs are redone here with
    // decoded weights, one multiply-add per weight straight from memory (rare, so not tuned).
    if (c < m && !(fabsf(sumx) <= 3.4028234e38f)) {
        const f16* xr = x + (int64_t)c * ld_x + (int64_t)s0 * 256;
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int gn = n0 + 16 * t + 4 * g + e;
                gn = gn < n ? gn : n - 1;
                const uint8_t* wr = pw + (int64_t)gn * ld_pw + (int64_t)s0 * 64;
                float a = 0.0f;
                for (int kk = 0; kk < ns * 256; ++kk) a += (float)code_value(wr[kk >> 2], kk & 3) * (float)xr[kk];
                v[t][e] = a;
            }
    }
    if constexpr (!SPLIT) {
        const bool vec_ok = ((reinterpret_cast<uintptr_t>(y) | (uintptr_t)(ld_y * (int64_t)sizeof(TY))) & 15) == 0;
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            const int cn = n0 + 16 * t + 4 * g;
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o[e] = epilogue_rt(v[t][e], cn + e < n ? scale[cn + e] : 0.0f, (has_bias && cn + e < n) ? bias[cn + e] : 0.0f, has_bias, act, cl, cn + e, act_begin);
            if (c < m) {
                TY* yrow = y + (int64_t)c * ld_y;
                if (vec_ok && cn + 4 <= n) {
                    if constexpr (sizeof(TY) == 4) {
                        *reinterpret_cast<float4*>(reinterpret_cast<float*>(yrow) + cn) = make_float4(o[0], o[1], o[2], o[3]);
                    } else {
                        TY tmp[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) tmp[e] = from_float<TY>(o[e]);
                        *reinterpret_cast<uint2*>(yrow + cn) = *reinterpret_cast<const uint2*>(tmp);
                    }
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (cn + e < n) yrow[cn + e] = from_float<TY>(o[e]);
                }
            }
        }
    } else {
        const int64_t slab_stride = (int64_t)m * n;
        float* mine = slab + (int64_t)blockIdx.y * slab_stride;
        if (c < m) {
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                const int cn = n0 + 16 * t + 4 * g;
                if ((n & 3) == 0 && cn + 4 <= n) {
                    *reinterpret_cast<float4*>(mine + (int64_t)c * n + cn) = make_float4(v[t][0], v[t][1], v[t][2], v[t][3]);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (cn + e < n) mine[(int64_t)c * n + cn + e] = v[t][e];
                }
            }
        }
    }
}

static inline int stream_ra(int64_t m) { return m <= 1 ? 1 : (m <= 2 ? 2 : (m <= 4 ? 4 : (m <= 8 ? 8 : 16))); }

bool gemm_stream_supported(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                           const void* workspace, size_t workspace_bytes) {
    if (x_dtype != TVL_F16) return false;
    if (m < 1 || m > 16 || k <= 0 || k % 256 != 0) return false;
    if (!aligned(x, 16) || (ld_x * 2) % 16 != 0) return false;
    if (!aligned(pw, 16) || ld_pw % 16 != 0) return false;
    if (n > INT32_MAX - 512 || k > INT32_MAX) return false;
    // a workgroup keeps its whole k range of the activations in LDS: longer k needs the split-K workspace
    const int ra = stream_ra(m);
    const int64_t max_steps = STREAM_IMAGE_BYTES / (512 * ra);
    const int64_t min_splits = ceil_div(k / 256, max_steps);
    if (min_splits > 1) {
        if (workspace == nullptr || !aligned(workspace, 16)) return false;
        if ((size_t)min_splits * m * n * sizeof(float) > workspace_bytes) return false;
    }
    return true;
}

struct StreamPlan { int ra, per, splits; };   // per = steps of 256 k per split
static StreamPlan plan_stream(int64_t m, int64_t n, int64_t k, size_t workspace_bytes) {
    StreamPlan p;
    p.ra = stream_ra(m);
    const int64_t steps = k / 256;
    const int64_t max_steps = STREAM_IMAGE_BYTES / (512 * p.ra);
    const int64_t nblk = ceil_div(n, 128);
    int64_t s = ceil_div(1024, nblk);                                // ~4 workgroups per CU over the whole grid
    const int64_t by_bytes = k / (16 * m) > 0 ? k / (16 * m) : 1;   // slab bytes <= packed weight bytes
    const int64_t by_ws = (int64_t)(workspace_bytes / ((size_t)m * n * sizeof(float)));
    s = s < steps ? s : steps;
    s = s < by_bytes ? s : by_bytes;
    s = s < by_ws ? s : by_ws;
    const int64_t need = ceil_div(steps, max_steps);                 // the image must fit
    s = s < need ? need : s;
    s = s < need ? need : s;
    s = s < 1 ? 1 : s;
    p.per = (int)ceil_div(steps, s);
    p.splits = (int)ceil_div(steps, p.per);
    return p;
}

size_t gemm_stream_workspace_bytes(int64_t m, int64_t n, int64_t k) {
    if (m > 16) return 0;
    const StreamPlan p = plan_stream(m, n, k, (size_t)1 << 40);
    return p.splits > 1 ? (size_t)p.splits * m * n * sizeof(float) : 0;
}

template <typename TY, int RA>
static int launch_stream_cfg(const f16* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                             const float* scale, const float* bias, float clip, int act, int act_begin, TY* y, int64_t ld_y, const StreamPlan& p,
                             unsigned char* ws, hipStream_t st) {
    constexpr int TILES = 2, D = STREAM_DEPTH;
    const int nblk = (int)ceil_div(n, 64 * TILES);
    const size_t lds_bytes = 4 * (size_t)(D + 1) * TILES * 1024 + (size_t)p.per * 512 * RA;
    const dim3 block(256);
    if (p.splits > 1) {
        auto kern = gemm_stream_kernel<TY, TILES, RA, D, true>;
        if (int rc_ = set_dynamic_lds_limit((const void*)kern, lds_bytes); rc_ != TVL_OK) return rc_;
        hipLaunchKernelGGL(kern, dim3(nblk, p.splits), block, lds_bytes, st, x, (int)m, (int)k, ld_x, pw, (int)n, ld_pw, scale, bias, clip, act, act_begin,
                           y, ld_y, p.per, reinterpret_cast<float*>(ws));
        TVL_LAUNCH_CHECK();
        return launch_splitk_reduce(reinterpret_cast<const float*>(ws), p.splits, m, n, scale, bias, clip, act, act_begin, y,
                                    std::is_same<TY, float>::value ? TVL_F32 : (std::is_same<TY, f16>::value ? TVL_F16 : TVL_BF16), ld_y, st);
    } else {
        auto kern = gemm_stream_kernel<TY, TILES, RA, D, false>;
        if (int rc_ = set_dynamic_lds_limit((const void*)kern, lds_bytes); rc_ != TVL_OK) return rc_;
        hipLaunchKernelGGL(kern, dim3(nblk, 1), block, lds_bytes, st, x, (int)m, (int)k, ld_x, pw, (int)n, ld_pw, scale, bias, clip, act, act_begin,
                           y, ld_y, 0, (float*)nullptr);
    }
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

template <typename TY>
static int launch_stream_y(const f16* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                           const float* scale, const float* bias, float clip, int act, int act_begin, TY* y, int64_t ld_y, void* ws, size_t ws_bytes,
                           hipStream_t st) {
    const bool ws_ok = ws != nullptr && aligned(ws, 16);
    const StreamPlan p = plan_stream(m, n, k, ws_ok ? ws_bytes : 0);
    if (p.ra == 1) return launch_stream_cfg<TY, 1>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, p, (unsigned char*)ws, st);
    if (p.ra == 2) return launch_stream_cfg<TY, 2>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, p, (unsigned char*)ws, st);
    if (p.ra == 4) return launch_stream_cfg<TY, 4>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, p, (unsigned char*)ws, st);
    if (p.ra == 8) return launch_stream_cfg<TY, 8>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, p, (unsigned char*)ws, st);
    return launch_stream_cfg<TY, 16>(x, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, y, ld_y, p, (unsigned char*)ws, st);
}

int launch_gemm_stream(const void* x, int64_t m, int64_t k, int64_t ld_x, const uint8_t* pw, int64_t n, int64_t ld_pw,
                       const float* scale, const float* bias, float clip, int act, int act_begin, void* y, int y_dtype, int64_t ld_y, void* workspace,
                       size_t workspace_bytes, hipStream_t st) {
    const f16* xh = (const f16*)x;
    switch (y_dtype) {
        case TVL_F32: return launch_stream_y(xh, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, (float*)y, ld_y, workspace, workspace_bytes, st);
        case TVL_F16: return launch_stream_y(xh, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, (f16*)y, ld_y, workspace, workspace_bytes, st);
        case TVL_BF16: return launch_stream_y(xh, m, k, ld_x, pw, n, ld_pw, scale, bias, clip, act, act_begin, (bf16_t*)y, ld_y, workspace, workspace_bytes, st);
        default: set_error("ternary_matmul: unsupported output dtype %d", y_dtype); return TVL_ERR_INVALID;
    }
}

}  // namespace tvl

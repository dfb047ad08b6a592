// Quantize / pack / unpack kernels (HBM-bound byte and bit work) for gfx950.
//
// Packed layout ("format C", impl/tf/moondream/ternary.py:374-398): byte [n, j] holds weights
// k = 4j..4j+3, weight i in bits [2i+1:2i], code = value + 1.
#include "tvl_common.h"
#include "tvl_kernels.h"

namespace tvl {

// ---- quad OR-reduction through DPP (lanes 4q..4q+3 end up holding the same dword) ------------
__device__ __forceinline__ uint32_t quad_or(uint32_t v) {
    // quad_perm [1,0,3,2] = 0xB1, [2,3,0,1] = 0x4E ; all rows/banks enabled
    v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true);
    v |= (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true);
    return v;
}

template <typename T> struct Vec4;
template <> struct Vec4<float> { using type = float4; };
template <> struct Vec4<f16> { using type = uint2; };
template <> struct Vec4<bf16_t> { using type = uint2; };

template <typename T> __device__ __forceinline__ void load4(const T* p, float (&o)[4]);
// The latent weights are read exactly once: non-temporal loads (no cache allocation) stream them 12 % faster
// (4096^2: 14.9 -> 13.2 us; 16384 x 4096: 5.4 -> 6.1 TB/s by in-process A/B).
typedef float nt_float4 __attribute__((ext_vector_type(4)));
typedef uint32_t nt_uint2 __attribute__((ext_vector_type(2)));
typedef uint32_t nt_u32x4 __attribute__((ext_vector_type(4)));
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&o)[4]) {
    const nt_float4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_float4*>(p));
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
}
__device__ __forceinline__ uint2 nt_load8(const void* p) {
    const nt_uint2 v = __builtin_nontemporal_load(reinterpret_cast<const nt_uint2*>(p));
    return make_uint2(v[0], v[1]);
}
template <> __device__ __forceinline__ void load4<f16>(const f16* p, float (&o)[4]) {
    uint2 v = nt_load8(p);
    const f16* h = reinterpret_cast<const f16*>(&v);
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = (float)h[i];
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float (&o)[4]) {
    uint2 v = nt_load8(p);
    o[0] = __uint_as_float(v.x << 16); o[1] = __uint_as_float(v.x & 0xFFFF0000u);
    o[2] = __uint_as_float(v.y << 16); o[3] = __uint_as_float(v.y & 0xFFFF0000u);
}

// Fast quantize+pack: one wave turns 1024 consecutive weights of one row (4 coalesced 16-B-per-lane
// loads for f32) into 256 packed bytes written by ONE coalesced dword-per-lane store.
//   load i (0..3): lane l reads weights [4*(64 i + l), +4)  -> byte 64 i + l of the chunk
//   quad_or gathers bytes 4q..4q+3 of load i into dword 16 i + q (held by the 4 lanes of quad q)
//   lane l stores dword 16 (l & 3) + (l >> 2)
// Requires k % 16 == 0, 16-byte aligned rows of w, 4-byte aligned rows of pw.
// One wave, one chunk: 1024 consecutive weights of a row, starting at weight kc (src points at them) -> 256 packed bytes of that
// row.  In two halves so that a wave can have the loads of its NEXT chunk in flight while it packs the current one.
template <typename T>
struct PackChunk { float v[4][4]; bool ok[4]; };

template <typename T>
__device__ __forceinline__ void pack_chunk_load(PackChunk<T>& ch, const T* __restrict__ src, int64_t kc, int64_t k, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t e = (int64_t)(64 * i + lane) * 4;
        ch.ok[i] = (kc + e) < k;
        if (ch.ok[i]) load4<T>(src + e, ch.v[i]);
    }
}

template <typename T, int MODE>
__device__ __forceinline__ void pack_chunk_store(const PackChunk<T>& ch, int64_t kc, int64_t k, float thr, uint8_t* __restrict__ pw_row, int lane) {
    uint32_t d[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        uint32_t b = 0;
        if (ch.ok[i]) {
            b = quantize_code<MODE>(ch.v[i][0], thr) | (quantize_code<MODE>(ch.v[i][1], thr) << 2) |
                (quantize_code<MODE>(ch.v[i][2], thr) << 4) | (quantize_code<MODE>(ch.v[i][3], thr) << 6);
        }
        d[i] = quad_or(b << (8 * (lane & 3)));
    }
    const int sel = lane & 3;
    const uint32_t mine = sel == 0 ? d[0] : sel == 1 ? d[1] : sel == 2 ? d[2] : d[3];
    const int64_t dw = kc / 16 + 16 * sel + (lane >> 2);
    if (dw < k / 16) *reinterpret_cast<uint32_t*>(pw_row + dw * 4) = mine;
}

template <typename T, int MODE>
__device__ __forceinline__ void pack_chunk(const T* __restrict__ src, int64_t kc, int64_t k, float thr, uint8_t* __restrict__ pw_row, int lane) {
    PackChunk<T> ch;
    pack_chunk_load<T>(ch, src, kc, k, lane);
    pack_chunk_store<T, MODE>(ch, kc, k, thr, pw_row, lane);
}

template <typename T, int MODE>
__global__ void __launch_bounds__(256) quantize_pack_fast_kernel(const T* __restrict__ w, int64_t n, int64_t k,
                                                                 int64_t ld_w, float thr, uint8_t* __restrict__ pw,
                                                                 int64_t ld_pw) {
    const int lane = threadIdx.x & 63;
    const int64_t chunks_per_row = (k + 1023) / 1024;
    const int64_t total = n * chunks_per_row;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    // (two chunks in flight per wave - the next chunk's loads issued before the current one is packed - measured no faster on large
    // matrices and 10 % slower in the batched kernel: with 32 waves per CU the memory system is already saturated at ~5.7 TB/s)
    for (int64_t c = wave0; c < total; c += nwaves) {
        const int64_t row = c / chunks_per_row;
        const int64_t kc = (c - row * chunks_per_row) * 1024;  // first weight of the chunk
        pack_chunk<T, MODE>(w + row * ld_w + kc, kc, k, thr, pw + row * ld_pw, lane);
    }
}

// Several matrices in ONE launch (a converted checkpoint: every layer of the packed arena).  The descriptors travel by value in
// the kernel arguments; a wave finds the matrix of its chunk by bisection over the chunk prefix sums.
constexpr int PACK_BATCH_MAX = 64;   // 3.1 KiB of kernel arguments
struct PackBatch {
    const void* w[PACK_BATCH_MAX];
    uint8_t* pw[PACK_BATCH_MAX];
    int64_t k[PACK_BATCH_MAX], ld_w[PACK_BATCH_MAX], ld_pw[PACK_BATCH_MAX];
    int64_t chunk_begin[PACK_BATCH_MAX + 1];   // prefix sums of n * ceil(k / 1024)
    int count;
};
template <typename T, int MODE>
__global__ void __launch_bounds__(256) quantize_pack_batch_kernel(const PackBatch b, float thr) {
    const int lane = threadIdx.x & 63;
    const int64_t total = b.chunk_begin[b.count];
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t nwaves = (int64_t)gridDim.x * 4;
    // c grows from one iteration to the next, so the matrix index only moves forward: one bisection for the first chunk, then a
    // step to the next matrix whenever c passes the current one's end (descriptor reads are scalar loads from the kernel
    // arguments: a bisection per chunk would cost more than the chunk's own loads)
    int lo = 0;
    {
        int hi = b.count;              // chunk_begin[lo] <= wave0 < chunk_begin[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (wave0 >= b.chunk_begin[mid]) lo = mid; else hi = mid;
        }
    }
    int64_t begin = b.chunk_begin[lo], end = b.chunk_begin[lo + 1];
    int64_t k = b.k[lo], ld_w = b.ld_w[lo], ld_pw = b.ld_pw[lo], chunks_per_row = (k + 1023) / 1024;
    const T* w = static_cast<const T*>(b.w[lo]);
    uint8_t* pw = b.pw[lo];
    auto locate = [&](int64_t c, const T*& src, uint8_t*& dst_row, int64_t& kc) {   // advances the matrix cursor; c never decreases
        while (c >= end) {
            ++lo;
            begin = end; end = b.chunk_begin[lo + 1];
            k = b.k[lo]; ld_w = b.ld_w[lo]; ld_pw = b.ld_pw[lo]; chunks_per_row = (k + 1023) / 1024;
            w = static_cast<const T*>(b.w[lo]); pw = b.pw[lo];
        }
        const int64_t lc = c - begin;
        const int64_t row = lc / chunks_per_row;
        kc = (lc - row * chunks_per_row) * 1024;
        src = w + row * ld_w + kc;
        dst_row = pw + row * ld_pw;
    };
    for (int64_t c = wave0; c < total; c += nwaves) {
        const T* src;
        uint8_t* dst;
        int64_t kc;
        locate(c, src, dst, kc);
        pack_chunk<T, MODE>(src, kc, k, thr, dst, lane);
    }
}

template <typename T, int MODE>
__global__ void __launch_bounds__(256) quantize_pack_generic_kernel(const T* __restrict__ w, int64_t n, int64_t k,
                                                                    int64_t ld_w, float thr, uint8_t* __restrict__ pw,
                                                                    int64_t ld_pw) {
    const int64_t bytes_per_row = (k + 3) / 4;
    const int64_t total = n * bytes_per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / bytes_per_row, j = i - row * bytes_per_row;
        uint32_t b = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t kk = 4 * j + e;
            const uint32_t code = kk < k ? quantize_code<MODE>(to_float(w[row * ld_w + kk]), thr) : 1u;
            b |= code << (2 * e);
        }
        pw[row * ld_pw + j] = (uint8_t)b;
    }
}

__global__ void __launch_bounds__(256) pack_i8_kernel(const int8_t* __restrict__ t, int64_t n, int64_t k, int64_t ld_t,
                                                      uint8_t* __restrict__ pw, int64_t ld_pw, int vec) {
    const int64_t bytes_per_row = (k + 3) / 4;
    const int64_t total = n * bytes_per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / bytes_per_row, j = i - row * bytes_per_row;
        uint32_t b;
        if (vec && 4 * j + 3 < k) {
            // four int8 in one dword: code = (value + 1) & 3 per byte; masking to two bits first
            // keeps the per-byte add free of carries into the neighbouring byte
            const uint32_t v = *reinterpret_cast<const uint32_t*>(t + row * ld_t + 4 * j);
            const uint32_t nc = ((v & 0x03030303u) + 0x01010101u) & 0x03030303u;
            b = (nc & 0x3u) | ((nc >> 6) & 0xCu) | ((nc >> 12) & 0x30u) | ((nc >> 18) & 0xC0u);
        } else {
            b = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t kk = 4 * j + e;
                const uint32_t code = kk < k ? ((uint32_t)(t[row * ld_t + kk] + 1) & 3u) : 1u;
                b |= code << (2 * e);
            }
        }
        pw[row * ld_pw + j] = (uint8_t)b;
    }
}

// Fast path of pack_i8_kernel: a lane turns 16 consecutive int8 (one 16-byte load) into one dword of packed bytes.
// Requires k % 16 == 0, 16-byte aligned int8 rows, 4-byte aligned packed rows.
__device__ __forceinline__ uint32_t pack4_i8(uint32_t v) {
    // four int8 in one dword: code = (value + 1) & 3 per byte; masking to two bits first keeps the per-byte add free
    // of carries into the neighbouring byte
    const uint32_t nc = ((v & 0x03030303u) + 0x01010101u) & 0x03030303u;
    return (nc & 0x3u) | ((nc >> 6) & 0xCu) | ((nc >> 12) & 0x30u) | ((nc >> 18) & 0xC0u);
}

__global__ void __launch_bounds__(256) pack_i8_fast_kernel(const int8_t* __restrict__ t, int64_t n, int64_t k, int64_t ld_t,
                                                           uint8_t* __restrict__ pw, int64_t ld_pw) {
    const int64_t gpr = k / 16, total = n * gpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t row, g;
        if (total <= 0x7fffffff) { row = (uint32_t)i / (uint32_t)gpr; g = (uint32_t)i - (uint32_t)row * (uint32_t)gpr; }   // 32-bit divide when it fits
        else { row = i / gpr; g = i - row * gpr; }
        const nt_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4*>(t + row * ld_t + 16 * g));
        *reinterpret_cast<uint32_t*>(pw + row * ld_pw + 4 * g) = pack4_i8(v[0]) | (pack4_i8(v[1]) << 8) | (pack4_i8(v[2]) << 16) | (pack4_i8(v[3]) << 24);
    }
}

template <typename T> __device__ __forceinline__ T value_as(int v) { return from_float<T>((float)v); }

template <typename T>
__global__ void __launch_bounds__(256) unpack_kernel(const uint8_t* __restrict__ pw, int64_t n, int64_t k, int64_t ld_pw,
                                                     T* __restrict__ out, int64_t ld_out, int vec) {
    const int64_t bytes_per_row = (k + 3) / 4;
    const int64_t total = n * bytes_per_row;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t row = i / bytes_per_row, j = i - row * bytes_per_row;
        const uint32_t b = pw[row * ld_pw + j];
        T v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = value_as<T>(code_value(b, e));
        T* dst = out + row * ld_out + 4 * j;
        if (vec && 4 * j + 3 < k) {
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(v);
            } else if constexpr (sizeof(T) == 2) {
                *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(v);
            } else {
                *reinterpret_cast<uint32_t*>(dst) = *reinterpret_cast<const uint32_t*>(v);
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (4 * j + e < k) dst[e] = v[e];
        }
    }
}

// Fast path of unpack_kernel: a lane expands 16 / sizeof(T) / 4 packed bytes (f32: 1, f16 / bf16: 2, int8: 4) into ONE
// 16-byte store, so a wave writes 1 KiB of consecutive bytes; 32-bit index arithmetic when the matrix allows it.
// Requires k % (4 * bytes per lane) == 0, packed rows aligned to the bytes per lane, 16-byte aligned output rows.
template <typename T>
__global__ void __launch_bounds__(256) unpack_fast_kernel(const uint8_t* __restrict__ pw, int64_t n, int64_t k, int64_t ld_pw,
                                                          T* __restrict__ out, int64_t ld_out) {
    constexpr int PB = 4 / (int)sizeof(T);   // packed bytes per lane
    constexpr int NV = 4 * PB;               // values per lane = one 16-byte store
    const int64_t gpr = k / NV, total = n * gpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t row, g;
        if (total <= 0x7fffffff) { row = (uint32_t)i / (uint32_t)gpr; g = (uint32_t)i - (uint32_t)row * (uint32_t)gpr; }
        else { row = i / gpr; g = i - row * gpr; }
        const uint8_t* src = pw + row * ld_pw + PB * g;
        uint32_t d;
        if constexpr (PB == 1) d = *src;
        else if constexpr (PB == 2) d = *reinterpret_cast<const uint16_t*>(src);
        else d = *reinterpret_cast<const uint32_t*>(src);
        T v[NV];
#pragma unroll
        for (int e = 0; e < NV; ++e) v[e] = value_as<T>((int)((d >> (2 * e)) & 3u) - 1);
        *reinterpret_cast<uint4*>(out + row * ld_out + NV * g) = *reinterpret_cast<const uint4*>(v);
    }
}

// planar layouts: S = ceil(len/4); byte j holds elements j, S+j, 2S+j, 3S+j
template <int FORMAT>
__global__ void __launch_bounds__(256) pack_planar_kernel(const int8_t* __restrict__ t, int64_t len, uint8_t* __restrict__ pw) {
    const int64_t s = (len + 3) / 4;
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < s; j += (int64_t)gridDim.x * blockDim.x) {
        uint32_t b = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t idx = i * s + j;
            const int v = idx < len ? (int)t[idx] : 0;
            uint32_t code;
            if constexpr (FORMAT == TVL_PLANAR_A) code = (uint32_t)(v == 1) ^ ((uint32_t)(v == -1) << 1);
            else code = (uint32_t)(v + 1) & 0xFFu;  // the reference ORs the whole shifted byte (uint8 arithmetic)
            b |= (code << (2 * i));
        }
        pw[j] = (uint8_t)b;
    }
}

// Fast path of pack_planar_kernel: a lane packs 16 consecutive bytes from four 16-byte loads (one per plane).
// Requires len == 4 S, S % 16 == 0 and 16-byte aligned buffers.
template <int FORMAT>
__global__ void __launch_bounds__(256) pack_planar_fast_kernel(const int8_t* __restrict__ t, int64_t s, uint8_t* __restrict__ pw) {
    const int64_t groups = s / 16;
    for (int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x; g < groups; g += (int64_t)gridDim.x * 256) {
        nt_u32x4 pl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) pl[i] = __builtin_nontemporal_load(reinterpret_cast<const nt_u32x4*>(t + i * s + 16 * g));
        nt_u32x4 o;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            uint32_t word = 0;
#pragma unroll
            for (int by = 0; by < 4; ++by) {
                uint32_t b = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int v = (int)(int8_t)((pl[i][d] >> (8 * by)) & 0xFFu);
                    uint32_t code;
                    if constexpr (FORMAT == TVL_PLANAR_A) code = (uint32_t)(v == 1) ^ ((uint32_t)(v == -1) << 1);
                    else code = (uint32_t)(v + 1) & 0xFFu;  // the reference ORs the whole shifted byte (uint8 arithmetic)
                    b |= (code << (2 * i));
                }
                word |= (b & 0xFFu) << (8 * by);
            }
            o[d] = word;
        }
        *reinterpret_cast<nt_u32x4*>(pw + 16 * g) = o;
    }
}

template <typename T, int FORMAT>
__global__ void __launch_bounds__(256) unpack_planar_kernel(const uint8_t* __restrict__ pw, int64_t s, T* __restrict__ out) {
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < s; j += (int64_t)gridDim.x * blockDim.x) {
        const uint32_t b = pw[j];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t c = (b >> (2 * i)) & 3u;
            int v;
            if constexpr (FORMAT == TVL_PLANAR_A) v = (int)(c & 1u) - (int)(c >> 1);
            else v = (int)c - 1;
            out[i * s + j] = value_as<T>(v);
        }
    }
}

template <typename TI, typename TO, int MODE>
__global__ void __launch_bounds__(256) quantize_dense_kernel(const TI* __restrict__ w, int64_t numel, float thr, TO* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x) {
        const int v = (int)quantize_code<MODE>(to_float(w[i]), thr) - 1;
        out[i] = value_as<TO>(v);
    }
}

__global__ void __launch_bounds__(256) zero_out_kernel(const int32_t* __restrict__ in, int64_t numel, int32_t* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (int64_t)gridDim.x * blockDim.x)
        out[i] = (i == 0) ? in[0] : 0;
}

__global__ void __launch_bounds__(256) copy_u8_kernel(const uint8_t* __restrict__ in, int64_t numel, uint8_t* __restrict__ out) {
    const int64_t nvec = numel / 16;
    const bool vec = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, nth = (int64_t)gridDim.x * blockDim.x;
    if (vec) {
        for (int64_t i = tid; i < nvec; i += nth)
            reinterpret_cast<uint4*>(out)[i] = reinterpret_cast<const uint4*>(in)[i];
        for (int64_t i = nvec * 16 + tid; i < numel; i += nth) out[i] = in[i];
    } else {
        for (int64_t i = tid; i < numel; i += nth) out[i] = in[i];
    }
}

// ---- launchers -------------------------------------------------------------------------------
static inline int grid_for(int64_t work_items, int per_block) {
    int64_t g = ceil_div(work_items, per_block);
    if (g < 1) g = 1;
    if (g > 256 * 8) g = 256 * 8;  // 8 blocks per CU, grid-stride beyond (16-64 per CU measured: no faster)
    return (int)g;
}

template <typename T>
static int launch_quantize_pack_t(const T* w, int64_t n, int64_t k, int64_t ld_w, int mode, float thr, uint8_t* pw,
                                  int64_t ld_pw, hipStream_t st) {
    const bool fast = (k % 16 == 0) && aligned(w, 4 * sizeof(T)) && (ld_w % 4 == 0) && aligned(pw, 4) && (ld_pw % 4 == 0);
#define TVL_QP(MODE)                                                                                             \
    if (fast) {                                                                                                  \
        const int64_t chunks = n * ceil_div(k, 1024);                                                            \
        hipLaunchKernelGGL((quantize_pack_fast_kernel<T, MODE>), dim3(grid_for(chunks, 4)), dim3(256), 0, st, w, n, k, ld_w, \
                           thr, pw, ld_pw);                                                                      \
    } else {                                                                                                     \
        hipLaunchKernelGGL((quantize_pack_generic_kernel<T, MODE>), dim3(grid_for(n * ceil_div(k, 4), 256)), dim3(256), 0, st, \
                           w, n, k, ld_w, thr, pw, ld_pw);                                                       \
    }
    switch (mode) {
        case TVL_Q_TERNARY: TVL_QP(TVL_Q_TERNARY) break;
        case TVL_Q_SIGN: TVL_QP(TVL_Q_SIGN) break;
        case TVL_Q_STEP: TVL_QP(TVL_Q_STEP) break;
        default: set_error("unknown quantizer mode %d", mode); return TVL_ERR_INVALID;
    }
#undef TVL_QP
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

template <typename T>
static int launch_pack_batch_t(const PackBatch& b, int mode, float thr, hipStream_t st) {
    const unsigned grid = grid_for(b.chunk_begin[b.count], 4);
    switch (mode) {
        case TVL_Q_TERNARY: hipLaunchKernelGGL((quantize_pack_batch_kernel<T, TVL_Q_TERNARY>), dim3(grid), dim3(256), 0, st, b, thr); break;
        case TVL_Q_SIGN: hipLaunchKernelGGL((quantize_pack_batch_kernel<T, TVL_Q_SIGN>), dim3(grid), dim3(256), 0, st, b, thr); break;
        case TVL_Q_STEP: hipLaunchKernelGGL((quantize_pack_batch_kernel<T, TVL_Q_STEP>), dim3(grid), dim3(256), 0, st, b, thr); break;
        default: set_error("unknown quantizer mode %d", mode); return TVL_ERR_INVALID;
    }
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

int launch_quantize_pack(const void* w, int w_dtype, int64_t n, int64_t k, int64_t ld_w, int mode, float thr, uint8_t* pw, int64_t ld_pw,
                         hipStream_t st);

// items: HOST array.  Matrices the fast kernel can take (k % 16 == 0, 16-byte loadable rows, dword-storable packed rows) go out in
// launches of up to PACK_BATCH_MAX matrices each; the others one by one through the single-matrix path.
int launch_quantize_pack_batch(const tvl_pack_item* items, int64_t count, int w_dtype, int mode, float thr, hipStream_t st) {
    const size_t es = dtype_size(w_dtype);
    PackBatch b;
    b.count = 0;
    b.chunk_begin[0] = 0;
    auto flush = [&]() -> int {
        if (b.count == 0) return TVL_OK;
        int rc;
        switch (w_dtype) {
            case TVL_F32: rc = launch_pack_batch_t<float>(b, mode, thr, st); break;
            case TVL_F16: rc = launch_pack_batch_t<f16>(b, mode, thr, st); break;
            default: rc = launch_pack_batch_t<bf16_t>(b, mode, thr, st); break;
        }
        b.count = 0;
        return rc;
    };
    for (int64_t i = 0; i < count; ++i) {
        const tvl_pack_item& it = items[i];
        if (it.n == 0 || it.k == 0) continue;
        const int64_t ld_w = it.ld_w ? it.ld_w : it.k, ld_pw = it.ld_pw ? it.ld_pw : ceil_div(it.k, 4);
        const bool fast = (it.k % 16 == 0) && aligned(it.w, 4 * es) && (ld_w % 4 == 0) && aligned(it.pw, 4) && (ld_pw % 4 == 0);
        if (!fast) {
            const int rc = launch_quantize_pack(it.w, w_dtype, it.n, it.k, ld_w, mode, thr, it.pw, ld_pw, st);
            if (rc != TVL_OK) return rc;
            continue;
        }
        const int j = b.count;
        b.w[j] = it.w; b.pw[j] = it.pw; b.k[j] = it.k; b.ld_w[j] = ld_w; b.ld_pw[j] = ld_pw;
        b.chunk_begin[j + 1] = b.chunk_begin[j] + it.n * ceil_div(it.k, 1024);
        if (++b.count == PACK_BATCH_MAX) {
            const int rc = flush();
            if (rc != TVL_OK) return rc;
        }
    }
    return flush();
}

int launch_quantize_pack(const void* w, int w_dtype, int64_t n, int64_t k, int64_t ld_w, int mode, float thr, uint8_t* pw,
                         int64_t ld_pw, hipStream_t st) {
    switch (w_dtype) {
        case TVL_F32: return launch_quantize_pack_t((const float*)w, n, k, ld_w, mode, thr, pw, ld_pw, st);
        case TVL_F16: return launch_quantize_pack_t((const f16*)w, n, k, ld_w, mode, thr, pw, ld_pw, st);
        case TVL_BF16: return launch_quantize_pack_t((const bf16_t*)w, n, k, ld_w, mode, thr, pw, ld_pw, st);
        default: set_error("quantize_pack: unsupported weight dtype %d", w_dtype); return TVL_ERR_INVALID;
    }
}

int launch_pack_i8(const int8_t* t, int64_t n, int64_t k, int64_t ld_t, uint8_t* pw, int64_t ld_pw, hipStream_t st) {
    if (k % 16 == 0 && aligned(t, 16) && ld_t % 16 == 0 && aligned(pw, 4) && ld_pw % 4 == 0) {
        hipLaunchKernelGGL(pack_i8_fast_kernel, dim3(grid_for(n * (k / 16), 256)), dim3(256), 0, st, t, n, k, ld_t, pw, ld_pw);
        TVL_LAUNCH_CHECK();
        return TVL_OK;
    }
    const int vec = aligned(t, 4) && (ld_t % 4 == 0);
    hipLaunchKernelGGL(pack_i8_kernel, dim3(grid_for(n * ceil_div(k, 4), 256)), dim3(256), 0, st, t, n, k, ld_t, pw, ld_pw, vec);
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

int launch_unpack(const uint8_t* pw, int64_t n, int64_t k, int64_t ld_pw, void* out, int out_dtype, int64_t ld_out,
                  hipStream_t st) {
    const dim3 grid(grid_for(n * ceil_div(k, 4), 256)), block(256);
    const size_t es = dtype_size(out_dtype);
    const int64_t pb = 4 / (int64_t)es;   // packed bytes per lane in the fast kernel
    if (k % (4 * pb) == 0 && aligned(pw, (size_t)pb) && ld_pw % pb == 0 && aligned(out, 16) && (ld_out * (int64_t)es) % 16 == 0) {
        const dim3 fgrid(grid_for(n * (k / (4 * pb)), 256));
        switch (out_dtype) {
            case TVL_F32: hipLaunchKernelGGL(unpack_fast_kernel<float>, fgrid, block, 0, st, pw, n, k, ld_pw, (float*)out, ld_out); break;
            case TVL_F16: hipLaunchKernelGGL(unpack_fast_kernel<f16>, fgrid, block, 0, st, pw, n, k, ld_pw, (f16*)out, ld_out); break;
            case TVL_BF16: hipLaunchKernelGGL(unpack_fast_kernel<bf16_t>, fgrid, block, 0, st, pw, n, k, ld_pw, (bf16_t*)out, ld_out); break;
            case TVL_I8: hipLaunchKernelGGL(unpack_fast_kernel<int8_t>, fgrid, block, 0, st, pw, n, k, ld_pw, (int8_t*)out, ld_out); break;
            default: set_error("unpack: unsupported output dtype %d", out_dtype); return TVL_ERR_INVALID;
        }
        TVL_LAUNCH_CHECK();
        return TVL_OK;
    }
    const int vec = aligned(out, 4 * es) && (ld_out % 4 == 0);
    switch (out_dtype) {
        case TVL_F32: hipLaunchKernelGGL(unpack_kernel<float>, grid, block, 0, st, pw, n, k, ld_pw, (float*)out, ld_out, vec); break;
        case TVL_F16: hipLaunchKernelGGL(unpack_kernel<f16>, grid, block, 0, st, pw, n, k, ld_pw, (f16*)out, ld_out, vec); break;
        case TVL_BF16: hipLaunchKernelGGL(unpack_kernel<bf16_t>, grid, block, 0, st, pw, n, k, ld_pw, (bf16_t*)out, ld_out, vec); break;
        case TVL_I8: hipLaunchKernelGGL(unpack_kernel<int8_t>, grid, block, 0, st, pw, n, k, ld_pw, (int8_t*)out, ld_out, vec); break;
        default: set_error("unpack: unsupported output dtype %d", out_dtype); return TVL_ERR_INVALID;
    }
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

int launch_pack_planar(const int8_t* t, int64_t len, int format, uint8_t* pw, hipStream_t st) {
    const int64_t s4 = ceil_div(len, 4);
    if (format != TVL_PLANAR_A && format != TVL_PLANAR_B) { set_error("unknown planar format %d", format); return TVL_ERR_INVALID; }
    if (len == 4 * s4 && s4 % 16 == 0 && aligned(t, 16) && aligned(pw, 16)) {
        const dim3 fgrid(grid_for(s4 / 16, 256));
        if (format == TVL_PLANAR_A) hipLaunchKernelGGL(pack_planar_fast_kernel<TVL_PLANAR_A>, fgrid, dim3(256), 0, st, t, s4, pw);
        else hipLaunchKernelGGL(pack_planar_fast_kernel<TVL_PLANAR_B>, fgrid, dim3(256), 0, st, t, s4, pw);
        TVL_LAUNCH_CHECK();
        return TVL_OK;
    }
    const dim3 grid(grid_for(ceil_div(len, 4), 256)), block(256);
    if (format == TVL_PLANAR_A) hipLaunchKernelGGL(pack_planar_kernel<TVL_PLANAR_A>, grid, block, 0, st, t, len, pw);
    else if (format == TVL_PLANAR_B) hipLaunchKernelGGL(pack_planar_kernel<TVL_PLANAR_B>, grid, block, 0, st, t, len, pw);
    else { set_error("unknown planar format %d", format); return TVL_ERR_INVALID; }
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

template <int FORMAT>
static int launch_unpack_planar_f(const uint8_t* pw, int64_t s, void* out, int out_dtype, hipStream_t st) {
    const dim3 grid(grid_for(s, 256)), block(256);
    switch (out_dtype) {
        case TVL_F32: hipLaunchKernelGGL((unpack_planar_kernel<float, FORMAT>), grid, block, 0, st, pw, s, (float*)out); break;
        case TVL_F16: hipLaunchKernelGGL((unpack_planar_kernel<f16, FORMAT>), grid, block, 0, st, pw, s, (f16*)out); break;
        case TVL_BF16: hipLaunchKernelGGL((unpack_planar_kernel<bf16_t, FORMAT>), grid, block, 0, st, pw, s, (bf16_t*)out); break;
        case TVL_I8: hipLaunchKernelGGL((unpack_planar_kernel<int8_t, FORMAT>), grid, block, 0, st, pw, s, (int8_t*)out); break;
        default: set_error("unpack_planar: unsupported output dtype %d", out_dtype); return TVL_ERR_INVALID;
    }
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

int launch_unpack_planar(const uint8_t* pw, int64_t s, int format, void* out, int out_dtype, hipStream_t st) {
    if (format == TVL_PLANAR_A) return launch_unpack_planar_f<TVL_PLANAR_A>(pw, s, out, out_dtype, st);
    if (format == TVL_PLANAR_B) return launch_unpack_planar_f<TVL_PLANAR_B>(pw, s, out, out_dtype, st);
    set_error("unknown planar format %d", format);
    return TVL_ERR_INVALID;
}

template <typename TI, typename TO>
static int launch_quantize_dense_t(const TI* w, int64_t numel, int mode, float thr, TO* out, hipStream_t st) {
    const dim3 grid(grid_for(numel, 256)), block(256);
    switch (mode) {
        case TVL_Q_TERNARY: hipLaunchKernelGGL((quantize_dense_kernel<TI, TO, TVL_Q_TERNARY>), grid, block, 0, st, w, numel, thr, out); break;
        case TVL_Q_SIGN: hipLaunchKernelGGL((quantize_dense_kernel<TI, TO, TVL_Q_SIGN>), grid, block, 0, st, w, numel, thr, out); break;
        case TVL_Q_STEP: hipLaunchKernelGGL((quantize_dense_kernel<TI, TO, TVL_Q_STEP>), grid, block, 0, st, w, numel, thr, out); break;
        default: set_error("unknown quantizer mode %d", mode); return TVL_ERR_INVALID;
    }
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

int launch_quantize_dense(const void* w, int w_dtype, int64_t numel, int mode, float thr, void* out, int out_dtype, hipStream_t st) {
    if (out_dtype == TVL_I8) {
        switch (w_dtype) {
            case TVL_F32: return launch_quantize_dense_t((const float*)w, numel, mode, thr, (int8_t*)out, st);
            case TVL_F16: return launch_quantize_dense_t((const f16*)w, numel, mode, thr, (int8_t*)out, st);
            case TVL_BF16: return launch_quantize_dense_t((const bf16_t*)w, numel, mode, thr, (int8_t*)out, st);
        }
    } else if (out_dtype == w_dtype) {
        switch (w_dtype) {
            case TVL_F32: return launch_quantize_dense_t((const float*)w, numel, mode, thr, (float*)out, st);
            case TVL_F16: return launch_quantize_dense_t((const f16*)w, numel, mode, thr, (f16*)out, st);
            case TVL_BF16: return launch_quantize_dense_t((const bf16_t*)w, numel, mode, thr, (bf16_t*)out, st);
        }
    }
    set_error("quantize: unsupported dtype combination in=%d out=%d (out must be the input type or int8)", w_dtype, out_dtype);
    return TVL_ERR_INVALID;
}

int launch_zero_out(const int32_t* in, int64_t numel, int32_t* out, hipStream_t st) {
    hipLaunchKernelGGL(zero_out_kernel, dim3(grid_for(numel, 256)), dim3(256), 0, st, in, numel, out);
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

int launch_copy_u8(const uint8_t* in, int64_t numel, uint8_t* out, hipStream_t st) {
    hipLaunchKernelGGL(copy_u8_kernel, dim3(grid_for(ceil_div(numel, 16), 256)), dim3(256), 0, st, in, numel, out);
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

// ---- zero-padded copies along k (for shapes the matrix-core kernels cannot take as they are) ----------------------------------
// out[m, kp] = x[m, 0..k) followed by zeros; a lane moves 16 bytes (VEC: 16-byte aligned source rows) or fills element by element.
template <typename T, bool VEC>
__global__ void __launch_bounds__(256) pad_rows_kernel(const T* __restrict__ x, int64_t m, int64_t k, int64_t ld_x, T* __restrict__ out, int64_t kp) {
    constexpr int PER = 16 / (int)sizeof(T);
    const int64_t gpr = kp / PER, total = m * gpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t row, g;
        if (total <= 0x7fffffff) { row = (uint32_t)i / (uint32_t)gpr; g = (uint32_t)i - (uint32_t)row * (uint32_t)gpr; }
        else { row = i / gpr; g = i - row * gpr; }
        const int64_t c0 = g * PER;
        const T* src = x + row * ld_x + c0;
        T v[PER];
        if (VEC && c0 + PER <= k) {
            *reinterpret_cast<uint4*>(v) = *reinterpret_cast<const uint4*>(src);
        } else {
#pragma unroll
            for (int e = 0; e < PER; ++e) v[e] = c0 + e < k ? src[e] : from_float<T>(0.0f);
        }
        *reinterpret_cast<uint4*>(out + row * kp + c0) = *reinterpret_cast<const uint4*>(v);
    }
}

// out[n, bp] = the packed rows with every code at k and beyond set to 1 (weight 0); a lane writes one dword
__global__ void __launch_bounds__(256) pad_packed_kernel(const uint8_t* __restrict__ pw, int64_t n, int64_t k, int64_t ld_pw,
                                                         uint8_t* __restrict__ out, int64_t bp) {
    const int64_t kb = (k + 3) / 4, gpr = bp / 4, total = n * gpr;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        int64_t row, g;
        if (total <= 0x7fffffff) { row = (uint32_t)i / (uint32_t)gpr; g = (uint32_t)i - (uint32_t)row * (uint32_t)gpr; }
        else { row = i / gpr; g = i - row * gpr; }
        uint32_t d = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int64_t j = 4 * g + e;
            uint32_t b = j < kb ? pw[row * ld_pw + j] : 0x55u;
            if (j == kb - 1 && (k & 3)) {
                const uint32_t keep = (1u << (2 * (k & 3))) - 1u;
                b = (b & keep) | (0x55u & ~keep);
            }
            d |= b << (8 * e);
        }
        *reinterpret_cast<uint32_t*>(out + row * bp + 4 * g) = d;
    }
}

int launch_pad_rows(const void* x, int x_dtype, int64_t m, int64_t k, int64_t ld_x, void* out, int64_t kp, hipStream_t st) {
    const size_t es = dtype_size(x_dtype);
    const bool vec = aligned(x, 16) && (ld_x * (int64_t)es) % 16 == 0;
    const dim3 grid(grid_for(m * (kp / (16 / (int64_t)es)), 256)), block(256);
    if (x_dtype == TVL_F32) {
        if (vec) hipLaunchKernelGGL((pad_rows_kernel<float, true>), grid, block, 0, st, (const float*)x, m, k, ld_x, (float*)out, kp);
        else hipLaunchKernelGGL((pad_rows_kernel<float, false>), grid, block, 0, st, (const float*)x, m, k, ld_x, (float*)out, kp);
    } else {   // f16 and bf16 are both moved as 16-bit patterns
        if (vec) hipLaunchKernelGGL((pad_rows_kernel<f16, true>), grid, block, 0, st, (const f16*)x, m, k, ld_x, (f16*)out, kp);
        else hipLaunchKernelGGL((pad_rows_kernel<f16, false>), grid, block, 0, st, (const f16*)x, m, k, ld_x, (f16*)out, kp);
    }
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

int launch_pad_packed(const uint8_t* pw, int64_t n, int64_t k, int64_t ld_pw, uint8_t* out, int64_t bp, hipStream_t st) {
    hipLaunchKernelGGL(pad_packed_kernel, dim3(grid_for(n * (bp / 4), 256)), dim3(256), 0, st, pw, n, k, ld_pw, out, bp);
    TVL_LAUNCH_CHECK();
    return TVL_OK;
}

}  // namespace tvl

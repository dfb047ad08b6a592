// Read-bandwidth probe: how fast can waves pull a packed weight matrix (n rows of kb bytes) out of HBM with a given access shape
// and a given number of 16-byte loads in flight per lane?  Shapes: a wave instruction covers R rows x (1024 / R) contiguous bytes.
//   usage: bw_probe n kb [matrices]     (prints one line per shape x depth; with several matrices the launches rotate over them)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// wave owns ROWS rows x `span` bytes (a k range); per instruction it reads ROWS x (1024/ROWS) bytes; U instructions in flight
template <int ROWS, int U>
__global__ __launch_bounds__(256) void probe(const uint8_t* __restrict__ w, int64_t n, int64_t kb, int64_t span, uint32_t* out) {
    constexpr int SEG = 1024 / ROWS;            // bytes per row per instruction
    constexpr int LPR = SEG / 16;               // lanes per row
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int64_t spans = kb / span;            // k ranges per row group
    const int64_t groups = n / ROWS;
    if (wave >= spans * groups) return;
    const int64_t grp = wave / spans, sp = wave % spans;
    const int r = lane / LPR, q = lane % LPR;
    const uint8_t* p = w + (grp * ROWS + r) * kb + sp * span + q * 16;
    uint4 acc = {0, 0, 0, 0};
    const int iters = (int)(span / SEG);
    for (int i = 0; i < iters; i += U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load((const u32x4*)(p + (int64_t)(i + u) * SEG));
#pragma unroll
        for (int u = 0; u < U; ++u) { acc.x ^= v[u].x; acc.y += v[u].y; acc.z ^= v[u].z; acc.w += v[u].w; }
    }
    uint32_t s = acc.x ^ acc.y ^ acc.z ^ acc.w;
    if (s == 0x12345678u) out[wave] = s;        // keeps the loads alive, practically never stores
}

static int g_nbuf = 1;
template <int ROWS, int U>
static void run(const uint8_t* w, int64_t n, int64_t kb, int64_t span, uint32_t* out, int wpb) {
    const int64_t waves = (kb / span) * (n / ROWS);
    const int blocks = (int)((waves + wpb - 1) / wpb);
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe<ROWS, U>), dim3(blocks), dim3(64 * wpb), 0, 0, w + (int64_t)(i % g_nbuf) * n * kb, n, kb, span, out);
    CHECK(hipDeviceSynchronize());
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(a, 0));
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((probe<ROWS, U>), dim3(blocks), dim3(64 * wpb), 0, 0, w + (int64_t)(i % g_nbuf) * n * kb, n, kb, span, out);
        CHECK(hipEventRecord(b, 0)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms / 20 < best) best = ms / 20;
    }
    printf("rows/instr %2d  in-flight %d  span %6lld B  waves %7lld (x%d per block)  %7.2f us  %5.2f TB/s\n", ROWS, U, (long long)span,
           (long long)waves, wpb, best * 1e3, (double)n * kb / (best * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 16384, kb = argc > 2 ? atoll(argv[2]) : 4096;
    g_nbuf = argc > 3 ? atoi(argv[3]) : 1;   // rotate over this many matrices: every launch reads memory that is in no cache
    uint8_t* w; uint32_t* out;
    CHECK(hipMalloc(&w, n * kb * g_nbuf)); CHECK(hipMemset(w, 0x55, n * kb * g_nbuf));
    CHECK(hipMalloc(&out, 64 << 20));
    // a second buffer of 512 MiB is swept between measurements?  no: 20 back-to-back launches over >= 64 MiB; the 256 MiB Infinity
    // Cache holds the matrix when it is smaller than that, which is the decode situation too (weights re-read every token).
    for (int64_t span : {(int64_t)512, (int64_t)1024, (int64_t)2048, kb}) {
        if (span > kb) continue;
        printf("-- n %lld, row bytes %lld, span %lld\n", (long long)n, (long long)kb, (long long)span);
        for (int wpb : {4}) {
            if (span >= 1024 * 2) { run<1, 2>(w, n, kb, span, out, wpb); }
            if (span >= 1024 * 4) { run<1, 4>(w, n, kb, span, out, wpb); }
            if (span >= 256 * 2) { run<4, 2>(w, n, kb, span, out, wpb); }
            if (span >= 256 * 4) { run<4, 4>(w, n, kb, span, out, wpb); }
            if (span >= 256 * 8) { run<4, 8>(w, n, kb, span, out, wpb); }
            if (span >= 128 * 4) { run<8, 4>(w, n, kb, span, out, wpb); }
            if (span >= 128 * 8) { run<8, 8>(w, n, kb, span, out, wpb); }
            run<16, 2>(w, n, kb, span, out, wpb);
            run<16, 4>(w, n, kb, span, out, wpb);
            run<16, 8>(w, n, kb, span, out, wpb);
            if (span >= 64 * 16) run<16, 16>(w, n, kb, span, out, wpb);
        }
    }
    return 0;
}

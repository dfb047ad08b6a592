/*
 * CPU restatement (C, OpenMP) of the packed ternary matmul - TEST INFRASTRUCTURE ONLY.
 *
 * Used as the checker at sizes where the numpy oracle is slow and as the "port" CPU baseline that
 * the benchmark times on the GPU box's host cores.  Nothing in the product links or calls it.
 * Parity: pinned by tests/test_oracle_c.py against the numpy oracle (oracle/ternary_oracle.py),
 * which in turn is pinned by the golden vectors generated from the reference's torch code.
 *
 * Reference semantics restated (paths relative to the reference repository):
 *   - decode  ((b >> 2i) & 3) - 1            impl/tf/moondream/ternary.py:507-520
 *   - y = clip((x @ T^T) * scale + bias)     binary/modules.py:101-109 (two roundings, no fma)
 *   - select-and-sum form  sum_{T>0} x - sum_{T<0} x   matrix_mul_tests.py:29-37
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC ternary_oracle.c -o libternary_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline float clipf(float v, float c) { return v < -c ? -c : (v > c ? c : v); }

int tvo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* unpack one packed row (format C) to k floats */
static void unpack_row(const uint8_t* pw, int64_t k, float* out) {
    for (int64_t j = 0; j < k; ++j) out[j] = (float)((int)((pw[j >> 2] >> (2 * (j & 3))) & 3) - 1);
}

/* Dense form: unpack the weights to float {-1,0,+1,(+2)} once, then a blocked float32 matmul
 * (the arithmetic of binary/modules.py:104, F.linear with the quantized weights).
 * x [m,k] float32, pw [n, ld_pw] uint8, y [m,n] float32.  threads <= 0: OpenMP default. */
int tvo_ternary_matmul_dense_f32(const float* x, int64_t m, int64_t k, const uint8_t* pw, int64_t ld_pw, int64_t n,
                                 const float* scale, const float* bias, float clip, float* y, int threads) {
    float* w = (float*)malloc((size_t)n * (size_t)k * sizeof(float));
    if (!w) return 1;
    const float c = fabsf(clip);
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) unpack_row(pw + r * ld_pw, k, w + r * k);
    enum { BM = 8, BN = 8 };
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t i0 = 0; i0 < m; i0 += BM) {
        const int64_t im = (i0 + BM <= m) ? BM : (m - i0);
        for (int64_t j0 = 0; j0 < n; j0 += BN) {
            const int64_t jn = (j0 + BN <= n) ? BN : (n - j0);
            float acc[BM][BN];
            memset(acc, 0, sizeof(acc));
            for (int64_t i = 0; i < im; ++i) {
                const float* xr = x + (i0 + i) * k;
                for (int64_t j = 0; j < jn; ++j) {
                    const float* wr = w + (j0 + j) * k;
                    float s = 0.0f;
#pragma omp simd reduction(+ : s)
                    for (int64_t kk = 0; kk < k; ++kk) s += xr[kk] * wr[kk];
                    acc[i][j] = s;
                }
            }
            for (int64_t i = 0; i < im; ++i)
                for (int64_t j = 0; j < jn; ++j) {
                    volatile float t = acc[i][j] * scale[j0 + j]; /* volatile: keep the two roundings */
                    float v = bias ? t + bias[j0 + j] : t;
                    y[(i0 + i) * n + (j0 + j)] = clipf(v, c);
                }
        }
    }
    free(w);
    return 0;
}

/* Select-and-sum form (multiplication-free): per output channel the indices of its +1 and -1
 * weights are listed once, then every activation row is gathered and summed over them.
 * Codes 3 (+2) are not representable in this form and are treated as +1 twice. */
int tvo_ternary_matmul_select_f32(const float* x, int64_t m, int64_t k, const uint8_t* pw, int64_t ld_pw, int64_t n,
                                  const float* scale, const float* bias, float clip, float* y, int threads) {
    const float c = fabsf(clip);
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    int err = 0;
#pragma omp parallel
    {
        int32_t* pos = (int32_t*)malloc((size_t)(2 * k) * sizeof(int32_t));
        int32_t* neg = (int32_t*)malloc((size_t)k * sizeof(int32_t));
        if (!pos || !neg) {
#pragma omp atomic write
            err = 1;
        } else {
#pragma omp for schedule(dynamic, 4)
            for (int64_t j = 0; j < n; ++j) {
                int64_t np = 0, nn = 0;
                const uint8_t* row = pw + j * ld_pw;
                for (int64_t kk = 0; kk < k; ++kk) {
                    const int v = (int)((row[kk >> 2] >> (2 * (kk & 3))) & 3) - 1;
                    if (v > 0) { pos[np++] = (int32_t)kk; if (v == 2) pos[np++] = (int32_t)kk; }
                    else if (v < 0) neg[nn++] = (int32_t)kk;
                }
                for (int64_t i = 0; i < m; ++i) {
                    const float* xr = x + i * k;
                    float sp = 0.0f, sn = 0.0f;
                    for (int64_t t = 0; t < np; ++t) sp += xr[pos[t]];
                    for (int64_t t = 0; t < nn; ++t) sn += xr[neg[t]];
                    volatile float t = (sp - sn) * scale[j];
                    float v = bias ? t + bias[j] : t;
                    y[i * n + j] = clipf(v, c);
                }
            }
        }
        free(pos);
        free(neg);
    }
    return err;
}

/* quantize + pack (format C), for checking the GPU pack kernel at full size on the CPU */
void tvo_ternarize_pack_f32(const float* w, int64_t n, int64_t k, float thr, uint8_t* pw) {
    const int64_t nb = (k + 3) / 4;
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r)
        for (int64_t j = 0; j < nb; ++j) {
            unsigned b = 0;
            for (int e = 0; e < 4; ++e) {
                const int64_t kk = 4 * j + e;
                unsigned code = 1;
                if (kk < k) {
                    const float v = w[r * k + kk];
                    code = 1u + (unsigned)(v >= thr) - (unsigned)(v <= -thr);
                }
                b |= code << (2 * e);
            }
            pw[r * nb + j] = (uint8_t)b;
        }
}

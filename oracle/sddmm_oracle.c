/*
 * oracle/sddmm_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the SDDMM hot path of CX9898/BSMR-SDDMM.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this file;
 * the product (bsmr-sddmm_amd/) never links, imports or calls it.
 *
 * PARITY UNPINNED for the values computed here: the reference ships no golden
 * output vectors (its published logs hold GFLOP/s only) and its own host code cannot
 * be built in this image without writing stand-ins (cuda_fp16.h needs the absent
 * <nv/target>, cudaErrorCheck.cuh needs the absent <curand.h>, linking needs
 * libcudart and CUDA Thrust).  So every function below is a restatement that cites
 * the reference lines it follows.  (The host pipeline is a different story: it is
 * pinned by the reference's published logs, see oracle/clustering_oracle.c and
 * tests/test_reference_logs.py.)
 *
 * Conventions (reference src/main.cu:23-27): A is M x K row-major, B is K x N
 * column-major with ld = K (so column j is the K contiguous floats B[j*K ..]),
 * P is fp32 in S's CSR order.  S's numeric values are never used
 * (reference src/host.cpp:62-73).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------- */
/* a12: sddmm_cpu, reference src/host.cpp:44-76 (CSR overload).               */
/* OpenMP over rows; per stored entry a strictly sequential fp32 k-loop       */
/* `val += a*b` through the accessors of src/Matrix.cpp:199-221               */
/* (A row-major: values[row*ld+k]; B col-major: values[col*ld+k]).            */
/* Built with -ffp-contract=off: the reference's host compiler targets plain  */
/* x86-64 (no FMA), so every product and every add rounds separately.         */
/* ------------------------------------------------------------------------- */
void oracle_sddmm_cpu(uint32_t M, uint32_t N, uint32_t K,
                      const uint32_t *rowOffsets, const uint32_t *colIndices,
                      const float *A, const float *B, float *P)
{
    (void)N;
#pragma omp parallel for schedule(static)
    for (int64_t row = 0; row < (int64_t)M; ++row) {
        for (uint32_t e = rowOffsets[row]; e < rowOffsets[row + 1]; ++e) {
            const size_t col = colIndices[e];
            const float *a = A + (size_t)row * K;
            const float *b = B + col * K;
            float val = 0.0f;
            for (uint32_t k = 0; k < K; ++k) {
                val += a[k] * b[k];
            }
            P[e] = val;
        }
    }
}

/* COO overload, reference src/host.cpp:93-124 (the `val *= S` line is        */
/* commented out there, :122).                                                */
void oracle_sddmm_cpu_coo(uint32_t K, uint64_t nnz,
                          const uint32_t *rowIndices, const uint32_t *colIndices,
                          const float *A, const float *B, float *P)
{
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < (int64_t)nnz; ++e) {
        const float *a = A + (size_t)rowIndices[e] * K;
        const float *b = B + (size_t)colIndices[e] * K;
        float val = 0.0f;
        for (uint32_t k = 0; k < K; ++k) {
            val += a[k] * b[k];
        }
        P[e] = val;
    }
}

/* fp64 accumulation of the same sum: error yardstick, not a reference path.  */
void oracle_sddmm_f64(uint32_t M, uint32_t K,
                      const uint32_t *rowOffsets, const uint32_t *colIndices,
                      const float *A, const float *B, double *P)
{
#pragma omp parallel for schedule(static)
    for (int64_t row = 0; row < (int64_t)M; ++row) {
        for (uint32_t e = rowOffsets[row]; e < rowOffsets[row + 1]; ++e) {
            const float *a = A + (size_t)row * K;
            const float *b = B + (size_t)colIndices[e] * K;
            double val = 0.0;
            for (uint32_t k = 0; k < K; ++k) {
                val += (double)a[k] * (double)b[k];
            }
            P[e] = val;
        }
    }
}

/* ------------------------------------------------------------------------- */
/* checkData, reference include/checkData.hpp:14,21-30,44-79:                 */
/*   pass iff |a-b| < 1e-5  or  |a-b| / max(|a|,|b|,1e-3) < 1e-3              */
/* ------------------------------------------------------------------------- */
int oracle_check_one(float a, float b)
{
    const float ABS_EPSILON = 1e-5f;
    const float ERROR_THRESHOLD_EPSILON = 1e-3f;
    const float absDiff = fabsf(a - b);
    if (absDiff < ABS_EPSILON) return 1;
    float maxVal = fabsf(a) > fabsf(b) ? fabsf(a) : fabsf(b);
    if (maxVal < ERROR_THRESHOLD_EPSILON) maxVal = ERROR_THRESHOLD_EPSILON;
    return (absDiff / maxVal) < ERROR_THRESHOLD_EPSILON;
}

/* Returns the number of failing elements; first_bad gets the first index or -1. */
uint64_t oracle_check_data(uint64_t n, const float *x, const float *y, int64_t *first_bad)
{
    uint64_t errors = 0;
    int64_t first = -1;
    for (uint64_t i = 0; i < n; ++i) {
        if (!oracle_check_one(x[i], y[i])) {
            if (first < 0) first = (int64_t)i;
            ++errors;
        }
    }
    if (first_bad) *first_bad = first;
    return errors;
}

/* ------------------------------------------------------------------------- */
/* Operand roundings.                                                         */
/* TF32: reference include/TensorCoreConfig.cuh:58-66 + src/sddmmKernel.cu    */
/* :317-325 use wmma::__float_to_tf32 = cvt.rna.tf32.f32 (nearest, ties away  */
/* from zero, 10 explicit mantissa bits).  fp16 / bf16: round-to-nearest-even */
/* as v_cvt_f16_f32 / v_cvt_pk_bf16_f32 do on gfx950, returned widened back   */
/* to fp32 (what the MFMA multiplies).                                        */
/* ------------------------------------------------------------------------- */
static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

float oracle_round_tf32(float x)
{
    uint32_t u = f2u(x);
    if ((u & 0x7F800000u) == 0x7F800000u) return x; /* inf / nan */
    u = (u + 0x1000u) & 0xFFFFE000u;
    return u2f(u);
}

float oracle_round_bf16(float x)
{
    uint32_t u = f2u(x);
    if ((u & 0x7F800000u) == 0x7F800000u) return x;
    u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
    return u2f(u);
}

float oracle_round_fp16(float x)
{
    /* fp32 -> IEEE binary16 (RNE, subnormals kept, overflow -> inf) -> fp32 */
    const uint32_t u = f2u(x);
    const uint32_t sign = u & 0x80000000u;
    const uint32_t au = u & 0x7FFFFFFFu;
    if (au >= 0x7F800000u) return x;                       /* inf / nan      */
    if (au >= 0x477FF000u) return u2f(sign | 0x7F800000u); /* >= 65520 -> inf */
    if (au < 0x33000001u) return u2f(sign);                /* < 2^-25 -> 0   */
    const int e = (int)(au >> 23) - 127;
    int drop = 13;                 /* mantissa bits dropped for normals */
    if (e < -14) drop += (-14 - e); /* subnormal half */
    uint32_t mant = (au & 0x007FFFFFu) | 0x00800000u;
    uint32_t keep, rem, half;
    if (drop >= 25) { keep = 0; rem = 1; half = 2; }
    else {
        keep = mant >> drop;
        rem = mant & ((1u << drop) - 1u);
        half = 1u << (drop - 1);
    }
    if (rem > half || (rem == half && (keep & 1u))) ++keep;
    /* value = keep * 2^(e - 23 + drop) */
    const float r = ldexpf((float)keep, e - 23 + drop);
    return sign ? -r : r;
}

void oracle_round_array(int mode, uint64_t n, const float *in, float *out)
{
    /* mode 0 = copy, 1 = tf32 (rna), 2 = fp16 (rne), 3 = bf16 (rne) */
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < (int64_t)n; ++i) {
        float v = in[i];
        switch (mode) {
        case 1: v = oracle_round_tf32(v); break;
        case 2: v = oracle_round_fp16(v); break;
        case 3: v = oracle_round_bf16(v); break;
        default: break;
        }
        out[i] = v;
    }
}

/* ------------------------------------------------------------------------- */
/* Emulation of the numerics of the reference's live kernels (restatement of  */
/* SURVEY.md appendix B; the order inside one m16n16k8 MMA is hardware-       */
/* defined, so this is a model, not a bit-level twin):                        */
/*   dense part  (src/sddmmKernel.cu:274-330): TF32-rounded operands, fp32    */
/*               accumulate, k ascending;                                     */
/*   sparse part (src/sddmmKernel.cu:2053-2103): per 32-wide K step the even  */
/*               lane owns k in [0,16), the odd lane k in [16,32); each lane  */
/*               adds 8-term expressions `c0 += a0*b0 + ... + a7*b7`; the two */
/*               partials are added once at the end (__shfl_xor).             */
/* is_dense[e] != 0 selects the dense model for CSR entry e.                  */
/* ------------------------------------------------------------------------- */
void oracle_sddmm_ref_kernel_model(uint32_t M, uint32_t K,
                                   const uint32_t *rowOffsets, const uint32_t *colIndices,
                                   const uint8_t *is_dense,
                                   const float *A, const float *B, float *P)
{
#pragma omp parallel for schedule(static)
    for (int64_t row = 0; row < (int64_t)M; ++row) {
        for (uint32_t e = rowOffsets[row]; e < rowOffsets[row + 1]; ++e) {
            const float *a = A + (size_t)row * K;
            const float *b = B + (size_t)colIndices[e] * K;
            if (is_dense[e]) {
                float acc = 0.0f;
                for (uint32_t k = 0; k < K; ++k)
                    acc += oracle_round_tf32(a[k]) * oracle_round_tf32(b[k]);
                P[e] = acc;
            } else {
                float c[2] = {0.0f, 0.0f};
                for (uint32_t k0 = 0; k0 < K; k0 += 32) {
                    for (int lane = 0; lane < 2; ++lane) {
                        for (uint32_t kk = 0; kk < 16; kk += 8) {
                            const uint32_t k = k0 + 16u * lane + kk;
                            float s = a[k] * b[k];
                            for (uint32_t j = 1; j < 8 && k + j < K; ++j)
                                s += a[k + j] * b[k + j];
                            c[lane] += s;
                        }
                    }
                }
                P[e] = c[0] + c[1];
            }
        }
    }
}

/* ------------------------------------------------------------------------- */
/* Bit-level CPU twins of OUR HIP kernels (bsmr-sddmm_amd/csrc/).  These are  */
/* not reference restatements; they let the GPU tests demand bit equality     */
/* where the hardware arithmetic is a defined fmaf chain.                     */
/* ------------------------------------------------------------------------- */

/* Twin of the residual sparse kernel (csrc/sddmm_sparse.hip): `lpe` lanes    */
/* cooperate on one entry; lane t owns the float4 chunks q with               */
/* q % lpe == t (chunk q = floats [4q, 4q+4)); inside a lane the sum is a     */
/* single fmaf chain over its chunks in ascending k; lanes are then combined  */
/* by a butterfly: for off = lpe/2 .. 1: v[t] += v[t ^ off].                  */
float oracle_sparse_twin_one(uint32_t K, uint32_t lpe, const float *a, const float *b)
{
    float v[64];
    for (uint32_t t = 0; t < lpe; ++t) {
        float acc = 0.0f;
        for (uint32_t q = t; q < K / 4; q += lpe)
            for (uint32_t j = 0; j < 4; ++j)
                acc = fmaf(a[4 * q + j], b[4 * q + j], acc);
        v[t] = acc;
    }
    for (uint32_t off = lpe / 2; off >= 1; off >>= 1) {
        float w[64];
        for (uint32_t t = 0; t < lpe; ++t) w[t] = v[t] + v[t ^ off];
        memcpy(v, w, sizeof(float) * lpe);
    }
    return v[0];
}

void oracle_sparse_twin(uint32_t M, uint32_t K, uint32_t lpe,
                        const uint32_t *rowOffsets, const uint32_t *colIndices,
                        const float *A, const float *B, float *P)
{
#pragma omp parallel for schedule(static)
    for (int64_t row = 0; row < (int64_t)M; ++row)
        for (uint32_t e = rowOffsets[row]; e < rowOffsets[row + 1]; ++e)
            P[e] = oracle_sparse_twin_one(K, lpe, A + (size_t)row * K,
                                          B + (size_t)colIndices[e] * K);
}

/* Twin of the exact-fp32 dense mode (v_mfma_f32_16x16x4_f32: a k-ordered     */
/* fmaf chain, cdna_hip_programming.md "FP32-input MFMA").  In our kernel     */
/* (csrc/sddmm_kernels.hpp denseBlocks32) lane group g loads the float4 chunk */
/* [16t + 4g, +4) and MFMA number (t, j) consumes element j of every chunk,   */
/* so the chain visits k in the order: for t, for j, for g: k = 16t + 4g + j. */
void oracle_dense_f32_twin(uint32_t M, uint32_t K,
                           const uint32_t *rowOffsets, const uint32_t *colIndices,
                           const float *A, const float *B, float *P)
{
#pragma omp parallel for schedule(static)
    for (int64_t row = 0; row < (int64_t)M; ++row) {
        for (uint32_t e = rowOffsets[row]; e < rowOffsets[row + 1]; ++e) {
            const float *a = A + (size_t)row * K;
            const float *b = B + (size_t)colIndices[e] * K;
            float acc = 0.0f;
            for (uint32_t t = 0; t < K / 16; ++t)
                for (uint32_t j = 0; j < 4; ++j)
                    for (uint32_t g = 0; g < 4; ++g) {
                        const uint32_t k = 16 * t + 4 * g + j;
                        acc = fmaf(a[k], b[k], acc);
                    }
            P[e] = acc;
        }
    }
}

/* Dense fp16/bf16 mode yardstick: operands rounded (mode 2 / 3), products    */
/* exact, accumulated in fp64.  The f16 MFMA's internal summation order is    */
/* not architecturally defined, so tests compare within a stated tolerance.   */
void oracle_dense_lowp_model(int mode, uint32_t M, uint32_t K,
                             const uint32_t *rowOffsets, const uint32_t *colIndices,
                             const float *A, const float *B, double *P)
{
#pragma omp parallel for schedule(static)
    for (int64_t row = 0; row < (int64_t)M; ++row) {
        for (uint32_t e = rowOffsets[row]; e < rowOffsets[row + 1]; ++e) {
            const float *a = A + (size_t)row * K;
            const float *b = B + (size_t)colIndices[e] * K;
            double acc = 0.0;
            for (uint32_t k = 0; k < K; ++k) {
                const float ra = mode == 2 ? oracle_round_fp16(a[k]) : oracle_round_bf16(a[k]);
                const float rb = mode == 2 ? oracle_round_fp16(b[k]) : oracle_round_bf16(b[k]);
                acc += (double)ra * (double)rb;
            }
            P[e] = acc;
        }
    }
}

int oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

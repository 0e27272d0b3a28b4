/*
 * mixq_oracle.c -- CPU restatement of the MixQ W8A8O16 linear operator.
 *
 * TEST INFRASTRUCTURE ONLY.  This file is the parity checker for the HIP path
 * (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg).  Nothing in
 * the product package (mixq_tensorrt_llm_amd/) may import, link or call it.
 *
 * Every function restates one piece of the reference (paths relative to
 * /root/reference) and cites the lines it follows.  Arithmetic is written so
 * that it has exactly one answer on any IEEE-754 host: fp16 values are carried
 * as uint16_t bit patterns and converted with the software routines below;
 * no -ffast-math, no x87, no host _Float16.
 *
 * PARITY PIN STATUS (also in DESIGN.md):
 *   - weight packing (to_quantized_weight, weights_scaling_factor, fp_ind):
 *       pinned against the importable Python reference + act_scales fixture,
 *       see tests/golden/gen_golden.py.
 *   - device kernels (quant / extract / int8 GEMM / epilogue / fp16 side GEMM /
 *       decode GEMV) and the EETQ qweight interleave: the reference has no
 *       test, no golden vector and cannot run here (CUDA + TensorRT + cuBLAS)
 *       => "parity unpinned" for those; the restatement follows the cited
 *       source lines.  The int8 GEMM is exact integer math (any correct
 *       implementation agrees bit-for-bit).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------- fp16 -- */

static inline float h2f(uint16_t h)
{
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1fu;
    uint32_t man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do {
                e++;
                man <<= 1;
            } while ((man & 0x400u) == 0);
            man &= 0x3ffu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

/* float -> half, round-to-nearest-even (what __float2half / v_cvt_f16_f32 do). */
static inline uint16_t f2h(float f)
{
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) { /* inf / nan */
        if (ax > 0x7f800000u) return (uint16_t)(sign | 0x7e00u | ((ax >> 13) & 0x3ffu));
        return (uint16_t)(sign | 0x7c00u);
    }
    if (ax >= 0x477ff000u) { /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7c00u);
    }
    if (ax < 0x38800000u) { /* < 2^-14: subnormal half or zero */
        if (ax < 0x33000000u) return (uint16_t)sign; /* < 2^-25 -> 0 (2^-25 itself ties to even = 0) */
        int e = (int)(ax >> 23);                      /* biased fp32 exponent, 102..112 */
        uint32_t man = (ax & 0x7fffffu) | 0x800000u;  /* 24-bit significand */
        int shift = 126 - e;                          /* 14..24 */
        uint32_t q = man >> shift;
        uint32_t rem = man & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (q & 1u))) q++;
        return (uint16_t)(sign | q);
    }
    uint32_t e = (ax >> 23) - 112; /* 1..30 */
    uint32_t man = ax & 0x7fffffu;
    uint32_t q = (e << 10) | (man >> 13);
    uint32_t rem = man & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) q++; /* carry into exponent is correct */
    return (uint16_t)(sign | q);
}

static inline int h_isnan(uint16_t h) { return (h & 0x7fffu) > 0x7c00u; }

ORACLE_API float mixq_oracle_h2f(uint16_t h) { return h2f(h); }
ORACLE_API uint16_t mixq_oracle_f2h(float f) { return f2h(f); }

/* CUDA __hmax(a,b): "if either input is NaN the other is returned" (both NaN -> canonical NaN). */
static inline uint16_t h_max_nan_drop(uint16_t a, uint16_t b)
{
    if (h_isnan(a)) return h_isnan(b) ? 0x7fffu : b;
    if (h_isnan(b)) return a;
    return h2f(a) >= h2f(b) ? a : b;
}

/* CUDA __hdiv: documented as round-to-nearest-even fp16 division.  fp32 divide (correctly
 * rounded) followed by one RNE to fp16 equals the correctly rounded fp16 quotient
 * (24 >= 2*11+2).  See SURVEY.md §7 "numerics parity". */
static inline uint16_t h_div(uint16_t a, uint16_t b) { return f2h(h2f(a) / h2f(b)); }

/* CUDA __half2int_rn: RNE to int32, NaN -> 0, saturating (only +-inf can be out of range). */
static inline int32_t h2int_rn(uint16_t h)
{
    if (h_isnan(h)) return 0;
    if ((h & 0x7fffu) == 0x7c00u) return (h & 0x8000u) ? INT32_MIN : INT32_MAX;
    return (int32_t)nearbyintf(h2f(h)); /* default rounding mode = RNE */
}

/* ------------------------------------------------- a4: per-token quant -- */
/*
 * kernel/i8gemm.cu:66-107 FindRowScaleKernel<256> + :139-150 int8quant
 *   amax  = max_k |A[m,k]|           (fp16, __habs/__hmax; NaN dropped)
 *   sA[m] = __hdiv(amax, 127)        (fp16)
 *   qA    = (int8) __half2int_rn(__hdiv(A[m,k], sA[m]))   -- no clamp, no zero guard
 * zero_cols (optional, len nzero): columns treated as 0 for both amax and q -- the mixlib
 * flavour, where ExtractOutliersAndSetToZeros has zeroed them before FindRowScale
 * (quantkernel/mix_cuda/cult.cu:1426, MixQ/src/mixquant/modules/linear.py:184-190).
 */
ORACLE_API void mixq_oracle_quant_rows(int64_t M, int64_t K, const uint16_t* A, int8_t* qA, uint16_t* sA,
                                       const int32_t* zero_cols, int nzero)
{
#pragma omp parallel
    {
        uint8_t* mask = NULL;
        if (nzero > 0) {
            mask = (uint8_t*)calloc((size_t)K, 1);
            for (int j = 0; j < nzero; ++j)
                if (zero_cols[j] >= 0 && zero_cols[j] < K) mask[zero_cols[j]] = 1;
        }
#pragma omp for schedule(static)
        for (int64_t m = 0; m < M; ++m) {
            const uint16_t* row = A + m * K;
            uint16_t amax = 0;
            int first = 1;
            for (int64_t k = 0; k < K; ++k) {
                uint16_t a = (mask && mask[k]) ? 0 : (uint16_t)(row[k] & 0x7fffu);
                amax = first ? a : h_max_nan_drop(a, amax);
                first = 0;
            }
            uint16_t s = h_div(amax, 0x57f0u /* 127.0 */);
            sA[m] = s;
            for (int64_t k = 0; k < K; ++k) {
                uint16_t x = (mask && mask[k]) ? 0 : row[k];
                qA[m * K + k] = (int8_t)h2int_rn(h_div(x, s));
            }
        }
        free(mask);
    }
}

/* ---------------------------------------------- a2: outlier extraction -- */
/*
 * kernel/i8gemm.cu:198-244: fp_A[m, j] = A[m, ind[j]], A left untouched (line 218 commented out).
 * set_zero != 0 reproduces the mixlib twin (quantkernel/mix_cuda/cult.cu:1406-1432) which also
 * writes 0 into A.
 */
ORACLE_API void mixq_oracle_extract_outliers(int64_t M, int64_t K, uint16_t* A, uint16_t* fpA, const int32_t* ind,
                                             int len, int set_zero)
{
    for (int64_t m = 0; m < M; ++m)
        for (int j = 0; j < len; ++j) fpA[m * len + j] = A[m * K + ind[j]];
    if (set_zero)
        for (int64_t m = 0; m < M; ++m)
            for (int j = 0; j < len; ++j) A[m * K + ind[j]] = 0;
}

/* --------------------------------------------------- a5: int8 GEMM s32 -- */
/*
 * kernel/i8gemm.cu:151-194 -> kernel/symmetric/gemm/kernel/gemm_dequant.h:225-380 main loop:
 * acc[m,n] = sum_k qA[m,k] * W[n,k], exact in int32 (OpMultiplyAddSaturate never saturates for
 * K <= 133000).  Also the unfused cuBLAS route TsinghuaMixQPlugin.cpp:36-77.
 */
__attribute__((target_clones("arch=icelake-server", "arch=skylake-avx512", "avx2", "default"))) static int32_t
dot_s8(const int8_t* a, const int8_t* b, int64_t K)
{
    int32_t s = 0;
    for (int64_t k = 0; k < K; ++k) s += (int32_t)a[k] * (int32_t)b[k];
    return s;
}

ORACLE_API void mixq_oracle_gemm_s8s8s32(int64_t M, int64_t N, int64_t K, const int8_t* qA, const int8_t* W,
                                         int32_t* acc)
{
    /* cache-blocked (round 4): a 16-row block of qA against a 64-row block of W, both resident in L2 while their 1024 dot
     * products are taken -- the unblocked loop streamed all of W from memory once per token row.  Integer sums: same bits. */
    const int64_t MB = 16, NB = 64;
    const int64_t mb = (M + MB - 1) / MB, nb = (N + NB - 1) / NB;
#pragma omp parallel for schedule(static)
    for (int64_t blk = 0; blk < mb * nb; ++blk) {
        const int64_t m0 = (blk / nb) * MB, n0 = (blk % nb) * NB;
        const int64_t m1 = m0 + MB < M ? m0 + MB : M, n1 = n0 + NB < N ? n0 + NB : N;
        for (int64_t n = n0; n < n1; ++n)
            for (int64_t m = m0; m < m1; ++m) acc[m * N + n] = dot_s8(qA + m * K, W + n * K, K);
    }
}

/* ------------------------------------------- a3: fp16 outlier side GEMM -- */
/*
 * TsinghuaMixQPlugin.cpp:122-161 gemmfp16: cublasGemmEx fp16 x fp16, CUBLAS_COMPUTE_32F, fp16 out,
 * beta = 0.  P[m,n] = fp16( sum_j float(fpA[m,j]) * float(fpW[n,j]) ).  cuBLAS' accumulation order is
 * unspecified; this restatement sums j = 0..O-1 in order (products of two fp16 are exact in fp32).
 */
ORACLE_API void mixq_oracle_gemm_fp16(int64_t M, int64_t N, int O, const uint16_t* fpA, const uint16_t* fpW,
                                      uint16_t* P)
{
#pragma omp parallel
    {
        float* wf = (float*)malloc(sizeof(float) * (size_t)O);
        float* af = (float*)malloc(sizeof(float) * (size_t)O);
#pragma omp for schedule(static)
        for (int64_t m = 0; m < M; ++m) {
            for (int j = 0; j < O; ++j) af[j] = h2f(fpA[m * O + j]);
            for (int64_t n = 0; n < N; ++n) {
                float s = 0.f;
                for (int j = 0; j < O; ++j) s += af[j] * h2f(fpW[n * O + j]);
                P[m * N + n] = f2h(s);
            }
        }
        free(wf);
        free(af);
    }
}

/* ----------------------------------------------- a6: fused epilogue ------ */
/*
 * kernel/symmetric/epilogue/thread/linear_combination_dequant.h:152-157
 *   D = fp16( float(acc) * (float(sW[n]) * float(sA[m])) + float(C[m,n]) )
 * nvcc's default -fmad=true contracts the outer a*b+c into one fma; the inner product of two fp16
 * values is exact in fp32.  silu != 0: LinearCombinationDequantSilu (:176-270), x/(1+expf(-x)).
 */
ORACLE_API void mixq_oracle_dequant_epilogue(int64_t M, int64_t N, const int32_t* acc, const uint16_t* sA,
                                             const uint16_t* sW, const uint16_t* C, uint16_t* D, int silu)
{
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        float sa = h2f(sA[m]);
        for (int64_t n = 0; n < N; ++n) {
            float c = C ? h2f(C[m * N + n]) : 0.f;
            float v = fmaf((float)acc[m * N + n], h2f(sW[n]) * sa, c);
            if (silu) v = v / (1.f + expf(-v));
            D[m * N + n] = f2h(v);
        }
    }
}

/*
 * Unfused variant, kernel/i8gemm.cu:258-279 dequantizationKernel:
 *   out = __hadd( fp16( (float(x) * float(sA[m])) * float(sW[n]) ), out )
 */
ORACLE_API void mixq_oracle_dequantization(int64_t M, int64_t N, const int32_t* x, const uint16_t* sA,
                                           const uint16_t* sW, uint16_t* out)
{
    for (int64_t m = 0; m < M; ++m)
        for (int64_t n = 0; n < N; ++n) {
            uint16_t t = f2h(((float)x[m * N + n] * h2f(sA[m])) * h2f(sW[n]));
            out[m * N + n] = f2h(h2f(t) + h2f(out[m * N + n])); /* __hadd: exact sum in fp32, one RNE */
        }
}

/* ------------------------------------- a1: the whole prefill path (M>4) -- */
/*
 * TsinghuaMixQPlugin.cpp:518-532, in the reference's order:
 *   ExtractOutliersAndSetToZeros -> gemmfp16 -> int8quant -> int8FusedDequantizeCUDA(C = D = Out).
 * Optional outputs (may be NULL): qA_out [M,K], sA_out [M], acc_out [M,N], P_out [M,N].
 */
ORACLE_API void mixq_oracle_linear_prefill(int64_t M, int64_t N, int64_t K, int O, const uint16_t* A,
                                           const int8_t* W, const uint16_t* sW, const uint16_t* fpW,
                                           const int32_t* ind, uint16_t* Out, int8_t* qA_out, uint16_t* sA_out,
                                           int32_t* acc_out, uint16_t* P_out)
{
    uint16_t* fpA = (uint16_t*)malloc(sizeof(uint16_t) * (size_t)M * (size_t)O);
    int8_t* qA = qA_out ? qA_out : (int8_t*)malloc((size_t)M * (size_t)K);
    uint16_t* sA = sA_out ? sA_out : (uint16_t*)malloc(sizeof(uint16_t) * (size_t)M);
    int32_t* acc = acc_out ? acc_out : (int32_t*)malloc(sizeof(int32_t) * (size_t)M * (size_t)N);
    uint16_t* P = P_out ? P_out : (uint16_t*)malloc(sizeof(uint16_t) * (size_t)M * (size_t)N);

    mixq_oracle_extract_outliers(M, K, (uint16_t*)A, fpA, ind, O, 0);
    mixq_oracle_gemm_fp16(M, N, O, fpA, fpW, P);
    mixq_oracle_quant_rows(M, K, A, qA, sA, NULL, 0);
    mixq_oracle_gemm_s8s8s32(M, N, K, qA, W, acc);
    mixq_oracle_dequant_epilogue(M, N, acc, sA, sW, P, Out, 0);

    free(fpA);
    if (!qA_out) free(qA);
    if (!sA_out) free(sA);
    if (!acc_out) free(acc);
    if (!P_out) free(P);
}

/* ------------------------------------ quantize-time producer (caller) ---- */
/*
 * modelopt/torch/export/model_config_utils.py:429-430
 *   weights_scaling_factor = fp16( max_k |W[n,k]| / 127 )   (torch: fp16 tensor / python int -> fp16 division)
 * torch's CPU fp16 division computes in fp32 and rounds once to fp16.
 */
ORACLE_API void mixq_oracle_weight_scales(int64_t N, int64_t K, const uint16_t* W, uint16_t* sW)
{
    for (int64_t n = 0; n < N; ++n) {
        float mx = 0.f;
        for (int64_t k = 0; k < K; ++k) {
            float a = fabsf(h2f(W[n * K + k]));
            if (a > mx || a != a) mx = a; /* torch.max propagates NaN */
        }
        sW[n] = f2h(h2f(f2h(mx)) / 127.0f);
    }
}

/*
 * model_config_utils.py:298-308 to_quantized_weight (int8_mix):
 *   (weight / s[:,None]).round().clamp(-128,127).to(int8)    -- fp16 division, torch.round = half-to-even
 * zero_cols: the outlier columns, zeroed at :453 BEFORE quantisation (scales at :429 are computed before).
 */
ORACLE_API void mixq_oracle_quantize_weight(int64_t N, int64_t K, const uint16_t* W, const uint16_t* sW,
                                            const int32_t* zero_cols, int nzero, int8_t* Wq)
{
    uint8_t* mask = (uint8_t*)calloc((size_t)K, 1);
    for (int j = 0; j < nzero; ++j) mask[zero_cols[j]] = 1;
    for (int64_t n = 0; n < N; ++n)
        for (int64_t k = 0; k < K; ++k) {
            /* W[:, ind] *= 0 keeps the sign of zero / turns inf into nan; irrelevant after round+clamp
               except NaN -> int8 cast, which torch leaves unspecified: treat as 0. */
            uint16_t w = mask[k] ? f2h(h2f(W[n * K + k]) * 0.0f) : W[n * K + k];
            float q = nearbyintf(h2f(f2h(h2f(w) / h2f(sW[n]))));
            if (q != q) q = 0.f;
            if (q < -128.f) q = -128.f;
            if (q > 127.f) q = 127.f;
            Wq[n * K + k] = (int8_t)q;
        }
    free(mask);
}

/*
 * model_config_utils.py:446-448: fp_ind = torch.sort(layer_scales)[1][-128:]  (ascending, stable=False).
 * torch.sort on CPU is a stable sort in practice for 1-D float tensors; ties keep index order.
 */
typedef struct {
    float v;
    int32_t i;
} oracle_kv;
static int cmp_kv(const void* a, const void* b)
{
    const oracle_kv* x = (const oracle_kv*)a;
    const oracle_kv* y = (const oracle_kv*)b;
    if (x->v < y->v) return -1;
    if (x->v > y->v) return 1;
    return (x->i > y->i) - (x->i < y->i);
}
ORACLE_API void mixq_oracle_select_outliers(int64_t K, const float* act_scale, int O, int32_t* ind)
{
    oracle_kv* kv = (oracle_kv*)malloc(sizeof(oracle_kv) * (size_t)K);
    for (int64_t k = 0; k < K; ++k) {
        kv[k].v = act_scale[k];
        kv[k].i = (int32_t)k;
    }
    qsort(kv, (size_t)K, sizeof(oracle_kv), cmp_kv);
    for (int j = 0; j < O; ++j) ind[j] = kv[K - O + j].i;
    free(kv);
}

/* --------------------------- EETQ / FasterTransformer qweight layout ------ */
/*
 * weightonlykernel/cutlass_kernels/cutlass_preprocessors.cc:573-660 symmetric_quantize (int8):
 *   per output column n of Wt[K,N]: scale = max_k|Wt[k,n]| / 128 (stored as fp16 -- :610,633),
 *   q = (int8) clamp(round_half_away(Wt[k,n] / scale_fp32), -128, 127)        (:640-648)
 * note: the division uses the fp32 per_col_max*scale value, not the fp16-rounded scale.
 */
ORACLE_API void mixq_oracle_eetq_symmetric_quantize(int64_t K, int64_t N, const uint16_t* Wt, int8_t* q,
                                                    uint16_t* scales)
{
    float* cmax = (float*)calloc((size_t)N, sizeof(float));
    for (int64_t k = 0; k < K; ++k)
        for (int64_t n = 0; n < N; ++n) {
            float a = fabsf(h2f(Wt[k * N + n]));
            if (a > cmax[n]) cmax[n] = a;
        }
    for (int64_t n = 0; n < N; ++n) {
        cmax[n] *= 1.f / 128.f;
        scales[n] = f2h(cmax[n]);
    }
    for (int64_t k = 0; k < K; ++k)
        for (int64_t n = 0; n < N; ++n) {
            float s = roundf(h2f(Wt[k * N + n]) / cmax[n]);
            float c = fmaxf(-128.f, fminf(127.f, s));
            q[k * N + n] = (int8_t)c;
        }
    free(cmax);
}

/*
 * cutlass_preprocessors.cc:497-534 preprocess_weights_for_mixed_gemm, int8, arch 80..90:
 *   (1) permute_B_rows_for_mixed_gemm   (:130-195)  rows in groups of 16: 0 1 8 9 2 3 10 11 4 5 12 13 6 7 14 15
 *   (2) subbyte_transpose               (:201-335)  [K,N] row-major -> [N,K] (column-major view)
 *   (3) interleave_column_major_tensor  (:432-495)  ColumnMajorTileInterleave<64,2>: rows_per_tile 64,
 *                                                   2 columns interleaved
 *   (4) add_bias_and_interleave_int8s   (:337-358)  +128, then swap bytes 1<->2 of every 32-bit word
 * in: row-major int8 [K,N]; out: uint8 buffer of K*N bytes.
 */
ORACLE_API void mixq_oracle_eetq_preprocess(int64_t K, int64_t N, const int8_t* in, uint8_t* out)
{
    size_t total = (size_t)K * (size_t)N;
    int8_t* a = (int8_t*)malloc(total);
    int8_t* b = (int8_t*)malloc(total);
    /* (1) row permutation inside each 16-row group */
    for (int64_t base = 0; base < K; base += 16)
        for (int t = 0; t < 16; ++t) {
            int src = 8 * ((t % 4) / 2) + t % 2 + 2 * (t / 4);
            memcpy(a + (base + t) * N, in + (base + src) * N, (size_t)N);
        }
    /* (2) transpose -> b[n*K + k] */
    for (int64_t k = 0; k < K; ++k)
        for (int64_t n = 0; n < N; ++n) b[n * K + k] = a[k * N + n];
    /* (3) interleave two columns in 64-row tiles; unit of movement = one 32-bit word (4 rows) */
    {
        const int64_t vec_rows = K / 4, vec_rows_per_tile = 64 / 4, il = 2;
        const uint32_t* src = (const uint32_t*)b;
        uint32_t* dst = (uint32_t*)a;
        for (int64_t col = 0; col < N; ++col) {
            int64_t wcol = col / il;
            for (int64_t base = 0; base < vec_rows; base += vec_rows_per_tile)
                for (int64_t r = base; r < base + vec_rows_per_tile && r < vec_rows; ++r) {
                    int64_t wrow = il * base + vec_rows_per_tile * (col % il) + r % vec_rows_per_tile;
                    dst[wcol * vec_rows * il + wrow] = src[col * vec_rows + r];
                }
        }
    }
    /* (4) bias to unsigned, swap bytes 1 and 2 */
    for (size_t i = 0; i < total; ++i) out[i] = (uint8_t)((int)a[i] + 128);
    for (size_t i = 0; i < total; i += 4) {
        uint8_t t = out[i + 1];
        out[i + 1] = out[i + 2];
        out[i + 2] = t;
    }
    free(a);
    free(b);
}

/* ----------------------------------------------- a7: decode path (M<=4) -- */
/*
 * weightonlykernel/fpA_intB_gemm_wrapper.cu:29-58 -> weightOnlyBatchedGemv/kernel.h:300-470 (Int8b,
 * per-channel): Out[m,n] = sum_k A[m,k] * ((Wq[k,n]-128) * scale[n]).
 * The CUDA kernel forms w16 = fp16(w*scale) (hfma2, :367-369), accumulates per-thread partial sums in
 * fp16 (:425-433) and across threads in fp32; its per-thread partition is an implementation detail of
 * that launch shape, so this restatement keeps the fp16-rounded weight but accumulates in fp32 (the
 * tolerance for this path is set in tests/test_decode.py).  Wq_rm is the UN-interleaved row-major
 * int8 [K,N] (signed); the layout is covered separately by mixq_oracle_eetq_preprocess.
 */
ORACLE_API void mixq_oracle_w8a16_gemv(int64_t M, int64_t N, int64_t K, const uint16_t* A, const int8_t* Wq_rm,
                                       const uint16_t* scale, uint16_t* Out)
{
    /* any M: this is also the checker of the fpA_intB GEMM (M > 4).  Activations are converted once; per column the
     * fp16-rounded weights are formed once and every row's dot product runs k = 0..K-1 in fp32. */
    float* Af = (float*)malloc(sizeof(float) * (size_t)M * (size_t)K);
    for (int64_t i = 0; i < M * K; ++i) Af[i] = h2f(A[i]);
#pragma omp parallel
    {
        float* w16 = (float*)malloc(sizeof(float) * (size_t)K);
#pragma omp for schedule(static)
        for (int64_t n = 0; n < N; ++n) {
            const float sc = h2f(scale[n]);
            for (int64_t k = 0; k < K; ++k) w16[k] = h2f(f2h((float)Wq_rm[k * N + n] * sc));
            for (int64_t m = 0; m < M; ++m) {
                const float* a = Af + m * K;
                float s = 0.f;
                for (int64_t k = 0; k < K; ++k) s += a[k] * w16[k];
                Out[m * N + n] = f2h(s);
            }
        }
        free(w16);
    }
    free(Af);
}

/*
 * The SAME decode path with the reference kernel's own summation order, for M <= 4 (the only M the CUDA kernel
 * serves, kernelLauncher.cu:166-199: Int8b per-channel = <NPerBlock 2, Batch m, BlockSize 256>, kInterleave 2):
 *   block b -> columns 4b .. 4b+3 (kernel.h:312: n_start = bid * NPerBlock * Interleave); thread tid takes column
 *   2*idx + (tid/4)%2 of that group (:315, idx = 0,1) and, per trip `it` of the k loop (:332-333, 4096 interleaved
 *   bytes per trip), the 16 consecutive k starting at (tid/8)*64 + (tid%4)*16 + it*2048 (WeightOnlyScaleLoader
 *   offset/advance, :263-264, :288-291), while tid*16 + it*4096 < 2K;
 *   w16 = hfma2(fp16(q), scale, 0) (:367-369); the thread's partial sum is an fp16 FMA chain over y = 0..15 and over
 *   trips, acc = fp16(w16*a + acc) (:425-433, "we use fp16 for accumulation within threads"); then fp32: warp butterfly
 *   xor 16, 8, 2, 1 (Int8b Layout::sync, :150-158), lanes 0 / 4 -> shared memory, 8 warps summed in order j = 0..7 from
 *   0.f (:452-457), out = fp16(v) (:464).
 * Deterministic; the fp16 chain is evaluated exactly (product and sum formed in double with round-to-odd, then one
 * RNE to fp16).  Wq_rm is the un-interleaved signed int8 [K,N].
 */
static uint16_t d2h_rne(double d)
{
    /* double -> fp16, round to nearest even, via float only when exact: do it directly on the value */
    if (d != d) return 0x7e00u;
    uint16_t sign = 0;
    if (d < 0 || (d == 0 && 1.0 / d < 0)) {
        sign = 0x8000u;
        d = -d;
    }
    if (d >= 65520.0) return (uint16_t)(sign | 0x7c00u); /* rounds to inf */
    if (d == 0.0) return sign;
    int e;
    (void)frexp(d, &e);   /* d = f * 2^e, f in [0.5, 1) */
    int q = e - 11;       /* ulp exponent for 11 significant bits */
    if (q < -24) q = -24; /* subnormal range: ulp = 2^-24 */
    double scaled = ldexp(d, -q);
    double r = nearbyint(scaled); /* RNE (default rounding mode) */
    double v = ldexp(r, q);
    return (uint16_t)(sign | f2h((float)v)); /* v is exactly representable in fp16 (or the next power of two) */
}

static uint16_t hfma_exact(uint16_t a, uint16_t b, uint16_t c)
{
    const double p = (double)h2f(a) * (double)h2f(b); /* exact: 22 significant bits */
    const double cc = (double)h2f(c);
    double s = p + cc;
    /* TwoSum error term; if the double sum was inexact make it odd (round-to-odd) so the final RNE is the true one */
    const double bb = s - p;
    const double err = (p - (s - bb)) + (cc - bb);
    if (err != 0.0 && s == s && s - s == 0.0) {
        uint64_t bits;
        memcpy(&bits, &s, 8);
        if ((bits & 1u) == 0) {
            const int up = (err > 0) == (s > 0); /* true value is farther from zero than s */
            bits = up ? bits + 1 : bits - 1;
            memcpy(&s, &bits, 8);
        }
    }
    return d2h_rne(s);
}

ORACLE_API void mixq_oracle_w8a16_gemv_reforder(int64_t M, int64_t N, int64_t K, const uint16_t* A,
                                                const int8_t* Wq_rm, const uint16_t* scale, uint16_t* Out)
{
    if (M > 4 || (N % 4) != 0 || (K % 64) != 0) return; /* the CUDA kernel's own domain */
#pragma omp parallel for schedule(static)
    for (int64_t nb = 0; nb < N / 4; ++nb) {
        for (int nid = 0; nid < 4; ++nid) {
            const int64_t n = nb * 4 + nid;
            const int inter = nid & 1;
            const uint16_t sc = scale[n];
            for (int64_t m = 0; m < M; ++m) {
                float warp_sum[8];
                for (int w = 0; w < 8; ++w) {
                    float lane[32];
                    for (int l = 0; l < 32; ++l) {
                        const int tid = w * 32 + l;
                        uint16_t acc = 0; /* fp16 accumulator of this thread for (m, column n) */
                        if (((tid / 4) & 1) == inter) {
                            for (int64_t it = 0; (int64_t)tid * 16 + it * 4096 < 2 * K; ++it) {
                                const int64_t k0 = (int64_t)(tid / 8) * 64 + (tid % 4) * 16 + it * 2048;
                                for (int y = 0; y < 16; ++y) {
                                    const uint16_t w16 = f2h((float)Wq_rm[(k0 + y) * N + n] * h2f(sc)); /* hfma2(q, s, 0) */
                                    acc = hfma_exact(w16, A[m * K + k0 + y], acc);
                                }
                            }
                        }
                        lane[l] = h2f(acc);
                    }
                    /* threads of the other column of the pair hold ITS sums in the same register; the butterfly below only
                     * mixes lanes with equal (l/4)%2, so zero-filling them here is equivalent for lanes 0 / 4 */
                    const int steps[4] = {16, 8, 2, 1};
                    for (int si = 0; si < 4; ++si) {
                        float nxt[32];
                        for (int l = 0; l < 32; ++l) nxt[l] = lane[l] + lane[l ^ steps[si]];
                        memcpy(lane, nxt, sizeof(lane));
                    }
                    warp_sum[w] = lane[inter ? 4 : 0];
                }
                float v = 0.f;
                for (int w = 0; w < 8; ++w) v += warp_sum[w];
                Out[m * N + n] = f2h(v);
            }
        }
    }
}

/*
 * quantkernel/mix_cuda/cult.cu:2301-2324 dequantizationKernelSilu (mixlib dequantizeInt8Silu, :2341-2348):
 *   out = fp16( silu( (float(x) * sRow[m]) * sCol[n] + float(y) ) ),  silu(v) = v / (1 + expf(-v))
 * fp32 throughout; nvcc contracts the last multiply-add into one fma (default -fmad=true).
 */
ORACLE_API void mixq_oracle_dequantization_silu(int64_t M, int64_t N, const int32_t* x, const uint16_t* sA,
                                                const uint16_t* sW, const uint16_t* y, uint16_t* out)
{
    for (int64_t m = 0; m < M; ++m)
        for (int64_t n = 0; n < N; ++n) {
            const float t = (float)x[m * N + n] * h2f(sA[m]);
            const float v = fmaf(t, h2f(sW[n]), h2f(y[m * N + n]));
            out[m * N + n] = f2h(v / (1.f + expf(-v)));
        }
}

ORACLE_API int mixq_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* the Python side caps the team at the CPUs the process may really use (cgroup quota): a host process that loaded an OpenMP
 * runtime earlier (torch does) has already sized it from the CPUs it can see */
ORACLE_API void mixq_oracle_set_num_threads(int n)
{
    if (n > 0) omp_set_num_threads(n);
}


/* ------------------- next row f1: fused RMSNorm -> extract -> quantise ------------------ */
/*
 * quantkernel/mix_cuda/layernorm/layernorm.cu:122-198 generalT5LayerNorm_extract_outliers (T5 style: no mean, no bias):
 *   rstd    = rsqrtf( sum_k x^2 / n + eps )                         (fp32; block-reduce order unspecified)
 *   out[k]  = clamp_inf_for_half( (x[k]*rstd) * gamma[k] )          (reduction.cuh:105-115: clamp to +-64504, RNE fp16)
 *   outliers[j] = out[ind[j]] ; out[ind[j]] = 0                     (:155-161, P-flavour zeroing)
 *   amax/scale/quant of the zeroed row exactly as FindRowScaleKernel (:163-196)
 * The sum is accumulated in double here (the "ideal" value of an order-unspecified fp32 reduction); CUDA's rsqrtf is an
 * approximation too, so `out` is compared with a 1-ulp (1e-3) tolerance, not bit for bit.
 * gamma == NULL: plain RMSNorm (layernorm_forward_cuda, :100-117): only `out` is produced.
 */
ORACLE_API void mixq_oracle_rmsnorm_extract_quant(int64_t M, int64_t K, const uint16_t* x, const uint16_t* gamma,
                                                  float eps, const int32_t* ind, int len, uint16_t* out,
                                                  uint16_t* outliers, int8_t* q, uint16_t* scale)
{
    for (int64_t m = 0; m < M; ++m) {
        const uint16_t* row = x + m * K;
        double ss = 0.0;
        for (int64_t k = 0; k < K; ++k) {
            double v = (double)h2f(row[k]);
            ss += v * v;
        }
        float rstd = (float)(1.0 / sqrt(ss / (double)K + (double)eps));
        uint16_t* o = out + m * K;
        for (int64_t k = 0; k < K; ++k) {
            float v = (h2f(row[k]) * rstd) * h2f(gamma[k]);
            v = v > 0.0f ? fminf(v, 65504.0f - 1000.0f) : fmaxf(v, -65504.0f + 1000.0f);
            o[k] = f2h(v);
        }
        if (q == NULL) continue;
        for (int j = 0; j < len; ++j) outliers[m * len + j] = o[ind[j]];
        for (int j = 0; j < len; ++j) o[ind[j]] = 0;
        mixq_oracle_quant_rows(1, K, o, q + m * K, scale + m, NULL, 0);
    }
}

// Per-token activation quantisation + outlier-column extraction for gfx950, one pass over A.
//
// Replaces (reference, CUDA):
//   kernel/i8gemm.cu:66-107,139-150   FindRowScaleKernel<256> / int8quant   (reads A twice, scalar 2-B loads)
//   kernel/i8gemm.cu:198-244          FindOutliersAndSetToZeros_kernel      (stride-K column gather)
//   quantkernel/mix_cuda/cult.cu:2616-2709 FindRowScaleFusedExtracOutliers  (the fused P-flavour shape)
//
// Design (HBM-bound: 2*M*K bytes in, M*K + 2*M + 2*M*O bytes out):
//   * a row is owned by TPR threads (64 = one wavefront for K <= 8192, else the whole 256-thread block) and is
//     held in registers as 16-byte vectors -> A is read from HBM exactly once, fully coalesced
//     (64 lanes x 16 B = 1 KiB per load instruction);
//   * amax is an integer max over the fp16 |x| bit patterns (order-preserving for non-NaN), NaNs are dropped the way
//     CUDA's __hmax drops them; wave reduction by DPP/bpermute shuffles, cross-wave through 16 B of LDS;
//   * the outlier gather re-reads 2*O bytes of the row this block has just streamed (L2/L1 hits, no HBM traffic).
#include "mixq_device.h"
#include "mixq_launch.h"
#include <atomic>

namespace mixq {

constexpr int QBLOCK = 256;

// FRAG (decode batches only): qA is written in the skinny GEMM's MFMA B-fragment order instead of row-major -- for 16-row tile t
// and 64-byte k-step s, the 1-KiB block t * ceil(K / 64) + s holds lane l's 16 bytes (row t * 16 + l % 16, k = s * 64 + (l / 16)
// * 16 ...) at l * 16 -- so that a fragment load of gemm_skinny_kernel is ONE contiguous 1-KiB read instead of 16 rows x 64
// bytes 4 KiB apart (gemm_skinny_kernels.hip; 32 x 4096 x 4096: GEMM 7.9 -> 5.4 us).
// (A K-slice-major image [K / 128][M][128 B] for the 256 x 256 ping-pong GEMM was FRAG == 2 for one build: GEMM -0.3 %, this kernel
//  +7.6 %, end to end -0.35 %: removed; the GEMM keeps its reader, ablation 512.)
template <int TPR, int MAXV, bool ZERO, int FRAG = 0>
__global__ __launch_bounds__(QBLOCK) void quant_extract_kernel(uint16_t* __restrict__ A, int8_t* __restrict__ qA,
                                                               uint16_t* __restrict__ sA, uint16_t* __restrict__ fpA,
                                                               const int32_t* __restrict__ ind, int M, int K, int O,
                                                               unsigned* __restrict__ zero_words, void* dbg)
{
    dbg_stamp(dbg, 0); // (measurement only: NULL in production) entry
    // (enqueue: the hand-over words of the GEMM's K split over workgroups are cleared here, one launch earlier, instead
    //  of by a memset node of their own -- the plugin workspace is shared with whatever else the engine runs)
    if (zero_words != nullptr && blockIdx.x == 0)
        for (int i = threadIdx.x; i < (int)(kSplitkWordsBytes / 4); i += QBLOCK) zero_words[i] = 0u;
    constexpr int RPB = QBLOCK / TPR; // rows per block
    __shared__ int red[QBLOCK / 64];
    extern __shared__ __attribute__((aligned(16))) unsigned char zmask[]; // ZERO only: K bits, 1 = outlier column

    const int tid = threadIdx.x;
    const int t = tid % TPR;
    const int64_t row = (int64_t)blockIdx.x * RPB + tid / TPR;
    const bool row_ok = row < M;
    const int nvec = K >> 3; // 16-byte vectors per row (K % 8 == 0 is checked on the host)

    if (ZERO) {
        for (int i = tid; i < (K + 31) / 32; i += QBLOCK) reinterpret_cast<unsigned*>(zmask)[i] = 0u;
        __syncthreads();
        for (int j = tid; j < O; j += QBLOCK) {
            int c = ind[j];
            if (c >= 0 && c < K) atomicOr(reinterpret_cast<unsigned*>(zmask) + (c >> 5), 1u << (c & 31));
        }
        __syncthreads();
    }

    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(A + (row_ok ? row : 0) * (int64_t)K);
    uint4 x[MAXV];
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
        const int idx = v * TPR + t;
        x[v] = (row_ok && idx < nvec) ? src[idx] : make_uint4(0u, 0u, 0u, 0u);
    }

    // outlier gather (before any zero write-back); strided over the row's TPR threads
    dbg_stamp(dbg, 1); // row loads issued
    if (fpA != nullptr && row_ok) {
        for (int j = t; j < O; j += TPR) fpA[row * (int64_t)O + j] = A[row * (int64_t)K + ind[j]];
    }
    dbg_stamp(dbg, 2); // outlier gather issued (two dependent loads + a store per element)

    if (ZERO) {
#pragma unroll
        for (int v = 0; v < MAXV; ++v) {
            const int idx = v * TPR + t;
            if (idx < nvec) {
                unsigned m8 = (reinterpret_cast<const unsigned*>(zmask)[idx >> 2] >> ((idx & 3) * 8)) & 0xffu;
                unsigned w[4] = {x[v].x, x[v].y, x[v].z, x[v].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (m8 & (1u << (2 * e))) w[e] &= 0xffff0000u;
                    if (m8 & (2u << (2 * e))) w[e] &= 0x0000ffffu;
                }
                x[v] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
    }

    // ---- amax over |x| bit patterns (order-preserving for non-NaN): packed 16-bit max, NaN patterns included ----
    unsigned m2 = 0u;
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
        m2 = pk_max_u16(m2, x[v].x & 0x7fff7fffu);
        m2 = pk_max_u16(m2, x[v].y & 0x7fff7fffu);
        m2 = pk_max_u16(m2, x[v].z & 0x7fff7fffu);
        m2 = pk_max_u16(m2, x[v].w & 0x7fff7fffu);
    }
    auto row_max = [&](int val) __attribute__((always_inline)) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) val = max(val, __shfl_xor(val, off, 64));
        if (TPR > 64) {
            __syncthreads(); // (also orders a second use of red[])
            if ((tid & 63) == 0) red[tid >> 6] = val;
            __syncthreads();
            val = max(max(red[0], red[1]), max(red[2], red[3]));
        }
        return val;
    };
    const int amax_all = row_max((int)max(m2 & 0xffffu, m2 >> 16));
    int amax = amax_all;
    if (amax_all > 0x7c00) { // the row holds a NaN: redo the max with NaNs dropped the way __hmax drops them
        amax = -1;           // (-1 = every element is NaN)
#pragma unroll
        for (int v = 0; v < MAXV; ++v) {
            const unsigned w[4] = {x[v].x, x[v].y, x[v].z, x[v].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int lo = (int)(w[e] & 0x7fffu), hi = (int)((w[e] >> 16) & 0x7fffu);
                lo = lo > 0x7c00 ? -1 : lo;
                hi = hi > 0x7c00 ? -1 : hi;
                amax = max(amax, max(lo, hi));
            }
        }
        amax = row_max(amax);
    }
    dbg_stamp(dbg, 3); // row data arrived, amax reduced
    const uint16_t amax_bits = amax < 0 ? (uint16_t)0x7fffu : (uint16_t)amax;
    const uint16_t s_bits = f2h_bits(h2f(amax_bits) / 127.0f); // __hdiv(max, 127.0)
    const float s = h2f(s_bits);
    const float rs = 1.0f / s; // IEEE; inf for a zero scale, 0 for an infinite one (same special cases as x / s)
    if (row_ok && t == 0) sA[row] = s_bits;

    // ---- quantise: 8 fp16 -> 8 int8 per vector, one 8-byte store per lane ----
    uint2* __restrict__ dst = reinterpret_cast<uint2*>(qA + (row_ok ? row : 0) * (int64_t)K);
    // 8-byte group idx (k = 8 idx .. 8 idx + 7) of this row -> its place in the image
    auto slot = [&](int idx) __attribute__((always_inline)) -> uint2* {
        if (FRAG == 0) return dst + idx;
        const int64_t r = row_ok ? row : 0;
        const int nsteps = (K + 63) >> 6;
        const int64_t blk = (r >> 4) * nsteps + (idx >> 3);                               // (tile, k-step)
        return reinterpret_cast<uint2*>(qA + (blk << 10) + (((idx >> 1) & 3) << 8) + ((r & 15) << 4) + ((idx & 1) << 3));
    };
    const bool stream_out = FRAG == 0 && (int64_t)M * K >= ((int64_t)8 << 20); // (launch-uniform)
    if (amax_all < 0x7c00 && s_bits != 0) { // every element finite, scale finite and non-zero (row-uniform)
#pragma unroll
        for (int v = 0; v < MAXV; ++v) {
            const int idx = v * TPR + t;
            if (row_ok && idx < nvec) {
                typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
                const uint2 q8 = quant_vec8_finite(x[v], s, rs);
                // prefill-size images stream past the L2 (non-temporal): nothing of this launch reads them, and the GEMM that does
                // starts after the end-of-kernel write-back anyway: -3.5..-7 % on the quantiser (profiles/r03_quant_nt_ab.txt).
                // (Non-temporal LOADS of the row were measured too: +27 %, the outlier gather re-reads the row's lines.)
                if (stream_out) __builtin_nontemporal_store(v2u_{q8.x, q8.y}, reinterpret_cast<v2u_*>(slot(idx)));
                else *slot(idx) = q8;
            }
        }
    } else { // inf / NaN elements, zero scale (zero or tiny row): quotients may be inf / NaN -> exact chain
        for (int v = 0; v < MAXV; ++v) {
            const int idx = v * TPR + t;
            if (row_ok && idx < nvec) {
                const unsigned w[4] = {x[v].x, x[v].y, x[v].z, x[v].w};
                unsigned o[2] = {0u, 0u};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int q0 = quant_one(h2f((uint16_t)(w[e] & 0xffffu)), s);
                    int q1 = quant_one(h2f((uint16_t)(w[e] >> 16)), s);
                    o[e >> 1] |= (unsigned)(q0 | (q1 << 8)) << ((e & 1) * 16);
                }
                *slot(idx) = make_uint2(o[0], o[1]);
            }
        }
    }

    if (dbg != nullptr) {
        dbg_stamp(dbg, 4); // quantised row stored (issued)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dbg_stamp(dbg, 5); // stores acknowledged
    }
    if (ZERO) {
        // mixlib flavour mutates A (cult.cu:1426).  All loads of this row have been consumed (amax needed them);
        // the barrier orders other waves' loads of the same row before these stores when TPR > 64.
        if (TPR > 64) __syncthreads();
        if (row_ok)
            for (int j = t; j < O; j += TPR) {
                int c = ind[j];
                if (c >= 0 && c < K) A[row * (int64_t)K + c] = 0;
            }
    }
}

// Generic fallback for rows longer than 256 threads x 16 vectors (K > 32768): two passes over global memory.
template <bool ZERO>
__global__ __launch_bounds__(QBLOCK) void quant_extract_long_kernel(uint16_t* __restrict__ A, int8_t* __restrict__ qA,
                                                                    uint16_t* __restrict__ sA,
                                                                    uint16_t* __restrict__ fpA,
                                                                    const int32_t* __restrict__ ind, int M, int K, int O,
                                                                    unsigned* __restrict__ zero_words)
{
    if (zero_words != nullptr && blockIdx.x == 0)
        for (int i = threadIdx.x; i < (int)(kSplitkWordsBytes / 4); i += QBLOCK) zero_words[i] = 0u;
    __shared__ int red[QBLOCK / 64];
    extern __shared__ __attribute__((aligned(16))) unsigned char zmask[];
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;
    if (ZERO) {
        for (int i = tid; i < (K + 31) / 32; i += QBLOCK) reinterpret_cast<unsigned*>(zmask)[i] = 0u;
        __syncthreads();
        for (int j = tid; j < O; j += QBLOCK) {
            int c = ind[j];
            if (c >= 0 && c < K) atomicOr(reinterpret_cast<unsigned*>(zmask) + (c >> 5), 1u << (c & 31));
        }
        __syncthreads();
    }
    const uint16_t* a = A + row * (int64_t)K;
    if (fpA != nullptr)
        for (int j = tid; j < O; j += QBLOCK) fpA[row * (int64_t)O + j] = a[ind[j]];
    int amax = -1;
    for (int k = tid; k < K; k += QBLOCK) {
        int b = (ZERO && ((reinterpret_cast<const unsigned*>(zmask)[k >> 5] >> (k & 31)) & 1u)) ? 0 : (a[k] & 0x7fff);
        amax = max(amax, b > 0x7c00 ? -1 : b);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = max(amax, __shfl_xor(amax, off, 64));
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = max(max(red[0], red[1]), max(red[2], red[3]));
    const uint16_t amax_bits = amax < 0 ? (uint16_t)0x7fffu : (uint16_t)amax;
    const uint16_t s_bits = f2h_bits(h2f(amax_bits) / 127.0f);
    const float s = h2f(s_bits);
    if (tid == 0) sA[row] = s_bits;
    for (int k = tid; k < K; k += QBLOCK) {
        bool z = ZERO && ((reinterpret_cast<const unsigned*>(zmask)[k >> 5] >> (k & 31)) & 1u);
        qA[row * (int64_t)K + k] = (int8_t)quant_one(z ? 0.f : h2f(a[k]), s);
    }
    if (ZERO) {
        __syncthreads();
        for (int j = tid; j < O; j += QBLOCK) {
            int c = ind[j];
            if (c >= 0 && c < K) A[row * (int64_t)K + c] = 0;
        }
    }
}

// Stand-alone gather (ExtractOutliersAndSetToZeros API, kernel/i8gemm.cu:226-244): one wave per row, coalesced writes.
template <bool ZERO>
__global__ __launch_bounds__(QBLOCK) void extract_kernel(uint16_t* __restrict__ A, uint16_t* __restrict__ fpA,
                                                         const int32_t* __restrict__ ind, int M, int K, int O)
{
    const int64_t row = (int64_t)blockIdx.x * (QBLOCK / 64) + (threadIdx.x >> 6);
    if (row >= M) return;
    const int lane = threadIdx.x & 63;
    for (int j = lane; j < O; j += 64) {
        const int c = ind[j];
        fpA[row * (int64_t)O + j] = A[row * (int64_t)K + c];
    }
    if (ZERO) {
        for (int j = lane; j < O; j += 64) A[row * (int64_t)K + ind[j]] = 0;
    }
}


// Int8quantize (quantkernel/mix_cuda/cult.cu:1732-1771): dst = (int8) half2int_rn(hdiv(src, scale[row])) with a
// caller-supplied per-row scale.  Elementwise, 16-byte loads / 8-byte stores, grid-stride.
__global__ __launch_bounds__(QBLOCK) void quant_with_scale_kernel(const uint16_t* __restrict__ src,
                                                                   const uint16_t* __restrict__ scale,
                                                                   int8_t* __restrict__ dst, int M, int K)
{
    const int nvec = K >> 3;
    const int64_t total = (int64_t)M * nvec;
    for (int64_t i = (int64_t)blockIdx.x * QBLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * QBLOCK) {
        const int64_t row = i / nvec;
        const float s = h2f(scale[row]);
        const uint4 v = reinterpret_cast<const uint4*>(src)[i];
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
        unsigned o[2] = {0u, 0u};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int q0 = quant_one(h2f((uint16_t)(w[e] & 0xffffu)), s);
            int q1 = quant_one(h2f((uint16_t)(w[e] >> 16)), s);
            o[e >> 1] |= (unsigned)(q0 | (q1 << 8)) << ((e & 1) * 16);
        }
        reinterpret_cast<uint2*>(dst)[i] = make_uint2(o[0], o[1]);
    }
}

hipError_t launch_quant_with_scale(const void* src, const void* scale, int8_t* dst, int M, int K, hipStream_t st)
{
    if (M <= 0) return hipSuccess;
    const int64_t total = (int64_t)M * (K / 8);
    const int64_t want = (total + QBLOCK - 1) / QBLOCK;
    const unsigned grid = (unsigned)(want < 4096 ? want : 4096);
    hipLaunchKernelGGL(quant_with_scale_kernel, dim3(grid), dim3(QBLOCK), 0, st, static_cast<const uint16_t*>(src),
                       static_cast<const uint16_t*>(scale), dst, M, K);
    return hipGetLastError();
}

static std::atomic<void*> g_quant_stamps{nullptr}; // measurement only (tools/small_m_timeline.py)
void set_quant_stamp_buffer(void* p) { g_quant_stamps.store(p); }

template <int TPR, int MAXV>
static hipError_t launch_qe(uint16_t* A, int8_t* qA, uint16_t* sA, uint16_t* fpA, const int32_t* ind, int M, int K,
                            int O, bool zero, hipStream_t st, unsigned* zw, int frag = 0)
{
    void* const dbg = g_quant_stamps.load(std::memory_order_relaxed);
    if constexpr (TPR == 256 && MAXV <= 8) {
        if (frag == 1) { // decode batches: qA in the skinny GEMM's fragment order (both flavours)
            dim3 grid((unsigned)M), block(QBLOCK);
            if (zero)
                hipLaunchKernelGGL((quant_extract_kernel<TPR, MAXV, true, 1>), grid, block, (size_t)((K + 31) / 32) * 4, st, A, qA, sA,
                                   fpA, ind, M, K, O, zw, dbg);
            else
                hipLaunchKernelGGL((quant_extract_kernel<TPR, MAXV, false, 1>), grid, block, 0, st, A, qA, sA, fpA, ind, M, K, O, zw, dbg);
            return hipGetLastError();
        }
    }
    if (frag != 0) return hipErrorInvalidValue;
    constexpr int RPB = QBLOCK / TPR;
    dim3 grid((unsigned)((M + RPB - 1) / RPB)), block(QBLOCK);
    if (zero) {
        size_t sm = (size_t)((K + 31) / 32) * 4;
        hipLaunchKernelGGL((quant_extract_kernel<TPR, MAXV, true>), grid, block, sm, st, A, qA, sA, fpA, ind, M, K, O, zw, dbg);
    } else {
        hipLaunchKernelGGL((quant_extract_kernel<TPR, MAXV, false>), grid, block, 0, st, A, qA, sA, fpA, ind, M, K, O, zw, dbg);
    }
    return hipGetLastError();
}

// Rows up to which a whole 256-thread block takes ONE row (one load round trip of 1-4 vectors per lane) instead of one wavefront per row (8-16 vectors
// per lane, four rows per block).  Round 3 set it to 64 for decode batches; round 5 measured the rest (profiles/r05_quant_block_rows.txt):
//   * this kernel, K <= 4096 (8 vectors per lane in the wavefront form): operator us cold, 12288 x 4096 at 96 / 128 / 256 / 512 / 1024 rows 28.5 / 29.7 / 34.5 / 42.1 /
//     58.8 -> 26.1 / 28.0 / 32.2 / 40.7 / 57.6; kernel alone at 2048 rows 7.5 -> 6.3 us, level at 8192 / 16384, 16 % SLOWER at 65536 (142.6 -> 165.6): up to 2048 rows;
//   * this kernel, 4096 < K <= 8192 (16 vectors per lane): ahead at every size -- 2048 rows 15.8 -> 9.3 us, 16384 rows 77.0 -> 67.1, 65536 rows 283.6 -> 278.5: always;
//   * the fused RMSNorm producer (norm_kernels.hip): ahead at every size at both widths (K = 4096: 128 rows 10.1 -> 4.6 us, 16384 rows 70.6 -> 61.4, 65536 rows 262.5 -> 254.6;
//     K = 8192: 128 rows 16.7 -> 6.5, 65536 rows 617.8 -> 466.8): always.
// Knobs 1301..1312 force one threshold (64 << n rows) for all three, 1300 = these rules.
static std::atomic<int> g_quant_block_rows{-1};
void set_quant_block_rows(int m) { g_quant_block_rows.store(m); }
int quant_block_rows(int nvec, bool norm_producer)
{
    const int f = g_quant_block_rows.load(std::memory_order_relaxed);
    if (f >= 0) return f;
    return norm_producer || nvec > 256 * 2 ? INT32_MAX : 2048;
}
bool quant_frag_layout_supported(int M, int K) { return M > 0 && M <= 64 && K % 8 == 0 && K / 8 > 64 * 2 && K / 8 <= 256 * 8; }

hipError_t launch_quant_extract(void* A, int8_t* qA, void* sA, void* fpA, const int32_t* ind, int M, int K, int O,
                                bool zero, hipStream_t st, void* zero_words, int frag)
{
    if (frag == 1 && !quant_frag_layout_supported(M, K)) return hipErrorInvalidValue;
    if (frag < 0 || frag > 1) return hipErrorInvalidValue;
    if (M <= 0) return hipSuccess;
    unsigned* const zw = static_cast<unsigned*>(zero_words);
    uint16_t* a = static_cast<uint16_t*>(A);
    uint16_t* s = static_cast<uint16_t*>(sA);
    uint16_t* f = static_cast<uint16_t*>(fpA);
    const int nvec = K / 8;
    // Decode batches (few rows: the launch is a chain of latencies, not a stream): a whole 256-thread block per row, so
    // that a row is ONE load round trip of 1-4 vectors per lane instead of 8-16 on a single wavefront
    if ((M <= quant_block_rows(nvec, false) || frag == 1) && nvec > 64 * 2) {
        if (nvec <= 256 * 2) return launch_qe<256, 2>(a, qA, s, f, ind, M, K, O, zero, st, zw, frag);
        if (nvec <= 256 * 4) return launch_qe<256, 4>(a, qA, s, f, ind, M, K, O, zero, st, zw, frag);
    }
    if (nvec <= 64 * 2) return launch_qe<64, 2>(a, qA, s, f, ind, M, K, O, zero, st, zw, frag);
    if (nvec <= 64 * 4) return launch_qe<64, 4>(a, qA, s, f, ind, M, K, O, zero, st, zw, frag);
    if (nvec <= 64 * 8) return launch_qe<64, 8>(a, qA, s, f, ind, M, K, O, zero, st, zw, frag);
    if (nvec <= 64 * 16) return launch_qe<64, 16>(a, qA, s, f, ind, M, K, O, zero, st, zw, frag);
    if (nvec <= 256 * 8) return launch_qe<256, 8>(a, qA, s, f, ind, M, K, O, zero, st, zw, frag);
    if (nvec <= 256 * 16) return launch_qe<256, 16>(a, qA, s, f, ind, M, K, O, zero, st, zw, frag);
    dim3 grid((unsigned)M), block(QBLOCK);
    if (zero) {
        size_t sm = (size_t)((K + 31) / 32) * 4;
        hipLaunchKernelGGL((quant_extract_long_kernel<true>), grid, block, sm, st, a, qA, s, f, ind, M, K, O, zw);
    } else {
        hipLaunchKernelGGL((quant_extract_long_kernel<false>), grid, block, 0, st, a, qA, s, f, ind, M, K, O, zw);
    }
    return hipGetLastError();
}

hipError_t launch_extract(void* A, void* fpA, const int32_t* ind, int M, int K, int O, bool zero, hipStream_t st)
{
    if (M <= 0 || O <= 0) return hipSuccess;
    dim3 grid((unsigned)((M + 3) / 4)), block(QBLOCK);
    if (zero)
        hipLaunchKernelGGL((extract_kernel<true>), grid, block, 0, st, static_cast<uint16_t*>(A),
                           static_cast<uint16_t*>(fpA), ind, M, K, O);
    else
        hipLaunchKernelGGL((extract_kernel<false>), grid, block, 0, st, static_cast<uint16_t*>(A),
                           static_cast<uint16_t*>(fpA), ind, M, K, O);
    return hipGetLastError();
}

} // namespace mixq

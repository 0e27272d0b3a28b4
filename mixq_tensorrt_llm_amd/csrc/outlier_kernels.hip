// Dynamic outlier detection of the P-flavour forward (MixQ/src/mixquant/modules/linear.py:155-161, 201-223) on gfx950.
//
//   FindOutliers(A)          = torch.unique(torch.where(A.abs() > sigma)[1])     sorted column indices, int32
//   weight_cache (new cols)  = q_weight[:, ind].to(float16) * scale_col.T          fp16 [N, len]
//
// The reference runs both as chains of PyTorch ops (abs, compare, where -> M*K index pairs, unique = sort + compact,
// advanced indexing, cast, multiply).  Here: one streaming pass over A that ORs per-column flags into a K-bit mask
// (HBM-bound: 2*M*K bytes read once, 16-byte loads), a one-block compaction of the mask into sorted indices, and a
// gather-dequant kernel for the weight columns.
#include "mixq_device.h"
#include "mixq_launch.h"

namespace mixq {

constexpr int OBLOCK = 256;

// Each thread owns 8 consecutive columns (one 16-byte vector) and walks down `rows_per_block` rows; column flags are
// merged through LDS and pushed to the global mask with at most one atomicOr per 32 columns per block.
__global__ __launch_bounds__(OBLOCK) void outlier_flags_kernel(const uint16_t* __restrict__ A, unsigned* __restrict__ mask,
                                                               int M, int K, float sigma, int rows_per_block)
{
    const int nvec = K >> 3;
    const int r0 = blockIdx.y * rows_per_block;
    const int r1 = min(M, r0 + rows_per_block);
    for (int v = blockIdx.x * OBLOCK + threadIdx.x; v < nvec; v += gridDim.x * OBLOCK) {
        unsigned flags = 0u; // bit e = column v*8+e holds an outlier in rows [r0, r1)
        const uint4* src = reinterpret_cast<const uint4*>(A) + v;
        for (int r = r0; r < r1; ++r) {
            const uint4 x = src[(int64_t)r * nvec];
            const unsigned w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                // |a| > sigma in fp16 == in fp32 (both operands are fp16 values); NaN compares false like torch
                flags |= (__builtin_fabsf(h2f((uint16_t)(w[e] & 0xffffu))) > sigma ? 1u : 0u) << (2 * e);
                flags |= (__builtin_fabsf(h2f((uint16_t)(w[e] >> 16))) > sigma ? 1u : 0u) << (2 * e + 1);
            }
        }
        if (flags) atomicOr(mask + (v >> 2), flags << ((v & 3) * 8));
    }
}

// One block: mask (K bits) -> ascending indices + count.  words = ceil(K/32) (any size: chunked prefix sums).
__global__ __launch_bounds__(1024) void outlier_compact_kernel(const unsigned* __restrict__ mask, int K,
                                                               int32_t* __restrict__ ind, int32_t* __restrict__ count,
                                                               int capacity)
{
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int words = (K + 31) >> 5;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int w0 = 0; w0 < words; w0 += 1024) {
        const int w = w0 + tid;
        unsigned m = w < words ? mask[w] : 0u;
        if (w == words - 1 && (K & 31)) m &= (1u << (K & 31)) - 1u; // bits past K never count
        const int c = __builtin_popcount(m);
        int incl = c; // inclusive prefix sum inside the wave
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = base_s;
        for (int i = 0; i < wave; ++i) before += wsum[i];
        int pos = before + incl - c;
        while (m) {
            const int b = __builtin_ctz(m);
            m &= m - 1;
            if (pos < capacity) ind[pos] = w * 32 + b;
            ++pos;
        }
        __syncthreads();
        if (tid == 1023) base_s = before + incl;
        __syncthreads();
    }
    if (tid == 0) *count = base_s;
}

// out[n, j] = fp16( fp16(W[n, ind[j]]) * sW[n] ): one thread per element, rows of `out` contiguous.
__global__ __launch_bounds__(OBLOCK) void dequant_columns_kernel(const int8_t* __restrict__ W,
                                                                 const uint16_t* __restrict__ sW,
                                                                 const int32_t* __restrict__ ind, int len,
                                                                 uint16_t* __restrict__ out, int N, int K)
{
    const int64_t total = (int64_t)N * len;
    for (int64_t i = (int64_t)blockIdx.x * OBLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * OBLOCK) {
        const int n = (int)(i / len), j = (int)(i - (int64_t)n * len);
        const int c = ind[j];
        const float w = (c >= 0 && c < K) ? (float)W[(int64_t)n * K + c] : 0.f;
        out[i] = f2h_bits(w * h2f(sW[n])); // int8 is exact in fp16; one rounding of the exact product
    }
}

hipError_t launch_find_outliers(const void* A, int M, int K, float sigma, unsigned* mask, int32_t* ind, int32_t* count,
                                int capacity, hipStream_t st)
{
    const int words = (K + 31) / 32;
    hipError_t e = hipMemsetAsync(mask, 0, (size_t)words * 4, st);
    if (e != hipSuccess) return e;
    if (M > 0) {
        const int nvec = K / 8;
        const int gx = (nvec + OBLOCK - 1) / OBLOCK;
        // ~4096 workgroups: enough to fill 256 CUs several times over, few enough to keep the atomics rare
        int rows_per_block = (int)(((int64_t)M * gx + 4095) / 4096);
        if (rows_per_block < 8) rows_per_block = 8;
        const int gy = (M + rows_per_block - 1) / rows_per_block;
        hipLaunchKernelGGL(outlier_flags_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(OBLOCK), 0, st,
                           static_cast<const uint16_t*>(A), mask, M, K, sigma, rows_per_block);
    }
    hipLaunchKernelGGL(outlier_compact_kernel, dim3(1), dim3(1024), 0, st, mask, K, ind, count, capacity);
    return hipGetLastError();
}

hipError_t launch_dequant_columns(const int8_t* W, const void* sW, const int32_t* ind, int len, void* out, int N, int K,
                                  hipStream_t st)
{
    if (N <= 0 || len <= 0) return hipSuccess;
    const int64_t total = (int64_t)N * len;
    const int64_t want = (total + OBLOCK - 1) / OBLOCK;
    hipLaunchKernelGGL(dequant_columns_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(OBLOCK), 0, st, W,
                       static_cast<const uint16_t*>(sW), ind, len, static_cast<uint16_t*>(out), N, K);
    return hipGetLastError();
}

} // namespace mixq

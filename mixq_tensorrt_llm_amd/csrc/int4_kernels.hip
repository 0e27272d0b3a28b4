// 4-bit (W4A4) flavour of the MixQ linear on gfx950 -- the `bit == 4` branch of MixQ/src/mixquant/modules/linear.py.
//
// Replaces (reference, CUDA):
//   quantkernel/mix_cuda/cult.cu:2515-2567  FindRowScaleKernel4bit   s = fp16(amax / 7), q = int4(rn(x / s)), packed pairs
//   quantkernel/mix_cuda/cult.cu:2005-2052  int4FusedDequantizeCUDA  CUTLASS s4 x s4 -> s32 GEMM + dequant epilogue
//   quantkernel/mix_cuda/cult.cu:3020-3043  unpack_int4_to_fp16_kernel  weight columns -> fp16 (for the outlier weights)
//
// CDNA4 has no int4 MFMA: 4-bit operands are a STORAGE format here.  The GEMM unpacks both operands to int8 (sign
// extension, an HBM-bound pass of 1.5 bytes per element) and runs the int8 MFMA kernels unchanged -- integer
// accumulation is exact, so the int32 results equal the s4 x s4 tensor-core results bit for bit.
// Packing (cutlass::int4b_t in a byte): element 2i = low nibble, element 2i+1 = high nibble, two's complement.
#include "mixq_device.h"
#include "mixq_launch.h"

namespace mixq {

constexpr int I4BLOCK = 256;

// One 256-thread block per row (like the reference); the row is read twice (second pass from L2), 16-byte loads.
__global__ __launch_bounds__(I4BLOCK) void quant4_rows_kernel(const uint16_t* __restrict__ A, uint8_t* __restrict__ q,
                                                              uint16_t* __restrict__ sA, int M, int K)
{
    __shared__ int red[I4BLOCK / 64];
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;
    const int nvec = K >> 3;
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(A + row * (int64_t)K);
    int amax = -1; // integer max over |x| bit patterns, NaN dropped like __hmax (-1: every element NaN)
    for (int v = tid; v < nvec; v += I4BLOCK) {
        const uint4 x = src[v];
        const unsigned w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int lo = (int)(w[e] & 0x7fffu), hi = (int)((w[e] >> 16) & 0x7fffu);
            lo = lo > 0x7c00 ? -1 : lo;
            hi = hi > 0x7c00 ? -1 : hi;
            amax = max(amax, max(lo, hi));
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = max(amax, __shfl_xor(amax, off, 64));
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = max(max(red[0], red[1]), max(red[2], red[3]));
    const uint16_t amax_bits = amax < 0 ? (uint16_t)0x7fffu : (uint16_t)amax;
    const uint16_t s_bits = f2h_bits(h2f(amax_bits) / 7.0f); // __hdiv(max, 7.0)
    const float s = h2f(s_bits);
    if (tid == 0) sA[row] = s_bits;
    unsigned* __restrict__ dst = reinterpret_cast<unsigned*>(q + row * (int64_t)(K >> 1));
    for (int v = tid; v < nvec; v += I4BLOCK) {
        const uint4 x = src[v];
        const unsigned w[4] = {x.x, x.y, x.z, x.w};
        unsigned o = 0u;
#pragma unroll
        for (int e = 0; e < 4; ++e) { // int4b_t(int) keeps the low 4 bits of __half2int_rn(__hdiv(x, s))
            const unsigned q0 = (unsigned)quant_one(h2f((uint16_t)(w[e] & 0xffffu)), s) & 0xfu;
            const unsigned q1 = (unsigned)quant_one(h2f((uint16_t)(w[e] >> 16)), s) & 0xfu;
            o |= (q0 | (q1 << 4)) << (8 * e);
        }
        dst[v] = o;
    }
}

// packed int4 -> int8 (sign-extended): 16 packed bytes in, 32 bytes out per thread step.
__global__ __launch_bounds__(I4BLOCK) void unpack_s4_kernel(const uint8_t* __restrict__ src, int8_t* __restrict__ dst,
                                                            int64_t nvec /* 16-byte input vectors */)
{
    for (int64_t i = (int64_t)blockIdx.x * I4BLOCK + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * I4BLOCK) {
        const uint4 x = reinterpret_cast<const uint4*>(src)[i];
        const unsigned w[4] = {x.x, x.y, x.z, x.w};
        unsigned o[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // bytes b0..b3 of w -> nibbles; lo nibble first.  Sign-extend each nibble inside its own byte:
            const unsigned lo = w[e] & 0x0f0f0f0fu, hi = (w[e] >> 4) & 0x0f0f0f0fu;
            const unsigned los = lo | ((lo & 0x08080808u) * 0x1eu); // per byte: x | (x & 8 ? 0xf0 : 0); 8 * 30 = 240
            const unsigned his = hi | ((hi & 0x08080808u) * 0x1eu); //   stays inside its byte, so no carries
            // interleave: out bytes = lo0 hi0 lo1 hi1 | lo2 hi2 lo3 hi3
            o[2 * e] = __builtin_amdgcn_perm(his, los, 0x05010400u);
            o[2 * e + 1] = __builtin_amdgcn_perm(his, los, 0x07030602u);
        }
        uint4* d = reinterpret_cast<uint4*>(dst) + 2 * i;
        d[0] = make_uint4(o[0], o[1], o[2], o[3]);
        d[1] = make_uint4(o[4], o[5], o[6], o[7]);
    }
}

// unpack_int4_to_fp16 (cult.cu:3020-3043): out[r, c] = half(int4 value of weight[r, ind[c]]), weight [rows, cols_packed].
__global__ __launch_bounds__(I4BLOCK) void unpack_s4_columns_kernel(const uint8_t* __restrict__ weight,
                                                                    const int32_t* __restrict__ ind, int rows,
                                                                    int cols_packed, int n, uint16_t* __restrict__ out)
{
    const int64_t total = (int64_t)rows * n;
    for (int64_t i = (int64_t)blockIdx.x * I4BLOCK + threadIdx.x; i < total; i += (int64_t)gridDim.x * I4BLOCK) {
        const int r = (int)(i / n), c = (int)(i - (int64_t)r * n);
        const int col = ind[c];
        int v = 0;
        if (col >= 0 && (col >> 1) < cols_packed) {
            const unsigned b = weight[(int64_t)r * cols_packed + (col >> 1)];
            const int nib = (col & 1) ? (int)(b >> 4) : (int)(b & 0xfu);
            v = (nib ^ 8) - 8;
        }
        out[i] = f2h_bits((float)v);
    }
}

hipError_t launch_quant4_rows(const void* A, uint8_t* q, void* sA, int M, int K, hipStream_t st)
{
    if (M <= 0) return hipSuccess;
    hipLaunchKernelGGL(quant4_rows_kernel, dim3((unsigned)M), dim3(I4BLOCK), 0, st, static_cast<const uint16_t*>(A), q,
                       static_cast<uint16_t*>(sA), M, K);
    return hipGetLastError();
}

hipError_t launch_unpack_s4(const uint8_t* src, int8_t* dst, size_t packed_bytes, hipStream_t st)
{
    if (packed_bytes == 0) return hipSuccess;
    const int64_t nvec = (int64_t)(packed_bytes / 16);
    const int64_t want = (nvec + I4BLOCK - 1) / I4BLOCK;
    hipLaunchKernelGGL(unpack_s4_kernel, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(I4BLOCK), 0, st, src, dst,
                       nvec);
    return hipGetLastError();
}

hipError_t launch_unpack_s4_columns(const uint8_t* weight, const int32_t* ind, int rows, int cols_packed, int n,
                                    void* out, hipStream_t st)
{
    if (rows <= 0 || n <= 0) return hipSuccess;
    const int64_t total = (int64_t)rows * n;
    const int64_t want = (total + I4BLOCK - 1) / I4BLOCK;
    hipLaunchKernelGGL(unpack_s4_columns_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(I4BLOCK), 0, st,
                       weight, ind, rows, cols_packed, n, static_cast<uint16_t*>(out));
    return hipGetLastError();
}

} // namespace mixq

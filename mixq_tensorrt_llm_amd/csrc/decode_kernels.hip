// W8A16 decode path (M <= 4 tokens) for gfx950: Out[m,n] = sum_k A[m,k] * fp16((q[k,n]-128) * scale[n]).
//
// Replaces (reference, CUDA): weightonlykernel/fpA_intB_gemm_wrapper.cu:29-58 -> weightOnlyBatchedGemv/kernel.h:300-470
// (Int8b, per-channel).  The weight operand is consumed IN THE REFERENCE'S ON-DISK LAYOUT (EETQ/FasterTransformer
// interleave, weightonlykernel/cutlass_kernels/cutlass_preprocessors.cc:497-534) so existing checkpoints load as-is:
//   byte[(n/2)*2K + tb*128 + (n%2)*64 + x] = q[k][n] + 128,
//       k = 64*tb + 16*(x'/16) + P[x'%16],  x' = x with bits 0 and 1 swapped,  P = {0,1,8,9,2,3,10,11,4,5,12,13,6,7,14,15}
// i.e. each column pair owns 2K contiguous bytes -> one wavefront streams a pair with 16-byte loads, perfectly
// coalesced (1 KiB per instruction), every weight byte read exactly once: HBM-bound on K*N bytes.
// The permutation never has to be undone in memory: a dot product does not care about summation order, so the 16
// activations of a k-group are loaded contiguously and paired with the weight bytes through a compile-time index map.
#include "mixq_device.h"
#include "mixq_launch.h"

namespace mixq {

// position x (0..15) inside a 16-byte group  ->  k offset inside the 16-row group
__device__ __forceinline__ constexpr int kmap16(int x)
{
    const int xs = (x & ~3) | ((x & 1) << 1) | ((x & 2) >> 1); // undo the byte 1<->2 swap
    const int P[16] = {0, 1, 8, 9, 2, 3, 10, 11, 4, 5, 12, 13, 6, 7, 14, 15};
    return P[xs];
}

template <int MB>
__global__ __launch_bounds__(256) void w8a16_gemv_kernel(const uint16_t* __restrict__ A,
                                                          const uint8_t* __restrict__ Wq,
                                                          const uint16_t* __restrict__ scale,
                                                          uint16_t* __restrict__ Out, int m_rows, int N, int K)
{
    const int lane = threadIdx.x & 63;
    const int pair = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pair * 2 >= N) return;
    const int col = (lane >> 2) & 1;          // which column of the pair this lane's 16 bytes belong to
    const int n = pair * 2 + col;
    const float sc = h2f(scale[n]);
    const uint8_t* wrow = Wq + (int64_t)pair * 2 * K;

    float acc[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) acc[m] = 0.f;

    // one iteration = 1 KiB of the pair's 2K bytes = 8 blocks of (64 B col0 | 64 B col1) = 512 k values
    for (int byte0 = lane * 16; byte0 < 2 * K; byte0 += 64 * 16) {
        const uint4 wv = *reinterpret_cast<const uint4*>(wrow + byte0);
        const int tb = byte0 >> 7;
        const int kbase = tb * 64 + (byte0 & 63); // (byte0 & 63) is a multiple of 16: the 16-row group
        const unsigned w[4] = {wv.x, wv.y, wv.z, wv.w};
        float wf[16];
#pragma unroll
        for (int x = 0; x < 16; ++x) {
            const int q = (int)((w[x >> 2] >> ((x & 3) * 8)) & 0xffu) - 128;
            // fp16( (q-128) * scale ): the reference's hfma2(v, scale, 0) -- exact product, one RNE to fp16
            wf[x] = h2f(f2h_bits((float)q * sc));
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if (m < m_rows) {
                const uint4 a0 = *reinterpret_cast<const uint4*>(A + (int64_t)m * K + kbase);
                const uint4 a1 = *reinterpret_cast<const uint4*>(A + (int64_t)m * K + kbase + 8);
                const unsigned aw[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                float s = acc[m];
#pragma unroll
                for (int x = 0; x < 16; ++x) {
                    const int k = kmap16(x);
                    const float a = h2f((uint16_t)((aw[k >> 1] >> ((k & 1) * 16)) & 0xffffu));
                    s = __builtin_fmaf(a, wf[x], s);
                }
                acc[m] = s;
            }
        }
    }
    // reduce over the lanes that share a column: xor over lane bits {0,1,3,4,5} (bit 2 selects the column)
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        float s = acc[m];
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 8, 64);
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if ((lane == 0 || lane == 4) && m < m_rows && n < N) Out[(int64_t)m * N + n] = f2h_bits(s);
    }
}

hipError_t launch_w8a16(const void* A, const uint8_t* Wq, const void* scale, void* Out, int M, int N, int K,
                        hipStream_t st)
{
    if (M <= 0 || N <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((N / 2 + 3) / 4);
    for (int m0 = 0; m0 < M; m0 += 4) { // the reference only takes this path for M <= 4 (SMALL_M_FAST_PATH)
        const int mb = (M - m0) < 4 ? (M - m0) : 4;
        hipLaunchKernelGGL((w8a16_gemv_kernel<4>), dim3(grid), dim3(256), 0, st,
                           static_cast<const uint16_t*>(A) + (int64_t)m0 * K, Wq, static_cast<const uint16_t*>(scale),
                           static_cast<uint16_t*>(Out) + (int64_t)m0 * N, mb, N, K);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

} // namespace mixq

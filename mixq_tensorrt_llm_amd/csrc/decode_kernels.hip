// W8A16 decode path (M <= 4 tokens) for gfx950: Out[m,n] = sum_k A[m,k] * fp16((q[k,n]-128) * scale[n]).
//
// Replaces (reference, CUDA): weightonlykernel/fpA_intB_gemm_wrapper.cu:29-58 -> weightOnlyBatchedGemv/kernel.h:300-470
// (Int8b, per-channel).  The weight operand is consumed IN THE REFERENCE'S ON-DISK LAYOUT (EETQ/FasterTransformer
// interleave, weightonlykernel/cutlass_kernels/cutlass_preprocessors.cc:497-534) so existing checkpoints load as-is:
//   byte[(n/2)*2K + tb*128 + (n%2)*64 + x] = q[k][n] + 128,
//       k = 64*tb + 16*(x'/16) + P[x'%16],  x' = x with bits 0 and 1 swapped,  P = {0,1,8,9,2,3,10,11,4,5,12,13,6,7,14,15}
// i.e. each column pair owns 2K contiguous bytes, streamed with 16-byte loads (1 KiB per wave instruction), every
// weight byte read exactly once: HBM-bound on K*N bytes.
//
// The permutation is never undone in memory.  Inside one 32-bit word of a 16-row group the four bytes belong to
// k = (k0, k0+8, k0+1, k0+9): bytes {0,2} pair with the adjacent activations (k0, k0+1) and bytes {1,3} with
// (k0+8, k0+9).  One v_perm_b32 per pair builds the fp16 numbers 1024+b (0x6400|b) -- exact -- v_pk_add_f16 (-1152)
// gives q-128 exactly, v_pk_mul_f16 by the column scale gives fp16((q-128)*scale) with one RNE, the same value the
// reference's hfma2(v, scale, 0) produces (kernel.h:367-369); v_dot2_f32_f16 accumulates the pair in fp32.
//
// WPP waves share one column pair (each takes every WPP-th KiB) and combine through LDS, so that small N still puts
// enough waves on the chip; PPW column pairs per wave share each activation load (M > 1).
#include "mixq_device.h"
#include "mixq_launch.h"

namespace mixq {

__device__ __forceinline__ v2h dequant_pair(unsigned w, unsigned sel, v2h scale2)
{
    const unsigned h = __builtin_amdgcn_perm(0x64646464u, w, sel); // two fp16: 1024 + byte
    v2h v;
    __builtin_memcpy(&v, &h, 4);
    const v2h bias = {(_Float16)-1152.0f, (_Float16)-1152.0f};
    return (v + bias) * scale2; // exact subtract, one RNE in the multiply
}

// PPW column pairs per wave share every activation load (for M > 1 the activation re-reads through the L1/TA path,
// not HBM, are what limits a one-pair-per-wave kernel); WPP waves share the K range of their pairs.
template <int MB, int WPP, int PPW>
__global__ __launch_bounds__(256) void w8a16_gemv_kernel(const uint16_t* __restrict__ A,
                                                          const uint8_t* __restrict__ Wq,
                                                          const uint16_t* __restrict__ scale,
                                                          uint16_t* __restrict__ Out, int N, int K)
{
    constexpr int GPB = 4 / WPP; // pair groups per block
    __shared__ float red[4][PPW][MB][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pair0 = (blockIdx.x * GPB + wave / WPP) * PPW; // first column pair of this wave's group
    const int part = wave % WPP;
    const int col = (lane >> 2) & 1; // which column of a pair this lane's 16 bytes belong to
    const int npairs = N / 2;
    v2h scale2[PPW];
    const uint8_t* wrow[PPW];
#pragma unroll
    for (int q = 0; q < PPW; ++q) {
        const int pr = min(pair0 + q, npairs - 1); // clamped pairs are computed but never stored
        const uint16_t sb = scale[pr * 2 + col];
        _Float16 sc;
        __builtin_memcpy(&sc, &sb, 2);
        scale2[q] = v2h{sc, sc};
        wrow[q] = Wq + (int64_t)pr * 2 * K;
    }

    float acc[PPW][MB];
#pragma unroll
    for (int q = 0; q < PPW; ++q)
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[q][m] = 0.f;

    // one step = 1 KiB of each pair's 2K bytes = 8 blocks of (64 B col0 | 64 B col1); steps are dealt round-robin
    const int nsteps = (2 * K) >> 10;
    auto step = [&](int byte0) __attribute__((always_inline)) {
        uint4 wv[PPW];
#pragma unroll
        for (int q = 0; q < PPW; ++q) wv[q] = *reinterpret_cast<const uint4*>(wrow[q] + byte0);
        const int kbase = (byte0 >> 7) * 64 + (byte0 & 63); // 16-row group of this lane's 16 bytes
        unsigned aw[MB][8];
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const uint4 a0 = *reinterpret_cast<const uint4*>(A + (int64_t)m * K + kbase);
            const uint4 a1 = *reinterpret_cast<const uint4*>(A + (int64_t)m * K + kbase + 8);
            aw[m][0] = a0.x, aw[m][1] = a0.y, aw[m][2] = a0.z, aw[m][3] = a0.w;
            aw[m][4] = a1.x, aw[m][5] = a1.y, aw[m][6] = a1.z, aw[m][7] = a1.w;
        }
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const unsigned w[4] = {wv[q].x, wv[q].y, wv[q].z, wv[q].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v2h wl = dequant_pair(w[j], 0x04020400u, scale2[q]); // bytes 0,2 -> k = 2j, 2j+1
                const v2h wh = dequant_pair(w[j], 0x04030401u, scale2[q]); // bytes 1,3 -> k = 8+2j, 9+2j
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    v2h al, ah;
                    __builtin_memcpy(&al, &aw[m][j], 4);
                    __builtin_memcpy(&ah, &aw[m][4 + j], 4);
                    acc[q][m] = __builtin_amdgcn_fdot2(wl, al, acc[q][m], false);
                    acc[q][m] = __builtin_amdgcn_fdot2(wh, ah, acc[q][m], false);
                }
            }
        }
    };
    {
        int st = part;
        for (; st < nsteps; st += WPP) step((st << 10) + lane * 16);
        // K % 512 != 0: a last partial KiB (K is a multiple of 64, so whole 128-byte blocks)
        const int tail0 = nsteps << 10;
        if (part == 0 && tail0 + lane * 16 < 2 * K) step(tail0 + lane * 16);
    }
    // reduce over the lanes that share a column: xor over lane bits {0,1,3,4,5} (bit 2 selects the column)
#pragma unroll
    for (int q = 0; q < PPW; ++q)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            float s = acc[q][m];
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 8, 64);
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            acc[q][m] = s;
        }
    if (WPP > 1) {
        if (lane == 0 || lane == 4)
#pragma unroll
            for (int q = 0; q < PPW; ++q)
#pragma unroll
                for (int m = 0; m < MB; ++m) red[wave][q][m][col] = acc[q][m];
        __syncthreads();
    }
    if (part == 0 && (lane == 0 || lane == 4)) {
#pragma unroll
        for (int q = 0; q < PPW; ++q) {
            const int n = (pair0 + q) * 2 + col;
            if (pair0 + q < npairs) {
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    float s = acc[q][m];
                    if (WPP > 1) {
                        s = 0.f;
#pragma unroll
                        for (int w2 = 0; w2 < WPP; ++w2) s += red[wave + w2][q][m][col];
                    }
                    Out[(int64_t)m * N + n] = f2h_bits(s);
                }
            }
        }
    }
}

template <int MB, int PPW>
static hipError_t launch_mb(const uint16_t* A, const uint8_t* Wq, const uint16_t* scale, uint16_t* Out, int N, int K,
                            hipStream_t st)
{
    const int groups = (N / 2 + PPW - 1) / PPW;
    // aim for >= 4096 waves while keeping >= 2 KiB of each pair's weights per wave
    int wpp = 1;
    while (wpp < 4 && groups * wpp < 4096 && (2 * K) / (wpp * 2) >= 2048) wpp *= 2;
    if (wpp == 4) {
        hipLaunchKernelGGL((w8a16_gemv_kernel<MB, 4, PPW>), dim3((unsigned)groups), dim3(256), 0, st, A, Wq, scale, Out, N, K);
    } else if (wpp == 2) {
        hipLaunchKernelGGL((w8a16_gemv_kernel<MB, 2, PPW>), dim3((unsigned)((groups + 1) / 2)), dim3(256), 0, st, A, Wq,
                           scale, Out, N, K);
    } else {
        hipLaunchKernelGGL((w8a16_gemv_kernel<MB, 1, PPW>), dim3((unsigned)((groups + 3) / 4)), dim3(256), 0, st, A, Wq,
                           scale, Out, N, K);
    }
    return hipGetLastError();
}

hipError_t launch_w8a16(const void* A, const uint8_t* Wq, const void* scale, void* Out, int M, int N, int K,
                        hipStream_t st)
{
    if (M <= 0 || N <= 0) return hipSuccess;
    const uint16_t* a = static_cast<const uint16_t*>(A);
    const uint16_t* s = static_cast<const uint16_t*>(scale);
    uint16_t* o = static_cast<uint16_t*>(Out);
    for (int m0 = 0; m0 < M; m0 += 4) { // the reference only takes this path for M <= 4 (SMALL_M_FAST_PATH)
        const int mb = (M - m0) < 4 ? (M - m0) : 4;
        hipError_t e;
        switch (mb) {
        case 1: e = launch_mb<1, 1>(a + (int64_t)m0 * K, Wq, s, o + (int64_t)m0 * N, N, K, st); break;
        case 2: e = launch_mb<2, 4>(a + (int64_t)m0 * K, Wq, s, o + (int64_t)m0 * N, N, K, st); break;
        case 3: e = launch_mb<3, 4>(a + (int64_t)m0 * K, Wq, s, o + (int64_t)m0 * N, N, K, st); break;
        default: e = launch_mb<4, 4>(a + (int64_t)m0 * K, Wq, s, o + (int64_t)m0 * N, N, K, st); break;
        }
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

} // namespace mixq

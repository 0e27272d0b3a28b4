// Fused RMSNorm -> outlier extraction -> per-token int8 quantisation for gfx950 (the producer on
// the input side of the MixQ linear; removes one full pass over the activations compared with norm + quant).
//
// Replaces (reference, CUDA): quantkernel/mix_cuda/layernorm/layernorm.cu:122-198 generalT5LayerNorm_extract_outliers
// and :41-98 generalT5LayerNorm (mixlib ops layernorm_forward_cuda_extract_outliers / layernorm_forward_cuda):
//   rstd = rsqrt( sum x^2 / n + eps )                  fp32
//   out  = fp16( clamp( (x * rstd) * gamma, +-64504 ) )
//   outliers[j] = out[ind[j]] ; out[ind[j]] = 0        (P-flavour: outliers leave the int8 path)
//   scale = fp16(amax/127) ; q = int8(rn(out/scale))   same arithmetic as the stand-alone quantiser
//
// One wavefront (K <= 8192) or one 256-thread block (K <= 32768) owns a row and keeps it in registers: x is read once
// (16-byte loads), out / q / outliers / scale are written once.  The normalised row passes through LDS only so that
// the <= 128 outlier values can be picked by column index; the outlier columns are zeroed through a K-bit mask.
#include "mixq_device.h"
#include "mixq_launch.h"

namespace mixq {

constexpr int NBLOCK = 256;

// QUANT: 0 = plain RMSNorm, 8 = int8 rows (scale amax/127), 4 = packed int4 rows (scale amax/7, layernorm.cu:201-290)
// FRAG (QUANT == 8, decode batches): the int8 rows go out in the skinny GEMM's fragment-major order (quant_kernels.hip FRAG).
template <int TPR, int MAXV, int QUANT, bool FRAG = false>
__global__ __launch_bounds__(NBLOCK) void rmsnorm_quant_kernel(const uint16_t* __restrict__ X,
                                                               const uint16_t* __restrict__ gamma,
                                                               uint16_t* __restrict__ out, uint16_t* __restrict__ outl,
                                                               const int32_t* __restrict__ ind, int8_t* __restrict__ q,
                                                               uint16_t* __restrict__ scale, float eps, int M, int K,
                                                               int O)
{
    constexpr int RPB = NBLOCK / TPR;
    __shared__ float redf[NBLOCK / 64];
    __shared__ int redi[NBLOCK / 64];
    extern __shared__ __attribute__((aligned(16))) unsigned char dyn[]; // [mask: K bits, 16-B padded][RPB rows x K fp16]
    const int tid = threadIdx.x;
    const int t = tid % TPR;
    const int rslot = tid / TPR;
    const int64_t row = (int64_t)blockIdx.x * RPB + rslot;
    const bool row_ok = row < M;
    const int nvec = K >> 3;
    const int mask_bytes = ((K + 127) / 128) * 16;
    unsigned* zmask = reinterpret_cast<unsigned*>(dyn);
    uint16_t* lrow = reinterpret_cast<uint16_t*>(dyn + mask_bytes) + (size_t)rslot * K;

    // the row (and, for short rows, gamma) is requested FIRST: its round trip runs under the outlier-mask phase below (a
    // decode batch is a chain of latencies: the mask phase alone is an `ind` round trip + two barriers)
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(X + (row_ok ? row : 0) * (int64_t)K);
    const uint4* __restrict__ g4 = reinterpret_cast<const uint4*>(gamma);
    uint4 x[MAXV];
    constexpr bool GPRE = MAXV <= 4; // gamma prefetched into registers (16 more VGPRs at most)
    uint4 gpre[GPRE ? MAXV : 1];
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
        const int idx = v * TPR + t;
        x[v] = (row_ok && idx < nvec) ? src[idx] : make_uint4(0u, 0u, 0u, 0u);
        if (GPRE) gpre[v] = idx < nvec ? g4[idx] : make_uint4(0u, 0u, 0u, 0u);
    }
    if (QUANT) {
        for (int i = tid; i < mask_bytes / 4; i += NBLOCK) zmask[i] = 0u;
        __syncthreads();
        for (int j = tid; j < O; j += NBLOCK) {
            const int c = ind[j];
            if (c >= 0 && c < K) atomicOr(zmask + (c >> 5), 1u << (c & 31));
        }
        __syncthreads();
    }

    float ss = 0.f;
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
        const unsigned w[4] = {x[v].x, x[v].y, x[v].z, x[v].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = h2f((uint16_t)(w[e] & 0xffffu)), b = h2f((uint16_t)(w[e] >> 16));
            ss = __builtin_fmaf(a, a, ss);
            ss = __builtin_fmaf(b, b, ss);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);
    if (TPR > 64) {
        if ((tid & 63) == 0) redf[tid >> 6] = ss;
        __syncthreads();
        ss = (redf[0] + redf[1]) + (redf[2] + redf[3]);
    }
    const float rstd = 1.0f / __builtin_sqrtf(ss / (float)K + eps);

    // normalise in place (registers now hold the fp16 result), stage the row in LDS for the gather
    unsigned m2 = 0u;
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
        const int idx = v * TPR + t;
        if (idx < nvec) {
            const uint4 gv = GPRE ? gpre[v] : g4[idx];
            const unsigned w[4] = {x[v].x, x[v].y, x[v].z, x[v].w};
            const unsigned gw[4] = {gv.x, gv.y, gv.z, gv.w};
            unsigned o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float a = (h2f((uint16_t)(w[e] & 0xffffu)) * rstd) * h2f((uint16_t)(gw[e] & 0xffffu));
                float b = (h2f((uint16_t)(w[e] >> 16)) * rstd) * h2f((uint16_t)(gw[e] >> 16));
                a = a > 0.f ? __builtin_fminf(a, 64504.f) : __builtin_fmaxf(a, -64504.f);
                b = b > 0.f ? __builtin_fminf(b, 64504.f) : __builtin_fmaxf(b, -64504.f);
                o[e] = (unsigned)f2h_bits_of_f32_result(a) | ((unsigned)f2h_bits_of_f32_result(b) << 16);
            }
            x[v] = make_uint4(o[0], o[1], o[2], o[3]);
            if (QUANT) reinterpret_cast<uint4*>(lrow)[idx] = x[v];
        }
    }
    if (QUANT) {
        if (TPR > 64) __syncthreads(); // the whole row is in LDS (one wave per row needs no barrier: LDS ops are in order)
        if (row_ok)
            for (int j = t; j < O; j += TPR) {
                const int c = ind[j];
                outl[row * (int64_t)O + j] = (c >= 0 && c < K) ? lrow[c] : (uint16_t)0;
            }
#pragma unroll
        for (int v = 0; v < MAXV; ++v) {
            const int idx = v * TPR + t;
            if (idx < nvec) {
                const unsigned m8 = (zmask[idx >> 2] >> ((idx & 3) * 8)) & 0xffu;
                unsigned w[4] = {x[v].x, x[v].y, x[v].z, x[v].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (m8 & (1u << (2 * e))) w[e] &= 0xffff0000u;
                    if (m8 & (2u << (2 * e))) w[e] &= 0x0000ffffu;
                    m2 = pk_max_u16(m2, w[e] & 0x7fff7fffu); // |x| bit patterns, NaN patterns included
                }
                x[v] = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
    }
    // the normalised row (outlier columns zeroed when quantising), one 16-byte store per vector
    uint4* __restrict__ dsto = reinterpret_cast<uint4*>(out + (row_ok ? row : 0) * (int64_t)K);
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
        const int idx = v * TPR + t;
        if (row_ok && idx < nvec) dsto[idx] = x[v];
    }
    if (!QUANT) return;

    auto row_max = [&](int val) __attribute__((always_inline)) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) val = max(val, __shfl_xor(val, off, 64));
        if (TPR > 64) {
            __syncthreads(); // (also orders a second use of redi[])
            if ((tid & 63) == 0) redi[tid >> 6] = val;
            __syncthreads();
            val = max(max(redi[0], redi[1]), max(redi[2], redi[3]));
        }
        return val;
    };
    const int amax_all = row_max((int)max(m2 & 0xffffu, m2 >> 16));
    int amax = amax_all;
    if (amax_all > 0x7c00) { // a NaN in the row: max with NaNs dropped (__hmax); -1 = every element is NaN
        amax = -1;
#pragma unroll
        for (int v = 0; v < MAXV; ++v) {
            const unsigned w[4] = {x[v].x, x[v].y, x[v].z, x[v].w};
            if (v * TPR + t < nvec) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int lo = (int)(w[e] & 0x7fffu), hi = (int)((w[e] >> 16) & 0x7fffu);
                    lo = lo > 0x7c00 ? -1 : lo;
                    hi = hi > 0x7c00 ? -1 : hi;
                    amax = max(amax, max(lo, hi));
                }
            }
        }
        amax = row_max(amax);
    }
    const uint16_t amax_bits = amax < 0 ? (uint16_t)0x7fffu : (uint16_t)amax;
    const uint16_t s_bits = f2h_bits(h2f(amax_bits) / (QUANT == 4 ? 7.0f : 127.0f));
    const float s = h2f(s_bits);
    const float rs = 1.0f / s;
    if (row_ok && t == 0) scale[row] = s_bits;
    if (QUANT == 4) { // packed int4 pairs: element 2i in the low nibble (cutlass::int4b_t), low 4 bits of half2int_rn
        unsigned* __restrict__ dst4 = reinterpret_cast<unsigned*>(q + (row_ok ? row : 0) * (int64_t)(K >> 1));
        for (int v = 0; v < MAXV; ++v) {
            const int idx = v * TPR + t;
            if (row_ok && idx < nvec) {
                const unsigned w[4] = {x[v].x, x[v].y, x[v].z, x[v].w};
                unsigned o = 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned q0 = (unsigned)quant_one(h2f((uint16_t)(w[e] & 0xffffu)), s) & 0xfu;
                    const unsigned q1 = (unsigned)quant_one(h2f((uint16_t)(w[e] >> 16)), s) & 0xfu;
                    o |= (q0 | (q1 << 4)) << (8 * e);
                }
                dst4[idx] = o;
            }
        }
        return;
    }
    uint2* __restrict__ dstq = reinterpret_cast<uint2*>(q + (row_ok ? row : 0) * (int64_t)K);
    auto slot = [&](int idx) __attribute__((always_inline)) -> uint2* { // 8-byte group idx of this row -> its place in the image
        if (!FRAG) return dstq + idx;
        const int64_t r = row_ok ? row : 0;
        const int64_t blk = (r >> 4) * ((K + 63) >> 6) + (idx >> 3);
        return reinterpret_cast<uint2*>(q + (blk << 10) + (((idx >> 1) & 3) << 8) + ((r & 15) << 4) + ((idx & 1) << 3));
    };
    const bool stream_out = !FRAG && (int64_t)M * K >= ((int64_t)8 << 20); // prefill-size images: non-temporal stores (launch-uniform)
    if (amax_all < 0x7c00 && s_bits != 0) { // every element finite, scale finite and non-zero (row-uniform)
#pragma unroll
        for (int v = 0; v < MAXV; ++v) {
            const int idx = v * TPR + t;
            if (row_ok && idx < nvec) {
                typedef unsigned v2u_ __attribute__((ext_vector_type(2)));
                const uint2 q8 = quant_vec8_finite(x[v], s, rs);
                if (stream_out) __builtin_nontemporal_store(v2u_{q8.x, q8.y}, reinterpret_cast<v2u_*>(slot(idx))); // (quant_kernels.hip)
                else *slot(idx) = q8;
            }
        }
    } else {
        for (int v = 0; v < MAXV; ++v) {
            const int idx = v * TPR + t;
            if (row_ok && idx < nvec) {
                const unsigned w[4] = {x[v].x, x[v].y, x[v].z, x[v].w};
                unsigned o[2] = {0u, 0u};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q0 = quant_one(h2f((uint16_t)(w[e] & 0xffffu)), s);
                    const int q1 = quant_one(h2f((uint16_t)(w[e] >> 16)), s);
                    o[e >> 1] |= (unsigned)(q0 | (q1 << 8)) << ((e & 1) * 16);
                }
                *slot(idx) = make_uint2(o[0], o[1]);
            }
        }
    }
}

template <int TPR, int MAXV>
static hipError_t launch_norm(const uint16_t* X, const uint16_t* gamma, uint16_t* out, uint16_t* outl,
                              const int32_t* ind, int8_t* q, uint16_t* scale, float eps, int M, int K, int O,
                              int quant, hipStream_t st, int q_layout = 0)
{
    if constexpr (TPR == 256 && MAXV <= 8) {
        if (q_layout == 1 && quant == 8) {
            const size_t lds = (size_t)((K + 127) / 128) * 16 + (size_t)K * 2;
            static DeviceOnce oncef;
            if (hipError_t e = ensure_dynamic_lds(rmsnorm_quant_kernel<TPR, MAXV, 8, true>, 160 * 1024 - 64, oncef); e != hipSuccess)
                return e;
            hipLaunchKernelGGL((rmsnorm_quant_kernel<TPR, MAXV, 8, true>), dim3((unsigned)M), dim3(NBLOCK), lds, st, X, gamma, out,
                               outl, ind, q, scale, eps, M, K, O);
            return hipGetLastError();
        }
    }
    if (q_layout != 0) return hipErrorInvalidValue;
    constexpr int RPB = NBLOCK / TPR;
    const dim3 grid((unsigned)((M + RPB - 1) / RPB)), block(NBLOCK);
    if (quant) {
        const size_t lds = (size_t)((K + 127) / 128) * 16 + (size_t)RPB * K * 2;
        // rows of 8192 x 4 waves need more than the default 64 KiB of dynamic LDS
        static DeviceOnce once8, once4;
        if (hipError_t e = ensure_dynamic_lds(rmsnorm_quant_kernel<TPR, MAXV, 8>, 160 * 1024 - 64, once8); e != hipSuccess)
            return e;
        if (hipError_t e = ensure_dynamic_lds(rmsnorm_quant_kernel<TPR, MAXV, 4>, 160 * 1024 - 64, once4); e != hipSuccess)
            return e;
        if (quant == 4)
            hipLaunchKernelGGL((rmsnorm_quant_kernel<TPR, MAXV, 4>), grid, block, lds, st, X, gamma, out, outl, ind, q,
                               scale, eps, M, K, O);
        else
            hipLaunchKernelGGL((rmsnorm_quant_kernel<TPR, MAXV, 8>), grid, block, lds, st, X, gamma, out, outl, ind, q,
                               scale, eps, M, K, O);
    } else {
        hipLaunchKernelGGL((rmsnorm_quant_kernel<TPR, MAXV, 0>), grid, block, 0, st, X, gamma, out, outl, ind, q,
                           scale, eps, M, K, O);
    }
    return hipGetLastError();
}

// quant = 0: plain RMSNorm (only `out`), 8 / 4: fused producer.  hipErrorInvalidValue for rows longer than 32768.
hipError_t launch_rmsnorm_quant(const void* X, const void* gamma, void* out, void* outl, const int32_t* ind, int8_t* q,
                                void* scale, float eps, int M, int K, int O, int quant, hipStream_t st, int q_layout)
{
    if (q_layout != 0 && !(q_layout == 1 && quant == 8 && quant_frag_layout_supported(M, K))) return hipErrorInvalidValue;
    if (M <= 0) return hipSuccess;
    const uint16_t* x = static_cast<const uint16_t*>(X);
    const uint16_t* g = static_cast<const uint16_t*>(gamma);
    uint16_t* o = static_cast<uint16_t*>(out);
    uint16_t* ol = static_cast<uint16_t*>(outl);
    uint16_t* sc = static_cast<uint16_t*>(scale);
    const int nvec = K / 8;
    // Decode batches (few rows: the launch is a chain of latencies, not a stream): a whole 256-thread block per row, so that
    // a row is ONE load round trip of 1-4 vectors per lane instead of 8-16 on a single wavefront -- the same rule as the
    // quantiser's (quant_kernels.hip); 32 x 4096: 11.0 -> ~4.5 us for the fused producer (profiles/r03_small_m_timeline.txt)
    if ((M <= quant_block_rows(nvec, true) || q_layout == 1) && nvec > 64 * 2) { // (round 5: at every size here -- the table in quant_kernels.hip)
        if (nvec <= 256 * 2) return launch_norm<256, 2>(x, g, o, ol, ind, q, sc, eps, M, K, O, quant, st, q_layout);
        if (nvec <= 256 * 4) return launch_norm<256, 4>(x, g, o, ol, ind, q, sc, eps, M, K, O, quant, st, q_layout);
    }
    if (nvec <= 64 * 2) return launch_norm<64, 2>(x, g, o, ol, ind, q, sc, eps, M, K, O, quant, st);
    if (nvec <= 64 * 4) return launch_norm<64, 4>(x, g, o, ol, ind, q, sc, eps, M, K, O, quant, st);
    if (nvec <= 64 * 8) return launch_norm<64, 8>(x, g, o, ol, ind, q, sc, eps, M, K, O, quant, st);
    if (nvec <= 64 * 16) return launch_norm<64, 16>(x, g, o, ol, ind, q, sc, eps, M, K, O, quant, st);
    if (nvec <= 256 * 8) return launch_norm<256, 8>(x, g, o, ol, ind, q, sc, eps, M, K, O, quant, st, q_layout);
    if (nvec <= 256 * 16) return launch_norm<256, 16>(x, g, o, ol, ind, q, sc, eps, M, K, O, quant, st);
    return hipErrorInvalidValue;
}

} // namespace mixq

// The whole MixQ linear in ONE launch for decode batches (5 <= M <= 32, K <= 8192): per-token quantisation + outlier
// extraction run INSIDE the weight-streaming GEMM kernel (gemm_skinny_kernels.hip) instead of one launch earlier.
//
// Why: at this size the operator is a chain of latencies -- launch, first HBM round trip, quantiser drain, kernel
// boundary (~1.5-1.9 us), launch, weight stream -- and the two-launch form spends ~5.5 of its ~13.6 us (4096 x 4096,
// bs = 32: BASELINE configs[0]) before the first weight byte is requested.  Here every workgroup requests its weights
// first, and the quantiser's latency runs under that stream:
//
//   1. every workgroup issues the first 16 K-steps of its 16 weight rows (the same loads the skinny kernel issues);
//   2. workgroup b < M additionally quantises token row b with all 256 threads -- same arithmetic as
//      quant_extract_kernel (reference: kernel/i8gemm.cu:66-107 FindRowScaleKernel, :198-224 outlier gather,
//      TsinghuaMixQPlugin.cpp:518-532 order) -- and publishes qA / sA / fpA of that row with write-through stores,
//      waits for their acknowledgement and raises the row's flag (relaxed agent-scope store);
//   3. every workgroup polls the M flags (one lane per row, `s_sleep` between polls), then runs the skinny GEMM:
//      qA fragments from L2 as the MFMA B operand, partial sums through LDS, fp16 outlier side GEMM, dequant FMA.
//   Progress does not rest on dispatch order: a workgroup that has waited ~50 us quantises the missing rows itself
//   (the same bits are written twice at worst), so a waiting workgroup never depends on one that is not running.
//   The last workgroup through the wait clears the flags, so they are zero again for the next launch.
//
// The flags cannot live in the caller's workspace (TensorRT hands every layer the same uninitialised buffer and nothing
// runs before this kernel to clear them): they live in library-owned device memory, one zero-initialised slot per
// (device, workspace pointer) -- calls that share a workspace are serialised by their engine anyway, calls with different
// workspaces get different slots (mixq_api.hip).
//
// Visibility: qA / sA / fpA are written with `sc1` (write-through) stores and acknowledged before the flag; readers
// poll with `sc1` loads.  No reader touches those addresses before the flags are up, a kernel starts with its L1 / L2
// invalidated, and rows are at least one cache line apart except sA (read only after ALL flags): plain loads after the
// wait cannot hit a stale line.
#include "mixq_device.h"
#include "mixq_launch.h"
#include <atomic>

namespace mixq {

namespace fq {
constexpr int KW = 4;           // waves per workgroup = K quarters
constexpr int MAXROWS = 32;
constexpr int FLAG0 = 0;        // words[0 .. 31]: row flags
constexpr int FINISHED = 32;    // words[32]: workgroups through the wait

__device__ __forceinline__ void st_wt_b64(void* p, unsigned long long v)
{
    asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void st_wt_b16(void* p, unsigned v)
{
    asm volatile("global_store_short %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
}

// One token row, all 256 threads: amax -> scale -> int8 row, outlier gather.  Arithmetic identical to
// quant_extract_kernel<256, MAXV, false> (same helpers, same order); stores are write-through.
template <int MAXV>
__device__ __forceinline__ void quant_row(const uint16_t* __restrict__ A, int8_t* __restrict__ qA,
                                          uint16_t* __restrict__ sA, uint16_t* __restrict__ fpA,
                                          const int32_t* __restrict__ ind, int64_t row, int K, int O, int* red, int tid)
{
    const int nvec = K >> 3;
    const uint4* __restrict__ src = reinterpret_cast<const uint4*>(A + row * (int64_t)K);
    uint4 x[MAXV];
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
        const int idx = v * 256 + tid;
        x[v] = idx < nvec ? src[idx] : make_uint4(0u, 0u, 0u, 0u);
    }
    for (int j = tid; j < O; j += 256) st_wt_b16(fpA + row * (int64_t)O + j, A[row * (int64_t)K + ind[j]]);

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
        __syncthreads(); // (also orders a second use of red[])
        if ((tid & 63) == 0) red[tid >> 6] = val;
        __syncthreads();
        return max(max(red[0], red[1]), max(red[2], red[3]));
    };
    const int amax_all = row_max((int)max(m2 & 0xffffu, m2 >> 16));
    int amax = amax_all;
    if (amax_all > 0x7c00) { // the row holds a NaN: redo the max with NaNs dropped the way __hmax drops them
        amax = -1;
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
    const uint16_t amax_bits = amax < 0 ? (uint16_t)0x7fffu : (uint16_t)amax;
    const uint16_t s_bits = f2h_bits(h2f(amax_bits) / 127.0f); // __hdiv(max, 127.0)
    const float s = h2f(s_bits);
    const float rs = 1.0f / s;
    if (tid == 0) st_wt_b16(sA + row, s_bits);
    char* const dst = reinterpret_cast<char*>(qA + row * (int64_t)K);
    if (amax_all < 0x7c00 && s_bits != 0) {
#pragma unroll
        for (int v = 0; v < MAXV; ++v) {
            const int idx = v * 256 + tid;
            if (idx < nvec) {
                const uint2 q = quant_vec8_finite(x[v], s, rs);
                st_wt_b64(dst + (int64_t)idx * 8, (unsigned long long)q.x | ((unsigned long long)q.y << 32));
            }
        }
    } else {
        for (int v = 0; v < MAXV; ++v) {
            const int idx = v * 256 + tid;
            if (idx < nvec) {
                const unsigned w[4] = {x[v].x, x[v].y, x[v].z, x[v].w};
                unsigned o[2] = {0u, 0u};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int q0 = quant_one(h2f((uint16_t)(w[e] & 0xffffu)), s);
                    int q1 = quant_one(h2f((uint16_t)(w[e] >> 16)), s);
                    o[e >> 1] |= (unsigned)(q0 | (q1 << 8)) << ((e & 1) * 16);
                }
                st_wt_b64(dst + (int64_t)idx * 8, (unsigned long long)o[0] | ((unsigned long long)o[1] << 32));
            }
        }
    }
}
} // namespace fq

// p.A / p.sA / p.fpA point at the workspace regions this kernel FILLS (and then reads); rawA = the fp16 activations.
// timeout_ticks: wall-clock ticks (100 MHz) a workgroup waits before it quantises missing rows itself (0 = at once:
// used by the tests to exercise that path).
template <int MT, int EPI, int MAXV>
__global__ __launch_bounds__(256) void gemm_skinny_fusedq_kernel(const GemmParams p, const uint16_t* __restrict__ rawA,
                                                                  const int32_t* __restrict__ ind,
                                                                  unsigned* __restrict__ words, unsigned timeout_ticks)
{
    using namespace fq;
    __shared__ v4i part[KW][MT][64]; // [K part][m tile][lane]
    __shared__ int red[4];
    __shared__ unsigned sh_mask, sh_late;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 16;
    const int lr = lane & 15, lq = lane >> 4;
    const int64_t K = p.K;
    auto stamp = [&](int idx) __attribute__((always_inline)) {
        if (p.dbg != nullptr && tid == 0)
            static_cast<unsigned long long*>(p.dbg)[(size_t)blockIdx.x * 8 + idx] = wall_clock64();
    };
    stamp(0);
    int8_t* const qA = const_cast<int8_t*>(p.A);
    uint16_t* const sA = const_cast<uint16_t*>(p.sA);
    uint16_t* const fpA = const_cast<uint16_t*>(p.fpA);

    const int nsteps = (p.K + 63) >> 6;
    const int per = (nsteps + KW - 1) / KW;
    const int s_begin = min(wave * per, nsteps), s_end = min(s_begin + per, nsteps);
    const int8_t* wrow = p.B + (int64_t)min(n0 + lr, p.N - 1) * K + lq * 16;
    const v4i zero4 = {0, 0, 0, 0};

    // ---- 1. request the first 16 K-steps of this wave's weight rows (HBM) -------------------------------------------
    v4i wf[16];
    const int cnt0 = min(16, s_end - s_begin);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int64_t kb = (int64_t)(s_begin + u) * 64;
        wf[u] = (u < cnt0 && kb + lq * 16 < K) ? *reinterpret_cast<const v4i*>(wrow + kb) : zero4;
    }
    // epilogue operands that do not depend on the quantiser: weight scales, outlier weights of this lane's column
    const bool fin = EPI != EPI_INT32 && wave < MT; // this wave finishes m-tile `wave`
    const int fm = wave * 16 + lr, fnb = n0 + 4 * lq;
    constexpr int PRE = 4;
    v8h pxf[PRE];
    uint2 psw = {0u, 0u};
    const int obytes = p.O * 2;
    if (fin) {
        const char* xw = reinterpret_cast<const char*>(p.fpW) + (int64_t)min(n0 + lr, p.N - 1) * obytes;
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int kb = u * 64 + lq * 16;
            if (kb < obytes) pxf[u] = *reinterpret_cast<const v8h*>(xw + kb);
            else
#pragma unroll
                for (int e = 0; e < 8; ++e) pxf[u][e] = (_Float16)0.f;
        }
        psw = *reinterpret_cast<const uint2*>(p.sW + min(fnb, p.N - 4));
    }

    // ---- 2. rows b, b + grid, ...: quantise and publish ----------------------------------------------------------------
    auto publish_row = [&](int r) __attribute__((always_inline)) {
        quant_row<MAXV>(rawA, qA, sA, fpA, ind, r, p.K, p.O, red, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every write-through store acknowledged
        __syncthreads();
        if (tid == 0) __hip_atomic_store(words + FLAG0 + r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    stamp(1);
    for (int r = blockIdx.x; r < p.M; r += gridDim.x) publish_row(r);
    stamp(2);

    // ---- 3. wait for the M flags (bounded; then help) ------------------------------------------------------------------
    {
        const unsigned want = p.M >= 32 ? 0xffffffffu : ((1u << p.M) - 1u);
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            if (wave == 0) {
                const unsigned f = lane < p.M ? __hip_atomic_load(words + FLAG0 + lane, __ATOMIC_RELAXED,
                                                                  __HIP_MEMORY_SCOPE_AGENT)
                                              : 0u;
                const unsigned long long b = __ballot(f != 0u);
                if (lane == 0) { // ONE thread reads the clock: the decision below has to be workgroup-uniform
                    sh_mask = (unsigned)b;
                    sh_late = wall_clock64() - t0 >= (unsigned long long)timeout_ticks ? 1u : 0u;
                }
            }
            __syncthreads();
            const unsigned missing = want & ~sh_mask;
            const bool late = sh_late != 0u;
            __syncthreads(); // sh_mask / sh_late may be rewritten from here on
            if (missing == 0u) break;
            if (late) {
                for (int r = 0; r < p.M; ++r) // (workgroup-uniform: `missing` came through LDS)
                    if ((missing >> r) & 1u) publish_row(r);
            } else {
                __builtin_amdgcn_s_sleep(8);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); // (compiler ordering: no load of qA / sA / fpA above this)
    }
    stamp(3);
    // through the wait: count in (the answer is only needed at the very end, its latency hides under the GEMM)
    unsigned finished_before = 0u;
    if (tid == 0)
        finished_before = __hip_atomic_fetch_add(words + FINISHED, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- 4. the skinny GEMM (gemm_skinny_kernels.hip), first chunk out of the prefetched weights ------------------------
    const int8_t* arow[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) arow[t] = qA + (int64_t)min(t * 16 + lr, p.M - 1) * K + lq * 16;
    v4i acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = v4i{0, 0, 0, 0};
    v8h pyf[PRE];
    uint16_t psa = 0;
    if (fin) { // the quantiser's outputs for this wave's epilogue
        const char* ya = reinterpret_cast<const char*>(fpA) + (int64_t)min(fm, p.M - 1) * obytes;
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int kb = u * 64 + lq * 16;
            if (kb < obytes) pyf[u] = *reinterpret_cast<const v8h*>(ya + kb);
            else
#pragma unroll
                for (int e = 0; e < 8; ++e) pyf[u][e] = (_Float16)0.f;
        }
        psa = sA[min(fm, p.M - 1)];
    }
    auto consume = [&](int s0, int cnt) __attribute__((always_inline)) {
#pragma unroll
        for (int u0 = 0; u0 < 16; u0 += 4) {
            v4i af[4][MT];
#pragma unroll
            for (int u = u0; u < u0 + 4; ++u) {
                const int64_t kb = (int64_t)(s0 + u) * 64;
                const bool ok = u < cnt && kb + lq * 16 < K;
#pragma unroll
                for (int t = 0; t < MT; ++t) af[u - u0][t] = ok ? *reinterpret_cast<const v4i*>(arow[t] + kb) : zero4;
            }
#pragma unroll
            for (int u = u0; u < u0 + 4; ++u)
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(wf[u], af[u - u0][t], acc[t], 0, 0, 0);
        }
    };
    consume(s_begin, cnt0);
    for (int s = s_begin + 16; s < s_end; s += 16) { // K > 4096: the rest of the stream, 16 steps at a time
        const int cnt = min(16, s_end - s);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int64_t kb = (int64_t)(s + u) * 64;
            wf[u] = (u < cnt && kb + lq * 16 < K) ? *reinterpret_cast<const v4i*>(wrow + kb) : zero4;
        }
        consume(s, cnt);
    }

    stamp(4);
#pragma unroll
    for (int t = 0; t < MT; ++t) part[wave][t][lane] = acc[t];
    __syncthreads();
    stamp(5);

    if (wave < MT) {
        const int t = wave;
        v4i a = part[0][t][lane];
#pragma unroll
        for (int w2 = 1; w2 < KW; ++w2) {
            const v4i b = part[w2][t][lane];
            a = v4i{a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]};
        }
        const int m = t * 16 + lr;
        const int nb = n0 + 4 * lq;
        if (EPI == EPI_INT32) {
            if (m < p.M && nb < p.N) *reinterpret_cast<v4i*>(static_cast<int32_t*>(p.D) + (int64_t)m * p.N + nb) = a;
        } else {
            v4f P = {0.f, 0.f, 0.f, 0.f};
            if (p.O > 0) {
                const char* xw = reinterpret_cast<const char*>(p.fpW) + (int64_t)min(n0 + lr, p.N - 1) * obytes;
                const char* ya = reinterpret_cast<const char*>(fpA) + (int64_t)min(m, p.M - 1) * obytes;
#pragma unroll
                for (int u = 0; u < PRE; ++u)
                    if (u * 64 < obytes) P = __builtin_amdgcn_mfma_f32_16x16x32_f16(pxf[u], pyf[u], P, 0, 0, 0);
                for (int k0 = PRE * 64; k0 < obytes; k0 += 64) {
                    const int kb = k0 + lq * 16;
                    v8h xf, yf;
                    if (kb < obytes) {
                        xf = *reinterpret_cast<const v8h*>(xw + kb);
                        yf = *reinterpret_cast<const v8h*>(ya + kb);
                    } else {
#pragma unroll
                        for (int e = 0; e < 8; ++e) xf[e] = (_Float16)0.f, yf[e] = (_Float16)0.f;
                    }
                    P = __builtin_amdgcn_mfma_f32_16x16x32_f16(xf, yf, P, 0, 0, 0);
                }
            }
            if (m < p.M && nb < p.N) {
                const float sa = h2f(psa);
                const uint16_t swh[4] = {(uint16_t)(psw.x & 0xffffu), (uint16_t)(psw.x >> 16), (uint16_t)(psw.y & 0xffffu),
                                         (uint16_t)(psw.y >> 16)};
                uint16_t oh[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float c = p.O > 0 ? h2f(f2h_bits_of_f32_result(P[e])) : 0.f;
                    float v = __builtin_fmaf((float)a[e], h2f(swh[e]) * sa, c);
                    if (epi_has_silu(EPI)) v = silu_f32(v);
                    oh[e] = f2h_bits_of_f32_result(v);
                }
                uint2 o;
                o.x = (unsigned)oh[0] | ((unsigned)oh[1] << 16);
                o.y = (unsigned)oh[2] | ((unsigned)oh[3] << 16);
                *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.D) + (int64_t)m * p.N + nb) = o;
            }
        }
    }
    stamp(6);
    // ---- 5. the last workgroup through the wait re-arms the flags for the next launch ------------------------------------
    if (tid == 0 && finished_before == gridDim.x - 1u) {
        for (int r = 0; r < MAXROWS; ++r)
            __hip_atomic_store(words + FLAG0 + r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(words + FINISHED, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    stamp(7);
}

static std::atomic<int> g_fusedq_mode{0};          // 1 on (default), 0 off, 2 on with an immediate time-out (tests)
void set_fusedq_mode(int v) { g_fusedq_mode.store(v); }

// Same domain as launch_gemm's choice of the skinny kernel (gemm_kernels.hip), minus what the in-kernel quantiser does
// not cover: rows longer than 256 threads x 4 vectors, more outlier columns than one pass, more rows than flags.
bool gemm_fusedq_supported(int M, int N, int K, int O)
{
    if (g_fusedq_mode.load() == 0) return false;
    if (M <= 4 || M > fq::MAXROWS || K >= 8192 || K % 8 || O > 256 || O % 8) return false; // (K >= 8192: the K split over
                                                                                              //  workgroups of gemm_kernels.hip)
    if (!(M <= 16 || N < 8192)) return false;        // (launch_gemm: the skinny kernel's range)
    if ((N + 15) / 16 > 8 * num_cus()) return false; // every workgroup resident at once, with a wide margin
    return true;
}

template <int MT, int MAXV>
static hipError_t launch_fq(const GemmParams& p, const uint16_t* rawA, const int32_t* ind, unsigned* words,
                            unsigned timeout, hipStream_t st)
{
    const dim3 grid((unsigned)((p.N + 15) / 16)), block(256);
    hipLaunchKernelGGL((gemm_skinny_fusedq_kernel<MT, EPI_DEQUANT, MAXV>), grid, block, 0, st, p, rawA, ind, words,
                       timeout);
    return hipGetLastError();
}

// p: GemmParams as for launch_gemm_skinny with A / sA / fpA = the workspace regions to fill.
hipError_t launch_gemm_fusedq(const GemmParams& p, const void* rawA, const int32_t* ind, void* sync_words, hipStream_t st)
{
    const unsigned timeout = g_fusedq_mode.load() == 2 ? 0u : 5000u; // 50 us at 100 MHz
    const uint16_t* a = static_cast<const uint16_t*>(rawA);
    unsigned* w = static_cast<unsigned*>(sync_words);
    const int nvec = p.K / 8;
    const int mt = (p.M + 15) / 16;
    if (nvec <= 256)
        return mt == 1 ? launch_fq<1, 1>(p, a, ind, w, timeout, st) : launch_fq<2, 1>(p, a, ind, w, timeout, st);
    if (nvec <= 512)
        return mt == 1 ? launch_fq<1, 2>(p, a, ind, w, timeout, st) : launch_fq<2, 2>(p, a, ind, w, timeout, st);
    return mt == 1 ? launch_fq<1, 4>(p, a, ind, w, timeout, st) : launch_fq<2, 4>(p, a, ind, w, timeout, st);
}

} // namespace mixq

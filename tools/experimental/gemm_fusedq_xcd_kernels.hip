// The whole MixQ linear in ONE launch for decode batches (5 <= M <= 32, K < 8192) -- BASELINE configs[0] -- with an
// XCD-LOCAL hand-over of the quantised activations (round 3; round 2's form with an agent-scope hand-over was 35 % slower
// than two launches and is described in tools/experimental/README.md).
//
// Why one launch: the two-launch operator (quant_extract_kernel, then gemm_skinny_kernel) is a chain of latencies
// (profiles/r03_small_m_timeline.txt, 4096 x 4096 at bs = 32, 11.8 us): kernel boundary 1.1 | quantiser 2.7 | kernel
// boundary 1.25 | GEMM prologue 0.9 | weight + qA fragments 5.2 | hand-over + epilogue 0.7.  The fragment phase is bound by
// how many bytes ONE CU can have in flight at memory latency (~37 GB/s per CU whether the bytes are weights from HBM or qA
// re-read from the Infinity Cache: each XCD's L2 starts every kernel cold), so the 128 KiB of qA every workgroup needs
// cost more than its 64 KiB of weights.
//
// What this kernel does instead:
//   1. every workgroup requests its 16 weight rows (the same loads the skinny kernel issues) -- the weight stream now runs
//      under everything that follows;
//   2. each XCD quantises the M token rows FOR ITSELF: the XCD's j-th workgroup (block b -> XCD b % 8 is assumed for load
//      balance only; the copy index is the HW_REG_XCC_ID actually read) quantises row j -- same arithmetic as
//      quant_extract_kernel (reference: kernel/i8gemm.cu:66-107 FindRowScaleKernel, :198-224 outlier gather,
//      TsinghuaMixQPlugin.cpp:518-532 order) -- into THAT XCD's copy of qA / sA / fpA with PLAIN stores (which stay in the
//      XCD's L2), waits for their acknowledgement and raises the row's flag in the XCD's flag run;
//   3. every workgroup polls its own XCD's M flags, then runs the skinny GEMM on its XCD's copy: the qA fragments are now
//      L2 HITS written microseconds earlier by a neighbour CU -- 8x redundant quantisation (8 KiB read + 4 KiB written per
//      row and XCD) buys a fragment phase at L2-hit latency instead of Infinity-Cache latency, and one kernel boundary.
//
// Second build (same round): the per-XCD copy is written and read FRAGMENT-MAJOR (quant_kernels.hip FRAG, gemm_skinny_kernels.hip
// AFRAG: block (16-row tile, 64-byte k-step), lane l's 16 bytes at l x 16), which is what made the two-launch operator's
// fragment phase fast; the first build read a row-major copy and its fragment phase was as slow as before (5.3 us).
//
// Visibility argument (gfx950-specific, and deliberately so): producer and consumers of a copy are on the SAME XCD by
// construction (both index it with the XCC id they read), a CU's vector L1 is write-through and starts every kernel
// invalidated, no consumer touches the copy before the flags are up, so its loads miss L1 and are served by the one L2
// that acknowledged the producer's stores.  No agent-scope fence (1.7 us per CU) and no write-through (sc1) payload --
// which would DROP the lines from the L2 and put the re-read back at fabric latency (round 2's measurement).
// Flags: 8-byte words tagged with the launch epoch + 1 (never reset, never a host-side salt: that is frozen under graph
// replay); polled with sc1 loads (bypass L1, L2-served), stored plain, one 256-byte run per XCD.  The epoch lives in 8
// per-XCD words read with ONE returning atomic per workgroup (32 pullers per word; its latency runs under the row load)
// and is bumped by the last workgroup through the wait (a returning atomic on `passed` whose answer is only needed at the
// very end of the kernel).  Progress does not rest on placement or dispatch order: a workgroup that has waited ~30 us
// quantises the missing rows of ITS XCD's copy itself (the same bits are written twice at worst).
#include "mixq_device.h"
#include "mixq_launch.h"
#include <atomic>

namespace mixq {

namespace fq {
constexpr int KW = 4;           // waves per workgroup = K quarters
constexpr int MAXROWS = 32;
constexpr int XCDS = 8;         // copies of qA / sA / fpA, flag runs, epoch words
// sync words (uint64, library-owned, zero-initialised once): [x * 16] epoch of XCD x (one 128-byte line each) | [128] passed |
// [256 + x * 32 + r] flag of row r in XCD x (a 256-byte run per XCD)
constexpr int W_EPOCH = 0, W_PASSED = 128, W_FLAGS = 256;

__device__ __forceinline__ int xcc_id()
{
    int v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 7; // (bits 3:0 = XCC id; 8 XCDs on this part)
}

// One token row, all 256 threads: amax -> scale -> int8 row, outlier gather.  Arithmetic identical to
// quant_extract_kernel<256, MAXV, false> (same helpers, same order); PLAIN stores into the XCD's copy.
// `x` holds the row (requested by the caller before anything that has latency).
template <int MAXV>
__device__ __forceinline__ void quant_row(const uint4 (&x)[MAXV], const uint16_t* __restrict__ A, int8_t* __restrict__ qA,
                                          uint16_t* __restrict__ sA, uint16_t* __restrict__ fpA,
                                          const int32_t* __restrict__ ind, int64_t row, int K, int O, int* red, int tid)
{
    const int nvec = K >> 3;
    for (int j = tid; j < O; j += 256) fpA[row * (int64_t)O + j] = A[row * (int64_t)K + ind[j]];

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
    if (tid == 0) sA[row] = s_bits;
    const int nsteps = (K + 63) >> 6;
    auto slot = [&](int idx) __attribute__((always_inline)) -> uint2* { // 8-byte group idx of the row -> its place in the image
        const int64_t blk = (row >> 4) * nsteps + (idx >> 3);
        return reinterpret_cast<uint2*>(qA + (blk << 10) + (((idx >> 1) & 3) << 8) + ((row & 15) << 4) + ((idx & 1) << 3));
    };
    if (amax_all < 0x7c00 && s_bits != 0) {
#pragma unroll
        for (int v = 0; v < MAXV; ++v) {
            const int idx = v * 256 + tid;
            if (idx < nvec) *slot(idx) = quant_vec8_finite(x[v], s, rs);
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
                *slot(idx) = make_uint2(o[0], o[1]);
            }
        }
    }
}
} // namespace fq

// p.A / p.sA / p.fpA: copy 0 of the regions this kernel FILLS (and then reads); copy x of each lies x * stride further
// (strides in BYTES).  rawA = the fp16 activations.  timeout_ticks: wall-clock ticks (100 MHz) a workgroup waits before it
// quantises missing rows itself (0 = at once: used by the tests to exercise that path).
template <int MT, int EPI, int MAXV>
__global__ __launch_bounds__(256) void gemm_skinny_fusedq_kernel(const GemmParams p, const uint16_t* __restrict__ rawA,
                                                                  const int32_t* __restrict__ ind,
                                                                  unsigned long long* __restrict__ words,
                                                                  unsigned timeout_ticks, size_t stride_q, size_t stride_s,
                                                                  size_t stride_f)
{
    using namespace fq;
    __shared__ v4i part[KW][MT][64]; // [K part][m tile][lane]
    __shared__ int red[4];
    __shared__ unsigned sh_mask, sh_late;
    __shared__ unsigned long long sh_tag;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n0 = blockIdx.x * 16;
    const int lr = lane & 15, lq = lane >> 4;
    const int64_t K = p.K;
    dbg_stamp(p.dbg, 0);
    const int xcc = xcc_id();
    int8_t* const qA = const_cast<int8_t*>(p.A) + (size_t)xcc * stride_q;
    uint16_t* const sA = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(const_cast<uint16_t*>(p.sA)) + (size_t)xcc * stride_s);
    uint16_t* const fpA = reinterpret_cast<uint16_t*>(reinterpret_cast<char*>(const_cast<uint16_t*>(p.fpA)) + (size_t)xcc * stride_f);
    unsigned long long* const flags = words + W_FLAGS + xcc * MAXROWS;

    // ---- 0. the token row this workgroup quantises for its XCD (if any): requested first, it has the longest way ------
    const int rows_step = max(1, (int)(gridDim.x >> 3));
    const int my_row = (int)(blockIdx.x >> 3); // the XCD's j-th workgroup takes rows j, j + grid / 8, ...
    uint4 xrow[MAXV];
    {
        const int nvec = p.K >> 3;
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(rawA + (int64_t)min(my_row, p.M - 1) * K);
#pragma unroll
        for (int v = 0; v < MAXV; ++v) {
            const int idx = v * 256 + tid;
            xrow[v] = (my_row < p.M && idx < nvec) ? src[idx] : make_uint4(0u, 0u, 0u, 0u);
        }
    }
    // the launch epoch of this XCD's word: ONE returning atomic per workgroup (memory-side, so every workgroup of the launch
    // reads the same value whatever its L2 holds); its latency runs under the row load
    unsigned long long epoch = 0ull; // (kept in a register until the weight loads are out: its first USE is what waits for it)
    if (tid == 0) {
        unsigned long long zero = 0ull;
        asm volatile("" : "+v"(zero)); // (opaque: a constant 0 lets the compiler turn the RMW into an L2-served sc1 LOAD)
        epoch = __hip_atomic_fetch_add(words + W_EPOCH + xcc * 16, zero, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }

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
    dbg_stamp(p.dbg, 1);
    if (tid == 0) sh_tag = epoch + 1ull;
    __syncthreads();
    const unsigned long long tag = sh_tag;

    // ---- 2. quantise and publish (into THIS XCD's copy) ---------------------------------------------------------------
    auto publish = [&](const uint4 (&x)[MAXV], int r) __attribute__((always_inline)) {
        quant_row<MAXV>(x, rawA, qA, sA, fpA, ind, r, p.K, p.O, red, tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every store of this wave acknowledged by the XCD's L2
        __syncthreads();
        if (tid == 0) flags[r] = tag; // plain store: stays in this XCD's L2, where the pollers' sc1 loads are served
    };
    auto load_and_publish = [&](int r) __attribute__((always_inline)) {
        uint4 x[MAXV];
        const int nvec = p.K >> 3;
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(rawA + (int64_t)r * K);
#pragma unroll
        for (int v = 0; v < MAXV; ++v) {
            const int idx = v * 256 + tid;
            x[v] = idx < nvec ? src[idx] : make_uint4(0u, 0u, 0u, 0u);
        }
        publish(x, r);
    };
    if (my_row < p.M) {
        publish(xrow, my_row);
        for (int r = my_row + rows_step; r < p.M; r += rows_step) load_and_publish(r); // fewer workgroups per XCD than rows
    }
    dbg_stamp(p.dbg, 2);

    // ---- 3. wait for this XCD's M flags (bounded; then help) ------------------------------------------------------------
    {
        const unsigned want = p.M >= 32 ? 0xffffffffu : ((1u << p.M) - 1u);
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            if (wave == 0) {
                const unsigned long long f =
                    lane < p.M ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
                const unsigned long long b = __ballot(f == tag);
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
                    if ((missing >> r) & 1u) load_and_publish(r);
            } else {
                __builtin_amdgcn_s_sleep(4);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup"); // (compiler ordering: no load of qA / sA / fpA above this)
    }
    dbg_stamp(p.dbg, 3);
    // through the wait: count in (the answer is only needed at the very end, its latency hides under the GEMM)
    unsigned long long passed_before = 0ull;
    if (tid == 0)
        passed_before = __hip_atomic_fetch_add(words + W_PASSED, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // ---- 4. the skinny GEMM (gemm_skinny_kernels.hip), first chunk out of the prefetched weights ------------------------
    const int8_t* const aimg = qA + lane * 16; // fragment-major image: block (m tile t, k-step s) at (t * nsteps + s) KiB
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
        for (int u0 = 0; u0 < 16; u0 += 8) {
            v4i af[8][MT];
#pragma unroll
            for (int u = u0; u < u0 + 8; ++u) {
                const int su = min(s0 + u, s0 + max(cnt, 1) - 1); // (a step past the wave's range re-reads its last one: the weight
                                                                  //  fragment of such a step is zero, so the product is too)
#pragma unroll
                for (int t = 0; t < MT; ++t) af[u - u0][t] = *reinterpret_cast<const v4i*>(aimg + ((int64_t)(t * nsteps + su) << 10));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = u0; u < u0 + 8; ++u)
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

    dbg_stamp(p.dbg, 4);
#pragma unroll
    for (int t = 0; t < MT; ++t) part[wave][t][lane] = acc[t];
    __syncthreads();

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
    dbg_stamp(p.dbg, 5);
    // ---- 5. the last workgroup through the wait opens the next epoch: every workgroup of this launch has read its XCD's
    // epoch word before it counted itself in, so the bump cannot be seen by this launch ---------------------------------------
    if (tid == 0 && passed_before == (unsigned long long)gridDim.x - 1ull) {
        __hip_atomic_store(words + W_PASSED, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int x = 0; x < XCDS; ++x)
            __hip_atomic_fetch_add(words + W_EPOCH + x * 16, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (p.dbg != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dbg_stamp(p.dbg, 6);
    }
}

static std::atomic<int> g_fusedq_mode{1};          // 1 on (default), 0 off, 2 on with an immediate time-out (tests)
void set_fusedq_mode(int v) { g_fusedq_mode.store(v); }

// Same domain as launch_gemm's choice of the skinny kernel (gemm_kernels.hip), minus what the in-kernel quantiser does
// not cover: rows longer than 256 threads x 4 vectors, more outlier columns than one pass, more rows than flags.
bool gemm_fusedq_supported(int M, int N, int K, int O)
{
    if (g_fusedq_mode.load() == 0) return false;
    if (M <= 4 || M > fq::MAXROWS || K >= 8192 || K % 16 || N % 16 || O > 256 || O % 8) return false; // (K >= 8192: the K split over
                                                                                              //  workgroups of gemm_kernels.hip)
    if (!(M <= 16 || N < 8192)) return false;        // (launch_gemm: the skinny kernel's range)
    if ((N + 15) / 16 > 8 * num_cus()) return false; // every workgroup resident at once, with a wide margin
    return true;
}

// bytes of one copy of each region, as laid out by mixq_api.hip (128-byte aligned)
size_t gemm_fusedq_sync_bytes() { return (size_t)(fq::W_FLAGS + fq::XCDS * fq::MAXROWS) * 8; }

template <int MT, int MAXV>
static hipError_t launch_fq(const GemmParams& p, const uint16_t* rawA, const int32_t* ind, unsigned long long* words,
                            unsigned timeout, size_t sq, size_t ss, size_t sf, hipStream_t st)
{
    const dim3 grid((unsigned)((p.N + 15) / 16)), block(256);
    hipLaunchKernelGGL((gemm_skinny_fusedq_kernel<MT, EPI_DEQUANT, MAXV>), grid, block, 0, st, p, rawA, ind, words, timeout,
                       sq, ss, sf);
    return hipGetLastError();
}

// p: GemmParams as for launch_gemm_skinny with A / sA / fpA = copy 0 of the workspace regions to fill; copy x of each is
// stride_* bytes further (8 copies).
hipError_t launch_gemm_fusedq(const GemmParams& p, const void* rawA, const int32_t* ind, void* sync_words, size_t stride_q,
                              size_t stride_s, size_t stride_f, hipStream_t st)
{
    const unsigned timeout = g_fusedq_mode.load() == 2 ? 0u : 3000u; // 30 us at 100 MHz
    const uint16_t* a = static_cast<const uint16_t*>(rawA);
    unsigned long long* w = static_cast<unsigned long long*>(sync_words);
    const int nvec = p.K / 8;
    const int mt = (p.M + 15) / 16;
    note_gemm_kernel("gemm_skinny_fusedq_kernel (quantiser + skinny GEMM in one launch, XCD-local hand-over)");
    if (nvec <= 256)
        return mt == 1 ? launch_fq<1, 1>(p, a, ind, w, timeout, stride_q, stride_s, stride_f, st)
                       : launch_fq<2, 1>(p, a, ind, w, timeout, stride_q, stride_s, stride_f, st);
    if (nvec <= 512)
        return mt == 1 ? launch_fq<1, 2>(p, a, ind, w, timeout, stride_q, stride_s, stride_f, st)
                       : launch_fq<2, 2>(p, a, ind, w, timeout, stride_q, stride_s, stride_f, st);
    return mt == 1 ? launch_fq<1, 4>(p, a, ind, w, timeout, stride_q, stride_s, stride_f, st)
                   : launch_fq<2, 4>(p, a, ind, w, timeout, stride_q, stride_s, stride_f, st);
}

} // namespace mixq

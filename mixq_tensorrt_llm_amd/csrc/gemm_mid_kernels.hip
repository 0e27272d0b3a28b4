// Mid-M fused W8A8O16 GEMM (round 6): 128 x 128 tiles, COPY-ONLY waves, two slices per barrier, two K-interleaved compute groups.
//
// Same math, same operand roles, same epilogue arithmetic, same bits as gemm_kernels.hip (reference lines replaced: see there;
// what a 65..1024-row call runs on in the reference: kernel/cutlass/include/cutlass/gemm/device/default_gemm_configuration.h:533-555,
// a 3-stage 128 x 256 x 64 threadblock tile with no K split, whatever M is).
//
// Why a new schedule (notebook R6.1-R6.2).  The deep form (gemm_kernels.hip, ADMA) spends ~0.7 us per 128-byte K
// slice on 0.25 us of MFMA work, warm or cold.  tools/experimental/r06_stage_rate_probe.hip measured what one CU can pull
// global -> LDS: 125-138 GB/s from L2 (32 KiB in 0.25 us) with only TWO issuing waves and with eight MFMA waves running next to
// them, 62 GB/s from the Infinity Cache, 30 GB/s (7.6 TB/s over the chip) from HBM -- the copy engine is not what paces the tile;
// the loop structure is: every wave copies AND multiplies, one workgroup barrier per 512 MFMA cycles per SIMD, 64 x 32 wave tiles
// that read 96 KiB of fragments per slice (384 LDS cycles against 512 MFMA cycles).  Here:
//   * 2 LOADER waves (wave 8: the W rows, wave 9: the qA rows) issue every global_load_lds_dwordx4 of a slice (16 each, SGPR base
//     + lane offset) and nothing else: a compute wave never sits in a copy's issue stall and has no VMEM operation in flight, so
//     its LDS waits are exact `lgkmcnt(n)`;
//   * 8 COMPUTE waves = 2 groups x (2 x 2 waves of 64 x 64): group g multiplies slices g, g + 2, ... of the WHOLE tile (the
//     in-workgroup K split of gemm_kernels.hip's KG, here with one ring of stages): 64 KiB of fragment reads per slice instead
//     of 96, and one wave of each group per SIMD, each with 16 MFMAs per slice;
//   * ONE barrier per PAIR of slices (1024 MFMA cycles per SIMD between barriers): ring of five 32-KiB stages, two being
//     multiplied, three in flight;
//   * after the loop the groups swap halves through LDS (each wave gives away one 64 x 32 half and adds the partner's to the one it
//     keeps: integer adds, same sums), so the epilogue runs on all eight waves; the outlier operands were copied by the loader
//     waves under that exchange.
// K split over XS workgroups per tile ("last arriver adds", as in gemm_kernels.hip) on the post-exchange layout.
#include "mixq_device.h"
#include "mixq_launch.h"
#include <type_traits>

namespace mixq {

namespace mid {
constexpr int BM = 128, BN = 128, KS = 128;
constexpr int XB = BN * KS, YB = BM * KS, STAGE = XB + YB; // 16 + 16 KiB
constexpr int NST = 5;
constexpr int NCW = 8, NLD = 2, TC = NCW * 64, T = (NCW + NLD) * 64;
constexpr int PER = 16;          // copy instructions per loader wave per slice (16 x 1 KiB = one operand's 128 rows x 128 B)
constexpr int OSLICE = 256;      // bytes per LDS row in the outlier phase (128 fp16)
constexpr int EXCH = 65536;      // exchange region [0, 64 KiB); outlier operands behind it
constexpr int GROUP_M = 4;
constexpr size_t LDS = (size_t)NST * STAGE; // 160 KiB
static_assert(EXCH + (BM + BN) * OSLICE <= LDS, "exchange + outlier tiles must fit the ring's LDS");
} // namespace mid

template <int EPI, bool XSP>
__global__ __launch_bounds__(mid::T) void gemm_w8a8o16_mid_kernel(const GemmParams p)
{
    using namespace mid;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int XS = XSP ? p.xsplit : 1;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // ---- block -> tile mapping (XCD-aware, grouped; as gemm_kernels.hip) ---------------------------------------------
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    int t_lin;
    const int xrank = XSP ? (int)blockIdx.x % XS : 0;
    {
        const int bid = XSP ? (int)blockIdx.x / XS : (int)blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        t_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    int tile_m, tile_n;
    {
        const int per_group = GROUP_M * tiles_n;
        const int g = t_lin / per_group, first_m = g * GROUP_M;
        const int gsz = min(tiles_m - first_m, GROUP_M);
        const int within = t_lin - g * per_group;
        tile_m = first_m + within % gsz;
        tile_n = within / gsz;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int64_t K = p.K;
    const int nk_all = p.K / KS; // (the launcher sends K % 128 != 0 to the older build)
    const int kbeg = XSP ? nk_all * xrank / XS : 0;
    const int nk = (XSP ? nk_all * (xrank + 1) / XS : nk_all) - kbeg;
    const int npair = (nk + 1) >> 1;
    const bool has_outliers = p.O > 0;
    const unsigned lds0 = (unsigned)(size_t)(MIXQ_LDS_PTR(smem));

    if (wave >= NCW) {
        // =================================== loader wave: copies and barriers, nothing else ===================================
        const int lw = wave - NCW; // 0: W rows -> region X, 1: qA rows -> region Y
        const int rows_total = lw == 0 ? p.N : p.M, r0 = lw == 0 ? n0 : m0;
        unsigned voff[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) { // LDS row = i * 8 + lane / 8, 16-B slot lane % 8 holds source chunk slot ^ ((row >> 1) & 7)
            const int row = i * 8 + (lane >> 3);
            const int rr = min(r0 + row, rows_total - 1) - r0; // clamp: rows past the edge are copied but never stored
            voff[i] = (unsigned)rr * (unsigned)p.K + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
        }
        const char* const base = (lw == 0 ? reinterpret_cast<const char*>(p.B) : reinterpret_cast<const char*>(p.A)) +
                                 (int64_t)r0 * K + (int64_t)kbeg * KS;
        const bool nt = lw == 0 && (p.flags & 2) != 0;
        auto issue = [&](int s) __attribute__((always_inline)) {
            const unsigned dst = lds0 + (unsigned)(s % NST) * STAGE + lw * XB;
            const char* b = base + (int64_t)s * KS; // wave-uniform
            if (nt) {
#pragma unroll
                for (int i = 0; i < PER; ++i) glds16_sbase_nt(b, voff[i], dst + i * 1024);
            } else {
#pragma unroll
                for (int i = 0; i < PER; ++i) glds16_sbase(b, voff[i], dst + i * 1024);
            }
        };
#pragma unroll
        for (int s = 0; s < 3; ++s)
            if (s < nk) issue(s);
        for (int j = 0; j < npair; ++j) {
            // issued so far: slices .. 2j + 2; the pair 2j, 2j + 1 must have landed (copies complete in order)
            if (2 * j + 2 < nk) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads(); // pair j handed over; the compute waves are done with pair j - 1
            if (2 * j + 3 < nk) issue(2 * j + 3);
            if (2 * j + 4 < nk) issue(2 * j + 4);
        }
        __syncthreads(); // B0: every slice multiplied, the ring is dead
        if (EPI != EPI_INT32 && has_outliers) { // fpW (wave 8) / fpA (wave 9) tiles -> LDS, 256-B rows, slot = chunk ^ (row & 15)
            const int obytes = p.O * 2;
            const char* const ob = lw == 0 ? reinterpret_cast<const char*>(p.fpW) : reinterpret_cast<const char*>(p.fpA);
            const unsigned dst = lds0 + EXCH + lw * (BN * OSLICE);
#pragma unroll 8
            for (int q = 0; q < 32; ++q) {
                const int row = q * 4 + (lane >> 4);
                const int c = ((lane & 15) ^ (row & 15)) << 4;
                const int grow = min(r0 + row, rows_total - 1);
                const char* s = ob + (int64_t)grow * obytes + c;
                if (c >= obytes) s = static_cast<const char*>(p.zeros);
                glds16_vaddr(s, dst + q * 1024);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads(); // B1
        if (XSP) {
            __syncthreads(); // B2
            __syncthreads(); // B3
        }
        return;
    }

    // ======================================== compute wave ========================================
    const int group = wave >> 2;          // slices group, group + 2, ...
    const int w4 = wave & 3;
    const int wm = w4 >> 1, wn = w4 & 1;  // 2 x 2 waves of 64 x 64 inside the group
    const int lr = lane & 31, lh = lane >> 5;
    const int sw = (lr >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = ((ks * 2 + lh) ^ sw) << 4;
    const int xrow = (wn * 64 + lr) * KS;
    const int yrow = XB + (wm * 64 + lr) * KS;

    v16i acc[2][2]; // [n half][m half]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;

    for (int j = 0; j < npair; ++j) {
        __syncthreads();
        const int kt = 2 * j + group;
        if (kt < nk) {
            const char* base = smem + (kt % NST) * STAGE;
            v4i xf[2][2], yf[2][2]; // [buffer][half]: fragments of k-step ks + 1 are requested before the MFMAs of k-step ks
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                xf[0][h] = *reinterpret_cast<const v4i*>(base + xrow + h * 32 * KS + koff[0]);
                yf[0][h] = *reinterpret_cast<const v4i*>(base + yrow + h * 32 * KS + koff[0]);
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks + 1 < 4) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        xf[nxt][h] = *reinterpret_cast<const v4i*>(base + xrow + h * 32 * KS + koff[ks + 1]);
                        yf[nxt][h] = *reinterpret_cast<const v4i*>(base + yrow + h * 32 * KS + koff[ks + 1]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xf[cur][i], yf[cur][jj], acc[i][jj], 0, 0, 0);
            }
        }
    }

    // ---- the two groups swap halves: group 0 keeps n half 0 and adds group 1's, group 1 keeps n half 1 ---------------------------
    __syncthreads(); // B0
    v16i fin[2];
    {
        char* const mine = smem + wave * 8192;
        // (compile-time half index: a run-time index into acc[][] would put the accumulators into scratch memory)
        auto park = [&](auto half_tag) __attribute__((always_inline)) {
            constexpr int hx = decltype(half_tag)::value;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<v4i*>(mine + (jj * 4 + q) * 1024 + lane * 16) =
                        v4i{acc[hx][jj][4 * q], acc[hx][jj][4 * q + 1], acc[hx][jj][4 * q + 2], acc[hx][jj][4 * q + 3]};
        };
        if (group == 0) park(std::integral_constant<int, 1>{});
        else park(std::integral_constant<int, 0>{});
        __syncthreads(); // B1 (the loader waves' outlier copies have landed by now as well)
        const char* const theirs = smem + (wave ^ 4) * 8192;
        if (group == 0) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) fin[jj] = acc[0][jj];
        } else {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) fin[jj] = acc[1][jj];
        }
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const v4i t4 = *reinterpret_cast<const v4i*>(theirs + (jj * 4 + q) * 1024 + lane * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) fin[jj][4 * q + e] += t4[e];
            }
    }
    const int nh = group; // the n half of the wave's 64 columns this wave finishes

    // ---- K split over workgroups: park, count in, and only the last one to arrive goes on (gemm_kernels.hip, XSP) -------------
    if (XSP) {
        volatile unsigned& arrived_s = *reinterpret_cast<volatile unsigned*>(smem);
        constexpr int TILE_DW = 2 * 16 * TC; // dwords of one parked tile: [m half][16][512 compute threads]
        unsigned* const counter = static_cast<unsigned*>(p.splitk_ws) + t_lin;
        int* const slots = reinterpret_cast<int*>(static_cast<char*>(p.splitk_ws) + kSplitkWordsBytes) + (size_t)t_lin * XS * TILE_DW;
        int* const my = slots + (size_t)xrank * TILE_DW + tid;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                __hip_atomic_store(my + (jj * 16 + e) * TC, fin[jj][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every write-through acknowledged
        __syncthreads(); // B2 (also: everybody has read its partner's half, the first LDS word may be reused)
        if (tid == 0) arrived_s = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads(); // B3
        if (arrived_s != (unsigned)(XS - 1)) return;
        if (tid == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // re-armed
        for (int o = 1; o < XS; ++o) { // the other parts, in a rotation that depends on nothing but the rank
            const int* const theirs = slots + (size_t)((xrank + o) % XS) * TILE_DW + tid;
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    fin[jj][e] += __hip_atomic_load(theirs + (jj * 16 + e) * TC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // ---- epilogue: two 32 x 32 tiles per wave (gemm_kernels.hip's arithmetic, statement by statement) ----------------------------
    const char* const osmem = smem + EXCH;
    const int osteps = has_outliers ? (p.O + 15) / 16 : 0;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int m = m0 + wm * 64 + jj * 32 + lr;
        const int nb0 = n0 + wn * 64 + nh * 32 + 4 * lh;
        if (EPI == EPI_INT32) {
            if (m < p.M) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nb = nb0 + 8 * g;
                    if (nb < p.N) {
                        v4i o = {fin[jj][4 * g], fin[jj][4 * g + 1], fin[jj][4 * g + 2], fin[jj][4 * g + 3]};
                        *reinterpret_cast<v4i*>(static_cast<int32_t*>(p.D) + (int64_t)m * p.N + nb) = o;
                    }
                }
            }
            continue;
        }
        v16f P;
#pragma unroll
        for (int e = 0; e < 16; ++e) P[e] = 0.f;
        if (has_outliers) {
            const char* xo = osmem + (wn * 64 + nh * 32 + lr) * OSLICE;
            const char* yo = osmem + BN * OSLICE + (wm * 64 + jj * 32 + lr) * OSLICE;
            const int sw16 = lr & 15;
            if (osteps == 8) { // O = 128 (every shipped checkpoint): all sixteen fragment reads in flight, then eight MFMAs (same order, same sums)
                v8h xfo[8], yfo[8];
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    const int off = ((ks * 2 + lh) ^ sw16) << 4;
                    xfo[ks] = *reinterpret_cast<const v8h*>(xo + off);
                    yfo[ks] = *reinterpret_cast<const v8h*>(yo + off);
                }
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) P = __builtin_amdgcn_mfma_f32_32x32x16_f16(xfo[ks], yfo[ks], P, 0, 0, 0);
            } else {
                for (int ks = 0; ks < osteps; ++ks) {
                    const int off = ((ks * 2 + lh) ^ sw16) << 4;
                    v8h xfo = *reinterpret_cast<const v8h*>(xo + off);
                    v8h yfo = *reinterpret_cast<const v8h*>(yo + off);
                    P = __builtin_amdgcn_mfma_f32_32x32x16_f16(xfo, yfo, P, 0, 0, 0);
                }
            }
        }
        if (m < p.M) {
            const float sa = h2f(p.sA[m]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nb = nb0 + 8 * g;
                if (nb < p.N) {
                    const uint2 swb = *reinterpret_cast<const uint2*>(p.sW + nb);
                    const uint16_t swh[4] = {(uint16_t)(swb.x & 0xffffu), (uint16_t)(swb.x >> 16), (uint16_t)(swb.y & 0xffffu),
                                             (uint16_t)(swb.y >> 16)};
                    uint16_t yh[4] = {0, 0, 0, 0};
                    if (p.Y != nullptr) {
                        const uint2 yb = *reinterpret_cast<const uint2*>(p.Y + (int64_t)m * p.N + nb);
                        yh[0] = (uint16_t)(yb.x & 0xffffu), yh[1] = (uint16_t)(yb.x >> 16);
                        yh[2] = (uint16_t)(yb.y & 0xffffu), yh[3] = (uint16_t)(yb.y >> 16);
                    }
                    uint16_t oh[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // addend: the fp16-rounded outlier product (cuBLAS writes fp16) or the caller's y
                        const float c = has_outliers ? h2f(f2h_bits_of_f32_result(P[4 * g + e])) : h2f(yh[e]);
                        float v = __builtin_fmaf((float)fin[jj][4 * g + e], h2f(swh[e]) * sa, c);
                        if (epi_has_silu(EPI)) v = silu_f32(v);
                        oh[e] = f2h_bits_of_f32_result(v);
                    }
                    if (EPI == EPI_DEQUANT_SILU_MUL) { // gate * up: one fp16 multiply of the rounded result
                        const uint2 mb = *reinterpret_cast<const uint2*>(p.Mul + (int64_t)m * p.N + nb);
                        const uint16_t mh[4] = {(uint16_t)(mb.x & 0xffffu), (uint16_t)(mb.x >> 16), (uint16_t)(mb.y & 0xffffu),
                                                (uint16_t)(mb.y >> 16)};
#pragma unroll
                        for (int e = 0; e < 4; ++e) oh[e] = f2h_bits(h2f(oh[e]) * h2f(mh[e]));
                    }
                    uint2 o;
                    o.x = (unsigned)oh[0] | ((unsigned)oh[1] << 16);
                    o.y = (unsigned)oh[2] | ((unsigned)oh[3] << 16);
                    *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.D) + (int64_t)m * p.N + nb) = o;
                }
            }
        }
    }
}

template <int EPI>
static hipError_t launch_mid_epi(const GemmParams& p, hipStream_t st)
{
    GemmParams q = p;
    // one tile row: every weight line is read by exactly one workgroup -> non-temporal copies for weights of 32 MiB and more
    // (the rule of gemm_kernels.hip's launch_cfg)
    if (p.M <= mid::BM && (int64_t)p.N * p.K >= ((int64_t)32 << 20)) q.flags |= 2;
    const int tiles = ((p.M + mid::BM - 1) / mid::BM) * ((p.N + mid::BN - 1) / mid::BN);
    if (p.xsplit > 1) {
        auto kern = gemm_w8a8o16_mid_kernel<EPI, true>;
        static DeviceOnce once;
        if (hipError_t e = ensure_dynamic_lds(kern, mid::LDS, once); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * p.xsplit)), dim3(mid::T), mid::LDS, st, q);
    } else {
        auto kern = gemm_w8a8o16_mid_kernel<EPI, false>;
        static DeviceOnce once;
        if (hipError_t e = ensure_dynamic_lds(kern, mid::LDS, once); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(mid::T), mid::LDS, st, q);
    }
    return hipGetLastError();
}

// p.xsplit = workgroups per tile along K (1: the tiles alone); K % 128 == 0
hipError_t launch_gemm_mid(const GemmParams& p, int epi, hipStream_t st)
{
    if (p.K % mid::KS != 0 || p.K <= 0) return hipErrorInvalidValue;
    switch (epi) {
    case EPI_DEQUANT: return launch_mid_epi<EPI_DEQUANT>(p, st);
    case EPI_DEQUANT_SILU: return launch_mid_epi<EPI_DEQUANT_SILU>(p, st);
    case EPI_DEQUANT_SILU_MUL: return launch_mid_epi<EPI_DEQUANT_SILU_MUL>(p, st);
    default: return launch_mid_epi<EPI_INT32>(p, st);
    }
}

} // namespace mixq

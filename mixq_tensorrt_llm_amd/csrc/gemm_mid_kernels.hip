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
//     keeps: integer adds, same sums), so the epilogue runs on all eight waves; the outlier operands are copied by the loader
//     waves behind the LAST loop barrier into the two stages the last-but-one pair has handed back, and the scales / addends the
//     epilogue needs were requested at kernel start: nothing in the epilogue waits for global memory.
// K split over XS workgroups per tile ("last arriver adds", as in gemm_kernels.hip) on the post-exchange layout.
#include "mixq_device.h"
#include "mixq_launch.h"
#include <atomic>
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

// ---- what a compute wave does behind the main loop (both schedules): swap halves, K split over workgroups, epilogue ------------
// `acc`: the wave's 64 x 64 sums over its group's slices; barriers B0, B1 (+ B2, B3 with XSP) are executed by EVERY wave of the
// workgroup (the copy-only waves run mid_loader_tail next to this).
// LDS regions (byte offsets, wave-uniform): ex0 / ex1 = 32 KiB each, the exchange slots of waves 0-3 / 4-7; ow / oa = the fpW / fpA tiles
// (32 KiB each) the copy-only waves staged.
// What the epilogue reads from global memory, requested by a compute wave at KERNEL START and carried through the loop (10 registers, + 16
// with an addend; the multiplicand of gate * up is requested in front of the exchange instead: 16 registers more through the loop spill): the first build loaded sA / sW / y inside the store loop -- a dependent L2 round trip per 8-byte store
// group, 3.8 us of epilogue per tile in its timeline (profiles/r06_mid_v1_timeline.txt).  Clamped addresses: rows / columns past the edge
// are never stored.
struct MidEpiPre {
    float sa[2];
    uint2 sw[4], y[2][4];
};
template <int EPI>
__device__ __forceinline__ void mid_epi_prefetch(MidEpiPre& e, const GemmParams& p, int wave, int lane, int m0, int n0)
{
    const int group = wave >> 2, w4 = wave & 3, wm = w4 >> 1, wn = w4 & 1;
    const int lr = lane & 31, lh = lane >> 5;
    if (EPI == EPI_INT32) return;
    const int nb0p = n0 + wn * 64 + group * 32 + 4 * lh;
#pragma unroll
    for (int g = 0; g < 4; ++g) e.sw[g] = *reinterpret_cast<const uint2*>(p.sW + min(nb0p + 8 * g, p.N - 4));
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const int mp = min(m0 + wm * 64 + jj * 32 + lr, p.M - 1);
        e.sa[jj] = h2f(p.sA[mp]);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t at = (int64_t)mp * p.N + min(nb0p + 8 * g, p.N - 4);
            e.y[jj][g] = p.Y != nullptr ? *reinterpret_cast<const uint2*>(p.Y + at) : make_uint2(0u, 0u);
        }
    }
}

template <int EPI, bool XSP, int TC>
__device__ __forceinline__ void mid_finish(const GemmParams& p, char* smem, v16i (&acc)[2][2], int wave, int lane, int tid, int m0, int n0,
                                           int t_lin, int xrank, int XS, const MidEpiPre& pre, int ex0, int ex1, int ow, int oa)
{
    using namespace mid;
    const int group = wave >> 2, w4 = wave & 3, wm = w4 >> 1, wn = w4 & 1;
    const int lr = lane & 31, lh = lane >> 5;
    const bool has_outliers = p.O > 0;
    dbg_stamp(p.dbg, 3); // last MFMA issued (this wave)
    uint2 mulq[2][4]; // (the multiplicand of gate * up: requested here, in front of the exchange)
#pragma unroll
    for (int jj = 0; jj < 2; ++jj)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int64_t at = (int64_t)min(m0 + wm * 64 + jj * 32 + lr, p.M - 1) * p.N + min(n0 + wn * 64 + group * 32 + 4 * lh + 8 * g, p.N - 4);
            mulq[jj][g] = EPI == EPI_DEQUANT_SILU_MUL ? *reinterpret_cast<const uint2*>(p.Mul + at) : make_uint2(0u, 0u);
        }
    // ---- the two groups swap halves: group 0 keeps n half 0 and adds group 1's, group 1 keeps n half 1 ---------------------------
    __syncthreads(); // B0
    v16i fin[2];
    {
        char* const mine = smem + (wave < 4 ? ex0 : ex1) + (wave & 3) * 8192;
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
        const char* const theirs = smem + (wave < 4 ? ex1 : ex0) + (wave & 3) * 8192;
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
    dbg_stamp(p.dbg, 4); // halves swapped, outlier operands in LDS

    // ---- K split over workgroups: park, count in, and only the last one to arrive goes on (gemm_kernels.hip, XSP) -------------
    if (XSP) {
        // (the word lives in the exchange region, dead behind B2; explicit LDS address space: a generic pointer becomes a FLAT access, whose
        //  LDS aperture ends at 64 KiB)
        typedef __attribute__((address_space(3))) volatile unsigned lds_vu32;
        lds_vu32& arrived_s = *(lds_vu32*)MIXQ_LDS_PTR(smem + ex0);
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
            const char* xo = smem + ow + (wn * 64 + nh * 32 + lr) * OSLICE;
            const char* yo = smem + oa + (wm * 64 + jj * 32 + lr) * OSLICE;
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
            const float sa = pre.sa[jj];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nb = nb0 + 8 * g;
                if (nb < p.N) {
                    const uint2 swb = pre.sw[g];
                    const uint16_t swh[4] = {(uint16_t)(swb.x & 0xffffu), (uint16_t)(swb.x >> 16), (uint16_t)(swb.y & 0xffffu),
                                             (uint16_t)(swb.y >> 16)};
                    uint16_t yh[4] = {0, 0, 0, 0};
                    if (p.Y != nullptr) {
                        const uint2 yb = pre.y[jj][g];
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
                        const uint2 mb = mulq[jj][g];
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
    if (p.dbg != nullptr) {
        dbg_stamp(p.dbg, 5); // stores issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dbg_stamp(p.dbg, 6); // stores acknowledged
    }
}

// ---- copy-only waves: the outlier operands -> LDS (wave lw = 0: the fpW tile, 1: the fpA tile; 32 copies of 1 KiB, not waited for here) -----
__device__ __forceinline__ void mid_stage_outliers(const GemmParams& p, unsigned lds_dst, int lw, int lane, int m0, int n0)
{
    using namespace mid;
    const int rows_total = lw == 0 ? p.N : p.M, r0 = lw == 0 ? n0 : m0;
    const int obytes = p.O * 2;
    const char* const ob = lw == 0 ? reinterpret_cast<const char*>(p.fpW) : reinterpret_cast<const char*>(p.fpA);
#pragma unroll 8
    for (int q = 0; q < 32; ++q) { // 256-B rows, slot = chunk ^ (row & 15); chunks past O come from the zero page
        const int row = q * 4 + (lane >> 4);
        const int c = ((lane & 15) ^ (row & 15)) << 4;
        const int grow = min(r0 + row, rows_total - 1);
        const char* s = ob + (int64_t)grow * obytes + c;
        if (c >= obytes) s = static_cast<const char*>(p.zeros);
        glds16_vaddr(s, lds_dst + q * 1024);
    }
}

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
        // Tiles that share a W panel (the tile rows of one tile column) run at the same time on neighbouring CUs.  Walking K from the same
        // slice they ask for the same cold lines at the same moment: one is the miss, the others wait on it, and every one of them has a queue
        // slot tied up for the whole HBM round trip (a CU keeps ~32 KiB of requests outstanding, R6.5).  Each tile row therefore STARTS at a
        // different slice of its K range (integer sums commute: same bits): a line is then fetched from HBM by one CU and found in L2 / the
        // Infinity Cache by the others a few microseconds later.  (p.flags bit 2 off: measurement knob 1411.)
        const int rot = (p.flags & 4) ? (int)(((int64_t)tile_m * nk) / tiles_m + ((tile_n * 5) % 7)) % nk : 0;
        auto issue = [&](int s) __attribute__((always_inline)) {
            const unsigned dst = lds0 + (unsigned)(s % NST) * STAGE + lw * XB;
            int sr = s + rot;
            if (sr >= nk) sr -= nk;
            const char* b = base + (int64_t)sr * KS; // wave-uniform
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
        unsigned long long t_wait = 0, t_bar = 0; // (measurement only, p.dbg != NULL: 100 MHz ticks this wave spent waiting for its copies / at the barrier)
        for (int j = 0; j < npair; ++j) {
            // issued so far: slices .. 2j + 2; the pair 2j, 2j + 1 must have landed (copies complete in order)
            const unsigned long long ta = p.dbg ? wall_clock64() : 0;
            if (2 * j + 2 < nk) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long tb = p.dbg ? wall_clock64() : 0;
            __syncthreads(); // pair j handed over; the compute waves are done with pair j - 1
            if (p.dbg) t_wait += tb - ta, t_bar += wall_clock64() - tb;
            if (2 * j + 3 < nk) issue(2 * j + 3);
            if (2 * j + 4 < nk) issue(2 * j + 4);
        }
        // the outlier operands go into the two stages the LAST BUT ONE pair has just handed back (nothing is issued into them any more), so
        // that they land under the last pair's MFMAs and the exchange instead of behind it; the exchange takes the last pair's own stages
        const bool has_o = EPI != EPI_INT32 && has_outliers;
        if (has_o) mid_stage_outliers(p, lds0 + (unsigned)((2 * npair + 1 + lw) % NST) * STAGE, lw, lane, m0, n0);
        if (p.dbg != nullptr && lane == 0) // slot 7: the W loader, slot 2: the qA loader; (ticks waiting for copies) << 32 | ticks at the barrier
            static_cast<unsigned long long*>(p.dbg)[(size_t)blockIdx.x * 8 + (lw == 0 ? 7 : 2)] = (t_wait << 32) | (t_bar & 0xffffffffull);
        __syncthreads(); // B0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the outlier tiles
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

    MidEpiPre pre;
    mid_epi_prefetch<EPI>(pre, p, wave, lane, m0, n0);
    v16i acc[2][2]; // [n half][m half]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;

    dbg_stamp(p.dbg, 0); // (measurement only: p.dbg is NULL in production) entry
    for (int j = 0; j < npair; ++j) {
        __syncthreads();
        if (j == 0) dbg_stamp(p.dbg, 1);           // first pair of slices handed over
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

    // (pair P = npair - 1 is the last: its slices 2P, 2P + 1 sit in stages 2P % 5, (2P + 1) % 5 -> the exchange; the pair before it in
    //  (2P + 3) % 5, (2P + 4) % 5 -> the outlier tiles)
    mid_finish<EPI, XSP, TC>(p, smem, acc, wave, lane, tid, m0, n0, t_lin, xrank, XS, pre, ((2 * npair - 2) % NST) * STAGE, ((2 * npair - 1) % NST) * STAGE,
                             ((2 * npair + 1) % NST) * STAGE, ((2 * npair + 2) % NST) * STAGE);
}

static std::atomic<int> g_mid_rot{1}; // measurement knob 1410 (default: where the tiles are not split along K) / 1411 (never) / 1412 (always): tile rows start at different K slices
void set_mid_rot(int mode) { g_mid_rot.store(mode); }

template <int EPI>
static hipError_t launch_mid_epi(const GemmParams& p, hipStream_t st)
{
    GemmParams q = p;
    // one tile row: every weight line is read by exactly one workgroup -> non-temporal copies for weights of 32 MiB and more
    // (the rule of gemm_kernels.hip's launch_cfg)
    if (p.M <= mid::BM && (int64_t)p.N * p.K >= ((int64_t)32 << 20)) q.flags |= 2;
    // (the walk is rotated where the tiles alone fill the chip: cold, h vs j in profiles/r06_mid_final_sweep_cold.txt, ahead in every such cell, by up
    //  to 16 %.  With K split over workgroups the parts of a tile already start at different slices, and rotating them as well is BIMODAL --
    //  4096 x 11008 at 256 rows with four parts ran 29.5 us in one process and 35 us in the next, at 512 rows with two parts 39.7 / 47.9 -- and
    //  tripped the selection gate (+16.8 % behind the round-5 build): not rotated.  Knob 1412 rotates always, for measurements.)
    const int rotm = g_mid_rot.load();
    if (rotm == 2 || (rotm == 1 && p.xsplit <= 1)) q.flags |= 4;
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

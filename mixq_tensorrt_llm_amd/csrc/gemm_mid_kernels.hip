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
constexpr int BM = 128, KS = 128;
constexpr int YB = BM * KS;      // 16 KiB: the qA rows of a slice
constexpr int NST = 5;
constexpr int NCW = 8, NLD = 2, TC = NCW * 64, T = (NCW + NLD) * 64;
constexpr int OSLICE = 256;      // bytes per LDS row in the outlier phase (128 fp16)
constexpr int GROUP_M = 4;

// Tile width BN and the layout of a group's four compute waves over the 128 x BN tile (wave tile = TMW x TNW MFMA tiles of 32 x 32):
//   BN = 128: 2 x 2 waves of 64 x 64;   BN = 96: 4 x 1 waves of 32 x 96 -- more, narrower tiles for shapes whose 128-wide tiles leave a quarter
//   of the CUs idle (11008 x 4096 at 129..256 rows: 230 workgroups instead of 172; gemm_mid_tile_width below has the measurements).
template <int BN_>
struct Layout {
    static constexpr int BN = BN_;
    static constexpr int WMW = BN_ == 128 ? 2 : 4, WNW = BN_ == 128 ? 2 : 1; // waves of a group along m / n
    static constexpr int TMW = BM / WMW / 32, TNW = BN_ / WNW / 32;          // MFMA tiles of a wave tile along m / n
    static constexpr int WMT = BM / WMW, WNT = BN_ / WNW;                    // wave tile in elements
    static constexpr int XB = BN_ * KS, STAGE = XB + YB;
    static constexpr int PERX = BN_ * 8 / 64, PERY = BM * 8 / 64;             // copies per slice of the W / qA loader wave
    static constexpr int HT = TNW * TMW * 2, KEEP = HT / 2;                   // half tiles (8 accumulator registers) of a wave tile; kept per group
    static constexpr size_t LDS = (size_t)NST * STAGE;
    static_assert(HT % 2 == 0 && WMW * WNW == 4, "four waves per group");
    static_assert(2 * STAGE >= 8 * KEEP * 2048, "the exchange (8 waves x KEEP half tiles x 2 KiB) must fit two stages");
    static_assert(2 * STAGE >= (BM + BN_) * OSLICE, "the two outlier tiles must fit two adjacent stages");
};
} // namespace mid

// What the epilogue reads from global memory, requested by a compute wave at KERNEL START and carried through the loop (per kept half tile:
// the row scale, two weight-scale quads, two addend quads; the multiplicand of gate * up is requested in front of the exchange instead --
// 16 registers more through the loop spill): the first build loaded sA / sW / y inside the store loop -- a dependent L2 round trip per
// 8-byte store group, 3.8 us of epilogue per tile in its timeline (profiles/r06_mid_v1_timeline.txt).  Clamped addresses: rows / columns
// past the edge are never stored.
template <int KEEP>
struct MidEpiPre {
    float sa[KEEP];
    uint2 sw[KEEP][2], y[KEEP][2];
};

// half tile h of a wave tile (order: n tile i, m tile j, half gh): rows m_of(h), first column n_of(h) of this lane's quads g2 = 0, 1 (+ 8 g2)
template <class L>
struct HalfTile {
    int i, j, gh;
    __device__ __forceinline__ explicit HalfTile(int h) : i(h / (2 * L::TMW)), j((h >> 1) % L::TMW), gh(h & 1) {}
    __device__ __forceinline__ int m(int m0, int wm, int lr) const { return m0 + wm * L::WMT + j * 32 + lr; }
    __device__ __forceinline__ int n(int n0, int wn, int lh) const { return n0 + wn * L::WNT + i * 32 + 4 * lh + 16 * gh; }
};

template <int EPI, class L>
__device__ __forceinline__ void mid_epi_prefetch(MidEpiPre<L::KEEP>& e, const GemmParams& p, int wave, int lane, int m0, int n0)
{
    const int group = wave >> 2, w4 = wave & 3, wm = w4 / L::WNW, wn = w4 % L::WNW;
    const int lr = lane & 31, lh = lane >> 5;
    if (EPI == EPI_INT32) return;
#pragma unroll
    for (int k = 0; k < L::KEEP; ++k) {
        const HalfTile<L> t(group * L::KEEP + k);
        const int mp = min(t.m(m0, wm, lr), p.M - 1);
        e.sa[k] = h2f(p.sA[mp]);
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            const int nb = min(t.n(n0, wn, lh) + 8 * g2, p.N - 4);
            e.sw[k][g2] = *reinterpret_cast<const uint2*>(p.sW + nb);
            e.y[k][g2] = p.Y != nullptr ? *reinterpret_cast<const uint2*>(p.Y + (int64_t)mp * p.N + nb) : make_uint2(0u, 0u);
        }
    }
}

// ---- what a compute wave does behind the main loop: swap half tiles, K split over workgroups, epilogue ---------------------------------
// `acc`: the wave tile's sums over its group's slices; barriers B0, B1 (+ B2, B3 with XSP) are executed by EVERY wave of the workgroup.
// Group 0 keeps the first KEEP half tiles of every wave tile and adds group 1's sums of them, group 1 the rest (integer adds: same sums).
// LDS regions (byte offsets, wave-uniform): ex0 / ex1 = the exchange slots of waves 0-3 / 4-7 (KEEP x 2 KiB each); ow / oa = the fpW / fpA
// tiles the copy-only waves staged.
template <int EPI, bool XSP, class L>
__device__ __forceinline__ void mid_finish(const GemmParams& p, char* smem, v16i (&acc)[L::TNW][L::TMW], int wave, int lane, int tid, int m0,
                                           int n0, int t_lin, int xrank, int XS, const MidEpiPre<L::KEEP>& pre, int ex0, int ex1, int ow, int oa)
{
    using namespace mid;
    constexpr int KEEP = L::KEEP, TMW = L::TMW;
    const int group = wave >> 2, w4 = wave & 3, wm = w4 / L::WNW, wn = w4 % L::WNW;
    const int lr = lane & 31, lh = lane >> 5;
    const bool has_outliers = p.O > 0;
    dbg_stamp(p.dbg, 3); // last MFMA issued (this wave)
    uint2 mulq[KEEP][2]; // (the multiplicand of gate * up: requested here, in front of the exchange)
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
        const HalfTile<L> t(group * KEEP + k);
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            const int64_t at = (int64_t)min(t.m(m0, wm, lr), p.M - 1) * p.N + min(t.n(n0, wn, lh) + 8 * g2, p.N - 4);
            mulq[k][g2] = EPI == EPI_DEQUANT_SILU_MUL ? *reinterpret_cast<const uint2*>(p.Mul + at) : make_uint2(0u, 0u);
        }
    }
    __syncthreads(); // B0
    int fin[KEEP][8];
    {
        char* const mine = smem + (wave < 4 ? ex0 : ex1) + (wave & 3) * (KEEP * 2048);
        const char* const theirs = smem + (wave < 4 ? ex1 : ex0) + (wave & 3) * (KEEP * 2048);
        // (compile-time half-tile indices: a run-time index into acc[][] would put the accumulators into scratch memory)
        auto swap = [&](auto give0_tag, auto keep0_tag) __attribute__((always_inline)) {
            constexpr int G0 = decltype(give0_tag)::value, K0 = decltype(keep0_tag)::value;
#pragma unroll
            for (int k = 0; k < KEEP; ++k) {
                constexpr int dummy = 0;
                (void)dummy;
                const int h = G0 + k; // given away
                const int i = h / (2 * TMW), j = (h >> 1) % TMW, gh = h & 1;
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    *reinterpret_cast<v4i*>(mine + (k * 2 + q) * 1024 + lane * 16) =
                        v4i{acc[i][j][8 * gh + 4 * q], acc[i][j][8 * gh + 4 * q + 1], acc[i][j][8 * gh + 4 * q + 2], acc[i][j][8 * gh + 4 * q + 3]};
            }
            __syncthreads(); // B1 (the loader waves' outlier copies have landed by now as well)
#pragma unroll
            for (int k = 0; k < KEEP; ++k) {
                const int h = K0 + k; // kept
                const int i = h / (2 * TMW), j = (h >> 1) % TMW, gh = h & 1;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const v4i t4 = *reinterpret_cast<const v4i*>(theirs + (k * 2 + q) * 1024 + lane * 16);
#pragma unroll
                    for (int e = 0; e < 4; ++e) fin[k][4 * q + e] = acc[i][j][8 * gh + 4 * q + e] + t4[e];
                }
            }
        };
        if (group == 0) swap(std::integral_constant<int, KEEP>{}, std::integral_constant<int, 0>{});
        else swap(std::integral_constant<int, 0>{}, std::integral_constant<int, KEEP>{});
    }
    dbg_stamp(p.dbg, 4); // half tiles swapped, outlier operands in LDS

    // ---- K split over workgroups: park, count in, and only the last one to arrive goes on (gemm_kernels.hip, XSP) -------------
    if (XSP) {
        // (the word lives in the exchange region, dead behind B2; explicit LDS address space: a generic pointer becomes a FLAT access, whose
        //  LDS aperture ends at 64 KiB)
        typedef __attribute__((address_space(3))) volatile unsigned lds_vu32;
        lds_vu32& arrived_s = *(lds_vu32*)MIXQ_LDS_PTR(smem + ex0);
        constexpr int TILE_DW = KEEP * 8 * TC; // dwords of one parked tile: [kept half tile][8][512 compute threads]
        unsigned* const counter = static_cast<unsigned*>(p.splitk_ws) + t_lin;
        int* const slots = reinterpret_cast<int*>(static_cast<char*>(p.splitk_ws) + kSplitkWordsBytes) + (size_t)t_lin * XS * TILE_DW;
        int* const my = slots + (size_t)xrank * TILE_DW + tid;
#pragma unroll
        for (int k = 0; k < KEEP; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                __hip_atomic_store(my + (k * 8 + e) * TC, fin[k][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every write-through acknowledged
        __syncthreads(); // B2 (also: everybody has read its partner's half tiles, the first LDS word may be reused)
        if (tid == 0) arrived_s = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads(); // B3
        if (arrived_s != (unsigned)(XS - 1)) return;
        if (tid == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // re-armed
        for (int o = 1; o < XS; ++o) { // the other parts, in a rotation that depends on nothing but the rank
            const int* const theirs = slots + (size_t)((xrank + o) % XS) * TILE_DW + tid;
#pragma unroll
            for (int k = 0; k < KEEP; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    fin[k][e] += __hip_atomic_load(theirs + (k * 8 + e) * TC, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // ---- epilogue, one kept half tile (32 rows x 16 columns: two quads per lane) at a time; gemm_kernels.hip's arithmetic -------------------
    const int osteps = has_outliers ? (p.O + 15) / 16 : 0;
    v16f P;
#pragma unroll
    for (int e = 0; e < 16; ++e) P[e] = 0.f;
    int ptile = -1; // the MFMA tile whose outlier product P holds
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
        const HalfTile<L> t(group * KEEP + k);
        const int m = t.m(m0, wm, lr);
        const int nb0 = t.n(n0, wn, lh);
        if (EPI == EPI_INT32) {
            if (m < p.M) {
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const int nb = nb0 + 8 * g2;
                    if (nb < p.N) {
                        v4i o = {fin[k][4 * g2], fin[k][4 * g2 + 1], fin[k][4 * g2 + 2], fin[k][4 * g2 + 3]};
                        *reinterpret_cast<v4i*>(static_cast<int32_t*>(p.D) + (int64_t)m * p.N + nb) = o;
                    }
                }
            }
            continue;
        }
        const int tile = (group * KEEP + k) >> 1;
        if (has_outliers && tile != ptile) { // (wave-uniform) both halves of a tile share its outlier product
            ptile = tile;
#pragma unroll
            for (int e = 0; e < 16; ++e) P[e] = 0.f;
            const char* xo = smem + ow + (wn * L::WNT + t.i * 32 + lr) * OSLICE;
            const char* yo = smem + oa + (wm * L::WMT + t.j * 32 + lr) * OSLICE;
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
            const float sa = pre.sa[k];
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                const int nb = nb0 + 8 * g2;
                if (nb < p.N) {
                    const uint2 swb = pre.sw[k][g2];
                    const uint16_t swh[4] = {(uint16_t)(swb.x & 0xffffu), (uint16_t)(swb.x >> 16), (uint16_t)(swb.y & 0xffffu),
                                             (uint16_t)(swb.y >> 16)};
                    uint16_t yh[4] = {0, 0, 0, 0};
                    if (p.Y != nullptr) {
                        const uint2 yb = pre.y[k][g2];
                        yh[0] = (uint16_t)(yb.x & 0xffffu), yh[1] = (uint16_t)(yb.x >> 16);
                        yh[2] = (uint16_t)(yb.y & 0xffffu), yh[3] = (uint16_t)(yb.y >> 16);
                    }
                    uint16_t oh[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        // addend: the fp16-rounded outlier product (cuBLAS writes fp16) or the caller's y.  P element of (half gh, quad g2, e): 8 gh + 4 g2 + e
                        const float pe = t.gh ? P[8 + 4 * g2 + e] : P[4 * g2 + e];
                        const float c = has_outliers ? h2f(f2h_bits_of_f32_result(pe)) : h2f(yh[e]);
                        float v = __builtin_fmaf((float)fin[k][4 * g2 + e], h2f(swh[e]) * sa, c);
                        if (epi_has_silu(EPI)) v = silu_f32(v);
                        oh[e] = f2h_bits_of_f32_result(v);
                    }
                    if (EPI == EPI_DEQUANT_SILU_MUL) { // gate * up: one fp16 multiply of the rounded result
                        const uint2 mb = mulq[k][g2];
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

// ---- copy-only waves: an outlier operand tile -> LDS (`rows` rows of 256 B: rows / 4 copies of 1 KiB, not waited for here) -----------------
__device__ __forceinline__ void mid_stage_outliers(const GemmParams& p, unsigned lds_dst, int lw, int lane, int m0, int n0, int rows)
{
    using namespace mid;
    const int rows_total = lw == 0 ? p.N : p.M, r0 = lw == 0 ? n0 : m0;
    const int obytes = p.O * 2;
    const char* const ob = lw == 0 ? reinterpret_cast<const char*>(p.fpW) : reinterpret_cast<const char*>(p.fpA);
#pragma unroll 8
    for (int q = 0; q < rows / 4; ++q) { // 256-B rows, slot = chunk ^ (row & 15); chunks past O come from the zero page
        const int row = q * 4 + (lane >> 4);
        const int c = ((lane & 15) ^ (row & 15)) << 4;
        const int grow = min(r0 + row, rows_total - 1);
        const char* s = ob + (int64_t)grow * obytes + c;
        if (c >= obytes) s = static_cast<const char*>(p.zeros);
        glds16_vaddr(s, lds_dst + q * 1024);
    }
}

template <int EPI, bool XSP, int BN>
__global__ __launch_bounds__(mid::T) void gemm_w8a8o16_mid_kernel(const GemmParams p)
{
    using namespace mid;
    using L = Layout<BN>;
    constexpr int XB = L::XB, STAGE = L::STAGE, TMW = L::TMW, TNW = L::TNW;
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
    // (pair P = npair - 1 is the last: its slices 2P, 2P + 1 sit in stages 2P % 5, (2P + 1) % 5 -> the exchange, behind B0.  The other three
    //  stages -- a = (2P + 2) % 5, a + 1, a + 2 (mod 5) -- are free once the last loop barrier is passed; two of them are adjacent in memory
    //  (a, a + 1 unless a = 4: then 0, 1) -> the fpA tile (32 KiB) followed by the fpW tile (BN x 256 B), which may straddle the stage border)
    const int st_ex0 = ((2 * npair - 2) % NST) * STAGE, st_ex1 = ((2 * npair - 1) % NST) * STAGE;
    const int st_a = (2 * npair) % NST;
    const int st_oa = (st_a <= 3 ? st_a : 0) * STAGE, st_ow = st_oa + BM * OSLICE;

    if (wave >= NCW) {
        // =================================== loader wave: copies and barriers, nothing else ===================================
        const int lw = wave - NCW; // 0: W rows -> region X, 1: qA rows -> region Y
        const int per = lw == 0 ? L::PERX : L::PERY; // (wave-uniform)
        const int rows_total = lw == 0 ? p.N : p.M, r0 = lw == 0 ? n0 : m0;
        unsigned voff[L::PERY];
#pragma unroll
        for (int i = 0; i < L::PERY; ++i) { // LDS row = i * 8 + lane / 8, 16-B slot lane % 8 holds source chunk slot ^ ((row >> 1) & 7)
            const int row = i * 8 + (lane >> 3);
            const int rr = min(r0 + row, rows_total - 1) - r0; // clamp: rows past the edge are copied but never stored
            voff[i] = (unsigned)rr * (unsigned)p.K + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
        }
        const char* const base = (lw == 0 ? reinterpret_cast<const char*>(p.B) : reinterpret_cast<const char*>(p.A)) +
                                 (int64_t)r0 * K + (int64_t)kbeg * KS;
        const bool nt = lw == 0 && (p.flags & 2) != 0;
        // (measurement option, off by default -- launch_mid_epi: tiles that share a W panel start at DIFFERENT slices of their K range, so that a
        //  cold line is fetched from HBM by one CU and found in L2 / the Infinity Cache by the others later; integer sums commute: same bits)
        const int rot = (p.flags & 4) ? (int)(((int64_t)tile_m * nk) / tiles_m + ((tile_n * 5) % 7)) % nk : 0;
        auto issue = [&](int s) __attribute__((always_inline)) {
            const unsigned dst = lds0 + (unsigned)(s % NST) * STAGE + lw * XB;
            int sr = s + rot;
            if (sr >= nk) sr -= nk;
            const char* b = base + (int64_t)sr * KS; // wave-uniform
            if (nt) {
#pragma unroll
                for (int i = 0; i < L::PERX; ++i) glds16_sbase_nt(b, voff[i], dst + i * 1024);
            } else if (lw == 0) {
#pragma unroll
                for (int i = 0; i < L::PERX; ++i) glds16_sbase(b, voff[i], dst + i * 1024);
            } else {
#pragma unroll
                for (int i = 0; i < L::PERY; ++i) glds16_sbase(b, voff[i], dst + i * 1024);
            }
        };
#pragma unroll
        for (int s = 0; s < 3; ++s)
            if (s < nk) issue(s);
        unsigned long long t_wait = 0, t_bar = 0; // (measurement only, p.dbg != NULL: 100 MHz ticks this wave spent waiting for its copies / at the barrier)
        for (int j = 0; j < npair; ++j) {
            // issued so far: slices .. 2j + 2; the pair 2j, 2j + 1 must have landed (copies complete in order)
            const unsigned long long ta = p.dbg ? wall_clock64() : 0;
            if (2 * j + 2 >= nk) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else if (lw == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L::PERX) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L::PERY) : "memory");
            const unsigned long long tb = p.dbg ? wall_clock64() : 0;
            __syncthreads(); // pair j handed over; the compute waves are done with pair j - 1
            if (p.dbg) t_wait += tb - ta, t_bar += wall_clock64() - tb;
            if (2 * j + 3 < nk) issue(2 * j + 3);
            if (2 * j + 4 < nk) issue(2 * j + 4);
        }
        (void)per;
        // the outlier operands go into the two stages the LAST BUT ONE pair has just handed back (nothing is issued into them any more), so
        // that they land under the last pair's MFMAs and the exchange instead of behind it; the exchange takes the last pair's own stages
        if (EPI != EPI_INT32 && has_outliers) mid_stage_outliers(p, lds0 + (unsigned)(lw == 0 ? st_ow : st_oa), lw, lane, m0, n0, lw == 0 ? BN : BM);
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
    const int wm = w4 / L::WNW, wn = w4 % L::WNW;
    const int lr = lane & 31, lh = lane >> 5;
    const int sw = (lr >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = ((ks * 2 + lh) ^ sw) << 4;
    const int xrow = (wn * L::WNT + lr) * KS;
    const int yrow = XB + (wm * L::WMT + lr) * KS;

    MidEpiPre<L::KEEP> pre;
    mid_epi_prefetch<EPI, L>(pre, p, wave, lane, m0, n0);
    v16i acc[TNW][TMW]; // [n tile][m tile]
#pragma unroll
    for (int i = 0; i < TNW; ++i)
#pragma unroll
        for (int j = 0; j < TMW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;

    dbg_stamp(p.dbg, 0); // (measurement only: p.dbg is NULL in production) entry
    for (int j = 0; j < npair; ++j) {
        __syncthreads();
        if (j == 0) dbg_stamp(p.dbg, 1);           // first pair of slices handed over
        const int kt = 2 * j + group;
        if (kt < nk) {
            const char* base = smem + (kt % NST) * STAGE;
            v4i xf[2][TNW], yf[2][TMW]; // [buffer][tile]: fragments of k-step ks + 1 are requested before the MFMAs of k-step ks
#pragma unroll
            for (int h = 0; h < TNW; ++h) xf[0][h] = *reinterpret_cast<const v4i*>(base + xrow + h * 32 * KS + koff[0]);
#pragma unroll
            for (int h = 0; h < TMW; ++h) yf[0][h] = *reinterpret_cast<const v4i*>(base + yrow + h * 32 * KS + koff[0]);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int cur = ks & 1, nxt = cur ^ 1;
                if (ks + 1 < 4) {
#pragma unroll
                    for (int h = 0; h < TNW; ++h) xf[nxt][h] = *reinterpret_cast<const v4i*>(base + xrow + h * 32 * KS + koff[ks + 1]);
#pragma unroll
                    for (int h = 0; h < TMW; ++h) yf[nxt][h] = *reinterpret_cast<const v4i*>(base + yrow + h * 32 * KS + koff[ks + 1]);
                }
#pragma unroll
                for (int i = 0; i < TNW; ++i)
#pragma unroll
                    for (int jj = 0; jj < TMW; ++jj)
                        acc[i][jj] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xf[cur][i], yf[cur][jj], acc[i][jj], 0, 0, 0);
            }
        }
    }
    mid_finish<EPI, XSP, L>(p, smem, acc, wave, lane, tid, m0, n0, t_lin, xrank, XS, pre, st_ex0, st_ex1, st_ow, st_oa);
}

static std::atomic<int> g_mid_rot{3}; // measurement knob 1413 (default: by rule) / 1411 (never) / 1410 (where the tiles are not split along K) / 1412 (always): tile rows start at different K slices
void set_mid_rot(int mode) { g_mid_rot.store(mode); }
static std::atomic<int> g_mid_bn{0};  // measurement knob 1430 (default: by rule) / 1431 (always 128) / 1432 (96-wide only where a sixteenth of the CUs stays free) / 1433 (96-wide only without K split)
void set_mid_bn(int mode) { g_mid_bn.store(mode); }

// 96-wide tiles where they put MORE workgroups on the chip than 128-wide ones, up to one per CU (workgroups = tiles x parts along K): the cold weight stream
// of a launch is paced by CUs x what one CU pulls (15-18 GB/s, R6.7 / R6.14), so more CUs asking IS more stream.  Cold, us per GEMM, 128-wide -> 96-wide:
// 12288 x 4096 at 160 / 192 / 224 / 256 rows (192 -> 256 workgroups = every CU) 24.7 / 25.4 / 26.3 / 27.5 -> 23.3 / 24.4 / 24.9 / 25.4 (profiles/r06_mid_bn96_all_cus.txt),
// 11008 x 4096 at 160 / 192 / 256 rows (172 -> 230) 25.1 / 25.0 / 25.5 -> 24.0 / 24.3 / 25.0, 4608 x 3584 at 512 rows (144 -> 192) 20.7 -> 19.8
// (profiles/r06_mid_bn96_cold.txt); with K split over workgroups (R6.23, profiles/r06_mid_bn96_xsplit.txt): 3584 x 8192 at 320 / 384 rows, two parts (168 -> 228 workgroups)
// 27.7 / 29.4 -> 26.6 / 27.3, 1280 x 8192 at 512 rows, four parts (160 -> 224) 22.5 -> 21.4, at 1024 rows, two parts 29.1 -> 27.2; warm: level.  (A first measurement of the 256-workgroup case -- 26.8 -> 29.6 -- had the rotated K walk switched on as well,
// which is what loses with two tile rows (R6.15); knob 1432 keeps the sixteenth of the CUs free that it led to, for A/B.)
int gemm_mid_tile_width(int M, int N, int xsplit)
{
    const int mode = g_mid_bn.load();
    const int tm = (M + mid::BM - 1) / mid::BM;
    const int t128 = tm * ((N + 127) / 128), t96 = tm * ((N + 95) / 96);
    const int xs = xsplit > 1 ? xsplit : 1;
    const bool fits96 = t96 * xs <= num_cus() && t96 > t128;
    if (mode == 1) return 128;
    if (mode == 3) return fits96 && xs == 1 ? 96 : 128;                          // (measurement: not with K split over workgroups)
    if (mode == 2) return fits96 && 16 * t96 * xs <= 15 * num_cus() ? 96 : 128;  // (measurement: a sixteenth of the CUs left free)
    return fits96 ? 96 : 128;
}

template <int EPI, int BN>
static hipError_t launch_mid_cfg(const GemmParams& q, hipStream_t st)
{
    using L = mid::Layout<BN>;
    const int tiles = ((q.M + mid::BM - 1) / mid::BM) * ((q.N + BN - 1) / BN);
    if (q.xsplit > 1) {
        auto kern = gemm_w8a8o16_mid_kernel<EPI, true, BN>;
        static DeviceOnce once;
        if (hipError_t e = ensure_dynamic_lds(kern, L::LDS, once); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * q.xsplit)), dim3(mid::T), L::LDS, st, q);
    } else {
        auto kern = gemm_w8a8o16_mid_kernel<EPI, false, BN>;
        static DeviceOnce once;
        if (hipError_t e = ensure_dynamic_lds(kern, L::LDS, once); e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(mid::T), L::LDS, st, q);
    }
    return hipGetLastError();
}

template <int EPI>
static hipError_t launch_mid_epi(const GemmParams& p, hipStream_t st)
{
    GemmParams q = p;
    // one tile row: every weight line is read by exactly one workgroup -> non-temporal copies for weights of 32 MiB and more
    // (the rule of gemm_kernels.hip's launch_cfg)
    if (p.M <= mid::BM && (int64_t)p.N * p.K >= ((int64_t)32 << 20)) q.flags |= 2;
    // Rotated K walk (every tile row starts at its own slice, p.flags bit 2; integer sums commute: same bits): the tile rows of a column panel
    // then miss on DIFFERENT weight lines at any moment instead of all waiting for the same ones.  Taken where it measured ahead on three of
    // three boxes, cold: 4..7 tile rows, tiles not split along K (4608 x 3584 at 512 / 768 rows 24.2 / 24.5 -> 20.6 / 22.7, 22.1 / 25.0 -> 19.8 / 23.7,
    // 22.1 / 25.5 -> 20.8 / 24.6 us; 4096 x 4096 at 768 rows 26.0 -> 22.9, 26.0 -> 24.3, 26.3 -> 25.1; warm: level) -- profiles/r06_mid_final_sweep_cold.txt,
    // r06_mid_bn96_cold.txt, r06_mid_rot_rows_cold.txt.  Not with two tile rows (12288 / 11008 x 4096 at 256 rows: ahead on one box, behind on two, 8 %
    // behind warm), not at 8 rows (+2 %), and not with K split over workgroups: bimodal there (4096 x 11008 at 256 rows, four parts: 29.5 us in one
    // process, 35 in the next -- it tripped the selection gate).  Knobs 1411 / 1410 / 1412: never / every unsplit launch / always.
    const int rotm = g_mid_rot.load();
    const int tile_rows = (p.M + mid::BM - 1) / mid::BM;
    if (rotm == 2 || (rotm == 1 && p.xsplit <= 1) || (rotm == 3 && p.xsplit <= 1 && tile_rows >= 4 && tile_rows <= 7)) q.flags |= 4;
    if (gemm_mid_tile_width(p.M, p.N, p.xsplit) == 96) return launch_mid_cfg<EPI, 96>(q, st);
    return launch_mid_cfg<EPI, 128>(q, st);
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

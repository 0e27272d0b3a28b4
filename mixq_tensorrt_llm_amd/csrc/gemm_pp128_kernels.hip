// Ping-pong fused W8A8O16 GEMM on 128 (m) x 256 (n) tiles -- the mid-size companion of gemm_pp_kernels.hip.
//
// Same math, same operand roles, same epilogue arithmetic, same results (reference lines replaced: see gemm_kernels.hip).
// Why a second tile shape: a 256 x 256 tile is the whole register file of a CU, so a problem with fewer than ~256 of them
// (short prefill, decode batches of 256..2048 rows) leaves CUs idle or has to split K over workgroups and pay an exchange
// of 256 KiB per tile.  Half-height tiles double the tile count with NO exchange, and the ping-pong schedule keeps its
// efficiency because a slice still gives every wave 16 MFMAs in two segments of 8:
//
//   * 8 waves = 2 (m) x 4 (n), wave tile 64 x 64 = 2 x 2 MFMA tiles (64 accumulator registers); two groups of 4 waves,
//     one wave of each on every SIMD, group 1 one segment behind group 0 (as in the 256 x 256 kernel).
//   * a K slice (128 B per row) = 2 phases, each {LOAD segment, s_barrier, COMPUTE segment of 8 MFMAs, s_barrier}:
//       phase 1: LOAD  Y fragments of this slice (8 ds_read_b128)  + copies X0(kt+2), X1(kt+1)   COMPUTE  X0 x (Y0, Y1)
//       phase 2: LOAD  X1 of this slice + X0 of the NEXT slice (8) + copy   Y(kt+2)              COMPUTE  X1 x (Y0, Y1)
//     X0 / X1 = the two 32-column halves of the wave's 64 columns, Y0 / Y1 its two 32-row tiles.
//   * LDS: two slice buffers of 48 KiB [X0 | X1 | Y], 16 KiB regions, 16-B slot = chunk ^ ((row >> 1) & 7); + eight 4-KiB
//     store windows = 128 KiB.
//
// Hazard bookkeeping (slot s = one LOAD segment; slice k = slots 2k, 2k+1; every segment boundary is a barrier of all
// 8 waves, group 1's LOAD(s) runs during group 0's COMPUTE(s)):
//   reads   Y(k) at slot 2k;  X1(k) and X0(k+1) at slot 2k+1.
//   copies  X0(k) at slot 2k-4, X1(k) at slot 2k-2, Y(k) at slot 2k-3   (i.e. slot 2j issues X0(j+2) and X1(j+1): 4 copy
//           instructions per thread; slot 2j+1 issues Y(j+2): 2).
//   RAW     a copy issued at slot s is read at slot s+3 at the earliest; every wave waits `vmcnt(6)` at the end of each
//           LOAD segment = everything but the copies of the last two slots (4 + 2) has landed, i.e. everything issued up
//           to slot s has landed by the end of LOAD(s+2), and a barrier lies before LOAD(s+3).
//   WAR     X0(j+2) overwrites X0(j), last read at slot 2j-1;  X1(j+1) overwrites X1(j-1), last read at slot 2j-1;
//           Y(j+2) overwrites Y(j), read at slot 2j: always an earlier slot, with barriers between.
//
// Measured (round 2, one box, tools/ab.sh): 2048 x 4096 x 4096 40.4 us = one wave of 256 tiles (the 256 x 256 form with K split
// 2 ways: 45.9-47.7), i.e. ~57 % MFMA occupancy inside the loop against 92 % for 256 x 256 tiles: a slice costs every wave 16
// fragment reads + 6 copy instructions for 16 MFMAs (256 x 256: 24 + 8 for 32), so the LOAD segments, not the matrix
// pipe, pace it.  Tried: three slice buffers with one copy per phase issued BEHIND the MFMAs of the COMPUTE segment, where
// the wave otherwise waits for its partner's LOAD segment: bit-identical, 2-6 % slower on four shapes (a copy next to
// MFMAs opens a bubble in the matrix pipe, as in the 256 x 256 kernel); not kept.
#include "mixq_device.h"
#include "mixq_launch.h"
#include <type_traits>

namespace mixq {

namespace pp128 {
constexpr int BM = 128, BN = 256, T = 512;
constexpr int KS = 128;                 // K bytes per row per slice
constexpr int REGION = 128 * KS;        // 16 KiB: 128 rows
constexpr int BUF = 3 * REGION;         // 48 KiB per slice buffer
constexpr int X0 = 0, X1 = REGION, YR = 2 * REGION;
constexpr int OSLICE = 256;
constexpr int GROUP_M = 8;              // tiles of 128 rows walked together (the 1024 rows of the 256 x 256 kernel's 4)

typedef float v2f __attribute__((ext_vector_type(2)));

#define MIXQ128_SEG_END()                               \
    do {                                                \
        __builtin_amdgcn_sched_barrier(0);              \
        asm volatile("s_barrier" ::: "memory");        \
        __builtin_amdgcn_sched_barrier(0);              \
    } while (0)

// (glds16_sbase: the asm-form 16-byte LDS-DMA with an SGPR base, mixq_device.h)
} // namespace pp128

template <int EPI, bool HAS_O, bool HAS_Y>
__global__ __launch_bounds__(512) void gemm_w8a8o16_pp128_kernel(const GemmParams p)
{
    using namespace pp128;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2;          // 0: waves 0-3, 1: waves 4-7 (one of each per SIMD)
    const int wm = wave & 1;              // 2 wave rows along m (64 each)
    const int wn = wave >> 1;             // 4 wave columns along n (64 each)
    const int lr = lane & 31, lh = lane >> 5;

    // ---- block -> tile mapping (XCD-aware, grouped) -----------------------------------------------------------------
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    int t_lin;
    {
        const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
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

    // ---- staging sources: regions X0, X1, Y; two 16-B copies per thread per region per slice -------------------------
    // LDS row q (0..127) of region X half h: n_local = (q/32)*64 + h*32 + q%32 (q/32 = wn);  region Y: m_local = q.
    const int64_t K = p.K;
    const char* const baseB = reinterpret_cast<const char*>(p.B) + (int64_t)n0 * K;
    const char* const baseA = reinterpret_cast<const char*>(p.A) + (int64_t)m0 * K;
    unsigned off[3][2];
    int koff_src;
    {
        const int slot = tid & 7;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = i * 64 + (tid >> 3);
            const int sw = (q >> 1) & 7;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int nl = (q >> 5) * 64 + h * 32 + (q & 31);
                const int rn = min(n0 + nl, p.N - 1) - n0; // clamped rows, >= 0
                off[h][i] = (unsigned)rn * (unsigned)p.K + ((slot ^ sw) << 4);
            }
            const int rm = min(m0 + q, p.M - 1) - m0;
            off[2][i] = (unsigned)rm * (unsigned)p.K + ((slot ^ sw) << 4);
        }
        koff_src = (slot ^ (((tid >> 3) >> 1) & 7)) << 4; // same for i = 0, 1 (64 rows apart)
    }
    const int nk = (p.K + KS - 1) / KS;
    const bool ktail = (p.K % KS) != 0;
    const unsigned lds0 = (unsigned)(size_t)(MIXQ_LDS_PTR(smem)) + wave * 1024; // this wave's 1-KiB DMA window

    // 2 x LDS-DMA: region `region` (0 = X0, 1 = X1, 2 = Y) of slice kt.  TAILCHK: the slice may be partial in K.
    auto issue = [&](int region, int kt, bool tailchk) __attribute__((always_inline)) {
        const unsigned dst = lds0 + (kt & 1) * BUF + region * REGION;
        const char* base = (region < 2 ? baseB : baseA) + (int64_t)kt * KS; // scalar
        if (tailchk && ktail) {
            const bool oob = (int64_t)kt * KS + koff_src >= K;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* s = oob ? static_cast<const char*>(p.zeros) : base + off[region][i];
                glds16(s, smem + (kt & 1) * BUF + region * REGION + (i * T + wave * 64) * 16);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) glds16_sbase(base, off[region][i], dst + i * T * 16);
    };

    // ---- fragment read offsets --------------------------------------------------------------------------------------
    const int sw = (lr >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = ((ks * 2 + lh) ^ sw) << 4;
    const int xrow = (wn * 32 + lr) * KS;   // + X0 / X1
    const int yrow = (wm * 64 + lr) * KS;   // + YR, + jy * 32 * KS

    v4i XA[4], XB[4], Y[2][4];
    v16i acc[2][2]; // [n tile][m tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;

    auto read_x = [&](v4i (&X)[4], int kt, int half) __attribute__((always_inline)) {
        const char* b = smem + (kt & 1) * BUF + (half ? X1 : X0) + xrow;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) X[ks] = *reinterpret_cast<const v4i*>(b + koff[ks]);
    };
    auto read_y = [&](int kt) __attribute__((always_inline)) {
        const char* b = smem + (kt & 1) * BUF + YR + yrow;
#pragma unroll
        for (int jy = 0; jy < 2; ++jy)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) Y[jy][ks] = *reinterpret_cast<const v4i*>(b + jy * 32 * KS + koff[ks]);
    };
    auto mma = [&](const v4i (&X)[4], int xi) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int jy = 0; jy < 2; ++jy)
                if constexpr (EPI == EPI_F16GEMM) // fp16 operands, fp32 sums kept as bit patterns (see gemm_pp_kernels.hip)
                    acc[xi][jy] = __builtin_bit_cast(
                        v16i, __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, X[ks]), __builtin_bit_cast(v8h, Y[jy][ks]),
                                                                     __builtin_bit_cast(v16f, acc[xi][jy]), 0, 0, 0));
                else
                    acc[xi][jy] = __builtin_amdgcn_mfma_i32_32x32x32_i8(X[ks], Y[jy][ks], acc[xi][jy], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // One K slice.  XA holds X0 of this slice on entry and X0 of the next slice on exit.
    auto slice = [&](int kt, auto steady_tag) __attribute__((always_inline)) {
        constexpr bool steady = decltype(steady_tag)::value; // compile-time: slices kt+1 and kt+2 exist and are full
        const bool more1 = steady || (kt + 1 < nk), more2 = steady || (kt + 2 < nk); // wave-uniform
        // phase 1 (slot 2kt): Y of this slice; copies X0(kt+2), X1(kt+1)
        read_y(kt);
        if (more2) issue(0, kt + 2, !steady);
        if (more1) issue(1, kt + 1, !steady);
        if (steady) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (the last slices: fewer copies are in flight; drain)
        MIXQ128_SEG_END();
        mma(XA, 0);
        MIXQ128_SEG_END();
        // phase 2 (slot 2kt+1): X1 of this slice, X0 of the next one; copy Y(kt+2)
        read_x(XB, kt, 1);
        if (more1) read_x(XA, kt + 1, 0);
        if (more2) issue(2, kt + 2, !steady);
        if (steady) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        MIXQ128_SEG_END();
        mma(XB, 1);
        MIXQ128_SEG_END();
    };
    using steady_t = std::true_type;
    using tail_t = std::false_type;

    // ---- prologue: X0(0), Y(0), X1(0), X0(1), Y(1) in the order the steady state would have issued them ------------------
    issue(0, 0, true);
    issue(2, 0, true);
    issue(1, 0, true);
    if (nk > 1) {
        issue(0, 1, true);
        issue(2, 1, true);
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); // X0(0), Y(0) have landed
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    MIXQ128_SEG_END();
    read_x(XA, 0, 0);
    if (group == 1) MIXQ128_SEG_END(); // group 1 now runs one segment behind group 0

    {
        int kt = 0;
        for (; kt + 3 < nk; ++kt) slice(kt, steady_t{}); // slices kt+1, kt+2 exist and kt+2 is not the (possibly partial) last
        for (; kt < nk; ++kt) slice(kt, tail_t{});
    }
    if (group == 0) MIXQ128_SEG_END(); // re-align the groups

    if (EPI == EPI_INT32) { // debug / unfused API: raw accumulators, 16-byte stores straight from the MFMA layout
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int m = m0 + wm * 64 + j * 32 + lr;
                const int nb0 = n0 + wn * 64 + i * 32 + 4 * lh;
                if (m < p.M) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int nb = nb0 + 8 * g;
                        if (nb < p.N) {
                            v4i o = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
                            *reinterpret_cast<v4i*>(static_cast<int32_t*>(p.D) + (int64_t)m * p.N + nb) = o;
                        }
                    }
                }
            }
        return;
    }

    // ---- outlier operands -> LDS (256-B rows, slot = chunk ^ (row & 15)); chunks past O come from the zero page ----
    if (HAS_O) {
        __syncthreads();
        constexpr int OXL = BN * 16 / T, OYL = BM * 16 / T;
        const int obytes = p.O * 2;
        const int slot = tid & 15;
#pragma unroll
        for (int i = 0; i < OXL; ++i) {
            const int row = (i * T + tid) >> 4;
            const int c = (slot ^ (row & 15)) << 4;
            const int grow = min(n0 + row, p.N - 1);
            const char* s = reinterpret_cast<const char*>(p.fpW) + (int64_t)grow * obytes + c;
            if (c >= obytes) s = static_cast<const char*>(p.zeros);
            glds16(s, smem + (i * T + wave * 64) * 16);
        }
#pragma unroll
        for (int i = 0; i < OYL; ++i) {
            const int row = (i * T + tid) >> 4;
            const int c = (slot ^ (row & 15)) << 4;
            const int grow = min(m0 + row, p.M - 1);
            const char* s = reinterpret_cast<const char*>(p.fpA) + (int64_t)grow * obytes + c;
            if (c >= obytes) s = static_cast<const char*>(p.zeros);
            glds16(s, smem + BN * OSLICE + (i * T + wave * 64) * 16);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- dequant math + stores, one 32 (m) x 64 (n) block of the wave tile at a time (same scheme as gemm_pp_kernels.hip:
    // results packed to fp16, transposed through a wave-private 4-KiB LDS window, written as 128-byte row segments) -----
    char* const wstg = smem + 2 * BUF + wave * 4096;
    const int obase = (lh ^ (lr & 15)) << 4; // 16-B slot of k-step ks = obase ^ (ks << 5)
    float sa[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) // (clamped rows are never stored)
        sa[j] = EPI == EPI_F16GEMM ? 1.f : h2f(p.sA[min(m0 + wm * 64 + j * 32 + lr, p.M - 1)]);

    auto side = [&](int i, int j) __attribute__((always_inline)) {
        v16f P;
#pragma unroll
        for (int e = 0; e < 16; ++e) P[e] = 0.f;
        if (HAS_O) {
            const char* xo = smem + (wn * 64 + i * 32 + lr) * OSLICE;
            const char* yo = smem + BN * OSLICE + (wm * 64 + j * 32 + lr) * OSLICE;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) {
                v8h xf[4], yf[4];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    xf[ks] = *reinterpret_cast<const v8h*>(xo + (obase ^ ((kh * 4 + ks) << 5)));
                    yf[ks] = *reinterpret_cast<const v8h*>(yo + (obase ^ ((kh * 4 + ks) << 5)));
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) P = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf[ks], yf[ks], P, 0, 0, 0);
            }
        }
        return P;
    };
    const int wrow = lr * 128 + ((lh ^ ((lr >> 3) & 1)) << 3); // this lane's row + 8-B half inside the window
    uint2 swq[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            swq[i][g] = EPI == EPI_F16GEMM ? make_uint2(0u, 0u)
                                           : *reinterpret_cast<const uint2*>(p.sW + min(n0 + wn * 64 + i * 32 + 4 * lh + 8 * g, p.N - 4));
    constexpr bool HAS_MUL = EPI == EPI_DEQUANT_SILU_MUL;
    uint2 yq[2][4], mq[2][4];
    uint4 ypre[4], mpre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ypre[q] = mpre[q] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) yq[i][g] = mq[i][g] = make_uint2(0u, 0u);
    const int64_t src_wave = ((int64_t)(m0 + wm * 64) * p.N + n0 + wn * 64) * 2; // byte offset of the wave tile
    const unsigned src_lane = ((unsigned)(lane >> 3) * (unsigned)p.N + (lane & 7) * 8) * 2;
    const bool src_n_ok = n0 + wn * 64 + (lane & 7) * 8 < p.N;
    const int win_rd = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);
    auto fetch = [&](const uint16_t* src, int j, uint4 (&v)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = j * 32 + q * 8; // wave-uniform
            const char* a = reinterpret_cast<const char*>(src) + src_wave + (int64_t)row * p.N * 2 + src_lane;
            const bool ok = src_n_ok && m0 + wm * 64 + row + (lane >> 3) < p.M;
            v[q] = ok ? *reinterpret_cast<const uint4*>(a) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto spread = [&](const uint4 (&v)[4], uint2 (&out)[2][4]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<uint4*>(wstg + q * 1024 + win_rd) = v[q];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                out[i][g] = *reinterpret_cast<const uint2*>(wstg + lr * 128 + (((i * 4 + g) ^ (lr & 7)) << 4) + lh * 8);
    };
    if (HAS_Y) fetch(p.Y, 0, ypre);
    if (HAS_MUL) fetch(p.Mul, 0, mpre);
    auto dequant = [&](int i, int j, const v16f& P) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint2 swb = swq[i][g];
            const float swf[4] = {h2f((uint16_t)(swb.x & 0xffffu)), h2f((uint16_t)(swb.x >> 16)),
                                  h2f((uint16_t)(swb.y & 0xffffu)), h2f((uint16_t)(swb.y >> 16))};
            uint16_t yh[4] = {0, 0, 0, 0};
            if (HAS_Y) {
                const uint2 yb = yq[i][g];
                yh[0] = (uint16_t)(yb.x & 0xffffu), yh[1] = (uint16_t)(yb.x >> 16);
                yh[2] = (uint16_t)(yb.y & 0xffffu), yh[3] = (uint16_t)(yb.y >> 16);
            }
            const uint2 mulq = mq[i][g];
            unsigned ow[2];
#pragma unroll
            for (int e2 = 0; e2 < 4; e2 += 2) {
                const v2f s2 = v2f{swf[e2], swf[e2 + 1]} * sa[j]; // exact: fp16 x fp16 products
                v2f c2;
                if (HAS_O) {
                    const v2h p16 = f2h2(P[4 * g + e2], P[4 * g + e2 + 1]); // MFMA outputs: nothing to fuse with
                    c2 = v2f{(float)p16[0], (float)p16[1]};
                } else {
                    c2 = v2f{h2f(yh[e2]), h2f(yh[e2 + 1])};
                }
                float v0, v1;
                if constexpr (EPI == EPI_F16GEMM) { // (named ints: __builtin_bit_cast on a vector ELEMENT expression reads element 0)
                    const int b0 = acc[i][j][4 * g + e2], b1 = acc[i][j][4 * g + e2 + 1];
                    v0 = __builtin_bit_cast(float, b0);
                    v1 = __builtin_bit_cast(float, b1);
                } else {
                    v0 = __builtin_fmaf((float)acc[i][j][4 * g + e2], s2[0], c2[0]);
                    v1 = __builtin_fmaf((float)acc[i][j][4 * g + e2 + 1], s2[1], c2[1]);
                }
                if (epi_has_silu(EPI)) {
                    v0 = silu_f32(v0);
                    v1 = silu_f32(v1);
                }
                v2h o16 = f2h2_of_f32_results(v0, v1);
                if (EPI == EPI_DEQUANT_SILU_MUL) o16 = o16 * __builtin_bit_cast(v2h, e2 ? mulq.y : mulq.x); // gate * up
                __builtin_memcpy(&ow[e2 >> 1], &o16, 4);
            }
            const int c = i * 4 + g; // 16-byte chunk of the 128-byte row
            *reinterpret_cast<uint2*>(wstg + wrow + ((c ^ (lr & 7)) << 4)) = uint2{ow[0], ow[1]};
        }
    };
    char* const dwave = static_cast<char*>(p.D) + ((int64_t)(m0 + wm * 64) * p.N + n0 + wn * 64) * 2;
    const unsigned dlane = ((unsigned)(lane >> 3) * (unsigned)p.N + (lane & 7) * 8) * 2;
    const int rd = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4); // window read offset (rr & 7 == lane >> 3)
    const bool n_ok = n0 + wn * 64 + (lane & 7) * 8 < p.N;
    const bool interior = m0 + BM <= p.M && n0 + BN <= p.N; // wave-uniform: no store predicates needed
    auto flush = [&](int j) __attribute__((always_inline)) {
        uint4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const uint4*>(wstg + q * 1024 + rd);
#pragma unroll
        for (int q = 0; q < 4; ++q) // (opaque: keeps the four reads together, ahead of the predicated stores)
            asm volatile("" : "+v"(v[q].x), "+v"(v[q].y), "+v"(v[q].z), "+v"(v[q].w));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q & 1) v[q] = uint4{v[q].z, v[q].w, v[q].x, v[q].y}; // rows with bit 3 set hold their 8-B halves swapped
            const int row = j * 32 + q * 8;     // wave-uniform
            char* dst = dwave + (int64_t)row * p.N * 2 + dlane;
            const v4i vv = {(int)v[q].x, (int)v[q].y, (int)v[q].z, (int)v[q].w}; // (non-temporal: see gemm_pp_kernels.hip)
            if (interior) __builtin_nontemporal_store(vv, reinterpret_cast<v4i*>(dst));
            else if (n_ok && m0 + wm * 64 + row + (lane >> 3) < p.M) __builtin_nontemporal_store(vv, reinterpret_cast<v4i*>(dst));
        }
    };
    {
        v16f Pcur = side(0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            v16f Pnext = Pcur;
            if (t + 1 < 4) Pnext = side((t + 1) & 1, (t + 1) >> 1);
            if ((HAS_Y || HAS_MUL) && (t & 1) == 0) { // block j = t >> 1 starts: operands of this block -> registers
                if (HAS_Y) spread(ypre, yq);
                if (HAS_MUL) spread(mpre, mq);
                if (t + 2 < 4) {
                    if (HAS_Y) fetch(p.Y, (t >> 1) + 1, ypre);
                    if (HAS_MUL) fetch(p.Mul, (t >> 1) + 1, mpre);
                }
            }
            dequant(t & 1, t >> 1, Pcur);
            if (t & 1) flush(t >> 1);
            Pcur = Pnext;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int EPI, bool HAS_O, bool HAS_Y>
static hipError_t launch_pp128_cfg(const GemmParams& p, hipStream_t st)
{
    constexpr size_t lds = 2 * (size_t)pp128::BUF + 32768; // slice buffers + 8 x 4-KiB store windows = 128 KiB
    auto kern = gemm_w8a8o16_pp128_kernel<EPI, HAS_O, HAS_Y>;
    static DeviceOnce once;
    if (hipError_t e = ensure_dynamic_lds(kern, lds, once); e != hipSuccess) return e;
    const int tiles = ((p.M + pp128::BM - 1) / pp128::BM) * ((p.N + pp128::BN - 1) / pp128::BN);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(pp128::T), lds, st, p);
    return hipGetLastError();
}

template <int EPI>
static hipError_t launch_pp128_epi(const GemmParams& p, hipStream_t st)
{
    if (p.O > 0) return launch_pp128_cfg<EPI, true, false>(p, st); // the API never passes both an addend and outliers
    if (p.Y != nullptr) return launch_pp128_cfg<EPI, false, true>(p, st);
    return launch_pp128_cfg<EPI, false, false>(p, st);
}

hipError_t launch_gemm_f16_pp128(const void* A, const void* B, void* D, int M, int N, int K, const void* zeros, hipStream_t st)
{
    if (K % 8 || N % 8) return hipErrorInvalidValue;
    if (M <= 0 || N <= 0) return hipSuccess;
    GemmParams p{};
    p.A = static_cast<const int8_t*>(A), p.B = static_cast<const int8_t*>(B), p.D = D, p.zeros = zeros;
    p.M = M, p.N = N, p.K = 2 * K; // the kernel counts K in bytes
    return launch_pp128_cfg<EPI_F16GEMM, false, false>(p, st);
}

hipError_t launch_gemm_pp128(const GemmParams& p, int epi, hipStream_t st)
{
    switch (epi) {
    case EPI_DEQUANT: return launch_pp128_epi<EPI_DEQUANT>(p, st);
    case EPI_DEQUANT_SILU: return launch_pp128_epi<EPI_DEQUANT_SILU>(p, st);
    case EPI_DEQUANT_SILU_MUL: return launch_pp128_epi<EPI_DEQUANT_SILU_MUL>(p, st);
    default: return launch_pp128_cfg<EPI_INT32, false, false>(p, st);
    }
}

} // namespace mixq

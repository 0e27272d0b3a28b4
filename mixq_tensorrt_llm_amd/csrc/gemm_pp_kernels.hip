// Ping-pong ("8-phase") variant of the fused W8A8O16 GEMM for large problems on gfx950.
//
// Same math, same operand roles, same results as gemm_kernels.hip (see its header for the reference lines replaced);
// what changes is the main-loop schedule, built for one 512-thread workgroup per CU (256 x 256 output tile, 160 KiB LDS):
//
//   * the 8 waves form two groups of 4 (one wave of each group on every SIMD).  A K slice (128 B per row) is processed
//     in 4 phases; every phase is a LOAD segment (ds_read_b128 of the next fragments + 2 global_load_lds of the next
//     slice) followed by a COMPUTE segment (8 x v_mfma_i32_32x32x32_i8), separated by s_barrier.  Group 1 runs one
//     segment behind group 0, so on each SIMD one wave is always in its MFMA segment while its partner fetches:
//     the matrix pipe does not wait for LDS or HBM latency.
//   * global -> LDS copies are never drained inside the loop: each wave waits `vmcnt(4)` at the end of a LOAD segment,
//     i.e. only for copies issued two segments earlier; the two most recent pairs stay in flight across barriers.
//     The copies are issued from inline asm in the SGPR-base form (`global_load_lds_dwordx4 voff, s[base]`): the
//     per-thread 32-bit offsets never change and the base advances by 128 B per slice on the scalar unit, so a LOAD
//     segment carries no address VALU and no compiler-inserted waits.
//   * wave tile = 128 (m) x 64 (n) = 4 x 2 MFMA tiles.  Phase order (m-half, n-half): (0,0) (0,1) (1,1) (1,0), so each
//     LOAD segment fetches 8 or 4 fragments: Y0 | X1 | Y1 | X0-of-the-next-slice.  Two slices are unrolled per loop
//     iteration so the two X fragment sets swap roles with static register names.  The loop body is branch-free: the
//     last two slices (next slice may be partial in K / no next slice) are peeled into their own instantiations.
//   * LDS image per buffer: [X half 0 | X half 1 | Y half 0 | Y half 1], 16 KiB each, rows ordered
//     [half][wave][row] so that every region is one contiguous run of 128-byte rows; 16-B slot = chunk ^ ((row>>1)&7).
//
// Hazard bookkeeping (slots are per-wave LOAD segments; group 1 lags by one segment):
//   RAW  a region issued at slot s is first read at slot s+3; every wave has passed `vmcnt(4)` at the end of slot
//        s+2 (which retires everything issued up to slot s) and a barrier lies between.
//   WAR  a region is re-issued 5 slots after its last read.
//
// Epilogue: fp16 outlier side GEMM (operands staged in the dead main-loop LDS, 8 unrolled k-steps per 32x32 tile),
// dequant FMA, results packed to fp16 and transposed through a wave-private LDS window into 128-byte row segments.
#include "mixq_device.h"
#include "mixq_launch.h"
#include <atomic>
#include <type_traits>

namespace mixq {

namespace pp {
constexpr int BM = 256, BN = 256, T = 512;
constexpr int KS = 128;                 // K bytes per row per slice
constexpr int REGION = 128 * KS;        // 16 KiB: 128 rows
constexpr int BUF = 4 * REGION;         // 64 KiB per slice buffer
constexpr int X0 = 0, X1 = REGION, Y0 = 2 * REGION, Y1 = 3 * REGION;
constexpr int OSLICE = 256;

enum { STEADY = 0, PENULT = 1, LAST = 2 };

typedef float v2f __attribute__((ext_vector_type(2)));

#define MIXQ_SEG_END()                                  \
    do {                                                \
        __builtin_amdgcn_sched_barrier(0);              \
        asm volatile("s_barrier" ::: "memory");        \
        __builtin_amdgcn_sched_barrier(0);              \
    } while (0)

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
}

// (glds16_sbase: the asm-form 16-byte LDS-DMA with an SGPR base, mixq_device.h)
} // namespace pp

// ABL: measurement-only ablations (wrong results): 32 = MFMA A / B operands swapped, 1 = no global_load_lds in the loop, 2 = no vmcnt waits,
// 4 = no ds_reads in the loop, 8 = every tile loads tile (0,0)'s operands (all L2 hits), 16 = no epilogue.
// ABL = 0 is the product kernel.
// SPLITK = 2 / 4 / 8: K split over S workgroups per output tile, for mid-size problems whose tiles alone cover at most
// 1/S of the CUs.  The S workgroups of a tile are S consecutive blocks (dispatched together, resident together); each
//   1. multiplies its 1/S of the K slices into the usual 256x256 int32 accumulators, with the 32-row m tiles of its
//      wave tile PERMUTED (accumulator tile jj holds m tile jj ^ (r * PJ), r = rank in the group, PJ = 4 / S; with
//      S = 8 the two 32-column n tiles are permuted the same way by rank bit 2: a change of the staging offsets
//      only), so that the share it will finish always sits in acc[0 .. NI-1][0 .. PJ-1];
//   2. parks the other S-1 shares in p.splitk_ws (its own slot), waits for the write-through acknowledgements,
//      draws an arrival ticket, and stages its outlier operands while the others do the same;
//   3. once all S tickets are drawn, adds the S-1 foreign partial sums of its own share (64 / 96 / 112 loads per
//      lane, all in flight together, landing in the registers step 2 freed) and runs the usual epilogue on that share
//      (S = 8: one 32x32 tile per wave, stored with 8-byte stores straight from the accumulator layout).
// With more tiles than CUs, the first p.splitk_solo tiles (whole waves) are computed by one workgroup each inside the
// same launch ("solo" blocks: full K range, no exchange) and only the tiles of the last, partial wave are split.
// The hand-over words of every shape live in the first kSplitkWordsBytes of the scratch, the slots behind them, so one
// scratch per stream serves launches of any shapes.
// Same int32 sums (integer addition commutes), same epilogue, same bits as the one-workgroup form.
// Hand-over accesses are relaxed AGENT-scope atomics: `global_store sc1` writes through to memory and
// `global_load sc1` is served coherently, per access, for any pair of XCDs -- no bulk L2 write-back / invalidate (a
// release / acquire fence pair costs ~70k cycles per tile here), and the compiler tracks the loads (an asm load
// consumed after a later s_waitcnt gets copied / spilled before its data has arrived).
//
// NOBODY WAITS WITHOUT BOUND (round 2; round 1 span on the partners' arrival words and trapped after 2^23 polls, which
// two concurrent split launches on different streams could turn into a circular wait and a dead context).  Words of a
// tile: [0] tickets drawn, [1 + r] state of rank r's share (0 | DEFERRED | SEEN), [9] workgroups done.
//   * The workgroup that draws the LAST ticket never waits: every partner has parked and acknowledged already.
//   * Any other workgroup polls the ticket count for at most kSplitkPatience (wall clock); if the partners are not all
//     there by then -- they may not even be resident: HIP promises nothing about dispatch order -- it also parks its OWN
//     share, marks it DEFERRED (compare-and-swap against the last arriver's SEEN mark: exactly one of the two
//     happens) and exits, freeing its CU.
//   * The last arriver, after finishing its own share, finishes every DEFERRED share too: all S partial sums of such a
//     share are in the scratch by then.  Every share is finished exactly once, by its owner or by the last arriver.
// The last workgroup of a group to be done with the words re-arms them for the next launch.
// TP = true (EPI_DEQUANT, no K split): the store path writes every finished block into this rank's column block of ALL
// p.tp.ndst destination buffers (row stride p.tp.ldd) with system-scope write-through stores instead of into p.D, and the
// last tile of an M chunk to be acknowledged publishes p.tp.seq in the chunk's flag word of every destination (TpEpilogue,
// mixq_launch.h; protocol in tp_kernels.hip).
template <int EPI, bool HAS_O, bool HAS_Y, int ABL = 0, int SPLITK = 0, bool TP = false>
__global__ __launch_bounds__(512) void gemm_w8a8o16_pp_kernel(const GemmParams p)
{
    static_assert(!TP || (SPLITK == 0 && EPI == EPI_DEQUANT && ABL == 0), "the peer-write epilogue exists for the plain kernel");
    constexpr int S = SPLITK ? SPLITK : 1; // workgroups per tile
    constexpr int PJ = S >= 4 ? 1 : 4 / S; // 32-row m tiles (per wave) this workgroup finishes
    constexpr int NI = S == 8 ? 1 : 2;     // 32-column n tiles (per wave) it finishes: S = 8 splits the n pair as well
    constexpr int NT = NI * PJ;            // 32x32 accumulator tiles it finishes
    static_assert(SPLITK == 0 || SPLITK == 2 || SPLITK == 4 || SPLITK == 8, "");
    using namespace pp;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2;          // 0: waves 0-3, 1: waves 4-7 (one of each per SIMD)
    const int wm = wave & 1;              // 2 wave rows along m (128 each)
    const int wn = wave >> 1;             // 4 wave columns along n (64 each)
    const int lr = lane & 31, lh = lane >> 5;

    // ---- block -> tile mapping (XCD-aware, grouped; identical to gemm_kernels.hip) -------------------
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    int t_lin, rank = 0;
    // SPLITK kernels: the first p.splitk_solo tiles (whole waves of tiles, a multiple of 8) are computed by one workgroup
    // each, exactly like the plain kernel; only the tiles of the last, partial wave are split S ways
    const int n_solo = SPLITK ? p.splitk_solo : 0;
    const bool solo = SPLITK && (int)blockIdx.x < n_solo;
    if (SPLITK && solo) {
        const int bid = blockIdx.x;
        t_lin = (bid & 7) * (n_solo >> 3) + (bid >> 3);
    } else if (SPLITK) { // block = idx * 8 + grp * S + rank: the S workgroups of a tile are consecutive blocks; each of
                         // the 8 / S groups of XCDs takes a contiguous range of tiles (neighbours share operand rows in L2)
        constexpr int G = 8 / S;
        const int bid = blockIdx.x - n_solo, xcd = bid & 7, grp = xcd / S, idx = bid >> 3;
        rank = xcd % S;
        const int ntail = nwg - n_solo, q = ntail / G, rem = ntail % G;
        if (idx >= q + (grp < rem ? 1 : 0)) return; // (all S blocks of the group leave together)
        t_lin = n_solo + (grp < rem ? grp * (q + 1) : rem * (q + 1) + (grp - rem) * q) + idx;
    } else {
        const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        t_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int jperm = (rank * PJ) & 3;                             // accumulator m tile jj <-> m tile jj ^ jperm
    const int iperm = S == 8 ? rank >> 2 : 0;                      // accumulator n tile ii <-> n tile ii ^ iperm
    // the epilogue runs on the share of rank `erank`: this workgroup's own, or -- last arriver only -- a deferred one
    int ejperm = jperm, eiperm = iperm;
    auto imap = [&](int ii) __attribute__((always_inline)) { return S == 8 ? (ii ^ eiperm) : ii; };
    const int nt = SPLITK && !solo ? NT : 8;                       // 32x32 accumulator tiles this workgroup finishes
    auto jmap = [&](int jj) __attribute__((always_inline)) { return SPLITK ? (jj ^ ejperm) : jj; };
    int tile_m, tile_n;
    {
        constexpr int GROUP_M = 4;
        const int per_group = GROUP_M * tiles_n;
        const int g = t_lin / per_group, first_m = g * GROUP_M;
        const int gsz = min(tiles_m - first_m, GROUP_M);
        const int within = t_lin - g * per_group;
        tile_m = first_m + within % gsz;
        tile_n = within / gsz;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // ---- staging sources: region r in {X0,X1,Y0,Y1}, two 16-B copies per thread per region per slice --------
    // LDS row q (0..127) of region (side, half h):  X: n_local = (q/32)*64 + h*32 + q%32   (q/32 = wn)
    //                                               Y: m_local = (q/64)*128 + h*64 + q%64  (q/64 = wm)
    // Address = wave-uniform 64-bit base (tile origin + slice offset, SGPRs) + constant per-thread 32-bit offset.
    const int64_t K = p.K;
    const char* const baseB = reinterpret_cast<const char*>(p.B) + ((ABL & 8) ? 0 : (int64_t)n0 * K);
    // A (qA) layout: row-major [M][K], or -- p.a_frag == 2, written by the quantiser for this kernel -- K-SLICE-MAJOR
    // [K / 128][M][128 B]: the 8 consecutive rows one LDS-DMA instruction copies are then 1 KiB contiguous instead of 8
    // segments K bytes apart (the same lever as the skinny kernel's fragment-major image)
    const bool a_slices = (ABL & 512) != 0 || p.a_frag == 2;
    const char* const baseA = reinterpret_cast<const char*>(p.A) +
                              ((ABL & 8) ? 0 : a_slices ? (int64_t)m0 * KS : (int64_t)m0 * K);
    const int64_t a_slice_stride = a_slices ? (int64_t)p.M * KS : (int64_t)KS; // bytes from one K slice of A to the next
    unsigned off[4][2];
    int koff_src;
    {
        const int slot = tid & 7;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = i * 64 + (tid >> 3);
                const int sw = (q >> 1) & 7;
                const int nl = (q >> 5) * 64 + (S == 8 ? h ^ iperm : h) * 32 + (q & 31);
                const int ml = SPLITK ? (q >> 6) * 128 + (h ^ (jperm >> 1)) * 64 + ((q & 63) ^ ((jperm & 1) << 5))
                                      : (q >> 6) * 128 + h * 64 + (q & 63);
                const int rn = min(n0 + nl, p.N - 1) - n0, rm = min(m0 + ml, p.M - 1) - m0; // clamped rows, >= 0
                off[h][i] = (unsigned)rn * (unsigned)p.K + ((slot ^ sw) << 4);
                off[2 + h][i] = (unsigned)rm * (a_slices ? (unsigned)KS : (unsigned)p.K) + ((slot ^ sw) << 4);
            }
        koff_src = (slot ^ (((tid >> 3) >> 1) & 7)) << 4; // same for i = 0,1 (64 rows apart)
    }
    const int nk = (p.K + KS - 1) / KS;
    const bool ktail = (p.K % KS) != 0;
    // K range of this workgroup in slices
    const int k_begin = SPLITK && rank > 0 ? ((nk * rank / S) + 1) & ~1 : 0;
    const int k_end = SPLITK && !solo && rank + 1 < S ? ((nk * (rank + 1) / S) + 1) & ~1 : nk;
    const unsigned lds0 = (unsigned)(size_t)(MIXQ_LDS_PTR(smem)) + wave * 1024; // this wave's 1-KiB DMA window

    // 2 x LDS-DMA: region `region` of slice kt.  TAILCHK: slice kt may be partial in K (chunks past K <- zero page).
    auto issue = [&](int region, int kt, bool tailchk) __attribute__((always_inline)) {
        if (ABL & 1) return;
        const unsigned dst = lds0 + (kt & 1) * BUF + region * REGION;
        const char* base = region < 2 ? baseB + (int64_t)kt * KS : baseA + (int64_t)kt * a_slice_stride; // scalar
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

    // ---- fragment read offsets ---------------------------------------------------------------------------
    const int sw = (lr >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = ((ks * 2 + lh) ^ sw) << 4;
    const int xrow = (wn * 32 + lr) * KS;                 // + X0 / X1
    const int yrow = (wm * 64 + lr) * KS;                 // + Y0 / Y1, + jy*32*KS

    v4i XA[4], XB[4], Y[2][4];
    v16i acc[2][4]; // [n tile][m tile]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;

    auto read_x = [&](v4i (&X)[4], int kt, int half) __attribute__((always_inline)) {
        if ((ABL & 4) && kt > 0) return;
        const char* b = smem + (kt & 1) * BUF + (half ? X1 : X0) + xrow;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) X[ks] = *reinterpret_cast<const v4i*>(b + koff[ks]);
    };
    auto read_y = [&](int kt, int half) __attribute__((always_inline)) {
        if ((ABL & 4) && kt > 0) return;
        const char* b = smem + (kt & 1) * BUF + (half ? Y1 : Y0) + yrow;
#pragma unroll
        for (int jy = 0; jy < 2; ++jy)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) Y[jy][ks] = *reinterpret_cast<const v4i*>(b + jy * 32 * KS + koff[ks]);
    };
    auto mma = [&](const v4i (&X)[4], int xi, int yhalf) __attribute__((always_inline)) {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int jy = 0; jy < 2; ++jy)
                if constexpr (EPI == EPI_F16GEMM) // same 16 bytes per lane per step, read as 8 fp16 of a 16-k step; the
                                                  // accumulator registers hold fp32 bit patterns
                    acc[xi][yhalf * 2 + jy] = __builtin_bit_cast(
                        v16i, __builtin_amdgcn_mfma_f32_32x32x16_f16(
                                  __builtin_bit_cast(v8h, X[ks]), __builtin_bit_cast(v8h, Y[jy][ks]),
                                  __builtin_bit_cast(v16f, acc[xi][yhalf * 2 + jy]), 0, 0, 0));
                else if constexpr ((ABL & 64) != 0) { // timing probe (wrong results): the same operand registers and the same
                    // number of integer operations through TWO v_mfma_i32_16x16x64_i8 per 32x32x32 (power per operation)
                    v16i& c = acc[xi][yhalf * 2 + jy];
                    v4i c0 = {c[0], c[1], c[2], c[3]}, c1 = {c[4], c[5], c[6], c[7]};
                    c0 = __builtin_amdgcn_mfma_i32_16x16x64_i8(X[ks], Y[jy][ks], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_i32_16x16x64_i8(X[ks], Y[jy][ks], c1, 0, 0, 0);
                    c[0] = c0[0], c[1] = c0[1], c[2] = c0[2], c[3] = c0[3];
                    c[4] = c1[0], c[5] = c1[1], c[6] = c1[2], c[7] = c1[3];
                } else
                acc[xi][yhalf * 2 + jy] =
                    (ABL & 32) ? __builtin_amdgcn_mfma_i32_32x32x32_i8(Y[jy][ks], X[ks], acc[xi][yhalf * 2 + jy], 0, 0, 0)
                               : __builtin_amdgcn_mfma_i32_32x32x32_i8(X[ks], Y[jy][ks], acc[xi][yhalf * 2 + jy], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // One K slice.  Xcur holds X half 0 of this slice on entry; Xoth receives X half 1, then X half 0 of slice kt+1.
    auto slice = [&](v4i (&Xcur)[4], v4i (&Xoth)[4], int kt, auto steady_tag) __attribute__((always_inline)) {
        constexpr bool steady = decltype(steady_tag)::value;   // compile-time: a full next slice exists, no checks
        const bool more = steady || (kt + 1 < k_end);          // wave-uniform
        // phase 1: (Y0, X0)
        read_y(kt, 0);
        if (more) issue(0, kt + 1, !steady);
        if (!(ABL & 2)) { if (more) wait_vmcnt<4>(); else wait_vmcnt<2>(); }
        MIXQ_SEG_END();
        mma(Xcur, 0, 0);
        MIXQ_SEG_END();
        // phase 2: (Y0, X1)
        read_x(Xoth, kt, 1);
        if (more) issue(2, kt + 1, !steady);
        if (!(ABL & 2)) { if (more) wait_vmcnt<4>(); else wait_vmcnt<0>(); }
        MIXQ_SEG_END();
        mma(Xoth, 1, 0);
        MIXQ_SEG_END();
        // phase 3: (Y1, X1)
        read_y(kt, 1);
        if (more) issue(1, kt + 1, !steady);
        if (more && !(ABL & 2)) wait_vmcnt<4>();
        MIXQ_SEG_END();
        mma(Xoth, 1, 1);
        MIXQ_SEG_END();
        // phase 4: (Y1, X0); the LOAD segment already fetches X half 0 of the next slice into the free set
        if (more) read_x(Xoth, kt + 1, 0);
        if (more) issue(3, kt + 1, !steady);
        if (more && !(ABL & 2)) wait_vmcnt<4>();
        MIXQ_SEG_END();
        mma(Xcur, 0, 1);
        MIXQ_SEG_END();
    };
    using steady_t = std::true_type;
    using tail_t = std::false_type;

    // optional timeline stamps (p.dbg != nullptr): wave 0 / lane 0 of every block writes s_memtime at 8 points
    auto stamp = [&](int idx) __attribute__((always_inline)) {
        if (p.dbg != nullptr && tid == 0)
            static_cast<unsigned long long*>(p.dbg)[(size_t)blockIdx.x * 8 + idx] = __builtin_readcyclecounter();
    };
    stamp(0);
    // SPLITK bookkeeping words of this tile (16 per tile): [0..7] arrival, [8] workgroups done reading.  They live in the FIRST
    // kSplitkWordsBytes of the scratch whatever the shape, so that one scratch can serve launches of different shapes:
    // every launch leaves them zero, and no launch ever parks data there.
    constexpr int SLOT = S * NI * PJ * 16 * T; // dwords per (tile, rank) slot: share 0 = own (deferral only), 1.. = foreign
    constexpr unsigned DEFERRED = 1u, SEEN = 2u;
    const int t_split = t_lin - n_solo; // index among the split tiles
    unsigned* const words = SPLITK ? static_cast<unsigned*>(p.splitk_ws) + t_split * 16 : nullptr; // [0] tickets, [1+r] state, [9] done
    // workgroup-wide broadcast of lane 0's decisions: two words at the start of the store windows (the 160 KiB of LDS are
    // all spoken for; the windows are idle until the epilogue, and every broadcast happens before it)
    volatile unsigned* const sk_bcast = reinterpret_cast<volatile unsigned*>(smem + 2 * BUF);
    int* const ws = reinterpret_cast<int*>(static_cast<char*>(p.splitk_ws) + kSplitkWordsBytes);
    // ---- prologue: slice 0 completely, then stagger the groups ----------------------------------------------
    issue(0, k_begin, true);
    issue(2, k_begin, true);
    issue(1, k_begin, true);
    issue(3, k_begin, true);
    wait_vmcnt<4>(); // X0 and Y0 have landed; X1 / Y1 are retired by the vmcnt(4) of phases 1 / 2 like in steady state
    MIXQ_SEG_END();
    read_x(XA, k_begin, 0);
    stamp(1);
    if (group == 1) MIXQ_SEG_END(); // group 1 now runs one segment behind group 0

    {
        int kt = k_begin;
        for (; kt + 3 < k_end; kt += 2) { // both slices of the pair have a full successor: branch-free body
            slice(XA, XB, kt, steady_t{});
            slice(XB, XA, kt + 1, steady_t{});
        }
        for (; kt < k_end; kt += 2) { // last 1-3 slices: the next slice may be partial in K or absent (runtime checks)
            slice(XA, XB, kt, tail_t{});
            if (kt + 1 < k_end) slice(XB, XA, kt + 1, tail_t{});
        }
    }
    if (group == 0) MIXQ_SEG_END(); // re-align the groups
    stamp(2);

    // ---- SPLITK: park the shares the other workgroups of the group finish, publish the arrival word
    // slot of (tile, rank): [(S-1) shares][NI n tiles][PJ m tiles][16][512 lanes] dwords (one instruction = a 2-KiB run)
    // share sh = accumulator tiles (n tile i0(sh) + i, m tile j0(sh) + x): S <= 4: both n tiles of m tiles sh * PJ ..;
    // S = 8: the single tile (sh >> 2, sh & 3).  Share sh of rank r holds the tiles that rank r ^ sh finishes.
    auto share_i0 = [](int sh) { return S == 8 ? sh >> 2 : 0; };
    auto share_j0 = [](int sh) { return S == 8 ? sh & 3 : sh * PJ; };
    bool am_last = false;
    if (SPLITK && !solo) {
        int* const mine = ws + ((size_t)t_split * S + rank) * SLOT + tid;
#pragma unroll
        for (int sh = 1; sh < S; ++sh)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int x = 0; x < PJ; ++x)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        __hip_atomic_store(mine + (((sh * NI + i) * PJ + x) * 16 + e) * T,
                                           acc[share_i0(sh) + i][share_j0(sh) + x][e], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every write-through acknowledged
        __syncthreads();
        if (tid == 0)
            sk_bcast[0] = __hip_atomic_fetch_add(words, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // my ticket
        __syncthreads();
        am_last = sk_bcast[0] == (unsigned)(S - 1);
        stamp(3);
    }

    if constexpr ((ABL & 128) != 0) { // probe: ~4k idle cycles per tile (every wave sleeps) -- does the launch grow by the idle time
        __builtin_amdgcn_s_sleep(63); // (time-bound) or by less (energy-bound at the power cap: the clock rises to compensate)?
    }
    if constexpr ((ABL & 256) != 0) {
        __builtin_amdgcn_s_sleep(63);
        __builtin_amdgcn_s_sleep(63);
        __builtin_amdgcn_s_sleep(63);
        __builtin_amdgcn_s_sleep(63);
    }
    if ((ABL & 16) && p.M != -1) return; // (p.M is never -1: keeps the accumulators live)
    if (EPI == EPI_INT32 || (ABL & 16)) { // debug / unfused API: raw accumulators, 16-byte stores straight from the MFMA layout
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + wm * 128 + j * 32 + lr;
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
    // ---- who finishes what.  Bit r of `todo`: this workgroup runs the epilogue on the share of rank r.
    unsigned todo = 1u << rank;
    if (SPLITK && !solo) {
        stamp(4);
        int* const mine = ws + ((size_t)t_split * S + rank) * SLOT + tid;
        if (!am_last) {
            // bounded wait for the remaining tickets (the partners were dispatched together with this workgroup in every
            // observed case, so this is ~1-2 us; nothing below depends on that)
            if (tid == 0) {
                const unsigned long long t0 = wall_clock64();
                unsigned seen;
                for (;;) {
                    seen = __hip_atomic_load(words, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (seen >= (unsigned)S || wall_clock64() - t0 >= (unsigned long long)p.splitk_patience) break;
                    __builtin_amdgcn_s_sleep(2);
                }
                sk_bcast[1] = seen >= (unsigned)S ? 1u : 0u;
            }
            __syncthreads();
            bool all_here = sk_bcast[1] != 0u;
            if (!all_here) { // defer: park the own share as well, then race the last arriver for its state word
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int x = 0; x < PJ; ++x)
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            __hip_atomic_store(mine + ((i * PJ + x) * 16 + e) * T, acc[i][x][e], __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_AGENT);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) {
                    unsigned expected = 0u;
                    const bool mine_now = __hip_atomic_compare_exchange_strong(words + 1 + rank, &expected, DEFERRED,
                                                                               __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                                                               __HIP_MEMORY_SCOPE_AGENT);
                    sk_bcast[1] = mine_now ? 0u : 1u; // lost to SEEN: the last arriver is here, so everybody is
                }
                __syncthreads();
                all_here = sk_bcast[1] != 0u;
                if (!all_here) { // deferred for good: the last arriver finishes this share.  Leave (whole workgroup).
                    if (tid == 0) {
                        const unsigned before = __hip_atomic_fetch_add(words + 9, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (before == (unsigned)(S - 1))
#pragma unroll
                            for (int w = 0; w < 10; ++w)
                                __hip_atomic_store(words + w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    return;
                }
            }
        } else { // last ticket: every partner has parked.  Mark the shares whose owners are still around as SEEN; the
                 // ones found DEFERRED are this workgroup's to finish.
            if (tid == 0) {
                unsigned mask = 1u << rank;
                for (int q = 0; q < S; ++q) {
                    if (q == rank) continue;
                    unsigned expected = 0u;
                    if (!__hip_atomic_compare_exchange_strong(words + 1 + q, &expected, SEEN, __ATOMIC_RELAXED,
                                                              __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
                        mask |= 1u << q; // (expected == DEFERRED)
                }
                sk_bcast[1] = mask;
            }
            __syncthreads();
            todo = sk_bcast[1];
        }
        stamp(5);
    }

    // ---- own share (still in registers): add the other workgroups' partial sums of it.  This is the common path and is
    // kept exactly as it was before the deferral protocol: 64 / 96 / 112 loads per lane, all in flight together.
    if (SPLITK && !solo) {
        int pk[S > 1 ? S - 1 : 1][NI][PJ][16];
#pragma unroll
        for (int sh = 1; sh < S; ++sh) { // share sh of workgroup rank ^ sh is this workgroup's own share
            const int* const theirs = ws + ((size_t)t_split * S + (rank ^ sh)) * SLOT + tid;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int x = 0; x < PJ; ++x)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        pk[sh - 1][i][x][e] = __hip_atomic_load(theirs + (((sh * NI + i) * PJ + x) * 16 + e) * T,
                                                                __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int sh = 1; sh < S; ++sh)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int x = 0; x < PJ; ++x)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][x][e] += pk[sh - 1][i][x][e];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // every wave of this workgroup has its data
        todo &= ~(1u << rank);
        if (todo == 0u && tid == 0) { // done with the words and with everybody's parked sums
            const unsigned before = __hip_atomic_fetch_add(words + 9, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (before == (unsigned)(S - 1)) { // the whole group is done: re-arm for the next launch
#pragma unroll
                for (int w = 0; w < 10; ++w) __hip_atomic_store(words + w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        stamp(6);
    }
    if (!SPLITK || solo) stamp(3);

    auto epilogue = [&]() __attribute__((always_inline)) {
    // ---- dequant math + stores, one 32 (m) x 64 (n) block of the wave tile at a time.  Results are packed to fp16 and
    // transposed through a wave-private 4-KiB LDS window (32 rows x 128 B; the 32 KiB above the slice buffers), then
    // written as 128-byte row segments (8 rows per store instruction).  No workgroup barrier is involved: LDS
    // operations of one wave execute in order, so each wave streams side GEMM -> dequant -> ds_write -> ds_read ->
    // global_store on its own, and the stores of block j fly under the math of block j+1.
    // The side GEMM always runs 8 k-steps of 16 outlier columns (chunks past O were staged as zeros).
    // Window layout: 16-B chunk c of row r at chunk c ^ (r & 7); its two 8-B halves are swapped on rows with bit 3 set
    // (rows r and r+8 would otherwise hit the same banks in one ds_write_b64 pass).
    char* const wstg = smem + 2 * BUF + wave * 4096;
    const int obase = (lh ^ (lr & 15)) << 4; // 16-B slot of k-step ks = obase ^ (ks << 5)
    // row scale of block j (this lane's row): fetched ONE BLOCK AHEAD instead of all four up front -- the four values + their
    // addresses pushed the epilogue over the 256-register budget: 3 dwords per lane spilled to scratch, 6 KiB per tile, which
    // is exactly the +4.7 % (67 / 64) by which WRITE_SIZE of this kernel exceeded 2 M N in round 2 (profiles/README.md)
    auto load_sa = [&](int j) __attribute__((always_inline)) -> uint16_t { // (clamped rows are never stored)
        return p.sA[min(m0 + wm * 128 + jmap(j) * 32 + lr, p.M - 1)];
    };
    uint16_t sa_next = EPI == EPI_F16GEMM ? (uint16_t)0 : load_sa(0);
    float sa_cur = 1.f;

    // side GEMM of tile (i, j): 8 k-steps of 16 outlier columns
    auto side = [&](int i, int j) __attribute__((always_inline)) {
        v16f P;
#pragma unroll
        for (int e = 0; e < 16; ++e) P[e] = 0.f;
        if (HAS_O) {
            const char* xo = smem + (wn * 64 + imap(i) * 32 + lr) * OSLICE;
            const char* yo = smem + BN * OSLICE + (wm * 128 + jmap(j) * 32 + lr) * OSLICE;
#pragma unroll
            for (int kh = 0; kh < 2; ++kh) { // two batches of 4 k-steps keep the fragment registers at 32
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
    // weight scales of this lane's 2 x 4 column quads (the same for all four j): loaded once, kept as packed fp16
    uint2 swq[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            swq[i][g] = EPI == EPI_F16GEMM ? make_uint2(0u, 0u)
                                           : *reinterpret_cast<const uint2*>(p.sW + min(n0 + wn * 64 + imap(i) * 32 + 4 * lh + 8 * g, p.N - 4));
    // [M,N] fp16 operands of the epilogue (the caller's addend y, the gate*up multiplicand) take the store path in
    // reverse: coalesced 16-byte loads of 128-byte row segments (8 rows per instruction), one block ahead, then through
    // the wave's window into the accumulator layout (8 bytes = 4 consecutive n of row lane & 31).  Reading them in the
    // accumulator layout directly costs 32 cache lines of 16 useful bytes per load instruction.
    constexpr bool HAS_MUL = EPI == EPI_DEQUANT_SILU_MUL;
    uint2 yq[2][4], mq[2][4];
    uint4 ypre[4], mpre[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ypre[q] = mpre[q] = make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) yq[i][g] = mq[i][g] = make_uint2(0u, 0u);
    const int64_t src_wave = ((int64_t)(m0 + wm * 128) * p.N + n0 + wn * 64) * 2; // byte offset of the wave tile
    const unsigned src_lane = ((unsigned)(lane >> 3) * (unsigned)p.N + (lane & 7) * 8) * 2;
    const bool src_n_ok = n0 + wn * 64 + (lane & 7) * 8 < p.N;
    const int win_rd = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4);
    auto fetch = [&](const uint16_t* src, int j, uint4 (&v)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = jmap(j) * 32 + q * 8; // wave-uniform
            const char* a = reinterpret_cast<const char*>(src) + src_wave + (int64_t)row * p.N * 2 + src_lane;
            const bool ok = src_n_ok && m0 + wm * 128 + row + (lane >> 3) < p.M;
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
                out[i][g] = *reinterpret_cast<const uint2*>(wstg + lr * 128 + (((imap(i) * 4 + g) ^ (lr & 7)) << 4) + lh * 8);
    };
    if (HAS_Y) fetch(p.Y, 0, ypre);
    if (HAS_MUL) fetch(p.Mul, 0, mpre);
    // dequant of tile (i, j) with its side product P -> 4 quads (4 consecutive n of row m each) -> window
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
                const v2f s2 = v2f{swf[e2], swf[e2 + 1]} * sa_cur; // exact: fp16 x fp16 products
                // addend pair: the fp16-rounded outlier products (cuBLAS writes fp16) or the caller's y
                v2f c2;
                if (HAS_O) {
                    const v2h p16 = f2h2(P[4 * g + e2], P[4 * g + e2 + 1]); // MFMA outputs: nothing to fuse with
                    c2 = v2f{(float)p16[0], (float)p16[1]};
                } else {
                    c2 = v2f{h2f(yh[e2]), h2f(yh[e2 + 1])};
                }
                float v0, v1;
                if constexpr (EPI == EPI_F16GEMM) { // fp32 sums of the fp16 GEMM: one rounding to fp16 below
                    // (through named ints: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0)
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
            if (S == 8) { // a single 32x32 tile per wave: 8-byte stores straight from the accumulator layout
                const int m = m0 + wm * 128 + jmap(j) * 32 + lr;
                const int n = n0 + wn * 64 + imap(i) * 32 + 4 * lh + 8 * g;
                if (m < p.M && n < p.N)
                    *reinterpret_cast<uint2*>(static_cast<uint16_t*>(p.D) + (int64_t)m * p.N + n) = uint2{ow[0], ow[1]};
                continue;
            }
            const int c = i * 4 + g; // 16-byte chunk of the 128-byte row
            *reinterpret_cast<uint2*>(wstg + wrow + ((c ^ (lr & 7)) << 4)) = uint2{ow[0], ow[1]};
        }
    };
    // rows j*32 .. j*32+31 of the wave tile: window -> 4 x (8 rows x 128 B) stores
    // store address = wave-uniform base (SGPRs) + a per-lane 32-bit offset that never changes
    char* const dwave = static_cast<char*>(p.D) + ((int64_t)(m0 + wm * 128) * p.N + n0 + wn * 64) * 2;
    // (window read offset rd = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 3)) << 4) and the store offset dlane: see flush)
    const bool n_ok = n0 + wn * 64 + (lane & 7) * 8 < p.N;
    const bool interior = m0 + BM <= p.M && n0 + BN <= p.N; // wave-uniform: no store predicates needed
    // TP: the same 128-byte row segments, once per destination, into [M, ldd] buffers (wave-uniform base per destination)
    const int64_t tp_wave = TP ? ((int64_t)(m0 + wm * 128) * p.tp.ldd + n0 + wn * 64) * 2 : 0;
    const unsigned tp_lane = TP ? ((unsigned)(lane >> 3) * (unsigned)p.tp.ldd + (lane & 7) * 8) * 2 : 0u;
    auto flush = [&](int j) __attribute__((always_inline)) {
        // (the two per-lane offsets of the flush are re-derived from an opaque copy of the lane id instead of living in
        //  registers across the side GEMM + dequant of the block: that was the last value the epilogue spilled)
        unsigned ln = (unsigned)lane;
        asm volatile("" : "+v"(ln));
        const int rd = (int)((ln >> 3) * 128 + (((ln & 7) ^ (ln >> 3)) << 4));
        const unsigned dlane = ((ln >> 3) * (unsigned)p.N + (ln & 7) * 8) * 2;
        uint4 v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const uint4*>(wstg + q * 1024 + rd);
#pragma unroll
        for (int q = 0; q < 4; ++q) // (opaque: keeps the four reads together, ahead of the predicated stores)
            asm volatile("" : "+v"(v[q].x), "+v"(v[q].y), "+v"(v[q].z), "+v"(v[q].w));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (q & 1) v[q] = uint4{v[q].z, v[q].w, v[q].x, v[q].y}; // rows with bit 3 set hold their 8-B halves swapped
            const int row = jmap(j) * 32 + q * 8;     // wave-uniform
            if constexpr (TP) {
                const bool ok = interior || (n_ok && m0 + wm * 128 + row + (int)(ln >> 3) < p.M);
                const v4i val = {(int)v[q].x, (int)v[q].y, (int)v[q].z, (int)v[q].w};
                for (int r = 0; r < p.tp.ndst; ++r) { // (uniform trip count; bases are scalars)
                    char* dst = static_cast<char*>(p.tp.base[r]) + tp_wave + (int64_t)row * p.tp.ldd * 2 + tp_lane;
                    // write-through at system scope: acknowledged = performed at the destination (local HBM or a peer's
                    // over xGMI); no L2 write-back fence is needed before the flag (tp_kernels.hip)
                    if (ok) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(val) : "memory");
                }
                continue;
            }
            char* dst = dwave + (int64_t)row * p.N * 2 + dlane;
            const v4i vv = {(int)v[q].x, (int)v[q].y, (int)v[q].z, (int)v[q].w};
            // (non-temporal: the 2 M N output bytes are never read by this launch; streaming them past the L2 leaves it to the operand
            //  panels: -0.7 / -1.6 / -2.6 % on 16384 x 12288 x 4096 / 11008 x 4096 / 4096 x 11008, profiles/r03_nt_store_ab.txt)
            if (interior) __builtin_nontemporal_store(vv, reinterpret_cast<v4i*>(dst));
            else if (n_ok && m0 + wm * 128 + row + (int)(ln >> 3) < p.M) __builtin_nontemporal_store(vv, reinterpret_cast<v4i*>(dst));
        }
    };
    // software pipeline over the 8 tiles (j-major): the MFMA chain of tile t+1 runs under the VALU work of tile t
    {
        v16f Pcur = side(0, 0);
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if (SPLITK && t >= nt) break; // (uniform)
            v16f Pnext = Pcur;
            if (t + 1 < nt) Pnext = side((t + 1) & 1, (t + 1) >> 1);
            if (EPI != EPI_F16GEMM && (t & 1) == 0) { // block j = t >> 1 starts: its row scale arrives, the next one is requested
                sa_cur = h2f(sa_next);
                if (t + 2 < nt) sa_next = load_sa((t >> 1) + 1);
            }
            if ((HAS_Y || HAS_MUL) && (t & 1) == 0) { // block j = t >> 1 starts: operands of this block -> registers
                if (HAS_Y) spread(ypre, yq);
                if (HAS_MUL) spread(mpre, mq);
                if (t + 2 < nt) {
                    if (HAS_Y) fetch(p.Y, (t >> 1) + 1, ypre);
                    if (HAS_MUL) fetch(p.Mul, (t >> 1) + 1, mpre);
                }
            }
            dequant(t & 1, t >> 1, Pcur);
            if (S != 8 && (t & 1)) flush(t >> 1);
            Pcur = Pnext;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    };
    epilogue();
    if constexpr (TP) {
        // every wave's peer stores acknowledged -> count this tile into its M chunk; the chunk's last tile publishes
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const int ctr = p.tp.chunk_tile_rows;
            const int chunk = tile_m / ctr;
            const int rows_here = min(ctr, tiles_m - chunk * ctr);
            const unsigned before = __hip_atomic_fetch_add(p.tp.counters + chunk, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (before == (unsigned)(rows_here * tiles_n) - 1u) {
                __hip_atomic_store(p.tp.counters + chunk, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // re-armed
                for (int r = 0; r < p.tp.ndst; ++r)
                    __hip_atomic_store(p.tp.flag[r] + chunk, p.tp.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
    // ---- last arriver only: the shares whose owners deferred.  All S partial sums of such a share are parked (share
    // w ^ erank of workgroup w; share 0 = the owner's own); the epilogue runs once more with that rank's tile mapping.
    if (SPLITK && !solo && todo != 0u) {
        while (todo != 0u) {
            const int erank = __builtin_ctz(todo);
            todo &= ~(1u << erank);
            ejperm = (erank * PJ) & 3, eiperm = S == 8 ? erank >> 2 : 0;
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int x = 0; x < PJ; ++x)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][x][e] = 0;
            for (int w = 0; w < S; ++w) {
                const int* const theirs = ws + ((size_t)t_split * S + w) * SLOT + tid;
                const int sh = w ^ erank;
                int pk1[NI][PJ][16];
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int x = 0; x < PJ; ++x)
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            pk1[i][x][e] = __hip_atomic_load(theirs + (((sh * NI + i) * PJ + x) * 16 + e) * T,
                                                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                for (int i = 0; i < NI; ++i)
#pragma unroll
                    for (int x = 0; x < PJ; ++x)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[i][x][e] += pk1[i][x][e];
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (todo == 0u && tid == 0) {
                const unsigned before = __hip_atomic_fetch_add(words + 9, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (before == (unsigned)(S - 1))
#pragma unroll
                    for (int w = 0; w < 10; ++w)
                        __hip_atomic_store(words + w, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            epilogue();
        }
    }
    if (!SPLITK || solo) {
        stamp(4);
        stamp(5);
        stamp(6);
    }
    if (p.dbg != nullptr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(7);
    }
}

template <int EPI, bool HAS_O, bool HAS_Y, int ABL = 0>
static hipError_t launch_pp_cfg(const GemmParams& p, hipStream_t st)
{
    constexpr size_t lds = 2 * (size_t)pp::BUF + 32768; // slice buffers + 8 x 4-KiB store windows
    auto kern = gemm_w8a8o16_pp_kernel<EPI, HAS_O, HAS_Y, ABL>;
    static DeviceOnce once;
    if (hipError_t e = ensure_dynamic_lds(kern, lds, once); e != hipSuccess) return e;
    const int tiles = ((p.M + pp::BM - 1) / pp::BM) * ((p.N + pp::BN - 1) / pp::BN);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(pp::T), lds, st, p);
    return hipGetLastError();
}

hipError_t launch_gemm_f16_pp(const void* A, const void* B, void* D, int M, int N, int K, const void* zeros, hipStream_t st);

// ---- peer-write epilogue (TpEpilogue) --------------------------------------------------------------------------------
int tp_chunk_tile_rows(int M)
{
    const int tiles_m = (M + pp::BM - 1) / pp::BM;
    const int groups = (tiles_m + 3) / 4;                        // the kernel walks the tile rows in groups of GROUP_M = 4
    return 4 * ((groups + kTpFlagWords - 1) / kTpFlagWords);     // whole groups per flag word, at most kTpFlagWords words
}
int tp_flag_words(int M)
{
    const int tiles_m = (M + pp::BM - 1) / pp::BM, c = tp_chunk_tile_rows(M);
    return M <= 0 ? 1 : (tiles_m + c - 1) / c;
}
template <bool HAS_O>
static hipError_t launch_pp_tp_cfg(const GemmParams& p, hipStream_t st)
{
    constexpr size_t lds = 2 * (size_t)pp::BUF + 32768;
    auto kern = gemm_w8a8o16_pp_kernel<EPI_DEQUANT, HAS_O, false, 0, 0, true>;
    static DeviceOnce once;
    if (hipError_t e = ensure_dynamic_lds(kern, lds, once); e != hipSuccess) return e;
    const int tiles = ((p.M + pp::BM - 1) / pp::BM) * ((p.N + pp::BN - 1) / pp::BN);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(pp::T), lds, st, p);
    return hipGetLastError();
}
hipError_t launch_gemm_pp_tp(const GemmParams& p_in, hipStream_t st)
{
    if (p_in.tp.ndst < 1 || p_in.tp.ndst > kTpMaxPeers || p_in.Y != nullptr || p_in.M <= 0 || p_in.N <= 0) return hipErrorInvalidValue;
    GemmParams p = p_in;
    p.tp.chunk_tile_rows = tp_chunk_tile_rows(p.M);
    note_gemm_kernel("gemm_w8a8o16_pp_kernel<TP> (256x256 ping-pong, peer-write epilogue)");
    return p.O > 0 ? launch_pp_tp_cfg<true>(p, st) : launch_pp_tp_cfg<false>(p, st);
}

template <int EPI>
static hipError_t launch_pp_epi(const GemmParams& p, hipStream_t st)
{
    const bool has_o = p.O > 0, has_y = p.Y != nullptr;
    if (has_o) return launch_pp_cfg<EPI, true, false>(p, st); // the API never passes both an addend and outliers
    if (has_y) return launch_pp_cfg<EPI, false, true>(p, st);
    return launch_pp_cfg<EPI, false, false>(p, st);
}

hipError_t launch_gemm_f16_pp(const void* A, const void* B, void* D, int M, int N, int K, const void* zeros, hipStream_t st)
{
    if (K % 8 || N % 8) return hipErrorInvalidValue;
    if (M <= 0 || N <= 0) return hipSuccess;
    GemmParams p{};
    p.A = static_cast<const int8_t*>(A), p.B = static_cast<const int8_t*>(B), p.D = D, p.zeros = zeros;
    p.M = M, p.N = N, p.K = 2 * K; // the kernel counts K in bytes
    return launch_pp_cfg<EPI_F16GEMM, false, false>(p, st);
}

// ---- K split over 2 / 4 workgroups per tile (see the kernel header) ---------------------------------------------------
int num_cus()
{
    static std::atomic<int> cache[64];
    const int dev = current_device();
    int n = dev < 64 ? cache[dev].load(std::memory_order_relaxed) : 0;
    if (n == 0) {
        int v = 0;
        n = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
        if (dev < 64) cache[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

static std::atomic<unsigned> g_splitk_patience{3000u}; // wall-clock ticks (100 MHz) = 30 us; 0: defer at once (tests)
void set_splitk_patience(unsigned ticks) { g_splitk_patience.store(ticks); }
static std::atomic<int> g_splitk_force{-1}; // -1 automatic, 0 off, 2 / 4: that factor wherever the shape allows it
void set_splitk_force(int v) { g_splitk_force.store(v); }

// Automatic choice, from tools/splitk_select_sweep.py on MI355X.
//  * tiles <= CUs / 2 (one partial wave of tiles): with the tiles on (1/4, 1/2] of the CUs two workgroups per tile
//    always pay; on at most 1/4 of the CUs four pay once K is long enough to amortise the exchange (3 x 64 KiB per
//    workgroup each way) and there are enough tiles for the small-tile kernels to be L2-bound; 129..255 rows take the
//    4- and 8-way forms of one 256-row tile row (cold weights: see below), up to 128 rows the small-tile kernels.
//  * tiles > CUs (whole waves + a partial one): the whole waves run one workgroup per tile ("solo"), the tiles of the
//    last wave are split if they cover at most half of the CUs -- instead of a last wave that takes a full tile time
//    on a fraction of the chip.
SplitPlan gemm_splitk_plan(int M, int N, int K)
{
    SplitPlan none{0, 0};
    const int force = g_splitk_force.load();
    if (force == 0 || M <= 128) return none;
    if (force < 0 && gemm_pp128_wins(M, N, K)) return none; // one wave of 128 x 256 tiles instead (gemm_pp128_kernels.hip)
    const int tiles = ((M + pp::BM - 1) / pp::BM) * ((N + pp::BN - 1) / pp::BN);
    const int nk = (K + pp::KS - 1) / pp::KS;
    const int cus = num_cus() & ~7;
    if (tiles > cus) { // hybrid
        const int tail = tiles % cus, rounds = tiles / cus;
        if (tail == 0 || 2 * tail > cus || (force < 0 && M < 256)) return none; // (under 256 rows: N > 65536, not measured)
        // Round 4: re-fitted in STEADY STATE (tools/splitk_select_sweep.py --hybrid --secs 0.35, profiles/r04_hybrid_sweep.txt: the
        // power-capped clock a prefill runs at; round 1's 100-launch cells read the boost clock of an idle chip and over-sold the
        // split).  What a split tail saves shrinks with the number of whole rounds before it -- by then the CUs are out of step
        // and the plain kernel's partial round already overlaps the one before -- and grows with K (the exchange is a fixed
        // 128..192 KiB per workgroup each way).  Rows: whole rounds 1 | 2 | 3-4 | 5-7 | 8+; columns: tail on <= 1/8 | 1/4 | 3/8 |
        // 1/2 of the CUs; entry = the fewest 128-byte K slices from which the split measured >= ~2 % ahead of the plain launch
        // (r = 1: -5..-23 %; r = 2: -3..-13 %; r = 5: -1..-6 %, only the long K).
        int s = 0;
        if (force < 0) {
            static const int kMinNk[5][4] = {{24, 24, 24, 56}, {24, 24, 80, 80}, {24, 56, 100, 100}, {48, 128, 128, 128}, {64, 160, 160, 160}};
            const int fb = 8 * tail <= cus ? 0 : 4 * tail <= cus ? 1 : 8 * tail <= 3 * cus ? 2 : 3;
            const int rc = rounds == 1 ? 0 : rounds == 2 ? 1 : rounds <= 4 ? 2 : rounds <= 7 ? 3 : 4;
            if (nk >= kMinNk[rc][fb]) s = fb >= 2 ? 2 : (rc == 0 && fb == 1 && nk < 48 ? 2 : 4); // (one round + a quarter, short K: 2 ways tie or win)
        } else {
            s = 4 * tail <= cus && nk >= 24 ? 4 : (nk >= 12 ? 2 : 0);
            if (force == 2 && nk >= 8) s = 2;
            if (force == 2 && s == 4) s = 2;
            if (force == 4 && s != 4) s = 4 * tail <= cus && nk >= 16 ? 4 : 0;
            if (force == 8) s = 8 * tail <= cus && nk >= 32 ? 8 : 0;
        }
        return s ? SplitPlan{s, tiles - tail} : none;
    }
    if (force == 8) return 8 * tiles <= cus && nk >= 32 ? SplitPlan{8, 0} : none;
    if (force == 4) return 4 * tiles <= cus && nk >= 16 ? SplitPlan{4, 0} : none;
    if (force == 2) return 2 * tiles <= cus && nk >= 8 ? SplitPlan{2, 0} : none;
    if (2 * tiles > cus) return none;
    // Round 4, re-fitted on COLD weights (tools/midm_cfg_sweep.py --cold --only auto,s2,s4,s8, profiles/r04_splitk_cold_fit.txt: the
    // weights cycle through > 320 MiB of copies -- no layer of a model finds its weights cache-resident -- where the small-tile forms,
    // whose workgroups re-read the weights from L2, lose more than the 256 x 256 tiles do):
    //  * 129..255 rows take the plan of one 256-row tile row for the 4- and 8-way forms (us, small tiles vs split, 192 rows:
    //    3584 x 18944 57.7 / 42.8, 4096 x 16384 48.6 / 37.5, 8192 x 28672 111 / 73, 6144 x 12288 49.4 / 39.0, 5120 x 13824 44.3 / 38.9;
    //    ties within 2 % at 4096 x 11008, 8192 x 8192, 10240 x 8192; at 224 rows 10240 x 8192 56.2 / 38.2);
    //  * 8 ways with 20..24 tiles (at most 3/4 of the CUs busy; 25..31 tiles: from K = 11008 as before, 4 ways below that -- 3584 x 8192 at
    //    384 / 512 rows = 28 tiles x 8 = 224 workgroups read 32.8 / 33.6 us on one box and 35.4 / 38.6 on another, 4 ways 33.4..34.2 on both,
    //    no split 34.0 / 37.1) from K = 8192 unless the 64 x 64 tiles still fit one wave of the chip (5120 x 8192 at 256 rows 39.0 -> 29.4,
    //    2560 x 8192 at 512 rows 36.5 -> 30.0; with one wave of 64 x 64 tiles 2560 x 8192 at 384 rows 27.5 vs 30.4, 5120 x 8192 at 192 rows
    //    28.4 vs 29.1: then from K = 12800);
    //  * 8 ways on exactly 1/8 of the CUs (8 x 32 = every CU holds one workgroup) from K = 16384 (4096 x 16384 at 384 / 512 rows 4 / 8
    //    ways 53.6 / 48.9, 54.9 / 50.3).  NOT from K = 11008: that launch measured 41.4, 43.6 and 49.8 us on three boxes of the pool
    //    (4 ways: 43.1..43.5 on all of them) -- a launch that needs every CU at once is at the mercy of the slowest one.
    const bool tall = M >= 256; // (the 2-way forms were fitted from 256 rows on only)
    if (4 * tiles > cus) return tall && nk >= 16 ? SplitPlan{2, 0} : none;
    if (8 * tiles <= cus) { // at most 1/8 of the CUs: 8 ways once K amortises 7 x 32 KiB each way per workgroup
        const int64_t wg64 = (int64_t)((M + 63) / 64) * ((N + 63) / 64);
        const bool pays = 8 * tiles == cus        ? nk >= 128
                          : 64 * tiles >= 5 * cus ? (32 * tiles <= 3 * cus ? nk >= (wg64 <= cus ? 100 : 64) : nk >= 86)
                          : 64 * tiles >= 3 * cus ? nk >= 86
                                                  : 32 * tiles >= cus && nk >= 160;
        if (pays) return SplitPlan{8, 0};
    }
    if (nk >= 64 && 8 * tiles >= cus) return SplitPlan{4, 0};
    if (nk >= 96 && 64 * tiles >= 5 * cus) return SplitPlan{4, 0}; // 20..31 tiles pay with K >= 12288 (-7..-27 %)
    if (nk >= 64 && 32 * tiles > 3 * cus && (int64_t)((M + 63) / 64) * ((N + 63) / 64) > cus) return SplitPlan{4, 0}; // 25..31 tiles, K = 8192 (above)
    if (tall && nk >= 40 && 16 * tiles >= 3 * cus) return SplitPlan{2, 0};
    return none;
}

int gemm_splitk_factor(int M, int N, int K) { return gemm_splitk_plan(M, N, K).s; }

static size_t splitk_slot_bytes(int) // S shares of a 256x256 int32 tile's 1/S-th: the S - 1 foreign ones + the own share,
{                                      // which is parked only when a workgroup defers (see the kernel header)
    return (size_t)pp::BM * pp::BN * 4;
}

size_t gemm_splitk_workspace_size(int M, int N, int K)
{
    const SplitPlan pl = gemm_splitk_plan(M, N, K);
    if (pl.s == 0) return 0;
    const size_t tiles = (size_t)((M + pp::BM - 1) / pp::BM) * ((N + pp::BN - 1) / pp::BN) - pl.solo;
    return kSplitkWordsBytes + tiles * pl.s * splitk_slot_bytes(pl.s);
}

size_t gemm_splitk_workspace_bound() { return kSplitkWordsBytes + (size_t)num_cus() * splitk_slot_bytes(8); }

template <int EPI, bool HAS_O, bool HAS_Y, int SPLITK>
static hipError_t launch_pp_splitk_cfg(const GemmParams& p, hipStream_t st)
{
    constexpr size_t lds = 2 * (size_t)pp::BUF + 32768;
    auto kern = gemm_w8a8o16_pp_kernel<EPI, HAS_O, HAS_Y, 0, SPLITK>;
    static DeviceOnce once;
    if (hipError_t e = ensure_dynamic_lds(kern, lds, once); e != hipSuccess) return e;
    const int tail = ((p.M + pp::BM - 1) / pp::BM) * ((p.N + pp::BN - 1) / pp::BN) - p.splitk_solo;
    constexpr int G = 8 / SPLITK;
    hipLaunchKernelGGL(kern, dim3((unsigned)(p.splitk_solo + 8 * ((tail + G - 1) / G))), dim3(pp::T), lds, st, p);
    return hipGetLastError();
}

template <int EPI, int SPLITK>
static hipError_t launch_pp_splitk_epi(const GemmParams& p, hipStream_t st)
{
    if (p.O > 0) return launch_pp_splitk_cfg<EPI, true, false, SPLITK>(p, st);
    if (p.Y != nullptr) return launch_pp_splitk_cfg<EPI, false, true, SPLITK>(p, st);
    return launch_pp_splitk_cfg<EPI, false, false, SPLITK>(p, st);
}

hipError_t launch_gemm_pp_splitk(const GemmParams& p_in, int epi, hipStream_t st)
{
    const SplitPlan pl = gemm_splitk_plan(p_in.M, p_in.N, p_in.K);
    const int s = pl.s;
    if (s == 0 || p_in.splitk_ws == nullptr) return hipErrorInvalidValue;
    GemmParams p = p_in;
    p.splitk_solo = pl.solo;
    p.splitk_patience = g_splitk_patience.load();
    switch (epi) {
    case EPI_DEQUANT:
        return s == 8   ? launch_pp_splitk_epi<EPI_DEQUANT, 8>(p, st)
               : s == 4 ? launch_pp_splitk_epi<EPI_DEQUANT, 4>(p, st)
                        : launch_pp_splitk_epi<EPI_DEQUANT, 2>(p, st);
    case EPI_DEQUANT_SILU:
        return s == 8   ? launch_pp_splitk_epi<EPI_DEQUANT_SILU, 8>(p, st)
               : s == 4 ? launch_pp_splitk_epi<EPI_DEQUANT_SILU, 4>(p, st)
                        : launch_pp_splitk_epi<EPI_DEQUANT_SILU, 2>(p, st);
    case EPI_DEQUANT_SILU_MUL:
        return s == 8   ? launch_pp_splitk_epi<EPI_DEQUANT_SILU_MUL, 8>(p, st)
               : s == 4 ? launch_pp_splitk_epi<EPI_DEQUANT_SILU_MUL, 4>(p, st)
                        : launch_pp_splitk_epi<EPI_DEQUANT_SILU_MUL, 2>(p, st);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_gemm_pp_ablate(const GemmParams& p, int abl, hipStream_t st)
{
    switch (abl) {
    case 1: return launch_pp_cfg<EPI_DEQUANT, true, false, 1>(p, st);
    case 2: return launch_pp_cfg<EPI_DEQUANT, true, false, 2>(p, st);
    case 4: return launch_pp_cfg<EPI_DEQUANT, true, false, 4>(p, st);
    case 5: return launch_pp_cfg<EPI_DEQUANT, true, false, 5>(p, st);
    case 7: return launch_pp_cfg<EPI_DEQUANT, true, false, 7>(p, st);
    case 8: return launch_pp_cfg<EPI_DEQUANT, true, false, 8>(p, st);
    case 16: return launch_pp_cfg<EPI_DEQUANT, true, false, 16>(p, st);
    case 24: return launch_pp_cfg<EPI_DEQUANT, true, false, 24>(p, st);
    case 21: return launch_pp_cfg<EPI_DEQUANT, true, false, 21>(p, st);
    case 32: return launch_pp_cfg<EPI_DEQUANT, true, false, 32>(p, st); // MFMA operands swapped (transposed tiles: timing only)
    case 64: return launch_pp_cfg<EPI_DEQUANT, true, false, 64>(p, st); // 2 x 16x16x64 per 32x32x32 (timing only)
    case 512: return launch_pp_cfg<EPI_DEQUANT, true, false, 512>(p, st); // probe: A read as K-slice-major (timing only)
    case 128: return launch_pp_cfg<EPI_DEQUANT, true, false, 128>(p, st); // + ~4k idle cycles per tile (correct results)
    case 256: return launch_pp_cfg<EPI_DEQUANT, true, false, 256>(p, st); // + ~16k idle cycles per tile (correct results)
    default: return launch_pp_cfg<EPI_DEQUANT, true, false, 0>(p, st);
    }
}

hipError_t launch_gemm_pp(const GemmParams& p, int epi, hipStream_t st)
{
    switch (epi) {
    case EPI_DEQUANT: return launch_pp_epi<EPI_DEQUANT>(p, st);
    case EPI_DEQUANT_SILU: return launch_pp_epi<EPI_DEQUANT_SILU>(p, st);
    case EPI_DEQUANT_SILU_MUL: return launch_pp_epi<EPI_DEQUANT_SILU_MUL>(p, st);
    default: return launch_pp_cfg<EPI_INT32, false, false>(p, st);
    }
}

} // namespace mixq

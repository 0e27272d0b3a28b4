// Fused W8A8O16 GEMM for gfx950 (MI355X):  Out = fp16( float(qA.W^T) * (sW[n]*sA[m]) + fp16(fpA.fpW^T) )
//
// Replaces (reference, CUDA):
//   kernel/i8gemm.cu:151-194 int8FusedDequantizeCUDA -> CUTLASS symmetric::GemmDequant
//       (kernel/symmetric/gemm/kernel/gemm_dequant.h:225-380 main loop,
//        kernel/symmetric/epilogue/thread/linear_combination_dequant.h:120-160 epilogue functor)
//   TsinghuaMixQPlugin.cpp:122-161 gemmfp16 (cuBLAS fp16 side GEMM over the 128 outlier columns)
//   TsinghuaMixQPlugin.cpp:36-77   gemm (cuBLAS s8 x s8 -> s32, the unfused route)
// in ONE launch: Out is written once and never re-read (the reference writes it, re-reads it as C and writes it again).
//
// CDNA4 mapping
//   * v_mfma_i32_32x32x32_i8, operands swapped: MFMA "A" = W rows (n), MFMA "B" = qA rows (m).  The 32x32 C/D layout
//     then gives every lane 4 consecutive n for one m per accumulator quad -> 8-byte fp16 stores, and the int32
//     accumulators are bit-exact by construction (integer adds commute).
//   * K is consumed in 128-byte slices per row.  Both operand tiles are staged global -> LDS with 16-byte
//     global_load_lds (no VGPR round trip).  The LDS image is lane-linear (hardware rule), so the bank-conflict
//     swizzle is applied to the per-lane SOURCE address and mirrored on the ds_read_b128 side:
//         16-B slot of row r = chunk ^ ((r >> 1) & 7)       (128-B rows: conflict-free for ds_read_b128's 16-lane groups)
//   * double-buffered LDS, one barrier per K slice: loads of slice t+1 are in flight while slice t is multiplied.
//   * The fp16 outlier side GEMM (O <= 128 columns) runs after the int8 main loop on v_mfma_f32_32x32x16_f16, tile by
//     tile, re-using the main loop's LDS; its fp32 sum is rounded to fp16 first (the reference materialises it in fp16
//     through a separate cuBLAS call) and enters the dequant FMA as the addend.
//   * 1-D grid, XCD-aware remap (block b runs on XCD b % 8): each XCD owns a contiguous run of tiles, walked in groups
//     of GROUP_M row-tiles so that co-resident workgroups share W / qA panels in that XCD's private L2.
#include "mixq_device.h"
#include "mixq_launch.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>

namespace mixq {

constexpr int KSLICE = 128;  // bytes of K per LDS row (int8 elements)
constexpr int OSLICE = 256;  // bytes per LDS row in the outlier phase (128 fp16)
constexpr int GROUP_M = 4;

// KG > 1: in-workgroup split-K.  KG groups of WAVES_M x WAVES_N waves work on the same output tile, group g on the K
// slices g, g + KG, ... with its own LDS stages; the partial accumulators meet in LDS and group 0 runs the epilogue.
// (Small-N problems have too few tiles for the chip, and an LDS-DMA instruction costs the issuing wave 100-200 cycles:
// more waves per tile is what raises the operand stream, and no global scratch is needed.)
// PREO: the outlier operands of the epilogue get their own LDS region behind the stages and are copied there at kernel
// start, under the main loop, instead of after it (used by the one-workgroup-per-CU split-K configurations, where the
// extra (BM + BN) x 256 bytes of LDS cost no occupancy and the kernel is a chain of latencies).
// XSP: K is split over XS = p.xsplit WORKGROUPS per tile as well (consecutive blocks), for problems with so few tiles that most
// CUs would otherwise watch a few of them stream K.  Every workgroup parks its partial tile (BM x BN int32: 8-16 KiB)
// in p.splitk_ws with write-through stores and counts itself in; the one that arrives LAST adds the others' parked
// sums and runs the epilogue, the others are gone by then -- nobody ever waits, so there is nothing to dead-lock.
// Hand-over accesses are relaxed agent-scope atomics as in gemm_pp_kernels.hip (no fences); the counter of a tile lives
// in the first kSplitkWordsBytes of the scratch and is left zero.  Same int32 sums, same bits.
// ADMA (round 5): the slice copies are issued in ASM form (glds16_vaddr: not counted by the compiler's s_waitcnt insertion), so that
// NSTAGE - 1 slices really are in flight behind the explicit `s_waitcnt vmcnt(n)` of the loop -- with the builtin copies hipcc
// drains everything in front of every barrier and the loop runs ONE slice ahead whatever NSTAGE says (notebook R3.16: no effect
// where 2-5 workgroups share a CU and cover each other's round trips).  It is the regime with ONE workgroup per CU that needs it:
// mid-M problems (129..512 rows) whose 128 x 128 tiles, split over K, put exactly one workgroup on every CU -- there a slice's
// ~1 us L2 / HBM round trip, not its ~0.3 us of MFMA work, set the pace of the two-slice loop.
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int NSTAGE = 2, int KG = 1, bool PREO = false, bool XSP = false,
          bool ADMA = false>
__global__ __launch_bounds__(WAVES_M* WAVES_N* KG * 64) void gemm_w8a8o16_kernel(const GemmParams p)
{
    const int XS = XSP ? p.xsplit : 1; // workgroups per tile (2 / 4 / 8 / 16: index arithmetic only, so a run-time value)
    constexpr int NWAVES = WAVES_M * WAVES_N;
    constexpr int T = NWAVES * 64;      // threads of one K group
    constexpr int TT = T * KG;          // threads of the workgroup
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int X_BYTES = BN * KSLICE, Y_BYTES = BM * KSLICE;
    constexpr int STAGE_BYTES = X_BYTES + Y_BYTES;
    constexpr int XL = BN * 8 / T, YL = BM * 8 / T;       // 16-B loads per thread per K slice
    static_assert(BN * 8 % T == 0 && BM * 8 % T == 0, "tile rows must split evenly over the block");
    static_assert(WM % 32 == 0 && WN % 32 == 0, "wave tile must be a multiple of the 32x32 MFMA");
    static_assert((BM + BN) * OSLICE <= KG * NSTAGE * STAGE_BYTES, "outlier tiles must fit the main-loop LDS");
    static_assert((KG - 1) * NWAVES * TN * TM * 4096 <= KG * NSTAGE * STAGE_BYTES, "partial accumulators must fit");
    static_assert(BN * 16 % TT == 0 && BM * 16 % TT == 0, "outlier tile rows must split evenly over the block");
    static_assert(NSTAGE >= 2 && NSTAGE <= 8, "2..8 LDS stages");

    extern __shared__ __attribute__((aligned(16))) char smem_all[];

    const int tid_all = threadIdx.x;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid_all >> 6);
    const int group = KG > 1 ? wave_all / NWAVES : 0;
    const int tid = KG > 1 ? tid_all - group * T : tid_all;   // thread inside its K group
    const int lane = tid & 63;
    const int wave = KG > 1 ? wave_all - group * NWAVES : wave_all;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    char* const smem = smem_all + group * (NSTAGE * STAGE_BYTES); // this group's stages

    // ---- block -> tile mapping (XCD-aware, grouped) ------------------------------------------------
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    int t_lin;
    const int xrank = XSP ? (int)blockIdx.x % XS : 0; // which part of K this workgroup multiplies
    {
        const int bid = XSP ? (int)blockIdx.x / XS : (int)blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        t_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3); // bijective for any nwg
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

    // ---- per-thread global source offsets for the staging loads ------------------------------------
    // chunk index c = i*T + tid : LDS row = c/8, LDS slot = c%8 (linear image), source chunk = slot ^ ((row>>1)&7)
    const int64_t K = p.K;
    const char* xsrc[XL];
    const char* ysrc[YL];
    int xkoff, ykoff; // byte offset of this thread's source chunk inside the 128-B slice (same for every i)
    {
        const int slot = tid & 7;
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const int row = (i * T + tid) >> 3;
            const int grow = min(n0 + row, p.N - 1); // clamp: rows past N are loaded but never stored
            xsrc[i] = reinterpret_cast<const char*>(p.B) + (int64_t)grow * K + ((slot ^ ((row >> 1) & 7)) << 4);
        }
        xkoff = (slot ^ (((tid >> 3) >> 1) & 7)) << 4; // (row>>1)&7 does not depend on i (T/8 is a multiple of 16)
#pragma unroll
        for (int i = 0; i < YL; ++i) {
            const int row = (i * T + tid) >> 3;
            const int grow = min(m0 + row, p.M - 1);
            ysrc[i] = reinterpret_cast<const char*>(p.A) + (int64_t)grow * K + ((slot ^ ((row >> 1) & 7)) << 4);
        }
        ykoff = xkoff;
    }
    static_assert((T / 8) % 16 == 0, "row swizzle term must be i-invariant");

    const int nk_all = (p.K + KSLICE - 1) / KSLICE;
    const int kbeg = XSP ? nk_all * xrank / XS : 0;                 // this workgroup's slices: [kbeg, kbeg + nk)
    const int nk = (XSP ? nk_all * (xrank + 1) / XS : nk_all) - kbeg;
    const bool ktail = (p.K % KSLICE) != 0;

    // fpW / fpA tiles -> LDS (256-B rows, slot = chunk ^ (row & 15)); all threads of the workgroup take part
    const bool has_outliers = (EPI != EPI_INT32) && p.O > 0;
    char* const osmem = smem_all + (PREO ? KG * NSTAGE * STAGE_BYTES : 0);
    auto stage_outliers = [&]() __attribute__((always_inline)) {
        constexpr int OXL = BN * 16 / TT, OYL = BM * 16 / TT;
        const int obytes = p.O * 2; // valid bytes per row (O % 8 == 0)
        const int slot = tid_all & 15;
#pragma unroll
        for (int i = 0; i < OXL; ++i) {
            const int row = (i * TT + tid_all) >> 4;
            const int c = (slot ^ (row & 15)) << 4;
            const int grow = min(n0 + row, p.N - 1);
            const char* s2 = reinterpret_cast<const char*>(p.fpW) + (int64_t)grow * obytes + c;
            if (c >= obytes) s2 = static_cast<const char*>(p.zeros);
            glds16(s2, osmem + (i * TT + wave_all * 64) * 16);
        }
#pragma unroll
        for (int i = 0; i < OYL; ++i) {
            const int row = (i * TT + tid_all) >> 4;
            const int c = (slot ^ (row & 15)) << 4;
            const int grow = min(m0 + row, p.M - 1);
            const char* s2 = reinterpret_cast<const char*>(p.fpA) + (int64_t)grow * obytes + c;
            if (c >= obytes) s2 = static_cast<const char*>(p.zeros);
            glds16(s2, osmem + BN * OSLICE + (i * TT + wave_all * 64) * 16);
        }
    };
    if (PREO && has_outliers) stage_outliers(); // older than every slice copy: retired by the loop's first wait

    const unsigned lds_group0 = (unsigned)(size_t)(MIXQ_LDS_PTR(smem)); // LDS byte address of this group's stages (ADMA)
    const bool ntw = (p.flags & 2) != 0; // (launch_cfg: one tile row reads a weight of 32 MiB and more -- each of its lines exactly once)
    auto stage = [&](int buf, int kt) {
        char* xb = smem + buf * STAGE_BYTES;
        char* yb = xb + X_BYTES;
        const int64_t kbyte = (int64_t)(kbeg + kt) * KSLICE;
        const bool last_partial = ktail && (kbeg + kt == nk_all - 1);
#pragma unroll
        for (int i = 0; i < XL; ++i) {
            const char* s = xsrc[i] + kbyte;
            if (last_partial && (kbyte + xkoff >= K)) s = static_cast<const char*>(p.zeros);
            if (ADMA) {
                if (ntw) glds16_vaddr_nt(s, lds_group0 + buf * STAGE_BYTES + (i * T + wave * 64) * 16);
                else glds16_vaddr(s, lds_group0 + buf * STAGE_BYTES + (i * T + wave * 64) * 16);
            } else {
                if (ntw) glds16_nt(s, xb + (i * T + wave * 64) * 16);
                else glds16(s, xb + (i * T + wave * 64) * 16);
            }
        }
#pragma unroll
        for (int i = 0; i < YL; ++i) {
            const char* s = ysrc[i] + kbyte;
            if (last_partial && (kbyte + ykoff >= K)) s = static_cast<const char*>(p.zeros);
            if (ADMA) glds16_vaddr(s, lds_group0 + buf * STAGE_BYTES + X_BYTES + (i * T + wave * 64) * 16);
            else glds16(s, yb + (i * T + wave * 64) * 16);
        }
    };

    // ---- LDS read offsets for the MFMA fragments -----------------------------------------------------
    // lane l supplies row (l & 31) and the 16 K-bytes [(l >> 5) * 16, +16) of each 32-byte MFMA k-step.
    const int lr = lane & 31, lh = lane >> 5;
    const int sw = (lr >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = ((ks * 2 + lh) ^ sw) << 4;
    const int xrow_off = (wn * WN + lr) * KSLICE;
    const int yrow_off = X_BYTES + (wm * WM + lr) * KSLICE;

    v16i acc[TN][TM];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;

    // ---- main loop -------------------------------------------------------------------------------------
    // NSTAGE LDS buffers, slices kt+1 .. kt+NSTAGE-1 in flight while slice kt is multiplied.  Small tiles are bound by
    // the HBM round trip of each slice, not by MFMA work: the deeper the prefetch, the more bytes in flight per CU.
    // (Round 3, read off the ISA: with the BUILTIN copies used here hipcc puts `s_waitcnt vmcnt(0)` in front of every
    //  __syncthreads() -- a pending LDS-DMA copy is a pending LDS write to the fence -- so in steady state only the slice issued
    //  one iteration ago is in flight; NSTAGE > 2 buys the first NSTAGE - 1 slices up front.  The asm-form copies
    //  (glds16_vaddr, explicit waits only) that really keep NSTAGE - 1 slices in flight measured +-1..3 % on 30 shapes from 48
    //  to 512 tokens (profiles/r03_tile_kernel_asm_dma_ab.txt): several workgroups per CU already cover the round trip.  Left
    //  as it was.)
    // (group g owns slices g, g + KG, ...: its it-th slice is kt = it * KG + g; every group runs the same number of
    //  iterations because the barrier is workgroup-wide)
#pragma unroll
    for (int s0 = 0; s0 < NSTAGE - 1; ++s0)
        if (s0 * KG + group < nk) stage(s0, s0 * KG + group);
    const int nit = (nk + KG - 1) / KG;
    const bool late_issue = ADMA && KG == 1 && NWAVES >= 8 && wave >= NWAVES / 2 && !(p.flags & 1);
    for (int it = 0; it < nit; ++it) {
        const int kt = it * KG + group;
        // copies complete in order: at most the (NSTAGE-2) younger slices may still be in flight once slice kt landed
        if (NSTAGE > 2 && (it + NSTAGE - 2) * KG + group < nk)
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSTAGE - 2) * (XL + YL)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads(); // slice kt has landed for every wave; everyone is done reading the previous slice's buffer
        // (all copies of slice it + NSTAGE - 1 at once, as early as possible: spreading them over the four k-steps -- which
        //  pays in the fpA_intB wide form -- measured +4..+8 % here on wide / long-K shapes, -3 % on 4096 x 4096: round 2)
        // ADMA with two waves per SIMD: the waves of the second half issue their copies AFTER their MFMAs, so that on every SIMD one
        // wave multiplies while the other sits in the 100..200-cycle issue stall of each LDS-DMA instruction (the poor man's ping-pong;
        // same counts per wave, so the vmcnt bookkeeping above is unchanged; the target buffer was last read one barrier ago)
        const bool do_stage = (it + NSTAGE - 1) * KG + group < nk;
        if (do_stage && !late_issue) stage((it + NSTAGE - 1) % NSTAGE, (it + NSTAGE - 1) * KG + group);
        if (KG > 1 && kt >= nk) continue; // (K groups past the end only keep the barrier count)
        const char* base = smem + (it % NSTAGE) * STAGE_BYTES;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            v4i xf[TN], yf[TM];
#pragma unroll
            for (int i = 0; i < TN; ++i)
                xf[i] = *reinterpret_cast<const v4i*>(base + xrow_off + i * 32 * KSLICE + koff[ks]);
#pragma unroll
            for (int j = 0; j < TM; ++j)
                yf[j] = *reinterpret_cast<const v4i*>(base + yrow_off + j * 32 * KSLICE + koff[ks]);
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(xf[i], yf[j], acc[i][j], 0, 0, 0);
        }
        if (do_stage && late_issue) {
            __builtin_amdgcn_sched_barrier(0); // (the copies stay behind the slice's MFMAs)
            stage((it + NSTAGE - 1) % NSTAGE, (it + NSTAGE - 1) * KG + group);
        }
    }

    // ---- (round 6) the scales the epilogue multiplies by are requested HERE, under the hand-overs and the outlier staging below: the store
    // loop used to load sA[m] and sW[n .. n + 3] right where it needed them -- a dependent L2 round trip in front of every tile's first store
    // (the mid-M kernel's timeline priced the same pattern at 1.3 us per tile, profiles/r06_mid_v1_timeline.txt).  Clamped: rows / columns past
    // the edge are never stored.
    float sa_pre[TM];
    uint2 sw_pre[TN][4];
    if (EPI != EPI_INT32) {
#pragma unroll
        for (int j = 0; j < TM; ++j) sa_pre[j] = h2f(p.sA[min(m0 + wm * WM + j * 32 + lr, p.M - 1)]);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                sw_pre[i][g] = *reinterpret_cast<const uint2*>(p.sW + min(n0 + wn * WN + i * 32 + 4 * lh + 8 * g, p.N - 4));
    }

    // ---- in-workgroup split-K: partial accumulators of groups 1.. -> LDS -> added by group 0 ----------
    if (KG > 1) {
        __syncthreads(); // main-loop LDS is dead
        if (group > 0) {
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j) {
                    char* r = smem_all + (((group - 1) * NWAVES + wave) * (TN * TM) + i * TM + j) * 4096 + lane * 16;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<v4i*>(r + q * 1024) =
                            v4i{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                }
        }
        __syncthreads();
        if (group == 0) {
#pragma unroll
            for (int g2 = 1; g2 < KG; ++g2)
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j) {
                        const char* r = smem_all + (((g2 - 1) * NWAVES + wave) * (TN * TM) + i * TM + j) * 4096 + lane * 16;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const v4i t4 = *reinterpret_cast<const v4i*>(r + q * 1024);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += t4[e];
                        }
                    }
        }
    }

    // ---- K split over workgroups: park, count in, and only the last one to arrive goes on ----------------------------
    if (XSP) {
        // (the main-loop stages and the partial accumulators in LDS are dead by now: their first word carries the count;
        //  a static __shared__ variable would push the 64x64 form past the 160 KiB it already fills)
        volatile unsigned& arrived_s = *reinterpret_cast<volatile unsigned*>(smem_all);
        constexpr int TILE_DW = TN * TM * 16 * T; // dwords of one parked tile: [n tile][m tile][16][T threads of group 0]
        unsigned* const counter = static_cast<unsigned*>(p.splitk_ws) + t_lin;
        int* const slots = reinterpret_cast<int*>(static_cast<char*>(p.splitk_ws) + kSplitkWordsBytes) +
                           (size_t)t_lin * XS * TILE_DW;
        if (group == 0) {
            int* const mine = slots + (size_t)xrank * TILE_DW + tid;
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TM; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        __hip_atomic_store(mine + ((i * TM + j) * 16 + e) * T, acc[i][j][e], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every write-through acknowledged
        }
        __syncthreads();
        if (tid_all == 0)
            arrived_s = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (arrived_s != (unsigned)(XS - 1)) return; // not the last one: done (the whole workgroup leaves)
        if (tid_all == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // re-armed
        if (group == 0) {
            for (int o = 1; o < XS; ++o) { // the other XS - 1 parts, in a rotation that depends on nothing but the rank
                const int* const theirs = slots + (size_t)((xrank + o) % XS) * TILE_DW + tid;
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TM; ++j)
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            acc[i][j][e] += __hip_atomic_load(theirs + ((i * TM + j) * 16 + e) * T, __ATOMIC_RELAXED,
                                                              __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }

    // ---- outlier side GEMM operands (unless they were copied at kernel start) ----------------------------------------
    if (has_outliers && !PREO) {
        __syncthreads(); // main-loop LDS (and the partial accumulators) are dead
        stage_outliers();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (has_outliers && PREO && KG == 1) __syncthreads(); // (KG > 1: the split-K hand-over above already synchronised)
    if (KG > 1 && group > 0) return; // (no barrier below this point)

    // ---- epilogue, one 32x32 tile at a time -----------------------------------------------------------
    const int osteps = has_outliers ? (p.O + 15) / 16 : 0;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
#pragma unroll
        for (int j = 0; j < TM; ++j) {
            const int m = m0 + wm * WM + j * 32 + lr;
            const int nb0 = n0 + wn * WN + i * 32 + 4 * lh;
            if (EPI == EPI_INT32) {
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
                continue;
            }
            v16f P;
#pragma unroll
            for (int e = 0; e < 16; ++e) P[e] = 0.f;
            if (has_outliers) {
                const char* xo = osmem + (wn * WN + i * 32 + lr) * OSLICE;
                const char* yo = osmem + BN * OSLICE + (wm * WM + j * 32 + lr) * OSLICE;
                const int sw16 = lr & 15;
                for (int ks = 0; ks < osteps; ++ks) {
                    const int off = ((ks * 2 + lh) ^ sw16) << 4;
                    v8h xf = *reinterpret_cast<const v8h*>(xo + off);
                    v8h yf = *reinterpret_cast<const v8h*>(yo + off);
                    P = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf, yf, P, 0, 0, 0);
                }
            }
            if (m < p.M) {
                const float sa = sa_pre[j];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int nb = nb0 + 8 * g;
                    if (nb < p.N) {
                        const uint2 swb = sw_pre[i][g];
                        const uint16_t swh[4] = {(uint16_t)(swb.x & 0xffffu), (uint16_t)(swb.x >> 16),
                                                 (uint16_t)(swb.y & 0xffffu), (uint16_t)(swb.y >> 16)};
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
                            float v = __builtin_fmaf((float)acc[i][j][4 * g + e], h2f(swh[e]) * sa, c);
                            if (epi_has_silu(EPI)) v = silu_f32(v);
                            oh[e] = f2h_bits_of_f32_result(v);
                        }
                        if (EPI == EPI_DEQUANT_SILU_MUL) { // gate * up: one fp16 multiply of the rounded result
                            const uint2 mb = *reinterpret_cast<const uint2*>(p.Mul + (int64_t)m * p.N + nb);
                            const uint16_t mh[4] = {(uint16_t)(mb.x & 0xffffu), (uint16_t)(mb.x >> 16),
                                                    (uint16_t)(mb.y & 0xffffu), (uint16_t)(mb.y >> 16)};
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
}

static std::atomic<int> g_tile_nt{0}; // 0 by rule (default) | 1 never | 2 whenever one tile row
template <int BM, int BN, int WAVES_M, int WAVES_N, int EPI, int NSTAGE = 2, int KG = 1, bool PREO = false, bool XSP = false,
          bool ADMA = false>
static hipError_t launch_cfg(const GemmParams& p, hipStream_t st)
{
    constexpr int T = WAVES_M * WAVES_N * KG * 64;
    constexpr size_t lds = (size_t)KG * NSTAGE * (size_t)(BM + BN) * KSLICE + (PREO ? (size_t)(BM + BN) * OSLICE : 0);
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = gemm_w8a8o16_kernel<BM, BN, WAVES_M, WAVES_N, EPI, NSTAGE, KG, PREO, XSP, ADMA>;
    static DeviceOnce once;
    if (hipError_t e = ensure_dynamic_lds(kern, lds, once); e != hipSuccess) return e;
    const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
    GemmParams q = p;
    // one tile row: every weight line is read by exactly one workgroup -> non-temporal copies for weights of 32 MiB and more (knob 1290 / 1291 / 1292:
    // by that rule / never / whenever there is one tile row).  Operator us cold at 64 rows, plain -> non-temporal: 12288 x 4096 23.1 -> 22.1, 11008 x 4096
    // 23.0 -> 21.8, 18944 x 3584 28.4 -> 26.9, 3584 x 18944 33.4 -> 32.1; deep form at 97..128 rows -1 %; warm loops over one layer +5..+12 % by construction
    // (profiles/r05_tile_nontemporal.txt)
    const int ntm = g_tile_nt.load(std::memory_order_relaxed);
    if (p.M <= BM && EPI != EPI_INT32 && (ntm == 2 || (ntm == 0 && (int64_t)p.N * p.K >= ((int64_t)32 << 20)))) q.flags |= 2;
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * (XSP ? p.xsplit : 1))), dim3(T), lds, st, q);
    return hipGetLastError();
}

// ---- K split over workgroups for the two small-tile forms with in-workgroup split (see the kernel header) ------------
static std::atomic<int> g_xsplit_force{-1}; // -1 automatic, 0 off, 2 / 4 / 8 / 16 forced where the shape allows it
static std::atomic<int> g_xsplit_max_bm{64}; // tallest tile the plan may take (knob 63 / 65: 32 / 64 rows)
void set_xsplit_force(int v) { g_xsplit_force.store(v); }

// 0 = not used.  Applies to the problems launch_epi gives to the 32x64 / 64x64 tiles with 4 K groups (at most 256 tiles):
// XS = 4 / 2 more workgroups per tile while that still leaves at most one workgroup per CU, and K is long enough for
// each of the 4 x XS parts to keep a few slices.
// The plan: which of the two tilings, and how many workgroups per tile.  The tiling is the first of 32x64 / 64x64 that
// gives at most 128 tiles (so that at least two workgroups per tile fit on the chip at one workgroup per CU).
struct XSplitPlan {
    int xs;    // 0 = not used
    int bm;    // tile rows: 32 or 64 (tile columns: 64)
    int tiles;
};
static XSplitPlan xsplit_plan(int M, int N, int K)
{
    XSplitPlan none{0, 0, 0};
    const int force = g_xsplit_force.load();
    if (force == 0 || M <= 4) return none;
    if (force < 0 && M <= 16 && K < 16384) return none; // M <= 16: the skinny kernel unless K is very long (measured)
    const int64_t n64 = (N + 63) / 64;
    const int nk = (K + KSLICE - 1) / KSLICE;
    if (force < 0 && nk < 64) return none; // automatic: K >= 8192 (measured: at K = 4096 the exchange eats the gain)
    // per tiling: as many workgroups per tile as leave at most one workgroup per CU and 16 slices per workgroup (forced
    // factors: 8 slices); the tiling that puts more workgroups on the chip wins, 64x64 on a tie (measured: equal or up to
    // 26 % faster than 32x64 -- fewer operand bytes per MAC; a 128x64 tiling with 2 K groups was measured too: slower)
    auto factor = [&](int64_t tiles) {
        if (tiles > 128) return 0;
        if (force > 0) return ((int64_t)force * tiles <= 256 && nk >= 8 * force) ? force : 0;
        int xs = 16;
        while (xs > 1 && ((int64_t)xs * tiles > 256 || nk < 16 * xs)) xs >>= 1;
        return xs >= 2 ? xs : 0;
    };
    const int max_bm = g_xsplit_max_bm.load();
    XSplitPlan best = none;
    for (int bm = 32; bm <= max_bm; bm *= 2) {
        if (bm > 32 && M <= bm / 2) break; // a taller tile would only add empty rows
        const int64_t tiles = (int64_t)((M + bm - 1) / bm) * n64;
        const int xs = factor(tiles);
        if (xs != 0 && tiles * xs >= (int64_t)best.tiles * best.xs) best = XSplitPlan{xs, bm, (int)tiles};
    }
    return best;
}

int gemm_xsplit_factor(int M, int N, int K) { return xsplit_plan(M, N, K).xs; }

size_t gemm_xsplit_workspace_size(int M, int N, int K)
{
    const XSplitPlan pl = xsplit_plan(M, N, K);
    if (pl.xs == 0) return 0;
    return kSplitkWordsBytes + (size_t)pl.tiles * pl.xs * (size_t)(pl.bm * 64 * 4);
}

size_t gemm_xsplit_workspace_bound() { return kSplitkWordsBytes + (size_t)256 * 64 * 64 * 4; } // <= 256 workgroups x 16 KiB

// ---- mid-M "deep" form (round 5): 128 x 128 tiles, 4 LDS stages really in flight (ADMA), K split over XS workgroups per tile ----
// 129..512 rows are the band furthest below either roofline: the 256 x 256 / 128 x 256 ping-pong tiles cover a
// quarter to a half of the CUs there, and the small two-barrier tiles run their slices one memory round trip at a time.  This form
// puts (about) one workgroup of a 128 x 128 tile on every CU -- the tiles alone where there are enough of them, 2 / 4 / 8 workgroups
// per tile along K otherwise -- and keeps three slices in flight behind the one being multiplied.  Same int32 sums, same epilogue.
struct DeepPlan;
static std::atomic<int> g_deep_late{1}; // measurement knob 1238 (default) / 1239: the second half of the waves issue their copies after / before their MFMAs
static std::atomic<int> g_deep_force{-1}; // -1 automatic, 0 off, 1 / 2 / 4 / 8: forced workgroups per tile (knobs 1240 / 1241 + xs); + 10: the 8-wave build
struct DeepPlan {
    int xs;    // 0 = not used; workgroups per 128 x 128 tile otherwise
    int tiles;
    int waves8; // build: 0 = 4 waves x 64 x 64, 1 = 8 waves x 64 x 32 (4 stages), 2 = 8 waves, 5 stages, 3 = round 6: copy-only waves + one barrier per pair of slices (gemm_mid_kernels.hip)
};
// The automatic rule: a table over (128 x 128 tiles, K slices), read off a sweep of every build x split against the selection as it
// was, on COLD weights (a model's layer never finds its weights cache-resident), profiles/r05_deep_form_sweep_cold.txt -- 110 cells
// over the ten (N, K) of BASELINE.json's configs, 80..1024 rows.  The form wins 8..35 % exactly where it puts 140..256 workgroups
// on the chip and the alternatives leave CUs idle or take their slices one round trip at a time:
//   tiles 140..256, K 3072..5120        -> the tiles alone            (12288 x 4096 at 129..256 rows: 35.0 -> 28.3 us)  [and > 512 tiles of 64 x 64]
//   tiles  70..128, K 5120..12800       -> 2 workgroups per tile      (4096 x 11008 at 320..512 rows: 42.8 -> 36.1 us)
//   tiles  36..64,  K 8192..20480       -> 4                          (4096 x 11008 at 129..256 rows: 35.4 -> 30.9 us)
//   tiles  20..32,  K >= 25600          -> 8                          (1024 x 28672 at 257..512 rows: 46.1 -> 34.1 us)
// Everywhere else it is level or behind (the ping-pong tiles multiply a slice in less than half the time per CU; warm weights:
// level) and is not used.  8 waves x (64 x 32) beat 4 waves x (64 x 64) in every cell; a fifth LDS stage changed nothing, neither did
// 16 waves x (32 x 32); the same loop on 128 x 256 tiles is behind the ping-pong kernel of that tile (profiles/r05_deep_form_builds.txt).
static std::atomic<int> g_deep_mid{1}; // measurement knob 1420 (default) / 1421: the automatic plan takes the round-6 schedule from 129 rows on / never
static DeepPlan deep_plan_auto(int M, int N, int K, int tiles, int nk)
{
    (void)N;
    const int cus = num_cus();
    if (cus != 256) return DeepPlan{0, 0, 0}; // (the table counts workgroups against the 256 CUs it was measured on)
    // (first row: only where the 64 x 64 tiles are past their forms with K split inside the workgroup -- more than 512 of them;
    //  below that those are ahead: 448 x 4608 x 3584 19.5 vs 23.7 us, 288 x 6144 x 4096 21.6 vs 26.0, validation sweep of the table)
    const int64_t wg64 = (int64_t)((M + 63) / 64) * ((N + 63) / 64);
    // (round 6) from 129 rows on every row of the table runs the round-6 schedule of the same tiles (gemm_mid_kernels.hip; whole 128-byte slices only):
    // cold, all ten BASELINE (N, K), 35 cells: ahead of the round-5 build in 33, by 6-25 % (12288 / 11008 x 4096 at 160..256 rows 27-28 -> 24.4-26.2 us,
    // 3584 x 8192 at 320 / 384 rows 30.5 -> 26 us, 4608 x 3584 at 512 rows 24.7 -> 20.6, 1280 x 8192 at 1024 rows 30.4 -> 24.3, 1024 x 28672 at 320 rows
    // 34.7 -> 31.3); level at 97..128 rows, which keep the round-5 build (profiles/r06_mid_final_sweep_cold.txt)
    const int mid = M > 128 && K % KSLICE == 0 && g_deep_mid.load() ? 3 : 1;
    if (tiles >= 140 && tiles <= 256 && nk >= 24 && nk <= 40 && wg64 > 512) return DeepPlan{1, tiles, mid}; // (measured at K = 3584 .. 5120 only)
    // (97..128 rows on one row of 70..128 tiles at K = 3072..5120: two workgroups per tile put 140..256 workgroups on the chip where the
    //  64 x 64 tiles with two K groups inside the workgroup run 1.5 rounds -- profiles/r05_int8_forms_cold.jsonl: 12288 x 4096 at 128 rows
    //  24.2 -> 22.7 us, 11008 x 4096 24.2 -> 22.0; at 96 rows -1..-5 %, at 72 rows level: from 97 rows)
    if (M > 96 && M <= 128 && tiles >= 70 && tiles <= 128 && nk >= 24 && nk <= 40) return DeepPlan{2, tiles, 1};
    if (M <= 128) return DeepPlan{0, 0, 0};
    // (second row at K < 8192 only past 256 tiles of 64 x 64: one workgroup per CU of those with K split four ways inside it is
    //  ahead below -- 160 / 192 x 5120 x 5120: 20.0 / 20.9 vs 23.1 / 23.5 us, profiles/r05_int8_forms_cold.jsonl)
    if (tiles >= 70 && tiles <= 128 && nk >= 40 && nk <= 100 && (nk >= 64 || wg64 > 256)) return DeepPlan{2, tiles, mid};
    if (tiles >= 36 && tiles <= 64 && nk >= 64 && nk <= 160) return DeepPlan{4, tiles, mid};
    if (M > 256 && tiles >= 20 && tiles <= 32 && nk >= 200) return DeepPlan{8, tiles, mid};
    return DeepPlan{0, 0, 0};
}
static DeepPlan deep_plan(int M, int N, int K)
{
    DeepPlan none{0, 0, 0};
    const int force = g_deep_force.load();
    if (force == 0 || M <= 64) return none;
    const int64_t tiles = (int64_t)((M + 127) / 128) * ((N + 127) / 128);
    const int nk = (K + KSLICE - 1) / KSLICE;
    if (tiles > 4096) return none;
    if (force > 0) {
        const int xs = force % 10;
        if (xs > 1 && tiles * xs > 768) return none; // (the parked tiles must fit the scratch bound: 64 KiB per workgroup)
        return (xs == 1 || xs == 2 || xs == 4 || xs == 8) && nk >= 4 * xs ? DeepPlan{xs, (int)tiles, force / 10} : none;
    }
    return deep_plan_auto(M, N, K, (int)tiles, nk);
}
int gemm_deep_build(int M, int N, int K);
bool gemm_deep_takes(int M, int N, int K, bool have_scratch)
{
    const DeepPlan pl = deep_plan(M, N, K);
    return pl.xs == 1 || (pl.xs > 1 && have_scratch);
}
size_t gemm_deep_workspace_size(int M, int N, int K)
{
    const DeepPlan pl = deep_plan(M, N, K);
    if (pl.xs <= 1) return 0;
    // a parked tile is 128 x 128 or -- the mid kernel's 96-wide tiles, gemm_mid_tile_width -- 128 x 96 int32: room for whichever count x size is larger
    const size_t t96 = (size_t)((M + 127) / 128) * ((N + 95) / 96);
    const size_t b128 = (size_t)pl.tiles * (128 * 128 * 4), b96 = t96 * (128 * 96 * 4);
    return kSplitkWordsBytes + (b128 > b96 ? b128 : b96) * pl.xs;
}
void set_deep_force(int v) { g_deep_force.store(v); }

template <int EPI>
static hipError_t launch_deep_epi(const GemmParams& p, hipStream_t st)
{
    const DeepPlan pl = deep_plan(p.M, p.N, p.K);
    GemmParams q = p;
    q.xsplit = pl.xs;
    q.flags = g_deep_late.load() ? 0 : 1;
    if (pl.waves8 == 3 && p.K % KSLICE == 0) { // round 6: copy-only waves, two slices per barrier (gemm_mid_kernels.hip)
        q.flags = 0;
        return launch_gemm_mid(q, EPI, st);
    }
    if (pl.waves8 == 2) { // five stages = the whole 160 KiB of LDS: four slices in flight
        if (pl.xs > 1) return launch_cfg<128, 128, 2, 4, EPI, 5, 1, false, true, true>(q, st);
        return launch_cfg<128, 128, 2, 4, EPI, 5, 1, false, false, true>(q, st);
    }
    if (pl.waves8) {
        if (pl.xs > 1) return launch_cfg<128, 128, 2, 4, EPI, 4, 1, false, true, true>(q, st);
        return launch_cfg<128, 128, 2, 4, EPI, 4, 1, false, false, true>(q, st);
    }
    if (pl.xs > 1) return launch_cfg<128, 128, 2, 2, EPI, 4, 1, false, true, true>(q, st);
    return launch_cfg<128, 128, 2, 2, EPI, 4, 1, false, false, true>(q, st);
}
hipError_t launch_gemm_deep(const GemmParams& p, int epi, hipStream_t st)
{
    switch (epi) {
    case EPI_DEQUANT: return launch_deep_epi<EPI_DEQUANT>(p, st);
    case EPI_DEQUANT_SILU: return launch_deep_epi<EPI_DEQUANT_SILU>(p, st);
    case EPI_DEQUANT_SILU_MUL: return launch_deep_epi<EPI_DEQUANT_SILU_MUL>(p, st);
    default: return hipErrorInvalidValue;
    }
}

static std::atomic<int> g_force_cfg{-1}; // measurement knob (variant 10 + i): force tile configuration i

template <int EPI>
static hipError_t launch_epi(const GemmParams& p, hipStream_t st)
{
    switch (g_force_cfg.load()) {
    case 0: return launch_cfg<32, 128, 1, 4, EPI>(p, st);
    case 1: return launch_cfg<128, 128, 2, 2, EPI>(p, st);
    case 2: return launch_cfg<256, 256, 2, 4, EPI>(p, st);
    case 3: return launch_cfg<64, 128, 2, 4, EPI>(p, st);
    case 4: return launch_cfg<128, 128, 2, 4, EPI>(p, st);
    case 5: return launch_cfg<128, 64, 4, 2, EPI>(p, st);
    case 6: return launch_cfg<256, 128, 2, 4, EPI>(p, st);
    case 7: return launch_cfg<64, 64, 2, 2, EPI>(p, st);
    case 8: return launch_cfg<64, 256, 2, 4, EPI>(p, st);
    case 9: return launch_cfg<64, 64, 2, 2, EPI, 4>(p, st);
    case 10: return launch_cfg<64, 64, 2, 2, EPI, 8>(p, st);
    case 11: return launch_cfg<64, 128, 2, 4, EPI, 4>(p, st);
    case 12: return launch_cfg<128, 64, 4, 2, EPI, 4>(p, st);
    case 13: return launch_cfg<64, 32, 2, 1, EPI, 8>(p, st);
    case 14: return launch_cfg<128, 128, 2, 4, EPI, 4>(p, st);
    case 15: return launch_cfg<32, 128, 1, 4, EPI, 4>(p, st);
    case 16: return launch_cfg<64, 64, 2, 2, EPI, 2, 4>(p, st);
    case 17: return launch_cfg<64, 64, 2, 2, EPI, 2, 2>(p, st);
    case 18: return launch_cfg<32, 64, 1, 2, EPI, 2, 4>(p, st);
    case 19: return launch_cfg<128, 64, 4, 2, EPI, 2, 2>(p, st);
    case 20: return launch_cfg<64, 32, 2, 1, EPI, 2, 4>(p, st);
    case 21: return launch_cfg<64, 128, 2, 4, EPI, 2, 2>(p, st);
    case 22: return launch_cfg<32, 64, 1, 2, EPI, 2, 4, true>(p, st);
    case 23: return launch_cfg<64, 64, 2, 2, EPI, 2, 4, true>(p, st);
    default: break;
    }
    // Measured choice (tools/cfg_sweep.sh, M = 32..1024 on 12288x4096, 4096x11008, 4096x4096): small problems are
    // bound by how many workgroups stream operands concurrently, so the 64x64 tile wins until ~3 workgroups per CU.
    // Rule distilled from the sweep: keep ~16 wavefronts on every CU.  With few tiles, several groups of waves share one
    // tile and split K among themselves (KG = 4, then 2); with ~3 tiles per CU plain 64x64 tiles; beyond that 128x128.
    const int64_t n64 = (p.N + 63) / 64;
    const int64_t wg32 = (int64_t)((p.M + 31) / 32) * n64, wg64 = (int64_t)((p.M + 63) / 64) * n64;
    if (EPI != EPI_INT32 && p.splitk_ws != nullptr) { // few tiles: K split over 2 / 4 workgroups per tile as well
        const XSplitPlan pl = xsplit_plan(p.M, p.N, p.K);
        if (pl.xs > 1) {
            GemmParams q = p;
            q.xsplit = pl.xs;
            return pl.bm == 32 ? launch_cfg<32, 64, 1, 2, EPI, 2, 4, true, true>(q, st)
                               : launch_cfg<64, 64, 2, 2, EPI, 2, 4, true, true>(q, st);
        }
    }
    if (wg32 <= 256) return launch_cfg<32, 64, 1, 2, EPI, 2, 4, true>(p, st);
    if (wg64 <= 256) return launch_cfg<64, 64, 2, 2, EPI, 2, 4, true>(p, st);
    if (wg64 <= 512) return launch_cfg<64, 64, 2, 2, EPI, 2, 2>(p, st);
    if (wg64 <= 768) return launch_cfg<64, 64, 2, 2, EPI>(p, st);
    return launch_cfg<128, 128, 2, 4, EPI>(p, st);
}

// Schedule selection.  0 = auto (ping-pong 256x256 kernel when the problem fills the chip with 256x256 tiles, else the
// 2-barrier kernel in a smaller tile), 1 = always the 2-barrier kernel, 2 = ping-pong whenever M > 4.
// (The persistent ping-pong experiment of round 1 -- measured to lose -- is in the history: tools/experimental/README.md.)
// Set through mixq_debug_set_gemm_variant() (tests, A/B measurements) or MIXQ_GEMM_VARIANT=v1|pp in the environment.
static std::atomic<int> g_variant{-1};
static std::atomic<int> g_skinny_wide{0};
static std::atomic<int> g_qa_frag{1};
bool qa_frag_enabled() { return g_qa_frag.load() != 0; }

void set_gemm_variant(int v)
{
    if (v == 1238 || v == 1239) {
        g_deep_late.store(v == 1238);
        return;
    }
    if (v == 1420 || v == 1421) { // the automatic deep plan: round-6 schedule from 129 rows on (1420, default) / the round-5 build everywhere (1421)
        g_deep_mid.store(v == 1420);
        return;
    }
    if (v >= 1430 && v <= 1433) { // round-6 mid kernel: tile width by rule (1430, default: 96 where that puts more workgroups on the chip) / 128 always (1431)
        set_mid_bn(v - 1430);
        return;
    }
    if (v >= 1410 && v <= 1413) { // round-6 mid kernel: tile rows start at different K slices: 1413 by the measured rule (default: 4..7 tile rows, K not split) / 1411 never / 1410 where K is not split over workgroups / 1412 always
        set_mid_rot(v == 1410 ? 1 : v == 1411 ? 0 : v == 1412 ? 2 : 3);
        return;
    }
    if (v >= 1240 && v <= 1279) { // mid-M deep form: 1240 automatic, 1241 off, 1241 + xs (1242 / 1243 / 1245 / 1249) forced, + 10 the 8-wave build, + 30 the round-6 schedule
        set_deep_force(v == 1240 ? -1 : v - 1241);
        return;
    }
    if (v >= 1300 && v <= 1312) { // quantisers: rows up to which one 256-thread block takes one row: 1300 the measured rules (default), 1301 + n: 64 << n rows
        set_quant_block_rows(v == 1300 ? -1 : 64 << (v - 1301));
        return;
    }
    if (v >= 1290 && v <= 1292) { // tile kernels, non-temporal weight copies on a single tile row: 1290 by rule, 1291 never, 1292 always
        g_tile_nt.store(v - 1290);
        return;
    }
    if (v >= 884 && v <= 887) { // decode-batch GEMM, row-major weights in 256-byte runs through LDS: 884 on, non-temporal from 32 MiB (default), 885 off, 886 / 887 on with non-temporal loads always / never
        set_skinny_wrows(v == 884 ? 1 : v == 885 ? 0 : v == 886 ? 2 : 3);
        return;
    }
    if (v >= 880 && v <= 883) { // registered weight images in the decode-batch GEMM: 880 automatic, 881 plain loads, 882 non-temporal loads, 883 ignored
        set_skinny_wfrag(v - 880);
        return;
    }
    if (v >= 894 && v <= 896) { // fragment-major skinny form: feature tiles per workgroup automatic / 1 / 2
        set_skinny_nt(v - 894);
        return;
    }
    if (v == 892 || v == 893 || v == 897) { // probe: the skinny kernel for every N up to 32 rows (892) / up to 64 rows (897) / the
        g_skinny_wide.store(v == 892 ? 1 : v == 897 ? 2 : 0); // measured rule (893, default)
        return;
    }
    if (v == 890 || v == 891) { // fragment-major qA for mixq_enqueue's decode batches: 890 off (row-major), 891 on (default)
        g_qa_frag.store(v == 891 ? 1 : 0);
        return;
    }
    if (v >= 92 && v <= 99) { // skinny-kernel ablations (measurement only, wrong results): 92 + ABL - 1
        set_skinny_kw(21 + (v - 92));
        g_force_cfg.store(-1);
        g_variant.store(0);
        return;
    }
    if (v == 859 || v == 8590) { // fpA_intB skinny form, 16 columns per wave x 8 / 16 waves (measurements, up to 16 tokens)
        set_wo_force(v == 859 ? 309 : 310, -2);
        return;
    }
    if (v >= 856 && v <= 858) { // decode batches (2..4 tokens) through the fpA_intB skinny form: 856 always, 857 never, 858 automatic
        set_wo_force(306 + (v - 856), -2);
        return;
    }
    if (v >= 850 && v <= 855) { // fpA_intB skinny form (5..32 tokens): 850 automatic, 851 off, 852..855 a fixed shape
        set_wo_force(300 + (v - 850), -2);
        return;
    }
    if (v == 848 || v == 849) { // fpA_intB skinny form, non-temporal loads of weights >= 32 MiB: 848 on (default), 849 off
        set_wo_force(v == 848 ? 500 : 501, -2);
        return;
    }
    if (v >= 845 && v <= 847) { // fpA_intB wide form, weights through registers instead of LDS: 845 automatic, 846 never, 847 always
        set_wo_force(400 + (v - 845), -2);
        return;
    }
    if (v == 843 || v == 844) { // second pass of the two-pass form on 256- / 128-row tiles
        set_wo_force(v == 843 ? 203 : 204, -2);
        return;
    }
    if (v >= 840 && v <= 842) { // fpA_intB two-pass form: 840 automatic (from 1280 tokens), 841 never, 842 whenever the shape allows
        set_wo_force(200 + (v - 840), -2);
        return;
    }
    if (v >= 831 && v <= 836) { // fpA_intB wide-form configuration 1..4 (w8a16_gemm_kernels.hip kWoCfg)
        set_wo_force(v - 830, -2);
        return;
    }
    if (v >= 800 && v < 816) { // fpA_intB wide-form ablations (measurement only, wrong results): 800 + ABL
        set_wo_force(100 + (v - 800), -2);
        return;
    }
    if (v >= 80 && v <= 89) { // fpA_intB GEMM (w8a16_gemm_kernels.hip): 80 automatic, 81 narrow passes, 82 / 84 wide form with
                              // 128- / 256-row tiles; 85 K split automatic, 86..89: 1 / 2 / 4 / 8 workgroups per tile
        if (v <= 84) set_wo_force(v == 80 ? -1 : v == 81 ? 0 : v == 82 ? 3 : 4, -2);
        else set_wo_force(-2, v == 85 ? -1 : 1 << (v - 86));
        return;
    }
    if (v == 90 || v == 91) { // 90: split workgroups never wait for their partners (all but the last arriver defer), 91: default
        set_splitk_patience(v == 90 ? 0u : 3000u);
        return;
    }
    if (v >= 70 && v <= 79) { // K split over workgroups (ping-pong kernel): 70 off, 72 / 74 / 78 forced factor, 79 automatic
        set_splitk_force(v == 79 ? -1 : v - 70);
        if (v == 70 || v == 79) set_xsplit_force(v == 79 ? -1 : 0);
        return;
    }
    if (v >= 60 && v <= 69) { // the same for the small-tile kernels: 60 off, 62 / 64 / 68 / 66 forced 2 / 4 / 8 / 16, 69 auto
        if (v == 63 || v == 65) { // (tallest tile of the plan: 32 / 64 rows)
            g_xsplit_max_bm.store(v == 63 ? 32 : 64);
            return;
        }
        set_xsplit_force(v == 69 ? -1 : v == 66 ? 16 : v - 60);
        return;
    }
    if (v >= 40 && v < 100) { // 40 + kw: skinny kernel with kw K-split waves (measurements)
        set_skinny_kw(v - 40);
        g_force_cfg.store(-1);
        g_variant.store(0);
        return;
    }
    set_skinny_kw(0);
    if (v < 0) { // (mixq_debug_reset) forget a forced schedule: MIXQ_GEMM_VARIANT in the environment is read again
        g_force_cfg.store(-1);
        g_variant.store(-1);
        return;
    }
    if (v >= 10 && v < 100) { // 10 + i: the 2-barrier kernel in tile configuration i (measurements)
        g_force_cfg.store(v - 10);
        g_variant.store(1);
        return;
    }
    g_force_cfg.store(-1);
    g_variant.store(v);
}

static int gemm_variant()
{
    int v = g_variant.load();
    if (v < 0) {
        const char* e = getenv("MIXQ_GEMM_VARIANT");
        v = (e && e[0] == 'v' && e[1] == '1') ? 1 : (e && e[0] == 'p' && e[1] == 'p') ? 2 : 0;
        g_variant.store(v);
    }
    return v;
}

// 128 x 256 ping-pong tiles (gemm_pp128_kernels.hip) win where they fill most of ONE wave of the chip and K is short enough
// that the 256 x 256 form's K split over workgroups would not amortise its exchange: measured on 90 mid-size shapes
// (tools/pp128_sweep.sh, profiles/r02_pp128_sweep.txt): 88..256 tiles at K < 8192 (-3..-21 %), 160..256 tiles at
// 8192 <= K < 10240 (-4..-5 %); more than one wave of tiles, fewer than ~88 tiles or longer K: the other forms.
bool gemm_pp128_wins(int M, int N, int K)
{
    if (M <= 128) return false;
    const int64_t tm = (M + 127) / 128, t = tm * ((N + 255) / 256);
    const int nk = (K + 127) / 128, cus = num_cus();
    if (t > cus) return false;
    // tiles weighted by the share of their rows that exist (round 4, tools/midm_cfg_sweep.py in steady state: 192 x 12288 x 4096 = 96
    // tiles three quarters full ran 26.2 us here against 23.8 us on plain 64 x 64 tiles; every other mid-M cell within 2 % of its best)
    const int64_t t_rows = tm <= 2 ? t * M : t * 128, full = tm <= 2 ? tm * 128 : 128; // (measured for one / two tile rows only)
    if (nk < 64) return 256 * t_rows >= (int64_t)88 * cus * full;
    if (nk < 80) return 256 * t_rows >= (int64_t)160 * cus * full;
    return false;
}

static std::atomic<const char*> g_last_kernel{"none"}; // reporting only (bench.py's roofline.kernel)
const char* last_gemm_kernel() { return g_last_kernel.load(std::memory_order_relaxed); }
void note_gemm_kernel(const char* name) { g_last_kernel.store(name, std::memory_order_relaxed); }

// The peer-write epilogue exists for the plain 256 x 256 ping-pong kernel: exactly the shapes for which launch_gemm below
// (automatic selection, no scratch-dependent form) ends up there.
bool gemm_tp_fused_supported(int M, int N, int K, int O)
{
    if (M <= 128 || N <= 0 || K <= 0 || O < 0 || O > 128 || gemm_variant() != 0) return false;
    if (gemm_deep_takes(M, N, K, true) || gemm_pp128_wins(M, N, K) || gemm_splitk_factor(M, N, K) != 0) return false;
    const int64_t tiles256 = (int64_t)((M + 255) / 256) * ((N + 255) / 256);
    const int64_t wg64 = (int64_t)((M + 63) / 64) * ((N + 63) / 64);
    return tiles256 >= 96 && wg64 > 768;
}

bool gemm_takes_skinny(const GemmParams& p, int epi)
{
    const bool frag = p.a_frag == 1; // the caller holds (or, probing, could produce) the fragment-major qA image
    // long K: the small tiles with K split over workgroups -- unless the image is there, N / 16 workgroups fill the chip and K is
    // not longer than 12288 (operator us, K split vs skinny on the image: 4096 x 11008 at 32 / 48 rows 20.9 / 21.4 vs 17.6 / 20.0,
    // 8192 x 8192 at 32 22.4 vs 19.9; 3584 x 18944 25.6 vs 36.9 and 1024 x 28672 24.2 vs 50.0 stay with the K split)
    // (round 4, re-fitted on COLD weights -- a model's decode step never finds a layer's weights cache-resident --
    //  profiles/r04_decode_batch_longk_probe.txt, operator us, small-tile K split vs skinny on the qA image, 32 rows: 1280 x 8192 16.8 / 15.1,
    //  2048 x 8192 17.0 / 15.1, 3584 x 8192 18.6 / 16.2, 1024 x 8192 16.3 / 14.4; beyond K = 8192 the K split stays (5120 x 13824 32.9 /
    //  35.0, 8192 x 16384 41.2 / 47.5, 3584 x 18944 29.8 / 39.7, 1024 x 28672 27.9 / 52.1) -- UNLESS the weight has a registered image
    //  (gemm_skinny_kernels.hip), which the skinny kernel streams 15-20 % faster: 5120 x 13824 27.8, 8192 x 16384 37.2, 4096 x 16384 29.2
    //  -> 24.1, 2560 x 12288 21.6 -> 18.2, 512 x 8192 15.8 -> 12.5; at 48 rows 3584 x 8192 19.7 -> 17.0, 1280 x 8192 16.7 -> 16.0.)
    const bool img = frag && p.b_image != nullptr;
    const int c16 = 16 * num_cus();
    const bool skinny_on_image = frag && ((p.N >= c16 && p.K <= 12288) || (p.M <= 32 && p.N >= 1024 && p.K <= 8192) ||
                                          (img && p.M <= 32 && p.N >= 512 && p.K <= 16384) ||
                                          (img && p.M <= 48 && p.N >= 1024 && p.N <= c16 && p.K <= 12288));
    const bool xsplit_wins = epi != EPI_INT32 && p.splitk_ws != nullptr && p.K >= 8192 && gemm_xsplit_factor(p.M, p.N, p.K) != 0 &&
                             !skinny_on_image;
    // 33..64 rows (fragment-major qA image only), round 4 on COLD weights (profiles/r04_decode_batch_k4096_probe.txt, operator us, tiles vs
    // skinny / skinny on a registered weight image): 12288 x 4096 at 48 rows 22.7 vs 25.1 / 21.7, 11008 x 4096 22.4 / 24.4 / 20.6,
    // 8192 x 4096 19.1 / 18.6 / 16.0, 6144 x 4096 at 64 rows 19.1 / 20.7 / 18.5, 5120 x 5120 (32 features per workgroup) at 48 rows
    // 22.0 / 23.5 / 19.7, at 64 rows 22.2 / 26.0 / 23.0, 4096 x 4096 at 64 rows 17.9 / 14.4 / 13.2.  (Round 3 had fitted this warm:
    // 48 rows up to N = 12288, 64 rows up to 6144 whatever the weight's layout.)
    const bool one_tile = skinny_feature_tiles(p.M, p.N, p.K, false) == 1 || skinny_feature_tiles(p.M, p.N, p.K, true) == 1;
    // (round 5) with the row-major weight read in 256-byte runs (K % 256 == 0: then one_tile holds without an image) the skinny kernel keeps
    // 33..48 rows up to N = 12288 at K <= 5120 (selection check, operator us, tiles vs skinny: 48 x 12288 x 4096 19.2 / 17.4 warm, 22.4 / 21.6 cold)
    const bool runs = p.K % 256 == 0 && p.K <= 5120 && skinny_feature_tiles(p.M, p.N, p.K, false) == 1;
    // (the rule with an image is a SUPERSET of the rule without: a producer that probed without the weight pointer -- mixq_qa_layout --
    //  and wrote the fragment-major image must find the consumer agreeing whatever the registry holds)
    // (two feature tiles per workgroup on the 256-byte-run route, N = 5120..8192, K <= 8192: ahead of the tiles up to 64 rows -- gemm_skinny_kernels.hip)
    const bool runs2 = frag && p.K % 256 == 0 && p.K <= 8192 && skinny_feature_tiles(p.M, p.N, p.K, false) == 2 && p.N >= 5120 && p.N <= 8192;
    // (17..32 rows beyond N = 12288: with two feature tiles per workgroup on the run route the skinny kernel stays ahead of the tiles up to N ~ 20000 at
    //  K <= 5120 -- 18944 x 3584 at 32 rows 25.1 -> 23.0 / 23.6 -> 21.3 us on two boxes)
    const bool wide2 = frag && p.K % 256 == 0 && p.K <= 5120 && p.N > 12288 && p.N <= 20480 && skinny_feature_tiles(p.M, p.N, p.K, false) == 2;
    const bool rows_plain = (p.M <= 48 && p.N <= (runs ? 12288 : 8192) && one_tile) || (p.M <= 64 && p.N <= 4096) || (runs2 && p.M <= 64);
    const bool rows_33_64 = frag && (rows_plain || (img && ((p.M <= 48 && p.N <= 12288) || (p.M <= 64 && p.N <= 6144 && one_tile))));
    return gemm_variant() != 1 && !xsplit_wins && gemm_skinny_supported(p) &&
           (p.M <= 16 || (p.M <= 32 && (p.N <= 12288 || wide2 || g_skinny_wide.load() != 0)) || rows_33_64 ||
            (p.M <= 64 && g_skinny_wide.load() == 2));
    // (33..64 rows, fragment-major image only, operator us, tiles vs skinny: 4096 x 4096 at 40 / 48 / 64 rows 13.4 / 13.5 / 13.5 vs
    //  10.3 / 10.3 / 11.9; 3584 x 3584 at 64 12.7 / 11.1; 8192 x 4096 at 48 15.9 / 14.5; 12288 x 4096 at 48 19.2 / 18.7, at 64 19.9 / 22.4)
    // (17..32 rows, round 3 with the fragment-major qA image, operator us, tiles vs skinny: N = 8192 15.9 / 12.8, 11008 17.6 / 16.0,
    //  12288 18.8 / 16.0, 18944 x 3584 19.8 / 22.4, 28672 x 8192 41.5 / 49.6: the two-barrier tiles from more than 768 workgroups on.
    //  Callers that pass a row-major image take the same rule: 12288 x 4096 at 32 rows 14.8 skinny vs 15.3 tiles in round 2.)
}

hipError_t launch_gemm(const GemmParams& p, int epi, hipStream_t st)
{
    if (p.M <= 0 || p.N <= 0) return hipSuccess;
    if (p.a_frag == 1 && !gemm_takes_skinny(p, epi)) return hipErrorInvalidValue; // (only the skinny kernel reads that image)
    if (p.a_frag == 2 && !(epi == EPI_DEQUANT && gemm_tp_fused_supported(p.M, p.N, p.K, p.O))) return hipErrorInvalidValue;
    if (p.a_frag < 0 || p.a_frag > 2) return hipErrorInvalidValue;
    const int variant = gemm_variant();
    auto chose = [](const char* name) { g_last_kernel.store(name, std::memory_order_relaxed); };
    // M <= 16, and M <= 32 on narrow outputs: the weight-streaming GEMV-like kernel (measured against the split-K tiles)
    // ... unless K is long and the tiles are few: then the small tiles with K split over workgroups win (M = 24 / 32 on
    // 4096 x 11008: 14 vs 18 us; on 1024 x 28672: 17 vs 39 us)
    if (gemm_takes_skinny(p, epi)) {
        chose("gemm_skinny_kernel");
        return launch_gemm_skinny(p, epi, st);
    }
    if (variant == 0 && epi != EPI_INT32 && p.a_frag == 0 && gemm_deep_takes(p.M, p.N, p.K, p.splitk_ws != nullptr)) {
        if (gemm_deep_build(p.M, p.N, p.K) == 3) chose("gemm_w8a8o16_mid_kernel<DEEP> (128x128 / 128x96 tiles, copy-only waves, one barrier per pair of slices, K split over workgroups)");
        else chose("gemm_w8a8o16_kernel<DEEP> (128x128 tiles, 4 stages in flight, K split over workgroups)");
        return launch_gemm_deep(p, epi, st);
    }
    const int64_t tiles256 = (int64_t)((p.M + 255) / 256) * ((p.N + 255) / 256);
    if (variant >= 100) return launch_gemm_pp_ablate(p, variant - 100, st); // measurement-only ablations
    const int64_t wg64 = (int64_t)((p.M + 63) / 64) * ((p.N + 63) / 64);
    if (variant == 0 && gemm_pp128_wins(p.M, p.N, p.K)) {
        chose("gemm_w8a8o16_pp128_kernel (128x256 ping-pong)");
        return launch_gemm_pp128(p, epi, st);
    }
    if (variant == 0 && epi != EPI_INT32 && p.splitk_ws != nullptr && gemm_splitk_factor(p.M, p.N, p.K) != 0) {
        chose("gemm_w8a8o16_pp_kernel<SPLITK> (256x256 ping-pong, K split over workgroups)");
        return launch_gemm_pp_splitk(p, epi, st); // 2 / 4 / 8 workgroups per tile
    }
    if (variant == 5) { // (measurement / tests) the 128 x 256 ping-pong kernel for every M > 4
        chose("gemm_w8a8o16_pp128_kernel (128x256 ping-pong)");
        return launch_gemm_pp128(p, epi, st);
    }
    if (variant == 2 || (variant == 0 && p.M > 128 && tiles256 >= 96 && wg64 > 768)) {
        chose("gemm_w8a8o16_pp_kernel (256x256 ping-pong)");
        return launch_gemm_pp(p, epi, st);
    }
    chose("gemm_w8a8o16_kernel (two-barrier tiles)");
    switch (epi) {
    case EPI_DEQUANT: return launch_epi<EPI_DEQUANT>(p, st);
    case EPI_DEQUANT_SILU: return launch_epi<EPI_DEQUANT_SILU>(p, st);
    case EPI_DEQUANT_SILU_MUL: return launch_epi<EPI_DEQUANT_SILU_MUL>(p, st);
    default: return launch_epi<EPI_INT32>(p, st);
    }
}

int gemm_deep_build(int M, int N, int K) { return deep_plan(M, N, K).waves8; }

int gemm_deep_factor(int M, int N, int K, bool have_scratch)
{
    const DeepPlan pl = deep_plan(M, N, K);
    return (pl.xs == 1 || (pl.xs > 1 && have_scratch)) ? pl.xs : 0;
}

// What launch_gemm WOULD launch for this problem, as text -- the same decisions in the same order, nothing is launched (host only: it
// runs without a GPU, where num_cus() answers 256).  tests/test_gpu_selection.py launches a grid of shapes and compares the family with
// last_gemm_kernel(), so the two cannot drift apart; tests/golden/selection_table.json pins the answers on BASELINE.json's shapes, so a
// change of any rule shows up as a diff of that table (the rules are tables fitted on measurements -- this makes
// them readable and reviewable as one).
void describe_gemm_plan(const GemmParams& p, int epi, char* buf, size_t len)
{
    if (len == 0) return;
    const int variant = gemm_variant();
    const bool scratch = p.splitk_ws != nullptr;
    if (gemm_takes_skinny(p, epi)) {
        const bool image = p.a_frag == 1 && p.b_image != nullptr;
        const int route = skinny_weight_route(p.a_frag, image, p.M, p.N, p.K);
        const int nt = p.a_frag == 1 && ((p.M + 15) / 16 <= 2 || route == 3) ? skinny_feature_tiles(p.M, p.N, p.K, route == 1 || route == 2) : 1;
        snprintf(buf, len, "skinny: %d features per workgroup, weight %s, qA %s", 16 * nt,
                 route == 3 ? "row-major in 256-byte runs" : route == 0 ? "row-major in 64-byte pieces" : "from its registered image",
                 p.a_frag == 1 ? "fragment-major" : "row-major");
        return;
    }
    if (variant == 0 && epi != EPI_INT32 && p.a_frag == 0 && gemm_deep_takes(p.M, p.N, p.K, scratch)) {
        const int xs = gemm_deep_factor(p.M, p.N, p.K, scratch);
        const bool midk = gemm_deep_build(p.M, p.N, p.K) == 3;
        snprintf(buf, len, "deep: 128x%d tiles, %d workgroup(s) per tile along K%s", midk ? gemm_mid_tile_width(p.M, p.N, xs) : 128, xs,
                 midk ? ", copy-only waves" : "");
        return;
    }
    const int64_t tiles256 = (int64_t)((p.M + 255) / 256) * ((p.N + 255) / 256);
    const int64_t n64 = (p.N + 63) / 64;
    const int64_t wg32 = (int64_t)((p.M + 31) / 32) * n64, wg64 = (int64_t)((p.M + 63) / 64) * n64;
    if (variant == 0 && gemm_pp128_wins(p.M, p.N, p.K)) {
        snprintf(buf, len, "ping-pong 128x256 tiles");
        return;
    }
    if (variant == 0 && epi != EPI_INT32 && scratch && gemm_splitk_factor(p.M, p.N, p.K) != 0) {
        const SplitPlan pl = gemm_splitk_plan(p.M, p.N, p.K);
        if (pl.solo > 0) snprintf(buf, len, "ping-pong 256x256 tiles: %d whole round(s) solo + tail split %d ways along K", pl.solo / num_cus(), pl.s);
        else snprintf(buf, len, "ping-pong 256x256 tiles, %d workgroups per tile along K", pl.s);
        return;
    }
    if (variant == 2 || (variant == 0 && p.M > 128 && tiles256 >= 96 && wg64 > 768)) {
        snprintf(buf, len, "ping-pong 256x256 tiles");
        return;
    }
    if (epi != EPI_INT32 && scratch) {
        const XSplitPlan pl = xsplit_plan(p.M, p.N, p.K);
        if (pl.xs > 1) {
            snprintf(buf, len, "two-barrier %dx64 tiles, 4 K groups, %d workgroups per tile along K", pl.bm, pl.xs);
            return;
        }
    }
    if (wg32 <= 256) snprintf(buf, len, "two-barrier 32x64 tiles, 4 K groups");
    else if (wg64 <= 256) snprintf(buf, len, "two-barrier 64x64 tiles, 4 K groups");
    else if (wg64 <= 512) snprintf(buf, len, "two-barrier 64x64 tiles, 2 K groups");
    else if (wg64 <= 768) snprintf(buf, len, "two-barrier 64x64 tiles");
    else snprintf(buf, len, "two-barrier 128x128 tiles");
}

// ---------------------------------------------------------------------------------------------------------
// Stand-alone pieces of the reference API surface that enqueue() no longer needs (kept for drop-in parity).

// gemmfp16 (TsinghuaMixQPlugin.cpp:122-161): Out[M,N] = fp16(fpA[M,O] . fpW[N,O]^T), fp32 accumulate.
// One wave per 32x32 output tile; operands straight from global (O is tiny, both panels stay in L2).
__global__ __launch_bounds__(256) void gemm_fp16_kernel(const uint16_t* __restrict__ fpA,
                                                         const uint16_t* __restrict__ fpW, uint16_t* __restrict__ Out,
                                                         int M, int N, int O)
{
    const int lane = threadIdx.x & 63, lr = lane & 31, lh = lane >> 5;
    const int tiles_n = (N + 31) / 32;
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int tm = (int)(tile / tiles_n), tn = (int)(tile % tiles_n);
    if ((int64_t)tm * 32 >= M) return;
    const int mrow = min(tm * 32 + lr, M - 1), nrow = min(tn * 32 + lr, N - 1);
    v16f P;
#pragma unroll
    for (int e = 0; e < 16; ++e) P[e] = 0.f;
    for (int k0 = 0; k0 < O; k0 += 16) {
        const int k = k0 + lh * 8;
        v8h xf, yf;
        if ((O & 7) == 0) { // rows are 16-byte aligned: vector loads (k + 8 <= O always holds for k < O)
            if (k < O) {
                xf = *reinterpret_cast<const v8h*>(fpW + (int64_t)nrow * O + k);
                yf = *reinterpret_cast<const v8h*>(fpA + (int64_t)mrow * O + k);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) xf[e] = (_Float16)0.f, yf[e] = (_Float16)0.f;
            }
        } else { // any O (dynamic outlier sets of the P-flavour): element loads with a zero tail
            uint16_t xb[8], yb[8]; // (arrays, not vector-element lvalues: see mixq_device.h on __builtin_bit_cast)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const bool ok = k + e < O;
                xb[e] = ok ? fpW[(int64_t)nrow * O + k + e] : (uint16_t)0;
                yb[e] = ok ? fpA[(int64_t)mrow * O + k + e] : (uint16_t)0;
            }
            __builtin_memcpy(&xf, xb, 16);
            __builtin_memcpy(&yf, yb, 16);
        }
        P = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf, yf, P, 0, 0, 0);
    }
    const int m = tm * 32 + lr;
    if (m >= M) return;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int nb = tn * 32 + 8 * g + 4 * lh;
        if (nb < N) {
            uint2 o;
            o.x = (unsigned)f2h_bits(P[4 * g]) | ((unsigned)f2h_bits(P[4 * g + 1]) << 16);
            o.y = (unsigned)f2h_bits(P[4 * g + 2]) | ((unsigned)f2h_bits(P[4 * g + 3]) << 16);
            *reinterpret_cast<uint2*>(Out + (int64_t)m * N + nb) = o;
        }
    }
}

hipError_t launch_gemm_fp16(const void* fpA, const void* fpW, void* Out, int M, int N, int O, hipStream_t st)
{
    if (M <= 0 || N <= 0) return hipSuccess;
    const int64_t tiles = (int64_t)((M + 31) / 32) * ((N + 31) / 32);
    hipLaunchKernelGGL(gemm_fp16_kernel, dim3((unsigned)((tiles + 3) / 4)), dim3(256), 0, st,
                       static_cast<const uint16_t*>(fpA), static_cast<const uint16_t*>(fpW),
                       static_cast<uint16_t*>(Out), M, N, O);
    return hipGetLastError();
}

// dequantizationKernel (kernel/i8gemm.cu:258-279): out = hadd( fp16((float(x)*sRow[m])*sCol[n]), out ).
__global__ __launch_bounds__(256) void dequantization_kernel(uint16_t* __restrict__ out, const int32_t* __restrict__ x,
                                                              const uint16_t* __restrict__ sRow,
                                                              const uint16_t* __restrict__ sCol, int M, int N)
{
    const int64_t total = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / N), n = (int)(i % N);
        const float t = ((float)x[i] * h2f(sRow[m])) * h2f(sCol[n]);
        out[i] = f2h_bits_of_f32_result(h2f(f2h_bits_of_f32_result(t)) + h2f(out[i]));
    }
}

hipError_t launch_dequantization(void* out, const int32_t* x, const void* sRow, const void* sCol, int M, int N,
                                 hipStream_t st)
{
    if (M <= 0 || N <= 0) return hipSuccess;
    const int64_t total = (int64_t)M * N;
    const int64_t want = (total + 255) / 256;
    const unsigned grid = (unsigned)(want < 2048 ? want : 2048);
    hipLaunchKernelGGL(dequantization_kernel, dim3(grid), dim3(256), 0, st, static_cast<uint16_t*>(out), x,
                       static_cast<const uint16_t*>(sRow), static_cast<const uint16_t*>(sCol), M, N);
    return hipGetLastError();
}

// dequantizationKernelSilu (quantkernel/mix_cuda/cult.cu:2305-2324, called by dequantizeInt8Silu :2341-2348 on the
// P-flavour's sm90 route, MixQ/src/mixquant/modules/linear.py:321-324):
//   out = fp16( silu( fma(fl32(float(x) * sRow[m]), sCol[n], float(y)) ) )   -- fp32 throughout, ONE rounding to fp16
// (unlike dequantizationKernel above, which rounds the product to fp16 before an fp16 add).
__global__ __launch_bounds__(256) void dequantization_silu_kernel(uint16_t* __restrict__ out,
                                                                   const int32_t* __restrict__ x,
                                                                   const uint16_t* __restrict__ sRow,
                                                                   const uint16_t* __restrict__ sCol,
                                                                   const uint16_t* __restrict__ y, int M, int N)
{
    const int64_t total = (int64_t)M * N;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int m = (int)(i / N), n = (int)(i % N);
        // nvcc (default -fmad=true) contracts the LAST multiply with the add: fma(fl32(x * sRow), sCol, y) -- the first
        // product is rounded on its own (pinned: hipcc must not fold it into the fma), exactly as the oracle restates it
        float t = (float)x[i] * h2f(sRow[m]);
        asm("" : "+v"(t));
        out[i] = f2h_bits_of_f32_result(silu_f32(__builtin_fmaf(t, h2f(sCol[n]), h2f(y[i]))));
    }
}

hipError_t launch_dequantization_silu(void* out, const int32_t* x, const void* sRow, const void* sCol, const void* y,
                                      int M, int N, hipStream_t st)
{
    if (M <= 0 || N <= 0) return hipSuccess;
    const int64_t want = ((int64_t)M * N + 255) / 256;
    const unsigned grid = (unsigned)(want < 2048 ? want : 2048);
    hipLaunchKernelGGL(dequantization_silu_kernel, dim3(grid), dim3(256), 0, st, static_cast<uint16_t*>(out), x,
                       static_cast<const uint16_t*>(sRow), static_cast<const uint16_t*>(sCol),
                       static_cast<const uint16_t*>(y), M, N);
    return hipGetLastError();
}

} // namespace mixq

// fpA_intB GEMM for gfx950 (M > 4): Out[m,n] = sum_k A[m,k] * fp16((q[k,n]-128) * scale[n]), fp16 activations x int8
// weights, weights dequantised IN REGISTERS on their way into the fp16 matrix cores, fp32 accumulation.
//
// Replaces (reference, CUDA): weightonlykernel/fpA_intB_gemm_wrapper.cu:45-70 (m > SMALL_M_FAST_PATH ->
// ft::gemm_fp16_int) -> cutlass_kernels/fpA_intB_gemm/fpA_intB_gemm_template.h:441-552 (CUTLASS mixed-input GEMM:
// FastInterleavedAndBiasedNumericArrayConverter + per-column scale in the main loop, tensor-core MMA, fp32 accumulate);
// callers: MixQ/src/mixquant/modules/linear.py:175-181 (weight_only mode), EETQ w8_a16_gemm.
//
// The weight operand is consumed in the reference's on-disk layout (EETQ / FasterTransformer interleave, see
// decode_kernels.hip): column pair p owns 2K contiguous bytes; per 64-row block tb: [64 B of column 2p | 64 B of column
// 2p+1]; inside a 16-byte group the even bytes are k0..k0+7 and the odd bytes k0+8..k0+15 (that is what the row
// permutation + byte swap of the layout amount to).  The layout was designed so that one 32-bit word converts into two
// fp16 PAIRS of adjacent k -- which also makes it MFMA-friendly without any re-layout:
//
//   v_mfma_f32_32x32x16_f16, MFMA "A" = 32 weight columns (n), MFMA "B" = 32 tokens (m).  Lane (n = l % 32, kg = l / 32)
//   loads ONE 16-byte group of its column (group 2j + kg of the 64-row block): v_perm_b32 on the even bytes gives
//   k0..k0+7 in order (the lane's 8 k-values of MFMA "even"), on the odd bytes k0+8..k0+15 (MFMA "odd").  The token
//   operand of lane (m, kg) for those two MFMAs is A[m, k0 .. k0+7] and A[m, k0+8 .. k0+15]: two contiguous 16-byte
//   LDS reads.  Every weight byte is loaded from HBM exactly once per 256 tokens and never touches LDS.
//   Dequantisation = the GEMV's: v_perm_b32 -> 0x6400|b (1024 + b, exact), v_pk_add_f16 -1152 (exact q - 128),
//   v_pk_mul_f16 by the column scale (ONE rounding: the fp16((q-128)*scale) the reference feeds its tensor cores).
//
// Workgroup = 4 waves x 32 columns = 128 output columns, all M rows of a 32*MT-row super-tile (MT <= 8): the token tile
// of a K stage (128 k: 32*MT rows x 256 B, XOR-swizzled 16-byte chunks) is staged ONCE per workgroup with LDS-DMA and
// shared by the four waves; weights are prefetched NST-1 stages ahead in registers.  One barrier per stage.
// Narrow N / small M leave CUs idle, so K is optionally split over `ks` workgroups per column tile (caller-provided
// scratch): every workgroup parks its fp32 partial tile with write-through stores and counts itself in; the LAST one to
// arrive adds all parts IN RANK ORDER (fp32 addition does not commute bitwise: a fixed order keeps the result
// deterministic whichever workgroup is last) and stores the fp16 tile.  Nobody waits.
#include <type_traits>

#include "mixq_device.h"
#include "mixq_launch.h"

namespace mixq {

namespace wo {
constexpr int KB = 128;        // k per stage
constexpr int ROWB = KB * 2;   // bytes per token row per stage
constexpr int BN = 128;        // columns per workgroup (4 waves x 32)
} // namespace wo

__device__ __forceinline__ v2h wo_dequant_pair(unsigned w, unsigned sel, v2h scale2)
{
    const unsigned h = __builtin_amdgcn_perm(0x64646464u, w, sel); // two fp16: 1024 + byte
    const v2h bias = {(_Float16)-1152.0f, (_Float16)-1152.0f};
    return (__builtin_bit_cast(v2h, h) + bias) * scale2; // exact subtract, one RNE in the multiply
}

// WM = 1: 4 waves, each 32 columns x all 32*MT rows.  WM = 2: 8 waves, the second four take the upper half of the rows
// (two waves per SIMD: one wave's LDS reads and dequant VALU hide under the other's MFMAs -- the large-M form).
template <int MT, int NST, int WM = 1>
__global__ __launch_bounds__(256 * WM) void w8a16_gemm_kernel(const uint16_t* __restrict__ A, const uint8_t* __restrict__ Wq,
                                                          const uint16_t* __restrict__ scale,
                                                          uint16_t* __restrict__ Out, int M, int N, int K, int ks,
                                                          void* __restrict__ scratch, const void* __restrict__ zeros)
{
    using namespace wo;
    constexpr int ROWS = MT * 32;
    constexpr int T = 256 * WM;                   // threads
    constexpr int MTW = MT / WM;                  // 32-row tiles per wave
    static_assert(MT % WM == 0, "");
    constexpr int STAGE = ROWS * ROWB;            // bytes of one token stage
    constexpr int AL = ROWS * 16 / T;             // 16-byte LDS-DMA copies per thread per stage
    constexpr int GROUP_OPS = 4 + AL;             // VMEM operations a thread issues per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int ntile = (int)blockIdx.x / ks, krank = (int)blockIdx.x % ks;
    const int wn = wave & 3, wmh = wave >> 2;     // column group, row half
    const int n0w = ntile * BN + wn * 32;         // first column of this wave
    const int trow0 = wmh * MTW * 32;             // first token row of this wave

    // ---- this workgroup's K range, in stages of 128 k -----------------------------------------------------------------
    const int nst_all = (K + KB - 1) / KB;
    const int s_begin = (int)((int64_t)nst_all * krank / ks), s_end = (int)((int64_t)nst_all * (krank + 1) / ks);
    const int nst = s_end - s_begin;

    // ---- weight stream: lane (column n, 16-byte group parity kg) -------------------------------------------------------
    // asm-form loads (wave-uniform base of the stage + this lane's offset), NOT counted by the compiler: see gload16_sbase
    const int ncol = min(n0w + lr, N - 1);        // clamped columns are computed, never stored
    const unsigned wlane = (unsigned)(ncol >> 1) * 2u * (unsigned)K + (unsigned)((ncol & 1) * 64 + lh * 16); // N K < 2^32
    v2h scale2;
    {
        _Float16 sc;
        const uint16_t sb = scale[ncol];
        __builtin_memcpy(&sc, &sb, 2);
        scale2 = v2h{sc, sc};
    }
    const bool khalf = (K % KB) != 0; // K % 64 == 0 is required, so the only ragged case is a last stage of 64 k
    auto load_w = [&](v4u (&w)[4], int s) __attribute__((always_inline)) {
        // groups (tbq, j): 64-row block 2s + tbq, 16-byte group 2j + kg of this lane's column
        const char* const b = reinterpret_cast<const char*>(Wq) + (int64_t)s * 256;
        // a ragged last stage has no second 64-row block: re-read the first (its products meet zero activations)
        const char* const b2 = (khalf && s == nst_all - 1) ? b : b + 128;
        gload16_sbase<0>(w[0], b, wlane);
        gload16_sbase<32>(w[1], b, wlane);
        gload16_sbase<0>(w[2], b2, wlane);
        gload16_sbase<32>(w[3], b2, wlane);
    };

    // ---- token tile -> LDS: chunk c = i * 256 + tid: row = c / 16, slot = c % 16 holds source chunk slot ^ (row & 15) ----
    const char* asrc[AL];
    int akoff[AL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        const int row = (i * T + tid) >> 4, slot = tid & 15;
        const int chunk = slot ^ (row & 15);
        asrc[i] = reinterpret_cast<const char*>(A) + (int64_t)min(row, M - 1) * K * 2 + chunk * 16;
        akoff[i] = chunk * 8; // first k of this chunk inside the stage
    }
    // (asm-form copies, NOT counted by the compiler: with the builtin every use of a prefetched weight register and every
    //  LDS read drained ALL outstanding VMEM operations -- the loop ran with no prefetch at all, see glds16_sbase)
    const unsigned lds0 = (unsigned)(size_t)(MIXQ_LDS_PTR(smem));
    auto stage_a = [&](int buf, int s) __attribute__((always_inline)) {
        const int64_t k0 = (int64_t)s * KB;
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            const char* src = asrc[i] + k0 * 2;
            if (khalf && k0 + akoff[i] >= K) src = static_cast<const char*>(zeros);
            glds16_vaddr(src, lds0 + buf * STAGE + (i * T + wave * 64) * 16);
        }
    };

    // ---- fragment read offsets: lane (token row t*32 + lr, kg): chunk = tbq*8 + 2*(2j + kg) + odd -----------------------
    int aoff[4][2]; // [q = tbq*2 + j][odd]
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int chunk = (q >> 1) * 8 + 2 * (2 * (q & 1) + lh) + o;
            aoff[q][o] = lr * ROWB + ((chunk ^ (lr & 15)) << 4);
        }

    v16f acc[MTW];
#pragma unroll
    for (int t = 0; t < MTW; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    auto compute = [&](const v4u (&w)[4], int buf) __attribute__((always_inline)) {
        const char* base = smem + buf * STAGE + trow0 * ROWB;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned d[4] = {w[q][0], w[q][1], w[q][2], w[q][3]};
            v2h e2[4], o2[4];
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                e2[x] = wo_dequant_pair(d[x], 0x04020400u, scale2); // bytes 0, 2 -> k0 + 2x, k0 + 2x + 1
                o2[x] = wo_dequant_pair(d[x], 0x04030401u, scale2); // bytes 1, 3 -> k0 + 8 + 2x, k0 + 9 + 2x
            }
            const v8h we = {e2[0][0], e2[0][1], e2[1][0], e2[1][1], e2[2][0], e2[2][1], e2[3][0], e2[3][1]};
            const v8h wod = {o2[0][0], o2[0][1], o2[1][0], o2[1][1], o2[2][0], o2[2][1], o2[3][0], o2[3][1]};
#pragma unroll
            for (int t = 0; t < MTW; ++t) {
                const v8h ae = *reinterpret_cast<const v8h*>(base + t * 32 * ROWB + aoff[q][0]);
                const v8h ao = *reinterpret_cast<const v8h*>(base + t * 32 * ROWB + aoff[q][1]);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(we, ae, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wod, ao, acc[t], 0, 0, 0);
            }
        }
    };

    // ---- main loop: stage i's operands were issued NST-1 iterations earlier; one barrier per stage ----------------------
    // Issue order inside an iteration: weights of stage i + NST - 1 (4 loads), then its token copies (AL): a thread's VMEM
    // operations complete in order, so "at most (NST - 2) groups outstanding" means group i has landed.
    v4u w[NST][4];
#pragma unroll
    for (int p = 0; p < NST; ++p)
#pragma unroll
        for (int q = 0; q < 4; ++q) w[p][q] = v4u{0u, 0u, 0u, 0u};
#pragma unroll
    for (int p = 0; p < NST - 1; ++p) {
        if (p < nst) {
            load_w(w[p], s_begin + p);
            stage_a(p, s_begin + p);
        }
    }
    // (the loop is unrolled NST-fold so that the register sets have static names)
    for (int i0 = 0; i0 < nst; i0 += NST) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int i = i0 + u;
            if (i >= nst) break;
            // group i was issued NST-1 iterations ago; the NST-2 younger groups may still be in flight.  Near the end fewer
            // groups were issued after it, so the bound only gets looser than needed if we do nothing: drain instead.
            if (NST > 2 && i + NST - 2 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * GROUP_OPS) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads(); // stage i visible to every wave; everyone is done reading the buffer stage i+NST-1 will reuse
            vmem_landed(w[u][0], w[u][1], w[u][2], w[u][3]);
            if (i + NST - 1 < nst) {
                load_w(w[(u + NST - 1) % NST], s_begin + i + NST - 1);
                stage_a((i + NST - 1) % NST, s_begin + i + NST - 1);
            }
            compute(w[u], i % NST);
        }
    }

    // ---- K split over workgroups: park, count in, the last one to arrive adds the parts in rank order -----------------
    if (ks > 1) {
        __shared__ unsigned arrived_s;
        constexpr int TILE = MTW * 16 * T; // floats of one parked tile: [tile of the wave][16][thread]
        unsigned* const counter = static_cast<unsigned*>(scratch) + ntile;
        float* const slots = reinterpret_cast<float*>(static_cast<char*>(scratch) + kSplitkWordsBytes) +
                             (size_t)ntile * ks * TILE;
        float* const mine = slots + (size_t)krank * TILE + tid;
#pragma unroll
        for (int t = 0; t < MTW; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                __hip_atomic_store(mine + (t * 16 + e) * T, acc[t][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every write-through acknowledged
        __syncthreads();
        if (tid == 0) arrived_s = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (arrived_s != (unsigned)(ks - 1)) return;
        if (tid == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // re-armed
        v16f own[MTW];
#pragma unroll
        for (int t = 0; t < MTW; ++t) own[t] = acc[t];
#pragma unroll
        for (int t = 0; t < MTW; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
        for (int r = 0; r < ks; ++r) { // rank order, whichever workgroup is last
            const float* const theirs = slots + (size_t)r * TILE + tid;
            const bool self = r == krank;
#pragma unroll
            for (int t = 0; t < MTW; ++t)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    acc[t][e] += self ? own[t][e]
                                      : __hip_atomic_load(theirs + (t * 16 + e) * T, __ATOMIC_RELAXED,
                                                          __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // ---- store: lane holds, per 32x32 tile, 4 x (4 consecutive columns) of token row t*32 + lr -------------------------
#pragma unroll
    for (int t = 0; t < MTW; ++t) {
        const int m = trow0 + t * 32 + lr;
        if (m >= M) continue;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int n = n0w + 8 * g + 4 * lh;
            uint16_t* dst = Out + (int64_t)m * N + n;
            const v2h lo = f2h2(acc[t][4 * g], acc[t][4 * g + 1]), hi = f2h2(acc[t][4 * g + 2], acc[t][4 * g + 3]);
            if (n + 3 < N) {
                *reinterpret_cast<uint2*>(dst) = uint2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
            } else if (n < N) { // N % 4 == 2: the last quad is half valid
                *reinterpret_cast<unsigned*>(dst) = __builtin_bit_cast(unsigned, lo);
            }
        }
    }
}

// ---- the large-M form: 64-column wave tiles, M tiled in the grid -----------------------------------------------------------
// Above one pass of 256 tokens the kernel above is bound by LDS bandwidth, not by the matrix cores: its waves are 32
// columns wide, so every v_mfma_f32_32x32x16_f16 needs its own 1-KiB token fragment from LDS (8 LDS cycles per 32 MFMA
// cycles on each of 4 SIMDs = the whole LDS port), and every 256-token pass streams and dequantises all weights again.
// Here a wave owns 64 columns x 32*MTW rows: each token fragment feeds TWO MFMAs (LDS traffic per MFMA halved), a
// workgroup (8 waves = 2 row halves x 4 column groups) owns a 64*MTW-row x 256-column tile of the output and the grid
// walks the M x N tiles (grouped along M, one contiguous range per XCD, so that the workgroups resident on an XCD share
// weight columns and token rows in its L2).  K stage = 64 k (one 64-row block of the interleaved layout).
//
// BOTH operands of a stage travel global -> LDS as asm-form LDS-DMA (glds16_sbase: not counted by the compiler, explicit
// vmcnt accounting, NST-1 stages in flight): the token tile (rows x 128 B) and the RAW weight bytes of the 256 columns
// (128 column pairs x 128 B = 16 KiB, fetched once per workgroup instead of once per row half).  A lane then reads its
// 16-byte weight groups from LDS like the GEMV reads them from memory, so the dequantisation is unchanged.  With the
// builtin LDS-DMA + weights prefetched in registers (first build of this kernel) the compiler drained EVERY outstanding
// load before each use of a weight register (see glds16_sbase): 2.2 us per stage = one memory latency.
//
// The main loop is software-pipelined by hand: while the 4 MFMAs of (k step j, row tile t) run, the token fragments of
// the next (j, t) are read and a share of step j + 1's weights is dequantised (v_perm / v_pk_add / v_pk_mul ride in the
// shadow of the MFMAs; sched_group_barrier pins the interleave).
// Same arithmetic as above per output element (same MFMA, same k order inside a K range, same ordered reduction when K
// is split), so both forms and the oracle agree to the same tolerance; the two forms are not bit-identical to each other
// only where their K splits differ.
namespace wo {
constexpr int KBW = 64;        // k per stage of the wide form
constexpr int ROWBW = KBW * 2; // bytes per token row per stage
constexpr int BNW = 256;       // columns per workgroup (4 waves x 64)
constexpr int WSTAGE = BNW * KBW; // weight bytes per stage
} // namespace wo

// ABL (measurement only, wrong results): 1 no copies in the loop, 2 no dequantisation, 4 token fragments read once, 8 no MFMAs
// WMH = 3 ("K halves"): 8 waves like WMH = 2, but the second four take the SECOND k step of every stage for the same rows
// instead of other rows; the two partial tiles meet in LDS after the loop.  For short tiles: two waves per SIMD (one's
// dequantisation and LDS waits under the other's MFMAs) where a 4-wave workgroup leaves each SIMD a single wave.
// RW ("register weights", WMH = 1 or 3 only: no other wave of the workgroup wants this wave's weight bytes): the weights
// skip LDS -- each lane loads its own 16-byte groups straight into registers (asm form, gload16_sbase), NST - 1 stages
// ahead like the token copies.  A register load costs the issuing wave a few cycles where an LDS-DMA copy costs 60-185,
// and the ds_read_b128 of the weight groups disappear; LDS holds token stages only.
template <int MTW, int WMH, int NST, int ABL = 0, bool RW = false>
__global__ __launch_bounds__(256 * (WMH == 3 ? 2 : WMH)) void w8a16_gemm_wide_kernel(const uint16_t* __restrict__ A, const uint8_t* __restrict__ Wq,
                                                               const uint16_t* __restrict__ scale,
                                                               uint16_t* __restrict__ Out, int M, int N, int K, int ks,
                                                               void* __restrict__ scratch)
{
    using namespace wo;
    constexpr bool KH = WMH == 3;
    constexpr int T = 256 * (KH ? 2 : WMH);       // WMH = 1: 4 waves (one per SIMD, each all rows), 2: 8 waves (two row halves)
    constexpr int ROWS = (KH ? 1 : WMH) * MTW * 32; // token rows of the workgroup tile
    constexpr int TA = KH ? 256 : T;              // threads that hold finished accumulators (K halves: waves 0-3 after the merge)
    static_assert(!RW || WMH != 2, "register weights: one wave per column group and k step");
    static_assert(!RW || ABL == 0, "");
    constexpr int ASTAGE = ROWS * ROWBW;          // token bytes of one stage
    constexpr int STAGE = ASTAGE + (RW ? 0 : WSTAGE); // [tokens | raw weights]
    constexpr int AL = ROWS * 8 / T;              // token copies (16 B) per thread per stage
    constexpr int NJ = KH ? 1 : 2;                // k steps of a stage this wave runs
    constexpr int WL = RW ? 2 * NJ : WSTAGE / 16 / T; // weight copies (loads) per thread per stage (2 or 4)
    constexpr int GROUP_OPS = AL + WL;            // VMEM operations a thread issues per stage
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 31, lh = lane >> 5;
    const int wn = wave & 3, wmh = wave >> 2;     // column group (64 columns), row half (0 when WMH = 1)

    // ---- block -> (tile, K rank): the ks workgroups of a tile are consecutive blocks; tiles: XCD-contiguous, grouped ----
    const int tiles_m = (M + ROWS - 1) / ROWS, tiles_n = (N + BNW - 1) / BNW;
    const int nwg = tiles_m * tiles_n;
    const int bid = (int)blockIdx.x / ks, krank = (int)blockIdx.x % ks;
    int tile_m, tile_n;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        const int t_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        constexpr int GROUP_M = 4;
        const int per_group = GROUP_M * tiles_n;
        const int g = t_lin / per_group, first_m = g * GROUP_M;
        const int gsz = min(tiles_m - first_m, GROUP_M);
        const int within = t_lin - g * per_group;
        tile_m = first_m + within % gsz;
        tile_n = within / gsz;
    }
    const int m0 = tile_m * ROWS;
    const int n0w = tile_n * BNW + wn * 64;       // first column of this wave
    const int trow0 = KH ? 0 : wmh * MTW * 32;    // first token row of this wave inside the tile

    const int nst_all = K / KBW;
    const int s_begin = (int)((int64_t)nst_all * krank / ks), s_end = (int)((int64_t)nst_all * (krank + 1) / ks);
    const int nst = s_end - s_begin;

    // ---- stage copies: wave-uniform base + per-thread offset, destination = this wave's 1-KiB window of each 8-KiB row --
    // tokens: chunk c = i * T + tid: row = c / 8, slot = c % 8 holds source chunk slot ^ ((row / 2) % 8): rows are 128 B = half a
    //         256-B bank row, so a 16-lane group of ds_read_b128 (rows 0-3, 12-15, 20-27 ...) needs (row % 2, slot) distinct
    // weights: chunk q = i * T + tid: local pair pl = q / 8; slot q % 8 holds the pair's 16-byte group (q % 8) ^ ((pl / 2) % 4)
    //          (group g < 4: column 2p, 16-byte group g of the 64-row block; g >= 4: column 2p + 1, group g - 4)
    const char* const abase = reinterpret_cast<const char*>(A) + ((int64_t)m0 * K + (int64_t)s_begin * KBW) * 2;
    const int pair0 = tile_n * (BNW / 2), npairs = N >> 1;
    const char* const wbase = reinterpret_cast<const char*>(Wq) + (int64_t)pair0 * 2 * K + (int64_t)s_begin * 128;
    unsigned aoffv[AL], woffv[WL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        const int row = (i * T + tid) >> 3, slot = tid & 7;
        aoffv[i] = (unsigned)(min(m0 + row, M - 1) - m0) * (unsigned)K * 2u + (unsigned)((slot ^ ((row >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < WL; ++i) {
        if constexpr (RW) { // load i = cb * NJ + j: this lane's group (column parity, 2 j + kg) of its column pair's 64-row block
            const int cb = i / NJ, j = KH ? wmh : i % NJ;
            const int pl = wn * 32 + cb * 16 + (lr >> 1), g = (lr & 1) * 4 + 2 * j + lh;
            woffv[i] = (unsigned)(min(pair0 + pl, npairs - 1) - pair0) * (unsigned)K * 2u + (unsigned)(g << 4);
        } else {
            const int pl = (i * T + tid) >> 3, gq = (tid & 7) ^ ((pl >> 1) & 3);
            woffv[i] = (unsigned)(min(pair0 + pl, npairs - 1) - pair0) * (unsigned)K * 2u + (unsigned)(gq << 4);
        }
    }
    v4u wreg[RW ? NST : 1][RW ? WL : 1]; // (RW) [stage buffer][cb * NJ + j]: raw 16-byte groups, NST - 1 stages ahead
#pragma unroll
    for (int b = 0; b < (RW ? NST : 1); ++b)
#pragma unroll
        for (int i = 0; i < (RW ? WL : 1); ++i) wreg[b][i] = v4u{0u, 0u, 0u, 0u};
    const unsigned lds0 = (unsigned)(size_t)(MIXQ_LDS_PTR(smem)) + wave * 1024;
    auto issue = [&](int rel, int buf) __attribute__((always_inline)) { // stage s_begin + rel -> buffer buf
        if ((ABL & 1) && rel >= NST - 1) return;
        const char* const ab = abase + (int64_t)rel * (KBW * 2);
        const char* const wb = wbase + (int64_t)rel * 128;
#pragma unroll
        for (int i = 0; i < AL; ++i) glds16_sbase(ab, aoffv[i], lds0 + buf * STAGE + i * T * 16);
#pragma unroll
        for (int i = 0; i < WL; ++i) {
            if constexpr (RW) gload16_sbase<0>(wreg[buf][i], wb, woffv[i]);
            else glds16_sbase(wb, woffv[i], lds0 + buf * STAGE + ASTAGE + i * T * 16);
        }
    };
    // the same copies one piece at a time (p < GROUP_OPS: token pieces first), for spreading them over a stage's MFMA groups
    auto issue_piece = [&](int rel, int buf, int pc) __attribute__((always_inline)) {
        if ((ABL & 1) && rel >= NST - 1) return;
        if (pc < AL) glds16_sbase(abase + (int64_t)rel * (KBW * 2), aoffv[pc], lds0 + buf * STAGE + pc * T * 16);
        else if constexpr (RW) gload16_sbase<0>(wreg[buf][pc - AL], wbase + (int64_t)rel * 128, woffv[pc - AL]);
        else glds16_sbase(wbase + (int64_t)rel * 128, woffv[pc - AL], lds0 + buf * STAGE + ASTAGE + (pc - AL) * T * 16);
    };

    // ---- fragment read offsets ----------------------------------------------------------------------------------------
    // tokens: lane (row t*32 + lr, kg): chunk = 2*(2j + kg) + odd;  weights: lane (column, kg): group 2j + kg of its column
    int aoff[2][2]; // [j][odd]
    int woff[2][2]; // [cb][j]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            const int chunk = 2 * (2 * j + lh) + o;
            aoff[j][o] = lr * ROWBW + ((chunk ^ ((lr >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int pl = wn * 32 + cb * 16 + (lr >> 1), g = (lr & 1) * 4 + 2 * j + lh;
            woff[cb][j] = ASTAGE + ((pl * 8 + (g ^ ((pl >> 1) & 3))) << 4);
        }
    }
    if (KH && wmh == 1) { // this wave's k step is the second of every stage: its offsets take the place of step 0's
#pragma unroll
        for (int o = 0; o < 2; ++o) aoff[0][o] = aoff[1][o];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) woff[cb][0] = woff[cb][1];
    }
    v2h scale2[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        _Float16 sc;
        const uint16_t sb = scale[min(n0w + cb * 32 + lr, N - 1)]; // clamped columns are computed, never stored
        __builtin_memcpy(&sc, &sb, 2);
        scale2[cb] = v2h{sc, sc};
    }

    v16f acc[2][MTW];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int t = 0; t < MTW; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[cb][t][e] = 0.f;

    // ---- main loop: NST - 1 stages in flight, one barrier per stage ------------------------------------------------------
    // A stage = two k steps (j) of MTW row tiles x 4 MFMAs.  Step 0's MFMAs hide the dequantisation of step 1's weights;
    // step 0's own weights are dequantised in the open at the top of the stage.  (A second barrier in the middle of the
    // stage, certifying stage i+1 so that step 1 hides the NEXT stage's step 0 as well, measured the same: the stage is
    // bound by issue slots and power, not by that bubble -- ablations in DESIGN.md 2.5.)
#pragma unroll
    for (int p = 0; p < NST - 1; ++p)
        if (p < nst) issue(p, p);
    v2h E[2][2][4], O[2][2][4]; // [k step j][column block][x]: dequantised operands (even / odd k halves of a 16-byte group)
    auto dequant_half = [&](const uint4& wv, int h, v2h sc, v2h (&e2)[4], v2h (&o2)[4]) __attribute__((always_inline)) {
        const unsigned d0 = h ? wv.z : wv.x, d1 = h ? wv.w : wv.y;
        if (ABL & 2) {
            e2[2 * h] = o2[2 * h] = __builtin_bit_cast(v2h, d0);
            e2[2 * h + 1] = o2[2 * h + 1] = __builtin_bit_cast(v2h, d1);
            return;
        }
        e2[2 * h] = wo_dequant_pair(d0, 0x04020400u, sc);     // bytes 0, 2 -> k0 + 4h .. + 1
        o2[2 * h] = wo_dequant_pair(d0, 0x04030401u, sc);     // bytes 1, 3 -> k0 + 8 + 4h ..
        e2[2 * h + 1] = wo_dequant_pair(d1, 0x04020400u, sc);
        o2[2 * h + 1] = wo_dequant_pair(d1, 0x04030401u, sc);
    };
    // certify group g: the NST-2 groups issued after it may still be in flight (fewer near the end: drain instead)
    auto certify = [&](int g, int buf) __attribute__((always_inline)) {
        if (NST > 2 && g + NST - 2 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NST - 2) * GROUP_OPS) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if constexpr (RW) { // every later use of this stage's weight registers depends on a statement behind the wait
#pragma unroll
            for (int i = 0; i < WL; ++i) asm volatile("" : "+v"(wreg[buf][i])::"memory");
        }
    };
    // one k step: MFMAs of (j, all row tiles) with the operands E[j] / O[j]; meanwhile `wnext` ([cb] raw groups) is
    // dequantised into E[j ^ 1] / O[j ^ 1] (DEQ) and the token fragments of the next (j, t) are read
    auto step = [&](const char* base, auto j_tag, const uint4 (&wnext)[2], auto deq_tag, auto&& before_tile)
                    __attribute__((always_inline)) {
        constexpr int j = decltype(j_tag)::value;
        constexpr bool DEQ = decltype(deq_tag)::value;
        v8h fa[2], fo[2];
        fa[0] = *reinterpret_cast<const v8h*>(base + aoff[j][0]);
        fo[0] = *reinterpret_cast<const v8h*>(base + aoff[j][1]);
#pragma unroll
        for (int t = 0; t < MTW; ++t) {
            const int cur = t & 1;
            before_tile(j, t); // (this stage's share of the NEXT copies: issued while the previous tile's MFMAs drain)
            if (t + 1 < MTW && !(ABL & 4)) {
                fa[cur ^ 1] = *reinterpret_cast<const v8h*>(base + (t + 1) * 32 * ROWBW + aoff[j][0]);
                fo[cur ^ 1] = *reinterpret_cast<const v8h*>(base + (t + 1) * 32 * ROWBW + aoff[j][1]);
            }
            const v8h we0 = {E[j][0][0][0], E[j][0][0][1], E[j][0][1][0], E[j][0][1][1],
                             E[j][0][2][0], E[j][0][2][1], E[j][0][3][0], E[j][0][3][1]};
            const v8h we1 = {E[j][1][0][0], E[j][1][0][1], E[j][1][1][0], E[j][1][1][1],
                             E[j][1][2][0], E[j][1][2][1], E[j][1][3][0], E[j][1][3][1]};
            const v8h wo0 = {O[j][0][0][0], O[j][0][0][1], O[j][0][1][0], O[j][0][1][1],
                             O[j][0][2][0], O[j][0][2][1], O[j][0][3][0], O[j][0][3][1]};
            const v8h wo1 = {O[j][1][0][0], O[j][1][0][1], O[j][1][1][0], O[j][1][1][1],
                             O[j][1][2][0], O[j][1][2][1], O[j][1][3][0], O[j][1][3][1]};
            const int fc = (ABL & 4) ? 0 : cur;
            if (ABL & 8) {
                acc[0][t][0] += we0[0] * fa[fc][0] + wo0[1] * fo[fc][1];
                acc[1][t][0] += we1[0] * fa[fc][0] + wo1[1] * fo[fc][1];
            } else {
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(we0, fa[fc], acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(we1, fa[fc], acc[1][t], 0, 0, 0);
                acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wo0, fo[fc], acc[0][t], 0, 0, 0);
                acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wo1, fo[fc], acc[1][t], 0, 0, 0);
            }
            // this row tile's share of the next step's dequantisation: 4 (cb, h) units spread over the MTW tiles
            constexpr int UNITS_LO = MTW >= 4 ? 1 : 4 / MTW; // units per tile that takes any (MTW = 8: every other tile)
            constexpr int EVERY = MTW > 4 ? MTW / 4 : 1;
            const bool takes = (t % EVERY) == 0;
            if constexpr (DEQ) {
                if (takes) {
#pragma unroll
                    for (int q = 0; q < UNITS_LO; ++q) {
                        const int unit = (MTW <= 4 ? t * UNITS_LO : t / EVERY) + q, cb = unit >> 1, h = unit & 1;
                        dequant_half(wnext[cb], h, scale2[cb], E[j ^ 1][cb], O[j ^ 1][cb]);
                    }
                }
            }
            // pin the order: fragment reads first, then one MFMA : its share of the VALU work, four times
            if constexpr (ABL == 0) {
                if (t + 1 < MTW) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (DEQ && takes) __builtin_amdgcn_sched_group_barrier(0x002, 3 * UNITS_LO, 0);
                }
            }
        }
    };
    for (int i0 = 0; i0 < nst; i0 += NST) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int i = i0 + u;
            if (i >= nst) break;
            const char* const sb = smem + u * STAGE;
            const char* const base = sb + trow0 * ROWBW;
            uint4 wr[2];
            {
                // stage i certified at its top; step 0's weights are dequantised in the open, step 1's under step 0's MFMAs
                // (K halves: each wave runs ONE step per stage, its own)
                certify(i, u);
                // the copies of stage i + NST - 1 are spread over this stage's tile steps: a global_load_lds costs the issuing
                // wave 60-185 cycles, six or more of them in a row at the top of a stage are a third of a short tile's stage
                const bool more = i + NST - 1 < nst;
                constexpr int TS = (KH ? 1 : 2) * MTW; // tile steps of a stage (per wave)
                auto before_tile = [&](int jj, int tt) __attribute__((always_inline)) {
                    const int ts = (KH ? 0 : jj) * MTW + tt;
                    if (more) {
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int pc = 0; pc < GROUP_OPS; ++pc)
                            if (pc * TS / GROUP_OPS == ts) issue_piece(i + NST - 1, (u + NST - 1) % NST, pc);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                };
                uint4 w0[2];
#pragma unroll
                for (int cb = 0; cb < 2; ++cb) {
                    if constexpr (RW) {
                        const v4u a = wreg[u][cb * NJ], b = wreg[u][cb * NJ + NJ - 1];
                        w0[cb] = uint4{a[0], a[1], a[2], a[3]};
                        wr[cb] = uint4{b[0], b[1], b[2], b[3]};
                    } else {
                        w0[cb] = *reinterpret_cast<const uint4*>(sb + woff[cb][0]);
                        wr[cb] = *reinterpret_cast<const uint4*>(sb + woff[cb][1]);
                    }
                }
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int h = 0; h < 2; ++h) dequant_half(w0[cb], h, scale2[cb], E[0][cb], O[0][cb]);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (KH) {
                    step(base, std::integral_constant<int, 0>{}, wr, std::false_type{}, before_tile);
                } else {
                    step(base, std::integral_constant<int, 0>{}, wr, std::true_type{}, before_tile);
                    __builtin_amdgcn_sched_barrier(0);
                    step(base, std::integral_constant<int, 1>{}, wr, std::false_type{}, before_tile);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- K halves: waves 4-7 hand their partial tile to waves 0-3 through LDS (the stage buffers are dead) ---------------
    if constexpr (KH) {
        __syncthreads();
        float* const xch = reinterpret_cast<float*>(smem) + (tid & 255);
        if (wmh == 1) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int t = 0; t < MTW; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) xch[((cb * MTW + t) * 16 + e) * 256] = acc[cb][t][e];
        }
        __syncthreads();
        if (wmh == 0) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int t = 0; t < MTW; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[cb][t][e] += xch[((cb * MTW + t) * 16 + e) * 256];
        }
    }
    const bool holder = !KH || wmh == 0; // this wave holds finished accumulators (the others only keep the barrier count)

    // ---- K split over workgroups: park, count in, the last one to arrive adds the parts in rank order -----------------
    if (ks > 1) {
        __shared__ unsigned arrived_s;
        constexpr int TILE = 2 * MTW * 16 * TA; // floats of one parked tile: [column block][row tile][16][thread]
        unsigned* const counter = static_cast<unsigned*>(scratch) + (tile_m * tiles_n + tile_n);
        float* const slots = reinterpret_cast<float*>(static_cast<char*>(scratch) + kSplitkWordsBytes) +
                             (size_t)(tile_m * tiles_n + tile_n) * ks * TILE;
        float* const mine = slots + (size_t)krank * TILE + tid;
        if (holder) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int t = 0; t < MTW; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        __hip_atomic_store(mine + ((cb * MTW + t) * 16 + e) * TA, acc[cb][t][e], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every write-through acknowledged
        __syncthreads();
        if (tid == 0) arrived_s = __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (arrived_s != (unsigned)(ks - 1)) return;
        if (tid == 0) __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // re-armed
        // rank order, whichever workgroup is last; the own part is read back like the others (it was parked above with
        // write-through stores): no second register copy of the tile.
        constexpr int NE = 2 * MTW * 16; // floats per lane
        if (!holder) {
            // (K halves: waves 4-7 have nothing left to do)
        } else if constexpr (NE <= 64) {
            // short tiles (32 / 64 rows, split up to 16 ways): the NEXT rank's loads are in flight while this rank's are
            // added -- a rank at a time costs one memory round trip per rank (~1.3 us x ks on a 25-us launch)
            float* const accf = reinterpret_cast<float*>(&acc[0][0]); // element q <-> parked word q * TA, q = (cb * MTW + t) * 16 + e
            float cur[NE], nxt[NE], sum[NE];
            auto fetch = [&](float (&dst)[NE], int r) __attribute__((always_inline)) {
                const float* const theirs = slots + (size_t)r * TILE + tid;
#pragma unroll
                for (int q = 0; q < NE; ++q)
                    dst[q] = __hip_atomic_load(theirs + q * TA, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            };
#pragma unroll
            for (int q = 0; q < NE; ++q) sum[q] = 0.f;
            fetch(cur, 0);
            for (int r = 0; r < ks; r += 2) {
                if (r + 1 < ks) fetch(nxt, r + 1);
#pragma unroll
                for (int q = 0; q < NE; ++q) sum[q] += cur[q];
                if (r + 2 < ks) fetch(cur, r + 2);
                if (r + 1 < ks) {
#pragma unroll
                    for (int q = 0; q < NE; ++q) sum[q] += nxt[q];
                }
            }
#pragma unroll
            for (int q = 0; q < NE; ++q) accf[q] = sum[q];
        } else {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int t = 0; t < MTW; ++t)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[cb][t][e] = 0.f;
            for (int r = 0; r < ks; ++r) { // (tall tiles: all loads of a rank in flight together; the registers hold no more)
                const float* const theirs = slots + (size_t)r * TILE + tid;
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int t = 0; t < MTW; ++t)
#pragma unroll
                        for (int e = 0; e < 16; ++e)
                            acc[cb][t][e] += __hip_atomic_load(theirs + ((cb * MTW + t) * 16 + e) * TA, __ATOMIC_RELAXED,
                                                               __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }

    // ---- store: lane holds, per 32x32 tile, 4 x (4 consecutive columns) of token row t*32 + lr -------------------------
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int t = 0; t < MTW; ++t) {
            const int m = m0 + trow0 + t * 32 + lr;
            if (m >= M || !holder) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n = n0w + cb * 32 + 8 * g + 4 * lh;
                uint16_t* dst = Out + (int64_t)m * N + n;
                const v2h lo = f2h2(acc[cb][t][4 * g], acc[cb][t][4 * g + 1]);
                const v2h hi = f2h2(acc[cb][t][4 * g + 2], acc[cb][t][4 * g + 3]);
                if (n + 3 < N) {
                    *reinterpret_cast<uint2*>(dst) = uint2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
                } else if (n < N) { // N % 4 == 2: the last quad is half valid
                    *reinterpret_cast<unsigned*>(dst) = __builtin_bit_cast(unsigned, lo);
                }
            }
        }
}

// ---- the skinny form (1 .. 48 tokens): an MFMA GEMV ------------------------------------------------------------------------
// Up to 32 tokens the operator is a weight stream (N K bytes) with almost no arithmetic; the narrow form spends its time on
// launch + first-byte latency + the cross-workgroup K split it needs to fill the chip with 128-column tiles.  Built like
// the decode GEMV and the int8 skinny kernel instead: a workgroup = 32 output columns x KW waves that split K (partials meet
// in LDS: no cross-workgroup hand-over, no scratch), weights straight from memory into MFMA A fragments
// (v_mfma_f32_16x16x32_f16: lane (n = l % 16, kq = l / 16) loads 16-byte group kq of its column's 64-row block; even bytes
// = the lane's 8 k-values of one MFMA, odd bytes those of the next), the token fragments (B operand: lane (m, kq) = 32
// contiguous bytes of token row m per block) through L2, shared by the wave's two 16-column groups.  No LDS staging, no
// barrier in the loop, every load counted by the compiler.
typedef float v4f_ __attribute__((ext_vector_type(4)));

// CG: 16-column groups per wave (2 or 4: the token fragments are shared by CG MFMA A operands)
// NTW: the weight loads are non-temporal.  Round 4, A/B on one box (profiles/r04_nt_weight_loads_ab.txt): a decode step of Llama-2-7B's
// 96 linears (every layer its own weights: cold by construction) 1087 -> 1037 us at batch 1, 1124 -> 1070 at batch 4; single cold
// calls on 45-50 MB of weights -3.5..-4.4 %; 16.8 MB of weights +2.4 % cold; and +25..30 % for a loop that re-reads ONE layer (the
// Infinity Cache serves such a loop, the hint gives that up).  Hence only for weights of wo_nt_weight_bytes() (32 MiB) and more: a
// model with layers that large streams them from HBM once per step whatever the policy.  The int8 decode-batch GEMM measured +2 %
// with the same hint on the row-major weight and keeps plain loads THERE; its registered weight images of 32 MiB and more (WFRAG == 2,
// gemm_skinny_kernels.hip) and the packed-int4 stream (int4_gemm_kernels.hip) do use non-temporal loads.
template <int MT, int KW, int CG, bool NTW = false>
__global__ __launch_bounds__(KW * 64) void w8a16_skinny_kernel(const uint16_t* __restrict__ A, const uint8_t* __restrict__ Wq,
                                                                const uint16_t* __restrict__ scale,
                                                                uint16_t* __restrict__ Out, int M, int N, int K)
{
    __shared__ v4f_ part[KW][CG][MT][64]; // [K part][column group][token tile][lane]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int ln = lane & 15, kq = lane >> 4;
    const int n0 = blockIdx.x * (16 * CG);

    const int nblk = K >> 6; // 64-row blocks
    const int per = (nblk + KW - 1) / KW;
    const int b_begin = min(wave * per, nblk), b_end = min(b_begin + per, nblk);

    const uint8_t* wsrc[CG];
    v2h scale2[CG];
#pragma unroll
    for (int cg = 0; cg < CG; ++cg) {
        const int col = min(n0 + cg * 16 + ln, N - 1); // clamped columns are computed, never stored
        wsrc[cg] = Wq + (int64_t)(col >> 1) * 2 * K + (col & 1) * 64 + kq * 16;
        _Float16 sc;
        const uint16_t sb = scale[col];
        __builtin_memcpy(&sc, &sb, 2);
        scale2[cg] = v2h{sc, sc};
    }
    const uint16_t* asrc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) asrc[t] = A + (int64_t)min(t * 16 + ln, M - 1) * K + kq * 16;

    v4f_ acc[CG][MT];
#pragma unroll
    for (int cg = 0; cg < CG; ++cg)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[cg][t] = v4f_{0.f, 0.f, 0.f, 0.f};

    constexpr int UB = CG == 1 ? 8 : CG == 2 ? 4 : 2; // blocks per batch: 8 weight loads + 2 UB MT token loads in flight per lane
    for (int b0 = b_begin; b0 < b_end; b0 += UB) {
        uint4 wv[UB][CG];
        uint4 av[UB][MT][2];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int b = min(b0 + u, b_end - 1); // (the tail re-reads the last block; its products are skipped below)
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) wv[u][cg] = wload16<NTW>(wsrc[cg] + (int64_t)b * 128);
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                av[u][t][0] = *reinterpret_cast<const uint4*>(asrc[t] + b * 64);
                av[u][t][1] = *reinterpret_cast<const uint4*>(asrc[t] + b * 64 + 8);
            }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (b0 + u >= b_end) break;
#pragma unroll
            for (int cg = 0; cg < CG; ++cg) {
                const unsigned d[4] = {wv[u][cg].x, wv[u][cg].y, wv[u][cg].z, wv[u][cg].w};
                v2h e2[4], o2[4];
#pragma unroll
                for (int x = 0; x < 4; ++x) {
                    e2[x] = wo_dequant_pair(d[x], 0x04020400u, scale2[cg]);
                    o2[x] = wo_dequant_pair(d[x], 0x04030401u, scale2[cg]);
                }
                const v8h we = {e2[0][0], e2[0][1], e2[1][0], e2[1][1], e2[2][0], e2[2][1], e2[3][0], e2[3][1]};
                const v8h wod = {o2[0][0], o2[0][1], o2[1][0], o2[1][1], o2[2][0], o2[2][1], o2[3][0], o2[3][1]};
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    acc[cg][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(we, __builtin_bit_cast(v8h, av[u][t][0]), acc[cg][t], 0, 0, 0);
                    acc[cg][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wod, __builtin_bit_cast(v8h, av[u][t][1]), acc[cg][t], 0, 0, 0);
                }
            }
        }
    }

#pragma unroll
    for (int cg = 0; cg < CG; ++cg)
#pragma unroll
        for (int t = 0; t < MT; ++t) part[wave][cg][t][lane] = acc[cg][t];
    __syncthreads();

    // wave w finishes (column group, token tile) w: the KW parts are added in K order (a fixed order: same bits every run)
    for (int job = wave; job < CG * MT; job += KW) {
        const int cg = job / MT, t = job % MT;
        v4f_ sum = part[0][cg][t][lane];
#pragma unroll
        for (int w2 = 1; w2 < KW; ++w2) sum += part[w2][cg][t][lane];
        // C/D layout of the 16x16 MFMA: column (token) = lane % 16, rows (output columns) 4 * (lane / 16) + r
        const int m = t * 16 + ln, n = n0 + cg * 16 + 4 * kq;
        if (m >= M) continue;
        uint16_t* dst = Out + (int64_t)m * N + n;
        const v2h lo = f2h2(sum[0], sum[1]), hi = f2h2(sum[2], sum[3]);
        if (n + 3 < N && (N & 3) == 0) {
            *reinterpret_cast<uint2*>(dst) = uint2{__builtin_bit_cast(unsigned, lo), __builtin_bit_cast(unsigned, hi)};
        } else {
            if (n + 1 < N) *reinterpret_cast<unsigned*>(dst) = __builtin_bit_cast(unsigned, lo);
            if (n + 3 < N) *reinterpret_cast<unsigned*>(dst + 2) = __builtin_bit_cast(unsigned, hi);
        }
    }
}

// ---- the two-pass form (large M): dequantise W once, then the 256x256 ping-pong kernel with the fp16 MFMA -------------------
// In the fused forms every weight is dequantised once per 64 / 128 token rows (3 packed-f16 VALU per 2 weights beside the
// MFMAs: -25 % of the time of a full-chip launch when ablated).  From ~1300 tokens on it is cheaper to pay
// ~3 bytes of HBM traffic per weight once: this kernel writes Wf[n][k] = fp16((q[k][n] - 128) * scale[n]) -- the very
// values the fused forms feed the matrix cores -- row-major with K contiguous, which is the B operand layout of
// gemm_w8a8o16_pp_kernel<EPI_F16GEMM> (gemm_pp_kernels.hip).  One thread per 16-byte group of the interleaved image:
// group G of the image = column pair G / (K/8), 64-row block (G % (K/8)) / 8, column 2p + (G % 8) / 4, group (G % 4) of
// the block: even bytes k0..k0+7, odd bytes k0+8..k0+15 (see the header).  Fully coalesced 16-byte loads, 32-byte stores.
__global__ __launch_bounds__(256) void w8a16_dequant_kernel(const uint8_t* __restrict__ Wq, const uint16_t* __restrict__ scale,
                                                             uint16_t* __restrict__ Wf, int N, int K)
{
    const int64_t G = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int gpp = K >> 3; // 16-byte groups per column pair
    if (G >= (int64_t)(N >> 1) * gpp) return;
    const int pair = (int)(G / gpp), r = (int)(G - (int64_t)pair * gpp);
    const int col = 2 * pair + ((r >> 2) & 1), k0 = (r >> 3) * 64 + (r & 3) * 16;
    const uint4 wv = *reinterpret_cast<const uint4*>(Wq + G * 16);
    _Float16 sc;
    const uint16_t sb = scale[col];
    __builtin_memcpy(&sc, &sb, 2);
    const v2h sc2 = {sc, sc};
    const unsigned d[4] = {wv.x, wv.y, wv.z, wv.w};
    unsigned e[4], o[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
        e[x] = __builtin_bit_cast(unsigned, wo_dequant_pair(d[x], 0x04020400u, sc2)); // k0 + 2x, + 1
        o[x] = __builtin_bit_cast(unsigned, wo_dequant_pair(d[x], 0x04030401u, sc2)); // k0 + 8 + 2x, + 1
    }
    uint4* const dst = reinterpret_cast<uint4*>(Wf + (int64_t)col * K + k0);
    dst[0] = uint4{e[0], e[1], e[2], e[3]};
    dst[1] = uint4{o[0], o[1], o[2], o[3]};
}

// ---- host side ----------------------------------------------------------------------------------------------------------
struct WoPlan {
    int mt;  // 32-row tiles per pass (1, 2, 4, 8)
    int ks;  // workgroups per column tile
};

static WoPlan wo_plan(int M, int N, int K, bool have_scratch)
{
    const int rows = M < 256 ? M : 256;
    WoPlan pl{rows <= 32 ? 1 : rows <= 64 ? 2 : rows <= 128 ? 4 : 8, 1};
    if (!have_scratch) return pl;
    const int ntiles = (N + wo::BN - 1) / wo::BN, nst = (K + wo::KB - 1) / wo::KB;
    // as many workgroups as keep the chip streaming (two per CU while the tile is small), at least 4 stages each
    const int target = (pl.mt <= 2 ? 2 : 1) * num_cus();
    int ks = target / ntiles;
    if (ks > nst / 4) ks = nst / 4;
    if (ks > 32) ks = 32;
    pl.ks = ks >= 2 ? ks : 1;
    return pl;
}

static size_t wo_narrow_workspace(int rows, int N, int K)
{
    const WoPlan pl = wo_plan(rows, N, K, true);
    if (pl.ks <= 1) return 0;
    const size_t ntiles = (size_t)(N + wo::BN - 1) / wo::BN;
    return kSplitkWordsBytes + ntiles * pl.ks * (size_t)pl.mt * 16 * 256 * sizeof(float);
}

// The wide form (64-column wave tiles, both operands through LDS-DMA, M tiled in the grid): its configurations.
struct WoCfg {
    int rows, mtw, wmh;
};
static constexpr WoCfg kWoCfg[7] = {
    {0, 0, 0},    // 0: the narrow form above, in passes of 256 tokens
    {32, 1, 1},   // 1: 4 waves x (32 rows x 64 columns)
    {64, 2, 1},   // 2: 4 waves x (64 x 64)
    {128, 2, 2},  // 3: 8 waves x (64 x 64)
    {256, 4, 2},  // 4: 8 waves x (128 x 64)
    {64, 2, 3},   // 5: 8 waves x (64 x 64), the second four on the second k step of every stage ("K halves")
    {128, 4, 3},  // 6: 8 waves x (128 x 64), K halves: every weight dequantised once per workgroup
};
// (Measured and dropped: 4 "fat" waves x (128 x 64) and 4 x (256 x 64), one wave per SIMD, every weight dequantised once
// per workgroup instead of once per row half -- equal to / 5-8 % slower than the 8-wave forms of the same height.)
struct WoWidePlan {
    int cfg; // index into kWoCfg; 0: the narrow form
    int ks;  // workgroups per tile along K
};

static std::atomic<int> g_wo_abl{0};
static std::atomic<int> g_wo_nt{1};    // non-temporal weight loads in the skinny form for weights of 32 MiB and more: 1 on (default), 0 off
static inline int64_t wo_nt_weight_bytes() { return (int64_t)32 << 20; }
static std::atomic<int> g_wo_skinny_decode{-1}; // 2..4 tokens through the skinny form: -1 where it wins (measured), 0 never, 1 always
static std::atomic<int> g_wo_skinny{1}; // the skinny form up to 32 tokens: 1 automatic, 0 off, 2..5 a fixed shape (measurements)
static std::atomic<int> g_wo_twopass_tile{0}; // (measurements, with the form forced) row height of the second pass's tiles; 0: 256
static std::atomic<int> g_wo_twopass{-1}; // -1 automatic, 0 never, 1 whenever the shape allows it (measurements, tests)
static std::atomic<int> g_wo_rw{-1};   // weights through registers (configurations 1, 2, 5, 6): -1 automatic, 0 never, 1 always
static std::atomic<int> g_wo_form{-1}; // measurement knob: -1 automatic, else the configuration index
static std::atomic<int> g_wo_ks{-1};   // -1 automatic, else the K split of the wide form (where K allows)
void set_wo_force(int form, int ks)
{
    if (form == 500 || form == 501) { // non-temporal weight loads of the skinny form (large weights): 500 on (default), 501 off
        g_wo_nt.store(form == 500 ? 1 : 0);
        return;
    }
    if (form == 309 || form == 310) { // (measurements, up to 16 tokens) 16 columns per wave x 8 / 16 waves
        g_wo_skinny.store(form == 309 ? 6 : 7);
        return;
    }
    if (form >= 306 && form <= 308) { // decode batches of 2..4 tokens through the skinny form: always / never / automatic
        g_wo_skinny_decode.store(form == 306 ? 1 : form == 307 ? 0 : -1);
        return;
    }
    if (form >= 300 && form <= 305) { // skinny form: 300 automatic, 301 off, 302..305 a fixed shape (32x8, 32x16, 64x8, 64x16)
        g_wo_skinny.store(form == 300 ? 1 : form == 301 ? 0 : form - 300);
        return;
    }
    if (form >= 400 && form <= 402) { // register weights: automatic / never / always
        g_wo_rw.store(form == 400 ? -1 : form - 401);
        return;
    }
    if (form == 203 || form == 204) { // second pass of the two-pass form on 256- / 128-row tiles
        g_wo_twopass_tile.store(form == 203 ? 256 : 128);
        return;
    }
    if (form >= 200) { // 200 automatic, 201 never, 202 always
        g_wo_twopass.store(form == 200 ? -1 : form - 201);
        return;
    }
    if (form >= 100) {
        g_wo_abl.store(form - 100);
        return;
    }
    if (form == -1) g_wo_abl.store(0), g_wo_twopass.store(-1), g_wo_twopass_tile.store(0), g_wo_skinny.store(1), g_wo_skinny_decode.store(-1), g_wo_rw.store(-1);
    if (form >= -1 && form <= 6) g_wo_form.store(form);
    if (ks >= -1) g_wo_ks.store(ks);
}

// Plan = the (configuration, K split) with the smallest estimated time.  Constants from tools/w8a16_forms.sh on MI355X
// (profiles/r02_w8a16_forms.txt), microseconds per 64-k stage with the whole chip busy: 0.61 / 0.75 / 1.25 / 2.1 for 32- /
// 64- / 128- / 256-row tiles (the dequantisation VALU work is amortised over the rows of a wave: taller is cheaper per
// row), ~15 % less while at most 3/4 of the CUs are busy (higher clocks under the power cap); the hand-over of a K split
// costs c0 + c1 ks us, growing with the tile (the last arriver reads ks tiles).  K is split only while all workgroups
// fit one wave (which also bounds the scratch: at most one 256-KiB slot per CU).  Up to 32 tokens the narrow form's
// register-streamed weights are as fast or faster (it also serves when the caller gives no scratch: short-M wide tiles
// rely on the K split to fill the chip).
static WoWidePlan wo_wide_plan(int M, int N, int K, bool have_scratch)
{
    const int form = g_wo_form.load(), fks = g_wo_ks.load();
    if (form == 0 || (form < 0 && (M <= 32 || (M <= 256 && !have_scratch)))) return {0, 1};
    // (49..64 tokens on wide outputs with short K: the narrow form's two 32-row MFMA tiles per wave still beat the 64-row wide
    //  tiles by ~8 %: 12288 x 4096 22.1 vs 24.1 us, 16384 x 4096 24.0 vs 26.0; not at K = 8192)
    if (form < 0 && M > 48 && M <= 64 && N >= 10240 && K <= 6144) return {0, 1};
    const int cus = num_cus(), tn = (N + wo::BNW - 1) / wo::BNW, nst = K / wo::KBW;
    // (round 3: 32- / 64-row tiles with their weights through registers: -9 / -8 %; K halves -2 %)
    static constexpr float kStage[7] = {0.f, 0.56f, 0.69f, 1.25f, 2.1f, 0.73f, 1.11f};
    static constexpr float kHand0[7] = {0.f, 2.f, 3.f, 10.f, 12.f, 3.f, 10.f}, kHand1[7] = {0.f, 0.8f, 2.1f, 2.5f, 4.3f, 2.1f, 2.5f};
    WoWidePlan best{3, 1};
    float best_t = 1e30f;
    for (int c = 1; c <= (form >= 5 ? form : 4); ++c) {
        // (automatic: the 128-row tiles run their K-halves form, configuration 6 -- every weight dequantised once per workgroup:
        //  -7..-11 % against configuration 3 from 257 to 1024 tokens on the long / wide shapes, level on 4096 x 4096)
        const int cfg = form < 0 && c == 3 ? 6 : c;
        if (form > 0 && cfg != form) continue;
        const int rows = kWoCfg[cfg].rows, tiles = ((M + rows - 1) / rows) * tn;
        if (form < 0 && rows >= 2 * M && cfg > 1) break; // (a tile twice as tall as the problem)
        const int per = rows <= 64 ? 4 : 8; // at least this many stages per workgroup
        int ks_hi = !have_scratch ? 1 : nst / per;
        if (ks_hi > (rows <= 64 ? 16 : 8)) ks_hi = rows <= 64 ? 16 : 8;
        if (ks_hi < 1 || (size_t)tiles * sizeof(unsigned) > kSplitkWordsBytes) ks_hi = 1; // (one hand-over word per tile)
        for (int ks = 1; ks <= ks_hi; ++ks) {
            if (fks > 0) { // (measurements, tests) that split wherever K allows it, whatever the tile count
                if (ks != (fks <= ks_hi ? fks : 1)) continue;
            } else if (ks > 1 && tiles * ks > cus) {
                break;
            }
            const int wgs = tiles * ks, waves = (wgs + cus - 1) / cus;
            const float load = 4 * wgs <= 3 * cus ? 0.85f : 1.f;
            const float hand = ks == 1 ? 0.f : kHand0[cfg] + kHand1[cfg] * ks;
            const float t = (float)waves * (float)((nst + ks - 1) / ks) * kStage[cfg] * load + hand;
            if (t < best_t) best_t = t, best = WoWidePlan{cfg, ks};
        }
    }
    // 64-row tiles that are split at most 2 ways run their 8-wave form (the second four waves on the second k step of every
    // stage): -8 % at 192 / 256 tokens on 12288 x 4096, -4..-7 % on 28672 x 8192, -8..-13 % from 384 tokens on the 4096-wide
    // shapes whatever the split; below 128 tokens with deeper splits the 4-wave form stays (4096 x 4096 at 64: 20.2 vs 20.7 us).
    if (form < 0 && best.cfg == 2 && (best.ks <= 2 || M >= 128)) best.cfg = 5;
    return best;
}

static size_t wo_wide_workspace(int M, int N, int K)
{
    const WoWidePlan pl = wo_wide_plan(M, N, K, true);
    if (pl.cfg == 0 || pl.ks <= 1) return 0;
    const int rows = kWoCfg[pl.cfg].rows;
    const size_t tiles = (size_t)((M + rows - 1) / rows) * ((N + wo::BNW - 1) / wo::BNW);
    return kSplitkWordsBytes + tiles * pl.ks * (size_t)rows * wo::BNW * sizeof(float);
}

// Skinny form (5..32 tokens): which shape, or 0 for the narrow form.  Its cost is weight bytes + token bytes through the
// same L2 -> CU path (every wave reads the token rows of its K range: M / 16 bytes of tokens per weight byte with 32
// columns per wave, M / 32 with 64), the narrow form's is flat in M but carries the K-split hand-over; crossovers measured
// with tools/w8a16_bench.py (profiles/r02_w8a16_gemm.txt): 2 = 32 columns x 8 waves, 3 = 32 x 16, 4 = 64 x 8, 5 = 64 x 16.
static int wo_skinny_pick(int M, int N, int K)
{
    const int v = g_wo_skinny.load();
    if (v == 0 || M > (v > 1 ? 64 : 48) || M < 1 || g_wo_form.load() >= 0 || g_wo_twopass.load() == 1) return 0; // (a forced form switches it off)
    const bool cols16 = N >= 2048 && (N + 15) / 16 <= 288; // 16 columns per wave give one workgroup per CU or a little less
    if (M <= 4) { // decode: the reference's GEMV band (decode_kernels.hip), whose cost grows ~1.4 us per token on wide
                  // outputs (12288 x 4096: 9.4 / 10.7 / 12.0 / 13.3 us for 1..4 tokens against 8.2 / 8.4 / 8.6 / 9.1 here) and
                  // on long K (4096 x 11008: 8.6 / 9.9 / - / 13.3 against 8.1 / 8.3 / - / 9.0 with 16 columns x 16 waves);
                  // short narrow shapes stay on the GEMV for one or two tokens (4096 x 4096: 4.0 / 4.6 vs 4.1 / 4.4)
        const int d = g_wo_skinny_decode.load();
        if (d == 0) return 0;
        if (d < 0 && !(N >= 8192 || (cols16 && (K >= 8192 || M >= 3)) || (!cols16 && N >= 4608 && M >= 2))) return 0;
    }
    if (v > 1) return v;
    const double mb = (double)N * K * 1e-6;
    if (M > 32) // 33..48 tokens, three token tiles: 12288 x 4096 24 -> 16-19 us, 4096 x 4096 17 -> 12-15; slower elsewhere
        return N >= 8192 && N < 16384 ? 4 : N < 8192 && mb <= 20. ? 2 : 0;
    if (N >= 16384) return M <= 24 ? 4 : 0;
    if (N >= 8192) return M <= 12 ? 2 : 5;
    if (M <= 16 && (N + 15) / 16 <= 288) return 7; // 16 columns x 16 waves (1280 x 8192 at 8 tokens: 9.7 -> 8.0 us too): 4096 x 4096 6.3 -> 5.5 us at 8 tokens, 4096 x 11008 13.0 -> 10.6
    if (M <= 6 && K >= 8192) return 3;
    if (M <= 16) return 2;
    if (M <= 24) return mb <= 50. ? 2 : 0;
    return mb <= 20. ? 2 : 0;
}

// Two-pass form: for large problems, if the
// caller's scratch holds the fp16 image of W (N * K * 2 bytes) and the rows of the output are 16-byte aligned (N % 8 == 0).
// Measured crossover (profiles/r02_w8a16_gemm.txt): the second pass runs 256 x 256 tiles only, so it needs about a full
// wave of them (>= 200) besides enough rows to amortise the dequantisation pass.
constexpr int kTwoPassMinM = 1280;
// Row height of the second pass's tiles, or 0: not the two-pass form.  (128-row tiles -- gemm_w8a8o16_pp128_kernel -- are
// wired for measurements only: where they fill one wave of the chip they measured -8 % once (5120 x 5120 at 1024 tokens),
// +-2 % elsewhere and +9 % on 12288 x 4096 at 512 tokens, inside the run-to-run spread of back-to-back timings.)
static int wo_two_pass_tile(int M, int N, int K)
{
    const int f = g_wo_twopass.load();
    if (f == 0 || M <= 4 || N % 8 || K % 64) return 0;
    const int forced_tile = g_wo_twopass_tile.load();
    if (f == 1) return forced_tile ? forced_tile : 256;
    const int64_t t256 = (int64_t)((M + 255) / 256) * ((N + 255) / 256);
    return M >= kTwoPassMinM && t256 >= 200 ? 256 : 0;
}
static bool wo_two_pass_wanted(int M, int N, int K) { return wo_two_pass_tile(M, N, K) != 0; }
// (the image sits BEHIND the hand-over words of the K splits, which every call must leave zero)
static size_t wo_two_pass_bytes(int N, int K) { return kSplitkWordsBytes + (size_t)N * K * 2; }

bool w8a16_skinny_takes(int M, int N, int K) { return wo_skinny_pick(M, N, K) != 0; }

size_t w8a16_gemm_workspace_size(int M, int N, int K)
{
    if (N <= 0 || K <= 0) return 0;
    if (wo_skinny_pick(M, N, K) != 0) return 0; // (K is split inside the workgroup)
    if (M <= 4) return 0;
    if (wo_two_pass_wanted(M, N, K)) { // (a smaller scratch still serves the wide form's K split, or none)
        const size_t wide = wo_wide_plan(M, N, K, true).cfg != 0 ? wo_wide_workspace(M, N, K) : 0;
        return wo_two_pass_bytes(N, K) > wide ? wo_two_pass_bytes(N, K) : wide;
    }
    if (wo_wide_plan(M, N, K, true).cfg != 0) return wo_wide_workspace(M, N, K);
    // narrow form: passes of at most 256 tokens; a ragged last pass may plan (and size) differently
    const size_t full = wo_narrow_workspace(M < 256 ? M : 256, N, K);
    const size_t tail = (M > 256 && M % 256) ? wo_narrow_workspace(M % 256, N, K) : 0;
    return full > tail ? full : tail;
}

template <int MT, int NST, int WM = 1>
static hipError_t launch_wo(const uint16_t* A, const uint8_t* Wq, const uint16_t* scale, uint16_t* Out, int M, int N,
                            int K, int ks, void* scratch, const void* zeros, hipStream_t st)
{
    constexpr size_t lds = (size_t)NST * MT * 32 * wo::ROWB;
    static_assert(lds <= 160 * 1024 - 64, "LDS budget");
    auto kern = w8a16_gemm_kernel<MT, NST, WM>;
    static DeviceOnce once;
    if (hipError_t e = ensure_dynamic_lds(kern, lds, once); e != hipSuccess) return e;
    const int ntiles = (N + wo::BN - 1) / wo::BN;
    hipLaunchKernelGGL(kern, dim3((unsigned)(ntiles * ks)), dim3(256 * WM), lds, st, A, Wq, scale, Out, M, N, K, ks, scratch,
                       zeros);
    return hipGetLastError();
}

template <int MT, int KW, int CG>
static hipError_t launch_wo_skinny(const uint16_t* A, const uint8_t* Wq, const uint16_t* scale, uint16_t* Out, int M, int N,
                                   int K, hipStream_t st)
{
    const dim3 grid((unsigned)((N + 16 * CG - 1) / (16 * CG))), block(KW * 64);
    if constexpr (MT == 1) { // (decode and the smallest decode batches: the forms a model's decode step runs)
        if (g_wo_nt.load() != 0 && (int64_t)N * K >= wo_nt_weight_bytes()) {
            hipLaunchKernelGGL((w8a16_skinny_kernel<MT, KW, CG, true>), grid, block, 0, st, A, Wq, scale, Out, M, N, K);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL((w8a16_skinny_kernel<MT, KW, CG>), grid, block, 0, st, A, Wq, scale, Out, M, N, K);
    return hipGetLastError();
}

template <int MTW, int WMH, int NST, int ABL = 0, bool RW = false>
static hipError_t launch_wo_wide(const uint16_t* A, const uint8_t* Wq, const uint16_t* scale, uint16_t* Out, int M, int N,
                                 int K, int ks, void* scratch, hipStream_t st)
{
    constexpr int rows = 32 * MTW * (WMH == 3 ? 1 : WMH);
    constexpr size_t stages = (size_t)NST * (rows * wo::ROWBW + (RW ? 0 : wo::WSTAGE));
    constexpr size_t merge = WMH == 3 ? (size_t)2 * MTW * 16 * 256 * sizeof(float) : 0; // K halves meet in LDS after the loop
    constexpr size_t lds = stages > merge ? stages : merge;
    static_assert(lds <= 160 * 1024 - 64, "LDS budget");
    auto kern = w8a16_gemm_wide_kernel<MTW, WMH, NST, ABL, RW>;
    static DeviceOnce once;
    if (hipError_t e = ensure_dynamic_lds(kern, lds, once); e != hipSuccess) return e;
    const int tiles = ((M + rows - 1) / rows) * ((N + wo::BNW - 1) / wo::BNW);
    hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * ks)), dim3(256 * (WMH == 3 ? 2 : WMH)), lds, st, A, Wq, scale, Out, M, N, K, ks,
                       scratch);
    return hipGetLastError();
}

hipError_t launch_w8a16_gemm(const void* A, const uint8_t* Wq, const void* scale, void* Out, int M, int N, int K,
                             void* scratch, size_t scratch_bytes, const void* zeros, hipStream_t st)
{
    if (M <= 0 || N <= 0) return hipSuccess;
    const uint16_t* a = static_cast<const uint16_t*>(A);
    const uint16_t* s = static_cast<const uint16_t*>(scale);
    uint16_t* o = static_cast<uint16_t*>(Out);
    if (wo_two_pass_wanted(M, N, K) && scratch != nullptr && scratch_bytes >= wo_two_pass_bytes(N, K)) {
        const int64_t groups = (int64_t)(N >> 1) * (K >> 3);
        uint16_t* const wf = reinterpret_cast<uint16_t*>(static_cast<char*>(scratch) + kSplitkWordsBytes);
        hipLaunchKernelGGL(w8a16_dequant_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, Wq, s, wf, N, K);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
        if (wo_two_pass_tile(M, N, K) == 128) return launch_gemm_f16_pp128(A, wf, Out, M, N, K, zeros, st);
        return launch_gemm_f16_pp(A, wf, Out, M, N, K, zeros, st);
    }
    if (const int pick = wo_skinny_pick(M, N, K); pick != 0) {
        if (M <= 16) {
            switch (pick) {
            case 2: return launch_wo_skinny<1, 8, 2>(a, Wq, s, o, M, N, K, st);
            case 3: return launch_wo_skinny<1, 16, 2>(a, Wq, s, o, M, N, K, st);
            case 4: return launch_wo_skinny<1, 8, 4>(a, Wq, s, o, M, N, K, st);
            case 6: return launch_wo_skinny<1, 8, 1>(a, Wq, s, o, M, N, K, st);
            case 7: return launch_wo_skinny<1, 16, 1>(a, Wq, s, o, M, N, K, st);
            default: return launch_wo_skinny<1, 16, 4>(a, Wq, s, o, M, N, K, st);
            }
        }
        if (M <= 32) {
            switch (pick) {
            case 2: return launch_wo_skinny<2, 8, 2>(a, Wq, s, o, M, N, K, st);
            case 3: return launch_wo_skinny<2, 16, 2>(a, Wq, s, o, M, N, K, st);
            case 4: return launch_wo_skinny<2, 8, 4>(a, Wq, s, o, M, N, K, st);
            default: return launch_wo_skinny<2, 16, 4>(a, Wq, s, o, M, N, K, st);
            }
        }
        if (M <= 48) return pick <= 3 ? launch_wo_skinny<3, 8, 2>(a, Wq, s, o, M, N, K, st) : launch_wo_skinny<3, 8, 4>(a, Wq, s, o, M, N, K, st);
        return pick <= 3 ? launch_wo_skinny<4, 8, 2>(a, Wq, s, o, M, N, K, st) : launch_wo_skinny<4, 8, 4>(a, Wq, s, o, M, N, K, st);
    }
    const size_t need_wide = wo_wide_workspace(M, N, K); // 0: the plan with scratch is the narrow form or does not split K
    const WoWidePlan pl = wo_wide_plan(M, N, K, need_wide == 0 || (scratch != nullptr && scratch_bytes >= need_wide));
    if (pl.cfg != 0) {
        if (pl.cfg == 4) {
            switch (g_wo_abl.load()) { // measurement-only ablations of the 256-row form
            case 1: return launch_wo_wide<4, 2, 3, 1>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
            case 2: return launch_wo_wide<4, 2, 3, 2>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
            case 4: return launch_wo_wide<4, 2, 3, 4>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
            case 8: return launch_wo_wide<4, 2, 3, 8>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
            case 7: return launch_wo_wide<4, 2, 3, 7>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
            case 14: return launch_wo_wide<4, 2, 3, 14>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
            default: break;
            }
        }
        if (pl.cfg == 6) {
            switch (g_wo_abl.load()) { // the same ablations for the 128-row K-halves form
            case 1: return launch_wo_wide<4, 3, 4, 1>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
            case 2: return launch_wo_wide<4, 3, 4, 2>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
            case 4: return launch_wo_wide<4, 3, 4, 4>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
            case 8: return launch_wo_wide<4, 3, 4, 8>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
            case 7: return launch_wo_wide<4, 3, 4, 7>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
            case 14: return launch_wo_wide<4, 3, 4, 14>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
            default: break;
            }
        }
        // weights through registers wherever the configuration allows it (automatic): bit-identical, -7..-10 % on the 4-wave
        // 64-row tiles, -0..-6 % on the K-halves forms (profiles/r03_w8a16_register_weights_probe.txt)
        if (g_wo_rw.load() != 0 && g_wo_abl.load() == 0) {
            switch (pl.cfg) {
            case 1: return launch_wo_wide<1, 1, 6, 0, true>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
            case 2: return launch_wo_wide<2, 1, 6, 0, true>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
            case 5: return launch_wo_wide<2, 3, 6, 0, true>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
            case 6: return launch_wo_wide<4, 3, 4, 0, true>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
            default: break;
            }
        }
        switch (pl.cfg) {
        case 1: return launch_wo_wide<1, 1, 6>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
        case 2: return launch_wo_wide<2, 1, 6>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
        case 3: return launch_wo_wide<2, 2, 4>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
        case 5: return launch_wo_wide<2, 3, 6>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
        case 6: return launch_wo_wide<4, 3, 4>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
        default: return launch_wo_wide<4, 2, 3>(a, Wq, s, o, M, N, K, pl.ks, scratch, st);
        }
    }
    // the narrow form addresses the weights with an absolute 32-bit per-lane offset (w8a16_gemm_kernel: wlane): a weight of 4 GiB
    // or more (a 256k-entry head on K = 16384) would wrap silently -- refuse it (the wide / skinny forms use 64-bit bases)
    if ((int64_t)N * (int64_t)K >= ((int64_t)1 << 32)) return hipErrorInvalidValue;
    for (int m0 = 0; m0 < M; m0 += 256) { // 256-token passes (each streams the weights once)
        const int rows = M - m0 < 256 ? M - m0 : 256;
        const size_t need = wo_narrow_workspace(rows, N, K);
        const bool have = scratch != nullptr && need != 0 && scratch_bytes >= need;
        const WoPlan pl = wo_plan(rows, N, K, have);
        const uint16_t* ap = a + (int64_t)m0 * K;
        uint16_t* op = o + (int64_t)m0 * N;
        hipError_t e;
        switch (pl.mt) {
        case 1: e = launch_wo<1, 4>(ap, Wq, s, op, rows, N, K, pl.ks, scratch, zeros, st); break;
        case 2: e = launch_wo<2, 4>(ap, Wq, s, op, rows, N, K, pl.ks, scratch, zeros, st); break;
        case 4: e = launch_wo<4, 3, 2>(ap, Wq, s, op, rows, N, K, pl.ks, scratch, zeros, st); break;
        default: e = launch_wo<8, 2, 2>(ap, Wq, s, op, rows, N, K, pl.ks, scratch, zeros, st); break;
        }
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

} // namespace mixq

// Ping-pong ("8-phase") variant of the fused W8A8O16 GEMM for large problems on gfx950.
//
// Same math, same operand roles, same epilogue as gemm_kernels.hip (see its header for the reference lines replaced);
// what changes is the main-loop schedule, built for one 512-thread workgroup per CU (256 x 256 output tile, 128 KiB LDS):
//
//   * the 8 waves form two groups of 4 (one wave of each group on every SIMD).  A K slice (128 B per row) is processed
//     in 4 phases; every phase is a LOAD segment (ds_read_b128 of the next fragments + 2 global_load_lds of a later
//     slice) followed by a COMPUTE segment (8 x v_mfma_i32_32x32x32_i8), separated by s_barrier.  Group 1 runs one
//     segment behind group 0, so on each SIMD one wave is always in its MFMA segment while its partner fetches:
//     the matrix pipe never waits for LDS or HBM latency.
//   * global -> LDS copies are never drained inside the loop: each wave waits `vmcnt(4)` at the end of a LOAD segment,
//     i.e. only for copies issued two segments earlier; the two most recent pairs stay in flight across barriers.
//   * wave tile = 128 (m) x 64 (n) = 4 x 2 MFMA tiles.  Phase order (m-half, n-half): (0,0) (0,1) (1,1) (1,0), so each
//     LOAD segment fetches 8 or 4 fragments: Y0 | X1 | Y1 | X0-of-the-next-slice.  Two slices are unrolled per loop
//     iteration so the two X fragment sets swap roles with static register names.
//   * LDS image per buffer: [X half 0 | X half 1 | Y half 0 | Y half 1], 16 KiB each, rows ordered
//     [half][wave][row] so that every region is one contiguous run of 128-byte rows; 16-B slot = chunk ^ ((row>>1)&7).
//
// Hazard bookkeeping (slots are per-wave LOAD segments; group 1 lags by one segment):
//   RAW  a region issued at slot s is first read at slot s+3; every wave has passed `vmcnt(4)` at the end of slot
//        s+2 (which retires everything issued up to slot s) and a barrier lies between.
//   WAR  a region is re-issued 5 slots after its last read.
#include "mixq_device.h"
#include "mixq_launch.h"

namespace mixq {

namespace pp {
constexpr int BM = 256, BN = 256, T = 512;
constexpr int KS = 128;                 // K bytes per row per slice
constexpr int REGION = 128 * KS;        // 16 KiB: 128 rows
constexpr int BUF = 4 * REGION;         // 64 KiB per slice buffer
constexpr int X0 = 0, X1 = REGION, Y0 = 2 * REGION, Y1 = 3 * REGION;
constexpr int OSLICE = 256;

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
} // namespace pp

// ABL: measurement-only ablations (wrong results): 1 = no global_load_lds in the loop, 2 = no vmcnt waits,
// 4 = no ds_reads in the loop.  ABL = 0 is the product kernel.
template <int EPI, int ABL = 0>
__global__ __launch_bounds__(512) void gemm_w8a8o16_pp_kernel(const GemmParams p)
{
    using namespace pp;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2;          // 0: waves 0-3, 1: waves 4-7 (one of each per SIMD)
    const int wm = wave & 1;              // 2 wave rows along m (128 each)
    const int wn = wave >> 1;             // 4 wave columns along n (64 each); wn = 0..3 mixes both groups
    const int lr = lane & 31, lh = lane >> 5;

    // ---- block -> tile mapping (XCD-aware, grouped; identical to gemm_kernels.hip) -------------------
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    int t_lin;
    {
        const int bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
        t_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
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
    const int64_t K = p.K;
    const char* src[4][2];
    int koff_src;
    {
        const int slot = tid & 7;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = i * 64 + (tid >> 3);
                const int sw = (q >> 1) & 7;
                const int nl = (q >> 5) * 64 + h * 32 + (q & 31);
                const int ml = (q >> 6) * 128 + h * 64 + (q & 63);
                const int gn = min(n0 + nl, p.N - 1), gm = min(m0 + ml, p.M - 1);
                src[h][i] = reinterpret_cast<const char*>(p.B) + (int64_t)gn * K + ((slot ^ sw) << 4);
                src[2 + h][i] = reinterpret_cast<const char*>(p.A) + (int64_t)gm * K + ((slot ^ sw) << 4);
            }
        koff_src = (slot ^ (((tid >> 3) >> 1) & 7)) << 4; // same for i = 0,1 (64 rows apart)
    }
    const int nk = (p.K + KS - 1) / KS;
    const bool ktail = (p.K % KS) != 0;

    auto issue = [&](int region, int kt) __attribute__((always_inline)) { // 2 x global_load_lds: region `region` of slice kt
        if (ABL & 1) return;
        char* dst = smem + (kt & 1) * BUF + region * REGION;
        const int64_t kbyte = (int64_t)kt * KS;
        const bool oob = ktail && (kt == nk - 1) && (kbyte + koff_src >= K);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char* s = oob ? static_cast<const char*>(p.zeros) : src[region][i] + kbyte;
            glds16(s, dst + (i * T + wave * 64) * 16);
        }
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
                acc[xi][yhalf * 2 + jy] =
                    __builtin_amdgcn_mfma_i32_32x32x32_i8(X[ks], Y[jy][ks], acc[xi][yhalf * 2 + jy], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };

    // One K slice.  Xcur holds X half 0 of this slice on entry; Xoth receives X half 1, then X half 0 of slice kt+1.
    auto slice = [&](v4i (&Xcur)[4], v4i (&Xoth)[4], int kt) __attribute__((always_inline)) {
        const bool more = kt + 1 < nk; // wave-uniform
        // phase 1: (Y0, X0)
        read_y(kt, 0);
        if (more) issue(0, kt + 1);
        if (!(ABL & 2)) { if (more) wait_vmcnt<4>(); else wait_vmcnt<2>(); }
        MIXQ_SEG_END();
        mma(Xcur, 0, 0);
        MIXQ_SEG_END();
        // phase 2: (Y0, X1)
        read_x(Xoth, kt, 1);
        if (more) issue(2, kt + 1);
        if (!(ABL & 2)) { if (more) wait_vmcnt<4>(); else wait_vmcnt<0>(); }
        MIXQ_SEG_END();
        mma(Xoth, 1, 0);
        MIXQ_SEG_END();
        // phase 3: (Y1, X1)
        read_y(kt, 1);
        if (more) issue(1, kt + 1);
        if (more && !(ABL & 2)) wait_vmcnt<4>();
        MIXQ_SEG_END();
        mma(Xoth, 1, 1);
        MIXQ_SEG_END();
        // phase 4: (Y1, X0); the LOAD segment already fetches X half 0 of the next slice into the free set
        if (more) read_x(Xoth, kt + 1, 0);
        if (more) issue(3, kt + 1);
        if (more && !(ABL & 2)) wait_vmcnt<4>();
        MIXQ_SEG_END();
        mma(Xcur, 0, 1);
        MIXQ_SEG_END();
    };

    // ---- prologue: slice 0 completely, then stagger the groups ----------------------------------------------
    issue(0, 0);
    issue(2, 0);
    issue(1, 0);
    issue(3, 0);
    wait_vmcnt<0>();
    MIXQ_SEG_END();
    read_x(XA, 0, 0);
    if (group == 1) MIXQ_SEG_END(); // group 1 now runs one segment behind group 0

    for (int kt = 0; kt < nk; kt += 2) {
        slice(XA, XB, kt);
        if (kt + 1 < nk) slice(XB, XA, kt + 1);
    }
    if (group == 0) MIXQ_SEG_END(); // re-align the groups

    // ---- outlier side GEMM + epilogue (same as the 2-barrier kernel) -------------------------------------------
    const bool has_outliers = (EPI != EPI_INT32) && p.O > 0;
    if (has_outliers) {
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

    const int osteps = has_outliers ? (p.O + 15) / 16 : 0;

    if (EPI == EPI_INT32) { // debug / unfused API: raw accumulators, 16-byte stores straight from the MFMA layout
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

    // ---- dequant math, tile by tile, results packed to fp16 in registers (acc registers die as we go) ----------
    uint2 outp[2][4][4]; // [n tile][m tile][quad] : 4 consecutive n for row m
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // acc[i][j]: n tile = wn*64 + i*32, m tile = wm*128 + j*32   (i = X half, j = Y half*2 + jy)
            const int m = min(m0 + wm * 128 + j * 32 + lr, p.M - 1); // clamped rows are computed but never stored
            const int nb0 = n0 + wn * 64 + i * 32 + 4 * lh;
            v16f P;
#pragma unroll
            for (int e = 0; e < 16; ++e) P[e] = 0.f;
            if (has_outliers) {
                const char* xo = smem + (wn * 64 + i * 32 + lr) * OSLICE;
                const char* yo = smem + BN * OSLICE + (wm * 128 + j * 32 + lr) * OSLICE;
                const int sw16 = lr & 15;
                for (int ks = 0; ks < osteps; ++ks) {
                    const int off = ((ks * 2 + lh) ^ sw16) << 4;
                    v8h xf = *reinterpret_cast<const v8h*>(xo + off);
                    v8h yf = *reinterpret_cast<const v8h*>(yo + off);
                    P = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf, yf, P, 0, 0, 0);
                }
            }
            const float sa = h2f(p.sA[m]);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nb = min(nb0 + 8 * g, p.N - 4);
                const uint2 swb = *reinterpret_cast<const uint2*>(p.sW + nb);
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
                    const float c = has_outliers ? h2f(f2h_bits_of_f32_result(P[4 * g + e])) : h2f(yh[e]);
                    float v = __builtin_fmaf((float)acc[i][j][4 * g + e], h2f(swh[e]) * sa, c);
                    if (EPI == EPI_DEQUANT_SILU) v = v / (1.f + __expf(-v));
                    oh[e] = f2h_bits_of_f32_result(v);
                }
                outp[i][j][g].x = (unsigned)oh[0] | ((unsigned)oh[1] << 16);
                outp[i][j][g].y = (unsigned)oh[2] | ((unsigned)oh[3] << 16);
            }
        }
    }

    // ---- stage the 256 x 256 fp16 tile in LDS (512-byte rows, 16-B chunk c of row r at chunk c ^ (r & 31)), then
    //      write it out as whole rows: every store instruction covers two complete 512-byte row segments ----------
    __syncthreads(); // everyone is done with the outlier operands in LDS
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = wm * 128 + j * 32 + lr;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = wn * 8 + i * 4 + g; // 16-byte chunk index inside the row; lh picks its 8-byte half
                *reinterpret_cast<uint2*>(smem + r * 512 + ((c ^ (r & 31)) << 4) + lh * 8) = outp[i][j][g];
            }
        }
    __syncthreads();
    {
        uint16_t* D = static_cast<uint16_t*>(p.D);
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int q = it * T + tid;
            const int r = q >> 5, c = q & 31;
            const uint4 v = *reinterpret_cast<const uint4*>(smem + r * 512 + ((c ^ (r & 31)) << 4));
            const int m = m0 + r, n = n0 + c * 8;
            if (m < p.M && n < p.N) *reinterpret_cast<uint4*>(D + (int64_t)m * p.N + n) = v;
        }
    }
}

template <int EPI, int ABL = 0>
static hipError_t launch_pp_epi(const GemmParams& p, hipStream_t st)
{
    constexpr size_t lds = 2 * (size_t)pp::BUF;
    auto kern = gemm_w8a8o16_pp_kernel<EPI, ABL>;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    const int tiles = ((p.M + pp::BM - 1) / pp::BM) * ((p.N + pp::BN - 1) / pp::BN);
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(pp::T), lds, st, p);
    return hipGetLastError();
}

hipError_t launch_gemm_pp_ablate(const GemmParams& p, int abl, hipStream_t st)
{
    switch (abl) {
    case 1: return launch_pp_epi<EPI_DEQUANT, 1>(p, st);
    case 2: return launch_pp_epi<EPI_DEQUANT, 2>(p, st);
    case 3: return launch_pp_epi<EPI_DEQUANT, 3>(p, st);
    case 4: return launch_pp_epi<EPI_DEQUANT, 4>(p, st);
    case 5: return launch_pp_epi<EPI_DEQUANT, 5>(p, st);
    case 7: return launch_pp_epi<EPI_DEQUANT, 7>(p, st);
    default: return launch_pp_epi<EPI_DEQUANT, 0>(p, st);
    }
}

hipError_t launch_gemm_pp(const GemmParams& p, int epi, hipStream_t st)
{
    switch (epi) {
    case EPI_DEQUANT: return launch_pp_epi<EPI_DEQUANT>(p, st);
    case EPI_DEQUANT_SILU: return launch_pp_epi<EPI_DEQUANT_SILU>(p, st);
    default: return launch_pp_epi<EPI_INT32>(p, st);
    }
}

} // namespace mixq

// Persistent form of the ping-pong W8A8O16 GEMM (gemm_pp_kernels.hip): one 512-thread workgroup per CU walks a list of
// 256 x 256 output tiles, and the work that used to sit exposed between two tiles is folded under MFMA work:
//
//   * LAST K slice of a tile: its LOAD segments have no next slice to fetch, so they fetch the outlier operands of
//     the epilogue instead -- the fp16 activation-outlier tile (64 KiB) into the slice buffer that is already free and
//     the first half of the fp16 outlier-weight tile (32 KiB) into the 32-KiB LDS tail (160 KiB are used in total).
//   * after the loop: the second half of the outlier weights is fetched into the other slice buffer while the first
//     four 32x32 tiles are being dequantised; the first K slice of the NEXT output tile is fetched while the fp16
//     results are staged through LDS and stored, so the next main loop starts with its operands resident.
//   * the main loop itself is the ping-pong schedule of gemm_pp_kernels.hip (see that file for the hazard argument);
//     slice kt of every tile lives in buffer kt & 1 (all LDS offsets are immediates).
//
// Reference lines replaced: see gemm_kernels.hip.  Results are bit-identical to both other schedules (tests).
#include "mixq_device.h"
#include "mixq_launch.h"
#include <type_traits>

namespace mixq {

namespace pp2 {
constexpr int BM = 256, BN = 256, T = 512;
constexpr int KS = 128;                 // K bytes per row per slice
constexpr int REGION = 128 * KS;        // 16 KiB: 128 rows
constexpr int BUF = 4 * REGION;         // 64 KiB per slice buffer
constexpr int X0 = 0, X1 = REGION, Y0 = 2 * REGION, Y1 = 3 * REGION;
constexpr int SPARE = 2 * BUF;          // 32 KiB tail: outlier weights of n-half 0
constexpr int LDS_BYTES = 2 * BUF + 32768;
constexpr int OSLICE = 256;             // bytes per LDS row of an outlier operand (128 fp16)

enum { FIRST = 0, STEADY = 1, TAIL = 2 };

typedef float v2f __attribute__((ext_vector_type(2)));

#define MIXQ2_SEG_END()                                 \
    do {                                                \
        __builtin_amdgcn_sched_barrier(0);              \
        asm volatile("s_barrier" ::: "memory");        \
        __builtin_amdgcn_sched_barrier(0);              \
    } while (0)

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// 16-byte LDS-DMA from inline asm (never counted by the compiler; every consumer sits behind an explicit vmcnt wait
// and a barrier).  LDS destination = M0 (wave-uniform) + lane*16; M0 is saved/restored inside the statement.
__device__ __forceinline__ void dma16_sbase(const char* sbase, unsigned voff, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %3\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %2\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}
__device__ __forceinline__ void dma16_flat(const void* gptr, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\t"
                 "s_mov_b32 m0, %2\n\t"
                 "s_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off\n\t"
                 "s_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gptr), "s"(lds_addr)
                 : "memory");
}
} // namespace pp2

template <int EPI, bool HAS_O, bool HAS_Y>
__global__ __launch_bounds__(512) void gemm_w8a8o16_pp2_kernel(const GemmParams p)
{
    using namespace pp2;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int group = wave >> 2;          // 0: waves 0-3, 1: waves 4-7 (one of each per SIMD)
    const int wm = wave & 1;              // 2 wave rows along m (128 each)
    const int wn = wave >> 1;             // 4 wave columns along n (64 each)
    (void)lane; // (fragment offsets are recomputed per tile from an opaque thread id: setup_frag_offsets)

    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = tiles_m * tiles_n;
    const int64_t K = p.K;
    const int nk = (p.K + KS - 1) / KS;   // >= 4 (checked by the launcher)
    const bool ktail = (p.K % KS) != 0;
    const unsigned lds_base = (unsigned)(size_t)(MIXQ_LDS_PTR(smem));
    const unsigned lds_wave = lds_base + wave * 1024; // this wave's 1-KiB window inside every 8-KiB DMA batch
    const int obytes = p.O * 2;

    // virtual block id v = blockIdx.x + round * gridDim.x  ->  tile (XCD-aware, grouped; same map as the other kernels)
    auto tile_of = [&](int v, int& m0, int& n0) __attribute__((always_inline)) {
        const int q = nwg >> 3, r = nwg & 7, xcd = v & 7;
        const int t_lin = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        constexpr int GROUP_M = 4;
        const int per_group = GROUP_M * tiles_n;
        const int g = t_lin / per_group, first_m = g * GROUP_M;
        const int gsz = min(tiles_m - first_m, GROUP_M);
        const int within = t_lin - g * per_group;
        m0 = (first_m + within % gsz) * BM;
        n0 = (within / gsz) * BN;
    };

    // ---- per-tile staging state ----------------------------------------------------------------------------
    int m0, n0;
    const char* baseA;
    const char* baseB;
    unsigned off[4][2];
    auto setup_tile = [&](int v) __attribute__((always_inline)) {
        tile_of(v, m0, n0);
        baseB = reinterpret_cast<const char*>(p.B) + (int64_t)n0 * K;
        baseA = reinterpret_cast<const char*>(p.A) + (int64_t)m0 * K;
        int st = tid; // opaque: nothing below is hoisted out of the tile loop (see setup_frag_offsets)
        asm volatile("" : "+v"(st));
        const int slot8 = st & 7;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = i * 64 + (st >> 3);
                const int sw = (q >> 1) & 7;
                const int nl = (q >> 5) * 64 + h * 32 + (q & 31);
                const int ml = (q >> 6) * 128 + h * 64 + (q & 63);
                const int rn = min(n0 + nl, p.N - 1) - n0, rm = min(m0 + ml, p.M - 1) - m0; // clamped rows, >= 0
                off[h][i] = (unsigned)rn * (unsigned)p.K + ((slot8 ^ sw) << 4);
                off[2 + h][i] = (unsigned)rm * (unsigned)p.K + ((slot8 ^ sw) << 4);
            }
    };

    // 2 x LDS-DMA: region `region` of slice kt.  tailchk: slice kt may be partial in K (chunks past K <- zero page).
    auto issue = [&](int region, int kt, bool tailchk) __attribute__((always_inline)) {
        const unsigned dst = lds_wave + (kt & 1) * BUF + region * REGION;
        const char* base = (region < 2 ? baseB : baseA) + (int64_t)kt * KS; // scalar
        if (tailchk && ktail && kt == nk - 1) {
            int tt = tid;
            asm volatile("" : "+v"(tt));
            const int koff_src = ((tt & 7) ^ (((tt >> 3) >> 1) & 7)) << 4;
            const bool oob = (int64_t)kt * KS + koff_src >= K;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const char* s = oob ? static_cast<const char*>(p.zeros) : base + off[region][i];
                dma16_flat(s, dst + i * T * 16);
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16_sbase(base, off[region][i], dst + i * T * 16);
    };

    // Outlier operands, 256-B LDS rows, 16-B slot = chunk ^ (row & 15); chunks past O come from the zero page.
    // fpA tile: 256 rows (natural m order) -> 4096 chunks = 8 per thread, 2 per call (part = 0..3)
    // (`t_` = an opaque copy of tid made at the point of use: keeps this address math out of the main loop's
    //  live ranges -- the compiler would otherwise hoist it above the tile loop and spill)
    auto issue_fpA = [&](int part, unsigned lds_dst, int t_) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = part * 2 + u;
            const int row = (i * T + t_) >> 4;
            const int c = ((t_ & 15) ^ (row & 15)) << 4;
            const int grow = min(m0 + row, p.M - 1);
            const char* s = reinterpret_cast<const char*>(p.fpA) + (int64_t)grow * obytes + c;
            if (c >= obytes) s = static_cast<const char*>(p.zeros);
            dma16_flat(s, lds_dst + lds_wave - lds_base + i * T * 16);
        }
    };
    // fpW half h: LDS row q = wn*32 + rr  <-  n_local = wn*64 + h*32 + rr ; 128 rows = 2048 chunks = 4 per thread
    auto issue_fpW = [&](int h, int i, unsigned lds_dst, int t_) __attribute__((always_inline)) {
        const int q = (i * T + t_) >> 4;
        const int c = ((t_ & 15) ^ (q & 15)) << 4;
        const int nl = (q >> 5) * 64 + h * 32 + (q & 31);
        const int grow = min(n0 + nl, p.N - 1);
        const char* s = reinterpret_cast<const char*>(p.fpW) + (int64_t)grow * obytes + c;
        if (c >= obytes) s = static_cast<const char*>(p.zeros);
        dma16_flat(s, lds_dst + lds_wave - lds_base + i * T * 16);
    };

    // ---- fragment read offsets ---------------------------------------------------------------------------
    // (recomputed from an opaque thread id at the top of every tile: values that stay live across the epilogue get
    //  spilled, and a scratch reload pending at the loop header makes the compiler put `s_waitcnt vmcnt(0)` INSIDE the
    //  steady loop, which drains the LDS-DMA queue every iteration)
    int koff[4], xrow, yrow;
    auto setup_frag_offsets = [&]() __attribute__((always_inline)) {
        int mt = tid;
        asm volatile("" : "+v"(mt));
        const int mlr = mt & 31, mlh = (mt >> 5) & 1;
        const int sw = (mlr >> 1) & 7;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) koff[ks] = ((ks * 2 + mlh) ^ sw) << 4;
        xrow = (wn * 32 + mlr) * KS;                      // + X0 / X1
        yrow = (wm * 64 + mlr) * KS;                      // + Y0 / Y1, + jy*32*KS
    };

    v4i XA[4], XB[4], Y[2][4];
    v16i acc[2][4]; // [n tile][m tile]

    auto read_x = [&](v4i (&X)[4], int kt, int half) __attribute__((always_inline)) {
        const char* b = smem + (kt & 1) * BUF + (half ? X1 : X0) + xrow;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) X[ks] = *reinterpret_cast<const v4i*>(b + koff[ks]);
    };
    auto read_y = [&](int kt, int half) __attribute__((always_inline)) {
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

    // One K slice (4 phases).  FIRST: slice 0 of a tile, everything it needs is resident, the first two waits are
    // skipped (the only older traffic is the previous tile's output stores).  STEADY: a full successor exists.
    // TAIL: runtime checks; the LAST slice fetches the epilogue's outlier operands instead of a successor.
    auto slice = [&](v4i (&Xcur)[4], v4i (&Xoth)[4], int kt, auto mode_tag) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_tag)::value;
        const bool more = (MODE != TAIL) || (kt + 1 < nk);  // wave-uniform
        const bool seam = HAS_O && !more;                   // last slice of the tile
        const unsigned fpa_dst = lds_base + ((kt & 1) ^ 1) * BUF;
        int t_ = tid;
        if (MODE == TAIL) asm volatile("" : "+v"(t_));
        // phase 1: (Y0, X0)
        read_y(kt, 0);
        if (more) issue(0, kt + 1, MODE == TAIL);
        if (seam) { issue_fpA(0, fpa_dst, t_); issue_fpW(0, 0, lds_base + SPARE, t_); }
        if (MODE != FIRST) { if (more) wait_vmcnt<4>(); else if (HAS_O) wait_vmcnt<5>(); else wait_vmcnt<2>(); }
        MIXQ2_SEG_END();
        mma(Xcur, 0, 0);
        MIXQ2_SEG_END();
        // phase 2: (Y0, X1)
        read_x(Xoth, kt, 1);
        if (more) issue(2, kt + 1, MODE == TAIL);
        if (seam) { issue_fpA(1, fpa_dst, t_); issue_fpW(0, 1, lds_base + SPARE, t_); }
        if (MODE != FIRST) { if (more) wait_vmcnt<4>(); else if (HAS_O) wait_vmcnt<6>(); else wait_vmcnt<0>(); }
        MIXQ2_SEG_END();
        mma(Xoth, 1, 0);
        MIXQ2_SEG_END();
        // phase 3: (Y1, X1)
        read_y(kt, 1);
        if (more) issue(1, kt + 1, MODE == TAIL);
        if (seam) { issue_fpA(2, fpa_dst, t_); issue_fpW(0, 2, lds_base + SPARE, t_); }
        if (more) wait_vmcnt<4>();
        MIXQ2_SEG_END();
        mma(Xoth, 1, 1);
        MIXQ2_SEG_END();
        // phase 4: (Y1, X0); the LOAD segment already fetches X half 0 of the next slice into the free set
        if (more) read_x(Xoth, kt + 1, 0);
        if (more) issue(3, kt + 1, MODE == TAIL);
        if (seam) { issue_fpA(3, fpa_dst, t_); issue_fpW(0, 3, lds_base + SPARE, t_); }
        if (more) wait_vmcnt<4>();
        MIXQ2_SEG_END();
        mma(Xcur, 0, 1);
        MIXQ2_SEG_END();
    };
    using first_t = std::integral_constant<int, FIRST>;
    using steady_t = std::integral_constant<int, STEADY>;
    using tail_t = std::integral_constant<int, TAIL>;

    int v = blockIdx.x; // virtual block id of the current tile
    auto stamp = [&](int idx) __attribute__((always_inline)) {
        if (p.dbg != nullptr && tid == 0 && v >= (int)gridDim.x && v < 2 * (int)gridDim.x) // second tile of each block
            static_cast<unsigned long long*>(p.dbg)[(size_t)blockIdx.x * 8 + idx] = __builtin_readcyclecounter();
    };

    // ---- prologue: slice 0 of the first tile ---------------------------------------------------------------
    setup_tile(v);
    issue(0, 0, true);
    issue(2, 0, true);
    issue(1, 0, true);
    issue(3, 0, true);
    wait_vmcnt<0>();
    MIXQ2_SEG_END();

    for (;;) {
        stamp(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0;
        setup_frag_offsets();
        read_x(XA, 0, 0);
        if (group == 1) MIXQ2_SEG_END(); // group 1 now runs one segment behind group 0
        {
            slice(XA, XB, 0, first_t{});
            slice(XB, XA, 1, steady_t{});
            stamp(1);
            int kt = 2;
            for (; kt + 3 < nk; kt += 2) { // both slices of the pair have a full successor: branch-free body
                slice(XA, XB, kt, steady_t{});
                slice(XB, XA, kt + 1, steady_t{});
            }
            stamp(2);
            for (; kt < nk; kt += 2) { // last 2-3 slices: successor may be partial in K or absent (runtime checks)
                slice(XA, XB, kt, tail_t{});
                if (kt + 1 < nk) slice(XB, XA, kt + 1, tail_t{});
            }
        }
        if (group == 0) MIXQ2_SEG_END(); // re-align the groups
        stamp(3);

        const int b_last = (nk - 1) & 1;                 // buffer of the last slice: free now
        const unsigned lds_fpa = ((b_last ^ 1) * BUF);   // fpA tile (fetched during the last slice)
        const unsigned lds_fpw1 = (b_last * BUF);        // fpW n-half 1 (fetched now)
        int et = tid; // opaque thread id for everything between two main loops (see issue_fpA)
        asm volatile("" : "+v"(et));
        const int elr = et & 31, elh = (et >> 5) & 1;
        if (HAS_O) {
#pragma unroll
            for (int i = 0; i < 4; ++i) issue_fpW(1, i, lds_base + lds_fpw1, et);
            wait_vmcnt<4>(); // fpA + fpW half 0 have landed; only the four copies above may still be in flight
            MIXQ2_SEG_END();
        }
        stamp(4);

        // ---- dequant math, tile by tile, results packed to fp16 in registers (acc registers die as we go) ----
        uint2 outp[2][4][4]; // [n tile][m tile][quad] : 4 consecutive n for row m
        const int obase = (elh ^ (elr & 15)) << 4; // 16-B slot of k-step ks = obase ^ (ks << 5)
        float sa[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) sa[j] = h2f(p.sA[min(m0 + wm * 128 + j * 32 + elr, p.M - 1)]);

        auto side = [&](int i, int j) __attribute__((always_inline)) {
            v16f P;
#pragma unroll
            for (int e = 0; e < 16; ++e) P[e] = 0.f;
            if (HAS_O) {
                const char* xo = smem + (i == 0 ? (unsigned)SPARE : lds_fpw1) + (wn * 32 + elr) * OSLICE;
                const char* yo = smem + lds_fpa + (wm * 128 + j * 32 + elr) * OSLICE;
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
        auto dequant = [&](int i, int j, const v16f& P) __attribute__((always_inline)) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nb = min(n0 + wn * 64 + i * 32 + 4 * elh + 8 * g, p.N - 4);
                const uint2 swb = *reinterpret_cast<const uint2*>(p.sW + nb);
                const float swf[4] = {h2f((uint16_t)(swb.x & 0xffffu)), h2f((uint16_t)(swb.x >> 16)),
                                      h2f((uint16_t)(swb.y & 0xffffu)), h2f((uint16_t)(swb.y >> 16))};
                uint16_t yh[4] = {0, 0, 0, 0};
                if (HAS_Y) {
                    const int m = min(m0 + wm * 128 + j * 32 + elr, p.M - 1);
                    const uint2 yb = *reinterpret_cast<const uint2*>(p.Y + (int64_t)m * p.N + nb);
                    yh[0] = (uint16_t)(yb.x & 0xffffu), yh[1] = (uint16_t)(yb.x >> 16);
                    yh[2] = (uint16_t)(yb.y & 0xffffu), yh[3] = (uint16_t)(yb.y >> 16);
                }
                uint16_t oh[4];
#pragma unroll
                for (int e2 = 0; e2 < 4; e2 += 2) {
                    const v2f s2 = v2f{swf[e2], swf[e2 + 1]} * sa[j]; // exact: fp16 x fp16 products
#pragma unroll
                    for (int e = e2; e < e2 + 2; ++e) {
                        const float c = HAS_O ? h2f(f2h_bits_of_f32_result(P[4 * g + e])) : h2f(yh[e]);
                        float vv = __builtin_fmaf((float)acc[i][j][4 * g + e], s2[e - e2], c);
                        if (EPI == EPI_DEQUANT_SILU) vv = vv / (1.f + __expf(-vv));
                        oh[e] = f2h_bits_of_f32_result(vv);
                    }
                }
                outp[i][j][g].x = (unsigned)oh[0] | ((unsigned)oh[1] << 16);
                outp[i][j][g].y = (unsigned)oh[2] | ((unsigned)oh[3] << 16);
            }
        };
        // n-half 0 (outlier weights in the LDS tail), then n-half 1 (outlier weights that were fetched meanwhile)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (HAS_O && i == 1) {
                wait_vmcnt<0>();
                MIXQ2_SEG_END();
            }
            v16f Pcur = side(i, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v16f Pnext = Pcur;
                if (j + 1 < 4) Pnext = side(i, j + 1);
                dequant(i, j, Pcur);
                Pcur = Pnext;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        stamp(5);
        MIXQ2_SEG_END(); // everyone is done with the outlier operands: both slice buffers are free

        // ---- fetch slice 0 of the next tile into buffer b_last while this tile's results are stored ----------
        const int m0_cur = m0, n0_cur = n0;
        const int vnext = v + gridDim.x;
        const bool have_next = vnext < nwg;
        if (have_next) { // slice 0 always lives in buffer 0; the results are staged through buffer 1
            setup_tile(vnext);
            issue(0, 0, true);
            issue(2, 0, true);
            issue(1, 0, true);
            issue(3, 0, true);
        }

        // ---- results -> LDS (buffer b_last^1, 64 KiB = 128 rows x 512 B at a time) -> whole-row stores --------
        char* stg = smem + BUF;
        uint16_t* D = static_cast<uint16_t*>(p.D);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (wm == h) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int r = j * 32 + elr;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int c = wn * 8 + i * 4 + g; // 16-byte chunk in the row; lh picks its 8-byte half
                            *reinterpret_cast<uint2*>(stg + r * 512 + ((c ^ (r & 31)) << 4) + elh * 8) = outp[i][j][g];
                        }
                    }
            }
            if (h == 0) {
                if (have_next) wait_vmcnt<0>(); // next tile's slice 0 (only DMA in flight: no stores issued yet)
                stamp(6);
            }
            MIXQ2_SEG_END();
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int q = it * T + et;
                const int r = q >> 5, c = q & 31;
                const uint4 val = *reinterpret_cast<const uint4*>(stg + r * 512 + ((c ^ (r & 31)) << 4));
                const int m = m0_cur + h * 128 + r, n = n0_cur + c * 8;
                if (m < p.M && n < p.N) *reinterpret_cast<uint4*>(D + (int64_t)m * p.N + n) = val;
            }
            MIXQ2_SEG_END();
        }
        stamp(7);
        if (!have_next) break;
        v = vnext;
    }
}

template <int EPI, bool HAS_O, bool HAS_Y>
static hipError_t launch_pp2_cfg(const GemmParams& p, hipStream_t st)
{
    auto kern = gemm_w8a8o16_pp2_kernel<EPI, HAS_O, HAS_Y>;
    static bool attr_done = false;
    static int num_cu = 256;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, pp2::LDS_BYTES);
        if (e != hipSuccess) return e;
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess &&
            prop.multiProcessorCount > 0)
            num_cu = prop.multiProcessorCount;
        attr_done = true;
    }
    const int tiles = ((p.M + pp2::BM - 1) / pp2::BM) * ((p.N + pp2::BN - 1) / pp2::BN);
    // one resident workgroup per CU (160 KiB LDS each); a multiple of 8 keeps "block b -> XCD b % 8" aligned
    int grid = tiles < num_cu ? tiles : (num_cu / 8) * 8;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(pp2::T), pp2::LDS_BYTES, st, p);
    return hipGetLastError();
}

template <int EPI>
static hipError_t launch_pp2_epi(const GemmParams& p, hipStream_t st)
{
    if (p.O > 0) return launch_pp2_cfg<EPI, true, false>(p, st); // the API never passes both an addend and outliers
    if (p.Y != nullptr) return launch_pp2_cfg<EPI, false, true>(p, st);
    return launch_pp2_cfg<EPI, false, false>(p, st);
}

// Persistent ping-pong GEMM.  Requires at least 4 K slices (K > 384) and an fp16 epilogue.
bool gemm_pp2_supported(const GemmParams& p, int epi)
{
    return (epi == EPI_DEQUANT || epi == EPI_DEQUANT_SILU) && p.K > 3 * pp2::KS;
}

hipError_t launch_gemm_pp2(const GemmParams& p, int epi, hipStream_t st)
{
    return epi == EPI_DEQUANT_SILU ? launch_pp2_epi<EPI_DEQUANT_SILU>(p, st) : launch_pp2_epi<EPI_DEQUANT>(p, st);
}

} // namespace mixq

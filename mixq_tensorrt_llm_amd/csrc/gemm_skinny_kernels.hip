// Skinny-M form of the fused W8A8O16 GEMM (5 <= M <= 64: small prefill chunks, BASELINE config 0 = bs 32): the
// operator is HBM-bound on the N x K int8 weight here (AI ~ 2M op/B), so the kernel is built like a GEMV:
//
//   * one workgroup per 16 output features: N/16 workgroups (256 for N = 4096: one per CU) of KW = 4, 8 or 16 waves
//     (chosen so that the chip holds >= ~12 wavefronts per CU); the waves split K evenly, so KW x N/16 wavefronts stream
//     W concurrently with 16-byte loads straight into
//     MFMA A-fragments (v_mfma_i32_16x16x64_i8: lane l = W row l%16, K bytes (l/16)*16.. of each 64-byte step): every
//     weight byte is read from HBM exactly once, no LDS staging, 16 steps (1 KiB/lane-group) of loads in flight per wave;
//   * qA (M x K int8, <= 64 rows) is the MFMA B operand, read through L2 (it is shared by every workgroup);
//   * the KW partial accumulators meet in LDS; wave t then owns m-tile t: fp16 outlier side GEMM on
//     v_mfma_f32_16x16x32_f16 (operands straight from L2), dequant FMA, 8-byte fp16 stores (4 consecutive n per lane).
//
// Same arithmetic and same results as the large-M kernels (integer accumulation is order-independent).
// Reference lines replaced: see gemm_kernels.hip.
#include "mixq_device.h"
#include "mixq_launch.h"
#include <atomic>
#include <mutex>
#include <shared_mutex>
#include <unordered_map>
#include <vector>
#include <type_traits>

namespace mixq {

// ABL: measurement-only ablations (wrong results): 1 = no qA loads, 2 = no weight loads, 4 = no epilogue operand loads.
// NT (round 3): 16-column feature tiles per workgroup.  NT = 1 is the form above; NT = 2 makes every qA fragment feed two MFMAs
// (32 output features per workgroup): the workgroups of a WIDE output then pull half as many qA bytes in total (N / 32 x M x K
// instead of N / 16 x M x K) while the chip is still full (N = 12288: 384 workgroups).  Output tile tt = nt * MT + t of the
// workgroup is finished by wave tt % KW.
template <int MT, int EPI, int KW, int ABL = 0, bool AFRAG = false, int NT = 1, int WFRAG = 0>
__global__ __launch_bounds__(KW * 64) void gemm_skinny_kernel(const GemmParams p)
{
    __shared__ v4i part[KW][MT * NT][64]; // [K part][output tile][lane]
    dbg_stamp(p.dbg, 0); // (measurement only: p.dbg is NULL in production) entry
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n0 = blockIdx.x * 16 * NT;
    const int lr = lane & 15, lq = lane >> 4;
    const int64_t K = p.K;

    // this wave's K range, in 64-byte MFMA steps
    const int nsteps = (p.K + 63) >> 6;
    const int per = (nsteps + KW - 1) / KW;
    const int s_begin = min(wave * per, nsteps), s_end = min(s_begin + per, nsteps);

    const int8_t* wrow[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) wrow[c] = p.B + (int64_t)min(n0 + c * 16 + lr, p.N - 1) * K + lq * 16;
    const int8_t* arow[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
        arow[t] = p.A + (int64_t)min(t * 16 + lr, p.M - 1) * K + lq * 16;

    v4i acc[NT][MT];
#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[c][t] = v4i{0, 0, 0, 0};

    // Epilogue operands of the FIRST output tile this wave will finish (tile `wave` = (m tile wave % MT, feature tile wave / MT)) are requested
    // NOW -- the first 128 outlier columns of the side GEMM, the row scale, the weight scales -- so that their L2 round
    // trip runs under the weight stream instead of after it (the kernel is a chain of latencies at this size).
    constexpr int PRE = 4; // pre-loaded side-GEMM steps (32 outlier columns each)
    v8h pxf[PRE], pyf[PRE];
    uint16_t psa = 0;
    uint2 psw = {0u, 0u};
    const bool fin = EPI != EPI_INT32 && wave < MT * NT && !(ABL & 4); // this wave runs an fp16 epilogue
    const int ft = wave % MT, fc = wave / MT;                              // its first tile: m tile ft, feature tile fc
    const int fm = ft * 16 + lr, fnb = n0 + fc * 16 + 4 * lq;
    if (fin) {
        const int obytes = p.O * 2;
        const char* xw = reinterpret_cast<const char*>(p.fpW) + (int64_t)min(n0 + fc * 16 + lr, p.N - 1) * obytes;
        const char* ya = reinterpret_cast<const char*>(p.fpA) + (int64_t)min(fm, p.M - 1) * obytes;
#pragma unroll
        for (int u = 0; u < PRE; ++u) {
            const int kb = u * 64 + lq * 16;
            if (kb < obytes) {
                pxf[u] = *reinterpret_cast<const v8h*>(xw + kb);
                pyf[u] = *reinterpret_cast<const v8h*>(ya + kb);
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) pxf[u][e] = (_Float16)0.f, pyf[u][e] = (_Float16)0.f;
            }
        }
        psa = p.sA[min(fm, p.M - 1)];
        psw = *reinterpret_cast<const uint2*>(p.sW + min(fnb, p.N - 4));
    }

    const v4i zero4 = {0, 0, 0, 0};
    // Main loop, round 3.  profiles/r03_small_m_timeline.txt: of the 6.8 us of this kernel at M = 32 on 4096 x 4096, 5.2 were
    // the fragment phase -- and the ISA showed why: every fragment load sat behind its own exec-mask branch (the `ok ? load :
    // 0` predicates), and the qA loads of a 4-step group were only ISSUED after the MFMAs of the group before had waited
    // `vmcnt(0)`: four (eight) serialised memory round trips per wave.  Now a batch is branch-free and entirely in flight:
    //   * addresses are clamped instead of predicated (a step past the wave's range re-reads its last valid step; a 16-byte
    //     group past K re-reads the row's last group), and only the WEIGHT fragment of such a step / group is zeroed -- a zero
    //     A operand makes the product zero whatever the qA lanes hold -- with selects on a wave-uniform / lane condition that
    //     exist only in the tail instantiation of the batch;
    //   * loads are issued in CONSUMPTION order (W_u, qA_u0, qA_u1, W_u+1, ...), all STEPS x (1 + MT) of a batch before the
    //     first MFMA, so the compiler's counted `vmcnt(n)` lets step u multiply while steps u+1.. are still on their way.
    // (Round 2's ablations of the old form, 4096 x 4096, GEMM only, us: M = 32 full 7.8 | no qA loads 4.7 | no weight loads
    //  5.9 | no loads at all 4.6.)
    constexpr int STEPS = (MT <= 2 && NT == 1) ? 16 : 8; // k-steps per batch: (NT + MT) x STEPS x 4 fragment registers
    const bool ktail = (p.K & 63) != 0;     // the row's last step is partial (K % 16 == 0 is checked on the host)
    const int koff_last = p.K - 16;
    auto do_steps = [&](int s0, int cnt, auto full_tag) __attribute__((always_inline)) {
        constexpr bool FULL = decltype(full_tag)::value; // cnt == STEPS and no partial step: no clamps, no selects
        v4i wf[STEPS][NT], af[STEPS][MT];
#pragma unroll
        for (int u = 0; u < STEPS; ++u) {
            const int su = FULL ? s0 + u : min(s0 + u, s0 + cnt - 1); // (wave-uniform)
            int off = su * 64;                                        // byte offset inside the row, + lq * 16 per lane
            if (!FULL) off = min(off + lq * 16, koff_last) - lq * 16;
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                if (WFRAG == 1 || WFRAG == 2) { // W from its registered fragment-major image (weight_image_kernel below): one contiguous 1-KiB read of
                                  // whole cache lines per load instead of 64 bytes of 16 rows K bytes apart; WFRAG == 2: non-temporal
                    const int8_t* src = p.B + ((int64_t)(min((n0 >> 4) + c, (p.N >> 4) - 1) * nsteps + su) << 10) + lane * 16; // (clamped tiles are computed, never stored)
                    if (WFRAG == 2) wf[u][c] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(src));
                    else wf[u][c] = *reinterpret_cast<const v4i*>(src);
                } else {
                    wf[u][c] = (ABL & 2) ? zero4 : *reinterpret_cast<const v4i*>(wrow[c] + off);
                }
            }
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                if (AFRAG) // fragment-major qA (quant_kernels.hip FRAG): block (m tile, k-step), lane l at l * 16 -- ONE
                           // contiguous 1-KiB read; the row-major form touches 16 rows x 64 B, 4 KiB apart (7.9 vs 5.4 us)
                    af[u][t] = (ABL & 1) ? zero4 : *reinterpret_cast<const v4i*>(p.A + ((int64_t)(t * nsteps + su) << 10) + lane * 16);
                else
                    af[u][t] = (ABL & 1) ? zero4 : *reinterpret_cast<const v4i*>(arow[t] + off);
            }
        }
        __builtin_amdgcn_sched_barrier(0); // every load of the batch is issued before the first MFMA (the scheduler otherwise
                                           // sinks loads between the MFMAs to save registers: ~10 in flight instead of 48)
#pragma unroll
        for (int u = 0; u < STEPS; ++u) {
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                v4i w = wf[u][c];
                if (!FULL) {
                    const bool dead = u >= cnt || (ktail && (s0 + u) * 64 + lq * 16 >= p.K);
                    if (dead) w = zero4;
                }
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    acc[c][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w, af[u][t], acc[c][t], 0, 0, 0);
            }
        }
    };
    dbg_stamp(p.dbg, 1); // epilogue operands requested
    if constexpr (WFRAG == 3 || WFRAG == 4) { // (4: the same with non-temporal loads)
        // ROW-MAJOR weight in whole 256-byte runs (round 5, R5.8).  The plain fragment load above takes 64 bytes of 16 rows per
        // instruction: every 256-byte DRAM chunk of the weight is asked for in four pieces by four different instructions, and a
        // cold stream of such pieces ran at ~3.5 TB/s against ~4.5 from a registered image (1-KiB reads) in the SAME kernel
        // (profiles/r05_stream_form_probe.txt, timelines).  Here an instruction reads 4 rows x 256 contiguous bytes (lane l: row
        // 4 rg + l / 16, 16-byte chunk l % 16), and the four registers of a 256-byte group (rg = 0..3) become the group's four MFMA
        // fragments through a WAVE-PRIVATE 4-KiB LDS tile: ds_write_b128 at [row][chunk ^ row], ds_read_b128 of (row l % 16, chunk
        // 4 s' + l / 16) -- both conflict-free (16 consecutive lanes touch 16 different 16-byte slots of the 256-byte bank row),
        // no barrier (the LDS operations of one wave execute in order).  K % 256 == 0 (host-checked); the wave's K range is counted
        // in 256-byte groups; a group past the range re-reads the last valid one with zeroed fragments.
        // NT = 2: the wave regroups its two feature tiles one after the other through the same LDS tile; every qA fragment then feeds
        // two MFMAs (from 33 rows on the qA re-reads -- N / 16 x M K bytes through L2, 3-4 x the weight bytes -- are what paces the launch).
        static_assert(NT <= 2 && KW == 4 && ABL == 0, "the transposing route: one or two feature tiles, four K parts");
        __shared__ v4i tr[KW][256];
        constexpr int GB = MT * NT <= 2 ? 2 : 1; // groups per batch (4 k-steps each)
        const int ngroups = p.K >> 8;
        const int gper = (ngroups + KW - 1) / KW;
        const int g_begin = min(wave * gper, ngroups), g_end = min(g_begin + gper, ngroups);
        const int lrow = lane >> 4, lch = lane & 15;
        const int8_t* wl[NT][4];
#pragma unroll
        for (int c = 0; c < NT; ++c)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) wl[c][rg] = p.B + (int64_t)min(n0 + c * 16 + 4 * rg + lrow, p.N - 1) * K + lch * 16;
        v4i* const mytr = tr[wave];
        for (int g0 = g_begin; g0 < g_end; g0 += GB) {
            const int cnt = min(GB, g_end - g0);
            v4i wr[GB][NT][4], af[GB * 4][MT];
#pragma unroll
            for (int gi = 0; gi < GB; ++gi) {
                const int gg = min(g0 + gi, g0 + cnt - 1); // (wave-uniform)
#pragma unroll
                for (int c = 0; c < NT; ++c)
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) {
                        if (WFRAG == 4) wr[gi][c][rg] = __builtin_nontemporal_load(reinterpret_cast<const v4i*>(wl[c][rg] + gg * 256));
                        else wr[gi][c][rg] = *reinterpret_cast<const v4i*>(wl[c][rg] + gg * 256);
                    }
#pragma unroll
                for (int sp = 0; sp < 4; ++sp)
#pragma unroll
                    for (int t = 0; t < MT; ++t)
                        af[gi * 4 + sp][t] = AFRAG ? *reinterpret_cast<const v4i*>(p.A + ((int64_t)(t * nsteps + gg * 4 + sp) << 10) + lane * 16)
                                                   : *reinterpret_cast<const v4i*>(arow[t] + (gg * 4 + sp) * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int gi = 0; gi < GB; ++gi) {
#pragma unroll
                for (int c = 0; c < NT; ++c) {
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) mytr[(4 * rg + lrow) * 16 + (lch ^ (4 * rg + lrow))] = wr[gi][c][rg];
#pragma unroll
                    for (int sp = 0; sp < 4; ++sp) {
                        v4i w = mytr[lr * 16 + ((4 * sp + lq) ^ lr)];
                        if (gi >= cnt) w = zero4;
#pragma unroll
                        for (int t = 0; t < MT; ++t)
                            acc[c][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(w, af[gi * 4 + sp][t], acc[c][t], 0, 0, 0);
                    }
                }
            }
            if (g0 == g_begin) dbg_stamp(p.dbg, 2);
        }
    } else {
        for (int s = s_begin; s < s_end; s += STEPS) {
            const int cnt = min(STEPS, s_end - s);
            if (cnt == STEPS && !(ktail && s + STEPS == nsteps)) do_steps(s, cnt, std::true_type{});
            else do_steps(s, cnt, std::false_type{});
            if (s == s_begin) dbg_stamp(p.dbg, 2); // first batch: first weight / qA data arrived and multiplied
        }
    }
    dbg_stamp(p.dbg, 3); // last MFMA issued

#pragma unroll
    for (int c = 0; c < NT; ++c)
#pragma unroll
        for (int t = 0; t < MT; ++t) part[wave][c * MT + t][lane] = acc[c][t];
    __syncthreads();
    dbg_stamp(p.dbg, 4); // LDS hand-over

    // wave w finishes output tiles w, w + KW, ...: tile tt = (m tile tt % MT, feature tile tt / MT); C/D layout of the 16x16
    // MFMA: m = lane & 15, n = 4 * (lane >> 4) + r
    for (int tt = wave; tt < MT * NT; tt += KW) {
        const int t = tt % MT, c0 = (tt / MT) * 16; // (c0: column offset of the feature tile inside the workgroup)
        v4i a = part[0][tt][lane];
#pragma unroll
        for (int w2 = 1; w2 < KW; ++w2) {
            const v4i b = part[w2][tt][lane];
            a = v4i{a[0] + b[0], a[1] + b[1], a[2] + b[2], a[3] + b[3]};
        }
        const int m = t * 16 + lr;
        const int nb = n0 + c0 + 4 * lq;
        if (EPI == EPI_INT32) {
            if (m < p.M && nb < p.N) *reinterpret_cast<v4i*>(static_cast<int32_t*>(p.D) + (int64_t)m * p.N + nb) = a;
            continue;
        }
        v4f P = {0.f, 0.f, 0.f, 0.f};
        if (p.O > 0) {
            const int obytes = p.O * 2;
            const char* xw = reinterpret_cast<const char*>(p.fpW) + (int64_t)min(n0 + c0 + lr, p.N - 1) * obytes;
            const char* ya = reinterpret_cast<const char*>(p.fpA) + (int64_t)min(m, p.M - 1) * obytes;
            if (tt == wave) { // (the wave's first tile) the first PRE steps were requested before the main loop
#pragma unroll
                for (int u = 0; u < PRE; ++u)
                    if (u * 64 < obytes) P = __builtin_amdgcn_mfma_f32_16x16x32_f16(pxf[u], pyf[u], P, 0, 0, 0);
            }
            for (int k0 = (tt == wave ? PRE * 64 : 0); k0 < obytes; k0 += 64) { // 32 outlier columns per step, 8 per lane
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
            const float sa = h2f(tt == wave ? psa : p.sA[m]);
            const uint2 swb = tt == wave ? psw : *reinterpret_cast<const uint2*>(p.sW + nb);
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
                const float c = p.O > 0 ? h2f(f2h_bits_of_f32_result(P[e])) : h2f(yh[e]);
                float v = __builtin_fmaf((float)a[e], h2f(swh[e]) * sa, c);
                if (epi_has_silu(EPI)) v = silu_f32(v);
                oh[e] = f2h_bits_of_f32_result(v);
            }
            if (EPI == EPI_DEQUANT_SILU_MUL) { // gate * up
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
    if (p.dbg != nullptr) {
        dbg_stamp(p.dbg, 5); // stores issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dbg_stamp(p.dbg, 6); // stores acknowledged
    }
}

static std::atomic<int> g_skinny_kw{0}; // measurement knob: force the K-split width (0 = auto)
void set_skinny_kw(int kw) { g_skinny_kw.store(kw); }
static std::atomic<int> g_skinny_nt{0}; // measurement knob: feature tiles per workgroup of the fragment-major form (0 = auto)
void set_skinny_nt(int nt) { g_skinny_nt.store(nt); }
// 16-column feature tiles per workgroup (fragment-major form).  Measured (tools/experimental/skinny_wide_probe.sh, 32 rows, us per
// operator call, 1 vs 2 tiles): 4096 x 4096 9.1 / 12.4, 6144 12.2 / 12.3, 8192 12.8 / 12.8, 11008 16.0 / 17.8, 12288 16.0 / 17.8 --
// halving the qA bytes does not pay for halving the workgroups -- except where N / 16 workgroups are a little more than one
// round of the chip (5120 x 5120: 320 workgroups, 18.0 -> 15.0 with 160): 2 tiles only there.
// (round 5: with the row-major weight read in 256-byte runs -- WFRAG == 3, one feature tile per workgroup -- 5120 x 5120 at 32 rows runs
//  12.9 us on one tile against 15.3 on two, profiles/r05_selection_check.txt: two tiles only where that route does not apply, i.e. on a
//  registered weight image or K % 256 != 0)
static std::atomic<int> g_skinny_wrows{1}; // knob 884 / 885 / 886 / 887: row-major weights in 256-byte runs through the wave-private LDS transposition: on, non-temporal by size (default) / off / on, always non-temporal / on, never
void set_skinny_wrows(int on) { g_skinny_wrows.store(on); }
// Non-temporal loads on that route for weights of 32 MiB and more -- cold by construction in any model (a decode step reads every layer's
// weights once; the same rule as the registered images' and the W8A16 stream's): operator us cold, plain -> non-temporal: 12288 x 4096 at 8 / 32 rows
// 15.9 / 19.3 -> 15.2 / 18.2, 4096 x 11008 17.5 / 19.6 -> 16.0 / 18.6, 8192 x 8192 19.5 / 20.9 -> 17.9 / 19.9; a loop over ONE such layer (warm: served by the
// Infinity Cache) is 4-12 % slower by construction (profiles/r05_skinny_256B_runs_cold.txt).  Knob 886 / 887: always / never.
static bool skinny_runs_nontemporal(int N, int K)
{
    const int m = g_skinny_wrows.load(std::memory_order_relaxed);
    return m == 2 || (m == 1 && (int64_t)N * K >= ((int64_t)32 << 20));
}
int skinny_feature_tiles(int M, int N, int K, bool image)
{
    const int f = g_skinny_nt.load();
    if (f == 1 || f == 2) return f;
    // ... and TWO tiles on that route where N / 32 workgroups are 160..256 (N = 5120..8192): every qA fragment feeds two MFMAs and the chip is
    // still (nearly) full -- operator us cold, one tile (or the tiles the rule took) -> two: 5120 x 5120 at 32 / 48 / 64 rows 16.8 / 18.8 / 22.4 -> 14.6 /
    // 17.2 / 18.5; 6144 x 4096 at 64 rows 18.7 -> 16.3; 8192 x 4096 at 64 rows 20.2 -> 17.5; 8192 x 8192 at 48 / 64 rows 26.1 / 27.4 -> 23.1 / 25.1; level at
    // N = 11008 / 12288 (1.3-1.5 rounds), behind at N = 4096 (128 workgroups) and 18944 from 48 rows (profiles/r05_skinny_256B_runs_cold.txt)
    // (4608 x 3584 -- Qwen2-7B's qkv, 144 workgroups of 32 features -- at 8 / 16 / 32 rows 11.1 / 11.2 / 12.3 -> 9.9 / 9.9 / 11.8, behind from 33 rows)
    // (... and up to 32 rows where N / 16 one-tile workgroups would not all be resident at once -- more than 3 per CU: 18944 x 3584 at 32 rows 25.1 -> 23.0 /
    //  23.6 -> 21.3 us on two boxes)
    if (!image && K % 256 == 0 && g_skinny_wrows.load(std::memory_order_relaxed) != 0)
        return ((N >= (M <= 32 ? 4608 : 5120) && N <= 8192) || (M <= 32 && (N + 15) / 16 > 3 * num_cus())) ? 2 : 1;
    const int wgs = (N + 15) / 16, cus = num_cus();
    return (wgs > cus && 8 * wgs < 11 * cus) ? 2 : 1; // (256, 352) workgroups on 256 CUs
}

// ---- weight images (MI355X extension, round 4) ---------------------------------------------------------------------------------
// The reference stores `weight` row-major int8 [N, K]; the skinny GEMM's MFMA A-fragment load then takes 64 bytes of 16 rows that
// lie K bytes apart: half a cache line per row per instruction.  A caller that can spare N K bytes per layer registers a
// FRAGMENT-MAJOR copy of the weight -- 1-KiB blocks [16-feature tile][64-byte k-step], lane l = feature % 16 + 16 * (k / 16 % 4)
// holding its 16 bytes at l * 16, the layout the quantiser already uses for qA -- and every decode-batch call on that weight
// pointer reads the copy instead: one contiguous 1-KiB read per load.  Same bytes into the same MFMA lanes: same bits.  Measured
// (tools/wfrag_probe.py, operator = quantiser + GEMM, 32 rows): 12288 x 4096 16.3 -> 14.4 us warm, 21.0 -> 18.1 cold;
// 4096 x 4096 (BASELINE configs[0]) 9.1 -> 8.3 warm, 10.9 -> 9.8 cold; 4096 x 11008 18.0 -> 15.8 / 22.3 -> 19.3.
// Loads of an image of 32 MiB or more are non-temporal (cold by construction in any model, see w8a16_gemm_kernels.hip NTW).
__global__ __launch_bounds__(256) void weight_image_kernel(const int8_t* __restrict__ W, int8_t* __restrict__ img, int N, int K)
{
    const int nsteps = K >> 6;
    const int64_t chunk = (int64_t)blockIdx.x * 256 + threadIdx.x;        // 16-byte chunk of the image
    const int64_t total = (int64_t)(N >> 4) * nsteps * 64;
    if (chunk >= total) return;
    const int lane = (int)(chunk & 63);
    const int64_t blk = chunk >> 6;
    const int64_t tile = blk / nsteps;
    const int step = (int)(blk - tile * nsteps);
    const int8_t* src = W + (tile * 16 + (lane & 15)) * (int64_t)K + step * 64 + (lane >> 4) * 16;
    *reinterpret_cast<v4i*>(img + (chunk << 4)) = *reinterpret_cast<const v4i*>(src);
}

hipError_t launch_weight_image(const int8_t* W, int8_t* img, int N, int K, hipStream_t st)
{
    if (N <= 0 || K <= 0 || N % 16 || K % 64) return hipErrorInvalidValue;
    const int64_t total = (int64_t)(N >> 4) * (K >> 6) * 64;
    hipLaunchKernelGGL(weight_image_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, W, img, N, K);
    return hipGetLastError();
}

static std::atomic<int> g_skinny_wfrag{0}; // knob 880 automatic | 881 images with plain loads | 882 with non-temporal loads | 883 images ignored

static bool g_skinny_wfrag_off() { return g_skinny_wfrag.load(std::memory_order_relaxed) == 3; }

// Content tag of an int8 [N, K] tensor in EITHER layout: the image is a permutation of the weight's 16-byte chunks, so a sum over
// chunks of a mix of each chunk's four words is the same for both -- tag(weight) == tag(image) says the image is a copy of THIS weight.
__global__ __launch_bounds__(256) void weight_tag_kernel(const uint4* __restrict__ src, int64_t nchunks, unsigned long long* __restrict__ tag)
{
    unsigned long long acc = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nchunks; i += (int64_t)gridDim.x * 256) {
        const uint4 v = src[i];
        unsigned long long h = ((unsigned long long)v.x << 32 | v.y) * 0x9E3779B97F4A7C15ull;
        h ^= ((unsigned long long)v.z << 32 | v.w) * 0xC2B2AE3D27D4EB4Full;
        h ^= h >> 29;
        acc += h * 0x165667B19E3779F9ull;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(tag, acc);
}

namespace {
// One registration.  Everything the first-use check needs is allocated HERE, at registration time (set-up work): a device word for the
// tag, a pinned host word it is copied to, an event.  The hot entries (mixq_enqueue, mixq_gemm_mixed*, the fused dequantize calls) never
// allocate, free or synchronise (the first build did hipMalloc + hipStreamSynchronize + hipFree inside `enqueue`, once per
// registration -- 96 syncs on the first decode step of a 96-linear model, and a prohibited call if another thread was capturing in
// global mode).
struct WeightImage {
    const void* image;
    int N, K;
    unsigned long long tag;   // content tag of the weight at registration time
    int state;                // 0 not checked since registration | 1 check in flight (tag kernel + copy enqueued, event recorded) | 2 verified
    unsigned long long* d_tag; // device word
    unsigned long long* h_tag; // pinned host word
    hipEvent_t ev;
};
std::shared_mutex g_wimg_mutex;
std::unordered_map<const void*, WeightImage> g_wimg;
std::vector<WeightImage> g_wimg_retired; // entries dropped by a hot entry: their buffers are released at the next set-up call, never inside enqueue
std::atomic<int> g_wimg_count{0};
std::atomic<int> g_wimg_stale{0};

void release_buffers(WeightImage& w)
{
    if (w.d_tag) (void)hipFree(w.d_tag);
    if (w.h_tag) (void)hipHostFree(w.h_tag);
    if (w.ev) (void)hipEventDestroy(w.ev);
    w.d_tag = nullptr, w.h_tag = nullptr, w.ev = nullptr;
}
void release_retired_locked() // (g_wimg_mutex held exclusively; set-up calls only)
{
    for (WeightImage& w : g_wimg_retired) release_buffers(w);
    g_wimg_retired.clear();
}

// enqueues: *d = 0; tag kernel over n_bytes of dev_ptr into *d; copy *d -> *h.  Nothing is waited for.
hipError_t enqueue_tag(const void* dev_ptr, size_t n_bytes, unsigned long long* d, unsigned long long* h, hipStream_t st)
{
    hipError_t e = hipMemsetAsync(d, 0, sizeof(*d), st);
    if (e != hipSuccess) return e;
    const int64_t nchunks = (int64_t)(n_bytes / 16);
    const int64_t want = (nchunks + 255) / 256;
    hipLaunchKernelGGL(weight_tag_kernel, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, st, static_cast<const uint4*>(dev_ptr), nchunks, d);
    e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof(*d), hipMemcpyDeviceToHost, st);
    return e;
}
} // namespace

// Builds nothing: records (weight -> image) with the weight's CONTENT TAG (two passes over N K bytes + a stream synchronisation:
// registration is set-up work).  VERDICT r4 weak #12: the registry is keyed by the weight's address, and an address says nothing
// about what lives there -- a weight freed and re-allocated at the same address with the same shape would silently be served the
// old tensor's image.  So the image is not trusted until the bytes behind the pointer have been compared with the tag once more,
// ASYNCHRONOUSLY, around its first use (resolve_weight_image), and mixq_weight_image_verify re-checks on demand.
hipError_t register_weight_image(const void* weight, const void* image, int N, int K, hipStream_t st)
{
    WeightImage w{image, N, K, 0, 0, nullptr, nullptr, nullptr};
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&w.d_tag), sizeof(*w.d_tag));
    if (e == hipSuccess) e = hipHostMalloc(reinterpret_cast<void**>(&w.h_tag), sizeof(*w.h_tag), hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&w.ev, hipEventDisableTiming);
    unsigned long long ti = 1;
    if (e == hipSuccess) e = enqueue_tag(image, (size_t)N * K, w.d_tag, w.h_tag, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) ti = *w.h_tag;
    if (e == hipSuccess) e = enqueue_tag(weight, (size_t)N * K, w.d_tag, w.h_tag, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e == hipSuccess) w.tag = *w.h_tag;
    if (e == hipSuccess && w.tag != ti) e = hipErrorInvalidValue; // (the image just built is not a permutation of the weight: never seen; refuse)
    if (e != hipSuccess) {
        release_buffers(w);
        return e;
    }
    std::unique_lock<std::shared_mutex> lk(g_wimg_mutex);
    release_retired_locked();
    const auto it = g_wimg.find(weight);
    if (it != g_wimg.end()) release_buffers(it->second);
    g_wimg[weight] = w;
    g_wimg_count.store((int)g_wimg.size(), std::memory_order_release);
    return hipSuccess;
}
bool unregister_weight_image(const void* weight)
{
    std::unique_lock<std::shared_mutex> lk(g_wimg_mutex);
    release_retired_locked();
    const auto it = g_wimg.find(weight);
    const bool had = it != g_wimg.end();
    if (had) {
        release_buffers(it->second);
        g_wimg.erase(it);
    }
    g_wimg_count.store((int)g_wimg.size(), std::memory_order_release);
    return had;
}
int weight_image_stale_count() { return g_wimg_stale.load(); }

// 1 = the weight behind the pointer still has the registered content, 0 = it does not (the entry is dropped), -1 = nothing registered, or the
// check itself failed (then the entry is dropped and counted as well: an image that cannot be verified is not served).  Synchronises `st`.
int verify_weight_image(const void* weight, hipStream_t st)
{
    std::unique_lock<std::shared_mutex> lk(g_wimg_mutex); // (set-up / diagnostic call: holds the registry for the duration of one pass over the weight)
    release_retired_locked();
    const auto it = g_wimg.find(weight);
    if (it == g_wimg.end()) return -1;
    WeightImage& w = it->second;
    hipError_t e = enqueue_tag(weight, (size_t)w.N * w.K, w.d_tag, w.h_tag, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    const bool same = e == hipSuccess && *w.h_tag == w.tag;
    if (same) {
        w.state = 2;
        return 1;
    }
    release_buffers(w);
    g_wimg.erase(it);
    g_wimg_count.store((int)g_wimg.size(), std::memory_order_release);
    g_wimg_stale.fetch_add(1);
    return e == hipSuccess ? 0 : -1;
}

// The image a call on `weight` may stream, or null.  ONE lookup per call (the API layer passes the result on in GemmParams::b_image), and
// nothing here blocks: an entry that has not been checked since its registration gets its tag kernel + an 8-byte copy ENQUEUED on `st`
// behind an event, and this call (and every call until the event has fired) reads `weight` itself; the first call that finds the event
// complete compares the tags -- equal: the image is trusted from then on; different: the entry is dropped (its buffers are released at
// the next set-up call) and counted.  While `st` is being captured nothing can be enqueued for the check and the image is simply not used
// (callers warm a graph's calls up eagerly first, or call mixq_weight_image_verify after registering, as bench.py does).
const void* resolve_weight_image(const void* weight, int N, int K, hipStream_t st)
{
    if (g_wimg_count.load(std::memory_order_acquire) == 0) return nullptr; // (the common case costs one atomic load)
    if (g_skinny_wfrag_off()) return nullptr;
    {
        std::shared_lock<std::shared_mutex> lk(g_wimg_mutex);
        const auto it = g_wimg.find(weight);
        if (it == g_wimg.end() || it->second.N != N || it->second.K != K) return nullptr; // (another shape: stale, ignored)
        if (it->second.state == 2) return it->second.image;
    }
    // not verified yet.  Inside a capture of `st` nothing may be enqueued for the check and no event may be queried: the call reads `weight`.
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return nullptr;
    // (another thread may be capturing in GLOBAL mode, which forbids event queries from every thread: this thread's calls below are relaxed)
    struct Relaxed {
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        Relaxed() { (void)hipThreadExchangeStreamCaptureMode(&mode); }
        ~Relaxed() { (void)hipThreadExchangeStreamCaptureMode(&mode); }
    } relaxed;
    std::unique_lock<std::shared_mutex> lk(g_wimg_mutex);
    const auto it = g_wimg.find(weight);
    if (it == g_wimg.end() || it->second.N != N || it->second.K != K) return nullptr;
    WeightImage& w = it->second;
    if (w.state == 2) return w.image;
    if (w.state == 0) {
        hipError_t e = enqueue_tag(weight, (size_t)w.N * w.K, w.d_tag, w.h_tag, st);
        if (e == hipSuccess) e = hipEventRecord(w.ev, st);
        if (e == hipSuccess) {
            w.state = 1;
            return nullptr;
        }
    } else if (hipEventQuery(w.ev) == hipErrorNotReady) {
        return nullptr; // (still on its way: this call reads `weight`)
    } else if (*w.h_tag == w.tag) {
        w.state = 2;
        return w.image;
    }
    // the check could not be enqueued, or the tags differ: the entry goes (no retry on every later call)
    g_wimg_retired.push_back(w);
    g_wimg.erase(it);
    g_wimg_count.store((int)g_wimg.size(), std::memory_order_release);
    g_wimg_stale.fetch_add(1);
    return nullptr;
}

void set_skinny_wfrag(int mode) { g_skinny_wfrag.store(mode); }

// the weight route launch_skinny_kw takes (GemmParams::b_frag): 0 row-major in 64-byte pieces | 1 / 2 the registered image | 3 row-major in 256-byte runs
int skinny_weight_route(int a_frag, bool image, int M, int N, int K)
{
    const int mode = g_skinny_wfrag.load(std::memory_order_relaxed);
    const bool runs = K % 256 == 0 && g_skinny_wrows.load(std::memory_order_relaxed) != 0 && (a_frag == 0 || skinny_feature_tiles(M, N, K, false) <= 2);
    // (round 6, VERDICT r5 weak #6: an opt-in that costs + N K bytes must never be slower.  From 49 rows on the row-major weight in 256-byte runs
    //  is ahead of the image -- decode_step bs 64, round 5: 2347 us with images, 2282 without, the difference being the 4096 x 11008 calls that stay
    //  on this kernel -- so the image is streamed up to 48 rows only wherever the run route exists; forced modes 1 / 2 keep it for measurements.)
    const bool image_pays = mode == 1 || mode == 2 || M <= 48 || !runs;
    if (a_frag == 1 && image && mode != 3 && K % 64 == 0 && N % 16 == 0 && image_pays)
        return mode == 1 ? 1 : mode == 2 ? 2 : ((int64_t)N * K >= ((int64_t)32 << 20) ? 2 : 1);
    if (runs) return 3;
    return 0;
}

template <int EPI, int KW, int ABL = 0>
static hipError_t launch_skinny_kw_impl(const GemmParams& p, hipStream_t st);

template <int EPI, int KW, int ABL = 0>
static hipError_t launch_skinny_kw(const GemmParams& p_in, hipStream_t st)
{
    GemmParams p = p_in;
    p.b_frag = 0;
    if (KW == 4 && ABL == 0 && EPI != EPI_INT32) {
        p.b_frag = skinny_weight_route(p.a_frag, p.b_image != nullptr, p.M, p.N, p.K); // 1 / 2 image | 3 row-major, 256-byte runs | 0 pieces
        if (p.b_frag == 1 || p.b_frag == 2) p.B = static_cast<const int8_t*>(p.b_image); // (resolved ONCE per call by the API layer)
    }
    return launch_skinny_kw_impl<EPI, KW, ABL>(p, st);
}

template <int MT, int EPI, int KW, int NT>
static hipError_t launch_skinny_frag(const GemmParams& p, dim3 grid, dim3 block, hipStream_t st)
{
    if constexpr (KW == 4) {
        if (p.b_frag == 3) {
            if (skinny_runs_nontemporal(p.N, p.K)) hipLaunchKernelGGL((gemm_skinny_kernel<MT, EPI, KW, 0, true, NT, 4>), grid, block, 0, st, p);
            else hipLaunchKernelGGL((gemm_skinny_kernel<MT, EPI, KW, 0, true, NT, 3>), grid, block, 0, st, p);
            return hipGetLastError();
        }
    }
    if (p.b_frag == 1) hipLaunchKernelGGL((gemm_skinny_kernel<MT, EPI, KW, 0, true, NT, 1>), grid, block, 0, st, p);
    else if (p.b_frag == 2) hipLaunchKernelGGL((gemm_skinny_kernel<MT, EPI, KW, 0, true, NT, 2>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_skinny_kernel<MT, EPI, KW, 0, true, NT>), grid, block, 0, st, p);
    return hipGetLastError();
}

template <int EPI, int KW, int ABL>
static hipError_t launch_skinny_kw_impl(const GemmParams& p, hipStream_t st)
{
    const dim3 grid((unsigned)((p.N + 15) / 16)), block(KW * 64);
    const int mt = (p.M + 15) / 16;
    if constexpr (KW == 4 && ABL == 0 && EPI != EPI_INT32) {
        if (p.a_frag == 1) { // (decode batches: M <= 64)
            if (mt > 4) return hipErrorInvalidValue;
            if (p.b_frag == 3 && mt >= 3 && skinny_feature_tiles(p.M, p.N, p.K, false) == 2) { // (256-byte runs: 32 features per workgroup)
                const dim3 grid2((unsigned)((p.N + 31) / 32));
                return mt == 3 ? launch_skinny_frag<3, EPI, KW, 2>(p, grid2, block, st) : launch_skinny_frag<4, EPI, KW, 2>(p, grid2, block, st);
            }
            if (mt == 3) return launch_skinny_frag<3, EPI, KW, 1>(p, grid, block, st);
            if (mt == 4) return launch_skinny_frag<4, EPI, KW, 1>(p, grid, block, st);
            if (skinny_feature_tiles(p.M, p.N, p.K, p.b_frag == 1 || p.b_frag == 2) == 2) { // 32 features per workgroup
                const dim3 grid2((unsigned)((p.N + 31) / 32));
                return mt == 1 ? launch_skinny_frag<1, EPI, KW, 2>(p, grid2, block, st) : launch_skinny_frag<2, EPI, KW, 2>(p, grid2, block, st);
            }
            return mt == 1 ? launch_skinny_frag<1, EPI, KW, 1>(p, grid, block, st) : launch_skinny_frag<2, EPI, KW, 1>(p, grid, block, st);
        }
    }
    if (p.a_frag != 0) return hipErrorInvalidValue;
    if constexpr (KW == 4 && ABL == 0 && EPI != EPI_INT32) {
        if (p.b_frag == 3) { // row-major qA, row-major weight in 256-byte runs
            if (skinny_runs_nontemporal(p.N, p.K)) {
                switch (mt) {
                case 1: hipLaunchKernelGGL((gemm_skinny_kernel<1, EPI, KW, 0, false, 1, 4>), grid, block, 0, st, p); break;
                case 2: hipLaunchKernelGGL((gemm_skinny_kernel<2, EPI, KW, 0, false, 1, 4>), grid, block, 0, st, p); break;
                case 3: hipLaunchKernelGGL((gemm_skinny_kernel<3, EPI, KW, 0, false, 1, 4>), grid, block, 0, st, p); break;
                default: hipLaunchKernelGGL((gemm_skinny_kernel<4, EPI, KW, 0, false, 1, 4>), grid, block, 0, st, p); break;
                }
                return hipGetLastError();
            }
            switch (mt) {
            case 1: hipLaunchKernelGGL((gemm_skinny_kernel<1, EPI, KW, 0, false, 1, 3>), grid, block, 0, st, p); break;
            case 2: hipLaunchKernelGGL((gemm_skinny_kernel<2, EPI, KW, 0, false, 1, 3>), grid, block, 0, st, p); break;
            case 3: hipLaunchKernelGGL((gemm_skinny_kernel<3, EPI, KW, 0, false, 1, 3>), grid, block, 0, st, p); break;
            default: hipLaunchKernelGGL((gemm_skinny_kernel<4, EPI, KW, 0, false, 1, 3>), grid, block, 0, st, p); break;
            }
            return hipGetLastError();
        }
    }
    switch (mt) {
    case 1: hipLaunchKernelGGL((gemm_skinny_kernel<1, EPI, KW, ABL>), grid, block, 0, st, p); break;
    case 2: hipLaunchKernelGGL((gemm_skinny_kernel<2, EPI, KW, ABL>), grid, block, 0, st, p); break;
    case 3: hipLaunchKernelGGL((gemm_skinny_kernel<3, EPI, KW, ABL>), grid, block, 0, st, p); break;
    default: hipLaunchKernelGGL((gemm_skinny_kernel<4, EPI, KW, ABL>), grid, block, 0, st, p); break;
    }
    return hipGetLastError();
}

template <int EPI>
static hipError_t launch_skinny_epi(const GemmParams& p, hipStream_t st)
{
    // KW = 4 everywhere: measured (tools/skinny_sweep.sh) 8 / 16 K-split waves are never faster, even for N = 4096
    // where KW = 4 leaves one wave per SIMD -- the kernel is bound by the qA re-reads through L1, not by occupancy.
    // (Round 2: 2 / 4 sixteen-column groups per wave sharing each qA fragment -- what made the fpA_intB twin of this kernel,
    //  w8a16_skinny_kernel, fast -- measured 25-100 % SLOWER here at every M <= 64 on five shapes: a quarter of the workgroups,
    //  four times the weight registers in flight per wave; the int8 fragments are half the bytes of the fp16 ones to begin with.
    //  Rotating the K-step order per workgroup, so that the workgroups do not all ask for the same qA lines at the same moment:
    //  no change either -- 7.9-8.2 vs 8.3-8.7 us at M = 32 on 4096 x 4096 -- the L2 serves the broadcast.)
    if (p.a_frag == 1) { // fragment-major qA (decode batches): the one configuration that reads that image
        if constexpr (EPI != EPI_INT32) return launch_skinny_kw<EPI, 4>(p, st);
        return hipErrorInvalidValue;
    }
    int kw = g_skinny_kw.load();
    if (EPI == EPI_DEQUANT && kw >= 21 && kw <= 28) { // measurement-only ablations (variant 40 + 20 + ABL): wrong results
        switch (kw - 20) {
        case 1: return launch_skinny_kw<EPI, 4, 1>(p, st);
        case 2: return launch_skinny_kw<EPI, 4, 2>(p, st);
        case 3: return launch_skinny_kw<EPI, 4, 3>(p, st);
        case 4: return launch_skinny_kw<EPI, 4, 4>(p, st);
        default: return launch_skinny_kw<EPI, 4, 7>(p, st);
        }
    }
    if (kw == 0 || kw > 16) kw = 4;
    if (kw >= 16) return launch_skinny_kw<EPI, 16>(p, st);
    if (kw >= 8) return launch_skinny_kw<EPI, 8>(p, st);
    return launch_skinny_kw<EPI, 4>(p, st);
}

bool gemm_skinny_supported(const GemmParams& p) { return p.M <= 64 && p.O <= 256; }

hipError_t launch_gemm_skinny(const GemmParams& p, int epi, hipStream_t st)
{
    switch (epi) {
    case EPI_DEQUANT: return launch_skinny_epi<EPI_DEQUANT>(p, st);
    case EPI_DEQUANT_SILU: return launch_skinny_epi<EPI_DEQUANT_SILU>(p, st);
    case EPI_DEQUANT_SILU_MUL: return launch_skinny_epi<EPI_DEQUANT_SILU_MUL>(p, st);
    default: return launch_skinny_epi<EPI_INT32>(p, st);
    }
}

} // namespace mixq

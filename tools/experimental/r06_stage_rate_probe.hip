// Round 6 probe (measurement only, not part of the library): how fast can ONE CU pull operand slices global -> LDS, by who issues
// the copies and how, and what does that do to MFMA waves on the same CU?  The numbers decide between the two forms VERDICT r5 #1
// names for the 65..1024-row band (a copy-only wave / two slices per barrier).
//
// One workgroup per CU (128 KiB of dynamic LDS forces that), 256 workgroups.  A "slice" is what the 128 x 128 deep form stages per
// 128 bytes of K: 32 instructions of 64 lanes x 16 B = 32 KiB, every instruction 8 rows x 128 contiguous bytes, rows `pitch` apart.
//   loader waves [0, NLOAD): issue PER = 32 / NLOAD instructions per slice each, keep DEPTH slices in flight (s_waitcnt vmcnt)
//   MFMA waves   [NLOAD, NLOAD + MF): chains of v_mfma_i32_32x32x32_i8, `mfma_per_slice` per slice per wave, no dependence on the copies
// No barriers: both sides run free, so each side's span is its own throughput next to the other.
//   MODE 0  global_load_lds_dwordx4, SGPR base + 32-bit lane offset      1  the same with a 64-bit per-lane address
//   MODE 2  global_load_dwordx4 into registers (xor-ed away)             3  the same + ds_write_b128 into the stage
// Source scenarios (panel = 256 rows x pitch bytes per workgroup):
//   l2priv   every CU its own 64 KiB (2 slices, re-read): 2 MiB per XCD, L2 hits        l2shared  all CUs the same 1 MiB panel
//   mall     128 panels of 1 MiB: 16 MiB per XCD > L2, Infinity Cache hits              hbm       a fresh 256 MiB window per launch
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experimental/r06_stage_rate_probe.hip -o tools/experimental/bin/r06_stage_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CK(x)                                                                          \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

__device__ __forceinline__ void dma_sbase(const char* sbase, unsigned voff, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(sbase), "s"(lds_addr)
                 : "memory");
}
__device__ __forceinline__ void dma_vaddr(const void* g, unsigned lds_addr)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(lds_addr)
                 : "memory");
}

struct Args {
    const char* src;
    unsigned long long* stamps; // [wg][2 sides][2]
    int* sink;
    size_t panel_stride;
    int panels;     // panel = (blockIdx.x + shift) % panels
    int shift;
    int pitch;      // bytes between rows
    int kcycle;     // distinct slices before the K offset wraps
    int slices;     // slices per workgroup
    int mfma_per_slice;
};

template <int MODE, int NLOAD, int DEPTH, int MF>
__global__ __launch_bounds__((NLOAD + MF) * 64) void probe(const Args a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int PER = 32 / NLOAD;
    constexpr int NST = 4; // 4 x 32 KiB stages
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const char* panel = a.src + (size_t)((blockIdx.x + a.shift) % a.panels) * a.panel_stride;
    unsigned long long t0 = 0, t1 = 0;
    if (wave < NLOAD) {
        unsigned voff[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) voff[i] = (unsigned)(((wave * PER + i) * 8 + (lane >> 3)) * a.pitch + (lane & 7) * 16);
        const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) void*)smem);
        t0 = wall_clock64();
        if (MODE <= 1) {
            for (int s = 0; s < a.slices; ++s) {
                const char* base = panel + (size_t)(s % a.kcycle) * 128;
                const unsigned dst = lds0 + (s % NST) * 32768 + wave * PER * 1024;
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    if (MODE == 0) dma_sbase(base, voff[i], dst + i * 1024);
                    else dma_vaddr(base + voff[i], dst + i * 1024);
                }
                constexpr int W = (DEPTH - 1) * PER > 63 ? 63 : (DEPTH - 1) * PER;
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            // register route: DEPTH slices of PER registers in flight (compiler-counted loads, unrolled ring)
            v4i r[DEPTH][PER];
            v4i acc = {0, 0, 0, 0};
#pragma unroll
            for (int d = 0; d < DEPTH; ++d)
#pragma unroll
                for (int i = 0; i < PER; ++i)
                    r[d][i] = *reinterpret_cast<const v4i*>(panel + (size_t)(d % a.kcycle) * 128 + voff[i]);
            for (int s = 0; s < a.slices; s += DEPTH) {
#pragma unroll
                for (int d = 0; d < DEPTH; ++d) {
#pragma unroll
                    for (int i = 0; i < PER; ++i) {
                        if (MODE == 3)
                            *reinterpret_cast<v4i*>(smem + ((s + d) % NST) * 32768 + (wave * PER + i) * 1024 + lane * 16) = r[d][i];
                        else acc ^= r[d][i];
                    }
                    const int sn = s + d + DEPTH;
                    if (sn < a.slices) {
#pragma unroll
                        for (int i = 0; i < PER; ++i)
                            r[d][i] = *reinterpret_cast<const v4i*>(panel + (size_t)(sn % a.kcycle) * 128 + voff[i]);
                    }
                }
            }
            if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678) a.sink[0] = 1;
        }
        t1 = wall_clock64();
        if (lane == 0 && wave == 0) {
            a.stamps[(size_t)blockIdx.x * 4 + 0] = t0;
        }
        if (lane == 0) atomicMax(&a.stamps[(size_t)blockIdx.x * 4 + 1], t1);
    } else {
        v4i x = {lane, 1, 2, 3}, y = {3, 2, 1, lane};
        v16i c0 = {}, c1 = {}, c2 = {}, c3 = {};
        t0 = wall_clock64();
        const int n = a.slices * a.mfma_per_slice / 4;
        for (int i = 0; i < n; ++i) {
            c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(x, y, c3, 0, 0, 0);
        }
        t1 = wall_clock64();
        int s = 0;
#pragma unroll
        for (int e = 0; e < 16; ++e) s ^= c0[e] ^ c1[e] ^ c2[e] ^ c3[e];
        if (s == 0x12345678) a.sink[1] = 1;
        if (lane == 0 && wave == NLOAD) a.stamps[(size_t)blockIdx.x * 4 + 2] = t0;
        if (lane == 0) atomicMax(&a.stamps[(size_t)blockIdx.x * 4 + 3], t1);
    }
}

struct Scenario {
    const char* name;
    int panels, pitch, kcycle;
    bool fresh; // a new window per launch
};

static char* g_buf;
static size_t g_buf_bytes;
static unsigned long long* g_stamps;
static int* g_sink;
static int g_launch = 0;

template <int MODE, int NLOAD, int DEPTH, int MF>
static void run(const Scenario& sc, int slices, int mfma_per_slice)
{
    constexpr int T = (NLOAD + MF) * 64;
    auto kern = probe<MODE, NLOAD, DEPTH, MF>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    Args a{};
    a.stamps = g_stamps, a.sink = g_sink;
    a.panel_stride = (size_t)256 * sc.pitch;
    a.panels = sc.panels, a.pitch = sc.pitch, a.kcycle = sc.kcycle, a.slices = slices, a.mfma_per_slice = mfma_per_slice;
    const size_t window = (size_t)sc.panels * a.panel_stride;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    std::vector<float> ms;
    std::vector<double> load_us, mfma_us;
    for (int rep = 0; rep < 6; ++rep) {
        const size_t nwin = std::max<size_t>(1, g_buf_bytes / window);
        a.src = g_buf + (sc.fresh ? (size_t)(g_launch++ % nwin) * window : 0);
        a.shift = 0;
        CK(hipMemsetAsync(g_stamps, 0, 256 * 4 * 8, 0));
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(kern, dim3(256), dim3(T), 131072, 0, a);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float t;
        CK(hipEventElapsedTime(&t, e0, e1));
        std::vector<unsigned long long> st(256 * 4);
        CK(hipMemcpy(st.data(), g_stamps, 256 * 4 * 8, hipMemcpyDeviceToHost));
        // per-workgroup spans (100 MHz clock -> us), mean over workgroups
        double ls = 0, fs = 0;
        for (int w = 0; w < 256; ++w) {
            ls += (double)(st[w * 4 + 1] - st[w * 4 + 0]) / 100.0;
            if (MF) fs += (double)(st[w * 4 + 3] - st[w * 4 + 2]) / 100.0;
        }
        if (rep >= 1) { // (the first launch warms caches / code)
            ms.push_back(t);
            load_us.push_back(ls / 256);
            mfma_us.push_back(fs / 256);
        }
    }
    std::sort(ms.begin(), ms.end());
    std::sort(load_us.begin(), load_us.end());
    std::sort(mfma_us.begin(), mfma_us.end());
    const double lus = load_us[load_us.size() / 2], fus = mfma_us[mfma_us.size() / 2];
    const double bytes_cu = (double)slices * 32768;
    // MFMA: 32 cycles each per SIMD; MF waves over 4 SIMDs
    const double mfma_cycles_simd = MF ? (double)slices * mfma_per_slice * 32.0 * ((MF + 3) / 4) : 0;
    printf("%-9s mode %d  loaders %d x depth %d  mfma waves %d (%2d/slice)  kernel %7.1f us | copy span %7.1f us = %6.1f GB/s per CU (%5.2f TB/s chip), "
           "%5.3f us per slice | mfma span %7.1f us",
           sc.name, MODE, NLOAD, DEPTH, MF, mfma_per_slice, ms[ms.size() / 2] * 1e3, lus, bytes_cu / lus / 1e3, bytes_cu * 256 / lus / 1e6,
           lus / slices, fus);
    if (MF) printf(" = %4.2f GHz-equivalent at 100 %% duty", mfma_cycles_simd / fus / 1e3);
    printf("\n");
    fflush(stdout);
    CK(hipEventDestroy(e0));
    CK(hipEventDestroy(e1));
}

int main(int argc, char** argv)
{
    g_buf_bytes = (size_t)2 << 30;
    CK(hipMalloc(&g_buf, g_buf_bytes));
    CK(hipMemset(g_buf, 1, g_buf_bytes));
    CK(hipMalloc(&g_stamps, 256 * 4 * 8));
    CK(hipMalloc(&g_sink, 64));
    const Scenario scs[] = {{"l2priv", 256, 4096, 2, false}, {"l2shared", 1, 4096, 32, false}, {"mall", 128, 4096, 32, false},
                            {"hbm", 256, 4096, 32, true}};
    const int slices = argc > 1 ? atoi(argv[1]) : 256;
    if (argc > 2) { // short form: the cold scenarios only, the GEMM-like issue shapes only (how much of a SHORT launch is ramp?)
        for (int rep = 0; rep < 2; ++rep)
            for (const Scenario& sc : {scs[3], scs[2], scs[0]}) {
                run<0, 2, 3, 0>(sc, slices, 0);
                run<0, 2, 3, 8>(sc, slices, 8);
                run<0, 8, 3, 0>(sc, slices, 0);
            }
        return 0;
    }
    for (const Scenario& sc : scs) {
        // who issues: 1 / 2 / 4 / 8 loader waves, LDS-DMA with SGPR base, 3 slices in flight (1 loader: vmcnt caps at 63 = 2 slices)
        run<0, 1, 2, 0>(sc, slices, 0);
        run<0, 2, 3, 0>(sc, slices, 0);
        run<0, 4, 3, 0>(sc, slices, 0);
        run<0, 8, 3, 0>(sc, slices, 0);
        run<0, 8, 1, 0>(sc, slices, 0);
        run<0, 8, 2, 0>(sc, slices, 0);
        // 64-bit lane addresses instead of SGPR base + offset
        run<1, 8, 3, 0>(sc, slices, 0);
        run<1, 4, 3, 0>(sc, slices, 0);
        // register route, with and without the LDS write
        run<2, 8, 3, 0>(sc, slices, 0);
        run<3, 8, 3, 0>(sc, slices, 0);
        run<2, 4, 2, 0>(sc, slices, 0);
        // next to MFMA waves (the deep form's ratio: 8 waves x 8 MFMAs per slice; and 4 waves x 16)
        run<0, 1, 2, 8>(sc, slices, 8);
        run<0, 2, 3, 8>(sc, slices, 8);
        run<0, 4, 3, 8>(sc, slices, 8);
        run<0, 4, 3, 4>(sc, slices, 16);
        run<0, 8, 3, 8>(sc, slices, 8);
        run<3, 4, 3, 8>(sc, slices, 8);
    }
    // MFMA waves alone (reference for the duty figure)
    {
        const Scenario sc{"none", 1, 4096, 32, false};
        run<0, 1, 2, 8>(sc, 1, 8 * 256);
    }
    return 0;
}

// MFMA-only throughput at the power cap: v_mfma_i32_32x32x32_i8 vs v_mfma_i32_16x16x64_i8 (and the fp16 pair), register operands
// with pseudo-random bytes, no memory traffic.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 tools/experimental/mfma_power_probe.hip -o /tmp/mfma_probe && /tmp/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ inline unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(512) void probe(int iters, int* sink)
{
    const unsigned t = blockIdx.x * 512 + threadIdx.x;
    v4i a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = v4i{(int)hash(t * 8 + i), (int)hash(t * 8 + i + 100), (int)hash(t * 8 + i + 200), (int)hash(t * 8 + i + 300)};
        b[i] = v4i{(int)hash(t * 8 + i + 400), (int)hash(t * 8 + i + 500), (int)hash(t * 8 + i + 600), (int)hash(t * 8 + i + 700)};
    }
    if (MODE == 0) { // 32x32x32 i8: 8 accumulators (128 regs), 2 x 4 operand reuse like the GEMM
        v16i acc[8];
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i & 1], b[i >> 1], acc[i], 0, 0, 0);
        int s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0];
        if (s == 0x12345) sink[t] = s;
    } else if (MODE == 1) { // 16x16x64 i8: 32 accumulators (128 regs), 4 x 8 ... operands reused 4 x 4
        v4i acc[32];
        for (int i = 0; i < 32; ++i) acc[i] = v4i{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 32; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i & 3], b[(i >> 2) & 3], acc[i], 0, 0, 0);
        int s = 0; for (int i = 0; i < 32; ++i) s += acc[i][0];
        if (s == 0x12345) sink[t] = s;
    } else if (MODE == 2) { // 32x32x16 f16
        v16f acc[8];
        for (int i = 0; i < 8; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a[i & 1] & 0x3fff3fff), __builtin_bit_cast(v8h, b[i >> 1] & 0x3fff3fff), acc[i], 0, 0, 0);
        float s = 0; for (int i = 0; i < 8; ++i) s += acc[i][0];
        if (s == 12345.f) sink[t] = 1;
    } else { // 16x16x32 f16
        v4f acc[32];
        for (int i = 0; i < 32; ++i) acc[i] = v4f{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it)
#pragma unroll
            for (int i = 0; i < 32; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(v8h, a[i & 3] & 0x3fff3fff), __builtin_bit_cast(v8h, b[(i >> 2) & 3] & 0x3fff3fff), acc[i], 0, 0, 0);
        float s = 0; for (int i = 0; i < 32; ++i) s += acc[i][0];
        if (s == 12345.f) sink[t] = 1;
    }
}

template <int MODE>
static void run(const char* name, double ops_per_mfma, int mfma_per_iter)
{
    int* sink; hipMalloc(&sink, 256 * 512 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        for (int k = 0; k < 10; ++k) hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, iters, sink);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double ops = 10.0 * 256 * 8 * (double)iters * mfma_per_iter * ops_per_mfma;
        printf("%s: %.1f ms for 10 launches -> %.0f T(FL)OP/s\n", name, ms, ops / (ms * 1e-3) / 1e12);
    }
    hipFree(sink);
}

int main()
{
    run<0>("i8 32x32x32", 2.0 * 32 * 32 * 32, 8);
    run<1>("i8 16x16x64", 2.0 * 16 * 16 * 64, 32);
    run<2>("f16 32x32x16", 2.0 * 32 * 32 * 16, 8);
    run<3>("f16 16x16x32", 2.0 * 16 * 16 * 32, 32);
    run<0>("i8 32x32x32 (again)", 2.0 * 32 * 32 * 32, 8);
    return 0;
}

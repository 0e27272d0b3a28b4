// Launch-floor probe (measurement only, not part of the library): what does a kernel cost on this box before it moves a
// byte?  Graph of 200 back-to-back launches on one stream (device-paced, no host launch cost), several kernel bodies:
//   empty                       nothing
//   one dependent load chain    1 / 2 dependent global loads per thread (first byte latency)
//   store                       one 8-byte store per thread (stores must drain before the kernel ends)
//   mfma chain                  32 dependent v_mfma_i32_16x16x64_i8 + an LDS hand-over + store (the skinny GEMM's skeleton)
// for grids of 32 / 256 / 1024 workgroups of 256 threads.
//   hipcc --offload-arch=gfx950 -O3 tools/experimental/launch_floor_probe.hip -o /tmp/floor && /tmp/floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void k_empty(const int* a, int* o) {}
__global__ void k_load1(const int* a, int* o)
{
    int v = a[(blockIdx.x * 256 + threadIdx.x) & 0xfffff];
    if (v == 0x7fffffff) o[0] = v;
}
__global__ void k_load2(const int* a, int* o)
{
    int v = a[(blockIdx.x * 256 + threadIdx.x) & 0xfffff];
    v = a[(v + threadIdx.x) & 0xfffff];
    if (v == 0x7fffffff) o[0] = v;
}
__global__ void k_store(const int* a, int* o) { o[blockIdx.x * 256 + threadIdx.x] = threadIdx.x; }
__global__ void k_load_store(const int* a, int* o)
{
    o[blockIdx.x * 256 + threadIdx.x] = a[(blockIdx.x * 256 + threadIdx.x) & 0xfffff];
}
__global__ void k_mfma(const int* a, int* o)
{
    __shared__ v4i part[4][64];
    v4i x = {(int)threadIdx.x, 1, 2, 3}, y = {3, 2, 1, (int)blockIdx.x}, c = {0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 32; ++i) c = __builtin_amdgcn_mfma_i32_16x16x64_i8(x, y, c, 0, 0, 0);
    part[threadIdx.x >> 6][threadIdx.x & 63] = c;
    __syncthreads();
    if (threadIdx.x < 64) {
        v4i s = part[0][threadIdx.x];
        for (int w = 1; w < 4; ++w) {
            v4i b = part[w][threadIdx.x];
            s = v4i{s[0] + b[0], s[1] + b[1], s[2] + b[2], s[3] + b[3]};
        }
        reinterpret_cast<v4i*>(o)[blockIdx.x * 64 + threadIdx.x] = s;
    }
}

template <class K>
static float time_graph(K kern, int grid, const int* a, int* o, int n = 200, int reps = 20)
{
    hipStream_t st;
    hipStreamCreate(&st);
    hipGraph_t g;
    hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), 0, st, a, o);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, st);
    hipStreamSynchronize(st);
    hipEventRecord(e0, st);
    for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st);
    hipStreamSynchronize(st);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipGraphExecDestroy(ge), hipGraphDestroy(g), hipStreamDestroy(st);
    return ms * 1e3f / (n * reps);
}

int main()
{
    int *a, *o;
    hipMalloc(&a, 4 << 20), hipMalloc(&o, 64 << 20);
    hipMemset(a, 0, 4 << 20);
    printf("# us per launch, graph of 200 back-to-back launches, 256 threads per workgroup\n");
    printf("%-22s %8s %8s %8s\n", "kernel \\ workgroups", "32", "256", "1024");
    auto row = [&](const char* name, auto kern) {
        printf("%-22s", name);
        for (int grid : {32, 256, 1024}) printf(" %8.2f", time_graph(kern, grid, a, o));
        printf("\n");
    };
    row("empty", k_empty);
    row("1 load", k_load1);
    row("2 dependent loads", k_load2);
    row("store", k_store);
    row("load -> store", k_load_store);
    row("32 mfma + lds + store", k_mfma);
    return 0;
}

// Concurrency probe (measurement only, not part of the library): can a small producer kernel and a 256-workgroup consumer
// kernel of ONE operator run side by side (two branches of a HIP graph = two streams forked with events), the consumer
// picking up a device-side flag the producer raises?  What does a call cost then, against the same two kernels back to back?
//   producer: 32 workgroups, each spins ~2 us (the quantiser's duration), then adds 1 to the flag (agent scope)
//   consumer: G workgroups, lane 0 polls the flag until 32 * epoch, then one dependent load + store (something to finish)
//   serial   = producer -> consumer on one stream (the consumer finds the flag raised)
//   parallel = fork: producer on a side stream, consumer on the main stream, join
// Stamps (s_memrealtime, 100 MHz): consumer start / flag seen, producer start / flag raised, of the last call.
//   hipcc --offload-arch=gfx950 -O3 tools/experimental/concurrent_probe.hip -o ab/concurrent_probe && ab/concurrent_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

__global__ void producer(unsigned* flag, unsigned long long* stamp, int spin_ticks)
{
    const unsigned long long t0 = wall_clock64();
    while ((long long)(wall_clock64() - t0) < spin_ticks) __builtin_amdgcn_s_sleep(1);
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(flag, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        stamp[2 * blockIdx.x] = t0;
        stamp[2 * blockIdx.x + 1] = wall_clock64();
    }
}

__global__ void consumer(unsigned* flag, unsigned target, const int* a, int* o, unsigned long long* stamp, unsigned* gave_up)
{
    __shared__ unsigned ok;
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        unsigned seen = 0;
        for (int it = 0; it < 20000; ++it) { // bounded: a probe must not hang the box
            seen = __hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            if (seen >= target) break;
            __builtin_amdgcn_s_sleep(2);
        }
        ok = seen >= target;
        if (!ok) atomicAdd(gave_up, 1u);
    }
    __syncthreads();
    const unsigned long long t1 = wall_clock64();
    int v = a[(blockIdx.x * 256 + threadIdx.x) & 0xfffff];
    o[blockIdx.x * 256 + threadIdx.x] = v + (int)ok;
    if (threadIdx.x == 0) {
        stamp[2 * blockIdx.x] = t0;
        stamp[2 * blockIdx.x + 1] = t1;
    }
}

int main()
{
    int *a, *o;
    unsigned *flag, *gave_up;
    unsigned long long *ps, *cs;
    hipMalloc(&a, 4 << 20);
    hipMalloc(&o, 4 << 20);
    hipMemset(a, 0, 4 << 20);
    hipMalloc(&flag, 256);
    hipMalloc(&gave_up, 256);
    hipMalloc(&ps, 64 * 16);
    hipMalloc(&cs, 4096 * 16);
    hipStream_t st, side;
    hipStreamCreate(&st);
    hipStreamCreateWithFlags(&side, hipStreamNonBlocking);
    hipEvent_t fork, join, e0, e1;
    hipEventCreateWithFlags(&fork, hipEventDisableTiming);
    hipEventCreateWithFlags(&join, hipEventDisableTiming);
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int n = 100;
    for (int spin : {0, 100, 200}) {
        for (int grid : {256, 768}) {
            for (int mode = 0; mode < 2; ++mode) {
                hipMemset(flag, 0, 256);
                hipMemset(gave_up, 0, 256);
                hipGraph_t g;
                hipGraphExec_t ge;
                hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
                for (int i = 0; i < n; ++i) {
                    // the flag counts up over the whole run: call i (of repetition r) waits for 32 * (calls so far + 1);
                    // the target is baked per node, so each replay continues from where the previous one ended
                    if (mode == 0) {
                        hipLaunchKernelGGL(producer, dim3(32), dim3(256), 0, st, flag, ps, spin);
                        hipLaunchKernelGGL(consumer, dim3(grid), dim3(256), 0, st, flag, 0u, a, o, cs, gave_up);
                    } else {
                        hipEventRecord(fork, st);
                        hipStreamWaitEvent(side, fork, 0);
                        hipLaunchKernelGGL(producer, dim3(32), dim3(256), 0, side, flag, ps, spin);
                        // parallel mode: poll for "at least one more full round than when this call began" is not
                        // expressible with a baked target across replays, so the producer count is reset per replay
                        hipLaunchKernelGGL(consumer, dim3(grid), dim3(256), 0, st, flag, 32u * (unsigned)(i + 1), a, o, cs, gave_up);
                        hipEventRecord(join, side);
                        hipStreamWaitEvent(st, join, 0);
                    }
                }
                hipStreamEndCapture(st, &g);
                if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) {
                    printf("instantiate failed\n");
                    return 1;
                }
                std::vector<float> ts;
                for (int r = 0; r < 12; ++r) {
                    hipMemsetAsync(flag, 0, 256, st);
                    hipEventRecord(e0, st);
                    hipGraphLaunch(ge, st);
                    hipEventRecord(e1, st);
                    hipStreamSynchronize(st);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    ts.push_back(ms * 1000.f / n);
                }
                std::sort(ts.begin(), ts.end());
                std::vector<unsigned long long> hp(64), hc(2 * grid);
                unsigned gu = 0;
                hipMemcpy(hp.data(), ps, 64 * 8, hipMemcpyDeviceToHost);
                hipMemcpy(hc.data(), cs, 2 * grid * 8, hipMemcpyDeviceToHost);
                hipMemcpy(&gu, gave_up, 4, hipMemcpyDeviceToHost);
                unsigned long long p0 = ~0ull, p1 = 0, c0 = ~0ull, c0max = 0, c1 = 0;
                for (int b = 0; b < 32; ++b) p0 = std::min(p0, hp[2 * b]), p1 = std::max(p1, hp[2 * b + 1]);
                for (int b = 0; b < grid; ++b)
                    c0 = std::min(c0, hc[2 * b]), c0max = std::max(c0max, hc[2 * b]), c1 = std::max(c1, hc[2 * b + 1]);
                printf("spin %3d ticks grid %4d %-8s: %6.2f us per call (median of 12 x %d)  gave_up %u | last call, 10-ns ticks rel. "
                       "to producer start: flag raised %lld, consumer first/last start %lld / %lld, last flag seen %lld\n",
                       spin, grid, mode ? "parallel" : "serial", ts[ts.size() / 2], n, gu, (long long)(p1 - p0), (long long)(c0 - p0),
                       (long long)(c0max - p0), (long long)(c1 - p0));
                hipGraphExecDestroy(ge);
                hipGraphDestroy(g);
            }
        }
    }
    return 0;
}

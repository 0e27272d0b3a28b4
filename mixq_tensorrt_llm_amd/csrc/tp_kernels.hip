// Row-sharded (TP) MixQ linear: the ONE collective of the path -- all-gather of the fp16 output columns -- as one-sided
// peer writes over xGMI instead of an RCCL all-gather followed by a column-placement pass.
//
// north_star / SURVEY 8e: rank r computes Out[:, n0:n1] (rows n0..n1 of W); every rank that needs the full [M, N] output
// gets it by each producer WRITING its [M, N/tp] block straight into its column block of every consumer's [M, N] buffer
// (the buffers are hipMalloc'ed once, exported with hipIpcGetMemHandle, opened by the peers: plain device pointers from
// then on).  xGMI is point-to-point -- 7 links per GPU, one per peer -- and one launch writes to all peers at once, so
// all links carry traffic concurrently and the data lands in its final place: no rank-major staging buffer, no
// permute-copy of [M, N] (the reference has no counterpart: its allreduce after an N-split, plugin.py:155-156, is
// shape-wrong and guarded off by `assert tp_size == 1`).
//
// Completion: after its stores a producer publishes a sequence number in each consumer's flag array (system-scope
// release); a consumer's stream waits for the tp flags of the current sequence number in a one-workgroup kernel
// (system-scope acquire loads), so the next kernel on that stream -- which starts with its caches invalidated -- reads
// the gathered tensor.  No host synchronisation, graph-capturable.  Two buffers alternate by call parity: a producer can
// be at most one call ahead of the slowest consumer (it waits for that consumer's flag of call i before its own call
// i + 1 is pushed ... see parallel.PeerGather), which is exactly what two buffers cover.
#include "mixq_launch.h"

namespace mixq {

constexpr int kTpMaxPeers = 8;

struct TpDest {
    void* base[kTpMaxPeers];      // [M, N] fp16 buffer of every destination rank (own rank included)
    unsigned* flag[kTpMaxPeers];  // that rank's flag word for THIS producer
};

// src [M, n_loc] fp16 (contiguous) -> dst_r[m, col0 + j] for every destination r.  16-byte vectors; each source vector
// is read once and written ndst times.
__global__ __launch_bounds__(256) void tp_push_columns_kernel(const uint4* __restrict__ src, TpDest d, int ndst, int M,
                                                               int vec_per_row /* n_loc / 8 */, int64_t dst_row_vecs
                                                               /* N / 8 */, int col0_vec, unsigned seq,
                                                               unsigned* __restrict__ done_counter)
{
    const int64_t total = (int64_t)M * vec_per_row;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t m = i / vec_per_row;
        const int j = (int)(i - m * vec_per_row);
        const uint4 v = src[i];
        const int64_t o = m * dst_row_vecs + col0_vec + j;
#pragma unroll
        for (int r = 0; r < kTpMaxPeers; ++r)
            if (r < ndst) static_cast<uint4*>(d.base[r])[o] = v;
    }
    // the last workgroup to finish publishes the sequence number to every destination
    __threadfence_system(); // this thread's peer stores are visible system-wide before the counter is bumped
    __syncthreads();
    __shared__ unsigned last;
    if (threadIdx.x == 0) last = __hip_atomic_fetch_add(done_counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (last == gridDim.x - 1u && threadIdx.x < (unsigned)ndst) {
        __hip_atomic_store(done_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // re-armed (stream-ordered reuse)
        __threadfence_system();
        __hip_atomic_store(d.flag[threadIdx.x], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// One workgroup: lane r waits until flags[r] == seq (flags written by producer r into THIS rank's flag array).
// Bounded: gives up after ~2 s of wall clock and raises *timeout_flag (the host checks it; a lost peer must not hang
// the stream forever).
__global__ __launch_bounds__(64) void tp_wait_flags_kernel(const unsigned* __restrict__ flags, int n, unsigned seq,
                                                            unsigned* __restrict__ timeout_flag)
{
    const int r = threadIdx.x;
    if (r < n) {
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(flags + r, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            __builtin_amdgcn_s_sleep(16);
            if (wall_clock64() - t0 > 200000000ull) { // 2 s at 100 MHz
                __hip_atomic_store(timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
    }
    __threadfence_system();
}

hipError_t launch_tp_push(const void* src, void* const* dst_bases, unsigned* const* dst_flags, int ndst, int M, int n_loc,
                          int N, int col0, unsigned seq, unsigned* done_counter, hipStream_t st)
{
    if (ndst < 1 || ndst > kTpMaxPeers || n_loc % 8 || N % 8 || col0 % 8) return hipErrorInvalidValue;
    if (M <= 0 || n_loc <= 0) return hipSuccess;
    TpDest d{};
    for (int r = 0; r < ndst; ++r) d.base[r] = dst_bases[r], d.flag[r] = dst_flags[r];
    const int64_t total = (int64_t)M * (n_loc / 8);
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 4; // enough to saturate 7 links; the rest of the chip keeps computing
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(tp_push_columns_kernel, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const uint4*>(src), d,
                       ndst, M, n_loc / 8, (int64_t)(N / 8), col0 / 8, seq, done_counter);
    return hipGetLastError();
}

hipError_t launch_tp_wait(const unsigned* flags, int n, unsigned seq, unsigned* timeout_flag, hipStream_t st)
{
    if (n < 1 || n > 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(tp_wait_flags_kernel, dim3(1), dim3(64), 0, st, flags, n, seq, timeout_flag);
    return hipGetLastError();
}

} // namespace mixq

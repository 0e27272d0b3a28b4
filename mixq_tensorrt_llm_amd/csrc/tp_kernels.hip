// Row-sharded (TP) MixQ linear: the ONE collective of the path -- all-gather of the fp16 output columns -- as one-sided
// peer writes over xGMI instead of an RCCL all-gather followed by a column-placement pass.
//
// north_star / SURVEY 8e: rank r computes Out[:, n0:n1] (rows n0..n1 of W); every rank that needs the full [M, N] output
// gets it by each producer WRITING its [M, N/tp] block straight into its column block of every consumer's [M, N] buffer.
// xGMI is point-to-point -- 7 links per GPU, one per peer -- and one launch writes to all peers at once, so all links
// carry traffic concurrently and the data lands in its final place: no rank-major staging buffer, no permute-copy of
// [M, N] (the reference has no counterpart: its allreduce after an N-split, plugin.py:155-156, is shape-wrong and guarded
// off by `assert tp_size == 1`).
//
// Memory (round 3): everything a REMOTE GPU writes and the local GPU reads -- the destination tiles and, above all, the
// flag words a running kernel polls -- is allocated fine-grained / uncached (hipExtMallocWithFlags, what RCCL uses for
// its own flags and buffers) BEFORE the IPC handle is taken.  System-scope atomics are only specified on fine-grained
// allocations; on coarse-grained hipMalloc memory they degrade to agent scope (mixq_api.hip: mixq_tp_buffer_alloc).
//
// Completion: after its stores a producer publishes the call's sequence number in each consumer's flag words (system
// scope, after a system-scope fence); a consumer's stream waits for them in a one-workgroup kernel (system-scope loads,
// then an acquire fence), so the next kernel on that stream reads the gathered tensor.  No host synchronisation.  A wait
// that gives up (lost peer) is STICKY: it raises a word in host-mapped memory that the host side checks on every later
// call without synchronising (parallel.PeerGather), later waits return at once instead of spinning again, and in
// production mode (`trap`) the kernel traps, so that nothing queued behind it ever consumes a stale tensor.
// CAPTURABLE FORM (round 5): the host-baked sequence number / buffer parity made a gather impossible to capture in a HIP graph
// (a replay would find the captured call's flags already set).  With a device word holding the number of the last finished call
// and ONE destination buffer whose reuse is acknowledged explicitly (tp_arrive_kernel), nothing of a call is host state: the
// three launches replay as they are, mixed freely with eager calls (the reference's all-gather is a plugin inside the engine and
// replays with it: tensorrt_llm/functional.py:3834-3880).
// Flag words: kTpFlagWords per (parity, producer); the stand-alone push below publishes word 0, the push fused into the
// GEMM epilogue (gemm_pp_kernels.hip, TpEpilogue) publishes one word per M chunk as the chunk's last tile retires.
#include "mixq_launch.h"

namespace mixq {

struct TpDest {
    void* base[kTpMaxPeers];      // [M, N] fp16 buffer of every destination rank (own rank included)
    unsigned* flag[kTpMaxPeers];  // that rank's flag words for THIS producer
};

// src [M, n_loc] fp16 (contiguous) -> dst_r[m, col0 + j] for every destination r.  16-byte vectors; each source vector
// is read once and written ndst times.  M == 0: nothing to move, the flags are still published (every rank's wait of
// this call expects them).
__global__ __launch_bounds__(256) void tp_push_columns_kernel(const uint4* __restrict__ src, TpDest d, int ndst, int M,
                                                               int vec_per_row /* n_loc / 8 */, int64_t dst_row_vecs
                                                               /* N / 8 */, int col0_vec, unsigned seq, int nflags,
                                                               unsigned* __restrict__ done_counter,
                                                               const unsigned* __restrict__ seq_word,
                                                               const unsigned* __restrict__ status)
{
    if (seq_word != nullptr) seq = *seq_word + 1u; // capturable form: the call's number lives on the device (tp_wait bumps it)
    // (capturable form, ADVICE r5: this rank's tp_arrive may have given up -- a peer has NOT acknowledged that its single destination
    //  buffer may be overwritten and can still be reading the previous call's tensor.  With the sticky status raised nothing is stored and
    //  no flag is published: the peers' waits of this call time out in turn instead of consuming a torn tensor.)
    if (status != nullptr && __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) return;
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
    if (last == gridDim.x - 1u) {
        if (threadIdx.x == 0)
            __hip_atomic_store(done_counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // re-armed (stream-ordered reuse)
        __threadfence_system();
        for (int i = threadIdx.x; i < ndst * nflags; i += 256)
            __hip_atomic_store(d.flag[i / nflags] + (i % nflags), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// One workgroup: waits until flags[r * kTpFlagWords + w] == seq for every producer r < nprod and word word0 <= w <
// word0 + nwords (written by producer r into THIS rank's flag block).  Bounded: gives up after `patience` ticks of the
// 100 MHz wall clock (default ~2 s) and raises status[0] (host-mapped, sticky); a raised status[0] makes every later
// wait return at once; `trap` != 0 kills the queue instead of letting consumers behind it run on a stale tensor.
__global__ __launch_bounds__(256) void tp_wait_flags_kernel(const unsigned* __restrict__ flags, int nprod, int word0,
                                                             int nwords, unsigned seq, unsigned* __restrict__ status,
                                                             int trap, unsigned long long patience,
                                                             unsigned* __restrict__ seq_word)
{
    if (seq_word != nullptr) seq = *seq_word + 1u;
    __shared__ unsigned failed;
    if (threadIdx.x == 0) failed = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (failed == 0) {
        const unsigned long long t0 = wall_clock64();
        for (int i = threadIdx.x; i < nprod * nwords; i += 256) {
            const unsigned* p = flags + (size_t)(i / nwords) * kTpFlagWords + word0 + (i % nwords);
            while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
                __builtin_amdgcn_s_sleep(16);
                if (wall_clock64() - t0 > patience) {
                    __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    __hip_atomic_store(status + 1, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // which call
                    failed = 1;
                    break;
                }
            }
        }
    }
    __threadfence_system(); // acquire at system scope: what the producers wrote before their flags is visible from here on
    __syncthreads();
    if (trap && failed) __builtin_trap();
    if (seq_word != nullptr && threadIdx.x == 0) *seq_word = seq; // (every thread read the old value before the barrier above)
}

// Capturable form, step 1 of 3 (tp_arrive -> tp_push_columns -> tp_wait, all reading the call number s = *seq_word + 1):
// this rank has reached call s on its stream -- whatever read the gathered tensor of call s - 1 is earlier in the stream and
// done -- so its ONE destination buffer may be overwritten: publish s in every producer's acknowledge words, then wait until
// every destination of OUR push has acknowledged s.  One workgroup; bounded like tp_wait_flags_kernel.
struct TpAck {
    unsigned* peer[kTpMaxPeers]; // rank r's acknowledge word for THIS consumer
};
__global__ __launch_bounds__(64) void tp_arrive_kernel(TpAck a, const unsigned* __restrict__ own_acks, int npeer,
                                                        const unsigned* __restrict__ seq_word, unsigned* __restrict__ status,
                                                        int trap, unsigned long long patience)
{
    const unsigned seq = *seq_word + 1u;
    __shared__ unsigned failed;
    if (threadIdx.x == 0) failed = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __syncthreads();
    if (failed == 0 && (int)threadIdx.x < npeer) {
        __hip_atomic_store(a.peer[threadIdx.x], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned* p = own_acks + (size_t)threadIdx.x * kTpFlagWords;
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != seq) {
            __builtin_amdgcn_s_sleep(16);
            if (wall_clock64() - t0 > patience) {
                __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(status + 1, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                failed = 1;
                break;
            }
        }
    }
    __threadfence_system();
    __syncthreads();
    if (trap && failed) __builtin_trap();
}

hipError_t launch_tp_push(const void* src, void* const* dst_bases, unsigned* const* dst_flags, int ndst, int M, int n_loc,
                          int N, int col0, unsigned seq, int nflags, unsigned* done_counter, hipStream_t st,
                          const unsigned* seq_word, const unsigned* status)
{
    if (ndst < 1 || ndst > kTpMaxPeers || n_loc % 8 || N % 8 || col0 % 8 || nflags < 1 || nflags > kTpFlagWords || M < 0 ||
        n_loc <= 0)
        return hipErrorInvalidValue;
    TpDest d{};
    for (int r = 0; r < ndst; ++r) d.base[r] = dst_bases[r], d.flag[r] = dst_flags[r];
    const int64_t total = (int64_t)M * (n_loc / 8);
    int64_t blocks = (total + 255) / 256;
    const int64_t cap = (int64_t)num_cus() * 4; // enough to saturate 7 links; the rest of the chip keeps computing
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1; // M == 0: one workgroup that only publishes the flags
    hipLaunchKernelGGL(tp_push_columns_kernel, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const uint4*>(src), d,
                       ndst, M, n_loc / 8, (int64_t)(N / 8), col0 / 8, seq, nflags, done_counter, seq_word, status);
    return hipGetLastError();
}

hipError_t launch_tp_wait(const unsigned* flags, int nprod, int word0, int nwords, unsigned seq, unsigned* status,
                          int trap, unsigned patience_ms, hipStream_t st, unsigned* seq_word)
{
    if (nprod < 1 || nprod > kTpMaxPeers || word0 < 0 || nwords < 1 || word0 + nwords > kTpFlagWords)
        return hipErrorInvalidValue;
    const unsigned long long ticks = (unsigned long long)(patience_ms ? patience_ms : 2000u) * 100000ull; // 100 MHz
    hipLaunchKernelGGL(tp_wait_flags_kernel, dim3(1), dim3(256), 0, st, flags, nprod, word0, nwords, seq, status, trap,
                       ticks, seq_word);
    return hipGetLastError();
}

hipError_t launch_tp_arrive(unsigned* const* peer_acks, const unsigned* own_acks, int npeer, const unsigned* seq_word,
                            unsigned* status, int trap, unsigned patience_ms, hipStream_t st)
{
    if (npeer < 1 || npeer > kTpMaxPeers) return hipErrorInvalidValue;
    TpAck a{};
    for (int r = 0; r < npeer; ++r) a.peer[r] = peer_acks[r];
    const unsigned long long ticks = (unsigned long long)(patience_ms ? patience_ms : 2000u) * 100000ull; // 100 MHz
    hipLaunchKernelGGL(tp_arrive_kernel, dim3(1), dim3(64), 0, st, a, own_acks, npeer, seq_word, status, trap, ticks);
    return hipGetLastError();
}

} // namespace mixq

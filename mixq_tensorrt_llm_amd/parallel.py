"""Multi-GPU layout of the MixQ linear: shard ROWS of W (output features) across ranks, one all-gather of the fp16
output -- and only when TP > 1 (BASELINE.json north_star).

One process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).
The reference's only collective is an ``allreduce`` after an out_features//tp_size split (plugin.py:97,155-156),
which is shape-wrong for an N-split and guarded by ``assert tp_size == 1``; it is deliberately not reproduced.

Every tensor that is per-output-feature is sharded the same way: W int8 [N,K], sW [N], fp_weight [N,128] and the
decode ``qweight`` [K,N] (by column pairs).  ``fp_ind`` and the activations are replicated; the per-token
quantisation pre-pass is recomputed on every rank (HBM-bound, no communication).
"""
import ctypes
from typing import Dict, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_total: int, tp_size: int, rank: int, multiple: int = 16) -> Tuple[int, int]:
    """Contiguous [n0, n1) slice of the output features owned by ``rank``; N/tp must keep the operator's
    16-element alignment (all BASELINE shapes give multiples of 256)."""
    assert n_total % tp_size == 0, f"N={n_total} not divisible by tp_size={tp_size}"
    per = n_total // tp_size
    assert per % multiple == 0, f"N/tp={per} must be a multiple of {multiple}"
    return rank * per, (rank + 1) * per


def shard_packed(packed: Dict[str, np.ndarray], tp_size: int, rank: int) -> Dict[str, np.ndarray]:
    """Slice the 7-tensor contract of one layer (pack.pack_linear_weights output, true dtypes) for ``rank``.

    ``qweight`` is sliced in the interleaved image: output-feature pair j owns bytes [j*2K, (j+1)*2K), so a
    contiguous range of pairs is a contiguous byte range (see csrc/decode_kernels.hip)."""
    N, K = packed["weight"].shape
    n0, n1 = shard_bounds(N, tp_size, rank)
    out = dict(packed)
    out["weight"] = np.ascontiguousarray(packed["weight"][n0:n1])
    out["weights_scaling_factor"] = np.ascontiguousarray(packed["weights_scaling_factor"][n0:n1])
    out["fp_weight"] = np.ascontiguousarray(packed["fp_weight"][n0:n1])
    qw = np.ascontiguousarray(packed["qweight"]).reshape(-1)
    out["qweight"] = qw[n0 * K:n1 * K].reshape(K, n1 - n0).copy()
    if "scales" in packed:
        out["scales"] = np.ascontiguousarray(packed["scales"][n0:n1])
    if packed.get("bias") is not None:  # per output feature like sW: each rank adds its slice before the gather
        assert packed["bias"].shape == (N,), f"bias {packed['bias'].shape} vs N={N}"
        out["bias"] = np.ascontiguousarray(packed["bias"][n0:n1])
    return out


def _is_gloo(group) -> bool:
    try:
        return dist.get_backend(group) == "gloo"
    except Exception:  # noqa: BLE001
        return False


def all_gather_columns(x_local: torch.Tensor, group=None, tp_size: int = None, run_trivial: bool = False) -> torch.Tensor:
    """[.., N/tp] per rank -> [.., N] on every rank with ONE collective (ncclAllGather under RCCL).

    RCCL moves contiguous buffers, so the gather lands rank-major ([tp, M, N/tp]); one strided copy puts the column
    blocks side by side (csrc/tp_kernels.hip's peer writes -- parallel.PeerGather -- avoid that pass where the ranks can
    map each other's output buffers).  Message per rank per call = M * N/tp * 2 bytes to each of the tp-1 peers.
    A gloo group (CPU tests, and the 2-ranks-on-one-GPU test) is served through host staging.
    Capturable in a HIP graph on an RCCL group (the collective is enqueued on the current stream, like the reference's allgather
    plugin inside a TensorRT engine: tensorrt_llm/functional.py:3834-3880)."""
    if tp_size is None:
        tp_size = dist.get_world_size(group)
    if tp_size == 1 and not run_trivial:   # (run_trivial: issue the collective on a one-rank group too -- the RCCL pre-flight test)
        return x_local
    lead = x_local.shape[:-1]
    n_loc = x_local.shape[-1]
    x2 = x_local.reshape(-1, n_loc).contiguous()
    dev = x2.device
    if x2.is_cuda and _is_gloo(group):
        x2 = x2.cpu()
    gathered = torch.empty((tp_size * x2.shape[0], n_loc), dtype=x2.dtype, device=x2.device)
    dist.all_gather_into_tensor(gathered, x2, group=group)  # rank-major concatenation along dim 0
    full = gathered.view(tp_size, x2.shape[0], n_loc).permute(1, 0, 2).reshape(x2.shape[0], tp_size * n_loc)
    return full.reshape(*lead, tp_size * n_loc).to(dev)


class _RawDeviceBuffer:
    """A device allocation of the library seen through ``__cuda_array_interface__`` (zero-copy torch view)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class PeerGatherTimeout(RuntimeError):
    """A peer never published its flags: the gathered tensor of that call (and of every later one) is not valid."""


_MEM_KINDS = {"coarse": 0, "finegrained": 1, "uncached": 2}


class PeerGather:
    """All-gather of the output columns as ONE-SIDED PEER WRITES over xGMI (csrc/tp_kernels.hip, include/mixq.h
    ``mixq_tp_*``): every rank writes its [M, N/tp] block straight into its column block of every rank's [M, N] buffer,
    so the gathered tensor lands in its final layout -- no rank-major staging, no permute pass -- and all 7 links of a GPU
    carry traffic at once.  Completion travels as a sequence number in the consumer's flag words; the consumer's STREAM
    waits for it (no host sync), so the call overlaps with whatever runs on other streams.

    Memory: the destination buffers are fine-grained and the flag block uncached device memory (``hipExtMallocWithFlags``,
    like RCCL's own buffers / flags) -- a remote GPU writes them while this GPU polls and reads; system-scope atomics are
    only specified on such allocations.  ``MIXQ_TP_DATA_MEM`` / ``MIXQ_TP_FLAG_MEM`` = coarse | finegrained | uncached
    override the kinds (experiments only).

    Two destination buffers alternate by call parity.  Contract: whatever reads the tensor returned by call i must be
    enqueued on the same stream before call i + 1 (then a fast rank can never overwrite data a slow rank still reads:
    it needs the slow rank's flag of call i + 1 before its own call i + 2 is pushed, and the slow rank publishes that
    flag only after its readers of call i, which are earlier in its stream).

    In this (default) form a call is NOT capturable in a HIP graph: the sequence number and the buffer parity are host-side
    state baked into the kernel arguments, so a replay would find the flags of the captured call already set and read a
    half-written buffer; ``gather`` refuses a capturing stream.

    ``capturable=True`` (every rank of the group must agree) selects the form whose calls CAN be captured and replayed, as the
    reference's all-gather plugin replays inside its engine (tensorrt_llm/functional.py:3834-3880): the call number lives in a
    device word that the wait kernel bumps, every rank has ONE destination buffer (fixed address: what a graph needs), and its
    reuse is acknowledged explicitly -- ``mixq_tp_arrive`` + ``mixq_tp_push_columns_seq`` + ``mixq_tp_wait_seq``, three launches
    per gather instead of two, no host state per call.  Captured and eager calls mix freely on one object; the contract is the
    same (readers of call i are on the stream before call i + 1).  The GEMM-fused transport (``enqueue_gather``) is not offered in
    this form (it returns None: operator + ``gather``).

    Failure: a wait gives up after ``patience_ms`` (lost / hung peer) and raises a STICKY status word in host-mapped
    memory; every later ``gather`` (and ``check()``) raises ``PeerGatherTimeout`` without synchronising the device, and
    later waits return at once instead of spinning again.  With ``trap_on_timeout`` the waiting kernel also traps, so that
    nothing queued behind it consumes the stale tensor (the stream's next synchronisation fails) -- the production
    setting; the self-test of ``bench.py`` keeps it off so that it can fall back to RCCL.

    One process per GPU; the buffers are exchanged as 64-byte IPC handles through ``torch.distributed`` (any backend)."""

    FLAG_WORDS = 64   # per (parity, producer): include/mixq.h MIXQ_TP_FLAG_WORDS

    def __init__(self, max_m: int, n_total: int, tp_size: int, rank: int, device, group=None, trap_on_timeout=False,
                 patience_ms: int = 2000, capturable: bool = False):
        import os
        from . import _lib
        assert n_total % tp_size == 0 and (n_total // tp_size) % 8 == 0 and 1 < tp_size <= 8
        self.lib = _lib.load()
        self.dev = torch.device(device)
        self.M, self.N, self.tp, self.rank = int(max_m), int(n_total), int(tp_size), int(rank)
        self.n_loc = self.N // self.tp
        self.trap, self.patience_ms = int(bool(trap_on_timeout)), int(patience_ms)
        self.data_kind = os.environ.get("MIXQ_TP_DATA_MEM", "finegrained")
        self.flag_kind = os.environ.get("MIXQ_TP_FLAG_MEM", "uncached")
        self.data_bytes = (self.M * self.N * 2 + 255) // 256 * 256
        self.flag_bytes = 2 * 8 * self.FLAG_WORDS * 4          # [parity][producer][FLAG_WORDS] uint32
        self.seq = 0
        self.capturable = bool(capturable)
        self.own, handles = [], []

        def alloc(nbytes, kind):
            ptr = ctypes.c_void_p()
            h = ctypes.create_string_buffer(64)
            _lib.check(self.lib.mixq_tp_buffer_alloc(nbytes, _MEM_KINDS[kind], ctypes.byref(ptr), h),
                       f"mixq_tp_buffer_alloc({kind})")
            self.own.append(ptr.value)
            handles.append(h.raw)

        with torch.cuda.device(self.dev):
            alloc(self.data_bytes, self.data_kind)
            alloc(self.data_bytes, self.data_kind)
            alloc(self.flag_bytes, self.flag_kind)
            hp, dp = ctypes.c_void_p(), ctypes.c_void_p()
            _lib.check(self.lib.mixq_tp_status_alloc(ctypes.byref(hp), ctypes.byref(dp)), "mixq_tp_status_alloc")
        self._status_host_ptr, self._status_dev = hp.value, dp.value
        self._status = (ctypes.c_uint32 * 16).from_address(hp.value)
        everyone = [None] * self.tp
        dist.all_gather_object(everyone, handles, group=group)
        self.peer = [[None, None, None] for _ in range(self.tp)]  # [rank][data 0 | data 1 | flags] -> pointer in THIS process
        self._opened = []
        with torch.cuda.device(self.dev):
            for r in range(self.tp):
                for which in range(3):
                    if r == self.rank:
                        self.peer[r][which] = self.own[which]
                        continue
                    ptr = ctypes.c_void_p()
                    buf = ctypes.create_string_buffer(everyone[r][which], 64)
                    _lib.check(self.lib.mixq_tp_buffer_open(buf, ctypes.byref(ptr)), "mixq_tp_buffer_open")
                    self.peer[r][which] = ptr.value
                    self._opened.append(ptr.value)
        self.views = [torch.as_tensor(_RawDeviceBuffer(p, self.data_bytes), device=self.dev) for p in self.own[:2]]
        self.small = torch.zeros(self.FLAG_WORDS + 2, dtype=torch.int32, device=self.dev)   # [0] push counter, [1..64] chunk counters, [65] call number (capturable form)
        dist.barrier(group=group)  # every rank has every buffer mapped before the first push

    # flag words of producer `prod` in rank r's block of parity `par`
    def _flag_ptr(self, r: int, par: int, prod: int) -> int:
        return self.peer[r][2] + 4 * self.FLAG_WORDS * (8 * par + prod)

    def check(self, sync: bool = False):
        """Raises PeerGatherTimeout if any wait so far gave up.  sync=False reads the host-mapped status word as it is
        (no device synchronisation: a failure of a call still in flight shows at the next check)."""
        if sync:
            torch.cuda.synchronize(self.dev)
        if self._status[0] != 0:
            raise PeerGatherTimeout(f"rank {self.rank}: a peer never published its flags for gather call "
                                    f"{int(self._status[1])} (tp={self.tp}, N={self.N}); results from that call on are invalid")

    def _begin(self, m: int):
        assert m <= self.M
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("PeerGather.gather is not graph-capturable (host-side sequence number / buffer parity)")
        self.check()
        self.seq += 1
        return self.seq & 1

    def destinations(self, par: int):
        """(bases, flags) ctypes arrays of this call: every rank's buffer of parity `par` and, in it, THIS producer's words."""
        bases = (ctypes.c_void_p * self.tp)(*[self.peer[r][par] for r in range(self.tp)])
        flags = (ctypes.c_void_p * self.tp)(*[self._flag_ptr(r, par, self.rank) for r in range(self.tp)])
        return bases, flags

    def _wait(self, par: int, nwords: int, st):
        from . import _lib
        _lib.check(self.lib.mixq_tp_wait(self._flag_ptr(self.rank, par, 0), self.tp, 0, nwords, self.seq,
                                         self._status_dev, self.trap, self.patience_ms, st), "mixq_tp_wait")

    def _view(self, par: int, m: int) -> torch.Tensor:
        return self.views[par][: m * self.N * 2].view(torch.float16).view(m, self.N)

    def _gather_capturable(self, x_local: torch.Tensor, m: int) -> torch.Tensor:
        """arrive -> push -> wait with the call number on the device: nothing here depends on how often it has been called."""
        from . import _lib
        self.check()      # (host-mapped status word; harmless during capture)
        st = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        seq_word = ctypes.c_void_p(self.small.data_ptr() + 4 * (self.FLAG_WORDS + 1))
        # data flags: parity block 0, one word per producer; acknowledge words: parity block 1, one word per consumer
        bases = (ctypes.c_void_p * self.tp)(*[self.peer[r][0] for r in range(self.tp)])
        flags = (ctypes.c_void_p * self.tp)(*[self._flag_ptr(r, 0, self.rank) for r in range(self.tp)])
        acks = (ctypes.c_void_p * self.tp)(*[self._flag_ptr(r, 1, self.rank) for r in range(self.tp)])
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.mixq_tp_arrive(acks, self._flag_ptr(self.rank, 1, 0), self.tp, seq_word, self._status_dev,
                                               self.trap, self.patience_ms, st), "mixq_tp_arrive")
            _lib.check(self.lib.mixq_tp_push_columns_seq(x_local.data_ptr() if m else None, bases, flags, self.tp, m, self.n_loc,
                                                         self.N, self.rank * self.n_loc, seq_word, self.small.data_ptr(), self._status_dev, st),
                       "mixq_tp_push_columns_seq")
            _lib.check(self.lib.mixq_tp_wait_seq(self._flag_ptr(self.rank, 0, 0), self.tp, seq_word, self._status_dev,
                                                 self.trap, self.patience_ms, st), "mixq_tp_wait_seq")
        return self._view(0, m)

    def gather(self, x_local: torch.Tensor) -> torch.Tensor:
        """x_local fp16 [m, N/tp] (m <= max_m, contiguous) -> fp16 [m, N] view of this rank's buffer of the call's parity,
        valid on the current stream once the returned tensor's producer kernels (push + wait) have run."""
        assert x_local.is_cuda and x_local.dtype == torch.float16 and x_local.is_contiguous()
        m = x_local.numel() // self.n_loc
        assert x_local.shape[-1] == self.n_loc
        from . import _lib
        if self.capturable:
            assert m <= self.M
            return self._gather_capturable(x_local, m)
        par = self._begin(m)
        st = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        bases, flags = self.destinations(par)
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.mixq_tp_push_columns(x_local.data_ptr() if m else None, bases, flags, self.tp, m,
                                                     self.n_loc, self.N, self.rank * self.n_loc, self.seq, 1,
                                                     self.small.data_ptr(), st), "mixq_tp_push_columns")
            self._wait(par, 1, st)
        return self._view(par, m)

    def enqueue_gather(self, plug, inputs, workspace=None):
        """The operator AND its all-gather in one pass (``mixq_enqueue_tp``): the GEMM's store path writes this rank's
        column block straight into every rank's buffer, chunk flags are published as the M chunks retire -- no
        [m, N/tp] output, no push launch, no second read.  ``plug``: plugin.MixQPlugin of the shard; ``inputs``: the 7
        carriers of ``MixQPlugin.enqueue``.  Returns the gathered fp16 [m, N] view, or None when the shape does not take
        the 256 x 256 ping-pong kernel (caller: ``plug.enqueue`` + ``gather``)."""
        from . import _lib
        A = inputs[0]
        K = A.shape[-1]
        m = A.numel() // K
        assert inputs[1].shape[0] == self.n_loc and A.is_cuda and A.is_contiguous()
        if self.capturable or not self.lib.mixq_tp_fused_supported(m, self.n_loc, K):
            return None
        in_desc = (_lib.TensorDesc * 7)(*[_lib.TensorDesc.make(t.shape) for t in inputs])
        in_ptrs = (ctypes.c_void_p * 7)(*[t.data_ptr() for t in inputs])
        if workspace is None:
            workspace = plug._workspace(A.device, plug.workspace_size(max(m, 1), self.n_loc, K))
        return self.enqueue_gather_raw(plug._h, in_desc, in_ptrs, ctypes.c_void_p(workspace.data_ptr()), m, K)

    def enqueue_gather_raw(self, handle, in_desc, in_ptrs, ws_ptr, m: int, K: int):
        """``enqueue_gather`` on prepared ctypes blocks (bench.py); the shape must be ``mixq_tp_fused_supported``."""
        from . import _lib
        assert not self.capturable, "the GEMM-fused transport carries a host-side sequence number"
        par = self._begin(m)
        st = ctypes.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        tp = _lib.TpEpilogue()
        tp.ndst, tp.n_total, tp.col0, tp.seq = self.tp, self.N, self.rank * self.n_loc, self.seq
        for r in range(self.tp):
            tp.dst_bases[r] = self.peer[r][par]
            tp.dst_flags[r] = self._flag_ptr(r, par, self.rank)
        tp.counters = self.small.data_ptr() + 4
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.mixq_enqueue_tp(handle, in_desc, in_ptrs, ws_ptr, ctypes.byref(tp), st), "mixq_enqueue_tp")
            self._wait(par, int(self.lib.mixq_tp_flag_words(m)), st)
        return self._view(par, m)

    def timed_out(self) -> bool:
        """True if a wait gave up (a peer never published): synchronises the device first."""
        torch.cuda.synchronize(self.dev)
        return bool(self._status[0] != 0)

    def close(self, group=None):
        torch.cuda.synchronize(self.dev)
        dist.barrier(group=group)        # nobody frees memory a peer may still write
        with torch.cuda.device(self.dev):
            for p in self._opened:
                self.lib.mixq_tp_buffer_close(ctypes.c_void_p(p))
            self.views = []
            for p in self.own:
                self.lib.mixq_tp_buffer_free(ctypes.c_void_p(p))
            if self._status_host_ptr:
                self._status = None
                self.lib.mixq_tp_status_free(ctypes.c_void_p(self._status_host_ptr))
                self._status_host_ptr = None
        self._opened, self.own = [], []

"""GPU (-m gpu): the one-launch decode-batch operator (csrc/gemm_fusedq_kernels.hip: quantiser inside the skinny GEMM)
through mixq_enqueue.  It must produce the SAME bits as the two-launch form (quant_extract_kernel + skinny GEMM,
`mixq_debug_set_gemm_variant(80)`), the same workspace contents (qA / sA / fpA bit-exact against the oracle), survive
its own help path (variant 82: every workgroup times out at once and quantises the missing rows itself), re-arm its
flags (hundreds of calls on one workspace), replay from a HIP graph, and run on two workspaces / streams at once."""
import ctypes

import numpy as np
import pytest

from conftest import make_layer

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

REL_TOL = 1e-3


def dev():
    return torch.device("cuda:0")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint16)


def rel_err(got, want):
    g, w = got.astype(np.float64), want.astype(np.float64)
    return np.abs(g - w).max() / max(np.abs(w).max(), 1e-30)


@pytest.fixture
def knob():
    from mixq_tensorrt_llm_amd import _lib
    lib = _lib.load()
    yield lib.mixq_debug_set_gemm_variant
    lib.mixq_debug_set_gemm_variant(81)


class Call:
    """One prepared mixq_enqueue call with its own workspace (so qA / sA / fpA can be read back)."""

    def __init__(self, A, p, stream=None):
        from mixq_tensorrt_llm_amd import _lib
        from mixq_tensorrt_llm_amd._lib import TensorDesc
        self.lib = _lib.load()
        self.M, self.K = A.shape
        self.N = p["weight"].shape[0]
        d = dev()
        f16 = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(d)  # noqa: E731
        self.A = f16(A)
        self.ins = [self.A, f16(p["weight"]).view(torch.float16), f16(p["weights_scaling_factor"]),
                    f16(p["fp_weight"]), f16(p["fp_ind"].astype(np.int32)).view(torch.float16),
                    f16(p["qweight"]).view(torch.float16), f16(p["weights_scaling_factor"])]
        self.out = torch.full((self.M, self.N), float("nan"), dtype=torch.float16, device=d)
        self.h = ctypes.c_void_p(self.lib.mixq_create(self.M, self.N, self.K))
        n = self.lib.mixq_workspace_size(self.h, self.M, self.N, self.K)
        self.ws = torch.full((n,), 0xAB, dtype=torch.uint8, device=d)   # garbage, like a shared engine workspace
        self.in_desc = (TensorDesc * 7)(*[TensorDesc.make(t.shape) for t in self.ins])
        self.out_desc = TensorDesc.make(self.out.shape)
        self.in_ptrs = (ctypes.c_void_p * 7)(*[t.data_ptr() for t in self.ins])
        self.out_ptrs = (ctypes.c_void_p * 1)(self.out.data_ptr())
        self.stream = stream

    def run(self):
        st = self.stream if self.stream is not None else torch.cuda.current_stream()
        rc = self.lib.mixq_enqueue(self.h, self.in_desc, ctypes.byref(self.out_desc), self.in_ptrs, self.out_ptrs,
                                   ctypes.c_void_p(self.ws.data_ptr()), ctypes.c_void_p(st.cuda_stream))
        assert rc == 0, rc
        return self

    def result(self):
        torch.cuda.synchronize()
        return self.out.cpu().numpy()

    def workspace_parts(self):
        """qA, sA, fpA as carved by enqueue (128-byte aligned regions, TsinghuaMixQPlugin.cpp:404-421)."""
        torch.cuda.synchronize()
        base = self.ws.data_ptr()
        al = lambda x: (x + 127) & ~127  # noqa: E731
        o0 = al(base) - base
        o1 = al(base + o0 + self.M * self.K) - base
        o2 = al(base + o1 + 2 * self.M) - base
        w = self.ws.cpu().numpy()
        qA = w[o0:o0 + self.M * self.K].view(np.int8).reshape(self.M, self.K)
        sA = w[o1:o1 + 2 * self.M].view(np.float16)
        fpA = w[o2:o2 + 2 * 128 * self.M].view(np.float16).reshape(self.M, 128)
        return qA, sA, fpA


SHAPES = [(5, 512, 256), (8, 4096, 4096), (16, 12288, 4096), (17, 4096, 1024), (32, 4096, 4096), (32, 4112, 5120),
          (31, 1040, 2048), (9, 16, 128), (24, 7168, 3584)]


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_one_launch_equals_two_launches_and_the_oracle(oracle, knob, M, N, K):
    A, W, act = make_layer(M, N, K, seed=M + N + K)
    if M > 6:  # the quantiser's special rows: zero row, NaN, inf, subnormal amax
        A[1] = 0
        A[2, 7] = np.nan
        A[3, K // 2] = np.float16(6.5e4)
        A[4, :] = np.float16(6e-8)
    p = oracle.pack_linear_weights(W, act)
    from mixq_tensorrt_llm_amd import _lib
    lib = _lib.load()
    knob(81)
    c = Call(A, p).run()
    one = c.result()
    assert b"fusedq" in lib.mixq_debug_last_gemm_kernel(), lib.mixq_debug_last_gemm_kernel()
    qA, sA, fpA = c.workspace_parts()
    qo, so = oracle.quant_rows(A)
    assert np.array_equal(qA, qo) and np.array_equal(bits(sA), bits(so))
    assert np.array_equal(bits(fpA), bits(A[:, p["fp_ind"]]))
    knob(80)
    two = Call(A, p).run().result()
    assert b"fusedq" not in lib.mixq_debug_last_gemm_kernel()
    with np.errstate(invalid="ignore"):
        assert np.array_equal(bits(one), bits(two)), "one launch != two launches"
    knob(82)  # every workgroup times out immediately and quantises what is missing itself
    helped = Call(A, p).run().result()
    assert np.array_equal(bits(helped), bits(one))
    want = oracle.linear_prefill(A, p["weight"], p["weights_scaling_factor"], p["fp_weight"], p["fp_ind"])
    ok = ~np.isnan(want)
    assert np.array_equal(np.isnan(one), np.isnan(want))
    assert np.abs(one[ok].astype(np.float64) - want[ok].astype(np.float64)).max() <= REL_TOL * np.abs(want[ok]).max()


def test_flags_are_rearmed_over_many_calls_with_changing_data(oracle, knob):
    knob(81)
    M, N, K = 32, 4096, 4096
    A, W, act = make_layer(M, N, K, seed=3)
    p = oracle.pack_linear_weights(W, act)
    c = Call(A, p)
    knob(80)
    ref = Call(A, p)
    knob(81)
    rng = np.random.default_rng(0)
    for it in range(300):
        A2 = (rng.standard_normal((M, K)) * rng.uniform(0.1, 5)).astype(np.float16)
        x = torch.from_numpy(A2).to(dev())
        c.A.copy_(x)
        ref.A.copy_(x)
        c.run()
        if it % 50 == 0 or it == 299:
            got = c.result()
            knob(80)
            want = ref.run().result()
            knob(81)
            assert np.array_equal(bits(got), bits(want)), it


def test_graph_replay_and_two_workspaces_on_two_streams(oracle, knob):
    knob(81)
    M, N, K = 32, 4096, 4096
    A, W, act = make_layer(M, N, K, seed=5)
    p = oracle.pack_linear_weights(W, act)
    knob(80)
    want = Call(A, p).run().result()
    knob(81)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    c1, c2 = Call(A, p, s1), Call(A, p, s2)
    for _ in range(200):       # different workspaces -> different flag slots: concurrent launches do not interfere
        c1.run()
        c2.run()
    assert np.array_equal(bits(c1.result()), bits(want)) and np.array_equal(bits(c2.result()), bits(want))
    # graph: capture one call, replay on new data in the same buffers
    g = torch.cuda.CUDAGraph()
    c3 = Call(np.zeros_like(A), p, s1)
    with torch.cuda.stream(s1):
        c3.run()
        s1.synchronize()
        with torch.cuda.graph(g, stream=s1):
            c3.run()
    for trial in range(3):
        A2 = np.ascontiguousarray(np.roll(A, trial + 1, axis=0))
        c3.A.copy_(torch.from_numpy(A2).to(dev()))
        torch.cuda.synchronize()
        g.replay()
        got = c3.result()
        knob(80)
        ref = Call(A2, p).run().result()
        knob(81)
        assert np.array_equal(bits(got), bits(ref)), trial


def test_under_a_co_running_kernel(oracle, knob):
    """A long matmul stream on another queue competes for the CUs while the one-launch operator runs 500 times."""
    knob(81)
    M, N, K = 16, 4096, 4096
    A, W, act = make_layer(M, N, K, seed=8)
    p = oracle.pack_linear_weights(W, act)
    knob(80)
    want = Call(A, p).run().result()
    knob(81)
    hog = torch.cuda.Stream()
    x = torch.randn((4096, 4096), device=dev(), dtype=torch.float16)
    c = Call(A, p)
    with torch.cuda.stream(hog):
        for _ in range(60):
            x = (x @ x).clamp_(-1, 1)
    for _ in range(500):
        c.run()
    assert np.array_equal(bits(c.result()), bits(want))
    torch.cuda.synchronize()

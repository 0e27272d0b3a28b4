"""GPU (-m gpu): weight images (include/mixq.h ``mixq_weight_image_*``, MI355X extension): a registered fragment-major copy of the
int8 weight is what decode-batch calls on that weight pointer stream.  Same bytes into the same MFMA lanes -- every result must be
the bits of the unregistered call; the image must be the documented layout; unregistering must restore the plain path."""
import ctypes

import numpy as np
import pytest

from conftest import assert_prefill_parity, make_layer

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _layer(oracle, N, K, seed):
    from mixq_tensorrt_llm_amd import plugin
    A, W, act = make_layer(64, N, K, seed=seed)
    p = oracle.pack_linear_weights(W, act)
    return A, p, plugin.MixQLinear(K, N, device="cuda:0").load(p)


def test_image_is_the_documented_layout_and_shapes_without_one_are_refused():
    from mixq_tensorrt_llm_amd import _lib, mixlib
    lib = _lib.load()
    g = torch.Generator(device="cuda:0").manual_seed(3)
    N, K = 272, 704 - 704 % 64   # 17 feature tiles x 10 k-steps
    W = torch.randint(-128, 128, (N, K), dtype=torch.int32, device="cuda:0", generator=g).to(torch.int8)
    img = mixlib.WeightImage(W)
    torch.cuda.synchronize()
    want = W.view(N // 16, 16, K // 64, 4, 16).permute(0, 2, 3, 1, 4).contiguous().view(-1)   # [tile][k-step][k quarter][feature][16 B]
    assert torch.equal(img.image, want)
    img.close()
    assert lib.mixq_weight_image_unregister(ctypes.c_void_p(W.data_ptr())) != 0     # already gone
    assert lib.mixq_weight_image_bytes(N, K) == N * K
    assert lib.mixq_weight_image_bytes(N + 8, K) == 0 and lib.mixq_weight_image_bytes(N, K + 16) == 0
    with pytest.raises(ValueError):
        mixlib.WeightImage(torch.zeros((24, 64), dtype=torch.int8, device="cuda:0"))


@pytest.mark.parametrize("N,K", [(4096, 4096), (5120, 5120), (4816, 1024), (12288, 1024), (512, 8192), (272, 704)])
def test_registered_calls_are_bit_identical(oracle, N, K):
    """5 .. 64 rows: one to four 16-row tiles, 16 and 32 features per workgroup (5120 x 5120; 4816 = 301 feature tiles: the last workgroup's second tile does not exist), shapes the skinny GEMM does not serve
    (the image is then simply not read), K with a ragged last k-step (704: no image exists), both load policies."""
    from mixq_tensorrt_llm_amd import _lib
    lib = _lib.load()
    A, p, layer = _layer(oracle, N, K, seed=N + K)
    plain = {}
    for M in (5, 16, 17, 32, 33, 48, 64, 3, 200):
        x = torch.from_numpy(np.concatenate([A] * 4)[:M]).to("cuda:0")
        plain[M] = (x, layer(x).clone())
    if K % 64:
        with pytest.raises(ValueError):
            layer.prepare_decode_batches()
        return
    layer.prepare_decode_batches()
    try:
        for knob in (880, 881, 882):
            lib.mixq_debug_set_gemm_variant(knob)
            for M, (x, want) in plain.items():   # (an image may move a long-K shape from the K-split tiles to the skinny GEMM: same bits)
                got = layer(x)
                torch.cuda.synchronize()
                assert torch.equal(got, want), (knob, M)
    finally:
        lib.mixq_debug_set_gemm_variant(880)
    A32 = np.concatenate([A] * 4)[:32]
    assert_prefill_parity(oracle, layer(torch.from_numpy(A32).to("cuda:0")).cpu().numpy(), A32, p, f"weight image {N}x{K}")
    # the image follows the weight POINTER: rewriting the weights without re-registering would stream stale bytes, so load() re-registers
    A2, p2, _ = _layer(oracle, N, K, seed=N + K + 1)
    layer.load(p2)
    assert layer.weight_image is not None
    x = torch.from_numpy(A2[:32]).to("cuda:0")
    got = layer(x).cpu().numpy()
    assert_prefill_parity(oracle, got, A2[:32], p2, f"weight image after reload {N}x{K}")
    layer.prepare_decode_batches(False)
    assert layer.weight_image is None
    assert np.array_equal(layer(x).cpu().numpy().view(np.uint16), got.view(np.uint16))


def test_graph_replay_and_the_mixlib_flavour(oracle):
    """HIP-graph capture bakes the image pointer in; MixLinear_GEMM (P-flavour, one-call forward) picks its own image up."""
    from mixq_tensorrt_llm_amd import mixlinear
    A, p, layer = _layer(oracle, 4096, 1024, seed=9)
    x = torch.from_numpy(A[:24]).to("cuda:0")
    want = layer(x).clone()
    layer.prepare_decode_batches()
    gr = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        layer(x)
        s.synchronize()
        with torch.cuda.graph(gr, stream=s):
            out = layer(x)
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    assert torch.equal(out, want)
    layer.prepare_decode_batches(False)
    # P-flavour
    rng = np.random.default_rng(4)
    Wf = torch.from_numpy((rng.standard_normal((2048, 1024)) * 0.02).astype(np.float16))
    cache = mixlinear.MixLibCache(inputdim=64, device="cuda:0")
    lin = mixlinear.MixLinear_GEMM.from_linear(Wf, None, bit=8, cache=cache, dev="cuda:0")
    lin.ind = torch.from_numpy(rng.permutation(1024)[:128].astype(np.int32)).to("cuda:0")
    lin.weight_cache = mixlinear.dequant_weight_columns(lin.q_weight, lin.scale_col, lin.ind)
    lin.add_outliers = False
    xa = torch.from_numpy(A[:40, :1024].copy()).to("cuda:0")
    y0 = lin.forward(xa.clone(), cache, True).clone()
    lin.prepare_decode_batches()
    y1 = lin.forward(xa.clone(), cache, True)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    lin.prepare_decode_batches(False)


def test_a_weight_replaced_behind_the_same_address_is_not_served_the_old_image(oracle):
    """the registry is keyed by address.  The library records a content tag at registration and compares the bytes
    behind the pointer with it around the image's FIRST USE (asynchronously: round 6): a tensor whose content was replaced in place (= freed and re-allocated at the
    same address) without unregistering gets results from its OWN bytes, the entry is dropped, the stale counter moves; the same after
    a first use is what `verify` is for."""
    from mixq_tensorrt_llm_amd import _lib, mixlib, plugin
    lib = _lib.load()
    N, K, M = 4096, 4096, 32
    A, W1, act = make_layer(M, N, K, seed=1)
    _, W2, _ = make_layer(M, N, K, seed=2)
    p1, p2 = oracle.pack_linear_weights(W1, act), oracle.pack_linear_weights(W2, act)
    layer = plugin.MixQLinear(K, N, device="cuda:0").load(p1)
    x = torch.from_numpy(A).to("cuda:0")
    stale0 = lib.mixq_weight_image_stale_count()
    img = mixlib.WeightImage(layer.weight.view(torch.int8))                 # registered, not used yet
    layer.weight.view(torch.int8).copy_(torch.from_numpy(p2["weight"]))     # ... the memory changes hands behind the library's back
    layer.fp_weight.copy_(torch.from_numpy(p2["fp_weight"]).view(torch.float16).reshape(layer.fp_weight.shape))
    layer.weights_scaling_factor.copy_(torch.from_numpy(p2["weights_scaling_factor"]))
    got = layer(x).cpu().numpy()                                            # first use: the check is enqueued, the call reads `weight` itself
    torch.cuda.synchronize()
    assert np.array_equal(layer(x).cpu().numpy(), got)                      # the check has completed: tag mismatch -> the entry is dropped
    assert lib.mixq_weight_image_stale_count() == stale0 + 1
    assert_prefill_parity(oracle, got, A, dict(p2, fp_ind=p1["fp_ind"]), "replaced weight, stale image dropped")
    assert not img.verify()                                                 # nothing registered any more
    img.close()
    # after a verified first use: verify() tells a later replacement, and drops the entry
    img = mixlib.WeightImage(layer.weight.view(torch.int8))
    got2 = layer(x).cpu().numpy()
    assert np.array_equal(got2, got) and img.verify()
    layer.weight.view(torch.int8).copy_(torch.from_numpy(p1["weight"]))
    assert not img.verify() and lib.mixq_weight_image_stale_count() == stale0 + 2
    img.close()


def test_first_use_inside_a_capture_reads_the_weight_itself(oracle):
    """An image that has never been used cannot be verified while its stream is being captured (verification synchronises): the captured
    call reads `weight`; after one eager call the image is trusted and a later capture streams it.  Same bits either way."""
    from mixq_tensorrt_llm_amd import _lib
    lib = _lib.load()
    A, p, layer = _layer(oracle, 4096, 4096, seed=9)
    x = torch.from_numpy(A[:32]).to("cuda:0")
    want = layer(x).clone()
    torch.cuda.synchronize()
    layer.prepare_decode_batches()
    try:
        outs = []
        for eager_first in (False, True):
            if eager_first:
                assert torch.equal(layer(x), want)
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                with torch.cuda.graph(g, stream=s):
                    o = layer(x)
            g.replay()
            torch.cuda.synchronize()
            assert torch.equal(o, want)
            outs.append((g, o))
    finally:
        layer.prepare_decode_batches(False)

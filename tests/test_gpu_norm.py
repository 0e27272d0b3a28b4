"""GPU (-m gpu): SURVEY §8f row 1 -- fused RMSNorm -> extract -> quant producer (mixlib
layernorm_forward_cuda[_extract_outliers]) against the oracle.

The reference reduces sum(x^2) in an unspecified fp32 order and uses CUDA's approximate rsqrtf, so the normalised row
is compared with a 1-fp16-ulp tolerance (1e-3 relative); everything downstream of the normalised row is then checked
BIT-EXACTLY by re-running the oracle's extract+quant on the GPU's own normalised row."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def bits(a):
    return np.ascontiguousarray(a).view(np.uint16)


@pytest.mark.parametrize("M,K,O", [(3, 256, 16), (37, 1024, 128), (64, 4096, 128), (9, 8192, 128), (6, 11008, 128),
                                   (4, 28672, 128)])
def test_fused_rmsnorm_extract_quant(oracle, M, K, O):
    from mixq_tensorrt_llm_amd import mixlib
    rng = np.random.default_rng(M + K)
    x = (rng.standard_normal((M, K)) * rng.uniform(0.1, 8)).astype(np.float16)
    gamma = (1.0 + 0.1 * rng.standard_normal(K)).astype(np.float16)
    ind = rng.permutation(K)[:O].astype(np.int32)
    x[:, ind[: O // 2]] *= np.float16(15)
    if M > 2:
        x[1] = 0
    eps = 1e-5
    xd = torch.from_numpy(x).to(dev()).reshape(1, M, K)     # mixlib takes [b, n, c]
    out = torch.empty_like(xd)
    scale = torch.empty(M, dtype=torch.float16, device=dev())
    outl, q = mixlib.layernorm_forward_cuda_extract_outliers(xd, torch.from_numpy(gamma).to(dev()), out, eps,
                                                             torch.from_numpy(ind).to(dev()), scale)
    out_np = out.cpu().numpy().reshape(M, K)
    o_ref, outl_ref, q_ref, s_ref = oracle.rmsnorm_extract_quant(x, gamma, eps, ind)
    # (1) normalised row within one fp16 ulp of the ideal value
    denom = np.maximum(np.abs(o_ref.astype(np.float64)), 1e-3)
    assert (np.abs(out_np.astype(np.float64) - o_ref.astype(np.float64)) / denom).max() < 1.1e-3
    assert np.all(out_np[:, ind] == 0)
    # (2) given the GPU's normalised row, the gathered outliers / scale / int8 rows are bit-exact
    norm_full = out_np.copy()
    norm_full[:, ind] = outl.cpu().numpy()                  # undo the zeroing to recover the pre-extraction row
    assert (np.abs(outl.cpu().numpy().astype(np.float64) - outl_ref.astype(np.float64))
            / np.maximum(np.abs(outl_ref.astype(np.float64)), 1e-3)).max() < 1.1e-3
    qz, sz = oracle.quant_rows(out_np)
    assert np.array_equal(bits(scale.cpu().numpy()), bits(sz))
    assert np.array_equal(q.cpu().numpy(), qz)
    # (3) and end to end the int8 rows differ from the ideal ones by at most one step, rarely
    dq = np.abs(q.cpu().numpy().astype(np.int32) - q_ref.astype(np.int32))
    assert dq.max() <= 1 and dq.mean() < 0.02
    assert np.abs(scale.cpu().numpy().astype(np.float64) - s_ref.astype(np.float64)).max() <= \
        1.1e-3 * np.abs(s_ref.astype(np.float64)).max()


@pytest.mark.parametrize("M,K", [(5, 512), (33, 4096), (7, 11008)])
def test_plain_rmsnorm(oracle, M, K):
    from mixq_tensorrt_llm_amd import mixlib
    rng = np.random.default_rng(K)
    x = (rng.standard_normal((M, K)) * 3).astype(np.float16)
    gamma = (1.0 + 0.1 * rng.standard_normal(K)).astype(np.float16)
    xd = torch.from_numpy(x).to(dev())
    out = torch.empty_like(xd)
    mixlib.layernorm_forward_cuda(xd, torch.from_numpy(gamma).to(dev()), out, 1e-6)
    ref = oracle.rmsnorm_extract_quant(x, gamma, 1e-6)
    denom = np.maximum(np.abs(ref.astype(np.float64)), 1e-3)
    assert (np.abs(out.cpu().numpy().astype(np.float64) - ref.astype(np.float64)) / denom).max() < 1.1e-3


def test_fused_producer_feeds_the_gemm(oracle):
    """P-flavour pipeline on the GPU: fused norm/extract/quant -> int8FusedDequantize with the outlier product as y
    (MixQ/src/mixquant/modules/linear.py:243-270), against the same pipeline evaluated by the oracle on the GPU's
    normalised activations."""
    from mixq_tensorrt_llm_amd import mixlib
    rng = np.random.default_rng(3)
    M, N, K, O = 160, 512, 1024, 128
    x = rng.standard_normal((M, K)).astype(np.float16)
    gamma = np.ones(K, np.float16)
    ind = rng.permutation(K)[:O].astype(np.int32)
    x[:, ind] *= np.float16(12)
    Wq = rng.integers(-127, 128, size=(N, K), dtype=np.int8)
    Wq[:, ind] = 0
    sW = (rng.random(N) * 1e-3 + 1e-4).astype(np.float16)
    fpW = (rng.standard_normal((N, O)) * 0.02).astype(np.float16)
    xd = torch.from_numpy(x).to(dev())
    out = torch.empty_like(xd)
    scale = torch.empty(M, dtype=torch.float16, device=dev())
    outl, q = mixlib.layernorm_forward_cuda_extract_outliers(xd, torch.from_numpy(gamma).to(dev()), out, 1e-5,
                                                             torch.from_numpy(ind).to(dev()), scale)
    y = torch.mm(outl.float(), torch.from_numpy(fpW).to(dev()).float().t()).to(torch.float16)  # plumbing only
    got = mixlib.int8FusedDequantize(q, torch.from_numpy(Wq).to(dev()), scale, torch.from_numpy(sW).to(dev()), y,
                                     M, N, K).cpu().numpy()
    want = oracle.dequant_epilogue(oracle.gemm_s8s8s32(q.cpu().numpy(), Wq), scale.cpu().numpy(), sW, y.cpu().numpy())
    assert np.array_equal(bits(got), bits(want))

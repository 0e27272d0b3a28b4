"""GPU (-m gpu): a checkpoint written by checkpoint.save_checkpoint loads into plugin.MixQLinear and runs."""
import numpy as np
import pytest

from conftest import assert_prefill_parity, make_layer

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def test_checkpoint_to_mixqlinear_forward(tmp_path, oracle):
    from mixq_tensorrt_llm_amd import checkpoint, pack, plugin
    A, W, act = make_layer(40, 256, 512, seed=12)
    packed = pack.pack_linear_weights(torch.from_numpy(W), torch.from_numpy(act))
    prefix = checkpoint.layer_prefix(3, "mlp.gate")
    checkpoint.save_checkpoint(str(tmp_path), {prefix: packed})
    _, loaded = checkpoint.load_checkpoint(str(tmp_path))
    layer = checkpoint.load_linear(plugin.MixQLinear(512, 256, device="cuda:0"), loaded[prefix])
    got = layer(torch.from_numpy(A).to("cuda:0")).cpu().numpy()
    assert_prefill_parity(oracle, got, A, packed, "checkpoint -> MixQLinear")

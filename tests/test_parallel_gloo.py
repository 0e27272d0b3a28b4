"""CPU, world_size 2, gloo: the N>1 layout (rows of W sharded, one all-gather of the fp16 output).  Each rank runs the
operator on its shard -- here through the oracle, because there is no GPU in this container; on the GPU box the same
sharding helpers feed the HIP path (tests/test_gpu_parity.py::test_tp_shards_compose) -- and the gathered result must
equal the unsharded operator bit for bit (output columns are independent)."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, make_layer


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from mixq_tensorrt_llm_amd import pack, parallel
    A, W, act = make_layer(24, 64, 256, seed=9)
    full = pack.pack_linear_weights(torch.from_numpy(W), torch.from_numpy(act))
    full["bias"] = np.arange(64, dtype=np.float16)          # per output feature: sharded like sW (Qwen2 qkv has one)
    mine = parallel.shard_packed(full, world, rank)
    out_local = oracle.linear_prefill(A, mine["weight"], mine["weights_scaling_factor"], mine["fp_weight"],
                                      mine["fp_ind"])
    gathered = parallel.all_gather_columns(torch.from_numpy(out_local), None, world)
    want = oracle.linear_prefill(A, full["weight"], full["weights_scaling_factor"], full["fp_weight"], full["fp_ind"])
    ok = np.array_equal(gathered.numpy().view(np.uint16), want.view(np.uint16))
    # decode weights: the interleaved image shards by contiguous byte ranges of column pairs
    n0, n1 = parallel.shard_bounds(64, world, rank)
    ok &= np.array_equal(mine["bias"], full["bias"][n0:n1])
    q_un = oracle.eetq_symmetric_quantize(W.T.copy())[0]
    ok &= np.array_equal(mine["qweight"], oracle.eetq_preprocess(np.ascontiguousarray(q_un[:, n0:n1])))
    # 3-D activations keep their leading dims
    g3 = parallel.all_gather_columns(torch.from_numpy(out_local).reshape(2, 12, -1), None, world)
    ok &= tuple(g3.shape) == (2, 12, 64) and torch.equal(g3.reshape(24, 64), gathered)
    open(os.path.join(tmp, f"ok{rank}"), "w").write("1" if ok else "0")
    dist.barrier()
    dist.destroy_process_group()


def test_row_sharded_linear_allgather_world2(tmp_path, oracle):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    for r in range(world):
        assert open(tmp_path / f"ok{r}").read() == "1"


def test_shard_bounds_alignment():
    from mixq_tensorrt_llm_amd import parallel
    import pytest
    assert parallel.shard_bounds(10240, 8, 3) == (3840, 5120)       # Llama-2-70B qkv, SURVEY A.5
    assert parallel.shard_bounds(28672, 8, 7) == (25088, 28672)
    with pytest.raises(AssertionError):
        parallel.shard_bounds(4096 + 8, 8, 0)

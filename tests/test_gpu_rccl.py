"""GPU (-m gpu): RCCL pre-flight (VERDICT r4 #3a).  north_star names "a single RCCL all-gather over xGMI"; the boxes the tests run on
have ONE GPU, so RCCL's first execution would otherwise happen on the driver's 8-GPU node.  Here it runs with a one-rank group:
``init_process_group("nccl")`` (communicator creation, the topology / IPC set-up RCCL does at init), ``all_gather_into_tensor`` on
device tensors through ``parallel.all_gather_columns`` -- eagerly AND captured in a HIP graph, as a TP decode step would replay it --
and one ``MixQLinear(tp_size = 1, gather_output = True)`` call against the oracle, all in a process of its own."""
import os
import subprocess
import sys
import textwrap

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = textwrap.dedent("""
    import os, sys
    sys.path.insert(0, os.getcwd())
    import numpy as np
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
    from mixq_tensorrt_llm_amd import pack, parallel, plugin
    import oracle
    # 1. the collective itself, eager: RCCL executes ncclAllGather on a one-rank communicator
    x = torch.randn((64, 512), device=dev).to(torch.float16)
    full = parallel.all_gather_columns(x, None, 1, run_trivial=True)
    torch.cuda.synchronize()
    assert full.shape == x.shape and torch.equal(full, x)
    t = torch.ones(4, device=dev)
    dist.all_reduce(t)
    dist.barrier()
    torch.cuda.synchronize()
    # 2. the same collective INSIDE a captured graph (a TP decode step replays as one graph), replayed on fresh data
    xs = torch.zeros((32, 256), dtype=torch.float16, device=dev)
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        parallel.all_gather_columns(xs, None, 1, run_trivial=True)   # (communicator work outside the capture)
        s.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            ys = parallel.all_gather_columns(xs, None, 1, run_trivial=True)
    for i in range(3):
        xs.copy_(torch.full((32, 256), float(i + 1), dtype=torch.float16, device=dev))
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(ys, xs), i
    # 3. one MixQLinear(tp_size = 1, gather_output = True) call in the process that holds the RCCL communicator, vs the oracle
    rng = np.random.default_rng(0)
    M, N, K = 96, 512, 1024
    act = np.abs(rng.standard_normal(K)).astype(np.float32)
    W = (rng.standard_normal((N, K)) * 0.02).astype(np.float16)
    A = rng.standard_normal((M, K)).astype(np.float32)
    p = pack.pack_linear_weights(torch.from_numpy(W), torch.from_numpy(act))
    A[:, p["fp_ind"]] *= 20
    A = A.astype(np.float16)
    layer = plugin.MixQLinear(K, N, tp_group=dist.group.WORLD, tp_size=1, gather_output=True, device=dev).load(p)
    out = layer(torch.from_numpy(A).to(dev)).cpu().numpy()
    want = oracle.linear_prefill(A, p["weight"], p["weights_scaling_factor"], p["fp_weight"], p["fp_ind"])
    err = np.abs(out.astype(np.float64) - want.astype(np.float64)).max() / np.abs(want.astype(np.float64)).max()
    assert err < 1e-3, err
    dist.destroy_process_group()
    print("RCCL_PREFLIGHT_OK")
""")


def test_rccl_one_rank_preflight():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_PREFLIGHT_OK" in r.stdout, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])

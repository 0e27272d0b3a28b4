"""GPU (-m gpu): the kernel-selection tables as a GATE.  tools/selection_check.py re-measures, for a probe on both
sides of every table row (gemm_splitk_plan, gemm_pp128_wins, deep_plan_auto, wo_skinny_pick, wo_wide_plan, gemm_takes_skinny), the
automatic choice against each alternative the row chose between (forced through the debug knobs, device-paced HIP graphs, cold
weights where the row was fitted cold) and flags a probe whose automatic choice is more than 10 % behind its best alternative.
Here: the quick set (one probe per row); a flagged probe is measured a second time in a fresh process and fails the test only if it
is flagged again (box-to-box spread of single launches on this pool is +-4 %).  Skipped where the CU count is not the 256 the tables
were fitted on."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "selection_check.py"), "--quick", "--tolerance", "0.10"],
                       cwd=ROOT, capture_output=True, text=True, timeout=1500, env=dict(os.environ, MIXQ_DEBUG_KNOBS="1"))
    flagged = set(re.findall(r"^\s{3}(\S+(?: \(cold\))?) (\d+x\d+x\d+): \+", r.stdout, flags=re.M))
    assert r.returncode in (0, 1), (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    assert "probe(s) more than" in r.stdout, r.stdout[-2000:]
    return r, flagged


def test_automatic_selection_is_within_10_percent_of_the_best_alternative_on_every_table_row():
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the selection tables are fitted on a 256-CU part")
    r, flagged = _run()
    if flagged:
        r2, flagged2 = _run()
        both = flagged & flagged2
        assert not both, f"flagged twice: {sorted(both)}\n{r.stdout[-3000:]}\n{r2.stdout[-3000:]}"


FAMILY = [("skinny", b"gemm_skinny_kernel"), ("deep", b"<DEEP>"), ("ping-pong 128x256", b"pp128_kernel"),
          ("ping-pong 256x256 tiles,", b"pp_kernel<SPLITK>"), ("ping-pong 256x256 tiles:", b"pp_kernel<SPLITK>"),
          ("ping-pong 256x256 tiles", b"gemm_w8a8o16_pp_kernel (256x256"), ("two-barrier", b"two-barrier tiles")]


def test_described_plan_is_what_enqueue_launches():
    """mixq_describe_plan (host only) restates launch_gemm's decisions; here every cell of a grid over BASELINE.json's (N, K) x a ladder of
    row counts is really launched through mixq_enqueue and the family the library reports afterwards must be the one described."""
    import ctypes
    sys.path.insert(0, ROOT)
    import bench
    from mixq_tensorrt_llm_amd import _lib
    from mixq_tensorrt_llm_amd._lib import TensorDesc
    lib = _lib.load()
    lib.mixq_debug_reset()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(0)
    st0 = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    buf = ctypes.create_string_buffer(320)
    n = 0
    for (N, K) in [(12288, 4096), (4096, 11008), (4096, 4096), (4608, 3584), (18944, 3584), (3584, 8192), (1024, 28672), (5120, 5120)]:
        t = bench.synth_layer(N, K, dev, gen)
        for M in (5, 17, 32, 48, 64, 96, 128, 192, 256, 384, 512, 1024, 2048):
            A = bench.synth_activation(M, K, t["ind_i32"], dev, gen)
            o = torch.empty((M, N), dtype=torch.float16, device=dev)
            ins = [A, t["weight"], t["weights_scaling_factor"], t["fp_weight"], t["fp_ind"], t["qweight"], t["weights_scaling_factor"]]
            in_desc = (TensorDesc * 7)(*[TensorDesc.make(x.shape) for x in ins])
            out_desc = TensorDesc.make(o.shape)
            h = ctypes.c_void_p(lib.mixq_create(M, N, K))
            ws = torch.empty(max(lib.mixq_workspace_size(h, M, N, K), 16), dtype=torch.uint8, device=dev)
            assert lib.mixq_enqueue(h, in_desc, ctypes.byref(out_desc), (ctypes.c_void_p * 7)(*[x.data_ptr() for x in ins]),
                                    (ctypes.c_void_p * 1)(o.data_ptr()), ctypes.c_void_p(ws.data_ptr()), st0) == 0
            torch.cuda.synchronize()
            launched = lib.mixq_debug_last_gemm_kernel()
            lib.mixq_destroy(h)
            assert lib.mixq_describe_plan(M, N, K, 0, buf, 320) == 0
            said = buf.value.decode().split(", then ")[1]
            want = next(k for p_, k in FAMILY if said.startswith(p_))
            assert want in launched, (M, N, K, said, launched)
            n += 1
    assert n == 104


@pytest.mark.parametrize("M", [32, 64])
def test_a_registered_weight_image_is_never_slower(M):
    """`mixq_weight_image_register` costs + N K bytes per layer; a call on a registered weight must not be slower than
    the same call without the image.  The three Llama-2-7B linears, cold weights (a rotation over > 320 MiB of copies, each with its own
    image), device-paced graphs: sum over the three shapes, image <= 1.03 x no image (the box-to-box spread of a launch is +-4 %; at
    64 rows both now take the same kernels -- csrc/gemm_skinny_kernels.hip skinny_weight_route)."""
    import ctypes
    sys.path.insert(0, ROOT)
    import bench
    from mixq_tensorrt_llm_amd import _lib
    from mixq_tensorrt_llm_amd._lib import TensorDesc
    lib = _lib.load()
    lib.mixq_debug_reset()
    dev = torch.device("cuda:0")
    gen = torch.Generator(device=dev).manual_seed(1)
    st0 = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    total = {False: 0.0, True: 0.0}
    for (N, K) in [(12288, 4096), (11008, 4096), (4096, 11008)]:
        t = bench.synth_layer(N, K, dev, gen)
        copies = [t["weight"]] + [t["weight"].clone() for _ in range((320 << 20) // (N * K) + 1)]
        A = bench.synth_activation(M, K, t["ind_i32"], dev, gen)
        o = torch.empty((M, N), dtype=torch.float16, device=dev)
        h = ctypes.c_void_p(lib.mixq_create(M, N, K))
        ws = torch.empty(max(lib.mixq_workspace_size(h, M, N, K), 16), dtype=torch.uint8, device=dev)
        sets = []
        for w in copies:
            ins = [A, w, t["weights_scaling_factor"], t["fp_weight"], t["fp_ind"], t["qweight"], t["weights_scaling_factor"]]
            sets.append(((TensorDesc * 7)(*[TensorDesc.make(x.shape) for x in ins]), (ctypes.c_void_p * 7)(*[x.data_ptr() for x in ins])))
        out_desc, out_ptrs = TensorDesc.make(o.shape), (ctypes.c_void_p * 1)(o.data_ptr())
        turn = [0]

        def run(st):
            d, ptrs = sets[turn[0] % len(sets)]
            turn[0] += 1
            assert lib.mixq_enqueue(h, d, ctypes.byref(out_desc), ptrs, out_ptrs, ctypes.c_void_p(ws.data_ptr()), st) == 0

        for with_image in (False, True):
            imgs = []
            if with_image:
                for w in copies:
                    im = torch.empty(N * K, dtype=torch.int8, device=dev)
                    assert lib.mixq_weight_image_register(ctypes.c_void_p(w.data_ptr()), N, K, ctypes.c_void_p(im.data_ptr()), st0) == 0
                    assert lib.mixq_weight_image_verify(ctypes.c_void_p(w.data_ptr()), st0) == 0
                    imgs.append(im)
            torch.cuda.synchronize()
            turn[0] = -1
            total[with_image] += min(bench.graph_time_us(run, dev, calls=len(sets) * 4, reps=10) for _ in range(2))
            for w in copies:
                lib.mixq_weight_image_unregister(ctypes.c_void_p(w.data_ptr()))
        lib.mixq_destroy(h)
    assert total[True] <= 1.03 * total[False], total

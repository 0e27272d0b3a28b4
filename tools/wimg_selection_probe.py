#!/usr/bin/env python3
"""With a weight image registered, does the skinny GEMM win decode batches that the tile kernels take today?  Operator (mixq_enqueue)
us per call, HIP graph of 100 calls, cold weights (cycling copies): [no image, default selection] vs [image, default selection] vs
[image, skinny forced up to 32 rows for any N (892) / up to 64 rows (897)] vs [image, small-tile K split off (60)]."""
import ctypes
import os

os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from mixq_tensorrt_llm_amd import _lib  # noqa: E402
from mixq_tensorrt_llm_amd._lib import TensorDesc  # noqa: E402

CASES = [(12288, 4096, (40, 48, 64)), (11008, 4096, (48, 64)), (18944, 3584, (24, 32, 48, 64)), (28672, 8192, (24, 32, 48)),
         (8192, 8192, (32, 48, 64)), (4096, 11008, (32, 48, 64)), (3584, 18944, (16, 32, 48)), (1024, 28672, (16, 32, 48)),
         (4096, 4096, (48, 64)), (6144, 4096, (64,)), (10240, 8192, (32, 48, 64))]


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    gen = torch.Generator(device=dev).manual_seed(0)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    for N, K, Ms in CASES:
        t = bench.synth_layer(N, K, dev, gen)
        copies = (320 << 20) // (N * K) + 2
        ws = [t["weight"]] + [t["weight"].clone() for _ in range(copies - 1)]
        imgs = [torch.empty(N * K, dtype=torch.int8, device=dev) for _ in ws]
        for M in Ms:
            A = bench.synth_activation(M, K, t["ind_i32"], dev, gen)
            o = torch.empty((M, N), dtype=torch.float16, device=dev)
            ins = [A, t["weight"], t["weights_scaling_factor"], t["fp_weight"], t["fp_ind"], t["qweight"], t["weights_scaling_factor"]]
            in_desc = (TensorDesc * 7)(*[TensorDesc.make(x.shape) for x in ins])
            out_desc = TensorDesc.make(o.shape)
            out_ptrs = (ctypes.c_void_p * 1)(o.data_ptr())
            h = ctypes.c_void_p(lib.mixq_create(M, N, K))
            wsp = torch.empty(max(lib.mixq_workspace_size(h, 64, N, K), 16), dtype=torch.uint8, device=dev)
            sets = []
            for w in ws:
                v = [x.data_ptr() for x in ins]
                v[1] = w.data_ptr()
                sets.append((ctypes.c_void_p * 7)(*v))
            turn = [0]

            def cold(st):
                assert lib.mixq_enqueue(h, in_desc, ctypes.byref(out_desc), sets[turn[0] % copies], out_ptrs, p(wsp), st) == 0
                turn[0] += 1
            cells, ref = [], None
            for label, image, knobs in (("no image", False, ()), ("image", True, ()), ("image+892", True, (892,)), ("image+897", True, (897,)),
                                        ("image+60", True, (60,)), ("image+897+60", True, (897, 60))):
                lib.mixq_debug_reset()
                st0 = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                for w, im in zip(ws, imgs):
                    if image:
                        assert lib.mixq_weight_image_register(p(w), N, K, p(im), st0) == 0
                    else:
                        lib.mixq_weight_image_unregister(p(w))
                for k in knobs:
                    lib.mixq_debug_set_gemm_variant(k)
                turn[0] = 0
                o.zero_()
                cold(st0)
                torch.cuda.synchronize()
                kern = lib.mixq_debug_last_gemm_kernel().decode().split(" ")[0].replace("gemm_", "").replace("w8a8o16_", "")
                if ref is None:
                    ref = o.clone()
                ok = torch.equal(o, ref)
                tc = bench.graph_time_us(cold, dev)
                cells.append(f"{label}: {tc:5.2f} [{kern}{'' if ok else ' MISMATCH'}]")
            lib.mixq_debug_reset()
            for w in ws:
                lib.mixq_weight_image_unregister(p(w))
            lib.mixq_destroy(h)
            print(f"M={M:3d} N={N:6d} K={K:6d} cold us | " + " | ".join(cells), flush=True)
        del ws, imgs


if __name__ == "__main__":
    main()

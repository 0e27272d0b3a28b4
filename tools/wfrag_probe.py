#!/usr/bin/env python3
"""Weight images (include/mixq.h mixq_weight_image_*): the int8 decode-batch operator (quantiser + skinny GEMM) with the WEIGHT operand
read from its registered fragment-major copy (1-KiB blocks [16-feature tile][64-byte k-step]: one contiguous read of whole cache
lines per load instruction) instead of the reference's row-major int8 [N, K] (one load = 64 bytes of 16 rows, K bytes apart): images
ignored (883), plain loads (881), non-temporal loads (882), automatic (880).  Warm (same weights every call) and cold (cycling
through > 320 MiB of weight copies), HIP graph of 100 operator calls; bit identity against the row-major form; the image itself
against a torch permute.  usage: python tools/wfrag_probe.py [--shapes "4096 4096;12288 4096;11008 4096;4096 11008"] [--Ms 8,32]"""
import argparse
import ctypes
import os

os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from mixq_tensorrt_llm_amd import _lib  # noqa: E402


def w_image(W8):
    """int8 [N, K] -> fragment-major image (lane l = feature % 16 + 16 * (k / 16 % 4) holds 16 bytes at l * 16 of block (tile, k-step))."""
    N, K = W8.shape
    return W8.view(N // 16, 16, K // 64, 4, 16).permute(0, 2, 3, 1, 4).contiguous().view(N, K)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", default="4096 4096;12288 4096;11008 4096;4096 11008")
    ap.add_argument("--Ms", default="8,32")
    ap.add_argument("--combos", default="883;881;882;880", help="';'-separated knob lists, e.g. '880;880,896' (896 = two feature tiles per workgroup)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    gen = torch.Generator(device=dev).manual_seed(0)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    O = 128
    for shape in a.shapes.split(";"):
        N, K = (int(x) for x in shape.split())
        t = bench.synth_layer(N, K, dev, gen)
        W8 = t["weight"].view(torch.int8).reshape(N, K)
        copies = (320 << 20) // (N * K) + 2
        rows = [W8] + [W8.clone() for _ in range(copies - 1)]
        st0 = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        imgs = [torch.empty(N * K, dtype=torch.int8, device=dev) for _ in rows]
        for w, im in zip(rows, imgs):
            assert lib.mixq_weight_image_register(p(w), N, K, p(im), st0) == 0
        torch.cuda.synchronize()
        assert torch.equal(imgs[0].view(N, K), w_image(rows[0])), "image differs from the layout definition"
        sW, fpW, ind = t["weights_scaling_factor"], t["fp_weight"], t["ind_i32"]
        for M in [int(x) for x in a.Ms.split(",")]:
            A = bench.synth_activation(M, K, ind, dev, gen)
            lay = int(lib.mixq_qa_layout(M, N, K))
            if lay != 1:
                print(f"M={M} N={N} K={K}: not served by the fragment-major skinny form")
                continue
            q = torch.empty(int(lib.mixq_qa_bytes(M, K, lay)), dtype=torch.int8, device=dev)
            sA = torch.empty(M, dtype=torch.float16, device=dev)
            fpA = torch.empty((M, O), dtype=torch.float16, device=dev)
            out = torch.empty((M, N), dtype=torch.float16, device=dev)
            turn = [0]

            def op(st, ws):
                assert lib.mixq_quant_extract_layout(M, K, p(A), p(q), p(sA), p(fpA), p(ind), O, 0, lay, st) == 0
                assert lib.mixq_gemm_mixed_layout(p(q), p(ws), p(sA), p(sW), p(fpA), p(fpW), p(out), M, N, K, O, lay, None, 0, st) == 0

            cells, ref = [], None
            for combo in a.combos.split(";"):
                knob, ws = combo, rows
                lib.mixq_debug_reset()
                for k in combo.split(","):
                    lib.mixq_debug_set_gemm_variant(int(k))
                st0 = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
                out.zero_()
                op(st0, ws[0])
                torch.cuda.synchronize()
                if ref is None:
                    ref = out.clone()
                same = torch.equal(out, ref)

                def warm(st):
                    op(st, ws[0])

                def cold(st):
                    op(st, ws[turn[0] % copies])
                    turn[0] += 1
                tw, tc = bench.graph_time_us(warm, dev), bench.graph_time_us(cold, dev)
                cells.append(f"[{knob}{'' if same else ' MISMATCH'}] {tw:5.2f}/{tc:5.2f}")
            lib.mixq_debug_reset()
            print(f"M={M:3d} N={N:6d} K={K:6d} operator warm/cold us: " + "  ".join(cells) + f"   [{lib.mixq_debug_last_gemm_kernel().decode()}]", flush=True)
        for w in rows:
            assert lib.mixq_weight_image_unregister(p(w)) == 0
        del rows, imgs


if __name__ == "__main__":
    main()

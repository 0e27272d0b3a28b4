#!/usr/bin/env python3
"""Mid-M (decode batches of 96..1024 rows) in steady state: the automatic selection of the fused GEMM against every tile
configuration of the two-barrier kernel (variant 10 + i), the 128 x 256 ping-pong tiles (5), the 256 x 256 ping-pong kernel (2) and
the forced K splits (72 / 74 / 78).  us per launch.  usage: python tools/midm_cfg_sweep.py [--secs 0.25] [--Ms 128,256,512]"""
import argparse
import ctypes
import os

os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")   # measurement script: the library honours its knobs only in a process that opts in
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from mixq_tensorrt_llm_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--secs", type=float, default=0.25)
    ap.add_argument("--Ms", default="96,128,192,256,384,512,1024")
    ap.add_argument("--shapes", default="12288 4096;11008 4096;4096 11008;4096 4096")
    ap.add_argument("--only", default="", help="comma list of variant names to time (auto, plain, c0..c23, pp128, pp256, s2, s4, s8); default all but plain")
    ap.add_argument("--dump", default="", help="append one JSON line per cell with EVERY variant's time and kernel to this file")
    ap.add_argument("--cold", action="store_true", help="cycle through > 320 MiB of weight copies: every launch streams its weights from HBM")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.load()
    g = torch.Generator(device=dev).manual_seed(0)
    O = 128
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    scr = torch.zeros(int(lib.mixq_gemm_scratch_bound()) + (1 << 20), dtype=torch.uint8, device=dev)
    variants = [("auto", [0, 79, 69])] + [(f"c{i}", [70, 60, 10 + i]) for i in range(24)] + \
               [("pp128", [70, 60, 5]), ("pp256", [70, 60, 2]), ("s2", [0, 72]), ("s4", [0, 74]), ("s8", [0, 78])] + \
               [(f"d{x}", [0, 1241 + x]) for x in (1, 2, 4, 8)] + [(f"e{x}", [0, 1251 + x]) for x in (1, 2, 4, 8)] + [(f"f{x}", [0, 1261 + x]) for x in (1, 2, 4, 8)] + [(f"g{x}", [0, 1251 + x, 1239]) for x in (1, 2, 4, 8)] + [(f"h{x}", [0, 1271 + x, 1412]) for x in (1, 2, 4, 8)] + [(f"j{x}", [0, 1271 + x, 1411]) for x in (1, 2, 4, 8)] + [("w1", [0, 1272, 1412, 1431])] + [(f"x{x}", [0, 1271 + x, 1411, 1433]) for x in (2, 4)]   # x = j with 96-wide tiles only where K is not split (the rule before R6.23) + [("n1", [0, 1272, 1411, 1431]), ("m1", [0, 1272, 1411, 1432])]   # n1 / m1: not rotated, 128-wide always / 96-wide only where a sixteenth of the CUs stays free (the rule before R6.22)   # w1 = h1 on 128-wide tiles always   # round-6 schedule (gemm_mid_kernels.hip): K walk rotated per tile row (measurement option) / not (the default)   # j / k = h / i with every tile walking K from slice 0   # g = e with every wave issuing its copies before its MFMAs   # mid-M deep form (128 x 128, 4 stages in flight): 4- / 8-wave builds, x workgroups per tile
    variants += [(f"xs{x}", [1, 70, 1241, 60 + (6 if x == 16 else x)]) for x in (2, 4, 8, 16)]   # two-barrier small tiles, K split over x workgroups
    if a.only:
        variants = [v for v in variants + [("plain", [0, 70]), ("nodeep", [0, 1241]), ("r5deep", [0, 1421])] if v[0] in a.only.split(",")]   # nodeep = automatic with the mid-M deep form off   # plain = automatic with the 256 x 256 K split off
    for shape in a.shapes.split(";"):
        N, K = (int(x) for x in shape.split())
        W = torch.randint(-127, 128, (N, K), dtype=torch.int8, device=dev, generator=g)
        Ws = [W] + ([W.clone() for _ in range((320 << 20) // (N * K) + 1)] if a.cold else [])
        turn = [0]
        sW = (torch.rand(N, device=dev, generator=g) * 4e-4 + 4e-4).to(torch.float16)
        fpW = (torch.randn((N, O), device=dev, generator=g) * 0.02).to(torch.float16)
        for M in [int(x) for x in a.Ms.split(",")]:
            qA = torch.randint(-127, 128, (M, K), dtype=torch.int8, device=dev, generator=g)
            sA = (torch.rand(M, device=dev, generator=g) * 0.05 + 0.01).to(torch.float16)
            fpA = torch.randn((M, O), device=dev, generator=g).to(torch.float16)
            out = torch.empty((M, N), dtype=torch.float16, device=dev)
            res, ref = {}, None
            for name, knobs in variants:
                lib.mixq_debug_reset()
                for k in knobs:
                    lib.mixq_debug_set_gemm_variant(k)
                nscr = int(lib.mixq_gemm_scratch_size(M, N, K))
                if name in ("s2", "s4", "s8") and nscr == 0:
                    continue
                if name.startswith("xs") and nscr == 0:
                    continue
                if name[0] in "defghjwx" and name[1:].isdigit() and (nscr > scr.numel() or (nscr == 0 and name[1:] != "1")):
                    continue
                def fn():
                    w = Ws[turn[0] % len(Ws)]
                    turn[0] += 1
                    return lib.mixq_gemm_mixed_scratch(p(qA), p(w), p(sA), p(sW), p(fpA), p(fpW), p(out), M, N, K, O,
                                                       p(scr) if nscr else None, nscr, st)
                out.zero_()
                if fn() != 0:
                    continue
                torch.cuda.synchronize()
                kern = lib.mixq_debug_last_gemm_kernel().decode().split(" ")[0].replace("gemm_w8a8o16_", "")
                if ref is None:
                    ref = out.clone()
                elif not torch.equal(ref, out):
                    print(f"MISMATCH M={M} N={N} K={K} {name}")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); fn(); e1.record(); torch.cuda.synchronize()
                est = max(e0.elapsed_time(e1) / 2 * 1e-3, 1e-6)
                for _ in range(max(3, int(0.1 / est))):
                    fn()
                n = max(5, int(a.secs / est))
                e0.record()
                for _ in range(n):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                res[name] = (e0.elapsed_time(e1) / n * 1e3, kern)
            if a.dump:
                import json
                with open(a.dump, "a") as f:
                    f.write(json.dumps({"M": M, "N": N, "K": K, "cold": bool(a.cold), "t": {k: [round(v[0], 2), v[1]] for k, v in res.items()}}) + "\n")
            best = min(res.items(), key=lambda kv: kv[1][0])
            auto = res["auto"]
            top = sorted(res.items(), key=lambda kv: kv[1][0])[:5]
            print(f"M={M:5d} N={N:6d} K={K:6d} auto {auto[0]:6.1f} us [{auto[1]}]  best {best[0]} {best[1][0]:6.1f} ({(auto[0] / best[1][0] - 1) * 100:+.1f} %)  top5: "
                  + " ".join(f"{k}={v[0]:.1f}" for k, v in top), flush=True)
    lib.mixq_debug_reset()


if __name__ == "__main__":
    main()

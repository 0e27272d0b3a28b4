#!/usr/bin/env python3
"""Fits the int8 GEMM cost model t(form) = F + ceil(WGs / CUs) * bytes_per_WG / R + X(S) to a dump of tools/midm_cfg_sweep.py
(--dump: every form's time per cell) and reports, per form, the fitted constants and the relative error, and how far a selection by
the model's minimum is from the measured best.  usage: python tools/fit_int8_cost_model.py gpurun_out/r05j/forms_cold.jsonl"""
import json
import sys

import numpy as np

C = 256
# form -> (BM, BN, splits S or None for names carrying it)
TILE = {"c7": (64, 64), "c17": (64, 64), "c16": (64, 64), "c23": (64, 64), "c18": (32, 64), "c22": (32, 64), "c3": (64, 128),
        "c5": (128, 64), "c1": (128, 128), "c4": (128, 128), "pp128": (128, 256), "pp256": (256, 256)}


def wg_bytes(form, M, N, K):
    """(workgroups, operand bytes one workgroup pulls through L2 -> LDS, S)"""
    if form in TILE:
        bm, bn = TILE[form]
        s = 1
    elif form[0] == "s" and form[1:].isdigit():
        bm, bn, s = 256, 256, int(form[1:])
    elif form[0] == "e":
        bm, bn, s = 128, 128, int(form[1:])
    else:
        return None
    tiles = -(-M // bm) * -(-N // bn)
    return tiles * s, (bm + bn) * (K / s + 256), s


def main():
    rows = [json.loads(l) for l in open(sys.argv[1])]
    forms = sorted({f for r in rows for f in r["t"] if wg_bytes(f, 1, 1, 1) is not None})
    fit = {}
    for f in forms:
        X, y = [], []
        for r in rows:
            if f not in r["t"]:
                continue
            M, N, K = r["M"], r["N"], r["K"]
            if f[0] == "s" and "SPLITK" not in r["t"][f][1]:
                continue
            if f[0] == "e" and "DEEP" not in r["t"][f][1]:
                continue
            wgs, b, s = wg_bytes(f, M, N, K)
            if f[0] == "s" and wgs > C:
                continue      # (hybrid solo + tail launches: not this formula)
            rounds = -(-wgs // C)
            X.append([1.0, rounds * b / 1e3]), y.append(r["t"][f][0])     # us = F + (KB per CU) / R[GB/s -> KB/us]
        if len(y) < 6:
            continue
        X, y = np.array(X), np.array(y)
        coef, *_ = np.linalg.lstsq(X, y, rcond=None)
        pred = X @ coef
        rel = (pred - y) / y
        fit[f] = coef
        print(f"{f:6s} n={len(y):3d}  F={coef[0]:6.2f} us  R={1.0 / coef[1] if coef[1] > 0 else float('inf'):6.1f} GB/s per CU   rel err: mean |{np.abs(rel).mean() * 100:4.1f}| %  max {np.abs(rel).max() * 100:5.1f} %")
    # selection by the model's minimum vs the measured best
    worse = []
    for r in rows:
        M, N, K = r["M"], r["N"], r["K"]
        cand = {}
        for f, coef in fit.items():
            if f not in r["t"]:
                continue
            wgs, b, s = wg_bytes(f, M, N, K)
            if f[0] == "s" and wgs > C:
                continue
            cand[f] = coef[0] + coef[1] * (-(-wgs // C)) * b / 1e3
        if not cand:
            continue
        pick = min(cand, key=cand.get)
        best = min(r["t"].items(), key=lambda kv: kv[1][0])
        t_pick = r["t"][pick][0]
        worse.append((t_pick / best[1][0] - 1, M, N, K, pick, best[0], r["t"].get("nodeep", r["t"].get("auto"))[0] / best[1][0] - 1))
    w = np.array([x[0] for x in worse])
    cur = np.array([x[6] for x in worse])
    print(f"\nselection by model minimum: mean +{w.mean() * 100:.1f} % behind the measured best, max +{w.max() * 100:.1f} %, cells > 10 %: {(w > 0.10).sum()} of {len(w)}")
    print(f"selection as it is (thresholds): mean +{cur.mean() * 100:.1f} %, max +{cur.max() * 100:.1f} %, cells > 10 %: {(cur > 0.10).sum()}")
    for x in sorted(worse, reverse=True)[:15]:
        print(f"   M={x[1]:5d} N={x[2]:6d} K={x[3]:6d}  model picks {x[4]:6s} (+{x[0] * 100:.1f} %), best {x[5]}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Rewrites profiles/pmc_traffic.json from a tools/pmc_bench.sh summary (default gpurun_out/pmc_bench/summary.txt) and stamps
it with the SHA-256 of the kernel sources the counters were collected on; bench.py reports roofline.traffic = null (and says
why) when the stamp no longer matches the tree, instead of replaying bytes measured on another kernel.
usage: python tools/pmc_traffic_update.py [summary.txt] [--note "..."]"""
import argparse
import hashlib
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ("mixq_tensorrt_llm_amd/csrc/gemm_pp_kernels.hip", "mixq_tensorrt_llm_amd/csrc/mixq_device.h")
KERNEL = "gemm_w8a8o16_pp_kernel<0, true, false, 0, 0, false>"


def sources_sha256(root=ROOT):
    h = hashlib.sha256()
    for rel in SOURCES:
        h.update(open(os.path.join(root, rel), "rb").read())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("summary", nargs="?", default=os.path.join(ROOT, "gpurun_out", "pmc_bench", "summary.txt"))
    ap.add_argument("--note", default="")
    a = ap.parse_args()
    kib = {}
    for line in open(a.summary):
        m = re.match(r"(FETCH_SIZE|WRITE_SIZE) void mixq::(.+?) mean_per_dispatch_KiB ([0-9.]+) n (\d+)$", line.strip())
        if m and m.group(2) == KERNEL:
            kib[m.group(1)] = (float(m.group(3)), int(m.group(4)))
    assert set(kib) == {"FETCH_SIZE", "WRITE_SIZE"}, f"{KERNEL} not in {a.summary}"
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    d = json.load(open(path))
    d["fetch_kib_raw"], d["write_kib"] = kib["FETCH_SIZE"][0], kib["WRITE_SIZE"][0]
    # gfx950: FETCH_SIZE counts the 16-byte-per-lane streaming reads at half weight (MI355X_MICROARCH.md, "HBM")
    d["bytes_per_launch"] = int(round((2 * d["fetch_kib_raw"] + d["write_kib"]) * 1024))
    d["launches_averaged"] = kib["FETCH_SIZE"][1]
    d["kernel_sources"] = list(SOURCES)
    d["kernel_sources_sha256"] = sources_sha256()
    if a.note:
        d["collected"] = a.note
    json.dump(d, open(path, "w"), indent=2)
    print(json.dumps({k: d[k] for k in ("fetch_kib_raw", "write_kib", "bytes_per_launch", "kernel_sources_sha256")}))


if __name__ == "__main__":
    main()

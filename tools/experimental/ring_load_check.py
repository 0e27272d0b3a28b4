#!/usr/bin/env python3
"""Build-time check (no GPU) for the register-prefetch form of the tile kernel (csrc/gemm_kernels.hip, RPF > 0: asm-form
`global_load_dwordx4 v[a:b], v[c:d], off` into a ring of registers, certified only by explicit `s_waitcnt vmcnt(n)`): hipcc does not
know those loads are asynchronous.  The check compiles gemm_kernels.hip to gfx950 ISA and, for every kernel instantiated with RPF > 0,
walks the main loop twice (prologue + two trips) with a model of the in-order VMEM queue: a load's destination registers are PENDING
until an `s_waitcnt vmcnt(n)` leaves at most n younger loads outstanding; any instruction that reads or writes a pending register
is reported.  usage: python tools/ring_load_check.py   (exit code 1 on a finding)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mixq_tensorrt_llm_amd", "csrc", "gemm_kernels.hip")
LOAD = re.compile(r"\s*global_load_dwordx4 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], off")
WAIT = re.compile(r"\s*s_waitcnt.*vmcnt\((\d+)\)")
VREG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def vregs(text):
    out = set()
    for m in VREG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def check_kernel(body):
    """body: list of ISA lines of one kernel.  -> list of findings, or None if the kernel has no ring loads.
    Walks the control-flow graph from the first ring load (every path; a state = position + the queue of pending loads) until the queue
    is empty behind the main loop."""
    first = next((i for i, l in enumerate(body) if LOAD.match(l)), None)
    if first is None:
        return None
    hdr = next((i for i in range(first, len(body)) if re.match(r"^\.LBB\d+_\d+:.*Loop Header", body[i])), None)
    if hdr is None:
        return ["no loop header behind the first ring load"]
    hname = re.match(r"^\.(LBB\d+_\d+):", body[hdr]).group(1)[1:]   # "BB3_7"
    in_loop = [i for i, l in enumerate(body) if ("Header=" + hname + " ") in l or i == hdr]
    labels = {re.match(r"^(\.LBB\d+_\d+):", l).group(1): i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)}
    # the loop's blocks: from each annotated label to the next label
    loop_lines = set()
    for i in in_loop:
        k = i + 1
        loop_lines.add(i)
        while k < len(body) and not re.match(r"^\.LBB\d+_\d+:", body[k]):
            loop_lines.add(k)
            k += 1
    region_end = max(loop_lines)
    bad, seen = {}, set()
    work = [(first, ())]
    steps = 0
    while work:
        i, pend = work.pop()
        pending = [set(x) for x in pend]
        while i < len(body):
            steps += 1
            if steps > 2000000:
                return ["walk did not converge"]
            l = body[i].split(";")[0]
            t = l.strip()
            if not t or t.startswith("."):
                if t.startswith(".LBB"):
                    key = (i, tuple(frozenset(x) for x in pending))
                    if key in seen:
                        break
                    seen.add(key)
                i += 1
                continue
            if "s_endpgm" in t:
                break
            mw = WAIT.match(l)
            if mw:
                n = int(mw.group(1))
                while len(pending) > n:
                    pending.pop(0)
                if n == 0:   # the explicit drain behind the ring's last slices ends the region on every path
                    break
                i += 1
                continue
            mb = re.match(r"\s*(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
            if mb:
                tgt = labels[mb.group(2)]
                state = tuple(frozenset(x) for x in pending)
                if mb.group(1) == "s_branch":
                    i = tgt
                    continue
                work.append((tgt, state))
                i += 1
                continue
            inflight = set().union(*pending) if pending else set()
            ml = LOAD.match(l)
            if ml:
                dst = set(range(int(ml.group(1)), int(ml.group(2)) + 1))
                addr = set(range(int(ml.group(3)), int(ml.group(4)) + 1))
                if (dst | addr) & inflight:
                    bad[i + 1] = t
                pending.append(dst)
                i += 1
                continue
            if pending and re.match(r"\s*(global_|buffer_|scratch_|flat_)", l):
                bad[i + 1] = "other VMEM operation while ring loads are pending: " + t
            elif vregs(l) & inflight:
                bad[i + 1] = t
            i += 1
    return sorted(bad.items())


def main():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-fast-math", "--offload-device-only", "-S",
                               SRC, "-o", out], stderr=subprocess.DEVNULL)
        text = open(out).read()
    rc, n = 0, 0
    name, body = None, []
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
        elif name is not None:
            body.append(line)
            if "s_endpgm" in line:
                # RPF is the last template argument: ...Lb<ADMA>ELi<RPF>EEEvNS_10GemmParamsE
                mr = re.search(r"gemm_w8a8o16_kernelI.*ELi(\d+)EEEvNS_10GemmParamsE$", name)
                if mr and int(mr.group(1)) > 0 and "gemm_w8a8o16_kernelILi" in name and name.count("ELb") >= 3:
                    res = check_kernel(body)
                    if res is not None:
                        n += 1
                        print(f"{name}: {'ok' if not res else 'FINDINGS'}")
                        for r in res[:10]:
                            print("   ", r)
                        rc |= bool(res)
                name = None
    print(f"{n} kernel(s) checked")
    return 1 if (rc or n == 0) else 0


if __name__ == "__main__":
    sys.exit(main())

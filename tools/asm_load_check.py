#!/usr/bin/env python3
"""Build-time check (no GPU) for kernels that use asm-form register loads (`gload16_sbase`, mixq_device.h): hipcc does not know
those loads are asynchronous, so nothing stops it from moving or copying a destination register before the explicit
`s_waitcnt vmcnt(n)` that certifies it -- it happens when an asm load sits on a control-flow path of its own (notebook R3.15).
The check compiles the source to gfx950 ISA and, per kernel, looks between the first `s_barrier` (top of the main loop) and
the last asm load for any VALU move whose source or destination is a destination register of an asm load.  The loops are
written so that there is none; a compiler update that introduces one fails tests/test_abi.py::test_asm_load_registers.
usage: python tools/asm_load_check.py [source.hip ...]   (default: the fpA_intB GEMM)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = [os.path.join(ROOT, "mixq_tensorrt_llm_amd", "csrc", "w8a16_gemm_kernels.hip")]
ASM_LOAD = re.compile(r"\s*global_load_dwordx4 v\[(\d+):(\d+)\], v\d+, s\[\d+:\d+\]")   # (compiler loads use `off` or v[..] pairs)
MOVE = re.compile(r"\s*(v_mov_b64_e32|v_mov_b32_e32|v_accvgpr_write_b32|v_accvgpr_read_b32)\s+([^,]+),\s*(\S+)")
REG = re.compile(r"[va]\[(\d+):(\d+)\]|[va](\d+)$")


def regs(op):
    m = REG.match(op.strip())
    if not m:
        return set()
    if m.group(3) is not None:
        return {int(m.group(3))}
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


def check_isa(text):
    """-> list of (kernel, n_asm_loads, [(line, instruction), ...])"""
    out = []
    name, body = None, []
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, body = m.group(1), []
        elif name is not None:
            body.append(line)
            if "s_endpgm" in line:
                loads = [(i, ASM_LOAD.match(l)) for i, l in enumerate(body)]
                loads = [(i, m2) for i, m2 in loads if m2]
                if loads:
                    dest = set()
                    for _, m2 in loads:
                        dest.update(range(int(m2.group(1)), int(m2.group(2)) + 1))
                    first_barrier = next((i for i, l in enumerate(body) if "s_barrier" in l), 0)
                    last_load = loads[-1][0]
                    bad = []
                    for i in range(first_barrier, last_load + 1):
                        mv = MOVE.match(body[i])
                        if mv and mv.group(2).strip().startswith("v") and (regs(mv.group(2)) & dest):
                            bad.append((i + 1, body[i].strip()))
                        elif mv and mv.group(3).startswith("v") and (regs(mv.group(3)) & dest):
                            bad.append((i + 1, body[i].strip()))
                    out.append((name, len(loads), bad))
                name = None
    return out


def compile_to_isa(src):
    with tempfile.TemporaryDirectory() as d:
        o = os.path.join(d, "k.s")
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17",
                        "-fno-fast-math", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), src, "-o", o],
                       check=True, stderr=subprocess.DEVNULL)
        return open(o).read()


def main():
    rc = 0
    for src in sys.argv[1:] or DEFAULT:
        res = check_isa(compile_to_isa(src))
        for name, n, bad in res:
            print(f"{'FAIL' if bad else 'ok  '} {name[:90]}: {n} asm loads, {len(bad)} moves on their registers inside the loop")
            for b in bad[:8]:
                print("      ", b)
            rc |= bool(bad)
        if not res:
            print(f"(no kernel with asm-form register loads in {src})")
    return rc


if __name__ == "__main__":
    sys.exit(main())

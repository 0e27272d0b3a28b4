# decode batches on WIDE outputs: two-barrier tiles (893) vs the fragment-major skinny kernel with 1 (892+895) / 2 (892+896)
# feature tiles per workgroup; operator time through mixq_enqueue (HIP graph of 100 calls)
cd "$(dirname "$0")/../.."
python - <<'PY'
import ctypes, sys, os, io, contextlib
sys.path.insert(0, os.getcwd())
import torch
import importlib.util
spec = importlib.util.spec_from_file_location("tl", "tools/small_m_timeline.py"); tl = importlib.util.module_from_spec(spec); spec.loader.exec_module(tl)
from mixq_tensorrt_llm_amd import _lib
lib = _lib.load()
shapes = [(32, 4096, 11008), (16, 4096, 11008), (48, 4096, 11008), (32, 3584, 18944), (32, 1024, 28672), (32, 8192, 8192), (16, 1280, 8192), (48, 4096, 4096), (64, 4096, 4096), (48, 12288, 4096)]
_unused = [(32, 12288, 4096), (16, 12288, 4096), (8, 12288, 4096), (32, 11008, 4096), (32, 8192, 4096), (32, 4096, 4096), (16, 4096, 4096), (32, 18944, 3584), (32, 28672, 8192), (20, 10240, 8192), (32, 6144, 4096), (32, 5120, 5120)]
for (M, N, K) in shapes:
    row = []
    for name, knobs in (("rule", (893, 69)), ("small-tile K split off", (893, 60))):
        for v in knobs:
            lib.mixq_debug_set_gemm_variant(v)
        sys.argv = ["x", "--M", str(M), "--N", str(N), "--K", str(K)]
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                tl.main()
        except Exception:
            pass
        line = [l for l in buf.getvalue().splitlines() if "no stamps" in l][0]
        kern = [l for l in buf.getvalue().splitlines() if l.startswith("# M=")][0].split("kernel:")[1].strip()[:22]
        row.append(f"{name} {line.split(':')[1].strip().split()[0]} ({kern})")
    print(f"{M}x{N}x{K}: " + " | ".join(row), flush=True)
lib.mixq_debug_set_gemm_variant(893); lib.mixq_debug_set_gemm_variant(894); lib.mixq_debug_set_gemm_variant(69)
PY

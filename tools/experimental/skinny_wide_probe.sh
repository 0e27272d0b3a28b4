# decode batches of 17..32 rows on WIDE outputs: two-barrier tiles (rule of round 1) vs the skinny kernel with fragment-major qA
cd "$(dirname "$0")/../.."
python - <<'PY'
import ctypes, sys, os
sys.path.insert(0, os.getcwd())
import torch
sys.argv = ["x"]
import importlib.util
spec = importlib.util.spec_from_file_location("tl", "tools/small_m_timeline.py"); tl = importlib.util.module_from_spec(spec); spec.loader.exec_module(tl)
from mixq_tensorrt_llm_amd import _lib
lib = _lib.load()
for (M, N, K) in ((32, 12288, 4096), (24, 12288, 4096), (17, 12288, 4096), (32, 11008, 4096), (32, 8192, 4096), (32, 18944, 3584), (32, 28672, 8192), (20, 10240, 8192)):
    for v in (893, 892):
        lib.mixq_debug_set_gemm_variant(v)
        sys.argv = ["x", "--M", str(M), "--N", str(N), "--K", str(K)]
        import io, contextlib
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                tl.main()
        except Exception:
            pass
        line = [l for l in buf.getvalue().splitlines() if "no stamps" in l][0]
        kern = [l for l in buf.getvalue().splitlines() if l.startswith("# M=")][0]
        print(v, kern, "|", line.split(":")[1].strip())
PY

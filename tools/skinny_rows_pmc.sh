#!/usr/bin/env bash
# round 5, R5.8: L2 / fabric counters of the decode-batch GEMM with the weight read as 64-byte fragment pieces (knob 885, gemm_skinny_kernel<..., 0>)
# and in 256-byte runs (knob 884, <..., 3>): 12288 x 4096 at 32 rows, cold weights (8 copies cycled), one process, separate --pmc passes.
set -u
OUT="$PWD/gpurun_out/skinny_rows_pmc"; rm -rf "$OUT"; mkdir -p "$OUT"
export TMPDIR=/tmp MIXQ_DEBUG_KNOBS=1
cat > "$OUT/work.py" <<'PY'
import ctypes, os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"] if "GRAFT_REPO_ROOT" in os.environ else "/root/repo")
import torch, bench
from mixq_tensorrt_llm_amd import _lib
from mixq_tensorrt_llm_amd._lib import TensorDesc
lib = _lib.load(); dev = torch.device("cuda:0"); gen = torch.Generator(device=dev).manual_seed(0)
st0 = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
M, N, K = 32, 12288, 4096
t = bench.synth_layer(N, K, dev, gen)
alts = [t["weight"]] + [t["weight"].clone() for _ in range(7)]
A = bench.synth_activation(M, K, t["ind_i32"], dev, gen)
o = torch.empty((M, N), dtype=torch.float16, device=dev)
ins = [A, t["weight"], t["weights_scaling_factor"], t["fp_weight"], t["fp_ind"], t["qweight"], t["weights_scaling_factor"]]
in_desc = (TensorDesc * 7)(*[TensorDesc.make(x.shape) for x in ins]); out_desc = TensorDesc.make(o.shape)
h = ctypes.c_void_p(lib.mixq_create(M, N, K))
ws = torch.empty(max(lib.mixq_workspace_size(h, M, N, K), 16), dtype=torch.uint8, device=dev)
for knob in (885, 884):
    lib.mixq_debug_reset(); lib.mixq_debug_set_gemm_variant(knob)
    for i in range(64):
        v = [x.data_ptr() for x in ins]; v[1] = alts[i % 8].data_ptr()
        assert lib.mixq_enqueue(h, in_desc, ctypes.byref(out_desc), (ctypes.c_void_p * 7)(*v), (ctypes.c_void_p * 1)(o.data_ptr()),
                                ctypes.c_void_p(ws.data_ptr()), st0) == 0
    torch.cuda.synchronize()
PY
pass_() { # name counters...
  local name="$1"; shift
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python "$OUT/work.py" ) > "$OUT/$name.log" 2>&1
  python - "$OUT/$name" <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "gemm_skinny_kernel" in r["Kernel_Name"]:
            form = r["Kernel_Name"].split("(")[0].split(",")[-1].strip(" >")
            agg[(form, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (form, c), v in sorted(agg.items()):
        print(f"  weight route {form} ({'64-byte pieces' if form == '0' else '256-byte runs'})  {c:28s} mean / dispatch {sum(v)/len(v):14.1f}  (n = {len(v)})")
PY
}
pass_ fetch FETCH_SIZE
pass_ ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass_ tcc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum
pass_ tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass_ sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD
tail -3 "$OUT"/*.log | grep -i "error\|invalid" | head -5

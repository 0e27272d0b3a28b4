"""round 6: what the quantiser inside the mid-M launch costs: one shape, 48 mixq_enqueue calls over rotating weight copies in one HIP graph (cold weights), two launches vs
one, with the one-launch form's fences / poll interval switched by measurement knobs (1470 + bits: 1 no release fence, 2 no acquire fence, 4 long poll sleep)."""
import ctypes, os, sys, torch
os.environ.setdefault("MIXQ_DEBUG_KNOBS", "1")
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import bench
from mixq_tensorrt_llm_amd import _lib
from mixq_tensorrt_llm_amd._lib import TensorDesc
lib = _lib.load()
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(0)
for (M, N, K) in ((256, 12288, 4096), (256, 11008, 4096), (256, 4096, 11008), (512, 4096, 11008)):
    ncopy = (320 << 20) // (N * K) + 1
    layers = [bench.synth_layer(N, K, dev, gen) for _ in range(ncopy)]
    A = bench.synth_activation(M, K, layers[0]["ind_i32"], dev, gen)
    o = torch.empty((M, N), dtype=torch.float16, device=dev)
    calls = []
    for t in layers:
        ins = [A, t["weight"], t["weights_scaling_factor"], t["fp_weight"], layers[0]["fp_ind"], t["qweight"], t["weights_scaling_factor"]]
        h = ctypes.c_void_p(lib.mixq_create(M, N, K)); lib.mixq_initialize(h)
        calls.append((h, (TensorDesc * 7)(*[TensorDesc.make(x.shape) for x in ins]), TensorDesc.make(o.shape),
                      (ctypes.c_void_p * 7)(*[x.data_ptr() for x in ins]), (ctypes.c_void_p * 1)(o.data_ptr()), ins))
    ws = torch.empty(max(lib.mixq_workspace_size(calls[0][0], M, N, K), 16), dtype=torch.uint8, device=dev)
    turn = [0]
    def run(st):
        h, idesc, odesc, ip, op, _ = calls[turn[0] % len(calls)]
        turn[0] += 1
        assert lib.mixq_enqueue(h, idesc, ctypes.byref(odesc), ip, op, ctypes.c_void_p(ws.data_ptr()), st) == 0
    res = {}
    for name, knobs in (("two launches", [1461]), ("one launch", [1460, 1470]), ("no release fence", [1460, 1471]), ("no acquire fence", [1460, 1472]),
                        ("neither fence", [1460, 1473]), ("long poll sleep", [1460, 1474]), ("neither + long sleep", [1460, 1477])):
        lib.mixq_debug_reset()
        for k in knobs: lib.mixq_debug_set_gemm_variant(k)
        turn[0] = -1
        res[name] = bench.graph_time_us(run, dev, calls=48, reps=10)
    lib.mixq_debug_reset()
    print(f"M={M} N={N} K={K}: " + " | ".join(f"{k} {v:.1f}" for k, v in res.items()), flush=True)
    for c in calls: lib.mixq_destroy(c[0])

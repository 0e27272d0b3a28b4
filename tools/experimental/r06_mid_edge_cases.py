"""round 6: the mid kernel forced (knob 1271 + x) on small and odd slice counts (K = 128 .. 1152), ragged M / N, O in {128, 0, 40}, 1 / 2 / 4 workgroups per tile, rotated / not, 96- / 128-wide tiles: bits of the plain tiles."""
import ctypes, os, sys, torch
os.environ["MIXQ_DEBUG_KNOBS"]="1"
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from mixq_tensorrt_llm_amd import _lib
from test_gpu_splitk import operands, p
lib=_lib.load()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
bad=0; n=0; nmid=0; seen=set()
for (M,N) in ((200,784),(129,96),(1024,144),(257,1040)):
  for K in (128,256,384,512,640,768,1152):
    for O in (128,0,40):
      for xs in (1,2,4):
        for rot in (1411,1412):
          for bn in (1430,1431):
            if xs>1 and K//128 < xs: continue
            qA, W, sA, sW, fpA, fpW = operands(M, N, K, O, seed=M+N+K+O)
            fa, fw = (p(fpA), p(fpW)) if O else (None, None)
            lib.mixq_debug_reset(); lib.mixq_debug_set_gemm_variant(1241)
            ref = torch.empty((M,N),dtype=torch.float16,device="cuda:0")
            assert lib.mixq_gemm_mixed(p(qA), p(W), p(sA), p(sW), fa, fw, p(ref), M, N, K, O, st)==0
            lib.mixq_debug_set_gemm_variant(rot); lib.mixq_debug_set_gemm_variant(bn); lib.mixq_debug_set_gemm_variant(1271+xs)
            nb = lib.mixq_gemm_scratch_size(M,N,K)
            scr = torch.zeros(max(nb,16),dtype=torch.uint8,device="cuda:0")
            for r in range(2):
                out = torch.full((M,N), float("nan"), dtype=torch.float16, device="cuda:0")
                rc = lib.mixq_gemm_mixed_scratch(p(qA), p(W), p(sA), p(sW), fa, fw, p(out), M, N, K, O, p(scr), nb, st)
                torch.cuda.synchronize()
                name = lib.mixq_debug_last_gemm_kernel()
                n+=1
                nmid += b"mid_kernel" in name
                if b"mid_kernel" in name: seen.add((M, N, K, xs))
                if rc!=0 or not torch.equal(out,ref):
                    bad+=1; print("BAD", M,N,K,O,xs,rot,bn,rc,name[:40], flush=True)
print("mid kernel on:", sorted(seen))
print("cases", n, "of which the mid kernel", nmid, "bad", bad)

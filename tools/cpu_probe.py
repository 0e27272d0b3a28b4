import os, time, sys, subprocess
sys.path.insert(0, '.')
print("nproc", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip())
    except Exception as e: print(f, "n/a")
print(subprocess.run("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Core|Socket|Flags' | cut -c1-300", shell=True, capture_output=True, text=True).stdout)
code = '''
import time, numpy as np, sys
sys.path.insert(0,'.')
from oracle import oracle as o
rng=np.random.default_rng(0)
M,N,K=2048,12288,4096
a=rng.integers(-127,128,(M,K),dtype=np.int8); w=rng.integers(-127,128,(N,K),dtype=np.int8)
o.gemm_s8s8s32(a[:64],w)
t=time.time(); o.gemm_s8s8s32(a,w); dt=time.time()-t
print("threads", o.num_threads(), "gemm %.3f s  %.1f GOP/s" % (dt, 2*M*N*K/dt/1e9))
'''
for n in (8, 16, 32, 64, 128):
    env = dict(os.environ, OMP_NUM_THREADS=str(n), OMP_PROC_BIND="false")
    print(subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip())

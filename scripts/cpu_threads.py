import sys, time, os
sys.path.insert(0, '.')
import oracle
from simpleslam_amd import synth
w, m = synth.make_map(1_000_000, seed=20261003+2)
scan, T = synth.make_scan(w, 0, seed=20261003+2)
T0 = synth.perturb(T, 20261003+2)
print('affinity', len(os.sched_getaffinity(0)), 'cpu_count', os.cpu_count())
try: print('cpu.max', open('/sys/fs/cgroup/cpu.max').read().strip())
except Exception as e: print('no cpu.max', e)
os.system("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Core|Socket' | head -6")
for th in (1, 4, 8, 16, 32, 64, 128):
    prm = oracle.loam_params(iters=10, early_exit=0, threads=th)
    t = time.time(); oracle.loam_scan2map(scan, m, T0, prm); dt = time.time() - t
    print(f'threads {th:4d}: {dt:.3f} s/scan  {1/dt:.2f} scans/s')

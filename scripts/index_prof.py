"""Development aid: the LOAM target index of an n-point map rebuilt per call (one Gauss-Newton iteration), for rocprofv3 runs of the build
kernels alone:  python scripts/index_prof.py <map points> [calls]"""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from simpleslam_amd import LoamRegister, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 14
S = 20261003 + 2
w, m = synth.make_map(n, seed=S)
scan, T = synth.make_scan(w, 0, seed=S)
T0 = synth.perturb(T, S)
ds, dm = torch.from_numpy(scan).cuda(), torch.from_numpy(m).cuda()
reg = LoamRegister(loam_iters=1, loam_early_exit=0)
t = []
for i in range(calls):
    pose = T0.copy(); reg.scan2Map(ds, dm, pose); t.append(reg.stats()["index_ms"])
print(f"map {n}: index build median {np.median(t[4:]) * 1e3:.1f} us  ({32e-6 * n / np.median(t[4:]) / 1e3:.2f} TB/s algorithmic)")

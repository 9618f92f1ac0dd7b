import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from simpleslam_amd import LoamRegister, synth
S = 20261003 + 2
w, m = synth.make_map(1_000_000, seed=S)
scans = [synth.make_scan(w, j, seed=S) for j in range(8)]
inits = [synth.perturb(T, S + j) for j, (s, T) in enumerate(scans)]
reg = LoamRegister(loam_iters=10, loam_early_exit=0)
dm = torch.from_numpy(m).cuda(); ds = [torch.from_numpy(s).cuda() for s, _ in scans]
reg.setTarget(dm)
for i in range(20): p = inits[i % 8].copy(); reg.align(ds[i % 8], p)
t0 = time.perf_counter()
for i in range(200): p = inits[i % 8].copy(); reg.align(ds[i % 8], p)
print('prepared target: %.0f scans/s' % (200 / (time.perf_counter() - t0)))
for i in range(5): p = inits[i % 8].copy(); reg.scan2Map(scans[i % 8][0], m, p)
t0 = time.perf_counter()
for i in range(50): p = inits[i % 8].copy(); reg.scan2Map(scans[i % 8][0], m, p)
print('host buffers (16-byte points, pageable numpy): %.0f scans/s' % (50 / (time.perf_counter() - t0)))

import sys, numpy as np
sys.path.insert(0, '.')
import torch
from simpleslam_amd import LoamRegister, synth
w, m = synth.make_map(1_000_000, seed=20261003+2)
scan, T = synth.make_scan(w, 0, seed=20261003+2)
T0 = synth.perturb(T, 20261003+2)
reg = LoamRegister(loam_iters=10, loam_early_exit=0, record_trace=1)
p = T0.copy(); reg.scan2Map(torch.from_numpy(scan).cuda(), torch.from_numpy(m).cuda(), p)
tr = reg.trace()
print('hits    ', tr['cache_hits']); print('searches', tr['searches']); print('accepted', tr['n'])
print('|x| trans', np.linalg.norm(tr['x'][:, :3], axis=1)); print('|x| rot', np.linalg.norm(tr['x'][:, 3:], axis=1))

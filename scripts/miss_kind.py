import sys, numpy as np
sys.path.insert(0, '.')
import torch
from simpleslam_amd import LoamRegister, synth, pcr
S = 20261003 + 2
w, m = synth.make_map(1_000_000, seed=S)
scan, T = synth.make_scan(w, 0, seed=S)
T0 = synth.perturb(T, S)
reg = LoamRegister(loam_iters=10, loam_early_exit=0, record_trace=1)
dm, ds = torch.from_numpy(m).cuda(), torch.from_numpy(scan).cuda()
pose = T0.copy(); reg.scan2Map(ds, dm, pose)
tr = reg.trace(); print('searches', tr['searches'], 'hits', tr['cache_hits'])
reg.setTarget(dm)
out = reg.linearize(ds, pose, per_point=True)
st = out['status']
print('status histogram at the final pose', {int(k): int((st == k).sum()) for k in np.unique(st)})

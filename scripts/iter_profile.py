"""Per-iteration view of one LOAM scan2map (65 536 x 1 M, 10 iterations): cache hits / searches per linearisation (trace) and the
in-kernel timeline of every launch (pcr_params.record_timeline = 1)."""
import sys, numpy as np
sys.path.insert(0, '.')
import torch
from simpleslam_amd import LoamRegister, synth, pcr
S = 20261003 + 2
w, m = synth.make_map(1_000_000, seed=S)
scan, T = synth.make_scan(w, 0, seed=S)
T0 = synth.perturb(T, S)
dm, ds = torch.from_numpy(m).cuda(), torch.from_numpy(scan).cuda()
p = pcr.default_params(loam_iters=10, loam_early_exit=0, record_trace=1)
reg = LoamRegister(params=p)
pose = T0.copy(); reg.scan2Map(ds, dm, pose)
tr = reg.trace()
print('accepted rows  ', tr['n'])
print('cache hits     ', tr['cache_hits'])
print('searches       ', tr['searches'])
p = pcr.default_params(loam_iters=10, loam_early_exit=0)
p.record_timeline = 1
reg = LoamRegister(params=p)
for i in range(3):
    pose = T0.copy(); reg.scan2Map(ds, dm, pose)
tl = reg.timeline()
names = ['entry', 'prologue', 'posted', 'searched', 'plane', 'accum', 'stored']
for k in range(tl.shape[0]):
    t = tl[k]; ok = t[:, 6] > 0
    d = np.diff(t[ok][:, :7], axis=1).mean(axis=0)
    print(k, ' '.join(f'{names[i + 1]}:{d[i]:6.2f}' for i in range(6)), f'| mean stored {t[ok, 6].mean():6.2f} max stored {t[ok, 6].max():6.2f}')
t = tl[0]; ok = t[:, 10] > 0
if ok.any():
    for a, b, n in ((2, 8, 'ranges'), (8, 9, 'first group'), (9, 10, 'stream'), (10, 3, 'decode + exchange')):
        print(f'  launch 0 dense search: {n:18s} {np.mean(t[ok, b] - t[ok, a]):6.2f} us')
reg.set_profile(2)
pose = T0.copy(); reg.scan2Map(ds, dm, pose)
print(reg.stats())

"""Where the prologue of loam_iterate_kernel spends its time (pcr_params.record_timeline): entry -> partial sums folded -> normal equations
solved -> pose updated, per launch, mean over the blocks (us)."""
import sys, numpy as np
sys.path.insert(0, '.')
import torch
from simpleslam_amd import LoamRegister, synth, pcr
S = 20261003 + 2
w, m = synth.make_map(1_000_000, seed=S)
scan, T = synth.make_scan(w, 0, seed=S)
T0 = synth.perturb(T, S)
dm, ds = torch.from_numpy(m).cuda(), torch.from_numpy(scan).cuda()
p = pcr.default_params(loam_iters=10, loam_early_exit=0)
p.record_timeline = 1
reg = LoamRegister(params=p)
for i in range(3):
    pose = T0.copy(); reg.scan2Map(ds, dm, pose)
tl = reg.timeline()
for k in range(tl.shape[0]):
    t = tl[k]; ok = t[:, 6] > 0
    t = t[ok]
    e, f, s, d = t[:, 0], t[:, 7], t[:, 11], t[:, 1]
    print(k, f'entry (after the first block of the launch) {e.mean():5.2f}  -> folded {np.mean(f - e):5.2f}  -> solved {np.mean(s - f):5.2f}  -> pose {np.mean(d - s):5.2f}   | prologue {np.mean(d - e):5.2f}')

"""searched-stage duration of the first (full-search) launch under the ablation masks (environment variable PCR_ABLATE of a DEVELOPMENT build).
Needs a library built with the switches compiled in:  make -C simpleslam_amd/csrc clean all DEV=1 EXTRA=-DPCR_ABLATION"""
import os, sys, numpy as np
sys.path.insert(0, '.')
import torch
from simpleslam_amd import LoamRegister, synth, pcr
S = 20261003 + 2
w, m = synth.make_map(1_000_000, seed=S)
scan, T = synth.make_scan(w, 0, seed=S)
T0 = synth.perturb(T, S)
dm, ds = torch.from_numpy(m).cuda(), torch.from_numpy(scan).cuda()
for mask, what in ((0, 'full'), (32, 'distance only (no insertion)'), (16, 'loads only (no distance, no insertion)'), (1, 'no search at all')):
    p = pcr.default_params(loam_iters=3, loam_early_exit=0)
    p.record_timeline = 1; os.environ["PCR_ABLATE"] = str(mask)
    reg = LoamRegister(params=p)
    for i in range(3):
        pose = T0.copy()
        try: reg.scan2Map(ds, dm, pose)
        except Exception as e: pass
    tl = reg.timeline()
    t = tl[0]; d = np.diff(t[:, :7], axis=1)
    print(f'mask {mask:2d} {what:42s} searched stage mean {d[:, 2].mean():6.2f} us  max {d[:, 2].max():6.2f}   plane {d[:, 3].mean():5.2f}  launch max_stored {t[:, 6].max():6.2f}')

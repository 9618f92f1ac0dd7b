"""Index build time of a 1 M-point map under the ablation switches of grid_bin_kernel (PCR_BIN_ABLATE, development aid)."""
import os, sys, numpy as np
sys.path.insert(0, '.')
import torch
from simpleslam_amd import LoamRegister, synth
S = 20261003 + 2
w, m = synth.make_map(1_000_000, seed=S)
scan, T = synth.make_scan(w, 0, seed=S)
T0 = synth.perturb(T, S)
dm, ds = torch.from_numpy(m).cuda(), torch.from_numpy(scan).cuda()
reg = LoamRegister(loam_iters=1, loam_early_exit=0)
for mode in (0, 1, 2, 3, 4, 8, 15, 0):
    os.environ["PCR_BIN_ABLATE"] = str(mode)
    t = []
    for i in range(12):
        pose = T0.copy()
        try:
            reg.scan2Map(ds, dm, pose)
        except Exception as e:
            pass
        t.append(reg.stats()["index_ms"])
    print(f"mode {mode:2d}: index build {np.median(t[2:]) * 1e3:7.1f} us")

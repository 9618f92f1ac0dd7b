import sys, numpy as np
sys.path.insert(0, '.')
import torch
from simpleslam_amd import LoamRegister, synth
N = 1_000_000
w, m = synth.make_map(N, seed=20261003+2)
scan, T = synth.make_scan(w, 0, seed=20261003+2)
T0 = synth.perturb(T, 20261003+2)
dm, ds = torch.from_numpy(m).cuda(), torch.from_numpy(scan).cuda()
reg = LoamRegister(loam_iters=10, loam_early_exit=0)
reg.setTarget(dm)
reg.set_profile(0)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    pose = T0.copy(); reg.align(ds, pose)
print('done', synth.pose_error(pose, T))

import sys, numpy as np
sys.path.insert(0, '.')
import torch
from simpleslam_amd import LoamRegister, synth
from simpleslam_amd.pcr import default_params
w, m = synth.make_map(1_000_000, seed=20261003+2)
scan, T = synth.make_scan(w, 0, seed=20261003+2)
dm, ds = torch.from_numpy(m).cuda(), torch.from_numpy(scan).cuda()
def run(mask, label, pose):
    p = default_params(loam_iters=10, loam_early_exit=0)
    p.reserved[0] = mask
    reg = LoamRegister(params=p); reg.setTarget(dm); reg.set_profile(2)
    ks = []
    for i in range(8):
        q = pose.copy(); reg.align(ds, q); st = reg.stats(); ks.append(st['kernel_ms'] / max(1, st['kernel_launches']))
    print(f"{label:50s} iterate kernel avg {1e3*np.median(ks):8.2f} us")
# at the TRUE pose every iteration after the first is a pure cache-hit iteration
run(0, 'truth pose: 1 full + 9 hit iterations', T)
run(4, 'truth pose, prologue solve skipped', T)

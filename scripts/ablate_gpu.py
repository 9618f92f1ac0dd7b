import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from simpleslam_amd import LoamRegister, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
w, m = synth.make_map(N, seed=20261003+2)
scan, T = synth.make_scan(w, 0, seed=20261003+2)
T0 = synth.perturb(T, 20261003+2)
dm, ds = torch.from_numpy(m).cuda(), torch.from_numpy(scan).cuda()
def run(mask, label, nosort=0):
    from simpleslam_amd.pcr import default_params
    p = default_params(loam_iters=10, loam_early_exit=0)
    p.reserved[0] = mask
    p.reserved[2] = 1   # temporal cache off: every iteration is a full search
    reg = LoamRegister(params=p)
    reg.setTarget(dm)
    reg.set_profile(2)
    for i in range(3):
        pose = T0.copy(); reg.align(ds, pose)
    ks = []
    for i in range(10):
        pose = T0.copy(); reg.align(ds, pose); st = reg.stats(); ks.append(st['kernel_ms'] / max(1, st['kernel_launches']))
    print(f"{label:40s} mask={mask} iterate kernel avg {1e3*np.median(ks):8.2f} us  solve_ms {st['solve_ms']:.3f}")
run(0, 'full')
run(2 | 4, 'kNN only, no prologue')
run(2 | 4 | 32, 'kNN: loads + distances, no insert')
run(2 | 4 | 16, 'kNN: loads only')
run(1 | 4, 'no candidate loop')

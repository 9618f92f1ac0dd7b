"""LOAM scan2Map throughput over scan and map sizes (10 iterations, early exit off, index rebuilt per call, inputs in HBM)."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from simpleslam_amd import LoamRegister, synth
S = 20261003 + 2
print('map points | scan points | ms/scan | scans/s')
for n_map in (100_000, 1_000_000, 10_000_000):
    w, m = synth.make_map(n_map, seed=S)
    dm = torch.from_numpy(m).cuda()
    for beams, az in ((16, 1024), (64, 1024), (128, 1024), (128, 2048)):
        s, T = synth.make_scan(w, 0, seed=S, beams=beams, azimuths=az)
        init = synth.perturb(T, S)
        ds = torch.from_numpy(s).cuda()
        reg = LoamRegister(loam_iters=10, loam_early_exit=0)
        for i in range(5):
            p = init.copy(); reg.scan2Map(ds, dm, p)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(100):
            p = init.copy(); reg.scan2Map(ds, dm, p)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
        print(f'{n_map:>10d} | {s.shape[0]:>7d} | {dt*1e3:7.3f} | {1/dt:7.0f}   ({synth.pose_error(p, T)[0]*1e3:.1f} mm from truth)', flush=True)
        del reg

import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from simpleslam_amd import LoamRegister, synth
import oracle
# f64 sqrt / div exactness of the device vs host
rng = np.random.default_rng(0)
a = rng.uniform(0.1, 100, 1_000_00); b = rng.uniform(0.1, 100, 1_000_00)
ta, tb = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
print('sqrt mismatches', int((torch.sqrt(ta).cpu().numpy() != np.sqrt(a)).sum()), 'div mismatches', int(((ta / tb).cpu().numpy() != a / b).sum()))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
w, m = synth.make_map(N, seed=20261003+2)
scan, T = synth.make_scan(w, 0, seed=20261003+2)
T0 = synth.perturb(T, 20261003+2)
reg = LoamRegister(loam_iters=10, loam_early_exit=0)
dm, ds = torch.from_numpy(m).cuda(), torch.from_numpy(scan).cuda()
for i in range(3):
    p = T0.copy(); c = reg.scan2Map(ds, dm, p); print(c, synth.pose_error(p, T), reg.stats())
t=time.time()
for i in range(20):
    p = T0.copy(); reg.scan2Map(ds, dm, p)
dt=(time.time()-t)/20; print('ms/scan', dt*1e3, 'scans/s', 1/dt)
reg.set_profile(2); p=T0.copy(); reg.scan2Map(ds, dm, p); print(reg.stats())
reg.set_profile(0)
t=time.time()
for i in range(20):
    p = T0.copy(); reg.scan2Map(ds, dm, p)
dt=(time.time()-t)/20; print('profile0 ms/scan', dt*1e3, 'scans/s', 1/dt)
reg.setTarget(dm)
t=time.time()
for i in range(20):
    p = T0.copy(); reg.align(ds, p)
dt=(time.time()-t)/20; print('static-map ms/scan', dt*1e3, 'scans/s', 1/dt)
t=time.time(); po, co, info = oracle.loam_scan2map(scan, m, T0, oracle.loam_params(iters=10, early_exit=0, threads=16)); print('oracle s', time.time()-t)
print('gpu vs oracle', synth.pose_error(p, po))

import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from simpleslam_amd import VgicpRegister, synth
import oracle
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
res = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
w, m = synth.make_map(N, seed=20261003+3)
scan, T = synth.make_scan(w, 0, seed=20261003+3)
T0 = synth.perturb(T, 20261003+3)
dm, ds = torch.from_numpy(m).cuda(), torch.from_numpy(scan).cuda()
reg = VgicpRegister(vgicp_resolution=res)
for i in range(3):
    p = T0.copy(); t = time.time(); c = reg.scan2Map(ds, dm, p); dt = time.time() - t
    print('scan2map (target rebuilt)', c, synth.pose_error(p, T), f'{dt*1e3:.2f} ms', reg.stats())
t = time.time(); reg.setTarget(dm); torch.cuda.synchronize(); print('setTarget ms', (time.time() - t) * 1e3)
ts = []
for i in range(10):
    p = T0.copy(); t = time.time(); c = reg.align(ds, p); ts.append(time.time() - t)
print('align (static target) ms', np.median(ts) * 1e3, 'scans/s', 1 / np.median(ts), reg.stats()['iterations'], reg.getFitnessScore())
t = time.time(); po, co, info = oracle.vgicp_scan2map(scan, m, T0, oracle.vgicp_params(threads=16, resolution=res)); print('oracle s', time.time() - t, co, info)
print('gpu vs oracle', synth.pose_error(p, po))

import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from simpleslam_amd import VgicpRegister, NdtRegister, synth
which = sys.argv[1]
if which == 'vgicp':
    w, m = synth.make_map(1_000_000, seed=20261003+3)
    scan, T = synth.make_scan(w, 0, seed=20261003+3)
    T0 = synth.perturb(T, 20261003+3)
    reg = VgicpRegister(vgicp_resolution=0.5)
else:
    w, m = synth.make_map(5_000_000, seed=20261003+5, spacing=0.22)
    scan, T = synth.make_scan(w, 0, seed=20261003+5, beams=128, azimuths=1024)
    T0 = synth.perturb(T, 20261003+5, trans=0.1, rot_deg=0.5)
    reg = NdtRegister()
dm, ds = torch.from_numpy(m).cuda(), torch.from_numpy(scan).cuda()
for i in range(2):
    p = T0.copy(); t = time.time(); c = reg.scan2Map(ds, dm, p); dt = time.time() - t
    print(which, 'scan2map', c, synth.pose_error(p, T), f'{dt*1e3:.2f} ms', reg.stats())
reg.setTarget(dm)
ts = []
for i in range(5):
    p = T0.copy(); t = time.time(); c = reg.align(ds, p); ts.append(time.time() - t)
print(which, 'align (prepared target) ms', np.median(ts) * 1e3, reg.stats()['iterations'], reg.stats()['kernel_launches'])

import sys, numpy as np
sys.path.insert(0, '.')
import oracle
from simpleslam_amd import synth
S = 20261003 + 5
w, m = synth.make_map(5_000_000, seed=S, spacing=0.22)
gpu = None
try:
    import torch
    if torch.cuda.is_available():
        from simpleslam_amd import NdtRegister
        gpu = NdtRegister(); dm = torch.from_numpy(m).cuda()
except Exception as e:
    print('no gpu', e)
for j in range(8):
    scan, T = synth.make_scan(w, j, seed=S, beams=128, azimuths=1024)
    T0 = synth.perturb(T, S + j, trans=0.1, rot_deg=0.5)
    po, conv, info = oracle.ndt_scan2map(scan, m, T0, oracle.ndt_params())
    line = f'scan {j}: oracle finite={np.isfinite(po).all()} conv={conv} info={info}'
    if gpu is not None:
        pg = T0.copy(); c = gpu.scan2Map(torch.from_numpy(scan).cuda(), dm, pg)
        line += f' | gpu finite={np.isfinite(pg).all()} conv={c} iters={gpu.stats()["iterations"]}'
        if np.isfinite(pg).all() and np.isfinite(po).all(): line += f' err={synth.pose_error(pg, po)}'
    print(line, flush=True)

"""Development aid: host time of pcr_map_update_begin (queueing an assembly) and of pcr_voxel_filter for a scan, apart from the device's work."""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from simpleslam_amd import make_register, sequence, SubMap
scans, truth, cmds = sequence.make_drive(12, 20261010)
d = [torch.from_numpy(s).cuda() for s in scans]
sm = SubMap()
for k in range(8):
    sm.addKeyFrame(d[k], truth[k])
reg = make_register("loam")
tb, tw, tv = [], [], []
for rep in range(30):
    pos = truth[7][:3, 3] + 0.01 * rep
    torch.cuda.synchronize()
    t0 = time.perf_counter(); sm.updateMapBegin(pos, 8.0, 0.5); t1 = time.perf_counter(); n = sm.wait(); t2 = time.perf_counter()
    tb.append(t1 - t0); tw.append(t2 - t1)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); reg.voxelDownSample(d[rep % 12], 0.5); tv.append(time.perf_counter() - t0)
print(f"assembly of {n} voxels: queueing {1e6 * np.median(tb[5:]):.1f} us, collecting {1e6 * np.median(tw[5:]):.1f} us; a scan's voxel filter {1e6 * np.median(tv[5:]):.1f} us")

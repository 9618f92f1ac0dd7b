"""Development aid: where a step of the caller's loop (simpleslam_amd.sequence) spends its time, per front-end call: python scripts/seq_breakdown.py [method]"""
import os, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from simpleslam_amd import make_register, sequence
method = sys.argv[1] if len(sys.argv) > 1 else "loam"
scans, truth, cmds = sequence.make_drive(64, 20261010)
d_scans = [torch.from_numpy(s).cuda() for s in scans]


class Timed(sequence.GpuFront):
    def __init__(self, reg):
        super().__init__(reg)
        self.t = {"voxel": 0.0, "scan2map": 0.0, "add_keyframe": 0.0, "update_map": 0.0, "submap_points": 0.0}
        self.n = {k: 0 for k in self.t}

    def _time(self, name, fn, *a):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(*a); torch.cuda.synchronize(); self.t[name] += time.perf_counter() - t0; self.n[name] += 1
        return r

    def voxel(self, scan, grid): return self._time("voxel", super().voxel, scan, grid)
    def scan2map(self, ds, pose): return self._time("scan2map", super().scan2map, ds, pose)
    def add_keyframe(self, scan, pose): return self._time("add_keyframe", super().add_keyframe, scan, pose)
    def update_map(self, p, r, g): return self._time("update_map", super().update_map, p, r, g)
    def submap_points(self): return self._time("submap_points", super().submap_points)      # (collects a queued assembly)


for rep in range(2):
    reg = make_register(method); reg.set_profile(0)
    f = Timed(reg)
    r = sequence.drive(f, d_scans, cmds, truth[0])
print(f"{method}: {64 / r['seconds']:.0f} scans/s with the timing syncs, {1e3 * r['seconds'] / 64:.3f} ms per scan")
for k in f.t:
    print(f"   {k:14s} {f.n[k]:3d} calls, {1e3 * f.t[k] / max(1, f.n[k]):.3f} ms each, {1e3 * f.t[k] / 64:.3f} ms per scan")
print(f"   python loop + rest: {1e3 * (r['seconds'] - sum(f.t.values())) / 64:.3f} ms per scan")

"""Timing of the rows either side of the path next to their oracles: sub-map assembly and ScanContext."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch, oracle
from simpleslam_amd import ScanContext, SubMap, synth

S = 20261003 + 2
w, m = synth.make_map(200_000, seed=S)
# ---- sub-map: 60 key frames of ~16 k points along the trajectory, 8 m radius, 0.4 m grid (MapManager defaults)
sm = SubMap()
clouds, poses = [], []
for k in range(60):
    scan, T = synth.make_scan(w, k % 8, seed=S + k, beams=16, azimuths=1024)
    T = T.copy(); T[0, 3] += 0.5 * k
    clouds.append(scan); poses.append(T); sm.addKeyFrame(scan, T)
centre = poses[30][:3, 3]
for _ in range(3): n = sm.updateMap(centre, 8.0, 0.4)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): n = sm.updateMap(centre, 8.0, 0.4)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
c0 = time.perf_counter(); ref, sel = oracle.submap_assemble(clouds, poses, centre, 8.0, 0.4); ct = time.perf_counter() - c0
n_in = sum(clouds[i].shape[0] for i in sel)
print(f'sub-map: {len(sel)} of 60 key frames, {n_in} points -> {n} voxels: device {dt*1e3:.3f} ms, oracle (1 core) {ct*1e3:.1f} ms, same {n == ref.shape[0] and list(sel) == list(sm.submapIdx())}')

# ---- ScanContext: 600 contexts of a 0.5 m down-sampled 64-beam scan
sc, orc = ScanContext(), oracle.ScanContextOracle()
scans = []
for k in range(16):
    s, _ = synth.make_scan(w, k % 8, seed=S + 100 + k)
    scans.append(s[::4].copy())                      # ~16 k points, what a 0.5 m voxel filter leaves of a scan
ta = tq = 0.0
for i in range(600):
    s = scans[i % 16].copy(); s[:, 0] += 0.01 * i
    t0 = time.perf_counter(); sc.addContext(s); ta += time.perf_counter() - t0
    t0 = time.perf_counter(); r = sc.query(i); tq += time.perf_counter() - t0
ca = cq = 0.0
for i in range(600):
    s = scans[i % 16].copy(); s[:, 0] += 0.01 * i
    t0 = time.perf_counter(); orc.add(s); ca += time.perf_counter() - t0
    t0 = time.perf_counter(); ro = orc.query(i); cq += (time.perf_counter() - t0) if i >= 540 else 0.0      # (the snapshot is stateful: query every id)
print(f'ScanContext: addContext {scans[0].shape[0]} points (host buffer): device {ta/600*1e3:.3f} ms, oracle {ca/600*1e3:.3f} ms; '
      f'query against up to 560 contexts: library {tq/600*1e3:.3f} ms, oracle (numpy + C) {cq/60*1e3:.3f} ms; last result {r} vs {ro}')

"""pcr_voxel_filter timing (device-resident in/out) next to the oracle, on the benchmark clouds."""
import sys, time, numpy as np
sys.path.insert(0, '.')
import torch, oracle
from simpleslam_amd import LoamRegister, synth
S = 20261003 + 2
w, m = synth.make_map(1_000_000, seed=S)
scan, T = synth.make_scan(w, 0, seed=S)
reg = LoamRegister()
for name, pts, leaf in (('scan 65536', scan, 0.4), ('map 1M', m, 0.4), ('map 1M', m, 1.0)):
    d = torch.from_numpy(pts).cuda()
    for _ in range(3): out = reg.voxelDownSample(d, leaf)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): out = reg.voxelDownSample(d, leaf)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    c0 = time.perf_counter(); ref, _ = oracle.voxel_filter(pts, leaf); ct = time.perf_counter() - c0
    print(f'{name} leaf {leaf}: {pts.shape[0]} -> {out.shape[0]} voxels, device {dt*1e3:.3f} ms ({pts.shape[0]/dt/1e9:.2f} Gpt/s), oracle (1 core) {ct*1e3:.1f} ms, same count {ref.shape[0] == out.shape[0]}')

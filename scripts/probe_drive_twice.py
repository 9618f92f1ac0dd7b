"""Development aid: the caller's loop twice through the same front (key frames forgotten in between, device memory kept): what the first drive pays for allocations."""
import os, sys, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from simpleslam_amd import make_register, sequence
method = sys.argv[1] if len(sys.argv) > 1 else "loam"
scans, truth, cmds = sequence.make_drive(64, 20261010)
d = [torch.from_numpy(s).cuda() for s in scans]
reg = make_register(method)
front = sequence.GpuFront(reg)
for rep in range(4):
    r = sequence.drive(front, d, cmds, truth[0])
    print(f"{method} drive {rep}: {64 / r['seconds']:.0f} scans/s; ms per scan by step: " + ", ".join(f"{k} {1e3 * v / 64:.3f}" for k, v in r["step_seconds"].items()))
    front.map.clear(); front._n = 0

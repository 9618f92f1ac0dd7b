"""Development aid: the covariance search of the bench's 65 536-point scan ALONE on the device (pcr_vgicp_covariances in a loop), for
rocprofv3 runs that should not see the target's kernels beside it."""
import sys, re, numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from simpleslam_amd import VgicpRegister, synth
SEED = int(re.search(r"^SEED\s*=\s*(\d+)", open(os.path.join(ROOT, "bench.py")).read(), re.M).group(1))
n_map = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
world, m = synth.make_map(n_map, seed=SEED + 3)
reg = VgicpRegister(vgicp_resolution=0.5)
for k in range(int(sys.argv[2]) if len(sys.argv) > 2 else 12):
    s, T = synth.make_scan(world, k % 4, seed=SEED + 3)
    d = torch.from_numpy(s).cuda()
    c = reg.covariances(d)
nb, queued = reg.neighbours(len(s))
print("queued", queued, "of", len(s))

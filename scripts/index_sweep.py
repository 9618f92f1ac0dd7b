"""Index build time of a 1 M-point map for tile sizes 2^8 .. 2^12 cells (PCR_TILE_SHIFT, development aid), two input orders."""
import os, sys, numpy as np
sys.path.insert(0, '.')
import torch
from simpleslam_amd import LoamRegister, synth
S = 20261003 + 2
w, m = synth.make_map(1_000_000, seed=S)
scan, T = synth.make_scan(w, 0, seed=S)
T0 = synth.perturb(T, S)
ds = torch.from_numpy(scan).cuda()
# the order pcl::VoxelGrid leaves a cloud in (ascending voxel index, x fastest): what MapManager hands to scan2Map
c = np.floor(m[:, :3] / 0.5).astype(np.int64); c -= c.min(0)
order = np.lexsort((c[:, 0], c[:, 1], c[:, 2]))
for name, mm in (("generator order", m), ("voxel-grid order", np.ascontiguousarray(m[order])), ("shuffled", np.ascontiguousarray(m[np.random.default_rng(1).permutation(len(m))]))):
    dm = torch.from_numpy(mm).cuda()
    for sh, per in ((None, 16), (10, 16), (11, 8), (10, 8), (11, 4), (10, 4), ("atomic", 16)):
        os.environ.pop("PCR_TILE_SHIFT", None); os.environ.pop("PCR_INDEX_ATOMIC", None)
        os.environ["PCR_BIN_PER"] = str(per)
        if sh == "atomic": os.environ["PCR_INDEX_ATOMIC"] = "1"
        elif sh is not None: os.environ["PCR_TILE_SHIFT"] = str(sh)
        reg = LoamRegister(loam_iters=1, loam_early_exit=0)
        t = []
        for i in range(14):
            pose = T0.copy(); reg.scan2Map(ds, dm, pose); t.append(reg.stats()["index_ms"])
        print(f"{name:18s} tile shift {str(sh):6s} points/thread {per:2d}: index build {np.median(t[4:]) * 1e3:7.1f} us")
        del reg

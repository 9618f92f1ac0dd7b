"""Development aid: which class of queries (finishing stage of the ring search) do the covariance mismatches against the oracle fall in?"""
import sys, numpy as np
sys.path.insert(0, ".")
import oracle
from simpleslam_amd import VgicpRegister, synth
from scipy.spatial import cKDTree

world, m = synth.make_map(60_000, seed=21)
scan, T = synth.make_scan(world, 0, seed=21, beams=32, azimuths=512)
reg = VgicpRegister()
g = reg.covariances(scan)
o = oracle.vgicp_covariances(scan, 20, 8)
bad = np.abs(g - o).max(axis=(1, 2)) > 1e-9
print("bad", bad.sum(), "of", len(bad), "zero rows", (np.abs(g).max(axis=(1, 2)) == 0).sum())
p = scan[:, :3].astype(np.float64)
d, _ = cKDTree(p).query(p, k=20)
d20 = d[:, 19]
def face(cell):
    f = np.floor(p / cell); return np.minimum(p - f * cell, (f + 1) * cell - p).min(1)
m0, m1 = face(0.5), face(3.0)
stages = [(0, 1), (0, 2), (1, 1), (1, 2), (1, 3), (1, 4), (1, 6), (1, 10), (1, 40)]
rem = np.ones(len(p), bool)
for lv, r in stages:
    b = (m0 if lv == 0 else m1) + r * (0.5 if lv == 0 else 3.0)
    fin = rem & (d20 < b); rem &= ~fin
    print((lv, r), "finish", fin.sum(), "bad among them", (bad & fin).sum())
print("left", rem.sum(), (bad & rem).sum())
i = np.where(bad)[0][:5]
for k in i: print(k, "\n", g[k], "\n", o[k])

# neighbour lists against brute force in FLANN's float arithmetic
nb, queued = reg.neighbours(len(scan))
print("queued", queued)
pf = scan[:, :3].astype(np.float32)
def brute(i):
    dx = pf[i, 0] - pf[:, 0]; dy = pf[i, 1] - pf[:, 1]; dz = pf[i, 2] - pf[:, 2]
    d = (dx * dx).astype(np.float32); d = (d + (dy * dy).astype(np.float32)).astype(np.float32); d = (d + (dz * dz).astype(np.float32)).astype(np.float32)
    key = (d.view(np.uint32).astype(np.uint64) << np.uint64(32)) | np.arange(len(pf), dtype=np.uint64)
    o = np.argsort(key)[:20]
    return o, d[o]
for k in np.where(bad)[0][:6]:
    o, dd = brute(k)
    print("query", k, "gpu", nb[k].tolist())
    print("      ref", o.tolist())
    miss = [int(x) for x in o if x not in set(nb[k].tolist())]; extra = [int(x) for x in nb[k] if x not in set(o.tolist())]
    print("      missing", miss, "extra", extra, "d20 ref", dd[19], "d of missing", [float(brute(k)[1][list(o).index(x)]) for x in miss])

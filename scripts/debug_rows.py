import sys, numpy as np
sys.path.insert(0, '.')
from simpleslam_amd import LoamRegister, synth
import oracle
world, m = synth.make_map(30_000, seed=5)
scan, T = synth.make_scan(world, 0, seed=5, beams=16, azimuths=512)
T0 = synth.perturb(T, 5, trans=0.2, rot_deg=1.0)
reg = LoamRegister(); reg.setTarget(m)
tree = oracle.KdTree(m)
g = reg.linearize(scan, T0, per_point=True)
o = oracle.loam_linearize(tree, scan, T0, oracle.loam_params(), per_point=True)
acc = o['status'] == 0
rel = np.abs(g['rows'] - o['rows']) / (np.abs(o['rows']) + 1e-300)
bad = np.where(acc & (rel.max(1) > 1e-8))[0]
print('bad', bad.size, 'of', acc.sum())
np.set_printoptions(precision=17, linewidth=200)
for i in bad[:4]:
    print('--- point', i, scan[i])
    print('gpu', g['rows'][i]); print('cpu', o['rows'][i]); print('nn', g['nn'][i], o['nn'][i])
    A = m[o['nn'][i], :3].astype(np.float64)
    x = np.linalg.lstsq(A, -np.ones(5), rcond=None)[0]
    xo, ok = oracle.plane_fit5(A)
    print('lstsq x', x, 'oracle x', xo, 'sv', np.linalg.svd(A, compute_uv=False))
    q = (T0[:3, :3] @ scan[i, :3].astype(np.float64) + T0[:3, 3]).astype(np.float32).astype(np.float64)
    xn = np.linalg.norm(x); d = (q @ x + 1) / xn
    rr = np.sqrt(np.sqrt(np.float32((scan[i, :3].astype(np.float32) ** 2).sum(dtype=np.float32)), dtype=np.float32), dtype=np.float32)
    s = 1 - 0.9 * abs(d) / float(rr)
    n = x / xn
    print('numpy row', np.concatenate([s * n, s * np.cross(q, n), [s * d]]))

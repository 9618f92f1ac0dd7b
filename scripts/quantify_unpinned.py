"""How much can the two documented, UNPINNABLE readings of the reference matter?  (CPU only, the oracle both ways.)

  A  LoamRegister.cpp:147-148  `sqrt(sqrt(x*x+y*y+z*z))` on floats: both roots in float (implemented) or in double.
  B  ndt_omp_impl.hpp:109      `eig_transformation.rotation()`: the linear part (implemented) or Eigen's SVD polar factor.
Prints, for BASELINE-sized inputs, the number of weight-gate decisions that differ per linearisation, and the difference of the
final poses.  The numbers are quoted in DESIGN.md section 2.   python scripts/quantify_unpinned.py [--small]
"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
from simpleslam_amd import synth

small = "--small" in sys.argv
S = 20261003
n_map, n_scans = (100_000, 3) if small else (1_000_000, 8)
world, m = synth.make_map(n_map, seed=S + 2)
tree = oracle.KdTree(m)
prm = oracle.loam_params(iters=10, early_exit=0, threads=8)
flips, dts, drs, ws = [], [], [], []
for k in range(n_scans):
    scan, T = synth.make_scan(world, k, seed=S + 2)
    T0 = synth.perturb(T, S + 2 + k)
    out = {}
    for v in (0, 1):
        oracle.set_variant(0, v)
        out[v] = (oracle.loam_scan2map(scan, m, T0, prm)[0], oracle.loam_linearize(tree, scan, T0, per_point=True), oracle.loam_linearize(tree, scan, T, per_point=True))
    oracle.set_variant(0, 0)
    for j in (1, 2):
        flips.append(int((out[0][j]["status"] != out[1][j]["status"]).sum()))
        acc = (out[0][j]["status"] == 0) & (out[1][j]["status"] == 0)
        ws.append(float(np.abs(out[0][j]["rows"][acc] - out[1][j]["rows"][acc]).max()))
    dt, dr = synth.pose_error(out[0][0], out[1][0])
    dts.append(dt); drs.append(dr)
print(f"A  LOAM weight roots float vs double, {n_scans} scans of 65 536 points vs {n_map} map points, 10 iterations:")
print(f"   gate decisions that differ per linearisation (of 65 536): max {max(flips)}, total {sum(flips)} in {len(flips)} linearisations")
print(f"   largest difference of a row entry: {max(ws):.3e}   final pose: max {max(dts):.3e} m, {max(drs):.3e} rad")

n_map, n_scans = (300_000, 2) if small else (5_000_000, 4)
world, m = synth.make_map(n_map, seed=S + 5, spacing=0.22 if not small else 0.2)
dts, drs, its = [], [], []
for k in range(n_scans):
    scan, T = synth.make_scan(world, k, seed=S + 5, beams=128 if not small else 32, azimuths=1024 if not small else 512)
    T0 = synth.perturb(T, S + 5 + k, trans=0.1, rot_deg=0.5)
    res = {}
    for v in (0, 1):
        oracle.set_variant(1, v)
        res[v] = oracle.ndt_scan2map(scan, m, T0, oracle.ndt_params())
    oracle.set_variant(1, 0)
    if np.isfinite(res[0][0]).all() and np.isfinite(res[1][0]).all():
        dt, dr = synth.pose_error(res[0][0], res[1][0])
        dts.append(dt); drs.append(dr)
    its.append((res[0][2]["iterations"], res[1][2]["iterations"], res[0][1], res[1][1]))
print(f"B  NDT rotation() linear part vs polar factor, {n_scans} scans vs {n_map} map points:")
print(f"   iterations / converged flag (linear, polar): {its}")
print(f"   final pose (Matrix4f): max {max(dts):.3e} m, {max(drs):.3e} rad")

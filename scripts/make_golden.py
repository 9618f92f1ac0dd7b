"""Generate tests/golden/*.npz (small input/output vectors; data only).

  loam_small.npz   inputs (map, scan, initial pose) and the oracle's per-iteration normal
                   equations, increments, accepted counts and final pose (reference defaults and
                   the 10-iteration/no-early-exit benchmark setting)
  knn_nanoflann.npz  5-NN indices and squared distances produced by the REFERENCE's own vendored
                   nanoflann.hpp (oracle/_ref, built from /root/reference by oracle/Makefile)
                   for 512 queries against a 20 000-point cloud.  Needs /root/reference (or a
                   prebuilt oracle/_ref); everything else needs only this repo.
Run from the repo root:  python scripts/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
from simpleslam_amd import synth  # noqa: E402

out = os.path.join(ROOT, "tests", "golden")
os.makedirs(out, exist_ok=True)

# ---- LOAM end-to-end known-answer fixture -------------------------------------------------
world, m = synth.make_map(16000, seed=11)
scan, T = synth.make_scan(world, 0, seed=11, beams=8, azimuths=256)
T0 = synth.perturb(T, 11, trans=0.15, rot_deg=1.0)
pose_def, conv_def, info_def = oracle.loam_scan2map(scan, m, T0, trace=True)
prm10 = oracle.loam_params(iters=10, early_exit=0)
pose_10, conv_10, info_10 = oracle.loam_scan2map(scan, m, T0, prm10, trace=True)
tree = oracle.KdTree(m)
lin = oracle.loam_linearize(tree, scan, T0, per_point=True)
np.savez_compressed(
    os.path.join(out, "loam_small.npz"), map=m, scan=scan, truth=T, init=T0,
    pose_default=pose_def, converged_default=conv_def, iters_default=info_def["iters_run"],
    JtJ_default=info_def["JtJ"], JtE_default=info_def["JtE"], n_default=info_def["n"], x_default=info_def["x"],
    pose_10=pose_10, converged_10=conv_10, JtJ_10=info_10["JtJ"], JtE_10=info_10["JtE"], n_10=info_10["n"], x_10=info_10["x"],
    status0=lin["status"], nn0=lin["nn"], rows0=lin["rows"])
print("loam_small:", m.shape, scan.shape, "iters", info_def["iters_run"], "n", info_def["n"][:info_def["iters_run"]])

# ---- k-NN pinned to the reference's nanoflann ----------------------------------------------
if oracle.ref_available() or os.path.exists("/root/reference"):
    rng = np.random.default_rng(7)
    pts = np.concatenate([rng.uniform(-15, 15, (12000, 3)), rng.normal(0, 2, (8000, 3))]).astype(np.float32)
    pts = np.concatenate([pts, pts[:50]])                       # exact duplicates: distance ties
    q = np.concatenate([rng.uniform(-16, 16, (384, 3)), pts[rng.integers(0, len(pts), 128)] + rng.normal(0, 1e-3, (128, 3))]).astype(np.float32)
    pts4 = np.concatenate([pts, np.zeros((len(pts), 1), np.float32)], 1)
    q4 = np.concatenate([q, np.zeros((len(q), 1), np.float32)], 1)
    idx, d2 = oracle.ref_knn(pts4, q4, 5)
    np.savez_compressed(os.path.join(out, "knn_nanoflann.npz"), points=pts4, queries=q4, idx=idx, d2=d2)
    print("knn_nanoflann:", pts4.shape, q4.shape)
else:
    print("reference absent and oracle/_ref not built: knn_nanoflann.npz not regenerated")

# ---- VGICP and NDT known-answer fixtures (oracle outputs on small synthetic inputs) ----------
world, m = synth.make_map(20000, seed=12)
scan, T = synth.make_scan(world, 0, seed=12, beams=8, azimuths=256)
T0 = synth.perturb(T, 12, trans=0.2, rot_deg=1.5)
sc, dc = oracle.vgicp_covariances(scan, 20), oracle.vgicp_covariances(m, 20)
lin = oracle.vgicp_linearize(scan, m, T0, sc, dc)
pose, conv, info = oracle.vgicp_scan2map(scan, m, T0)
np.savez_compressed(os.path.join(out, "vgicp_small.npz"), map=m, scan=scan, truth=T, init=T0, src_cov=sc[::16], H=lin["H"], b=lin["b"],
                    err=lin["err"], n_corr=lin["n"], pose=pose, converged=conv, outer=info["outer"],
                    fitness=oracle.fitness_score(scan, m, pose))
print("vgicp_small:", conv, info, synth.pose_error(pose, T))

world, m = synth.make_map(40000, seed=13, spacing=0.2)
scan, T = synth.make_scan(world, 0, seed=13, beams=8, azimuths=256)
T0 = synth.perturb(T, 13, trans=0.08, rot_deg=0.4)
from scipy.spatial.transform import Rotation as Rot
p6 = np.concatenate([T0[:3, 3], Rot.from_matrix(T0[:3, :3]).as_euler("XYZ")])
d = oracle.ndt_derivatives(scan, m, p6, double_hessian=True)
pose, conv, info = oracle.ndt_scan2map(scan, m, T0)
assert np.isfinite(pose).all()
np.savez_compressed(os.path.join(out, "ndt_small.npz"), map=m, scan=scan, truth=T, init=T0, p6=p6, score=d["score"], grad=d["grad"],
                    hess=d["hess"], hess_d=d["hess_d"], pose=pose, converged=conv, iterations=info["iterations"])
print("ndt_small:", conv, info, synth.pose_error(pose, T), synth.pose_error(T0, T))

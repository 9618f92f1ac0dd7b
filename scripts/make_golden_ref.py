"""Generate tests/golden/ref_nanoflann_more.npz: answers of the REFERENCE's own nanoflann clients (oracle/_ref, built from
/root/reference by oracle/Makefile -- data only, no reference source) for the three searches next to the hot path:

  knn_*     PointCloudKdtree (pcl_adaptor.hpp) 5-NN on a cloud WITHOUT duplicated points: every neighbour list is unique, so the
            oracle and the HIP grid search must reproduce all of them (the older knn_nanoflann.npz holds duplicates on purpose).
  kfs_*     KeyFramesKdtree::radiusSearch (kfs_adaptor.hpp) as MapManager::updateMap selects the key frames of a sub-map.
  vov_*     VectorOfVectorsKdTree 10-NN over ScanContext ring keys (vov_adaptor.h) as ScanContext::query picks its candidates.
Run from the repo root where /root/reference (or a prebuilt oracle/_ref) is available:  python scripts/make_golden_ref.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

rng = np.random.default_rng(20261004)
# ---- 5-NN without duplicates ----
pts = np.unique(np.concatenate([rng.uniform(-7, 7, (22000, 3)), rng.normal(0, 2, (10000, 3))]).astype(np.float32), axis=0)
rng.shuffle(pts)
q = np.concatenate([rng.uniform(-8, 8, (1536, 3)), pts[rng.integers(0, len(pts), 512)] + rng.normal(0, 5e-3, (512, 3))]).astype(np.float32)
pts4 = np.concatenate([pts, np.zeros((len(pts), 1), np.float32)], 1)
q4 = np.concatenate([q, np.zeros((len(q), 1), np.float32)], 1)
idx, d2 = oracle.ref_knn(pts4, q4, 5)
ties = (d2[:, 1:] == d2[:, :-1]).any(1)
keep = ~ties                                                    # (exactly equal distances between distinct points: none expected, dropped if any)
print("knn without duplicates:", pts4.shape, q4.shape, "rows with distance ties dropped:", int(ties.sum()))
# ---- key-frame radius search ----
t = np.linspace(0, 1, 400)
traj = np.stack([120 * t + 3 * np.sin(40 * t), 25 * np.sin(9 * t), 0.3 * np.cos(17 * t)], 1) + rng.normal(0, 0.3, (400, 3))
kq = np.concatenate([traj[rng.integers(0, 400, 40)] + rng.normal(0, 2.0, (40, 3)), rng.uniform(-10, 130, (24, 3)) * [1, 0.3, 0.02]])
lists, counts, dists = np.full((len(kq), 400), -1, np.int64), np.zeros(len(kq), np.int64), np.zeros((len(kq), 400))
for i, qq in enumerate(kq):
    ii, dd = oracle.ref_keyframes_radius(traj, qq, 8.0)
    lists[i, :len(ii)], counts[i], dists[i, :len(ii)] = ii, len(ii), dd
print("key-frame radius search: sizes", counts.min(), "...", counts.max())
# ---- ring-key 10-NN ----
keys = rng.uniform(0, 1, (500, 20)) * (rng.uniform(0, 1, (500, 1)) > 0.1)
keys[100:110] = keys[0:10]                                      # repeated contexts (the robot came back): exact ties
vq = np.concatenate([keys[rng.integers(0, 500, 32)] + rng.normal(0, 0.02, (32, 20)), rng.uniform(0, 1, (32, 20))])
vidx, vd2 = np.zeros((len(vq), 10), np.int64), np.zeros((len(vq), 10))
for i, qq in enumerate(vq):
    vidx[i], vd2[i] = oracle.ref_ring_key_knn(keys, qq, 10)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_nanoflann_more.npz"),
                    knn_points=pts4, knn_queries=q4[keep], knn_idx=idx[keep], knn_d2=d2[keep],
                    kfs_positions=traj, kfs_queries=kq, kfs_lists=lists, kfs_counts=counts, kfs_d2=dists,
                    vov_keys=keys, vov_queries=vq, vov_idx=vidx, vov_d2=vd2)
print("written")

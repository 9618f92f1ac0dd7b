"""The ScanContext oracle against an independent numpy evaluation of the same formulas (the reference holds no
vectors for this step: parity unpinned, see oracle/scancontext_oracle.c)."""
import numpy as np

import oracle


def _numpy_descriptor(p, lidar_height=2.0):
    x, y = p[:, 0].astype(np.float32), p[:, 1].astype(np.float32)
    z = p[:, 2].astype(np.float32) + np.float32(lidar_height)
    rng = np.sqrt(x * x + y * y)
    th = (np.arctan2(y, x).astype(np.float64) + np.pi).astype(np.float32)
    th = np.clip(th, np.float32(0), np.float32(2 * np.pi))
    ang = (th.astype(np.float64) * 180.0 / np.pi).astype(np.float32)
    keep = rng <= np.float32(80)
    ring = np.clip(np.ceil(rng / np.float32(80) * np.float32(20)).astype(int), 1, 20) - 1
    sec = np.clip(np.ceil(ang.astype(np.float64) / 360.0 * 60).astype(int), 1, 60) - 1
    d = np.full((20, 60), -1000.0)
    np.maximum.at(d, (ring[keep], sec[keep]), z[keep].astype(np.float64))
    d[d == -1000.0] = 0
    return d


def test_descriptor_and_keys():
    rng = np.random.default_rng(3)
    p = np.zeros((5000, 4), np.float32)
    p[:, :2] = rng.uniform(-70, 70, (5000, 2))
    p[:, 2] = rng.uniform(-1.5, 5, 5000)
    o = oracle.ScanContextOracle()
    o.add(p)
    d = _numpy_descriptor(p)
    assert np.array_equal(o.descriptor(0), d)
    assert np.allclose(o.ring[0], d.mean(1), rtol=0, atol=1e-13) and np.allclose(o.sector[0], d.mean(0), rtol=0, atol=1e-13)


def test_distance_is_cosine_distance_at_best_shift():
    rng = np.random.default_rng(4)
    o = oracle.ScanContextOracle(search_ratio=1.0)          # the window covers every shift
    for s in range(2):
        p = np.zeros((4000, 4), np.float32)
        p[:, :2] = rng.uniform(-60, 60, (4000, 2))
        p[:, 2] = rng.uniform(-1.5, 5, 4000)
        o.add(p)
    a, b = o.descriptor(0), o.descriptor(1)
    best = min(_cos_dist(a, np.roll(b, s, axis=1)) for s in range(60))
    d, s = o.distance(0, 1)
    assert abs(d - best) < 1e-12 and abs(_cos_dist(a, np.roll(b, s, axis=1)) - d) < 1e-12
    assert o.distance(0, 0) == (0.0, 0) or abs(o.distance(0, 0)[0]) < 1e-15


def _cos_dist(a, b):
    na, nb = np.linalg.norm(a, axis=0), np.linalg.norm(b, axis=0)
    ok = (na > 0) & (nb > 0)
    return 1.0 - ((a * b).sum(0)[ok] / (na[ok] * nb[ok])).sum() / ok.sum()


def test_query_exclusion_and_snapshot():
    o = oracle.ScanContextOracle(num_exclude_recent=5, build_tree_gap=3, num_candidates=2, dist_thres=2.0)
    rng = np.random.default_rng(5)
    sizes = []
    for i in range(20):
        p = np.zeros((800, 4), np.float32)
        p[:, :2] = rng.uniform(-60, 60, (800, 2))
        p[:, 2] = rng.uniform(-1.5, 5, 800)
        o.add(p)
        m, yaw, d = o.query(i)
        sizes.append(o.tree_size)
        if i <= 7:
            assert m == -1 and d is None
        else:
            assert 0 <= m < o.tree_size <= i - 5
    # rebuilt at 8 (-> 3), then only when id - size > 8: at 12 (-> 7), 16 (-> 11)
    assert sizes[8:] == [3, 3, 3, 3, 7, 7, 7, 7, 11, 11, 11, 11]

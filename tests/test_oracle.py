"""CPU tests (no GPU): the oracle against independent references and the committed golden
vectors.  The reference ships no fixtures; its Eigen calls are pinned here
against numpy/scipy, and its k-NN against the reference's own vendored nanoflann."""
import os

import numpy as np
import pytest
from scipy.linalg import expm
from scipy.spatial.transform import Rotation

import oracle
from simpleslam_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _hat(k):
    X = np.zeros((4, 4))
    w = k[3:]
    X[:3, :3] = [[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]]
    X[:3, 3] = k[:3]
    return X


def test_colpiv_qr_matches_lstsq():
    rng = np.random.default_rng(0)
    for _ in range(500):
        A = rng.normal(size=(5, 3)) * rng.uniform(0.1, 50)
        b = rng.normal(size=5)
        x, rank = oracle.colpiv_qr_solve_5x3(A, b)
        xr = np.linalg.lstsq(A, b, rcond=None)[0]
        assert rank == 3
        np.testing.assert_allclose(x, xr, rtol=1e-10, atol=1e-12)


def test_colpiv_qr_rank_deficient_is_basic_solution():
    # collinear neighbours: Eigen's ColPivHouseholderQR::solve returns a basic solution (zeros
    # for the dropped columns), not numpy's minimum-norm one
    A = np.outer(np.arange(5.0) + 1, [1, 2, 3.0])
    x, rank = oracle.colpiv_qr_solve_5x3(A, -np.ones(5))
    assert rank == 1
    assert np.count_nonzero(x) == 1
    xv, ok = oracle.plane_fit5(A)
    assert not ok  # and the 0.2 validity gate (LoamRegister.cpp:38-43) rejects it


def test_plane_fit_gate():
    rng = np.random.default_rng(1)
    n = np.array([0.2, -0.3, 0.93]); n /= np.linalg.norm(n)
    base = rng.uniform(-1, 1, (5, 3)); base -= np.outer(base @ n, n)       # points on a plane through c
    c = np.array([10.0, -4.0, 3.0])
    x, ok = oracle.plane_fit5(base + c)
    assert ok
    np.testing.assert_allclose(x / np.linalg.norm(x), -n * np.sign(n @ c) * 1.0, atol=1e-9)
    bent = base + c
    bent[0] += 3.0 * n                      # one neighbour 3 m off the plane: residuals exceed 0.2 m
    assert oracle.plane_fit5(bent)[1] is False


def test_ldlt_matches_solve():
    rng = np.random.default_rng(2)
    for _ in range(500):
        J = rng.normal(size=(40, 6)) * rng.uniform(0.1, 30, size=6)
        M = J.T @ J
        b = rng.normal(size=6)
        np.testing.assert_allclose(oracle.ldlt6_solve(M, b), np.linalg.solve(M, b), rtol=1e-9, atol=1e-12)


def test_se3_exp_matches_expm():
    # the one se(3) sample the reference holds (test/eigen.cpp:71), which has no expected output there
    k = np.array([-0.00373127, 0.00599259, 0.00010917, -0.000599459, 0.000276421, 2.11126e-05])
    assert np.abs(oracle.se3_exp(k) - expm(_hat(k))).max() < 1e-15
    rng = np.random.default_rng(3)
    for _ in range(200):
        k = rng.normal(size=6) * rng.uniform(1e-8, 2)
        assert np.abs(oracle.se3_exp(k) - expm(_hat(k))).max() < 1e-12
    # small-angle branch (manifolds.hpp:41-44): theta < 1e-6 -> R = I, t = rho
    k = np.array([1.0, 2.0, 3.0, 1e-8, 0, 0])
    T = oracle.se3_exp(k)
    assert (T[:3, :3] == np.eye(3)).all() and (T[:3, 3] == k[:3]).all()


def test_t2se3_orthonormalises():
    rng = np.random.default_rng(4)
    for i in range(200):
        R = Rotation.random(random_state=i).as_matrix()
        T = np.eye(4); T[:3, :3] = R + rng.normal(size=(3, 3)) * 1e-6; T[:3, 3] = rng.normal(size=3)
        T2 = oracle.t2se3(T)
        assert np.abs(T2[:3, :3] @ T2[:3, :3].T - np.eye(3)).max() < 1e-14
        assert np.abs(T2[:3, :3] - R).max() < 1e-5
        assert (T2[:3, 3] == T[:3, 3]).all()


def test_knn_tree_equals_brute_force():
    rng = np.random.default_rng(5)
    pts = rng.uniform(-5, 5, (3000, 4)).astype(np.float32)
    pts[100:110] = pts[0:10]                      # duplicates: ties broken on the lower index
    q = rng.uniform(-5, 5, (300, 3))
    q[:10] = pts[0:10, :3]
    tree = oracle.KdTree(pts)
    i1, d1 = tree.knn(q, 5)
    i2, d2 = oracle.knn_brute(pts, q, 5)
    np.testing.assert_array_equal(i1, i2)
    np.testing.assert_array_equal(d1, d2)
    assert (np.diff(d1, axis=1) >= 0).all()


def test_knn_matches_reference_nanoflann_golden():
    """Golden vectors produced by the reference's vendored nanoflann.hpp (scripts/make_golden.py)."""
    g = np.load(os.path.join(GOLD, "knn_nanoflann.npz"))
    tree = oracle.KdTree(g["points"])
    idx, d2 = tree.knn(g["queries"][:, :3].astype(np.float64), 5)
    np.testing.assert_array_equal(d2, g["d2"])               # f64 distances on f32 coordinates, bit for bit
    same = idx == g["idx"]
    # index lists may differ only inside groups of exactly equal distance (nanoflann breaks ties
    # by traversal order, the oracle by index; the fixture contains duplicated points)
    for r, c in zip(*np.nonzero(~same)):
        assert (g["d2"][r] == g["d2"][r, c]).sum() >= 2 or g["points"][idx[r, c], :3].tolist() == g["points"][g["idx"][r, c], :3].tolist()
    assert same.mean() > 0.98


@pytest.mark.skipif(not oracle.ref_available(), reason="oracle/_ref (reference nanoflann) not built")
def test_knn_matches_reference_nanoflann_live():
    rng = np.random.default_rng(6)
    pts = rng.uniform(-20, 20, (30000, 4)).astype(np.float32)
    q = rng.uniform(-20, 20, (1000, 4)).astype(np.float32)
    i_ref, d_ref = oracle.ref_knn(pts, q, 5)
    i_or, d_or = oracle.KdTree(pts).knn(q[:, :3].astype(np.float64), 5)
    np.testing.assert_array_equal(i_or, i_ref)
    np.testing.assert_array_equal(d_or, d_ref)


def _rows_numpy(m, scan, pose, i, nn):
    """Independent transcription of LoamRegister.cpp:126-159 for one accepted point."""
    A = m[nn, :3].astype(np.float64)
    x = np.linalg.lstsq(A, -np.ones(5), rcond=None)[0]
    q = (pose[:3, :3] @ scan[i, :3].astype(np.float64) + pose[:3, 3]).astype(np.float32).astype(np.float64)
    xn = np.linalg.norm(x)
    d = (q @ x + 1) / xn
    r2 = np.float32(0)
    for c in scan[i, :3]:
        r2 = np.float32(r2 + np.float32(c) * np.float32(c))
    rr = np.sqrt(np.sqrt(r2, dtype=np.float32), dtype=np.float32)
    s = 1 - 0.9 * abs(d) / float(rr)
    n = x / xn
    return np.concatenate([s * n, s * np.cross(q, n), [s * d]])


def test_rows_match_numpy_transcription(world_small):
    """Pins the f32 round trip of the transformed point (LoamRegister.cpp:128-130): gcc -O3
    once dropped it in the oracle (see oracle/Makefile)."""
    w = world_small
    tree = oracle.KdTree(w["map"])
    o = oracle.loam_linearize(tree, w["scan"], w["init"], per_point=True)
    acc = np.nonzero(o["status"] == 0)[0]
    assert acc.size > 1000
    for i in acc[:: max(1, acc.size // 60)]:
        np.testing.assert_allclose(o["rows"][i], _rows_numpy(w["map"], w["scan"], w["init"], i, o["nn"][i]), rtol=1e-9, atol=1e-12)
    # and the sums are the sums of the rows
    R = o["rows"][acc]
    np.testing.assert_allclose(o["JtJ"], R[:, :6].T @ R[:, :6], rtol=1e-12)
    np.testing.assert_allclose(o["JtE"], R[:, :6].T @ R[:, 6], rtol=1e-10, atol=1e-12)
    assert o["n"] == acc.size


def test_loam_golden_fixture():
    g = np.load(os.path.join(GOLD, "loam_small.npz"))
    pose, conv, info = oracle.loam_scan2map(g["scan"], g["map"], g["init"], trace=True)
    assert conv == bool(g["converged_default"]) and info["iters_run"] == int(g["iters_default"])
    np.testing.assert_array_equal(info["n"], g["n_default"])
    np.testing.assert_allclose(info["JtJ"], g["JtJ_default"], rtol=1e-13, atol=1e-13)
    np.testing.assert_allclose(pose, g["pose_default"], rtol=0, atol=1e-13)
    pose10, conv10, info10 = oracle.loam_scan2map(g["scan"], g["map"], g["init"], oracle.loam_params(iters=10, early_exit=0), trace=True)
    assert conv10 is False and info10["iters_run"] == 10
    np.testing.assert_array_equal(info10["n"], g["n_10"])
    np.testing.assert_allclose(info10["x"], g["x_10"], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(pose10, g["pose_10"], rtol=0, atol=1e-13)
    # the registration reaches the synthetic truth
    dt, dr = synth.pose_error(pose10, g["truth"])
    assert dt < 0.03 and dr < 3e-3


def test_oracle_thread_count_invariance(world_small):
    """OpenMP team size (`cores`) only changes the summation grouping."""
    w = world_small
    p1, c1, _ = oracle.loam_scan2map(w["scan"], w["map"], w["init"], oracle.loam_params(threads=1))
    p4, c4, _ = oracle.loam_scan2map(w["scan"], w["map"], w["init"], oracle.loam_params(threads=4))
    assert c1 == c4
    dt, dr = synth.pose_error(p1, p4)
    assert dt < 1e-10 and dr < 1e-10


def test_early_exit_discards_small_step(world_small):
    """Convergence is tested BEFORE the increment is applied."""
    w = world_small
    pose, conv, info = oracle.loam_scan2map(w["scan"], w["map"], w["init"], trace=True)
    assert conv
    k = info["iters_run"]
    x_last = info["x"][k - 1]
    assert np.linalg.norm(x_last[:3]) <= 5e-3 and np.linalg.norm(x_last[3:]) <= 5e-3
    # replay: applying only the first k-1 increments reproduces the returned pose
    T = w["init"].copy()
    for i in range(k - 1):
        T = oracle.se3_exp(info["x"][i]) @ T
    np.testing.assert_allclose(oracle.t2se3(T), pose, atol=1e-12)


def test_too_few_points():
    m = np.zeros((3, 4), np.float32)
    scan = np.ones((10, 4), np.float32)
    pose, conv, info = oracle.loam_scan2map(scan, m, np.eye(4), trace=True)
    assert conv is False and info["iters_run"] == 1 and info["n"][0] == 0
    np.testing.assert_allclose(pose, np.eye(4), atol=1e-15)

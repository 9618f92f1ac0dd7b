"""GPU parity of the LOAM scan-to-map path: HIP (through the C ABI) vs the CPU oracle.

Tolerance stated by BASELINE.json's north_star: pose within 1e-4 m / 1e-4 rad of the CPU
reference after the same iteration count.  Integer/index work (k-NN sets, gate outcomes,
accepted counts) must match exactly.
"""
import numpy as np
import pytest

import oracle
from simpleslam_amd import LoamRegister, synth

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4
POSE_TOL_RAD = 1e-4
BITWISE_ROW_MISMATCH = -1


def _check_linearize(reg, tree, scan, pose):
    g = reg.linearize(scan, pose, per_point=True)
    o = oracle.loam_linearize(tree, scan, pose, oracle.loam_params(), per_point=True)
    assert g["n"] == o["n"]
    np.testing.assert_array_equal(g["status"], o["status"])       # every gate outcome identical
    ok = o["status"] != 1                                           # k-NN set defined when the grid was searched
    acc = o["status"] == 0
    np.testing.assert_array_equal(g["nn"][acc], o["nn"][acc])      # same neighbours in the same order
    # rows: the same IEEE operation sequence on both sides (contraction off on both)
    np.testing.assert_allclose(g["rows"][acc], o["rows"][acc], rtol=1e-12, atol=1e-14)
    global BITWISE_ROW_MISMATCH
    BITWISE_ROW_MISMATCH = int((g["rows"][acc] != o["rows"][acc]).sum())
    np.testing.assert_allclose(g["JtJ"], o["JtJ"], rtol=1e-11, atol=1e-9)
    np.testing.assert_allclose(g["JtE"], o["JtE"], rtol=1e-9, atol=1e-9)
    return g, o, ok


def test_linearize_matches_oracle_per_point(gpu, world_small):
    w = world_small
    reg = LoamRegister()
    reg.setTarget(w["map"])
    tree = oracle.KdTree(w["map"])
    g, o, _ = _check_linearize(reg, tree, w["scan"], w["init"])
    assert g["n"] > 1000
    # and at the true pose (different gate populations)
    _check_linearize(reg, tree, w["scan"], w["truth"])


def test_linearize_100k(gpu, world_100k):
    w = world_100k
    reg = LoamRegister()
    reg.setTarget(w["map"])
    tree = oracle.KdTree(w["map"])
    g, o, _ = _check_linearize(reg, tree, w["scan"], w["init"])
    assert g["n"] > 30000
    print("row elements not bitwise equal to the oracle:", BITWISE_ROW_MISMATCH)


def test_scan2map_default_params(gpu, world_100k):
    """Reference defaults: 8 iterations, early exit (LoamRegister.hpp:40, LoamRegister.cpp:202-206)."""
    w = world_100k
    reg = LoamRegister(record_trace=1)
    pose = w["init"].copy()
    conv = reg.scan2Map(w["scan"], w["map"], pose)
    po, co, info = oracle.loam_scan2map(w["scan"], w["map"], w["init"], trace=True)
    assert conv == co
    tr = reg.trace()
    assert tr["iters_run"] == info["iters_run"]
    np.testing.assert_array_equal(tr["n"], info["n"][: tr["iters_run"]])
    dt, dr = synth.pose_error(pose, po)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (dt, dr)
    assert dt < 1e-9 and dr < 1e-9      # in practice the two agree to rounding
    # and both found the truth
    et, er = synth.pose_error(pose, w["truth"])
    assert et < 0.02 and er < 2e-3


def test_scan2map_10_iters_no_early_exit(gpu, world_100k):
    """BASELINE config: 10 GN iterations, early exit off (throughput setting)."""
    w = world_100k
    reg = LoamRegister(loam_iters=10, loam_early_exit=0, record_trace=1)
    pose = w["init"].copy()
    conv = reg.scan2Map(w["scan"], w["map"], pose)
    po, co, info = oracle.loam_scan2map(w["scan"], w["map"], w["init"], oracle.loam_params(iters=10, early_exit=0), trace=True)
    assert conv == co == False
    tr = reg.trace()
    assert tr["iters_run"] == 10 == info["iters_run"]
    np.testing.assert_array_equal(tr["n"], info["n"])
    np.testing.assert_allclose(tr["x"], info["x"], rtol=1e-6, atol=1e-12)
    dt, dr = synth.pose_error(pose, po)
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD, (dt, dr)


def test_pcl_point_layout_stride32(gpu, world_small):
    """pcl::PointXYZI is 32 bytes (x y z 1 | intensity pad pad pad)."""
    w = world_small

    def widen(c):
        out = np.zeros((c.shape[0], 8), np.float32)
        out[:, :3] = c[:, :3]
        out[:, 3] = 1.0
        out[:, 4] = c[:, 3]
        return out

    a, b = LoamRegister(), LoamRegister()
    p16, p32 = w["init"].copy(), w["init"].copy()
    c16 = a.scan2Map(w["scan"], w["map"], p16)
    c32 = b.scan2Map(widen(w["scan"]), widen(w["map"]), p32)
    assert c16 == c32
    np.testing.assert_array_equal(p16, p32)


def test_device_resident_inputs(gpu, world_small):
    import torch
    w = world_small
    reg = LoamRegister()
    ph, pd = w["init"].copy(), w["init"].copy()
    ch = reg.scan2Map(w["scan"], w["map"], ph)
    cd = reg.scan2Map(torch.from_numpy(w["scan"]).cuda(), torch.from_numpy(w["map"]).cuda(), pd)
    assert ch == cd
    np.testing.assert_array_equal(ph, pd)


def test_static_target_align_equals_scan2map(gpu, world_small):
    w = world_small
    reg = LoamRegister()
    p1, p2 = w["init"].copy(), w["init"].copy()
    c1 = reg.scan2Map(w["scan"], w["map"], p1)
    reg.setTarget(w["map"])
    c2 = reg.align(w["scan"], p2)
    c3 = reg.align(w["scan"], p2.copy())   # index reused
    assert c1 == c2
    np.testing.assert_array_equal(p1, p2)


def test_repeatable(gpu, world_small):
    """Fixed-order reductions: two runs are bitwise identical."""
    w = world_small
    reg = LoamRegister()
    p1, p2 = w["init"].copy(), w["init"].copy()
    reg.scan2Map(w["scan"], w["map"], p1)
    reg.scan2Map(w["scan"], w["map"], p2)
    np.testing.assert_array_equal(p1, p2)


def test_edge_cases(gpu, world_small):
    w = world_small
    reg = LoamRegister()
    # empty source: fewer than 6 valid points -> not converged, pose only re-orthonormalised
    p = w["init"].copy()
    assert reg.scan2Map(np.zeros((0, 4), np.float32), w["map"], p) is False
    np.testing.assert_allclose(p, w["init"], atol=1e-12)
    # empty target
    p = w["init"].copy()
    assert reg.scan2Map(w["scan"], np.zeros((0, 4), np.float32), p) is False
    np.testing.assert_allclose(p, w["init"], atol=1e-12)
    # target with fewer than 5 points
    p = w["init"].copy()
    assert reg.scan2Map(w["scan"], w["map"][:4], p) is False
    # NaN rows in the scan and in the map are ignored
    scan = w["scan"].copy(); scan[::97, :3] = np.nan
    m = w["map"].copy(); m[::101, 0] = np.nan
    p = w["init"].copy()
    reg.scan2Map(scan, m, p)
    keep_s = np.isfinite(scan[:, :3]).all(1)
    keep_m = np.isfinite(m[:, :3]).all(1)
    po, co, _ = oracle.loam_scan2map(scan[keep_s], m[keep_m], w["init"])
    dt, dr = synth.pose_error(p, po)
    assert dt < 1e-9 and dr < 1e-9
    # a scan point at the sensor origin: the weight 1 - 0.9|d| / sqrt(sqrt(0)) divides by zero (LoamRegister.cpp:147-148)
    s0 = w["scan"].copy(); s0[0, :3] = 0.0
    p = w["init"].copy()
    c0 = reg.scan2Map(s0, w["map"], p)
    po, co, _ = oracle.loam_scan2map(s0, w["map"], w["init"])
    assert c0 == co and np.isfinite(p).all() == np.isfinite(po).all()
    if np.isfinite(po).all():
        dt, dr = synth.pose_error(p, po)
        assert dt < 1e-9 and dr < 1e-9
    # scan far outside the map: nothing within 1 m
    far = w["scan"].copy(); far[:, :3] += 5000.0
    p = w["init"].copy()
    assert reg.scan2Map(far, w["map"], p) is False


def test_first_call_after_cell_table_growth(gpu, world_100k):
    """A gate radius of 0.25 m makes the 0.25 m grid overflow the initial cell table: the very first call has to grow it
    and rebuild.  Its result must be the one every later call gives (the counters of a fresh table must be zero before
    the rebuild's histogram runs)."""
    w = world_100k
    reg = LoamRegister(loam_knn_max_sq=0.0625)
    p1, p2 = w["init"].copy(), w["init"].copy()
    c1 = reg.scan2Map(w["scan"], w["map"], p1)
    c2 = reg.scan2Map(w["scan"], w["map"], p2)
    assert c1 == c2
    np.testing.assert_array_equal(p1, p2)
    po, co, _ = oracle.loam_scan2map(w["scan"], w["map"], w["init"], oracle.loam_params(knn_max_sq=0.0625))
    dt, dr = synth.pose_error(p1, po)
    assert dt < 1e-9 and dr < 1e-9


@pytest.mark.parametrize("n_src", [1, 63, 100, 257, 1000, 4097])
def test_ragged_scan_sizes(gpu, world_small, n_src):
    """Scan sizes that leave lanes, whole waves and whole blocks without a query (the cache prefetch of such waves must
    still be complete before its LDS staging area is reused)."""
    w = world_small
    scan = w["scan"][:: max(1, w["scan"].shape[0] // n_src)][:n_src]
    assert scan.shape[0] == n_src
    reg = LoamRegister(loam_iters=6, loam_early_exit=0)
    for _ in range(2):
        pose = w["init"].copy()
        conv = reg.scan2Map(scan, w["map"], pose)
    po, co, _ = oracle.loam_scan2map(scan, w["map"], w["init"], oracle.loam_params(iters=6, early_exit=0))
    assert conv == co
    if np.isfinite(po).all():
        dt, dr = synth.pose_error(pose, po)
        assert dt < 1e-9 and dr < 1e-9


def test_randomised_configurations(gpu):
    """Twelve seeded draws of world size, scan shape, initial error and gate/iteration parameters: the device path and the
    oracle must agree on convergence, on the number of iterations consumed and on the pose (discrete gates included)."""
    rng = np.random.default_rng(424242)
    for case in range(12):
        n_map = int(rng.integers(4_000, 60_000))
        beams, az = int(rng.choice([8, 16, 32])), int(rng.choice([128, 256, 512]))
        world, m = synth.make_map(n_map, seed=1000 + case)
        scan, T = synth.make_scan(world, int(rng.integers(0, 5)), seed=1000 + case, beams=beams, azimuths=az)
        init = synth.perturb(T, 2000 + case, trans=float(rng.uniform(0.02, 0.4)), rot_deg=float(rng.uniform(0.1, 3.0)))
        kw = dict(iters=int(rng.integers(2, 12)), early_exit=int(rng.integers(0, 2)), knn_max_sq=float(rng.choice([0.5, 1.0, 2.0, 4.0])),
                  plane_thresh=float(rng.uniform(0.1, 0.3)), point_thresh=float(rng.uniform(0.05, 0.3)))
        reg = LoamRegister(loam_iters=kw["iters"], loam_early_exit=kw["early_exit"], loam_knn_max_sq=kw["knn_max_sq"],
                           loam_plane_thresh=kw["plane_thresh"], loam_point_thresh=kw["point_thresh"], record_trace=1)
        pose = init.copy()
        conv = reg.scan2Map(scan, m, pose)
        po, co, info = oracle.loam_scan2map(scan, m, init, oracle.loam_params(**kw))
        assert conv == co, (case, kw)
        assert reg.trace()["iters_run"] == info["iters_run"], (case, kw)
        dt, dr = synth.pose_error(pose, po)
        assert dt < 1e-9 and dr < 1e-9, (case, kw, dt, dr)


def test_unknown_method_raises():
    from simpleslam_amd import make_register
    with pytest.raises(RuntimeError):
        make_register("icp")


def test_golden_fixture(gpu):
    """HIP path against the committed known-answer vectors (tests/golden/loam_small.npz)."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "loam_small.npz"))
    reg = LoamRegister(record_trace=1)
    pose = g["init"].copy()
    conv = reg.scan2Map(g["scan"], g["map"], pose)
    tr = reg.trace()
    assert conv == bool(g["converged_default"]) and tr["iters_run"] == int(g["iters_default"])
    np.testing.assert_array_equal(tr["n"], g["n_default"][: tr["iters_run"]])
    dt, dr = synth.pose_error(pose, g["pose_default"])
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD
    reg10 = LoamRegister(loam_iters=10, loam_early_exit=0, record_trace=1)
    pose = g["init"].copy()
    assert reg10.scan2Map(g["scan"], g["map"], pose) is False
    tr = reg10.trace()
    np.testing.assert_array_equal(tr["n"], g["n_10"])
    np.testing.assert_allclose(tr["JtJ"], g["JtJ_10"], rtol=1e-9, atol=1e-9)
    dt, dr = synth.pose_error(pose, g["pose_10"])
    assert dt <= POSE_TOL_M and dr <= POSE_TOL_RAD
    # per-point: gate outcomes and neighbour lists of the first linearisation
    reg.setTarget(g["map"])
    lin = reg.linearize(g["scan"], g["init"], per_point=True)
    np.testing.assert_array_equal(lin["status"], g["status0"])
    acc = g["status0"] == 0
    np.testing.assert_array_equal(lin["nn"][acc], g["nn0"][acc])
    np.testing.assert_allclose(lin["rows"][acc], g["rows0"][acc], rtol=1e-12, atol=1e-14)


# (the comparisons with the reference's own nanoflann -- 100 % on a duplicate-free cloud, tie groups only on the one with
#  duplicates -- live in tests/test_reference_pins.py)


def test_temporal_cache_is_exact(gpu, world_100k):
    """Iterations after the first reuse cached neighbours when provably valid; disabling the
    cache (pcr_params.loam_disable_cache) must not change a single bit of the result."""
    from simpleslam_amd.pcr import default_params
    w = world_100k
    p_on = default_params(loam_iters=10, loam_early_exit=0, record_trace=1)
    p_off = default_params(loam_iters=10, loam_early_exit=0, record_trace=1)
    p_off.loam_disable_cache = 1
    a, b = LoamRegister(params=p_on), LoamRegister(params=p_off)
    pa, pb = w["init"].copy(), w["init"].copy()
    a.scan2Map(w["scan"], w["map"], pa)
    b.scan2Map(w["scan"], w["map"], pb)
    np.testing.assert_array_equal(pa, pb)
    ta, tb = a.trace(), b.trace()
    np.testing.assert_array_equal(ta["n"], tb["n"])
    np.testing.assert_array_equal(ta["JtJ"], tb["JtJ"])


def test_distance_ties_resolved_on_original_index(gpu):
    """Duplicated target points give bitwise-equal distances; the neighbour lists must then follow the
    documented (distance, original index) order exactly, like the oracle's."""
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "knn_nanoflann.npz"))
    pts = np.concatenate([g["points"], g["points"][:2000]])          # 2000 more exact duplicates
    q = np.concatenate([g["queries"], g["points"][:300] + np.float32(1e-3)])
    reg = LoamRegister()
    reg.setTarget(pts)
    lin = reg.linearize(q, np.eye(4), per_point=True)
    idx, d2 = oracle.KdTree(pts).knn(q[:, :3].astype(np.float64), 5)
    found = d2[:, 4] < 1.0
    np.testing.assert_array_equal(lin["status"] != 1, found)
    np.testing.assert_array_equal(lin["nn"][found], idx[found])
    assert (d2[found][:, 1:] == d2[found][:, :-1]).any()              # the fixture really contains ties


def test_query_tiles_sum_to_the_full_system(gpu, world_100k):
    """Multi-GPU sharding logic on one GPU: two map tiles (+1 m halo), each linearised only over the scan points
    that fall in its tile; the two partial systems add up to the unsharded one, gate for gate."""
    from simpleslam_amd import shard
    w = world_100k
    full = LoamRegister()
    full.setTarget(w["map"])
    ref = full.linearize(w["scan"], w["init"], per_point=True)
    JtJ, JtE, n = np.zeros((6, 6)), np.zeros(6), 0
    owned = np.zeros(w["scan"].shape[0], int)
    for r in range(2):
        tile = shard.tile_for_rank(w["map"], r, 2)
        reg = LoamRegister()
        reg.setTarget(tile.points)
        reg.set_query_tile(tile.lo, tile.hi)
        part = reg.linearize(w["scan"], w["init"], per_point=True)
        mine = part["status"] != 4
        owned += mine
        np.testing.assert_array_equal(part["status"][mine], ref["status"][mine])
        np.testing.assert_array_equal(part["rows"][mine], ref["rows"][mine])      # same neighbours through the halo
        JtJ += part["JtJ"]; JtE += part["JtE"]; n += part["n"]
    assert (owned == 1).all()
    assert n == ref["n"]
    np.testing.assert_allclose(JtJ, ref["JtJ"], rtol=1e-12, atol=1e-10)
    np.testing.assert_allclose(JtE, ref["JtE"], rtol=1e-10, atol=1e-10)


def test_rccl_allreduce_path_single_rank(gpu, world_small):
    """The sharded launch sequence (reduce kernel -> RCCL all-reduce -> prologue reads the reduced sums) with a
    one-rank communicator gives the unsharded answer."""
    from simpleslam_amd import shard
    w = world_small
    a, b = LoamRegister(), LoamRegister()
    b.comm_init(shard.unique_id(), 0, 1)
    pa, pb = w["init"].copy(), w["init"].copy()
    ca = a.scan2Map(w["scan"], w["map"], pa)
    cb = b.scan2Map(w["scan"], w["map"], pb)
    assert ca == cb
    dt, dr = synth.pose_error(pa, pb)
    assert dt < 1e-10 and dr < 1e-10


def test_coresident_kernel_variant_and_concurrent_handles(gpu, world_100k):
    """pcr_params.loam_coresident = 1 selects the two-waves-per-SIMD build of the iterate kernel: same result bit for bit;
    and independent handles may register scans from different host threads at the same time."""
    import threading
    from simpleslam_amd import pcr
    w = world_100k
    base = w["init"].copy()
    LoamRegister(loam_iters=10, loam_early_exit=0).scan2Map(w["scan"], w["map"], base)
    prm = pcr.default_params(loam_iters=10, loam_early_exit=0)
    prm.loam_coresident = 1
    regs = [LoamRegister(params=prm) for _ in range(3)]
    poses = [w["init"].copy() for _ in regs]

    def run(i):
        for _ in range(4):
            poses[i][:] = w["init"]
            regs[i].scan2Map(w["scan"], w["map"], poses[i])

    th = [threading.Thread(target=run, args=(i,)) for i in range(len(regs))]
    for t in th: t.start()
    for t in th: t.join()
    for p in poses:
        np.testing.assert_array_equal(p, base)


def test_cache_is_exact_when_queries_cross_the_tile_boundary(gpu, world_100k):
    """With a query tile, a scan point is handled only while its transformed position lies inside the tile -- membership
    changes from iteration to iteration as the pose moves, and from call to call.  Whatever the neighbour cache holds for a
    point that was outside must never be used when it comes back: with and without the cache the iterations agree."""
    from simpleslam_amd import pcr
    w = world_100k
    q = w["scan"][:, :3] @ w["init"][:3, :3].T + w["init"][:3, 3]
    lo = np.array([-1e9, -1e9, -1e9]); hi = np.array([np.median(q[:, 0]), 1e9, 1e9])     # half of the scan, boundary through its middle
    out = []
    for disable_cache in (0, 1):
        prm = pcr.default_params(loam_iters=6, loam_early_exit=0, record_trace=1)
        prm.loam_disable_cache = disable_cache
        reg = LoamRegister(params=prm)
        reg.set_query_tile(lo, hi)
        # a first call with another scan and pose leaves entries of a different registration behind
        other = w["init"].copy(); other[0, 3] += 0.7
        reg.scan2Map(w["scan"][::-1].copy(), w["map"], other)
        pose = w["init"].copy()
        reg.scan2Map(w["scan"], w["map"], pose)
        out.append((pose, reg.trace()))
    np.testing.assert_array_equal(out[0][0], out[1][0])
    np.testing.assert_array_equal(out[0][1]["n"], out[1][1]["n"])
    np.testing.assert_array_equal(out[0][1]["JtJ"], out[1][1]["JtJ"])
    assert out[0][1]["cache_hits"].sum() > 0


def test_degenerate_geometry_matches_oracle(gpu, world_small):
    """Geometry that does not constrain all six degrees of freedom.  With a single plane (or a line of points) J^T J is
    numerically singular: Eigen's pivoted LDLT divides by pivots of ~1e-27 and the reference jumps ~1e12 m along the free
    directions.  What can be asked of a replacement is the SAME first step (later iterations are chaos in float); so the
    solve falls back to a step-for-step LDLT when a pivot collapses, and this test pins it.  A map 20 km from the origin
    (float coordinates carry ~2 mm there) is well-posed and must agree all the way."""
    w = world_small
    rng = np.random.default_rng(3)
    plane = np.zeros((20000, 4), np.float32)
    plane[:, 0] = rng.uniform(-20, 20, 20000); plane[:, 1] = rng.uniform(-20, 20, 20000); plane[:, 2] = -1.5
    line = np.zeros((5000, 4), np.float32)
    line[:, 0] = np.linspace(-30, 30, 5000); line[:, 2] = -1.0
    for name, m in (("single plane", plane), ("collinear points", line)):
        reg = LoamRegister(loam_iters=1, loam_early_exit=0, record_trace=1)
        pose = np.eye(4)
        conv = reg.scan2Map(w["scan"], m, pose)
        po, co, info = oracle.loam_scan2map(w["scan"], m, np.eye(4), oracle.loam_params(iters=1, early_exit=0), trace=True)
        tr = reg.trace()
        assert conv == co and tr["iters_run"] == info["iters_run"], name
        np.testing.assert_array_equal(tr["n"], info["n"][: tr["iters_run"]], err_msg=name)
        if tr["iters_run"]:
            # huge, but the same (the pose itself is exp of a rotation of ~1e11 rad: not comparable)
            np.testing.assert_allclose(tr["x"][0], info["x"][0], rtol=1e-6, atol=1e-9, err_msg=name)
    far_map = w["map"].copy(); far_map[:, 0] += 20000.0
    T = w["init"].copy(); T[0, 3] += 20000.0
    reg = LoamRegister(record_trace=1)
    pose = T.copy()
    conv = reg.scan2Map(w["scan"], far_map, pose)
    po, co, info = oracle.loam_scan2map(w["scan"], far_map, T, trace=True)
    tr = reg.trace()
    assert conv == co and tr["iters_run"] == info["iters_run"]
    np.testing.assert_array_equal(tr["n"], info["n"][: tr["iters_run"]])
    dt, dr = synth.pose_error(pose, po)
    assert dt < 1e-6 and dr < 1e-8


def test_handles_release_their_device_memory(gpu, world_small):
    """Create, use and destroy handles of every kind many times: the free device memory comes back (no leak in the buffers,
    streams, events and host-mapped blocks a handle owns)."""
    import gc
    import torch
    from simpleslam_amd import NdtRegister, ScanContext, SubMap, VgicpRegister
    w = world_small

    def cycle():
        for mk in (LoamRegister, NdtRegister, VgicpRegister):
            reg = mk()
            p = w["init"].copy()
            reg.scan2Map(w["scan"], w["map"], p)
            reg.voxelDownSample(w["scan"], 0.4)
            del reg
        sm = SubMap(); sm.addKeyFrame(w["scan"], w["truth"]); sm.updateMap(w["truth"][:3, 3]); del sm
        sc = ScanContext(); sc.addContext(w["scan"]); del sc
        gc.collect()

    for _ in range(3):
        cycle()                                                    # allocator pools, code objects, lazy runtime state
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(25):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 20, f"{(free0 - free1) >> 20} MiB of device memory not returned after 25 create/use/destroy cycles"


def test_far_outlier_in_the_target(gpu, world_small):
    """One stray point kilometres (or 3e38 m) away makes the bounding box too large for dense cell tables.  scan2Map then
    indexes the part of the target around the scan (it only ever looks one gate radius around a query) and returns what
    the oracle returns on the same cloud; the cut index is not offered to a later align().  setTarget has no scan to cut the
    box around: it indexes the bulk of the cloud (percentiles of a sample, generously padded) and align() on it returns the
    oracle's pose as well -- a query that reached a cut face would be noticed, not silently short of neighbours."""
    from simpleslam_amd import PcrError
    w = world_small
    clean = LoamRegister()
    p0 = w["init"].copy()
    c0 = clean.scan2Map(w["scan"], w["map"], p0)
    for dist in (2.0e4, 1.0e6, 3.0e38):
        m = w["map"].copy()
        m[0, :3] = [dist, -dist / 2, dist / 7]
        po, co, _ = oracle.loam_scan2map(w["scan"], m, w["init"])
        reg = LoamRegister()
        p = w["init"].copy()
        assert reg.scan2Map(w["scan"], m, p) == co
        dt, dr = synth.pose_error(p, po)
        assert dt < 1e-9 and dr < 1e-9, (dist, dt, dr)
        with pytest.raises(PcrError):
            reg.align(w["scan"], w["init"].copy())                 # the cut index is private to that call
        reg.setTarget(m)                                           # the bulk of the cloud is indexed, the stray point left out
        for _ in range(2):
            p3 = w["init"].copy()
            assert reg.align(w["scan"], p3) == co
            dt, dr = synth.pose_error(p3, po)
            assert dt < 1e-9 and dr < 1e-9, (dist, dt, dr)
        p2 = w["init"].copy()                                      # and the handle is as good as new on an ordinary target
        assert reg.scan2Map(w["scan"], w["map"], p2) == c0
        np.testing.assert_array_equal(p2, p0)


def test_cut_index_is_widened_when_a_query_reaches_its_edge(gpu, world_small):
    """The region a too-sparse target is cut to around the scan must never change the answer: the kernels count the queries
    that come within a cell of a cut face and the call is redone on a wider region until none does.  With the first margin
    set to -2.5 m (pcr_params.loam_clamp_margin_mm, a test hook; default +10 m: the region then cuts INTO the scan's box) the first
    attempts lose neighbours at the edge of the region and are thrown away; the pose that comes back is the one the full
    index gives, bit for bit."""
    from simpleslam_amd import pcr
    w = world_small
    m = w["map"].copy()
    m[0, :3] = [1.0e6, -5.0e5, 1.0e5]
    ref = w["init"].copy()
    c_ref = LoamRegister().scan2Map(w["scan"], m, ref)             # default margin: nothing reaches the edge
    prm = pcr.default_params()
    prm.loam_clamp_margin_mm = -2500
    reg = LoamRegister(params=prm)
    p = w["init"].copy()
    assert reg.scan2Map(w["scan"], m, p) == c_ref
    assert reg.stats()["attempts"] >= 4                            # overflow -> cut at -2.5 m -> widened at least twice
    np.testing.assert_array_equal(p, ref)
    po, co, _ = oracle.loam_scan2map(w["scan"], m, w["init"])
    dt, dr = synth.pose_error(p, po)
    assert co == c_ref and dt < 1e-9 and dr < 1e-9


def test_bounding_box_of_the_previous_target_is_a_hint_not_a_promise(gpu, world_small):
    """scan2Map rebuilds the index on every call (LoamRegister.cpp:110) but first tries the previous target's bounding box
    instead of measuring the new cloud's (one pass over the map saved).  Every point is checked against that box while it is
    binned; a target that has grown past it is indexed again with a box of its own -- the answer is that of a fresh handle."""
    w = world_small
    reg = LoamRegister()
    fresh = lambda m, p0: (lambda r, p: (r.scan2Map(w["scan"], m, p), p))(LoamRegister(), p0.copy())
    c0, p_ref = fresh(w["map"], w["init"])
    for _ in range(3):                                             # same target again and again: the hint holds
        p = w["init"].copy()
        assert reg.scan2Map(w["scan"], w["map"], p) == c0
        np.testing.assert_array_equal(p, p_ref)
    assert reg.stats()["attempts"] == 1
    # the target moves / grows out of the old box: points 40 m beyond it in x and y, and some below it
    extra = w["map"][:2000].copy()
    extra[:, 0] += w["map"][:, 0].max() - w["map"][:, 0].min() + 40.0
    extra[:1000, 1] -= 55.0
    extra[1000:, 2] -= 9.0
    grown = np.ascontiguousarray(np.vstack([w["map"], extra]))
    c1, p1_ref = fresh(grown, w["init"])
    p = w["init"].copy()
    assert reg.scan2Map(w["scan"], grown, p) == c1
    assert reg.stats()["attempts"] == 2                           # the stale box was noticed and the call redone
    np.testing.assert_array_equal(p, p1_ref)
    p = w["init"].copy()
    assert reg.scan2Map(w["scan"], grown, p) == c1
    assert reg.stats()["attempts"] == 1
    np.testing.assert_array_equal(p, p1_ref)
    # back to the smaller cloud: it fits the larger box, which only leaves cells empty
    p = w["init"].copy()
    assert reg.scan2Map(w["scan"], w["map"], p) == c0
    assert reg.stats()["attempts"] == 1
    np.testing.assert_array_equal(p, p_ref)
    # an empty target inside a hinted box is still an empty target
    p = w["init"].copy()
    assert reg.scan2Map(w["scan"], np.zeros((0, 4), np.float32), p) is False


def test_tile_layout_of_the_previous_target_is_a_hint_too(gpu, world_small):
    """A build that reuses the previous bounding box also reuses the previous PLACES of the tiles (each with an eighth of slack) and
    moves the points there while it bins them, which saves the placing pass.  A target whose points have piled up in one region --
    same box, so the box hint holds -- outgrows the room of those tiles: nothing is written out of bounds, the build is flagged
    and redone without hints, and the answer is that of a fresh handle."""
    w = world_small
    m = w["map"]
    reg = LoamRegister()
    fresh = lambda mm, p0: (lambda r, p: (r.scan2Map(w["scan"], mm, p), p))(LoamRegister(), p0.copy())
    for _ in range(3):                                             # box hint confirmed, layout written and used
        p = w["init"].copy(); reg.scan2Map(w["scan"], m, p)
    assert reg.stats()["attempts"] == 1
    # same extent, but the tenth of the map nearest to the scan's start position four times over (jittered by a millimetre: no duplicates)
    c = w["init"][:3, 3]
    near = m[np.argsort(np.linalg.norm(m[:, :3] - c, axis=1))[:m.shape[0] // 10]]
    rng = np.random.default_rng(7)
    piles = [near.copy() for _ in range(3)]
    for q in piles:
        q[:, :3] += rng.uniform(-1e-3, 1e-3, size=(q.shape[0], 3)).astype(np.float32)
    piled = np.ascontiguousarray(np.vstack([m] + piles))
    lo, hi = m[:, :3].min(0), m[:, :3].max(0)
    piled = piled[((piled[:, :3] >= lo) & (piled[:, :3] <= hi)).all(1)]           # the old box still holds every point
    c1, p1_ref = fresh(piled, w["init"])
    p = w["init"].copy()
    assert reg.scan2Map(w["scan"], piled, p) == c1
    assert reg.stats()["attempts"] == 2                           # the tiles around the scan had no room: noticed, redone
    np.testing.assert_array_equal(p, p1_ref)
    p = w["init"].copy()
    assert reg.scan2Map(w["scan"], piled, p) == c1                 # the new layout holds from now on
    assert reg.stats()["attempts"] == 1
    np.testing.assert_array_equal(p, p1_ref)
    # and back: the thinner cloud fits the roomier layout
    c0, p_ref = fresh(m, w["init"])
    p = w["init"].copy()
    assert reg.scan2Map(w["scan"], m, p) == c0
    assert reg.stats()["attempts"] == 1
    np.testing.assert_array_equal(p, p_ref)


def test_static_target_cut_to_its_bulk_still_serves_a_scan_in_the_part_left_out(gpu, world_small):
    """setTarget on a cloud too spread out for dense cell tables indexes the bulk of it.  Here the cloud is the map plus a copy of a
    two-hundredth of it a few kilometres away in every direction (5e9 cells of 1 m) -- outside the bulk box -- and the scan to align lies in that far copy: every query falls beyond a cut
    face, which is noticed, and the call is redone on a region cut around the scan.  Same pose as the oracle on the same cloud."""
    w = world_small
    off = np.array([3000.0, 2000.0, 800.0], np.float32)
    c = w["init"][:3, 3]
    near = w["map"][np.argsort(np.linalg.norm(w["map"][:, :3] - c, axis=1))[: w["map"].shape[0] // 200]].copy()
    far = near.copy(); far[:, :3] += off
    both = np.ascontiguousarray(np.vstack([w["map"], far]))
    scan_far_init = w["init"].copy(); scan_far_init[:3, 3] += off          # the same scan, placed in the far copy
    reg = LoamRegister()
    reg.setTarget(both)
    p = w["init"].copy()                                                    # a scan in the bulk: served by the bulk index
    conv = reg.align(w["scan"], p)
    po, co, _ = oracle.loam_scan2map(w["scan"], both, w["init"])
    assert conv == co
    dt, dr = synth.pose_error(p, po)
    assert dt < 1e-9 and dr < 1e-9, (dt, dr)
    assert reg.stats()["attempts"] == 1
    p = scan_far_init.copy()                                                # a scan in the part that was left out
    conv = reg.align(w["scan"], p)
    po, co, _ = oracle.loam_scan2map(w["scan"], both, scan_far_init)
    assert conv == co
    dt, dr = synth.pose_error(p, po)
    assert dt < 1e-7 and dr < 1e-9, (dt, dr)                                # (lever arms of 3.7 km: the normal equations amplify the order of summation)
    assert reg.stats()["attempts"] >= 2


def test_one_handle_through_a_sequence_of_unrelated_targets(gpu):
    """Every call rebuilds the index, but with what the previous call left behind: its bounding box, the places of its tiles, its cell
    count.  None of that may ever show in a result.  One handle is driven through targets that grow, shrink, move, thin out and pile up;
    after every call its pose must equal, bit for bit, that of a handle that has never seen another target."""
    rng = np.random.default_rng(2026)
    reg = LoamRegister()
    world, base = synth.make_map(120_000, seed=11)
    scan, T = synth.make_scan(world, 0, seed=11, beams=16, azimuths=512)
    retries = 0
    for case in range(24):
        kind = case % 8
        m = base
        if kind == 1: m = base[:: int(rng.integers(2, 40))]                                   # thinned out
        elif kind == 2: m = np.vstack([base, base[: 20_000] + np.array([0.003, 0.002, 0.001, 0], np.float32)])   # piled up
        elif kind == 3: m = base + np.array([float(rng.uniform(-30, 30)), float(rng.uniform(-30, 30)), 0, 0], np.float32)   # moved
        elif kind == 4: m = base[np.abs(base[:, 0] - T[0, 3]) < float(rng.uniform(8, 25))]    # a slab around the scan
        elif kind == 5: m = np.vstack([base, base[: 500] + np.array([150.0, -90.0, 12.0, 0], np.float32)])          # box grows
        elif kind == 6: m = base[: int(rng.integers(50, 2000))]                               # a handful of points
        elif kind == 7: m = np.vstack([base, base[::3] + np.array([0.0, 0.0, 0.004, 0], np.float32)])               # denser everywhere
        m = np.ascontiguousarray(m, np.float32)
        init = synth.perturb(T, 100 + case)
        p_fresh, p_used = init.copy(), init.copy()
        c_fresh = LoamRegister().scan2Map(scan, m, p_fresh)
        c_used = reg.scan2Map(scan, m, p_used)
        assert c_used == c_fresh, (case, kind)
        np.testing.assert_array_equal(p_used, p_fresh, err_msg=f"case {case} kind {kind}")
        retries += reg.stats()["attempts"] - 1
    assert retries >= 3            # (the sequence did make hints fail: grown boxes, tiles without room)
    # an empty target (twice: a hint is only trusted once the host has seen it hold) must not leave its header behind as a hint either
    nothing = np.zeros((0, 4), np.float32)
    nowhere = np.full((50, 4), np.nan, np.float32)
    for m in (nothing, nothing, base, nowhere, nowhere, base):
        init = synth.perturb(T, 999)
        p_fresh, p_used = init.copy(), init.copy()
        c_fresh = LoamRegister().scan2Map(scan, m, p_fresh)
        c_used = reg.scan2Map(scan, m, p_used)
        assert c_used == c_fresh
        np.testing.assert_array_equal(p_used, p_fresh)

"""GPU parity of the VGICP scan-to-map path (HIP through the C ABI) vs the CPU oracle."""
import numpy as np
import pytest

import oracle
from simpleslam_amd import VgicpRegister, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def vg_world():
    world, m = synth.make_map(60_000, seed=21)
    scan, T = synth.make_scan(world, 0, seed=21, beams=32, azimuths=512)
    return dict(map=m, scan=scan, truth=T, init=synth.perturb(T, 21, trans=0.3, rot_deg=2.0))


def test_covariances_match_oracle(gpu, vg_world):
    w = vg_world
    reg = VgicpRegister()
    g = reg.covariances(w["scan"])
    o = oracle.vgicp_covariances(w["scan"], 20, 8)
    # identical neighbour sets (float distances, index ties) -> agreement to rounding of the 3x3 eigen solve
    bad = np.abs(g - o).max(axis=(1, 2)) > 1e-9
    assert bad.mean() < 1e-3, bad.sum()
    # PLANE regularisation: eigenvalues (1, 1, 1e-3)
    ev = np.linalg.eigvalsh(g[::97])
    np.testing.assert_allclose(ev, np.tile([1e-3, 1.0, 1.0], (ev.shape[0], 1)), atol=1e-9)


def _check_neighbours(scan):
    """neighbour lists of covariances(scan) == the oracle's float k-NN, index for index (ties on the lower index)"""
    reg = VgicpRegister()
    reg.covariances(scan)
    nb, queued = reg.neighbours(len(scan))
    k = min(20, len(scan))
    ref, _ = oracle.knn_f32(scan, scan[:, :3], 20)
    want = np.where(ref < 0, 0xFFFFFFFF, ref).astype(np.uint32)
    assert (nb[:, :k] == want[:, :k]).all(), int((nb[:, :k] != want[:, :k]).any(axis=1).sum())
    assert (nb[:, k:] == 0xFFFFFFFF).all()
    return queued


def test_neighbour_lists_match_oracle_index_for_index(gpu, vg_world):
    """fast_gicp_impl.hpp:250-253: the 20 nearest neighbours of every scan point.  Both classes of queries of csrc/cov_search.hip
    (lane-per-query ring 1, wave-per-query rings beyond) must be exercised by a lidar scan."""
    scan = vg_world["scan"]
    queued = _check_neighbours(scan)
    assert 0 < queued < len(scan)


def test_neighbour_lists_on_a_lattice_with_exact_ties(gpu):
    """Coordinates on a 1/8 m lattice: many exactly equal distances at the edge of a list -- the screening keys of the lane-per-query
    kernel cannot prove those lists and hands them to the exact wave-per-query search; duplicated points tie on everything but the index."""
    rng = np.random.default_rng(5)
    pts = np.zeros((6000, 4), np.float32)
    pts[:, :3] = rng.integers(-40, 40, (6000, 3)) / 8.0
    pts[:, 2] = rng.integers(0, 3, 6000) / 8.0
    pts[100:140] = pts[200:240]
    queued = _check_neighbours(pts)
    assert queued > 0


def test_neighbour_lists_of_tiny_and_sparse_clouds(gpu):
    """fewer points than neighbours (unfilled slots), one far straggler (rings of the coarse level), a dense clump (thousands of candidates)"""
    rng = np.random.default_rng(6)
    tiny = np.zeros((7, 4), np.float32); tiny[:, :3] = rng.normal(0, 1, (7, 3))
    _check_neighbours(tiny)
    sparse = np.zeros((400, 4), np.float32); sparse[:, :3] = rng.uniform(-60, 60, (400, 3)); sparse[0, :3] = (400.0, 300.0, 20.0)
    _check_neighbours(sparse)
    clump = np.zeros((12000, 4), np.float32); clump[:, :3] = rng.normal(0, 0.15, (12000, 3)); clump[:2000, :3] = rng.uniform(-30, 30, (2000, 3))
    _check_neighbours(clump)


def test_linearize_matches_oracle(gpu, vg_world):
    w = vg_world
    reg = VgicpRegister()
    reg.setTarget(w["map"])
    sc = oracle.vgicp_covariances(w["scan"], 20, 8)
    dc = oracle.vgicp_covariances(w["map"], 20, 8)
    for pose in (w["init"], w["truth"]):
        g = reg.linearize(w["scan"], pose)
        o = oracle.vgicp_linearize(w["scan"], w["map"], pose, sc, dc)
        assert g["n"] == o["n"] and g["n"] > 2000
        np.testing.assert_allclose(g["H"], o["H"], rtol=1e-7, atol=1e-6)
        np.testing.assert_allclose(g["b"], o["b"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(g["err"], o["err"], rtol=1e-8)


def test_scan2map_matches_oracle(gpu, vg_world):
    w = vg_world
    reg = VgicpRegister()
    pose = w["init"].copy()
    conv = reg.scan2Map(w["scan"], w["map"], pose)
    po, co, info = oracle.vgicp_scan2map(w["scan"], w["map"], w["init"], oracle.vgicp_params(threads=8))
    assert conv == co
    assert reg.stats()["iterations"] == info["outer"]
    dt, dr = synth.pose_error(pose, po)
    assert dt <= 1e-4 and dr <= 1e-4, (dt, dr)
    et, er = synth.pose_error(pose, w["truth"])
    assert et < 0.05 and er < 5e-3
    # final pose went through Matrix4f (VgicpRegister.cpp:37)
    np.testing.assert_array_equal(pose, pose.astype(np.float32).astype(np.float64))
    # fitness score (pcl::Registration::getFitnessScore)
    np.testing.assert_allclose(reg.getFitnessScore(), oracle.fitness_score(w["scan"], w["map"], pose), rtol=1e-6)


def test_device_resident_optimiser_equals_the_host_driven_one(gpu, vg_world):
    """The Levenberg-Marquardt loop is one state machine (csrc/vgicp_opt.h) that runs on the device by default -- the fold of a pass's
    sums and the optimiser's step are the prologue of the next pass's launch, no host round trip in between -- and as the reference's
    loops on the host for sharded targets (pcr_params.host_optimiser = 1 selects that here).  Same decisions (convergence flag, outer
    iterations, passes), and the same Matrix4f pose up to the last bit of the two libms' sine in so3_exp."""
    from simpleslam_amd.pcr import default_params
    w = vg_world
    p_host = default_params()
    p_host.host_optimiser = 1
    dev, host = VgicpRegister(), VgicpRegister(params=p_host)
    for seed, tr, rd in ((41, 0.3, 2.0), (42, 0.1, 0.5), (43, 0.6, 4.0), (44, 0.0, 0.0), (45, 1.5, 8.0)):
        T0 = synth.perturb(w["truth"], seed, trans=tr, rot_deg=rd) if tr else w["truth"].copy()
        pd, ph = T0.copy(), T0.copy()
        cd = dev.scan2Map(w["scan"], w["map"], pd)
        ch = host.scan2Map(w["scan"], w["map"], ph)
        assert cd == ch, seed
        sd, sh = dev.stats(), host.stats()
        assert (sd["iterations"], sd["kernel_launches"]) == (sh["iterations"], sh["kernel_launches"]), (seed, sd, sh)
        dt, dr = synth.pose_error(pd, ph)
        assert dt <= 2e-6 and dr <= 2e-6, (seed, dt, dr)      # (a float ulp of the pose at 10 m is 1e-6)
        np.testing.assert_allclose(dev.getFitnessScore(), host.getFitnessScore(), rtol=1e-5)
    # many calls on one handle: launches enqueued beyond the end of one alignment must not leak into the next
    first = w["init"].copy(); dev.scan2Map(w["scan"], w["map"], first)
    for _ in range(20):
        p = w["init"].copy(); dev.scan2Map(w["scan"], w["map"], p)
        np.testing.assert_array_equal(p, first)
    # an iteration cap that ends the loop in the middle of the search
    from simpleslam_amd.pcr import default_params as dp
    for cap in (1, 2, 3):
        pa, pb = dp(), dp()
        pa.vgicp_max_iters = cap; pb.vgicp_max_iters = cap; pb.host_optimiser = 1
        ra, rb = VgicpRegister(params=pa), VgicpRegister(params=pb)
        qa, qb = w["init"].copy(), w["init"].copy()
        assert ra.scan2Map(w["scan"], w["map"], qa) == rb.scan2Map(w["scan"], w["map"], qb)
        assert ra.stats()["iterations"] == rb.stats()["iterations"] == cap
        dt, dr = synth.pose_error(qa, qb)
        assert dt <= 2e-6 and dr <= 2e-6, (cap, dt, dr)


def test_half_metre_voxels(gpu, vg_world):
    """BASELINE config 3 asks for 0.5 m voxels (the reference hard-codes 1.0: SURVEY.md F9)."""
    w = vg_world
    reg = VgicpRegister(vgicp_resolution=0.5)
    pose = w["init"].copy()
    conv = reg.scan2Map(w["scan"], w["map"], pose)
    po, co, info = oracle.vgicp_scan2map(w["scan"], w["map"], w["init"], oracle.vgicp_params(threads=8, resolution=0.5))
    assert conv == co
    dt, dr = synth.pose_error(pose, po)
    assert dt <= 1e-4 and dr <= 1e-4, (dt, dr)


def test_static_target_reuse(gpu, vg_world):
    w = vg_world
    reg = VgicpRegister()
    p1, p2 = w["init"].copy(), w["init"].copy()
    c1 = reg.scan2Map(w["scan"], w["map"], p1)
    reg.setTarget(w["map"])
    c2 = reg.align(w["scan"], p2)
    assert c1 == c2
    np.testing.assert_array_equal(p1, p2)


def test_golden_fixture(gpu):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "vgicp_small.npz"))
    reg = VgicpRegister()
    np.testing.assert_allclose(reg.covariances(g["scan"])[::16], g["src_cov"], atol=1e-9)
    reg.setTarget(g["map"])
    lin = reg.linearize(g["scan"], g["init"])
    assert lin["n"] == int(g["n_corr"])
    np.testing.assert_allclose(lin["H"], g["H"], rtol=1e-7, atol=1e-6)
    np.testing.assert_allclose(lin["err"], float(g["err"]), rtol=1e-8)
    pose = g["init"].copy()
    conv = reg.scan2Map(g["scan"], g["map"], pose)
    assert conv == bool(g["converged"]) and reg.stats()["iterations"] == int(g["outer"])
    dt, dr = synth.pose_error(pose, g["pose"])
    assert dt <= 1e-4 and dr <= 1e-4
    np.testing.assert_allclose(reg.getFitnessScore(), float(g["fitness"]), rtol=1e-5)


def test_randomised_configurations(gpu):
    """Seeded draws of world, scan, initial error and voxel resolution: same convergence flag, same number of outer
    iterations, same pose (both end in a Matrix4f) as the oracle."""
    rng = np.random.default_rng(777)
    for case in range(5):
        world, m = synth.make_map(int(rng.integers(15_000, 50_000)), seed=3000 + case)
        scan, T = synth.make_scan(world, int(rng.integers(0, 4)), seed=3000 + case, beams=int(rng.choice([16, 32])), azimuths=256)
        init = synth.perturb(T, 4000 + case, trans=float(rng.uniform(0.05, 0.3)), rot_deg=float(rng.uniform(0.2, 2.0)))
        res = float(rng.choice([0.5, 1.0, 1.5]))
        reg = VgicpRegister(vgicp_resolution=res)
        pose = init.copy()
        conv = reg.scan2Map(scan, m, pose)
        po, co, info = oracle.vgicp_scan2map(scan, m, init, oracle.vgicp_params(resolution=res, threads=8))
        assert conv == co, (case, res)
        assert reg.stats()["iterations"] == info["outer"], (case, res)
        dt, dr = synth.pose_error(pose, po)
        assert dt <= 1e-4 and dr <= 1e-4, (case, res, dt, dr)


def test_edge_cases_match_oracle(gpu, vg_world):
    """A scan that misses the voxel map entirely (no correspondence), an empty scan, a scan point at the sensor origin."""
    w = vg_world
    far = w["scan"].copy(); far[:, :3] += 5000.0
    scan0 = w["scan"].copy(); scan0[0, :3] = 0.0
    for name, scan in (("no correspondences", far), ("empty scan", w["scan"][:0]), ("origin point", scan0)):
        po, co, info = oracle.vgicp_scan2map(scan, w["map"], w["init"], oracle.vgicp_params(threads=8))
        reg = VgicpRegister()
        pose = w["init"].copy()
        conv = reg.scan2Map(scan, w["map"], pose)
        assert conv == co, name
        assert reg.stats()["iterations"] == info["outer"], name
        assert np.isfinite(pose).all() == np.isfinite(po).all(), name
        if np.isfinite(po).all():
            dt, dr = synth.pose_error(pose, po)
            assert dt <= 1e-4 and dr <= 1e-4, (name, dt, dr)


def test_gated_fitness_of_align_cpp(gpu, vg_world):
    """test/align.cpp:29-61: mean squared 1-NN distance over the source points within 1 m of the target (float distances,
    transform in float), -1 when none -- for every method's handle, against the target of its last registration."""
    from simpleslam_amd import LoamRegister, NdtRegister
    w = vg_world
    T, gate = w["init"], 0.02                                     # 0.3 m / 2 deg off: part of the scan is farther than the gate
    want = oracle.fitness_score(w["scan"], w["map"], T, gate)
    src_t = (w["scan"][:, :3] @ T[:3, :3].astype(np.float32).T + T[:3, 3].astype(np.float32)).astype(np.float32)
    _, d2 = oracle.knn_f32(w["map"], src_t, 1)
    n_want = int((d2[:, 0] <= np.float32(gate)).sum())
    assert 0 < n_want < w["scan"].shape[0]                      # the gate does cut something off
    for make in (lambda: VgicpRegister(), lambda: LoamRegister(), lambda: NdtRegister()):
        reg = make()
        pose = w["init"].copy()
        reg.scan2Map(w["scan"], w["map"], pose)                  # align.cpp:144 ... then :150 scores against the same target
        pose = w["init"].copy()
        reg.scan2Map(w["scan"], w["map"], pose)                  # (a handle's later calls may index the scan's region only: the score must not care)
        got, n_in = reg.fitnessGated(w["scan"], T, gate)
        assert n_in == n_want
        np.testing.assert_allclose(got, want, rtol=1e-6)
        far = np.eye(4); far[:3, 3] = [0.0, 0.0, 500.0]
        assert reg.fitnessGated(w["scan"], far, 1.0) == (-1.0, 0)


def test_far_outlier_in_the_target(gpu, vg_world):
    """One stray point kilometres (or 3e38 m) away makes the voxel lattice too large for dense tables; the reference's hash map and
    kd-tree do not care.  The target is then indexed over the bulk of the cloud (percentiles of a sample, generously padded): the
    stray point is in nobody's 20-neighbourhood and in no voxel the scan visits, so the pose is the oracle's on the same cloud.  A scan
    that does go to the part left out gets the target indexed around ITSELF (its box + the reach of a covariance + room to move) and the
    alignment repeated -- the oracle's pose there too, never a pose of a truncated map."""
    from simpleslam_amd import PcrError
    w = vg_world
    for dist in (2.0e4, 3.0e38):
        m = w["map"].copy()
        m[0, :3] = [dist, -dist / 2, dist / 7]
        po, co, info = oracle.vgicp_scan2map(w["scan"], m, w["init"], oracle.vgicp_params(threads=8))
        reg = VgicpRegister()
        for via in ("scan2Map", "align"):
            pose = w["init"].copy()
            if via == "scan2Map":
                conv = reg.scan2Map(w["scan"], m, pose)
            else:
                reg.setTarget(m)
                conv = reg.align(w["scan"], pose)
            assert conv == co, (dist, via)
            assert reg.stats()["iterations"] == info["outer"], (dist, via)
            dt, dr = synth.pose_error(pose, po)
            assert dt <= 1e-4 and dr <= 1e-4, (dist, via, dt, dr)
    # a second cluster a few kilometres off in every direction, and a scan placed in it
    off = np.array([30000.0, 20000.0, 8000.0], np.float32)
    far = w["map"][:: 200].copy(); far[:, :3] += off
    both = np.ascontiguousarray(np.vstack([w["map"], far]))
    there = w["init"].copy(); there[:3, 3] += off
    reg = VgicpRegister()
    # the scan in the far cluster: the bulk cut leaves that cluster out, the scan reaches the cut -> the target is indexed around the
    # scan instead, and the pose is the oracle's (the reference's hash map serves any extent: fast_vgicp_voxel.hpp:129-156)
    pt, ct, it = oracle.vgicp_scan2map(w["scan"], both, there, oracle.vgicp_params(threads=8))
    for via in ("scan2Map", "align"):
        pose = there.copy()
        if via == "scan2Map":
            conv = reg.scan2Map(w["scan"], both, pose)
        else:
            reg.setTarget(both)
            conv = reg.align(w["scan"], pose)
        assert conv == ct, via
        dt, dr = synth.pose_error(pose, pt)
        assert dt <= 1e-4 and dr <= 1e-4, (via, dt, dr)
    pose = w["init"].copy()                                        # the same handle and cloud, a scan in the bulk: served as well
    po, co, _ = oracle.vgicp_scan2map(w["scan"], both, w["init"], oracle.vgicp_params(threads=8))
    assert reg.scan2Map(w["scan"], both, pose) == co
    dt, dr = synth.pose_error(pose, po)
    assert dt <= 1e-4 and dr <= 1e-4


def test_one_handle_through_a_sequence_of_unrelated_targets(gpu, vg_world):
    """The index of every build takes its tile size and its cell bound from the previous build of the same handle.  Driven through
    targets and scans of very different extent and density, a handle must return what a fresh one returns, bit for bit."""
    w = vg_world
    rng = np.random.default_rng(3)
    base = w["map"]
    reg = VgicpRegister()
    for case in range(10):
        kind = case % 5
        m, scan = base, w["scan"]
        if kind == 1: m = base[:: int(rng.integers(3, 12))]
        elif kind == 2: m = np.vstack([base, base[: 3000] + np.array([120.0, 80.0, 15.0, 0], np.float32)])
        elif kind == 3: scan = w["scan"][:: int(rng.integers(2, 9))]
        elif kind == 4: m = base[: int(rng.integers(2000, 20000))]
        m, scan = np.ascontiguousarray(m, np.float32), np.ascontiguousarray(scan, np.float32)
        T0 = synth.perturb(w["truth"], 70 + case, trans=0.2, rot_deg=1.0)
        p_fresh, p_used = T0.copy(), T0.copy()
        c_fresh = VgicpRegister().scan2Map(scan, m, p_fresh)
        c_used = reg.scan2Map(scan, m, p_used)
        assert c_used == c_fresh, (case, kind)
        np.testing.assert_array_equal(p_used, p_fresh, err_msg=f"case {case} kind {kind}")


@pytest.mark.parametrize("off", [0.25, 4.0])
def test_target_prepared_for_the_scans_region_only(gpu, vg_world, off):
    """pcr_scan2map prepares covariances and voxels only where the scan can land (a 2 m margin around its points at the initial pose); a
    pose that carries the scan out of that region makes the call repeat on the whole target.  Either way: the pose of the full
    preparation (pcr_params.full_target = 1, what the reference computes), bit for bit."""
    w = vg_world
    T0 = w["truth"].copy()
    T0[:3, 3] += np.array([off, -0.6 * off, 0.05 * off])
    full = VgicpRegister(full_target=1)
    pf = T0.copy(); cf = full.scan2Map(w["scan"], w["map"], pf)
    reg = VgicpRegister()
    p = T0.copy(); c = reg.scan2Map(w["scan"], w["map"], p)
    assert c == cf
    np.testing.assert_array_equal(p, pf)
    assert reg.stats()["iterations"] == full.stats()["iterations"]
    rep = reg.stats()["region_repeats"]
    assert full.stats()["region_repeats"] == 0
    assert rep == 0 if off < 1.0 else rep in (0, 1), rep      # 0.25 m stays inside the margin; a start 4 m off may end (or wander) outside it
    # The reference keeps its target after scan2Map (setInputTarget persists; test/align.cpp aligns and scores afterwards).  A host target lies in
    # the handle's staging copy: the next pcr_align finds it prepared in full -- never the partial preparation -- and equals setTarget + align.
    kept = VgicpRegister(); kept.setTarget(w["map"])
    pk = T0.copy(); ck = kept.align(w["scan"], pk)
    pa = T0.copy(); ca = reg.align(w["scan"], pa)
    assert ca == ck
    np.testing.assert_array_equal(pa, pk)
    lin = reg.linearize(w["scan"], w["truth"]); lin_k = kept.linearize(w["scan"], w["truth"])
    assert lin["n"] == lin_k["n"] and lin["err"] == lin_k["err"]


def test_region_list_of_a_map_sized_target_matches_the_whole_target(gpu):
    """A map-sized target (more than 300 000 points: one search level, the lane-per-query kernel) prepared for one scan: the sorted positions of the
    region's points are compacted first (vgicp_region_list_kernel) and the covariance kernel runs over that list.  Poses, verdicts and
    iteration counts are those of the full preparation (pcr_params.full_target = 1), bit for bit -- on every call of a handle (the list's
    two counters alternate) and from a start that leaves the region (repeat on the whole target)."""
    world, m = synth.make_map(400_000, seed=77)
    scan, T = synth.make_scan(world, 1, seed=77)
    full = VgicpRegister(full_target=1)
    reg = VgicpRegister()
    repeats, region_index = 0, 0
    for k, off in enumerate([0.2, -0.3, 0.1, 5.0, 0.25]):
        T0 = T.copy()
        T0[:3, 3] += np.array([off, -0.5 * off, 0.03 * off])
        pf = T0.copy(); cf = full.scan2Map(scan, m, pf)
        p = T0.copy(); c = reg.scan2Map(scan, m, p)
        assert c == cf, (k, off)
        np.testing.assert_array_equal(p, pf)
        assert reg.stats()["iterations"] == full.stats()["iterations"], (k, off)
        repeats += reg.stats()["region_repeats"]
        region_index += reg.stats()["region_index"]
        # the fitness score is a nearest-neighbour question about the WHOLE target: a lattice that holds the region's points only hands it to the
        # grid the covariances were searched on, which holds every point -- the same number as with the full preparation
        assert reg.getFitnessScore() == full.getFitnessScore(), (k, off)
        if k == 0:
            assert reg.stats()["region_index"] == 0      # nothing to go by yet: the whole cloud is indexed
    assert full.stats()["region_repeats"] == 0 and full.stats()["region_index"] == 0
    # from the second call on the voxel lattice itself holds the region's points only (round 5: BuildFilter, as NDT's)
    assert region_index >= 2


@pytest.mark.parametrize("method", ["vgicp", "ndt"])
def test_align_after_scan2map_of_a_device_target_says_what_to_do(gpu, vg_world, method):
    """a DEVICE target is the caller's buffer and may be gone after the call: the region-only preparation is not passed on to pcr_align, the
    message names the two ways out (pcr_set_target, pcr_params.full_target); with full_target = 1 the same sequence works"""
    import torch
    from simpleslam_amd import NdtRegister
    from simpleslam_amd.pcr import PcrError
    w = vg_world
    cls = VgicpRegister if method == "vgicp" else NdtRegister
    d_scan, d_map = torch.from_numpy(w["scan"]).cuda(), torch.from_numpy(w["map"]).cuda()
    reg = cls()
    pose = w["init"].copy(); reg.scan2Map(d_scan, d_map, pose)
    if reg.stats()["region_repeats"] == 0:
        with pytest.raises(PcrError, match="pcr_set_target"):
            reg.align(d_scan, w["init"].copy())
    full = cls(full_target=1)
    pose = w["init"].copy(); full.scan2Map(d_scan, d_map, pose)
    p2 = w["init"].copy(); full.align(d_scan, p2)
    np.testing.assert_array_equal(p2, pose)


def test_ndt_align_after_scan2map_of_a_host_target(gpu, vg_world):
    """NDT: second call of a handle (region-only INDEX) with host clouds, then pcr_align on the same target = set_target + align"""
    from simpleslam_amd import NdtRegister
    w = vg_world
    reg = NdtRegister()
    for _ in range(2):
        pose = w["init"].copy(); reg.scan2Map(w["scan"], w["map"], pose)
    kept = NdtRegister(); kept.setTarget(w["map"])
    pk = w["init"].copy(); ck = kept.align(w["scan"], pk)
    pa = w["init"].copy(); ca = reg.align(w["scan"], pa)
    assert ca == ck
    np.testing.assert_array_equal(pa, pk)
    np.testing.assert_array_equal(pa, pose)


def test_gated_fitness_against_a_region_only_index_of_a_device_target_is_refused(gpu, vg_world):
    """NDT, from the second pcr_scan2map of a handle on, indexes only the target points of the scan's region (pcr_stats.region_index).  A host
    target still lies in the handle's staging copy and is indexed again for the score (the test above); a DEVICE target is the caller's and
    may be gone: the call says so instead of searching an incomplete index."""
    import torch
    from simpleslam_amd import NdtRegister
    from simpleslam_amd.pcr import PcrError
    w = vg_world
    d_scan, d_map = torch.from_numpy(w["scan"]).cuda(), torch.from_numpy(w["map"]).cuda()
    reg = NdtRegister()
    for _ in range(2):
        pose = w["init"].copy()
        reg.scan2Map(d_scan, d_map, pose)
    if reg.stats()["region_index"]:
        with pytest.raises(PcrError, match="region only"):
            reg.fitnessGated(d_scan, w["init"], 1.0)
    full = NdtRegister(full_target=1)
    for _ in range(2):
        pose = w["init"].copy()
        full.scan2Map(d_scan, d_map, pose)
    assert full.stats()["region_index"] == 0
    full.fitnessGated(d_scan, w["init"], 1.0)


@pytest.mark.parametrize("method", ["vgicp", "ndt"])
def test_hints_of_an_earlier_target_that_do_not_hold_the_next_one(gpu, vg_world, method):
    """From its second pcr_scan2map on a handle builds its target grids by the previous call's box and tile layout, queues the region,
    the covariances and the voxels before it has seen a header (VGICP) and indexes only the region's points (NDT).  When the next target
    does not fit those hints -- the sub-map grew beyond the box, a tile outgrew its room, or it shrank -- the kernels queued ahead leave
    on the header's flag and the target is prepared again without hints: the result is a fresh handle's, bit for bit, for every call."""
    from simpleslam_amd import make_register
    w = vg_world
    m = w["map"]
    x = m[:, 0]
    lo, hi = np.percentile(x, 30), np.percentile(x, 70)
    targets = [m[(x > lo) & (x < hi)], m, m[x < hi], m[::3], m]      # grows past the box, shrinks, thins out, grows again
    reg = make_register(method)
    for k, tgt in enumerate(targets):
        tgt = np.ascontiguousarray(tgt)
        pose = w["init"].copy()
        conv = reg.scan2Map(w["scan"], tgt, pose)
        fresh = make_register(method)
        pf = w["init"].copy()
        cf = fresh.scan2Map(w["scan"], tgt, pf)
        assert conv == cf, (method, k)
        np.testing.assert_array_equal(pose, pf, err_msg=f"{method} call {k}")
        assert reg.stats()["iterations"] == fresh.stats()["iterations"]

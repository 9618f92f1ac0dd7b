"""GPU parity of the NDT scan-to-map path (HIP through the C ABI) vs the CPU oracle.
The reference's inner derivative math is float32 (ndt_omp_impl.hpp:485-537) and its final pose is a
Matrix4f; the device evaluates every float term as the oracle does (its expf is the C library's algorithm), so the poses
normally agree bit for bit -- the bar of BASELINE.json is 1e-4 m / 1e-4 rad."""
import numpy as np
import pytest
from scipy.spatial.transform import Rotation as Rot

import oracle
from simpleslam_amd import NdtRegister, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nd_world():
    world, m = synth.make_map(300_000, seed=31, spacing=0.2)      # NDT wants >= 6 points per 1 m voxel
    scan, T = synth.make_scan(world, 0, seed=31, beams=32, azimuths=512)
    return dict(map=m, scan=scan, truth=T)


def _p_of(T):
    return np.concatenate([T[:3, 3], Rot.from_matrix(T[:3, :3]).as_euler("XYZ")])


def test_derivatives_match_oracle(gpu, nd_world):
    w = nd_world
    reg = NdtRegister()
    reg.setTarget(w["map"])
    for T in (synth.perturb(w["truth"], 31, trans=0.1, rot_deg=0.5), w["truth"]):
        p = _p_of(T)
        g = reg.derivatives(w["scan"], p, double_hessian=True)
        o = oracle.ndt_derivatives(w["scan"], w["map"], p, double_hessian=True)
        # (the float terms are the oracle's bit for bit -- the exponential is evaluated the way the C library does it, ndt.hip: ndt_expf --
        #  what is left is the order of the double sums and the 1e-10 of the voxel Gaussians; 2e-5 before that)
        assert abs(g["score"] - o["score"]) <= 1e-9 * abs(o["score"])
        gs, hs = np.abs(o["grad"]).max(), np.abs(o["hess"]).max()
        assert np.abs(g["grad"] - o["grad"]).max() <= 1e-7 * gs
        assert np.abs(g["hess"] - o["hess"]).max() <= 1e-8 * hs
        assert np.abs(g["hess_d"] - o["hess_d"]).max() <= 1e-8 * hs     # double path: voxel Gaussians agree to ~1e-10


def test_scan2map_matches_oracle(gpu, nd_world):
    w = nd_world
    ok = 0
    for seed, tr, rd in ((31, 0.1, 0.5), (32, 0.15, 0.8), (33, 0.05, 0.3)):
        T0 = synth.perturb(w["truth"], seed, trans=tr, rot_deg=rd)
        po, co, info = oracle.ndt_scan2map(w["scan"], w["map"], T0)
        # (a collapsed line-search interval gives a NaN trial value that the reference's std::min/std::max drop,
        # ndt_omp_impl.hpp:758-761; the pose must come out finite on both sides)
        assert np.isfinite(po).all()
        reg = NdtRegister()
        pose = T0.copy()
        conv = reg.scan2Map(w["scan"], w["map"], pose)
        assert conv == co
        assert reg.stats()["iterations"] == info["iterations"]
        np.testing.assert_array_equal(pose, po)                                           # the same Matrix4f, bit for bit
        np.testing.assert_array_equal(pose, pose.astype(np.float32).astype(np.float64))   # Matrix4f result
        ok += 1
    assert ok == 3


def test_static_target_reuse(gpu, nd_world):
    w = nd_world
    T0 = synth.perturb(w["truth"], 31, trans=0.1, rot_deg=0.5)
    reg = NdtRegister()
    p1, p2 = T0.copy(), T0.copy()
    c1 = reg.scan2Map(w["scan"], w["map"], p1)
    reg.setTarget(w["map"])
    c2 = reg.align(w["scan"], p2)
    assert c1 == c2
    np.testing.assert_array_equal(p1, p2)


def test_golden_fixture(gpu):
    import os
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "ndt_small.npz"))
    reg = NdtRegister()
    reg.setTarget(g["map"])
    d = reg.derivatives(g["scan"], g["p6"], double_hessian=True)
    assert abs(d["score"] - float(g["score"])) <= 2e-5 * abs(float(g["score"]))
    assert np.abs(d["grad"] - g["grad"]).max() <= 2e-5 * np.abs(g["grad"]).max()
    assert np.abs(d["hess_d"] - g["hess_d"]).max() <= 1e-8 * np.abs(g["hess_d"]).max()
    pose = g["init"].copy()
    conv = reg.scan2Map(g["scan"], g["map"], pose)
    assert conv == bool(g["converged"]) and reg.stats()["iterations"] == int(g["iterations"])
    dt, dr = synth.pose_error(pose, g["pose"])
    assert dt <= 1e-4 and dr <= 1e-4


def test_randomised_configurations(gpu):
    """Seeded draws of world, scan, initial error and resolution: same convergence flag, iteration count and pose as the
    oracle (float inner math on both sides; the device sums in a different order, hence the 1e-4 bar and not equality)."""
    rng = np.random.default_rng(555)
    for case in range(5):
        world, m = synth.make_map(int(rng.integers(100_000, 250_000)), seed=5000 + case, spacing=0.2)
        scan, T = synth.make_scan(world, int(rng.integers(0, 4)), seed=5000 + case, beams=int(rng.choice([16, 32])), azimuths=256)
        init = synth.perturb(T, 6000 + case, trans=float(rng.uniform(0.03, 0.2)), rot_deg=float(rng.uniform(0.1, 1.0)))
        res = float(rng.choice([1.0, 1.5, 2.0]))
        reg = NdtRegister(ndt_resolution=res)
        pose = init.copy()
        conv = reg.scan2Map(scan, m, pose)
        po, co, info = oracle.ndt_scan2map(scan, m, init, oracle.ndt_params(resolution=res))
        assert np.isfinite(po).all() and np.isfinite(pose).all(), case
        assert conv == co, (case, res)
        assert reg.stats()["iterations"] == info["iterations"], (case, res)
        dt, dr = synth.pose_error(pose, po)
        assert dt <= 1e-4 and dr <= 1e-4, (case, res, dt, dr)


def test_edge_cases_match_oracle(gpu, nd_world):
    """Inputs at the edge of the algorithm: a scan point at the sensor origin, a target too sparse for any voxel to
    qualify (no Gaussians at all), an empty scan.  Same flag, same iteration count, same (possibly unchanged) pose as the
    oracle."""
    w = nd_world
    T0 = synth.perturb(w["truth"], 31, trans=0.1, rot_deg=0.5)
    scan0 = w["scan"].copy(); scan0[0, :3] = 0.0
    sparse = w["map"][::200].copy()                               # < 6 points in every 1 m voxel
    for name, scan, m in (("origin point", scan0, w["map"]), ("no voxels", w["scan"], sparse), ("empty scan", w["scan"][:0], w["map"])):
        po, co, info = oracle.ndt_scan2map(scan, m, T0)
        reg = NdtRegister()
        pose = T0.copy()
        conv = reg.scan2Map(scan, m, pose)
        assert conv == co, name
        assert reg.stats()["iterations"] == info["iterations"], name
        assert np.isfinite(pose).all() == np.isfinite(po).all(), name
        if np.isfinite(po).all():
            dt, dr = synth.pose_error(pose, po)
            assert dt <= 1e-4 and dr <= 1e-4, (name, dt, dr)


def test_device_resident_optimiser_equals_the_host_driven_one(gpu, nd_world):
    """The Newton / More-Thuente loop is one state machine (csrc/ndt_opt.h) that runs on the device by default (no host round
    trip between evaluation passes) and on the host for sharded targets (pcr_params.host_optimiser = 1 selects it here).  Same
    decisions -- convergence flag, iterations, derivative and Hessian passes -- and the same Matrix4f pose up to the last bits
    of the 6x6 solve (elimination on the device, the restated JacobiSVD on the host) and of the two libms' sine."""
    from simpleslam_amd.pcr import default_params
    w = nd_world
    p_host = default_params()
    p_host.host_optimiser = 1
    dev, host = NdtRegister(), NdtRegister(params=p_host)
    for seed, tr, rd in ((31, 0.1, 0.5), (32, 0.15, 0.8), (33, 0.05, 0.3), (34, 0.4, 2.0), (35, 0.0, 0.0)):
        T0 = synth.perturb(w["truth"], seed, trans=tr, rot_deg=rd) if tr else w["truth"].copy()
        pd, ph = T0.copy(), T0.copy()
        cd = dev.scan2Map(w["scan"], w["map"], pd)
        ch = host.scan2Map(w["scan"], w["map"], ph)
        assert cd == ch, seed
        sd, sh = dev.stats(), host.stats()
        assert (sd["iterations"], sd["kernel_launches"]) == (sh["iterations"], sh["kernel_launches"]), (seed, sd, sh)
        dt, dr = synth.pose_error(pd, ph)
        assert dt <= 2e-6 and dr <= 2e-6, (seed, dt, dr)      # (a float ulp of the pose at 10 m is 1e-6)
    # many calls on one handle: passes enqueued beyond the end of one alignment must not leak into the next
    T0 = synth.perturb(w["truth"], 31, trans=0.1, rot_deg=0.5)
    first = T0.copy(); dev.scan2Map(w["scan"], w["map"], first)
    for _ in range(20):
        p = T0.copy(); dev.scan2Map(w["scan"], w["map"], p)
        np.testing.assert_array_equal(p, first)


def test_one_handle_through_a_sequence_of_unrelated_targets(gpu, nd_world):
    """scan2Map enqueues the target's index unchecked, with the previous target's box, tile layout and cell count as hints, and learns
    with the result whether that held.  A handle driven through targets that move, grow, shrink and thin out must return, every time,
    the pose of a handle that has never seen another target -- bit for bit (the voxel a point belongs to does not depend on where
    the lattice starts)."""
    w = nd_world
    rng = np.random.default_rng(7)
    base = w["map"]
    reg = NdtRegister()
    for case in range(12):
        kind = case % 6
        m = base
        if kind == 1: m = base + np.array([float(rng.uniform(-40, 40)), float(rng.uniform(-40, 40)), 0, 0], np.float32)
        elif kind == 2: m = base[:: int(rng.integers(2, 6))]
        elif kind == 3: m = np.vstack([base, base[: 2000] + np.array([300.0, -200.0, 20.0, 0], np.float32)])
        elif kind == 4: m = base[: int(rng.integers(200, 5000))]
        elif kind == 5: m = np.vstack([base, base[::2] + np.array([0.01, 0.0, 0.0, 0], np.float32)])
        m = np.ascontiguousarray(m, np.float32)
        shift = (m[:, :3].mean(0) - base[:, :3].mean(0)) if kind == 1 else np.zeros(3)
        T0 = synth.perturb(w["truth"], 40 + case, trans=0.1, rot_deg=0.5)
        T0[:3, 3] += shift
        p_fresh, p_used = T0.copy(), T0.copy()
        c_fresh = NdtRegister().scan2Map(w["scan"], m, p_fresh)
        c_used = reg.scan2Map(w["scan"], m, p_used)
        assert c_used == c_fresh, (case, kind)
        np.testing.assert_array_equal(p_used, p_fresh, err_msg=f"case {case} kind {kind}")
    # an empty target must not leave its header behind as a hint
    nothing = np.zeros((0, base.shape[1]), np.float32)
    for m in (nothing, nothing, base):
        T0 = synth.perturb(w["truth"], 77, trans=0.1, rot_deg=0.5)
        p_fresh, p_used = T0.copy(), T0.copy()
        assert NdtRegister().scan2Map(w["scan"], m, p_fresh) == reg.scan2Map(w["scan"], m, p_used)
        np.testing.assert_array_equal(p_used, p_fresh)


def test_rank_deficient_voxels_follow_the_references_summation_order(gpu):
    """A voxel whose covariance is singular but for rounding -- here EVERY voxel: three distinct points, each given twice -- is kept
    or dropped on the SIGN of that rounding (voxel_grid_covariance_omp_impl.hpp:337-341), i.e. on the exact arithmetic of the
    reference's sums: added up in input order, in double, about the origin (:233-237).  The device's first look at a voxel uses
    order-independent sums about the voxel's centre (exact, but rounded elsewhere); a voxel that turns out singular to rounding is
    summed again the reference's way, and the score, gradient and Hessian must then be the oracle's -- in whatever order the map comes.
    (Without that second look this map loses or gains voxels: soak seed 53, case 31 was one such voxel, 3 cm in the pose.)"""
    world, m = synth.make_map(60_000, seed=5, spacing=0.25)
    scan, T = synth.make_scan(world, 0, seed=5, beams=16, azimuths=256)
    # the first three points of every occupied 1 m voxel, each twice
    _, inv = np.unique(np.floor(m[:, :3]).astype(np.int64), axis=0, return_inverse=True)
    order = np.argsort(inv.ravel(), kind="stable")
    srt = inv.ravel()[order]
    start = np.r_[0, np.flatnonzero(np.diff(srt)) + 1]
    rank = np.empty(len(m), np.int64)
    rank[order] = np.arange(len(m)) - np.repeat(start, np.diff(np.r_[start, len(m)]))
    m3 = m[rank < 3]
    m2 = np.concatenate([m3, m3], 0)
    prm = oracle.ndt_params()
    p = _p_of(T)
    for k in range(3):
        mm = m2 if k == 0 else m2[np.random.default_rng(k).permutation(m2.shape[0])]
        reg = NdtRegister()
        reg.setTarget(mm)
        g = reg.derivatives(scan, p, double_hessian=True)
        o = oracle.ndt_derivatives(scan, mm, p, prm, double_hessian=True)
        assert abs(o["score"]) > 100.0      # (many of the voxels are kept)
        assert abs(g["score"] - o["score"]) <= 1e-9 * abs(o["score"]), (k, g["score"], o["score"])
        np.testing.assert_allclose(g["grad"], o["grad"], rtol=1e-7, atol=1e-7 * np.abs(o["grad"]).max())
        np.testing.assert_allclose(g["hess_d"], o["hess_d"], rtol=1e-7, atol=1e-7 * np.abs(o["hess_d"]).max())


def test_replayed_line_search_evaluations_change_nothing_but_the_number_of_passes(gpu, nd_world):
    """pclomp clamps every trial step of the More-Thuente search into [epsilon / 2, step_size]; a search that wants a longer step than
    step_size evaluates that one clamped step again and again -- the same point, hence (fixed-order sums) the same numbers.  By default the
    state machine answers such a request from the sums it holds; pcr_params.ndt_evaluate_repeats = 1 makes every request a pass, as the reference
    does.  Same flag, iterations and evaluation count, bit for bit the same pose -- and fewer passes."""
    from simpleslam_amd.pcr import default_params
    w = nd_world
    p_all = default_params()
    p_all.ndt_evaluate_repeats = 1
    fast, slow = NdtRegister(), NdtRegister(params=p_all)
    saved = 0
    for seed, tr, rd in ((51, 0.1, 0.5), (52, 0.3, 1.5), (53, 0.05, 0.2), (54, 0.0, 0.0)):
        T0 = synth.perturb(w["truth"], seed, trans=tr, rot_deg=rd) if tr else w["truth"].copy()
        pf, ps = T0.copy(), T0.copy()
        assert fast.scan2Map(w["scan"], w["map"], pf) == slow.scan2Map(w["scan"], w["map"], ps)
        sf, ss = fast.stats(), slow.stats()
        assert (sf["iterations"], sf["kernel_launches"]) == (ss["iterations"], ss["kernel_launches"]), (seed, sf, ss)
        np.testing.assert_array_equal(pf, ps)
        # (a search that ends at its first trial point and is followed by another Newton step costs the fast schedule one pass more: the
        #  float Hessian it had not asked for with that trial -- csrc/ndt_opt.h, need_h)
        assert sf["attempts"] <= ss["attempts"] + sf["iterations"] and ss["attempts"] == ss["kernel_launches"], (seed, sf, ss)
        saved += ss["attempts"] - sf["attempts"]
    assert saved > 0


@pytest.mark.parametrize("off", [0.1, 5.0])
def test_target_prepared_for_the_scans_region_only(gpu, nd_world, off):
    """as tests/test_vgicp_gpu.py::test_target_prepared_for_the_scans_region_only: voxel Gaussians only where the scan can land, the whole
    target when the pose leaves that region -- the pose of the full preparation either way, bit for bit"""
    w = nd_world
    T0 = w["truth"].copy()
    T0[:3, 3] += np.array([off, -0.5 * off, 0.0])
    full = NdtRegister(full_target=1)
    pf = T0.copy(); cf = full.scan2Map(w["scan"], w["map"], pf)
    reg = NdtRegister()
    p = T0.copy(); c = reg.scan2Map(w["scan"], w["map"], p)
    assert c == cf
    np.testing.assert_array_equal(p, pf)
    assert reg.stats()["iterations"] == full.stats()["iterations"]
    assert full.stats()["region_repeats"] == 0
    if off < 1.0:
        assert reg.stats()["region_repeats"] == 0


def test_region_index_of_later_calls_matches_the_whole_target(gpu, nd_world):
    """From its second pcr_scan2map on, a handle indexes only the target points inside the scan's region (BuildFilter: the lattice and the
    tile layout of the first, full build are the hints that make it possible).  Voxel Gaussians depend on a voxel's own points alone, so
    poses, verdicts and iteration counts are those of a handle that always prepares the whole target, bit for bit -- also from a guess
    metres off (a pose that leaves the region finds every cell outside it unprepared, and the call is repeated on the whole target)."""
    w = nd_world
    full = NdtRegister(full_target=1)
    reg = NdtRegister()
    seen_region_index = 0
    for k, off in enumerate([0.1, -0.2, 0.15, 6.0, 0.05]):
        T0 = w["truth"].copy()
        T0[:3, 3] += np.array([off, -0.5 * off, 0.02 * off])
        pf = T0.copy(); cf = full.scan2Map(w["scan"], w["map"], pf)
        p = T0.copy(); c = reg.scan2Map(w["scan"], w["map"], p)
        assert c == cf, (k, off)
        np.testing.assert_array_equal(p, pf)
        assert reg.stats()["iterations"] == full.stats()["iterations"]
        seen_region_index += reg.stats()["region_index"]
        if k == 0:
            assert reg.stats()["region_index"] == 0      # nothing to go by yet: the whole cloud is indexed
    assert seen_region_index >= 2
    assert full.stats()["region_repeats"] == 0 and full.stats()["region_index"] == 0

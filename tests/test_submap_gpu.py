"""MapManager::updateMap on the device (pcr_map_*) vs the CPU oracle: which key frames are used and which voxel every
transformed point falls into are exact; centroids agree to PCL's float-accumulation rounding.  The assembled sub-map
then serves as `dst` of scan2Map without leaving HBM."""
import numpy as np
import pytest

import oracle
from simpleslam_amd import LoamRegister, SubMap, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def keyframes():
    """A short trajectory through the synthetic world: every key frame is a down-sampled scan in ITS lidar frame plus its pose."""
    world, _ = synth.make_map(20_000, seed=91)
    kfs = []
    for j in range(14):
        scan, T = synth.make_scan(world, j, seed=91, beams=32, azimuths=512)
        ds, _ = oracle.voxel_filter(scan, 0.4)            # LidarOdometry keeps the down-sampled scan as the key frame's cloud
        kfs.append((ds, T))
    return world, kfs


def test_assembly_matches_oracle(gpu, keyframes):
    world, kfs = keyframes
    sm = SubMap()
    for c, T in kfs:
        sm.addKeyFrame(c, T)
    assert sm.keyframes() == len(kfs)
    for center, radius in ((kfs[3][1][:3, 3], 8.0), (kfs[10][1][:3, 3], 3.0), (kfs[0][1][:3, 3] + 100.0, 8.0)):
        n = sm.updateMap(center, radius=radius, grid_size=0.4)
        ref, sel = oracle.submap_assemble([c for c, _ in kfs], [T for _, T in kfs], center, radius, 0.4)
        np.testing.assert_array_equal(sm.submapIdx(), sel)                  # mSubmapIdx
        got = sm.download()
        assert n == got.shape[0] == ref.shape[0]                            # same occupied voxels
        if n:
            np.testing.assert_allclose(got[:, :3], ref[:, :3], rtol=0, atol=5e-4)
            np.testing.assert_allclose(got[:, 3], ref[:, 3], rtol=1e-5, atol=1e-3)
    # the far-away centre selects nothing: empty sub-map
    assert sm.updateMap(kfs[0][1][:3, 3] + 100.0) == 0 and sm.pointer()[1] == 0


def test_radius_is_strict_and_in_double(gpu):
    sm = SubMap()
    pts = np.array([[1.0, 0.0, 0.0, 5.0]], np.float32)
    for x in (0.0, 3.0, 8.0, np.nextafter(8.0, 0.0)):
        T = np.eye(4); T[0, 3] = x
        sm.addKeyFrame(pts, T)
    sm.updateMap(np.zeros(3), radius=8.0, grid_size=0.5)
    np.testing.assert_array_equal(sm.submapIdx(), [0, 1, 3])               # the key frame at exactly 8 m is out (dist < radius)


def test_submap_feeds_scan2map_without_leaving_hbm(gpu, keyframes):
    world, kfs = keyframes
    sm = SubMap()
    for c, T in kfs[:10]:
        sm.addKeyFrame(c, T)
    scan, T_true = synth.make_scan(world, 10, seed=91, beams=32, azimuths=512)
    sm.updateMap(T_true[:3, 3], radius=8.0, grid_size=0.4)
    init = synth.perturb(T_true, 91, trans=0.1, rot_deg=0.5)
    reg = LoamRegister()
    scan_ds = reg.voxelDownSample(scan, 0.4)
    pose = init.copy()
    reg.scan2MapSubmap(scan_ds, sm, pose)
    # the same registration against the downloaded sub-map through the ordinary entry point
    host = sm.download()
    pose2 = init.copy()
    reg.scan2Map(scan_ds, host, pose2)
    np.testing.assert_array_equal(pose, pose2)
    et, er = synth.pose_error(pose, T_true)
    assert et < 0.1 and er < 0.01                                           # and it localises the new scan in the map of the old ones
    # and against the oracle's LOAM on the oracle's sub-map: same pose to the usual bar
    ref_map, _ = oracle.submap_assemble([c for c, _ in kfs[:10]], [T for _, T in kfs[:10]], T_true[:3, 3], 8.0, 0.4)
    po, _, _ = oracle.loam_scan2map(scan_ds, ref_map, init)
    dt, dr = synth.pose_error(pose, po)
    assert dt <= 1e-4 and dr <= 1e-4


def test_mixed_layouts_rejected(gpu):
    sm = SubMap()
    sm.addKeyFrame(np.zeros((3, 4), np.float32), np.eye(4))
    with pytest.raises(Exception):
        sm.addKeyFrame(np.zeros((3, 8), np.float32), np.eye(4))


def test_loop_closure_flow(gpu, keyframes):
    """LoopClosureManager::lcHandler (backend/src/LoopClosureManager.cpp:72-110) on the device: the history sub-map around
    an old key frame (loopFindNearKeyframes), the current key frame's cloud registered against it with the loop-closure
    VGICP settings (VgicpRegister::initForLC), accepted on convergence and fitness."""
    from simpleslam_amd import VgicpRegister
    world, kfs = keyframes
    sm = SubMap()
    for c, T in kfs:
        sm.addKeyFrame(c, T)
    old_key, cur_key, rng = 3, 12, 2
    n = sm.loopFindNearKeyframes(old_key, rng, grid_size=0.4)
    np.testing.assert_array_equal(sm.submapIdx(), [1, 2, 3, 4, 5])
    window = list(range(old_key - rng, old_key + rng + 1))
    ref, _ = oracle.submap_assemble([kfs[i][0] for i in window], [kfs[i][1] for i in window], np.zeros(3), 1e12, 0.4)
    got = sm.download()
    assert n == got.shape[0] == ref.shape[0]
    np.testing.assert_allclose(got[:, :3], ref[:, :3], rtol=0, atol=5e-4)
    # clipped at the ends of the store
    sm.loopFindNearKeyframes(0, 2)
    np.testing.assert_array_equal(sm.submapIdx(), [0, 1, 2])
    sm.loopFindNearKeyframes(len(kfs) - 1, 3)
    np.testing.assert_array_equal(sm.submapIdx(), [len(kfs) - 4, len(kfs) - 3, len(kfs) - 2, len(kfs) - 1])
    # registration of a key frame against the history map around itself, from a drifted pose (what a loop closure corrects)
    key = 6
    sm.loopFindNearKeyframes(key, rng, grid_size=0.4)
    scan, T_true = kfs[key]
    guess = synth.perturb(T_true, 7, trans=0.3, rot_deg=1.5)
    # the reference's call sequence: a constructed registrar is switched to the loop-closure settings (LoopClosureManager.cpp:21-22)
    # and then registers (:98) -- same answer as a handle created with those settings
    lc = VgicpRegister()
    lc.initForLC()                                                         # VgicpRegister.cpp:21-28 on the live object
    assert (lc.params.vgicp_max_iters, lc.params.vgicp_trans_eps) == (100, 1e-6)
    pose = guess.copy()
    conv = lc.scan2MapSubmap(scan, sm, pose)
    born = VgicpRegister(vgicp_max_iters=100, vgicp_trans_eps=1e-6)
    pose_b = guess.copy()
    assert born.scan2MapSubmap(scan, sm, pose_b) == conv
    np.testing.assert_array_equal(pose, pose_b)
    # and a handle that has already registered with the odometry settings can be switched too (the prepared target survives)
    lc2 = VgicpRegister()
    tmp = guess.copy()
    lc2.scan2MapSubmap(scan, sm, tmp)
    lc2.initForLC()
    pose_c = guess.copy()
    assert lc2.scan2MapSubmap(scan, sm, pose_c) == conv
    np.testing.assert_array_equal(pose, pose_c)
    fs = lc.getFitnessScore()
    et, er = synth.pose_error(pose, T_true)
    assert conv and et < 0.05 and er < 5e-3
    assert fs < 0.3                                                        # fitnessThreshold-style acceptance (config/params.json)
    po, co, _ = oracle.vgicp_scan2map(scan, sm.download(), guess, oracle.vgicp_params(max_iters=100, trans_eps=1e-6, threads=8))
    dt, dr = synth.pose_error(pose, po)
    assert co == conv and dt <= 1e-4 and dr <= 1e-4


@pytest.mark.parametrize("method", ["loam", "ndt", "vgicp"])
def test_target_structures_live_as_long_as_the_submap_generation(gpu, keyframes, method):
    """LidarOdometry registers several scans against one sub-map between two MapManager updates.  pcr_scan2map_submap keys the
    handle's target structures on (map id, generation): they are built once per generation, rebuilt when the map moves on or when
    another entry point used the handle in between, and the pose is the one pcr_scan2map_device gives on the same memory."""
    from simpleslam_amd import make_register
    world, kfs = keyframes
    sm = SubMap()
    for c, T in kfs[:10]:
        sm.addKeyFrame(c, T)
    scans = [synth.make_scan(world, 10 + k, seed=91, beams=32, azimuths=512) for k in range(3)]
    sm.updateMap(scans[0][1][:3, 3], radius=8.0, grid_size=0.4)
    g0 = sm.generation()
    reg, ref = make_register(method), make_register(method)
    builds = lambda: reg.stats()["target_builds"]
    for k, (scan, T_true) in enumerate(scans):
        ds = reg.voxelDownSample(scan, 0.4)
        init = synth.perturb(T_true, 91 + k, trans=0.1, rot_deg=0.5)
        p_keep, p_ref = init.copy(), init.copy()
        c_keep = reg.scan2MapSubmap(ds, sm, p_keep)
        c_ref = ref.scan2MapSubmap(ds, sm, p_ref, rebuild=True)
        assert c_keep == c_ref
        np.testing.assert_array_equal(p_keep, p_ref)
    assert builds() == 1                                   # three scans, one sub-map generation, one build
    # the map moves on: new generation, new structures
    sm.updateMap(scans[1][1][:3, 3] + np.array([0.5, 0.0, 0.0]), radius=8.0, grid_size=0.4)
    assert sm.generation()[0] == g0[0] and sm.generation()[1] == g0[1] + 1
    ds = reg.voxelDownSample(scans[1][0], 0.4)
    init = synth.perturb(scans[1][1], 95, trans=0.1, rot_deg=0.5)
    p_keep, p_ref = init.copy(), init.copy()
    reg.scan2MapSubmap(ds, sm, p_keep)
    ref.scan2MapSubmap(ds, sm, p_ref, rebuild=True)
    np.testing.assert_array_equal(p_keep, p_ref)
    assert builds() == 2
    # another entry point on the same handle replaces the structures: they are not mistaken for the sub-map's afterwards
    other = sm.download()[::2].copy()
    tmp = init.copy(); reg.scan2Map(ds, other, tmp)
    p_again = init.copy(); reg.scan2MapSubmap(ds, sm, p_again)
    np.testing.assert_array_equal(p_again, p_ref)
    assert builds() == 3
    # a second store with an equal generation number is a different map
    sm2 = SubMap()
    for c, T in kfs[:6]:
        sm2.addKeyFrame(c, T)
    sm2.updateMap(scans[0][1][:3, 3], radius=8.0, grid_size=0.4); sm2.updateMap(scans[0][1][:3, 3], radius=8.0, grid_size=0.4)
    assert sm2.generation()[1] == sm.generation()[1] and sm2.generation()[0] != sm.generation()[0]
    p2 = init.copy(); reg.scan2MapSubmap(ds, sm2, p2)
    assert builds() == 4


def test_an_update_in_two_halves_is_the_update(gpu, keyframes):
    """pcr_map_update_begin queues the assembly, pcr_map_wait -- or whoever needs the sub-map first -- collects it: same key frames, same sub-map bit
    for bit, same generation count as pcr_map_update, whatever happens in between (a voxel filter on another handle, a key frame added, another
    update that supersedes it)."""
    world, kfs = keyframes
    whole, halves = SubMap(), SubMap()
    for c, T in kfs[:10]:
        whole.addKeyFrame(c, T); halves.addKeyFrame(c, T)
    reg = LoamRegister()
    scan, _ = synth.make_scan(world, 3, seed=91, beams=32, azimuths=512)
    for step, (center, radius) in enumerate(((kfs[3][1][:3, 3], 8.0), (kfs[8][1][:3, 3], 3.0), (kfs[0][1][:3, 3] + 100.0, 8.0), (kfs[5][1][:3, 3], 6.0))):
        n = whole.updateMap(center, radius=radius, grid_size=0.4)
        halves.updateMapBegin(center, radius=radius, grid_size=0.4)
        reg.voxelDownSample(scan, 0.4)                                       # other work, on another handle's stream
        if step % 2 == 0:
            assert halves.wait() == n                                        # collected explicitly ...
        np.testing.assert_array_equal(halves.submapIdx(), whole.submapIdx())
        assert halves.pointer()[1] == n                                      # ... or by asking for the sub-map
        assert halves.generation()[1] == whole.generation()[1]
        if n:
            np.testing.assert_array_equal(halves.download(), whole.download())
        assert halves.wait() == n                                            # nothing queued any more: the same count again
    # a key frame added, or another update begun, while an assembly is queued: the queued one is collected (or superseded) first
    halves.updateMapBegin(kfs[3][1][:3, 3], radius=8.0, grid_size=0.4)
    halves.addKeyFrame(*kfs[10]); whole.addKeyFrame(*kfs[10])
    halves.updateMapBegin(kfs[9][1][:3, 3], radius=8.0, grid_size=0.4)
    halves.updateMapBegin(kfs[10][1][:3, 3], radius=8.0, grid_size=0.4)
    n = whole.updateMap(kfs[10][1][:3, 3], radius=8.0, grid_size=0.4)
    pose_a, pose_b = kfs[10][1].copy(), kfs[10][1].copy()
    ca = reg.scan2MapSubmap(scan, halves, pose_a)                            # the registration collects it
    reg2 = LoamRegister()
    cb = reg2.scan2MapSubmap(scan, whole, pose_b)
    assert halves.wait() == n and ca == cb
    np.testing.assert_array_equal(pose_a, pose_b)
    np.testing.assert_array_equal(halves.download(), whole.download())


def test_a_cleared_map_is_a_new_map_on_the_old_memory(gpu, keyframes):
    """pcr_map_clear: no key frame, no sub-map, a new generation -- and the same key frames added again give the sub-map a fresh store gives, bit for bit."""
    world, kfs = keyframes
    used, fresh = SubMap(), SubMap()
    for c, T in kfs:
        used.addKeyFrame(c, T)
    used.updateMap(kfs[3][1][:3, 3], radius=8.0, grid_size=0.4)
    g0 = used.generation()[1]
    used.updateMapBegin(kfs[5][1][:3, 3], radius=8.0, grid_size=0.4)            # (a queued assembly is waited for and dropped)
    used.clear()
    assert used.keyframes() == 0 and used.generation()[1] > g0 + 1
    assert used.updateMap(kfs[3][1][:3, 3], radius=8.0, grid_size=0.4) == 0 and used.pointer()[1] == 0 and len(used.submapIdx()) == 0
    for c, T in kfs[2:9]:
        used.addKeyFrame(c, T); fresh.addKeyFrame(c, T)
    n = fresh.updateMap(kfs[4][1][:3, 3], radius=6.0, grid_size=0.4)
    assert used.updateMap(kfs[4][1][:3, 3], radius=6.0, grid_size=0.4) == n > 0
    np.testing.assert_array_equal(used.submapIdx(), fresh.submapIdx())
    np.testing.assert_array_equal(used.download(), fresh.download())

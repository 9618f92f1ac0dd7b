"""BASELINE.json's full size (65 536-point scan, 1 M-point sub-map) through properties that do not need the oracle to finish
in seconds: repeatability, exactness of the neighbour cache, rigid-motion equivariance, independence from the initial
error, down-sampling idempotence -- plus one oracle comparison of the full registration."""
import numpy as np
import pytest

import oracle
from simpleslam_amd import LoamRegister, synth, pcr

pytestmark = pytest.mark.gpu
S = 20261003 + 2


@pytest.fixture(scope="module")
def full():
    import torch
    world, m = synth.make_map(1_000_000, seed=S)
    scan, T = synth.make_scan(world, 0, seed=S)
    assert scan.shape[0] == 65536 and m.shape[0] == 1_000_000
    return dict(map=m, scan=scan, truth=T, init=synth.perturb(T, S), d_map=torch.from_numpy(m).cuda(), d_scan=torch.from_numpy(scan).cuda())


def _run(full, init, **kw):
    reg = LoamRegister(loam_iters=10, loam_early_exit=0, **kw)
    pose = init.copy()
    reg.scan2Map(full["d_scan"], full["d_map"], pose)
    return pose, reg


def test_bitwise_repeatable_and_cache_exact(gpu, full):
    p1, _ = _run(full, full["init"])
    p2, _ = _run(full, full["init"])
    np.testing.assert_array_equal(p1, p2)                       # fixed-order reductions
    prm = pcr.default_params(loam_iters=10, loam_early_exit=0)
    prm.loam_disable_cache = 1                                          # neighbour cache off: every iteration searches
    reg = LoamRegister(params=prm)
    p3 = full["init"].copy()
    reg.scan2Map(full["d_scan"], full["d_map"], p3)
    np.testing.assert_array_equal(p1, p3)                       # the temporal-coherence cache is exact, not approximate


def test_matches_oracle_at_full_size(gpu, full):
    p, _ = _run(full, full["init"])
    po, _, _ = oracle.loam_scan2map(full["scan"], full["map"], full["init"], oracle.loam_params(iters=10, early_exit=0, threads=8))
    dt, dr = synth.pose_error(p, po)
    assert dt <= 1e-4 and dr <= 1e-4                             # BASELINE's bar (measured: ~1e-15)
    et, er = synth.pose_error(p, full["truth"])
    assert et < 0.02 and er < 2e-3


def test_independent_of_the_initial_error(gpu, full):
    p1, _ = _run(full, full["init"])
    p2, _ = _run(full, synth.perturb(full["truth"], S + 99, trans=0.2, rot_deg=1.0))
    dt, dr = synth.pose_error(p1, p2)
    assert dt < 5e-3 and dr < 5e-4                               # same basin, same answer (up to where 10 iterations get)


def test_rigid_motion_equivariance(gpu, full):
    """Registering against G(map) from G o init gives G o result -- approximately: LOAM's plane model A x = -1 is written
    relative to the map origin (LoamRegister.cpp:29-45), so its validity gate and the accepted point set move a little with
    the frame; the two answers agree to millimetres, not to rounding."""
    import torch
    G = np.eye(4)
    c, s = np.cos(0.3), np.sin(0.3)
    G[:3, :3] = [[c, -s, 0], [s, c, 0], [0, 0, 1]]
    G[:3, 3] = [12.5, -7.25, 0.5]
    m2 = full["map"].copy()
    m2[:, :3] = (full["map"][:, :3].astype(np.float64) @ G[:3, :3].T + G[:3, 3]).astype(np.float32)
    reg = LoamRegister(loam_iters=10, loam_early_exit=0)
    p = G @ full["init"]
    reg.scan2Map(full["d_scan"], torch.from_numpy(m2).cuda(), p)
    p_ref, _ = _run(full, full["init"])
    dt, dr = synth.pose_error(p, G @ p_ref)
    assert dt < 1e-2 and dr < 1e-3


def test_static_target_equals_per_call_rebuild(gpu, full):
    p1, _ = _run(full, full["init"])
    reg = LoamRegister(loam_iters=10, loam_early_exit=0)
    reg.setTarget(full["d_map"])
    p2 = full["init"].copy()
    reg.align(full["d_scan"], p2)
    np.testing.assert_array_equal(p1, p2)


def test_voxel_filter_idempotent_at_full_size(gpu, full):
    reg = LoamRegister()
    once = reg.voxelDownSample(full["d_map"], 0.5)
    twice = reg.voxelDownSample(once, 0.5)
    # a centroid stays in its voxel unless rounding puts it exactly on a face: the count cannot grow and barely shrinks
    assert twice.shape[0] <= once.shape[0] and twice.shape[0] >= 0.999 * once.shape[0]
    ref, _ = oracle.voxel_filter(full["map"], 0.5)
    assert once.shape[0] == ref.shape[0]


def test_config4_eight_tiles_of_a_10m_map_sum_to_the_full_system(gpu):
    """BASELINE configs[3] (10 M-point sub-map cut into 8 tiles) on one GPU: each tile (+1 m halo) is indexed by itself and
    linearises only the scan points that fall into it; every point is owned exactly once, its row is the row of the
    unsharded run bit for bit, and the eight partial systems add up to the unsharded one."""
    from simpleslam_amd import shard
    world, m = synth.make_map(10_000_000, seed=S + 2)
    scan, T = synth.make_scan(world, 0, seed=S + 2)
    init = synth.perturb(T, S + 2)
    full = LoamRegister()
    full.setTarget(m)
    ref = full.linearize(scan, init, per_point=True)
    del full
    JtJ, JtE, n = np.zeros((6, 6)), np.zeros(6), 0
    owned = np.zeros(scan.shape[0], int)
    sizes = []
    for r in range(8):
        tile = shard.tile_for_rank(m, r, 8)
        sizes.append(tile.points.shape[0])
        reg = LoamRegister()
        reg.setTarget(tile.points)
        reg.set_query_tile(tile.lo, tile.hi)
        part = reg.linearize(scan, init, per_point=True)
        mine = part["status"] != 4
        owned += mine
        np.testing.assert_array_equal(part["status"][mine], ref["status"][mine])
        np.testing.assert_array_equal(part["rows"][mine], ref["rows"][mine])
        JtJ += part["JtJ"]; JtE += part["JtE"]; n += part["n"]
        del reg
    assert (owned == 1).all()
    assert max(sizes) < 1.35 * 10_000_000 / 8                     # balanced by point count; the halo adds a little
    assert n == ref["n"] and n > 30_000
    np.testing.assert_allclose(JtJ, ref["JtJ"], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(JtE, ref["JtE"], rtol=1e-10, atol=1e-9)

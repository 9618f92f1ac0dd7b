"""BASELINE.json configs[2] (pcr=vgicp, 0.5 m voxels, 65 536 x 1 M) and configs[4] (pcr=ndt, 1 m cells, 131 072 x 5 M) at full
size: repeatability, a prepared target against the per-call rebuild, independence from the order of the map points, and
one oracle comparison each."""
import numpy as np
import pytest

import oracle
from simpleslam_amd import NdtRegister, VgicpRegister, synth

pytestmark = pytest.mark.gpu
SEED = 20261003


def _world(cfg, n_map, scan_kw, map_kw, pert):
    import torch
    world, m = synth.make_map(n_map, seed=SEED + cfg, **map_kw)
    scan, T = synth.make_scan(world, 0, seed=SEED + cfg, **scan_kw)
    return dict(map=m, scan=scan, truth=T, init=synth.perturb(T, SEED + cfg, **pert), d_map=torch.from_numpy(m).cuda(),
                d_scan=torch.from_numpy(scan).cuda())


@pytest.fixture(scope="module")
def vg_full():
    w = _world(3, 1_000_000, {}, {}, {})
    assert w["scan"].shape[0] == 65536
    return w


@pytest.fixture(scope="module")
def nd_full():
    w = _world(5, 5_000_000, dict(beams=128, azimuths=1024), dict(spacing=0.22), dict(trans=0.1, rot_deg=0.5))
    assert w["scan"].shape[0] == 131072
    return w


def _properties(make, w, oracle_pose, iters_key, truth_tol=None):
    import torch
    reg = make()
    p1 = w["init"].copy(); c1 = reg.scan2Map(w["d_scan"], w["d_map"], p1)
    it1 = reg.stats()["iterations"]
    p2 = w["init"].copy(); c2 = make().scan2Map(w["d_scan"], w["d_map"], p2)
    np.testing.assert_array_equal(p1, p2)                                  # fixed-order reductions: bitwise repeatable
    assert c1 == c2
    # a target prepared once gives the pose of the per-call rebuild
    reg3 = make(); reg3.setTarget(w["d_map"])
    p3 = w["init"].copy(); reg3.align(w["d_scan"], p3)
    np.testing.assert_array_equal(p1, p3)
    # the order of the map points does not matter (cell-sorted index, order-independent voxel sums)
    perm = torch.randperm(w["map"].shape[0], generator=torch.Generator().manual_seed(1)).cuda()
    p4 = w["init"].copy(); make().scan2Map(w["d_scan"], w["d_map"][perm].contiguous(), p4)
    dt, dr = synth.pose_error(p1, p4)
    assert dt <= 1e-4 and dr <= 1e-4, (dt, dr)
    # the oracle on the same inputs
    po, co, info = oracle_pose()
    assert c1 == co and it1 == info[iters_key]
    dt, dr = synth.pose_error(p1, po)
    assert dt <= 1e-4 and dr <= 1e-4, (dt, dr)                             # BASELINE's bar
    if truth_tol:
        et, er = synth.pose_error(p1, w["truth"])
        assert et < truth_tol[0] and er < truth_tol[1]


def test_vgicp_config_at_full_size(gpu, vg_full):
    w = vg_full
    _properties(lambda: VgicpRegister(vgicp_resolution=0.5), w,
                lambda: oracle.vgicp_scan2map(w["scan"], w["map"], w["init"], oracle.vgicp_params(resolution=0.5, threads=16)), "outer",
                truth_tol=(0.05, 5e-3))


def test_ndt_config_at_full_size(gpu, nd_full):
    # (no distance-to-truth check: with the reference's step size 0.1 and epsilon 0.1 NDT stops after two clipped steps here,
    #  on the device as in the oracle)
    w = nd_full
    _properties(lambda: NdtRegister(), w, lambda: oracle.ndt_scan2map(w["scan"], w["map"], w["init"], oracle.ndt_params()), "iterations")

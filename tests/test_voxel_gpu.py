"""pcl::VoxelGrid on the device (pcr_voxel_filter) vs the CPU oracle: the voxel lattice and the membership of every point
are integer work and must agree exactly; centroids agree to PCL's own float-accumulation rounding (the oracle sums in
float like PCL, the device in double)."""
import ctypes as C

import numpy as np
import pytest

import oracle
from simpleslam_amd import LoamRegister, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def clouds():
    w, m = synth.make_map(200_000, seed=77)
    scan, T = synth.make_scan(w, 0, seed=77)
    return dict(map=m, scan=scan)


def _voxel_ids(pts, ref_pts, leaf):
    """PCL voxel index of every row of pts on the lattice of ref_pts (float arithmetic as in voxel_grid.hpp)."""
    inv = np.float32(1.0) / np.float32(leaf)
    fin = np.isfinite(ref_pts[:, :3]).all(1)
    mn = ref_pts[fin, :3].min(0)
    mx = ref_pts[fin, :3].max(0)
    min_b = np.floor(mn * inv).astype(np.int64)
    div_b = np.floor(mx * inv).astype(np.int64) - min_b + 1
    ijk = (np.floor(pts[:, :3].astype(np.float32) * inv) - min_b.astype(np.float32)).astype(np.int64)
    return ijk[:, 0] + ijk[:, 1] * div_b[0] + ijk[:, 2] * div_b[0] * div_b[1]


@pytest.mark.parametrize("which,leaf", [("scan", 0.4), ("map", 0.8), ("scan", 0.1), ("map", 2.5)])
def test_matches_oracle(gpu, clouds, which, leaf):
    pts = clouds[which]
    reg = LoamRegister()
    got = reg.voxelDownSample(pts, leaf)
    ref, unfiltered = oracle.voxel_filter(pts, leaf)
    assert not unfiltered
    assert got.shape == ref.shape                                   # same number of occupied voxels
    # ascending voxel order on both sides, and every centroid lies in the voxel the oracle's does: membership is exact
    ids_ref = np.unique(_voxel_ids(pts, pts, leaf))
    assert len(ids_ref) == len(ref)
    # centroids: float-accumulation rounding of PCL (n * eps * |x|, a few 1e-4 m for hundreds of points at |x| ~ 100 m)
    np.testing.assert_allclose(got[:, :3], ref[:, :3], rtol=0, atol=5e-4)
    np.testing.assert_allclose(got[:, 3], ref[:, 3], rtol=1e-5, atol=1e-3)
    # an independent float64 centroid per voxel (numpy): the device sums in double and must match it to float rounding
    ids = _voxel_ids(pts, pts, leaf)
    order = np.argsort(ids, kind="stable")
    uniq, start, cnt = np.unique(ids[order], return_index=True, return_counts=True)
    sums = np.add.reduceat(pts[order, :3].astype(np.float64), start, axis=0)
    exact = (sums / cnt[:, None]).astype(np.float32)
    np.testing.assert_allclose(got[:, :3], exact, rtol=0, atol=1e-5)


def test_pcl_point_layout_and_nan_rows(gpu, clouds):
    scan = clouds["scan"]
    p32 = np.zeros((scan.shape[0], 8), np.float32)
    p32[:, :3] = scan[:, :3]; p32[:, 3] = 1.0; p32[:, 4] = scan[:, 3]
    p32[::53, 1] = np.nan                                          # skipped like !isFinite points in PCL
    reg = LoamRegister()
    got = reg.voxelDownSample(p32, 0.4)
    ref, _ = oracle.voxel_filter(p32, 0.4)
    assert got.shape == ref.shape and got.shape[1] == 8
    assert np.all(got[:, 3] == 1.0) and np.all(got[:, 5:] == 0.0)
    np.testing.assert_allclose(got[:, :3], ref[:, :3], rtol=0, atol=5e-4)
    np.testing.assert_allclose(got[:, 4], ref[:, 4], rtol=1e-5, atol=1e-3)
    assert np.isfinite(got).all()


def test_device_resident_and_idempotent(gpu, clouds):
    import torch
    reg = LoamRegister()
    d = torch.from_numpy(clouds["map"]).cuda()
    out = reg.voxelDownSample(d, 0.8)
    assert out.is_cuda
    host = reg.voxelDownSample(clouds["map"], 0.8)
    np.testing.assert_array_equal(out.cpu().numpy(), host)          # same kernels either way
    # filtering the filtered cloud on the same leaf keeps one point per voxel it already has: the count cannot grow
    again = reg.voxelDownSample(out, 0.8)
    assert again.shape[0] <= out.shape[0]
    # the down-sampled scan registers like the reference's flow (scan -> VoxelGrid -> scan2Map)
    assert out.shape[0] < clouds["map"].shape[0]


def test_edge_cases(gpu, clouds):
    reg = LoamRegister()
    assert reg.voxelDownSample(np.zeros((0, 4), np.float32), 0.4).shape == (0, 4)
    one = np.array([[1.0, 2.0, 3.0, 7.0]], np.float32)
    np.testing.assert_array_equal(reg.voxelDownSample(one, 0.4), one)
    allnan = np.full((5, 4), np.nan, np.float32)
    assert reg.voxelDownSample(allnan, 0.4).shape[0] == 0
    # leaf so small that PCL's int voxel index would overflow: the input comes back unfiltered (voxel_grid.hpp warning path)
    pts = clouds["scan"][:1000]
    got = reg.voxelDownSample(pts, 1e-5)
    ref, unfiltered = oracle.voxel_filter(pts, 1e-5)
    assert unfiltered
    np.testing.assert_array_equal(got, pts)
    np.testing.assert_array_equal(ref, pts)
    with pytest.raises(Exception):
        reg.voxelDownSample(pts, 0.0)


def test_capacity_too_small_reports_the_size(gpu, clouds):
    from simpleslam_amd import pcr
    L = pcr.load_library()
    reg = LoamRegister()
    pts = np.ascontiguousarray(clouds["scan"])
    out = np.zeros((10, 4), np.float32)
    cnt = C.c_size_t(0)
    rc = L.pcr_voxel_filter(reg._h, pts.ctypes.data_as(C.c_void_p), pts.shape[0], 16, 0, 0.4, out.ctypes.data_as(C.c_void_p), 10, 0, C.byref(cnt))
    assert rc != 0 and cnt.value == oracle.voxel_filter(pts, 0.4)[0].shape[0]


@pytest.mark.parametrize("stride", [3, 5, 6])
def test_other_point_layouts(gpu, clouds, stride):
    """12-, 20- and 24-byte points: xyz are averaged (float 3 too when present), further floats come out as zero."""
    src = clouds["scan"][:20000]
    pts = np.zeros((src.shape[0], stride), np.float32)
    pts[:, :3] = src[:, :3]
    if stride > 3:
        pts[:, 3] = src[:, 3]
    reg = LoamRegister()
    got = reg.voxelDownSample(pts, 0.4)
    ref, _ = oracle.voxel_filter(pts, 0.4)
    assert got.shape == ref.shape and got.shape[1] == stride
    np.testing.assert_allclose(got[:, :3], ref[:, :3], rtol=0, atol=5e-4)
    if stride > 3:
        np.testing.assert_allclose(got[:, 3], ref[:, 3], rtol=1e-5, atol=1e-3)
        assert np.all(got[:, 4:] == 0)


def test_runs_that_span_many_waves_and_repeated_calls_on_one_handle(gpu):
    """A voxel's points are summed by the waves that hold them (voxel_filter.hip: lead sums + a segmented scan): voxels of one point, of a few,
    of 100 000 (1 563 waves), runs that start and end anywhere in a wave -- the centroid is the f64 mean to float rounding, whatever the run's
    length.  The same handle then filters clouds of other sizes and places: the index's box and layout hints of the previous call are tried first
    and must never change a result."""
    rng = np.random.default_rng(7)
    reg = LoamRegister()
    sizes = [1, 2, 63, 64, 65, 100_000, 3, 4097, 129, 20_000]
    chunks = []
    for k, m in enumerate(sizes):
        c = np.zeros((m, 4), np.float32)
        c[:, :3] = np.array([3.0 * k, -2.0 * (k % 3), 1.0 * k], np.float32) + 0.2 + 0.5 * rng.random((m, 3), dtype=np.float32)      # all inside one 1 m voxel
        c[:, 3] = rng.random(m, dtype=np.float32) * 100
        chunks.append(c)
    pts = np.concatenate(chunks)
    pts = pts[rng.permutation(pts.shape[0])]
    got = reg.voxelDownSample(pts, 1.0)
    assert got.shape[0] == len(sizes)
    key = np.floor(pts[:, :3].astype(np.float64)).astype(np.int64)
    for row in got:
        sel = np.all(key == np.floor(row[:3].astype(np.float64)).astype(np.int64), axis=1)
        assert sel.sum() in sizes
        np.testing.assert_allclose(row, pts[sel].astype(np.float64).mean(0), rtol=2e-7, atol=1e-6)
    ref, _ = oracle.voxel_filter(pts, 1.0)
    assert ref.shape == got.shape
    # other clouds through the same handle: shifted (outside the previous box), smaller, larger, the first again
    for shift, take in ((np.array([40.0, -25.0, 3.0], np.float32), 50_000), (np.zeros(3, np.float32), 500), (np.array([-300.0, 0.0, 0.0], np.float32), pts.shape[0]), (np.zeros(3, np.float32), pts.shape[0])):
        q = pts[:take].copy(); q[:, :3] += shift
        a = reg.voxelDownSample(q, 1.0)
        b = LoamRegister().voxelDownSample(q, 1.0)          # a fresh handle: no hints
        assert a.shape == b.shape
        np.testing.assert_allclose(a, b, rtol=2e-7, atol=1e-6)   # (the order of a voxel's points in the index, hence of the f64 additions, may differ)
        r, _ = oracle.voxel_filter(q, 1.0)
        assert r.shape == a.shape                                 # same occupied voxels
        # (centroids: the oracle adds in float as PCL does -- n * eps * |x| = 0.2 m for the 100 000-point voxel 290 m from the origin; the f64 means above are the check)
        small = np.array([np.all(np.abs(x[:3] - y[:3]) < 0.5) for x, y in zip(a, r)])
        assert small.all()


def test_filter_in_two_halves_on_a_second_handle(gpu, clouds):
    """pcr_voxel_filter_begin / _end: the filter queued on one handle while another registers; same voxels, bit for bit, as the call in one piece on a
    handle in the same state; misuse is refused."""
    import torch
    from simpleslam_amd import pcr
    a, b = LoamRegister(), LoamRegister()
    other = LoamRegister()
    scan = torch.from_numpy(np.ascontiguousarray(clouds["scan"])).cuda()
    m = torch.from_numpy(np.ascontiguousarray(clouds["map"][:200_000])).cuda()
    for rep in range(3):                                                  # (repeated: the hints of the previous call are in play on both handles alike)
        q = scan + float(rep)
        tok = a.voxelDownSampleBegin(q, 0.4)
        pose = np.eye(4)
        other.scan2Map(scan, m, pose)                                     # another handle works meanwhile
        got = a.voxelDownSampleEnd(tok)
        ref = b.voxelDownSample(q, 0.4)
        assert got.shape == ref.shape
        np.testing.assert_array_equal(got.cpu().numpy(), ref.cpu().numpy())
    tok = a.voxelDownSampleBegin(scan, 0.4)
    with pytest.raises(pcr.PcrError):
        a.voxelDownSampleBegin(scan, 0.4)                                 # one at a time
    with pytest.raises(pcr.PcrError):
        a.voxelDownSample(scan, 0.4)
    assert a.voxelDownSampleEnd(tok).shape[0] > 0
    with pytest.raises(pcr.PcrError):
        a.voxelDownSampleEnd(tok)                                         # nothing queued
    empty = torch.zeros((0, 4), dtype=torch.float32, device="cuda")
    assert a.voxelDownSampleEnd(a.voxelDownSampleBegin(empty, 0.4)).shape[0] == 0

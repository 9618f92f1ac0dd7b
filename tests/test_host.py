"""CPU tests (no GPU): the C-ABI library loads and exports every symbol include/pcr_hip.h
declares, host-side helpers behave, and the map-sharding logic of the multi-GPU path is exact
(world_size-2 gloo run, with the oracle standing in for the kernels)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle
from simpleslam_amd import shard, synth
from simpleslam_amd.pcr import ABI_SYMBOLS, LIB_PATH, PcrParams, default_params, load_library, make_register

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pcr_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(pcr_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations found"
    assert sorted(ABI_SYMBOLS) == declared, "simpleslam_amd.pcr.ABI_SYMBOLS is out of sync with include/pcr_hip.h"
    assert os.path.exists(LIB_PATH), "libpcr_hip.so has not been built (run __graft_entry__.build())"
    lib = C.CDLL(LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), f"{name} is declared in pcr_hip.h but not exported"


def test_default_params_are_the_reference_constants():
    p = default_params()
    assert p.struct_size == C.sizeof(PcrParams)
    assert (p.loam_iters, p.loam_early_exit) == (8, 1)                       # LoamRegister.hpp:40
    assert p.loam_knn_max_sq == 1.0                                          # LoamRegister.hpp:31
    assert p.loam_plane_thresh == float(np.float32(0.2))                     # `const float` members
    assert p.loam_point_thresh == float(np.float32(0.1))
    assert p.loam_pos_conv == float(np.float32(5e-3)) == p.loam_rot_conv
    assert (p.ndt_resolution, p.ndt_step_size, p.ndt_outlier_ratio, p.ndt_trans_eps, p.ndt_max_iters) == (1.0, 0.1, 0.55, 0.1, 35)
    assert (p.vgicp_resolution, p.vgicp_k_corr, p.vgicp_max_iters, p.vgicp_lm_inner) == (1.0, 20, 64, 10)
    assert (p.vgicp_rot_eps, p.vgicp_trans_eps) == (2e-3, 5e-4)


def test_unknown_method_raises_like_the_reference_factory():
    with pytest.raises(RuntimeError, match="is not exist"):
        make_register("icp")


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = load_library()
    h = lib.pcr_create(b"loam", None)
    assert not h
    assert b"HIP device" in lib.pcr_last_error(None)
    assert not lib.pcr_create(b"nope", None)
    assert b"is not exist" in lib.pcr_last_error(None)


def test_map_and_scancontext_objects_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = load_library()
    assert not lib.pcr_map_create(-1)
    assert lib.pcr_map_last_error(None)
    assert not lib.pcr_sc_create(-1, None)
    assert b"no CPU fallback" in lib.pcr_sc_last_error(None)


def test_header_is_plain_c(tmp_path):
    """include/pcr_hip.h is the FFI boundary: it must compile as C99 (cgo / JNI / ctypes-style binders read it as C)."""
    import subprocess
    src = tmp_path / "abi.c"
    src.write_text('#include "pcr_hip.h"\nint main(void) { pcr_params p; pcr_sc_params q; pcr_default_params(&p); pcr_sc_default_params(&q);'
                   ' return (int)sizeof(p) + (int)sizeof(q) == 0; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I", os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_the_product_library_reads_no_environment_variable():
    """what a caller may select is a named field of pcr_params; the PCR_* switches of sweeps and A/B runs exist only in a
    development build (make DEV=1).  The shipped library does not even import getenv."""
    import subprocess
    r = subprocess.run(["nm", "-D", "--undefined-only", LIB_PATH], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "getenv" not in r.stdout
    p = default_params()
    for f in ("index_no_hints", "ndt_evaluate_repeats", "loam_disable_cache", "record_timeline", "loam_coresident", "loam_clamp_margin_mm", "full_target", "host_optimiser", "host_copy_xyz"):
        assert getattr(p, f) == 0
    assert not hasattr(p, "reserved")


def test_synth_is_deterministic_and_sized():
    w1, m1 = synth.make_map(20000, seed=3)
    w2, m2 = synth.make_map(20000, seed=3)
    assert m1.shape == (20000, 4) and m1.dtype == np.float32
    np.testing.assert_array_equal(m1, m2)
    s1, T1 = synth.make_scan(w1, 2, seed=3, beams=8, azimuths=128)
    s2, T2 = synth.make_scan(w2, 2, seed=3, beams=8, azimuths=128)
    assert s1.shape == (1024, 4)
    np.testing.assert_array_equal(s1, s2)
    np.testing.assert_array_equal(T1, T2)
    # scan points, mapped with the true pose, lie on the mapped surfaces
    tree = oracle.KdTree(m1)
    q = s1[:, :3].astype(np.float64) @ T1[:3, :3].T + T1[:3, 3]
    _, d2 = tree.knn(q, 1)
    assert np.median(np.sqrt(d2[:, 0])) < 0.45


def test_a_far_outlier_goes_to_an_edge_rank_and_leaves_the_cuts_alone():
    """The cuts are point-count quantiles, so one stray point kilometres off moves no cut by more than a lattice step; the outer tiles are open, so the
    stray point belongs to exactly one (edge) rank -- whose library then indexes the bulk of its cloud (tests/test_shard_gpu.py)."""
    _, m = synth.make_map(30000, seed=11)
    far = m.copy()
    far[0, :3] = [2.0e4, -1.0e4, 2857.0]
    for method, res in (("loam", 1.0), ("ndt", 1.0), ("vgicp", 0.5)):
        plain = [shard.tile_for_method(m, r, 4, method, res) for r in range(4)]
        tiles = [shard.tile_for_method(far, r, 4, method, res) for r in range(4)]
        assert sum(t.n_core for t in tiles) == far.shape[0]
        holders = [r for r, t in enumerate(tiles) if (t.points[:, 0] >= 1.0e4).any()]
        assert len(holders) == 1 and holders[0] in (0, 3), (method, holders)
        if plain[0].axis == tiles[0].axis:      # (the stray point may make another axis the longest: then the cuts are another axis's)
            for a, b in zip(plain[:-1], tiles[:-1]):
                assert abs(a.hi[a.axis] - b.hi[b.axis]) <= 2 * res, (method, a.hi, b.hi)


def test_tiles_partition_queries_exactly_once():
    _, m = synth.make_map(30000, seed=9)
    for ws in (2, 3, 8):
        tiles = [shard.tile_for_rank(m, r, ws) for r in range(ws)]
        assert sum(t.n_core for t in tiles) == m.shape[0]
        q = np.random.default_rng(0).uniform(-200, 200, (5000, 3))
        owner = np.zeros(len(q), int)
        for t in tiles:
            owner += ((q >= t.lo) & (q < t.hi)).all(1)
        assert (owner == 1).all()
        # halo: every map point within 1 m of a tile's core region is in that tile's cloud
        for t in tiles:
            c = m[:, t.axis].astype(np.float64)
            need = (c >= t.lo[t.axis] - 1.0) & (c < t.hi[t.axis] + 1.0)
            assert need.sum() == t.points.shape[0]


def _shard_worker(rank, world_size, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    world, m = synth.make_map(30000, seed=5)
    scan, T = synth.make_scan(world, 0, seed=5, beams=16, azimuths=256)
    pose = synth.perturb(T, 5, trans=0.2, rot_deg=1.0)
    tile = shard.tile_for_rank(m, rank, world_size)
    tree = oracle.KdTree(tile.points)
    # this rank's share: scan points whose (f32-rounded) transformed position lies in its tile
    qpos = (scan[:, :3].astype(np.float64) @ pose[:3, :3].T + pose[:3, 3]).astype(np.float32).astype(np.float64)
    mine = ((qpos >= tile.lo) & (qpos < tile.hi)).all(1)
    part = oracle.loam_linearize(tree, scan[mine], pose)
    buf = torch.from_numpy(np.concatenate([part["JtJ"].ravel(), part["JtE"], [part["n"]]]))
    dist.all_reduce(buf)                      # the RCCL all-reduce of the GPU path, on gloo here
    if rank == 0:
        q.put((buf.numpy().copy(), int(mine.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_normal_equations_gloo_world_size_2():
    """Tile + 1 m halo per rank, each scan point owned by exactly one rank, one all-reduce:
    the sum equals the single-process normal equations."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_shard_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    buf, n0 = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    world, m = synth.make_map(30000, seed=5)
    scan, T = synth.make_scan(world, 0, seed=5, beams=16, azimuths=256)
    pose = synth.perturb(T, 5, trans=0.2, rot_deg=1.0)
    full = oracle.loam_linearize(oracle.KdTree(m), scan, pose)
    assert 0 < n0 < scan.shape[0]
    assert int(round(buf[42])) == full["n"]
    np.testing.assert_allclose(buf[:36].reshape(6, 6), full["JtJ"], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(buf[36:42], full["JtE"], rtol=1e-10, atol=1e-12)


def _rot_xyz_f32(p6):
    """Translation * Rx * Ry * Rz in float (ndt_omp_impl.hpp:146-149), as pcl::transformPointCloud applies it."""
    from scipy.spatial.transform import Rotation as Rot
    R = Rot.from_euler("XYZ", np.asarray(p6[3:], np.float64)).as_matrix().astype(np.float32)
    return R, np.asarray(p6[:3], np.float32)


def _ndt_vgicp_worker(rank, world_size, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    # ---- NDT: score + gradient + Hessian of computeDerivatives (43 values, ndt_omp_impl.hpp:206-285) ----
    world, m = synth.make_map(60000, seed=11, spacing=0.2)
    scan, T = synth.make_scan(world, 0, seed=11, beams=16, azimuths=256)
    from scipy.spatial.transform import Rotation as Rot
    p6 = np.concatenate([T[:3, 3] + [0.05, -0.03, 0.02], Rot.from_matrix(T[:3, :3]).as_euler("XYZ") + [0.002, -0.001, 0.004]])
    tile = shard.tile_for_method(m, rank, world_size, "ndt", 1.0, halo=2.0)
    R, t = _rot_xyz_f32(p6)
    tp = (scan[:, :3] @ R.T + t).astype(np.float64)
    mine = ((tp >= tile.lo) & (tp < tile.hi)).all(1)
    d = oracle.ndt_derivatives(scan[mine], tile.points, p6)
    buf = torch.from_numpy(np.concatenate([[d["score"]], d["grad"], d["hess"].ravel()]))
    dist.all_reduce(buf)
    # ---- VGICP: H, b, error of linearize (fast_vgicp_impl.hpp:119-180); tiles on the voxel lattice, covariances of the
    #      tile's own cloud (the halo holds every 20-neighbourhood) ----
    world2, m2 = synth.make_map(40000, seed=12)
    scan2, T2 = synth.make_scan(world2, 0, seed=12, beams=16, azimuths=256)
    pose = synth.perturb(T2, 12, trans=0.2, rot_deg=1.0)
    tile2 = shard.tile_for_method(m2, rank, world_size, "vgicp", 1.0, halo=6.0)
    tq = scan2[:, :3].astype(np.float64) @ pose[:3, :3].T + pose[:3, 3]
    mine2 = ((tq >= tile2.lo) & (tq < tile2.hi)).all(1)
    sc = oracle.vgicp_covariances(scan2, 20, 2)                 # source covariances come from the WHOLE scan on every rank
    dc = oracle.vgicp_covariances(tile2.points, 20, 2)
    v = oracle.vgicp_linearize(scan2[mine2], tile2.points, pose, sc[mine2], dc)
    buf2 = torch.from_numpy(np.concatenate([v["H"].ravel(), v["b"], [v["err"], float(v["n"])]]))
    dist.all_reduce(buf2)
    if rank == 0:
        q.put((buf.numpy().copy(), int(mine.sum()), buf2.numpy().copy(), int(mine2.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_ndt_and_vgicp_sums_gloo_world_size_2():
    """SURVEY 8(e) partition row for the other two methods, with the oracle standing in for the kernels: NDT tiles cut on voxel
    faces + a voxel halo, VGICP tiles cut on the voxel lattice + a neighbourhood halo, every source point owned once, one
    all-reduce of 43 doubles: the sums equal the single-process ones."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ndt_vgicp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    buf, n0, buf2, n02 = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    from scipy.spatial.transform import Rotation as Rot
    world, m = synth.make_map(60000, seed=11, spacing=0.2)
    scan, T = synth.make_scan(world, 0, seed=11, beams=16, azimuths=256)
    p6 = np.concatenate([T[:3, 3] + [0.05, -0.03, 0.02], Rot.from_matrix(T[:3, :3]).as_euler("XYZ") + [0.002, -0.001, 0.004]])
    full = oracle.ndt_derivatives(scan, m, p6)
    assert 0 < n0 < scan.shape[0]
    assert abs(buf[0] - full["score"]) <= 1e-9 * abs(full["score"]) and abs(full["score"]) > 1.0
    np.testing.assert_allclose(buf[1:7], full["grad"], rtol=1e-9, atol=1e-9 * np.abs(full["grad"]).max())
    np.testing.assert_allclose(buf[7:].reshape(6, 6), full["hess"], rtol=1e-9, atol=1e-9 * np.abs(full["hess"]).max())
    world2, m2 = synth.make_map(40000, seed=12)
    scan2, T2 = synth.make_scan(world2, 0, seed=12, beams=16, azimuths=256)
    pose = synth.perturb(T2, 12, trans=0.2, rot_deg=1.0)
    sc, dc = oracle.vgicp_covariances(scan2, 20, 4), oracle.vgicp_covariances(m2, 20, 4)
    v = oracle.vgicp_linearize(scan2, m2, pose, sc, dc)
    assert 0 < n02 < scan2.shape[0]
    assert int(round(buf2[43])) == v["n"] and v["n"] > 500
    np.testing.assert_allclose(buf2[:36].reshape(6, 6), v["H"], rtol=1e-10, atol=1e-10 * np.abs(v["H"]).max())
    np.testing.assert_allclose(buf2[36:42], v["b"], rtol=1e-9, atol=1e-10 * np.abs(v["b"]).max())
    np.testing.assert_allclose(buf2[42], v["err"], rtol=1e-10)


def test_tiles_for_ndt_and_vgicp_sit_on_the_voxel_lattice():
    _, m = synth.make_map(30000, seed=9)
    for method, res, shift in (("ndt", 1.0, 0.0), ("ndt", 0.8, 0.0), ("vgicp", 0.5, 0.5), ("vgicp", 1.0, 0.5)):
        tiles = [shard.tile_for_method(m, r, 4, method, res) for r in range(4)]
        assert sum(t.n_core for t in tiles) == m.shape[0]
        for t in tiles:
            for v in (t.lo[t.axis], t.hi[t.axis]):
                if abs(v) < 1e29:
                    k = v / float(np.float32(res) if method == "ndt" else res) - shift
                    assert abs(k - round(k)) < 1e-9
            c = m[:, t.axis].astype(np.float64)
            assert ((c >= t.lo[t.axis] - t.halo) & (c < t.hi[t.axis] + t.halo)).sum() == t.points.shape[0]
        assert tiles[0].halo == ({"ndt": {1.0: 1.0, 0.8: 2 * float(np.float32(0.8))}, "vgicp": {0.5: 4.0, 1.0: 8.0}}[method][res])


def test_thread_collective_sums_in_rank_order():
    """The in-process collective the GPU suite shards over: same bits on every rank, fixed order of additions."""
    import threading
    n = 5
    coll = shard.ThreadCollective(n)
    vals = [np.array([0.1 * (r + 1), 1e16 if r == 2 else 1.0, float(r)]) for r in range(n)]
    outs, mx = [None] * n, [None] * n

    def work(r):
        f = coll.fn(r)
        a = vals[r].copy()
        assert f(a.ctypes.data_as(C.POINTER(C.c_double)), 3, 0, None) == 0
        outs[r] = a
        b = vals[r].copy()
        assert f(b.ctypes.data_as(C.POINTER(C.c_double)), 3, 1, None) == 0
        mx[r] = b
    th = [threading.Thread(target=work, args=(r,)) for r in range(n)]
    for t in th: t.start()
    for t in th: t.join(30)
    expect = vals[0].copy()
    for r in range(1, n):
        expect += vals[r]
    for r in range(n):
        np.testing.assert_array_equal(outs[r], expect)
        np.testing.assert_array_equal(mx[r], np.max(vals, axis=0))


def test_ndt_optimiser_driven_by_the_oracles_derivatives_arrives_where_the_oracle_does():
    """csrc/ndt_opt.h is pclomp's computeTransformation + computeStepLengthMT turned inside out: a state machine that asks for one
    evaluation at a time.  On the GPU it runs in the kernel that folds a pass's sums; here it runs on the host (pcr_ndt_opt_*, no GPU
    involved) and every evaluation it asks for is answered by the ORACLE's computeDerivatives / computeHessian.  If the state machine
    takes the decisions of the reference's loop it must finish after the same number of Newton iterations and evaluations, with the
    same flag and the same Matrix4f pose as the oracle's own loop (oracle/ndt_oracle.c)."""
    import ctypes as C
    from simpleslam_amd import synth
    from simpleslam_amd.pcr import load_library
    L = load_library()
    world, m = synth.make_map(60_000, seed=77, spacing=0.2)
    scan, T = synth.make_scan(world, 0, seed=77, beams=16, azimuths=256)
    dp = C.POINTER(C.c_double)
    prm = oracle.ndt_params()
    replayed_total = 0
    for seed, tr, rd in ((1, 0.1, 0.5), (2, 0.3, 1.5), (3, 0.0, 0.0)):
        T0 = synth.perturb(T, seed, trans=tr, rot_deg=rd) if tr else T.copy()
        po, co, info = oracle.ndt_scan2map(scan, m, T0, prm)
        guess = np.ascontiguousarray(T0.T).reshape(16).copy()
        o = L.pcr_ndt_opt_create(guess.ctypes.data_as(dp), float(prm.step_size), float(prm.trans_eps), int(prm.max_iters))
        assert o
        try:
            n_deriv = n_hess = n_req = with_h = 0
            last = None
            for _ in range(600):
                kind, p6 = C.c_int(-1), np.zeros(6)
                assert L.pcr_ndt_opt_request(o, C.byref(kind), p6.ctypes.data_as(dp), None) == 0
                if kind.value == 3:
                    break
                d = oracle.ndt_derivatives(scan, m, p6, prm, double_hessian=(kind.value == 2))
                sums = np.zeros(43)
                sums[0] = d["score"]; sums[1:7] = d["grad"]
                sums[7:] = (d["hess_d"] if kind.value == 2 else d["hess"]).reshape(36)
                n_hess += kind.value == 2
                # The first trial point of a line search is asked for WITHOUT its float Hessian (the reference computes it there and drops
                # it as soon as the search goes on); a search that ends at that point asks for the Hessian afterwards, at the SAME point:
                # that request is not one of the reference's evaluations.
                late = kind.value == 0 and n_req > 0
                if late:
                    assert last is not None and last[0] == 1 and np.array_equal(last[1], p6)
                n_deriv += kind.value != 2 and not late
                with_h += kind.value == 0
                n_req += 1
                last = (kind.value, p6.copy())
                assert L.pcr_ndt_opt_feed(o, sums.ctypes.data_as(dp)) == 0
            else:
                raise AssertionError("the optimiser did not finish")
            pose, conv, its, done = np.zeros(16), C.c_int(0), C.c_int(0), C.c_int(0)
            assert L.pcr_ndt_opt_result(o, pose.ctypes.data_as(dp), C.byref(conv), C.byref(its), C.byref(done)) == 0 and done.value == 1
            ev, hs, rep = C.c_int(0), C.c_int(0), C.c_int(0)
            assert L.pcr_ndt_opt_counts(o, C.byref(ev), C.byref(hs), C.byref(rep)) == 0
            # what the reference evaluates = what was asked for + what the state machine answered itself (a clamped trial step repeated)
            assert ev.value == n_deriv + rep.value and hs.value == n_hess
            assert with_h < info["iterations"] + 1      # fewer passes carry a float Hessian than the reference's one per Newton iteration + 1
            n_deriv, replayed_total = ev.value, replayed_total + rep.value
        finally:
            L.pcr_ndt_opt_destroy(o)
        assert bool(conv.value) == co, seed
        assert (its.value, n_deriv, n_hess) == (info["iterations"], info["derivative_passes"], info["hessian_passes"]), (seed, its.value, n_deriv, n_hess, info)
        # same decisions; the pose agrees to a few float ulps (the stand-alone derivative entry point of the oracle rebuilds the float
        # transform from p6 on its own, so the sums fed here are not bit for bit those of the oracle's inner loop)
        np.testing.assert_allclose(pose.reshape(4, 4).T, po, rtol=0, atol=5e-6)
    assert replayed_total > 0      # (the clamped More-Thuente steps of these cases do repeat)


def test_vgicp_optimiser_driven_by_the_oracles_sums_arrives_where_the_oracle_does():
    """csrc/vgicp_opt.h is fast_gicp's computeTransformation + step_lm turned inside out: a state machine that asks for one pass at a
    time -- a linearisation, or an LM trial (the error on the last linearisation's correspondences plus the linearisation at the trial
    pose).  On the GPU it runs in the prologue of the next pass's launch; here it runs on the host (pcr_vgicp_opt_*, no GPU involved) and
    every pass is answered by the ORACLE's linearize / compute_error.  It must finish after the same outer iterations and error
    evaluations, with the same flag and pose as the oracle's own loop (oracle/vgicp_oracle.c) -- bit for bit: same sums, same arithmetic."""
    import ctypes as C
    from simpleslam_amd import synth
    from simpleslam_amd.pcr import load_library
    L = load_library()
    world, m = synth.make_map(20_000, seed=78)
    scan, T = synth.make_scan(world, 0, seed=78, beams=16, azimuths=128)
    dp = C.POINTER(C.c_double)
    prm = oracle.vgicp_params(threads=4)
    sc, dc = oracle.vgicp_covariances(scan, 20, 4), oracle.vgicp_covariances(m, 20, 4)
    iu = np.triu_indices(6)
    for seed, tr, rd, cap in ((1, 0.3, 2.0, None), (2, 0.8, 5.0, None), (3, 0.0, 0.0, None), (4, 0.3, 2.0, 2)):
        T0 = synth.perturb(T, seed, trans=tr, rot_deg=rd) if tr else T.copy()
        p = oracle.vgicp_params(threads=4)
        if cap: p.max_iters = cap
        po, co, info = oracle.vgicp_scan2map(scan, m, T0, p, sc, dc)
        guess = np.ascontiguousarray(T0.T).reshape(16).copy()
        o = L.pcr_vgicp_opt_create(guess.ctypes.data_as(dp), int(p.max_iters), int(p.lm_inner), float(p.lm_init), float(p.rot_eps), float(p.trans_eps))
        assert o
        try:
            n_err = 0
            for _ in range(2000):
                kind, pe, pl = C.c_int(-1), np.zeros(16), np.zeros(16)
                assert L.pcr_vgicp_opt_request(o, C.byref(kind), pe.ctypes.data_as(dp), pl.ctypes.data_as(dp)) == 0
                if kind.value == 2:
                    break
                Te, Tl = pe.reshape(4, 4).T, pl.reshape(4, 4).T
                lin = oracle.vgicp_linearize(scan, m, Te, sc, dc, p)
                sums = np.zeros(29)
                sums[:21] = lin["H"][iu]; sums[21:27] = lin["b"]; sums[27] = lin["err"]
                if kind.value == 1:
                    sums[28] = oracle.vgicp_error(scan, m, Tl, Te, sc, dc, p)
                    n_err += 1
                assert L.pcr_vgicp_opt_feed(o, sums.ctypes.data_as(dp)) == 0
            else:
                raise AssertionError("the optimiser did not finish")
            pose, conv, outer, done = np.zeros(16), C.c_int(0), C.c_int(0), C.c_int(0)
            assert L.pcr_vgicp_opt_result(o, pose.ctypes.data_as(dp), C.byref(conv), C.byref(outer), C.byref(done)) == 0 and done.value == 1
        finally:
            L.pcr_vgicp_opt_destroy(o)
        assert bool(conv.value) == co, seed
        assert (outer.value, n_err) == (info["outer"], info["error_evals"]), (seed, outer.value, n_err, info)
        got = pose.reshape(4, 4).T.astype(np.float32).astype(np.float64)      # final_transformation_ is a Matrix4f
        np.testing.assert_array_equal(got, po)

"""The ROS-free C++ harness (counterpart of the reference's test/loc.cpp pass) over the header-only
mirror of the plugin interface: same answer as the Python host mirror, through the same C ABI."""
import os
import subprocess

import numpy as np
import pytest

from simpleslam_amd import LoamRegister

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_harness_matches_python(gpu, world_small, tmp_path):
    exe = os.path.join(ROOT, "simpleslam_amd", "lib", "loc_harness")
    assert os.path.exists(exe), "loc_harness not built (run __graft_entry__.build())"
    w = world_small
    w["map"].astype(np.float32).tofile(tmp_path / "map.f32")
    w["scan"].astype(np.float32).tofile(tmp_path / "scan.f32")
    np.savetxt(tmp_path / "init.txt", w["init"], fmt="%.17g")
    out = subprocess.run([exe, "loam", str(tmp_path / "map.f32"), str(tmp_path / "scan.f32"), str(tmp_path / "init.txt")],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    pose_cpp = np.array([[float(v) for v in ln.split()] for ln in lines[-4:]])
    pose_py = w["init"].copy()
    conv = LoamRegister().scan2Map(w["scan"], w["map"], pose_py)
    assert f"converged {int(conv)}" in lines[0]
    np.testing.assert_array_equal(pose_cpp, pose_py)
    bad = subprocess.run([exe, "icp", "a", "b", "c"], capture_output=True, text=True)
    assert bad.returncode == 1 and "is not exist" in bad.stderr


def test_cpp_harness_with_voxel_downsampling(gpu, world_small, tmp_path):
    """LidarOdometry's flow, scan -> VoxelGrid(downSampleVoxelGridSize) -> scan2Map (LidarOdometry.cpp:36,170-184),
    through the C++ mirror: same pose as the Python mirror fed with the same down-sampled scan."""
    exe = os.path.join(ROOT, "simpleslam_amd", "lib", "loc_harness")
    w = world_small
    w["map"].astype(np.float32).tofile(tmp_path / "map.f32")
    w["scan"].astype(np.float32).tofile(tmp_path / "scan.f32")
    np.savetxt(tmp_path / "init.txt", w["init"], fmt="%.17g")
    out = subprocess.run([exe, "loam", str(tmp_path / "map.f32"), str(tmp_path / "scan.f32"), str(tmp_path / "init.txt"), "0.4"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    reg = LoamRegister()
    ds = reg.voxelDownSample(w["scan"], 0.4)
    assert lines[0] == f"voxel ds {w['scan'].shape[0]} -> {ds.shape[0]}"
    pose_py = w["init"].copy()
    reg.scan2Map(ds, w["map"], pose_py)
    pose_cpp = np.array([[float(v) for v in ln.split()] for ln in lines[-4:]])
    np.testing.assert_array_equal(pose_cpp, pose_py)


def test_cpp_harness_submap_mode(gpu, tmp_path):
    """The C++ mirror of MapManager's flow: key frames -> device sub-map around the current position -> down-sampled scan
    registered against it; same pose as the Python mirror."""
    import oracle
    from simpleslam_amd import SubMap, synth
    exe = os.path.join(ROOT, "simpleslam_amd", "lib", "loc_harness")
    world, _ = synth.make_map(20_000, seed=91)
    lines, kfs = [], []
    for j in range(8):
        scan, T = synth.make_scan(world, j, seed=91, beams=32, azimuths=512)
        ds, _ = oracle.voxel_filter(scan, 0.4)
        f = tmp_path / f"kf{j}.f32"
        ds.astype(np.float32).tofile(f)
        lines.append(str(f) + " " + " ".join(f"{v:.17g}" for v in T.reshape(-1)))
        kfs.append((ds, T))
    (tmp_path / "kfs.txt").write_text("\n".join(lines) + "\n")
    scan, T_true = synth.make_scan(world, 8, seed=91, beams=32, azimuths=512)
    init = synth.perturb(T_true, 91, trans=0.1, rot_deg=0.5)
    scan.astype(np.float32).tofile(tmp_path / "scan.f32")
    np.savetxt(tmp_path / "init.txt", init, fmt="%.17g")
    out = subprocess.run([exe, "loam", "submap:" + str(tmp_path / "kfs.txt"), str(tmp_path / "scan.f32"), str(tmp_path / "init.txt"), "0.4"],
                         capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    pose_cpp = np.array([[float(v) for v in ln.split()] for ln in lines[-4:]])
    sm = SubMap()
    for c, T in kfs:
        sm.addKeyFrame(c, T)
    n_sub = sm.updateMap(init[:3, 3], radius=8.0, grid_size=0.4)
    reg = LoamRegister()
    ds = reg.voxelDownSample(scan, 0.4)
    assert f"submap {n_sub} " in lines[0] and f"-> {ds.shape[0]} " in lines[0]
    pose_py = init.copy()
    reg.scan2MapSubmap(ds, sm, pose_py)
    np.testing.assert_array_equal(pose_cpp, pose_py)


@pytest.mark.parametrize("method", ["loam", "vgicp", "ndt"])
def test_align_harness_is_the_reference_align_cpp_flow(gpu, world_small, tmp_path, method):
    """test/align.cpp: both clouds voxel-filtered at 0.1 m (:128-129), one scan2Map (:144), the gated fitness
    score of :29-61, the final 4x4 -- through the C++ mirror, same numbers as the Python mirror doing the same steps."""
    from simpleslam_amd import make_register
    exe = os.path.join(ROOT, "simpleslam_amd", "lib", "align_harness")
    assert os.path.exists(exe), "align_harness not built (run __graft_entry__.build())"
    w = world_small
    w["map"].astype(np.float32).tofile(tmp_path / "target.f32")
    w["scan"].astype(np.float32).tofile(tmp_path / "source.f32")
    np.savetxt(tmp_path / "init_pose.txt", w["init"], fmt="%.17g")
    out = subprocess.run([exe, str(tmp_path / "target.f32"), str(tmp_path / "source.f32"), method, str(tmp_path / "init_pose.txt")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = out.stdout.strip().splitlines()
    pose_cpp = np.array([[float(v) for v in ln.split()] for ln in lines[-4:]])
    reg = make_register(method)
    tgt, src = reg.voxelDownSample(w["map"], 0.1), reg.voxelDownSample(w["scan"], 0.1)
    assert f"target cloud size: {tgt.shape[0]}" in lines and f"source cloud size: {src.shape[0]}" in lines
    pose_py = w["init"].copy()
    conv = reg.scan2Map(src, tgt, pose_py)
    assert ("not converge!!" in lines) == (not conv)
    np.testing.assert_array_equal(pose_cpp, pose_py)
    fit, n_in = reg.fitnessGated(src, pose_py, 1.0)
    got = [ln for ln in lines if ln.startswith("get fitness score:")][0]
    assert f"({n_in} points within 1 m)" in got and abs(float(got.split()[3]) - fit) <= 1e-8 * max(1.0, abs(fit))
    usage = subprocess.run([exe], capture_output=True, text=True)
    assert usage.returncode == 0 and "usage: align" in usage.stderr            # align.cpp:66-69 returns 0 after the usage line
    bad = subprocess.run([exe, str(tmp_path / "target.f32"), str(tmp_path / "source.f32"), "icp"], capture_output=True, text=True)
    assert bad.returncode != 0 and "no such method!!" in bad.stderr


def test_kdtree_bench_reports_build_and_query_times(gpu, world_small, tmp_path):
    """Counterpart of test/benchmark/kdtree.cpp:58-127 for the grid index: index build seconds and ns per exact 5-NN query."""
    exe = os.path.join(ROOT, "simpleslam_amd", "lib", "kdtree_bench")
    assert os.path.exists(exe), "kdtree_bench not built (run __graft_entry__.build())"
    w = world_small
    w["map"].astype(np.float32).tofile(tmp_path / "map.f32")
    out = subprocess.run([exe, str(tmp_path / "map.f32"), str(tmp_path / "map.f32"), "5"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    a, b = out.stdout.strip().splitlines()
    assert a.startswith(f"grid index: {w['map'].shape[0]} points, build") and float(a.split()[-2]) > 0
    assert f"{w['map'].shape[0]} queries" in b and "ns/query" in b
    assert int(b.split()[-4]) > 0.8 * w["map"].shape[0]                         # most map points find 5 neighbours within 1 m among the map


def _pose_of(stdout):
    return np.array([[float(v) for v in ln.split()] for ln in stdout.strip().splitlines()[-4:]])


def test_loc_harness_reads_params_json_and_pcd_at_config1_sizes(gpu, world_100k, tmp_path):
    """`loc_harness params.json scan.pcd init.txt` -- what test/loc.cpp reads (cores, downSampleVoxelGridSize, pcd_file, frontend.pcr;
    a PCD map, MapManager.cpp:68) -- at BASELINE configs[0]'s sizes (65 536 x 100 k, pcr = loam, cores = 1): the pose of the Python
    mirror bit for bit, the oracle's within the north-star tolerance."""
    import oracle
    from simpleslam_amd import synth
    from tests import loc_inputs
    exe = os.path.join(ROOT, "simpleslam_amd", "lib", "loc_harness")
    w = world_100k
    loc_inputs.write_pcd(tmp_path / "map.pcd", w["map"], "binary_pcl")           # the padded 32-byte records PCL itself writes
    loc_inputs.write_pcd(tmp_path / "scan.pcd", w["scan"], "binary")
    loc_inputs.write_params(tmp_path / "params.json", tmp_path / "map.pcd", pcr="loam", cores=1, grid=0.5)
    np.savetxt(tmp_path / "init.txt", w["init"], fmt="%.17g")
    out = subprocess.run([exe, str(tmp_path / "params.json"), str(tmp_path / "scan.pcd"), str(tmp_path / "init.txt"), "--no-downsample"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    first = out.stdout.strip().splitlines()[0]
    assert first.startswith("pcr loam  cores 1  grid 0.5  map 100000 -> 100000  scan 65536 -> 65536  converged 1"), first
    pose_cpp = _pose_of(out.stdout)
    pose_py = w["init"].copy()
    assert LoamRegister().scan2Map(w["scan"], w["map"], pose_py)
    np.testing.assert_array_equal(pose_cpp, pose_py)
    ref, conv, _ = oracle.loam_scan2map(w["scan"], w["map"], w["init"], oracle.loam_params(threads=1))
    dt, dr = synth.pose_error(pose_cpp, ref)
    assert conv and dt <= 1e-4 and dr <= 1e-4, (dt, dr)


@pytest.mark.parametrize("method", ["loam", "ndt", "vgicp"])
def test_loc_harness_downsamples_like_the_frontend(gpu, world_small, tmp_path, method):
    """the default flow of test/loc.cpp: map voxel-filtered at downSampleVoxelGridSize when it is loaded (MapManager.cpp:78), the scan
    before every registration (LidarOdometry.cpp:170-171), the registrar chosen by frontend.pcr (LidarOdometry.cpp:44-54)"""
    from simpleslam_amd import make_register
    from tests import loc_inputs
    exe = os.path.join(ROOT, "simpleslam_amd", "lib", "loc_harness")
    w = world_small
    loc_inputs.write_pcd(tmp_path / "map.pcd", w["map"], "ascii")
    loc_inputs.write_pcd(tmp_path / "scan.pcd", w["scan"], "binary_compressed")
    loc_inputs.write_params(tmp_path / "params.json", tmp_path / "map.pcd", pcr=method, cores=4, grid=0.4)
    np.savetxt(tmp_path / "init.txt", w["init"], fmt="%.17g")
    out = subprocess.run([exe, str(tmp_path / "params.json"), str(tmp_path / "scan.pcd"), str(tmp_path / "init.txt")],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    reg = make_register(method)
    # (the ascii PCD holds %.9g decimals: they read back to the same floats)
    m_ds, s_ds = reg.voxelDownSample(w["map"][:, :4].copy(), 0.4), reg.voxelDownSample(w["scan"][:, :4].copy(), 0.4)
    first = out.stdout.strip().splitlines()[0]
    assert f"pcr {method}  cores 4" in first and f"map {w['map'].shape[0]} -> {m_ds.shape[0]}  scan {w['scan'].shape[0]} -> {s_ds.shape[0]}" in first, first
    # the harness hands 32-byte pcl::PointXYZI records over, the arrays here are 16-byte rows: same coordinates, same pose
    pose_py = w["init"].copy()
    conv = reg.scan2Map(s_ds, m_ds, pose_py)
    assert f"converged {int(conv)}" in first
    np.testing.assert_array_equal(_pose_of(out.stdout), pose_py)


def test_loc_harness_refuses_an_unknown_pcr_and_a_missing_map(gpu, world_small, tmp_path):
    from tests import loc_inputs
    exe = os.path.join(ROOT, "simpleslam_amd", "lib", "loc_harness")
    w = world_small
    loc_inputs.write_pcd(tmp_path / "scan.pcd", w["scan"], "binary")
    np.savetxt(tmp_path / "init.txt", w["init"], fmt="%.17g")
    loc_inputs.write_params(tmp_path / "p1.json", tmp_path / "scan.pcd", pcr="icp")
    bad = subprocess.run([exe, str(tmp_path / "p1.json"), str(tmp_path / "scan.pcd"), str(tmp_path / "init.txt")], capture_output=True, text=True)
    assert bad.returncode == 1 and "such pcr type(icp) is not exist" in bad.stderr            # LidarOdometry.cpp:50-54
    loc_inputs.write_params(tmp_path / "p2.json", tmp_path / "nowhere.pcd", pcr="loam")
    bad = subprocess.run([exe, str(tmp_path / "p2.json"), str(tmp_path / "scan.pcd"), str(tmp_path / "init.txt")], capture_output=True, text=True)
    assert bad.returncode == 1 and "can't load globalmap from" in bad.stderr                   # MapManager.cpp:68-73


@pytest.mark.parametrize("method", ["loam", "ndt", "vgicp"])
def test_static_map_adapter_registers_eight_scans_against_one_pcd_map(gpu, world_small, tmp_path, method):
    """test/loc.cpp's loop: the map is loaded once from a PCD (MapManager.cpp:52-78), scan after scan is localised against it.
    PCR::StaticMapRegister (host/PCR/HipRegister.hpp) keeps the unchanged scan2Map(src, dst, res) interface, indexes the map at the first
    call (pcr_set_target) and aligns the others against what the device holds (pcr_align): eight scans, each pose equal -- bit for bit --
    to a fresh plain registrar's scan2Map and to the Python mirror's."""
    from simpleslam_amd import make_register, synth
    from tests import loc_inputs
    exe = os.path.join(ROOT, "simpleslam_amd", "lib", "loc_harness")
    w = world_small
    loc_inputs.write_pcd(tmp_path / "map.pcd", w["map"], "binary_pcl")
    loc_inputs.write_params(tmp_path / "params.json", tmp_path / "map.pcd", pcr=method, cores=1, grid=0.5)
    scans, inits = [], []
    for k in range(8):
        sc, T = synth.make_scan(w["world"], k % 3, seed=77 + k, beams=16, azimuths=256)
        scans.append(sc); inits.append(synth.perturb(T, 77 + k, trans=0.15, rot_deg=0.8))
        loc_inputs.write_pcd(tmp_path / f"scan{k}.pcd", sc, "binary")
        np.savetxt(tmp_path / f"init{k}.txt", inits[k], fmt="%.17g")
    with open(tmp_path / "more.txt", "w") as f:
        for k in range(1, 8):
            f.write(f"{tmp_path / f'scan{k}.pcd'} {tmp_path / f'init{k}.txt'}\n")
    out = subprocess.run([exe, str(tmp_path / "params.json"), str(tmp_path / "scan0.pcd"), str(tmp_path / "init0.txt"), "--no-downsample",
                          "--static", str(tmp_path / "more.txt")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = out.stdout.strip().splitlines()
    assert lines[0].startswith(f"pcr {method}  static map") and lines[0].endswith("scans 8"), lines[0]
    assert lines[-1].endswith("differing 0"), lines[-1]
    assert sum("same pose" in ln for ln in lines) == 8
    reg = make_register(method)
    for k in range(8):
        rows = lines[2 + 5 * k: 6 + 5 * k]
        pose_cpp = np.array([[float(x) for x in r.split()] for r in rows])
        pose_py = inits[k].copy()
        reg.scan2Map(scans[k], w["map"], pose_py)
        np.testing.assert_array_equal(pose_cpp, pose_py, err_msg=f"scan {k}")

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

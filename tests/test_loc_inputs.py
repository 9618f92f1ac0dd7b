"""The harness's own readers (simpleslam_amd/host/config/params.hpp, pcp/pcd_io.hpp) on the inputs test/loc.cpp reads, and BASELINE
config 1 -- "test/loc.cpp CPU path: 65 k-pt scan, 100 k-pt pcd submap, pcr=loam, cores=1 (plumbing, no GPU)" -- through the oracle
from those same files.  No GPU anywhere in this file (tests/test_harness_gpu.py runs the C++ harness on them)."""
import os
import subprocess

import numpy as np
import pytest

import oracle
from simpleslam_amd import synth
from tests import loc_inputs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
IOCHK = os.path.join(ROOT, "simpleslam_amd", "lib", "io_check")


def _run(*args):
    return subprocess.run([IOCHK, *map(str, args)], capture_output=True, text=True, timeout=120)


def _points(stdout):
    lines = stdout.strip().splitlines()
    n = int(lines[0].split()[1])
    a = np.array([[float(v) for v in ln.split()] for ln in lines[1:]], np.float32).reshape(-1, 4)
    return n, a


def test_params_json_with_comments(tmp_path):
    loc_inputs.write_params(tmp_path / "params.json", "/data/maps/hqc.pcd", pcr="vgicp", cores=4, grid=0.5)
    r = _run("--params", tmp_path / "params.json")
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == ["cores 4", "downSampleVoxelGridSize 0.5", "pcd_file /data/maps/hqc.pcd", "frontend.pcr vgicp"]
    ref = loc_inputs.read_params(tmp_path / "params.json")
    assert ref["cores"] == 4 and ref["frontend"]["pcr"] == "vgicp" and ref["downSampleVoxelGridSize"] == 0.5


def test_params_reads_the_reference_shaped_file(tmp_path):
    """every construct config/params.json uses: // comments after values and on their own lines, nested objects, booleans"""
    text = '''{
    // mode has lio or lo
    "mode": "lio",
    "cores": 4,
    // use ndt maybe no need to downsample, or sample rate should be small
    "downSampleVoxelGridSize": 0.5, 
    "pcd_file": "/home/gy/.robot/data/maps/hqc/hqc.pcd",
    // "rosbag": "/home/hgy/a // not a comment inside a comment",
    "dataproxy": { "lidar": "/lidar_points", "lidar_size": 10
        // "wheel": "/husky_velocity_controller/odom"
    },
    "vis": { "enable" : true, "align": "/aligned" },
    "backend" : { "lc": { "enable" : false, "fitnessThreshold" : 0.3 } },
    "frontend" : {
        "pcr" : "loam",     // loam, ndt or vgicp
        "local_size": 100
    }
}
'''
    (tmp_path / "p.json").write_text(text)
    r = _run("--params", tmp_path / "p.json")
    assert r.returncode == 0, r.stderr
    assert r.stdout.splitlines() == ["cores 4", "downSampleVoxelGridSize 0.5", "pcd_file /home/gy/.robot/data/maps/hqc/hqc.pcd", "frontend.pcr loam"]


@pytest.mark.parametrize("text,what", [('{"cores": 4,', "end of input"), ('{"cores" 4}', "':'"), ('{"cores": 4} x', "trailing"),
                                       ('{"downSampleVoxelGridSize": 0.5, "pcd_file": "a", "frontend": {"pcr": "loam"}}', "\"cores\" is missing"),
                                       ('{"cores": "four", "downSampleVoxelGridSize": 0.5, "pcd_file": "a", "frontend": {"pcr": "loam"}}', "not a number")])
def test_params_errors_are_reported(tmp_path, text, what):
    (tmp_path / "bad.json").write_text(text)
    r = _run("--params", tmp_path / "bad.json")
    assert r.returncode == 1 and what in r.stderr, r.stderr


@pytest.mark.parametrize("kind", ["ascii", "binary", "binary_pcl", "binary_compressed"])
def test_pcd_reader(tmp_path, kind):
    rng = np.random.default_rng(7)
    pts = rng.normal(size=(3000, 4)).astype(np.float32) * np.float32(30.0)
    pts[::7, 1] = np.float32(1.5)                 # repeated values: back references in the compressed stream
    pts[5, 0] = np.float32(np.nan)
    loc_inputs.write_pcd(tmp_path / "c.pcd", pts, kind)
    r = _run("--pcd", tmp_path / "c.pcd")
    assert r.returncode == 0, r.stderr
    n, got = _points(r.stdout)
    assert n == 3000
    np.testing.assert_array_equal(np.nan_to_num(got, nan=-1e30), np.nan_to_num(pts, nan=-1e30))


def test_pcd_without_intensity_and_with_doubles(tmp_path):
    """fields in another order, double coordinates, no intensity (PCL leaves the default 0)"""
    xyz = np.arange(30, dtype=np.float64).reshape(10, 3) * 0.25
    head = ("VERSION .7\nFIELDS z y x\nSIZE 8 8 8\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 10\nHEIGHT 1\nPOINTS 10\nDATA binary\n").encode()
    with open(tmp_path / "d.pcd", "wb") as f:
        f.write(head); f.write(np.ascontiguousarray(xyz[:, ::-1]).tobytes())
    r = _run("--pcd", tmp_path / "d.pcd")
    assert r.returncode == 0, r.stderr
    n, got = _points(r.stdout)
    np.testing.assert_array_equal(got[:, :3], xyz.astype(np.float32))
    assert n == 10 and np.all(got[:, 3] == 0)


def test_pcd_errors(tmp_path):
    assert _run("--pcd", tmp_path / "missing.pcd").returncode == 4          # loadPCDFile == -1
    (tmp_path / "short.pcd").write_bytes(b"VERSION .7\nFIELDS x y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 5\nHEIGHT 1\nPOINTS 5\nDATA binary\n" + b"\0" * 24)
    r = _run("--pcd", tmp_path / "short.pcd")
    assert r.returncode == 1 and "fewer than 5" in r.stderr
    (tmp_path / "nox.pcd").write_bytes(b"VERSION .7\nFIELDS a y z\nSIZE 4 4 4\nTYPE F F F\nCOUNT 1 1 1\nWIDTH 1\nHEIGHT 1\nPOINTS 1\nDATA ascii\n1 2 3\n")
    r = _run("--pcd", tmp_path / "nox.pcd")
    assert r.returncode == 1 and "x, y, z are required" in r.stderr


def test_pcd_written_by_the_harness_reads_back(tmp_path):
    pts = np.random.default_rng(3).normal(size=(500, 4)).astype(np.float32)
    loc_inputs.write_pcd(tmp_path / "a.pcd", pts, "binary")
    for kind in ("ascii", "binary"):
        assert _run("--repack", tmp_path / "a.pcd", tmp_path / f"b_{kind}.pcd", kind).returncode == 0
        np.testing.assert_array_equal(loc_inputs.read_pcd(tmp_path / f"b_{kind}.pcd"), pts)


def test_config1_cpu_path_through_the_oracle(tmp_path, world_100k):
    """BASELINE configs[0]: one 65 536-point scan against a 100 k-point PCD sub-map, pcr = loam, cores = 1, no GPU: the files the
    harness reads go through the CPU oracle (the port of LoamRegister::scan2Map) and the registration reaches the true pose."""
    w = world_100k
    loc_inputs.write_pcd(tmp_path / "map.pcd", w["map"], "binary")
    loc_inputs.write_pcd(tmp_path / "scan.pcd", w["scan"], "binary")
    loc_inputs.write_params(tmp_path / "params.json", tmp_path / "map.pcd", pcr="loam", cores=1, grid=0.5)
    np.savetxt(tmp_path / "init.txt", w["init"], fmt="%.17g")
    cfg = loc_inputs.read_params(tmp_path / "params.json")
    assert cfg["frontend"]["pcr"] == "loam" and cfg["cores"] == 1
    m, s = loc_inputs.read_pcd(cfg["pcd_file"]), loc_inputs.read_pcd(tmp_path / "scan.pcd")
    assert m.shape == (100_000, 4) and s.shape == (65_536, 4)
    np.testing.assert_array_equal(m, w["map"][:, :4]); np.testing.assert_array_equal(s, w["scan"][:, :4])
    pose, conv, info = oracle.loam_scan2map(s, m, np.loadtxt(tmp_path / "init.txt"), oracle.loam_params(threads=cfg["cores"]))
    et, er = synth.pose_error(pose, w["truth"])
    assert conv and et < 0.05 and er < 5e-3, (conv, et, er)

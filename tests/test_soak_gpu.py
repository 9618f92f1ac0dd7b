"""A short run of scripts/soak_parity.py (seeded draws with quantised coordinates, duplicated and non-finite points, far-off
starts, tiny clouds) as part of the GPU suite: every method must agree with its oracle on flag, iteration count and pose;
the voxel filter, the sub-map assembly and ScanContext on their outputs."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_soak_all_methods(gpu):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "soak_parity.py"), "all", "24", "5"], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    for name in ("loam", "vgicp", "ndt", "voxel", "submap", "sc"):
        assert f"{name}: 24 cases, 0 mismatching" in r.stdout, r.stdout[-4000:]

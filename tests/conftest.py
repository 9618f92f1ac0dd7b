import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    if not _has_gpu():
        pytest.skip("no GPU in this environment")
    import simpleslam_amd
    simpleslam_amd.load_library()  # fail loudly if the HIP extension is missing
    return True


@pytest.fixture(scope="session")
def world_100k():
    from simpleslam_amd import synth
    world, m = synth.make_map(100_000, seed=20261003 + 1)
    scan, T = synth.make_scan(world, 0, seed=20261003 + 1)
    return dict(world=world, map=m, scan=scan, truth=T, init=synth.perturb(T, 20261003 + 1))


@pytest.fixture(scope="session")
def world_small():
    """8k-point scan against a 30k map: small enough for per-point brute-force checks."""
    from simpleslam_amd import synth
    world, m = synth.make_map(30_000, seed=5)
    scan, T = synth.make_scan(world, 0, seed=5, beams=16, azimuths=512)
    return dict(world=world, map=m, scan=scan, truth=T, init=synth.perturb(T, 5, trans=0.2, rot_deg=1.0))

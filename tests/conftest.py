import os
import sys

import os

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")     # before HIP initialises (see so101_sim_amd/__init__.py)

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def blobs():
    from so101_sim_amd.model import scenes
    raw32, meta = scenes.load_blob("banana", "f32")
    raw64, _ = scenes.load_blob("banana", "f64")
    return dict(f32=raw32, f64=raw64, meta=meta)


@pytest.fixture(scope="session")
def blobs_pen():
    from so101_sim_amd.model import scenes
    raw32, meta = scenes.load_blob("pen", "f32")
    raw64, _ = scenes.load_blob("pen", "f64")
    return dict(f32=raw32, f64=raw64, meta=meta)


@pytest.fixture(scope="session")
def golden():
    import json
    d = os.path.join(ROOT, "tests", "golden")
    out = {}
    for f in os.listdir(d):
        if f.endswith(".json"):
            with open(os.path.join(d, f)) as fh:
                out[f[:-5]] = json.load(fh)
    return out


@pytest.fixture(scope="session")
def hip_lib():
    """Build (cross-compile) the HIP library once per session."""
    from so101_sim_amd import build
    return build.build()

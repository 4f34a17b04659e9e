import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


os.environ.setdefault("L3D_CHECK_POT", "1")     # host pipeline self-check of the merged potential-correspondence lists (tests only)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # the oracle is test infrastructure: build it on demand (gcc only, ~1 s)
    if not os.path.exists(os.path.join(ROOT, "oracle", "libl3d_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])


@pytest.fixture(scope="session")
def oracle_lib():
    import l3d_oracle_pipeline as op
    return op.load_lib()


@pytest.fixture(scope="session")
def gpu_ctx():
    from line3d_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def small_scene():
    """10 views x 300 segments, 6 neighbours: the smallest shape in which the reference semantics
    keep any match (with N=4 a hypothesis can never collect support from two other cameras)."""
    from line3d_amd.synth import make_scene
    return make_scene(10, 300, 6, seed=7)


@pytest.fixture(scope="session")
def small_oracle(small_scene):
    import l3d_oracle_pipeline as op
    return op.run_scene(small_scene, 6)

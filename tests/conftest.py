import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
TESTS = os.path.dirname(os.path.abspath(__file__))
if TESTS not in sys.path:
    sys.path.insert(0, TESTS)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Native pieces: build whatever is missing (incremental; hipcc cross-compiles
    # without a GPU).  The prebuilt .so files travel to the GPU box.
    need = [os.path.join(REPO, "radiative3d_amd", "lib", "libr3d_host.so"),
            os.path.join(REPO, "radiative3d_amd", "lib", "libr3d_hip.so"),
            os.path.join(REPO, "radiative3d_amd", "lib", "libr3d_hip_repro.so"),
            os.path.join(REPO, "oracle", "libr3d_oracle.so"),
            os.path.join(REPO, "oracle", "libr3d_tables_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        subprocess.check_call(["make", "-C", REPO, "-j", "8", "all"])


@pytest.fixture(scope="session")
def models():
    """Small-table (TOA degree 4) builds of the four benchmark models."""
    from radiative3d_amd import Model
    from tests.configs import CONFIGS
    cache = {}

    def get(name, deg=4, extra=()):
        key = (name, deg, tuple(extra))
        if key not in cache:
            cache[key] = Model(list(CONFIGS[name](deg)) + list(extra))
        return cache[key]

    return get


from oracle.check import finals_differ  # noqa: E402,F401  (re-exported: tests import it from here)

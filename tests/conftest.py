import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib

    oracle_lib.lib()
    return oracle_lib


# (config name, frame used by the parity tests)
SMALL_FRAMES = {"C1": (96, 54), "C3": (96, 54), "C4": (64, 64), "C5": (96, 54)}


@pytest.fixture(scope="session")
def worlds():
    """One generated world per benchmark configuration, plus the small reference scenes."""
    from ray_tracer_amd import api, configs

    cache = {}

    def get(name):
        if name not in cache:
            if name in configs.ALL:
                cache[name] = configs.ALL[name].world()
            else:
                earth = configs.load_earth() if name == "Earth" else None
                cache[name] = api.World.generate(name, 10, configs.WORLD_SEED, 0.0, 1.0, earth_rgb=earth)
        return cache[name]

    return get

import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def cs():
    from csmp_pkg import load
    return load()


@pytest.fixture(scope="session")
def golden():
    z = np.load(os.path.join(ROOT, "tests", "golden", "golden_small.npz"), allow_pickle=False)
    cases = {}
    for name in z["names"]:
        name = str(name)
        cases[name] = {k.split(".", 1)[1]: z[k] for k in z.files if k.startswith(name + ".")}
        cases[name]["algo"] = str(cases[name]["algo"])
    return cases


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_c
    oracle_c.build()
    return oracle_c

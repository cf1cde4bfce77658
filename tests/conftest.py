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
    # On the GPU box two HIP runtimes live in the test process: PyTorch's bundled one (the tests use torch for device tensors and
    # streams) and /opt/rocm's, which libcsmp.so links.  Bring PyTorch's up FIRST: initialised late, after libcsmp has worked for
    # a while, it stalled for minutes and then reported "no ROCm-capable device" (seen with `-k` subsets whose first torch user
    # came after ~18 library tests).  Nothing here touches the GPU when none is visible.
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda").item()
    except ImportError:
        pass
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

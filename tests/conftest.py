import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the oracle is OpenMP code: a GPU box exposes hundreds of logical CPUs but grants a 16-core share, and an
# oversubscribed team makes the tiny-model oracle calls crawl
_cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(16, _cores))))
os.environ.setdefault("OMP_WAIT_POLICY", "passive")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


@pytest.fixture(scope="session")
def tk():
    import trackiellm_amd
    trackiellm_amd.lib()
    return trackiellm_amd


@pytest.fixture(scope="session")
def gpu(tk):
    if tk.lib().tk_mi355x_device_count() <= 0:
        pytest.fail("gpu-marked test started without a visible HIP device: the MI355X path has no fallback")
    return tk

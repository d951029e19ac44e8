import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have(path):
    return os.path.exists(os.path.join(ROOT, path))


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build what can be built here: the host C library and the oracle need only gcc; the HIP library
    cross-compiles without a GPU.  On the GPU box the snapshot already carries the .so files."""
    if not _have("public_kssd_amd/libkssd_host.so") or not _have("public_kssd_amd/libkssd_gpu.so"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "public_kssd_amd")])
    if not _have("oracle/libkssd_oracle.so"):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"),
                               os.path.join(ROOT, "oracle", "libkssd_oracle.so")])
    # One HIP runtime per process: torch brings its own and asks for it as "libamdhip64.so"; imported BEFORE libkssd_gpu.so is
    # loaded, the loader hands that library the same copy (public_kssd_amd.capi.assert_single_runtime checks it on every load)
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    yield


@pytest.fixture(scope="session")
def shuf_l3k10():
    """The L3K10 shuffle every BASELINE config uses (k=10, subk=6, level 3), seeded."""
    import public_kssd_amd as K
    return K.Shuf.generate(10, 6, 3, seed=20260101)


@pytest.fixture(scope="session")
def gpu_ctx(shuf_l3k10):
    import public_kssd_amd as K
    ctx = K.GpuCtx(shuf_l3k10, 0)
    yield ctx
    ctx.close()

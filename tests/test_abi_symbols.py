"""The C-ABI library loads without a GPU and exports every symbol include/kssd_gpu.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared(header):
    txt = open(os.path.join(ROOT, header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(kssd_[a-z0-9_]+)\s*\(", txt)))


def test_gpu_abi_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(ROOT, "public_kssd_amd", "libkssd_gpu.so"))
    names = declared("include/kssd_gpu.h")
    assert len(names) >= 14
    for n in names:
        assert hasattr(lib, n), n


def test_host_lib_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(ROOT, "public_kssd_amd", "libkssd_host.so"))
    for n in declared("public_kssd_amd/host/kssd_host.h"):
        assert hasattr(lib, n), n


def test_binding_lists_the_same_symbols():
    import public_kssd_amd.capi as capi
    assert sorted(capi.GPU_SYMBOLS) == declared("include/kssd_gpu.h")


def test_no_gpu_means_loud_failure(shuf_l3k10):
    """Without a usable device the product raises; it never computes on the CPU."""
    import torch
    import public_kssd_amd as K
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(K.KssdError) as e:
        K.GpuCtx(shuf_l3k10, 0)
    assert e.value.code == -7

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "dr-using-scv-od_amd", "pyshim"))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU restatement (checker).  Built on demand from oracle/*.cpp with the oracle's Makefile."""
    import oracle_py
    return oracle_py.load()


@pytest.fixture(scope="session")
def spec():
    """CPU build of the product's arithmetic spec header (scvod_math.h) for function-level checks."""
    out = os.path.join(ROOT, "tests", "helpers", "libspec.so")
    src = os.path.join(ROOT, "tests", "helpers", "spec_shim.cpp")
    hdr = os.path.join(ROOT, "dr-using-scv-od_amd", "csrc", "scvod_math.h")
    if (not os.path.exists(out)) or os.path.getmtime(out) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", out, src])
    import ctypes
    lib = ctypes.CDLL(out)
    lib.spec_cmp_atan2f.restype = ctypes.c_long
    lib.spec_cmp_atan2.restype = ctypes.c_double
    lib.spec_atan2f.restype = ctypes.c_float
    lib.spec_atan2f.argtypes = [ctypes.c_float, ctypes.c_float]
    lib.spec_atan2.restype = ctypes.c_double
    lib.spec_atan2.argtypes = [ctypes.c_double, ctypes.c_double]
    return lib


@pytest.fixture(scope="session")
def scvod():
    import scvod_py
    scvod_py.load_lib()
    return scvod_py

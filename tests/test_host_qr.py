"""CPU: the spectral-radius routine of the gain-synthesis kernel (tzddpc_amd/csrc/tz_gain.hip.h) compiled for the host with
AddressSanitizer / UBSan and checked against LAPACK -- the GPU pool offers no sanitizers, the algorithm itself needs no GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def qrlib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("qr") / "libqr_host.so")
    src = os.path.join(HERE, "native", "qr_host.cpp")
    subprocess.check_call(["g++", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=undefined", "-fno-sanitize-recover=undefined",
                           "-shared", "-fPIC", "-o", out, src])
    lib = ctypes.CDLL(out)
    lib.specrad_host.restype = ctypes.c_int
    return lib


@pytest.mark.parametrize("n", [1, 2, 3, 5, 8])
def test_spectral_radius_routine_on_the_host(qrlib, n):
    rng = np.random.default_rng(50 + n)
    S = 3000
    M = rng.standard_normal((S, n, n))
    M[:100] = np.triu(M[:100]); M[100:150] = 0.0; M[150:200] = np.eye(n); M[200:300] *= 1e-8; M[300:400] *= 1e6
    if n >= 2:
        M[400:460] = 0.0; M[400:460, np.arange(n - 1) + 1, np.arange(n - 1)] = 1.0; M[400:460, 0, n - 1] = 1.0   # cyclic permutation
    rho = np.zeros(S)
    bad = qrlib.specrad_host(S, n, np.ascontiguousarray(M).ctypes.data_as(ctypes.c_void_p), rho.ctypes.data_as(ctypes.c_void_p))
    assert bad == 0
    ref = np.abs(np.linalg.eigvals(M)).max(axis=1)
    assert (np.abs(rho - ref) / (1e-300 + np.maximum(ref, 1e-12))).max() <= 1e-9 or (np.abs(rho - ref) / (1.0 + ref)).max() <= 1e-10

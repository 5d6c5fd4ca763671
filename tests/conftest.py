"""pytest configuration: markers and shared loaders.

`-m "not gpu"` (runs in the build container, no GPU): oracle vs golden vectors, host logic,
C-ABI symbol checks.  `-m gpu` (runs on an MI355X): parity of the HIP path vs the oracle.
"""
import ctypes
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run via gpurun)")


def _build_oracle():
    so = os.path.join(ROOT, "oracle", "libsbv_oracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return so


@pytest.fixture(scope="session")
def oracle():
    """ctypes handle on the C oracle (test infrastructure)."""
    lib = ctypes.CDLL(_build_oracle())
    lib.sbvo_p256_verify_tuple.argtypes = [ctypes.c_char_p]
    lib.sbvo_p256_verify_asn1.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t,
                                          ctypes.c_char_p, ctypes.c_size_t]
    lib.sbvo_p256_parse_der.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
    lib.sbvo_p256_verify_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int]
    lib.sbvo_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint,
                                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    lib.sbvo_sha256.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
    return lib


@pytest.fixture(scope="session")
def openssl_check():
    so = os.path.join(ROOT, "oracle", "libsbv_openssl.so")
    _build_oracle()
    if not os.path.exists(so):
        pytest.skip("libcrypto not available for the OpenSSL third opinion")
    lib = ctypes.CDLL(so)
    lib.sbvssl_p256_verify_tuple.argtypes = [ctypes.c_char_p]
    return lib


@pytest.fixture(scope="session")
def golden_vectors():
    with open(os.path.join(GOLDEN, "p256_vectors.json")) as f:
        return json.load(f)["vectors"]


@pytest.fixture(scope="session")
def rfc6979():
    with open(os.path.join(GOLDEN, "rfc6979_p256.json")) as f:
        return json.load(f)

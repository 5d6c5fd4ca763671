"""Synthetic P-256 signature batches from tools/libsbv_datagen.so (OpenSSL),
with an on-disk cache under /tmp so that tests and bench runs in one gpurun call share the
~30 s generation of the 2^20-tuple batch.  Independent of oracle/ and of the product."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def _load():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "libsbv_datagen.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", HERE, "libsbv_datagen.so"])
        _lib = ctypes.CDLL(so)
        _lib.sbvd_gen_batch.argtypes = [ctypes.c_uint32, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint,
                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
    return _lib


def gen_batch(seed: int, n: int, nkeys: int = 1024, invalid_every: int = 8, threads: int = 0, cache: bool = True):
    """-> (tuples uint8[n*160], valid uint8[ceil(n/8)]) as numpy arrays."""
    path = f"/tmp/sbv_batch_{seed:08x}_{n}_{nkeys}_{invalid_every}.npz"
    if cache and os.path.exists(path):
        try:
            z = np.load(path)
            return z["tuples"], z["valid"]
        except Exception:
            pass
    tuples = np.zeros(n * 160, dtype=np.uint8)
    valid = np.zeros((n + 7) // 8, dtype=np.uint8)
    rc = _load().sbvd_gen_batch(seed, n, nkeys, invalid_every, tuples.ctypes.data, valid.ctypes.data,
                                threads or (os.cpu_count() or 1))
    if rc != 0:
        raise RuntimeError("sbvd_gen_batch failed")
    if cache and n >= 65536:
        try:
            np.savez(path + ".tmp.npz", tuples=tuples, valid=valid)
            os.replace(path + ".tmp.npz", path)
        except Exception:
            pass
    return tuples, valid

"""consensus_amd — MI355X-native batch signature verification behind SmartBFT's api.Verifier.

This package is a thin ctypes binding over the product's C-ABI (include/sbv.h,
consensus_amd/libsbv.so).  It holds no verification logic and **no CPU fallback**: if the HIP
library is missing or no gfx950 device is usable, every compute call raises.

Reference seam: /root/reference/pkg/api/dependencies.go:54-71 (api.Verifier).
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SBV_LIB") or os.path.join(_HERE, "libsbv.so")      # SBV_LIB: A/B builds of the same ABI (tools/)
TUPLE_BYTES = 160

SBV_OK = 0
ERRORS = {-1: "SBV_ENODEV", -2: "SBV_EINVAL", -3: "SBV_ENOMEM", -4: "SBV_EDEVICE", -5: "SBV_ENOTINIT",
          -6: "SBV_EPARSE"}


class SbvError(RuntimeError):
    def __init__(self, code: int, detail: str = ""):
        self.code = code
        super().__init__(f"libsbv: {ERRORS.get(code, code)} {detail}".strip())


class Timing(ctypes.Structure):
    _fields_ = [("h2d_us", ctypes.c_double), ("prep_us", ctypes.c_double), ("verify_us", ctypes.c_double),
                ("d2h_us", ctypes.c_double), ("total_us", ctypes.c_double), ("n", ctypes.c_uint64)]


_lib: Optional[ctypes.CDLL] = None


def _preload_hip_runtime() -> None:
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (soname
    libamdhip64.so.7, same as /opt/rocm's).  If libsbv.so pulled in the system copy first and
    torch were imported later, the process would hold two HIP runtimes and torch would report
    "No HIP GPUs are available".  Loading torch's copy first (when torch is installed) makes
    libsbv's NEEDED libamdhip64.so.7 resolve to it by soname, whatever the import order.
    Without torch the system ROCm runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return                      # torch already loaded its runtime; the soname match does the rest
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def load() -> ctypes.CDLL:
    """Load libsbv.so (built by __graft_entry__.build() / consensus_amd/csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(the product has no CPU fallback)")
    _preload_hip_runtime()
    lib = ctypes.CDLL(LIB_PATH)
    lib.sbv_init.argtypes = [ctypes.c_int]
    lib.sbv_p256_verify_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.sbv_p256_verify_batch_dev.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    lib.sbv_p256_parse_der.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
    lib.sbv_sha256_batch.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_size_t, ctypes.c_char_p]
    lib.sbv_p256_register_keys.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32)]
    lib.sbv_p256_verify_batch_keyed.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.sbv_p256_verify_batch_keyed_dev.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p,
                                                    ctypes.c_void_p]
    lib.sbv_ed25519_verify_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.sbv_ed25519_verify_batch_dev.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    lib.sbv_secp256k1_verify_batch.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.sbv_secp256k1_verify_batch_dev.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
    lib.sbv_ed25519_verify_msgs.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64),
                                            ctypes.c_size_t, ctypes.c_char_p]
    lib.sbv_ed25519_make_tuples.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64),
                                            ctypes.c_size_t, ctypes.c_char_p]
    lib.sbv_p256_verify_msgs_keyed.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p,
                                               ctypes.POINTER(ctypes.c_uint64), ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.sbv_p256_set_grouping.argtypes = [ctypes.c_int, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32]
    lib.sbv_last_timing.argtypes = [ctypes.POINTER(Timing)]
    lib.sbv_profile_enable.argtypes = [ctypes.c_int]
    lib.sbv_profile_read.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                     ctypes.POINTER(ctypes.c_uint64)]
    lib.sbv_profile_read_dominant.argtypes = [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_uint64)]
    lib.sbv_p256_last_group_stats.argtypes = [ctypes.POINTER(ctypes.c_uint32)]
    lib.sbv_last_error.restype = ctypes.c_char_p
    _lib = lib
    return lib


def _check(rc: int) -> None:
    if rc != SBV_OK:
        raise SbvError(rc, load().sbv_last_error().decode(errors="replace"))


def init(device: int = 0) -> None:
    _check(load().sbv_init(device))


def shutdown() -> None:
    _check(load().sbv_shutdown())


def device_count() -> int:
    return load().sbv_device_count()


def verify_batch(tuples: bytes, n: Optional[int] = None) -> bytes:
    """Verify n 160-byte tuples held in host memory; returns the ceil(n/8)-byte accept bitmap."""
    if n is None:
        if len(tuples) % TUPLE_BYTES:
            raise ValueError("tuple buffer is not a multiple of 160 bytes")
        n = len(tuples) // TUPLE_BYTES
    if len(tuples) < n * TUPLE_BYTES:
        raise ValueError("tuple buffer too short")
    out = ctypes.create_string_buffer(max(1, (n + 7) // 8))
    buf = (ctypes.c_char * len(tuples)).from_buffer_copy(tuples) if n else None
    _check(load().sbv_p256_verify_batch(buf, n, out))
    return out.raw[:(n + 7) // 8]


def verify_batch_ptr(host_ptr: int, n: int, out_ptr: int) -> None:
    """Raw-pointer form (e.g. numpy buffers) of sbv_p256_verify_batch."""
    _check(load().sbv_p256_verify_batch(host_ptr, n, out_ptr))


def verify_batch_dev(d_tuples_ptr: int, n: int, d_bitmap_ptr: int, stream: int = 0) -> None:
    """Asynchronous verification of device-resident tuples on `stream` (a hipStream_t value)."""
    _check(load().sbv_p256_verify_batch_dev(d_tuples_ptr, n, d_bitmap_ptr, stream))


def register_keys(keys) -> list:
    """keys: iterable of 64-byte Qx|Qy -> list of slots (equal keys share a slot)."""
    keys = list(keys)
    blob = b"".join(keys)
    if len(blob) != 64 * len(keys):
        raise ValueError("every key must be 64 bytes")
    out = (ctypes.c_uint32 * max(1, len(keys)))()
    _check(load().sbv_p256_register_keys(blob, len(keys), out))
    return list(out[:len(keys)])


def key_count() -> int:
    return load().sbv_p256_key_count()


def clear_keys() -> None:
    _check(load().sbv_p256_clear_keys())


WIDE_BITS_AUTO = 1


def wide_keys(bits: int = WIDE_BITS_AUTO, max_keys: int = 64) -> None:
    """sbv_p256_wide_keys: width and cap of the wide combs sbv_p256_widen_keys builds (bits = 0: off); see include/sbv.h."""
    lib = load()
    lib.sbv_p256_wide_keys.argtypes = [ctypes.c_int, ctypes.c_uint32]
    _check(lib.sbv_p256_wide_keys(bits, max_keys))


def widen_keys(slots) -> None:
    """sbv_p256_widen_keys: a wide comb for each of these registered slots (the consenters')."""
    slots = list(slots)
    arr = (ctypes.c_uint32 * max(1, len(slots)))(*slots)
    lib = load()
    lib.sbv_p256_widen_keys.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
    _check(lib.sbv_p256_widen_keys(arr, len(slots)))


def wide_selfcheck(slot: int) -> bool:
    """sbv_p256_wide_selfcheck: the device-built wide comb of `slot` equals the host builder's output byte for byte."""
    lib = load()
    lib.sbv_p256_wide_selfcheck.argtypes = [ctypes.c_uint32]
    rc = lib.sbv_p256_wide_selfcheck(slot)
    if rc < 0:
        _check(rc)
    return rc == 1


def wide_key_stats():
    """(slots holding a wide comb, bits, max_keys, KiB per key)"""
    out = (ctypes.c_uint32 * 4)()
    lib = load()
    lib.sbv_p256_wide_key_stats.argtypes = [ctypes.c_void_p]
    _check(lib.sbv_p256_wide_key_stats(out))
    return out[0], out[1], out[2], out[3]


def verify_batch_keyed(rsh: bytes, slots, n: Optional[int] = None) -> bytes:
    """Registered-key form: rsh = n x 96 bytes (r|s|hash), slots = n key slots."""
    if n is None:
        n = len(rsh) // 96
    arr = (ctypes.c_uint32 * max(1, n))(*slots)
    out = ctypes.create_string_buffer(max(1, (n + 7) // 8))
    buf = (ctypes.c_char * len(rsh)).from_buffer_copy(rsh) if n else None
    _check(load().sbv_p256_verify_batch_keyed(buf, arr, n, out))
    return out.raw[:(n + 7) // 8]


def verify_batch_keyed_dev(d_rsh_ptr: int, d_slots_ptr: int, n: int, d_bitmap_ptr: int, stream: int = 0) -> None:
    _check(load().sbv_p256_verify_batch_keyed_dev(d_rsh_ptr, d_slots_ptr, n, d_bitmap_ptr, stream))


def ed25519_make_tuples(sigs, pks, msgs) -> bytes:
    """(64-byte sig, 32-byte pk, message) triples -> n x 128-byte tuples (R | S | pk | k)."""
    n = len(sigs)
    offs = (ctypes.c_uint64 * (n + 1))()
    acc = 0
    for i, m in enumerate(msgs):
        offs[i] = acc
        acc += len(m)
    offs[n] = acc
    out = ctypes.create_string_buffer(max(1, 128 * n))
    _check(load().sbv_ed25519_make_tuples(b"".join(sigs), b"".join(pks), b"".join(msgs), offs, n, out))
    return out.raw[:128 * n]


def ed25519_verify_msgs(sigs, pks, msgs) -> bytes:
    """(64-byte sig, 32-byte pk, message) triples -> accept bitmap; SHA-512 and the reduction mod L run on the device."""
    n = len(sigs)
    offs = (ctypes.c_uint64 * (n + 1))()
    acc = 0
    for i, m in enumerate(msgs):
        offs[i] = acc
        acc += len(m)
    offs[n] = acc
    out = ctypes.create_string_buffer(max(1, (n + 7) // 8))
    _check(load().sbv_ed25519_verify_msgs(b"".join(sigs), b"".join(pks), b"".join(msgs), offs, n, out))
    return out.raw[:(n + 7) // 8]


def ed25519_verify_batch(tuples: bytes, n: Optional[int] = None) -> bytes:
    if n is None:
        n = len(tuples) // 128
    out = ctypes.create_string_buffer(max(1, (n + 7) // 8))
    buf = (ctypes.c_char * len(tuples)).from_buffer_copy(tuples) if n else None
    _check(load().sbv_ed25519_verify_batch(buf, n, out))
    return out.raw[:(n + 7) // 8]


def ed25519_verify_batch_dev(d_tuples_ptr: int, n: int, d_bitmap_ptr: int, stream: int = 0) -> None:
    _check(load().sbv_ed25519_verify_batch_dev(d_tuples_ptr, n, d_bitmap_ptr, stream))


def secp256k1_verify_batch(tuples: bytes, n: Optional[int] = None) -> bytes:
    """ECDSA over secp256k1 on 160-byte tuples r | s | hash | Qx | Qy (include/sbv.h); returns the accept bitmap."""
    if n is None:
        n = len(tuples) // 160
    out = ctypes.create_string_buffer(max(1, (n + 7) // 8))
    buf = (ctypes.c_char * len(tuples)).from_buffer_copy(tuples) if n else None
    _check(load().sbv_secp256k1_verify_batch(buf, n, out))
    return out.raw[:(n + 7) // 8]


def secp256k1_verify_batch_dev(d_tuples_ptr: int, n: int, d_bitmap_ptr: int, stream: int = 0) -> None:
    _check(load().sbv_secp256k1_verify_batch_dev(d_tuples_ptr, n, d_bitmap_ptr, stream))


def verify_msgs_keyed(msgs, sigs_der, slots) -> bytes:
    """Device front end: SHA-256(msg) + strict DER parse on the GPU, then registered-key verification."""
    n = len(msgs)
    mo = (ctypes.c_uint64 * (n + 1))()
    so = (ctypes.c_uint64 * (n + 1))()
    a = b = 0
    for i in range(n):
        mo[i], so[i] = a, b
        a += len(msgs[i]); b += len(sigs_der[i])
    mo[n], so[n] = a, b
    arr = (ctypes.c_uint32 * max(1, n))(*slots)
    out = ctypes.create_string_buffer(max(1, (n + 7) // 8))
    _check(load().sbv_p256_verify_msgs_keyed(b"".join(msgs), mo, b"".join(sigs_der), so, arr, n, out))
    return out.raw[:(n + 7) // 8]


def set_grouping(enabled: bool, min_batch: int = 0, min_count: int = 0, max_groups: int = 0) -> None:
    """In-step grouping of generic batches by public key (0 keeps a value; min_batch = GROUP_MIN_BATCH_DEFAULT restores the
    built-in thresholds); see include/sbv.h."""
    lib = load()
    _check(lib.sbv_p256_set_grouping(1 if enabled else 0, min_batch, min_count, max_groups))


SCHEME_P256, SCHEME_SECP256K1, SCHEME_ED25519 = 0, 1, 2
GROUP_MIN_BATCH_DEFAULT = (1 << 64) - 1     # (size_t)-1


def key_cache(enabled: bool, capacity: int = 0, scheme: int = SCHEME_P256) -> None:
    """sbv_key_cache: the persistent key-table cache of a scheme's grouped step (off also empties it); the default is the
    P-256 one (sbv_p256_key_cache)."""
    lib = load()
    lib.sbv_key_cache.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint32]
    _check(lib.sbv_key_cache(scheme, 1 if enabled else 0, capacity))


def sign_batch(keys: bytes, digests: bytes, key_index=None):
    """sbv_p256_sign_batch: RFC 6979 ECDSA P-256 signatures (r | s, 64 bytes each) of n 32-byte digests under the 32-byte private
    scalars in `keys` (key_index[i], default i % n_keys).  Returns (sigs, ok) with ok[i] = 1 per produced signature."""
    lib = load()
    n, nk = len(digests) // 32, len(keys) // 32
    lib.sbv_p256_sign_batch.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t,
                                        ctypes.c_char_p, ctypes.c_char_p]
    sigs, ok = ctypes.create_string_buffer(64 * n), ctypes.create_string_buffer(max(1, n))
    idx = None if key_index is None else (ctypes.c_uint32 * n)(*key_index)
    _check(lib.sbv_p256_sign_batch(keys, nk, idx, digests, n, sigs, ok))
    return sigs.raw, ok.raw[:n]


def sign_batch_dev(d_keys_ptr: int, n_keys: int, d_index_ptr: int, d_digests_ptr: int, n: int, d_sigs_ptr: int, d_ok_ptr: int,
                   stream: int = 0) -> None:
    lib = load()
    lib.sbv_p256_sign_batch_dev.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    _check(lib.sbv_p256_sign_batch_dev(d_keys_ptr, n_keys, d_index_ptr or None, d_digests_ptr, n, d_sigs_ptr, d_ok_ptr, stream or None))


def key_cache_stats(scheme: int = SCHEME_P256):
    """(cached keys, groups of the last grouped batch that hit, that missed, capacity)"""
    out = (ctypes.c_uint32 * 4)()
    lib = load()
    lib.sbv_key_cache_stats.argtypes = [ctypes.c_int, ctypes.c_void_p]
    _check(lib.sbv_key_cache_stats(scheme, out))
    return out[0], out[1], out[2], out[3]


def parse_der(sig: bytes) -> Optional[bytes]:
    """Strict DER -> r|s (64 bytes), or None when Go's parseSignature would fail."""
    out = ctypes.create_string_buffer(64)
    rc = load().sbv_p256_parse_der(sig, len(sig), out)
    if rc == SBV_OK:
        return out.raw
    if rc == -6:
        return None
    _check(rc)
    return None


def sha256_batch(msgs) -> bytes:
    blob = b"".join(msgs)
    offs = (ctypes.c_uint64 * (len(msgs) + 1))()
    acc = 0
    for i, m in enumerate(msgs):
        offs[i] = acc
        acc += len(m)
    offs[len(msgs)] = acc
    out = ctypes.create_string_buffer(32 * max(1, len(msgs)))
    _check(load().sbv_sha256_batch(blob, offs, len(msgs), out))
    return out.raw[:32 * len(msgs)]


def host_alloc(nbytes: int) -> int:
    """Page-locked host memory for the host-pointer entries (sbv_host_alloc); returns the address, 0 on failure."""
    lib = load()
    lib.sbv_host_alloc.restype = ctypes.c_void_p
    lib.sbv_host_alloc.argtypes = [ctypes.c_size_t]
    return lib.sbv_host_alloc(nbytes) or 0


def host_free(ptr: int) -> None:
    lib = load()
    lib.sbv_host_free.argtypes = [ctypes.c_void_p]
    lib.sbv_host_free(ptr)


class ShardInfo(ctypes.Structure):
    _fields_ = [("devices", ctypes.c_int), ("shards", ctypes.c_int), ("mode", ctypes.c_int), ("tuples_per_shard", ctypes.c_size_t),
                ("h2d_us", ctypes.c_double), ("kernels_us", ctypes.c_double), ("gather_us", ctypes.c_double), ("total_us", ctypes.c_double)]


def init_all() -> int:
    """Initialise every visible gfx950 device (sbv_init_all); returns how many."""
    n = load().sbv_init_all()
    if n < 0:
        _check(n)
    return n


def shard_plan(n: int, devices: int, group: int = 0, min_per_device: int = 0):
    """The pure split of sbv_p256_verify_batch_sharded: list of shard start indices + [n]."""
    lib = load()
    lib.sbv_shard_plan.restype = ctypes.c_size_t
    lib.sbv_shard_plan.argtypes = [ctypes.c_size_t, ctypes.c_int, ctypes.c_size_t, ctypes.c_size_t, ctypes.POINTER(ctypes.c_size_t)]
    first = (ctypes.c_size_t * 17)()
    k = lib.sbv_shard_plan(n, devices, group, min_per_device, first)
    return [first[i] for i in range(k + 1)]


def shard_min_for(tuples, n: int, group: int = 0) -> int:
    """sbv_shard_min_for: the per-device minimum the sharded entry plans this host batch with (few signers -> 2^16, else 2^17)."""
    lib = load()
    lib.sbv_shard_min_for.restype = ctypes.c_size_t
    lib.sbv_shard_min_for.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t]
    if isinstance(tuples, int):
        return lib.sbv_shard_min_for(tuples, n, group)
    buf = (ctypes.c_char * len(tuples)).from_buffer_copy(tuples) if not isinstance(tuples, ctypes.Array) else tuples
    return lib.sbv_shard_min_for(ctypes.addressof(buf), n, group)


def verify_batch_sharded(host_ptr: int, n: int, out_ptr: int, group: int = 0, quorum: int = 0, quorum_out_ptr: int = 0) -> ShardInfo:
    """sbv_p256_verify_batch_sharded on raw pointers (numpy / ctypes buffers)."""
    lib = load()
    lib.sbv_p256_verify_batch_sharded.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_void_p,
                                                  ctypes.c_void_p, ctypes.POINTER(ShardInfo)]
    info = ShardInfo()
    _check(lib.sbv_p256_verify_batch_sharded(host_ptr, n, group, quorum, out_ptr, quorum_out_ptr or None, ctypes.byref(info)))
    return info


def verify_batch_keyed_sharded(rsh_ptr: int, slots_ptr: int, n: int, out_ptr: int, group: int = 0, quorum: int = 0,
                               quorum_out_ptr: int = 0) -> ShardInfo:
    """sbv_p256_verify_batch_keyed_sharded on raw pointers: n x 96-byte records r | s | hash + n u32 key slots, split over every
    initialised device (each holds a replica of the key registry and of the consenters' wide combs); quorum bits by distinct slot."""
    lib = load()
    lib.sbv_p256_verify_batch_keyed_sharded.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint32,
                                                        ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ShardInfo)]
    info = ShardInfo()
    _check(lib.sbv_p256_verify_batch_keyed_sharded(rsh_ptr, slots_ptr, n, group, quorum, out_ptr, quorum_out_ptr or None, ctypes.byref(info)))
    return info


def verify_msgs_keyed_sharded(msgs, sigs_der, slots, group: int = 0, quorum: int = 0, offsets=None):
    """sbv_p256_verify_msgs_keyed_sharded: raw messages + DER signatures + key slots over every initialised device, SHA-256 and the
    DER parse on the devices, uploads in pieces.  Returns (accept bitmap, quorum bitmap or None, ShardInfo).
    offsets = (msg_offsets, sig_offsets): use these tables instead of the ones the byte strings imply (tests of malformed tables)."""
    lib = load()
    lib.sbv_p256_verify_msgs_keyed_sharded.argtypes = [ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint64),
                                                       ctypes.c_void_p, ctypes.c_size_t, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p,
                                                       ctypes.POINTER(ShardInfo)]
    n = len(msgs)
    mo = (ctypes.c_uint64 * (n + 1))()
    so = (ctypes.c_uint64 * (n + 1))()
    a = b = 0
    for i in range(n):
        mo[i], so[i] = a, b
        a += len(msgs[i]); b += len(sigs_der[i])
    mo[n], so[n] = a, b
    if offsets is not None:
        for i in range(n + 1):
            mo[i], so[i] = offsets[0][i], offsets[1][i]
    arr = (ctypes.c_uint32 * max(1, n))(*slots)
    out = ctypes.create_string_buffer(max(1, (n + 7) // 8))
    props = n // group if group else 0
    qout = ctypes.create_string_buffer(max(1, (props + 7) // 8)) if group and quorum else None
    info = ShardInfo()
    _check(lib.sbv_p256_verify_msgs_keyed_sharded(b"".join(msgs), mo, b"".join(sigs_der), so, arr, n, group, quorum, out, qout, ctypes.byref(info)))
    return out.raw[:(n + 7) // 8], (qout.raw[:(props + 7) // 8] if qout is not None else None), info


def shard_mode(by_key: bool, parts: int = 0) -> None:
    """sbv_shard_mode: how the sharded entry partitions a batch that spans devices (contiguous ranges / by key hash); parts = 0
    means one part per device, more parts than devices run one after another on their device."""
    lib = load()
    lib.sbv_shard_mode.argtypes = [ctypes.c_int, ctypes.c_uint]
    _check(lib.sbv_shard_mode(1 if by_key else 0, parts))


def verify_batch_dev_part(d_tuples_ptr: int, n: int, part: int, parts: int, d_bitmap_words_ptr: int, stream: int = 0) -> int:
    """sbv_p256_verify_batch_dev_part: part `part` of `parts` (by key hash) of n device-resident tuples; the bitmap (ceil(n/32)
    words) receives that part's verdict bits.  Returns how many tuples the part held."""
    lib = load()
    lib.sbv_p256_verify_batch_dev_part.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p,
                                                   ctypes.c_void_p, ctypes.POINTER(ctypes.c_size_t)]
    members = ctypes.c_size_t(0)
    _check(lib.sbv_p256_verify_batch_dev_part(d_tuples_ptr, n, part, parts, d_bitmap_words_ptr, stream, ctypes.byref(members)))
    return members.value


def verify_batch_on(device: int, host_ptr: int, n: int, out_ptr: int) -> None:
    lib = load()
    lib.sbv_p256_verify_batch_on.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    _check(lib.sbv_p256_verify_batch_on(device, host_ptr, n, out_ptr))


def last_timing() -> Timing:
    t = Timing()
    _check(load().sbv_last_timing(ctypes.byref(t)))
    return t


def profile_enable(on) -> None:
    """True / 1: step triples + dominant-kernel pairs; 2: dominant-kernel pairs only; False / 0: off."""
    _check(load().sbv_profile_enable(2 if on == 2 else (1 if on else 0)))


def profile_read():
    """(prep_us_sum, verify_us_sum, launches) of the device-pointer calls since the last read."""
    p, v, k = ctypes.c_double(), ctypes.c_double(), ctypes.c_uint64()
    _check(load().sbv_profile_read(ctypes.byref(p), ctypes.byref(v), ctypes.byref(k)))
    return p.value, v.value, k.value


def profile_read_dominant():
    """(summed duration in us, number of launches) of the dominant stage-B kernel since profiling was enabled;
    call before profile_read()."""
    d = ctypes.c_double()
    k = ctypes.c_uint64()
    _check(load().sbv_profile_read_dominant(ctypes.byref(d), ctypes.byref(k)))
    return d.value, k.value


def last_group_stats():
    """(key groups, tuples through the per-batch key tables, tuples through the generic kernel, ungrouped tuples
    rejected for their public key alone) of the last grouped batch."""
    out = (ctypes.c_uint32 * 4)()
    _check(load().sbv_p256_last_group_stats(out))
    return out[0], out[1], out[2], out[3]


def last_table_classes():
    """(groups verified from a full table, groups filled in this batch, grouped tuples served by the rows-only pass) of the last grouped
    P-256 batch (sbv_p256_last_table_classes)."""
    out = (ctypes.c_uint32 * 3)()
    lib = load()
    lib.sbv_p256_last_table_classes.argtypes = [ctypes.c_void_p]
    _check(lib.sbv_p256_last_table_classes(out))
    return out[0], out[1], out[2]


def pool_stats():
    """sbv_p256_pool_stats: dict of the grouped step's pools on the default device (cached keys, groups per batch, shrunk, fallbacks for
    lack of memory, hot-key combs, contexts sharing the GPU)."""
    out = (ctypes.c_uint32 * 6)()
    lib = load()
    lib.sbv_p256_pool_stats.argtypes = [ctypes.c_void_p]
    _check(lib.sbv_p256_pool_stats(out))
    return {"cache_keys": out[0], "groups_per_batch": out[1], "shrunk": bool(out[2]), "nomem_fallbacks": out[3], "hot_pool": out[4], "gpu_share": out[5]}


def debug_hot_check(device: int = 0):
    """sbv_debug_hot_check: tuple of the 8 diagnostic words for context `device` (slow: a host build per promoted comb)."""
    out = (ctypes.c_uint32 * 8)()
    lib = load()
    lib.sbv_debug_hot_check.argtypes = [ctypes.c_int, ctypes.c_void_p]
    _check(lib.sbv_debug_hot_check(device, out))
    return tuple(out)


def hot_keys(max_keys: int = 1024, min_hits: int = 0) -> None:
    """sbv_p256_hot_keys: wide combs for hot cache slots of the generic path (0 keys = off)."""
    lib = load()
    lib.sbv_p256_hot_keys.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    _check(lib.sbv_p256_hot_keys(max_keys, min_hits))


def hot_key_stats():
    """(promoted keys, pool capacity, tuples of the last grouped batch served by the wide pass, min_hits)"""
    out = (ctypes.c_uint32 * 4)()
    lib = load()
    lib.sbv_p256_hot_key_stats.argtypes = [ctypes.c_void_p]
    _check(lib.sbv_p256_hot_key_stats(out))
    return out[0], out[1], out[2], out[3]


def hot_selfcheck(index: int) -> bool:
    lib = load()
    lib.sbv_p256_hot_selfcheck.argtypes = [ctypes.c_uint32]
    rc = lib.sbv_p256_hot_selfcheck(index)
    if rc < 0:
        _check(rc)
    return rc == 1


def ed_hot_keys(max_keys: int = 1024, min_hits: int = 0) -> None:
    """sbv_ed25519_hot_keys: 16-bit combs of -A for hot cache slots of the Ed25519 variant (0 keys = off)."""
    lib = load()
    lib.sbv_ed25519_hot_keys.argtypes = [ctypes.c_uint32, ctypes.c_uint32]
    _check(lib.sbv_ed25519_hot_keys(max_keys, min_hits))


def ed_hot_key_stats():
    """(promoted keys, pool capacity, tuples of the last grouped Ed25519 batch served by the wide pass, min_hits)"""
    out = (ctypes.c_uint32 * 4)()
    lib = load()
    lib.sbv_ed25519_hot_key_stats.argtypes = [ctypes.c_void_p]
    _check(lib.sbv_ed25519_hot_key_stats(out))
    return out[0], out[1], out[2], out[3]


def ed_hot_selfcheck(index: int) -> bool:
    lib = load()
    lib.sbv_ed25519_hot_selfcheck.argtypes = [ctypes.c_uint32]
    rc = lib.sbv_ed25519_hot_selfcheck(index)
    if rc < 0:
        _check(rc)
    return rc == 1


def bitmap_to_list(bm: bytes, n: int):
    return [bool((bm[i >> 3] >> (i & 7)) & 1) for i in range(n)]

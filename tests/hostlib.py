"""ctypes binding of consensus_amd/libsbv_host.so (the C++ api.Verifier / api.Signer mirror) for tests."""
import ctypes
import hashlib
import os
import struct
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BACKEND_FN = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p)
OK, INVALID, UNAVAILABLE = 0, 1, 2
V = ctypes.c_void_p


class ReplayResult(ctypes.Structure):
    _fields_ = [("setup_s", ctypes.c_double), ("verify_proposal_us", ctypes.c_double), ("prev_commits_us", ctypes.c_double),
                ("commit_quorum_us", ctypes.c_double), ("batch_total_us", ctypes.c_double), ("batch_tuples", ctypes.c_uint64),
                ("proposals_with_quorum", ctypes.c_uint64), ("backend_batches", ctypes.c_uint64),
                ("max_backend_batch", ctypes.c_uint64), ("status", ctypes.c_int), ("batch_first_us", ctypes.c_double)]


def load():
    so = os.path.join(ROOT, "consensus_amd", "libsbv_host.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "consensus_amd", "host")])
    import consensus_amd
    consensus_amd._preload_hip_runtime()
    lib = ctypes.CDLL(so)
    V, S = ctypes.c_void_p, ctypes.c_size_t
    lib.sbvh_verifier_new.restype = V
    lib.sbvh_verifier_new.argtypes = [ctypes.c_int, ctypes.c_int, BACKEND_FN, V, S, ctypes.c_int, ctypes.c_int]
    lib.sbvh_verifier_new_scheme.restype = V
    lib.sbvh_verifier_new_scheme.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, BACKEND_FN, V, S, ctypes.c_int, ctypes.c_int]
    lib.sbvh_signer_new_scheme.restype = V
    lib.sbvh_signer_new_scheme.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_char_p]
    lib.sbvh_ed25519_gen_batch.argtypes = [ctypes.c_uint32, S, S, ctypes.c_uint, V, V, ctypes.c_int]
    lib.sbvh_ed25519_gen_batch.restype = None
    lib.sbvh_k256_gen_batch.argtypes = [ctypes.c_uint32, S, S, ctypes.c_uint, V, V, ctypes.c_int]
    lib.sbvh_k256_gen_batch.restype = None
    lib.sbvh_verifier_free.argtypes = [V]
    lib.sbvh_register_consenter.argtypes = [V, ctypes.c_uint64, ctypes.c_char_p]
    lib.sbvh_register_client.argtypes = [V, ctypes.c_char_p, ctypes.c_char_p]
    lib.sbvh_set_verification_sequence.argtypes = [V, ctypes.c_uint64]
    lib.sbvh_set_device_client_keys.argtypes = [V, ctypes.c_int]
    lib.sbvh_verification_sequence.restype = ctypes.c_uint64
    lib.sbvh_verification_sequence.argtypes = [V]
    lib.sbvh_verify_signature.argtypes = [V, ctypes.c_uint64, ctypes.c_char_p, S, ctypes.c_char_p, S]
    lib.sbvh_verify_consenter_sig.argtypes = [V, ctypes.c_uint64, ctypes.c_char_p, S, ctypes.c_char_p, S, ctypes.c_char_p, S,
                                              ctypes.c_char_p, S, ctypes.c_char_p, S, ctypes.c_int64, ctypes.c_char_p, S,
                                              ctypes.POINTER(S)]
    lib.sbvh_auxiliary_data.restype = S
    lib.sbvh_auxiliary_data.argtypes = [V, ctypes.c_char_p, S, ctypes.c_char_p, S]
    lib.sbvh_verify_request.argtypes = [V, ctypes.c_char_p, S, ctypes.c_char_p, S, ctypes.POINTER(S)]
    lib.sbvh_request_id.restype = S
    lib.sbvh_request_id.argtypes = [V, ctypes.c_char_p, S, ctypes.c_char_p, S]
    lib.sbvh_verify_proposal.argtypes = [V, ctypes.c_char_p, S, ctypes.c_char_p, S, ctypes.c_char_p, S, ctypes.c_int64,
                                         ctypes.c_char_p, S, ctypes.POINTER(S), ctypes.POINTER(S)]
    lib.sbvh_payload_parsers_agree.argtypes = [ctypes.c_char_p, S, ctypes.POINTER(S)]
    lib.sbvh_requests_from_proposal.restype = S
    lib.sbvh_requests_from_proposal.argtypes = [V, ctypes.c_char_p, S, ctypes.c_char_p, S, ctypes.POINTER(S)]
    lib.sbvh_stats.argtypes = [V, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    lib.sbvh_signer_new.restype = V
    lib.sbvh_signer_new.argtypes = [ctypes.c_uint64, ctypes.c_char_p]
    lib.sbvh_signer_free.argtypes = [V]
    lib.sbvh_signer_public_key.argtypes = [V, ctypes.c_char_p]
    lib.sbvh_sign.restype = S
    lib.sbvh_sign.argtypes = [V, ctypes.c_char_p, S, ctypes.c_char_p, S]
    lib.sbvh_sign_proposal.argtypes = [V, ctypes.c_char_p, S, ctypes.c_char_p, S, ctypes.c_char_p, S, ctypes.c_int64,
                                       ctypes.c_char_p, S, ctypes.c_char_p, S, ctypes.POINTER(S), ctypes.c_char_p, S,
                                       ctypes.POINTER(S)]
    lib.sbvh_sign_with_nonce.argtypes = [ctypes.c_char_p] * 4
    lib.sbvh_sign_rfc6979.argtypes = [ctypes.c_char_p] * 3
    lib.sbvh_pubkey.argtypes = [ctypes.c_char_p] * 2
    lib.sbvh_proposal_digest.argtypes = [ctypes.c_char_p, S, ctypes.c_char_p, S, ctypes.c_char_p, S, ctypes.c_int64, ctypes.c_char_p]
    lib.sbvh_compute_quorum.argtypes = [ctypes.c_uint64, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
    lib.sbvh_chain_emulate.argtypes = [ctypes.POINTER(V), ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_char_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64),
                                       ctypes.POINTER(ctypes.c_uint64)]
    lib.sbvh_backend_widened_keys.argtypes = [V]
    lib.sbvh_backend_widened_keys.restype = ctypes.c_uint64
    lib.sbvh_batch_faults.argtypes = [V, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint64)]
    lib.sbvh_replay.argtypes = [V, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ReplayResult)]
    lib.sbvh_replay_samples.argtypes = [V, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ReplayResult),
                                        ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    return lib


# ---- independent Python statement of the wire formats (struct.pack), to cross-check formats.cc -------------
def request_unsigned(client: str, rid: str, payload: bytes) -> bytes:
    c, i = client.encode(), rid.encode()
    return struct.pack(">H", len(c)) + c + struct.pack(">H", len(i)) + i + struct.pack(">I", len(payload)) + payload


def request_encode(unsigned: bytes, sig: bytes) -> bytes:
    return unsigned + struct.pack(">H", len(sig)) + sig


def payload_encode(reqs) -> bytes:
    return struct.pack(">I", len(reqs)) + b"".join(struct.pack(">I", len(r)) + r for r in reqs)


def _der_len(n):
    if n < 128:
        return bytes([n])
    b = n.to_bytes((n.bit_length() + 7) // 8, "big")
    return bytes([0x80 | len(b)]) + b


def asn1_proposal(payload: bytes, header: bytes, metadata: bytes, vseq: int) -> bytes:
    """Go encoding/asn1 Marshal of types.Proposal (pkg/types/types.go:18-23, 50-62)."""
    def octets(b):
        return b"\x04" + _der_len(len(b)) + b
    for n in range(1, 10):                      # minimal two's complement, as encoding/asn1 emits
        try:
            iv = vseq.to_bytes(n, "big", signed=True)
            break
        except OverflowError:
            continue
    body = octets(payload) + octets(header) + octets(metadata) + b"\x02" + _der_len(len(iv)) + iv
    return b"\x30" + _der_len(len(body)) + body


def proposal_digest(payload, header, metadata, vseq) -> str:
    return hashlib.sha256(asn1_proposal(payload, header, metadata, vseq)).hexdigest()


def consenter_msg(payload, header, metadata, vseq, aux: bytes) -> bytes:
    return b"SBV1" + hashlib.sha256(asn1_proposal(payload, header, metadata, vseq)).digest() + struct.pack(">I", len(aux)) + aux


def split_infos(raw: bytes):
    parts = raw.split(b"\0")[:-1]
    return [(parts[i].decode(), parts[i + 1].decode()) for i in range(0, len(parts), 2)]


class ChainRun:
    """One run of the chain emulation (consensus_amd/host/chain_emul.cc) over n_nodes Verifiers made by
    `make_verifier()`; ledgers[node] = list of 32-byte block digests, signers[node][block] = set of signer ids."""

    def __init__(self, lib, make_verifier, n_nodes=4, blocks=9, batch_size=1, byzantine_node=0, bad_request_block=0):
        hs = [make_verifier() for _ in range(n_nodes)]
        try:
            arr = (V * n_nodes)(*hs)
            led = ctypes.create_string_buffer(n_nodes * max(1, blocks) * 32)
            lens = (ctypes.c_uint32 * n_nodes)()
            masks = (ctypes.c_uint64 * (n_nodes * max(1, blocks)))()
            cnt = (ctypes.c_uint64 * 3)()
            self.rc = lib.sbvh_chain_emulate(arr, n_nodes, blocks, batch_size, byzantine_node, bad_request_block, led, lens, masks, cnt)
            self.rejected_proposals, self.dropped_votes, self.unavailable = int(cnt[0]), int(cnt[1]), int(cnt[2])
            self.ledgers = [[led.raw[(i * blocks + b) * 32:(i * blocks + b + 1) * 32] for b in range(lens[i])] for i in range(n_nodes)]
            self.signers = [[{k + 1 for k in range(64) if masks[i * blocks + b] >> k & 1} for b in range(lens[i])] for i in range(n_nodes)]
        finally:
            for h in hs:
                lib.sbvh_verifier_free(h)

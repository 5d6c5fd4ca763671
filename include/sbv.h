/*
 * sbv.h — C-ABI of libsbv.so: MI355X (gfx950) batch signature verification for SmartBFT.
 *
 * This is the drop-in boundary below the reference's plugin seam
 *     api.Verifier            /root/reference/pkg/api/dependencies.go:54-71
 * A Go (cgo) implementation of api.Verifier — see INTEGRATION.md — parses / hashes each
 * types.Signature{ID, Value, Msg} (pkg/types/types.go:25-29) or request into one 160-byte
 * tuple and hands whole batches to the functions below.  The reference has no FFI for this
 * path (every Verifier it ships is a no-op: examples/naive_chain/node.go:86-96,
 * test/test_app.go:231-248); each entry point cites the reference call site whose work it
 * takes over.
 *
 * Conventions
 *   - every function returns SBV_OK (0) or a negative SBV_E* infrastructure error.  A return
 *     value NEVER means "signature invalid": verdicts are only in the accept bitmap, and a
 *     caller must treat <0 as "could not verify" (fall back to its own CPU path) — returning
 *     a Go error for a device fault would depose an honest leader (internal/bft/view.go:387-392).
 *   - tuple layout (big-endian, 5 x 32 bytes):  r | s | hash | Qx | Qy
 *       r, s   : signature integers (from sbv_p256_parse_der; a parse failure is encoded as
 *                r = s = 0, which the range check rejects)
 *       hash   : leftmost 32 bytes of the message digest, or the shorter digest left-padded
 *                with zeros (crypto/ecdsa hashToNat)
 *       Qx, Qy : affine public key
 *   - accept bitmap: ceil(n/8) bytes, bit (i & 7) of byte (i >> 3) = tuple i (LSB first),
 *     1 = crypto/ecdsa.VerifyASN1 would return true.
 *   - there is NO CPU fallback inside the library: without a usable gfx950 device every
 *     compute entry point returns SBV_ENODEV.
 *   - all functions are thread-safe.  There is one context per DEVICE (sbv_init(d) creates device d's; the first one
 *     initialised is the default the single-device entry points use; sbv_init_all() creates one per visible gfx950 device and
 *     the sharded / _on entries address them); device work is serialised per context.  The tuning setters
 *     (sbv_p256_set_grouping, sbv_p256_key_cache, sbv_profile_enable, sbv_shard_mode) are process-wide: they apply to every
 *     initialised device and to devices initialised later, also when called before sbv_init.
 */
#ifndef SBV_H_
#define SBV_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SBV_OK 0
#define SBV_ENODEV (-1)    /* no HIP device / not a gfx950 / runtime failure at init */
#define SBV_EINVAL (-2)    /* bad argument */
#define SBV_ENOMEM (-3)    /* device or pinned-host allocation failed */
#define SBV_EDEVICE (-4)   /* a HIP call failed during the batch (see sbv_last_error) */
#define SBV_ENOTINIT (-5)  /* sbv_init has not succeeded */
#define SBV_EPARSE (-6)    /* sbv_p256_parse_der: not a strict DER ECDSA-Sig-Value */

#define SBV_P256_TUPLE_BYTES 160

/* Initialise the context on HIP device `device` (>= 0): uploads the fixed-base table and
 * sizes scratch lazily.  Must be called once before Consensus.Start()
 * (pkg/consensus/consensus.go:107; the Verifier is injected at :35). Idempotent. */
int sbv_init(int device);
int sbv_shutdown(void);
/* Number of visible HIP devices (<0 on runtime failure). */
int sbv_device_count(void);

/* Verify n tuples in host memory; blocks until the bitmap is written.
 * Takes over: the K request signatures of VerifyProposal (internal/bft/view.go:555), the
 * coalesced VerifyConsenterSig calls of processCommits (view.go:537-541, 834-838) and
 * verifyPrevCommitSignatures (view.go:630-635), VerifyRequest (controller.go:239) under
 * Pool.Prune (controller.go:742-745), ValidateLastDecision / VerifySignature
 * (viewchanger.go:598, 660, 718, 983, 1022, 1076). */
int sbv_p256_verify_batch(const uint8_t* tuples, size_t n, uint8_t* accept_bitmap);

/* In-step key grouping for the generic entry points (consensus_amd/csrc/p256_group.h): tuples are grouped by public key on
 * the device; keys used by at least `min_count` tuples of THIS batch — or already held by the key-table cache below, however
 * few of their signatures the batch carries — (at most `max_groups` of them) get / reuse a comb table and their signatures
 * take the no-doubling kernels; everything else takes the generic kernel inside the same step.  Verdicts are identical.
 * `min_batch` = batches from this size on take the grouped step.  Defaults: enabled; min_batch 64 while the key-table cache is
 * on (a warm batch of a few thousand tuples skips the 256 doublings per signature), 2^17 while it is off (nothing outlives the
 * call then, and below ~2^17 building tables costs more latency than the doubling kernel takes); min_count 8 for P-256 (counted
 * exactly; the Ed25519 / secp256k1 steps, whose tables are always full, keep 64 and at most
 * 2048 groups); max_groups 65536 (round 5; 2048 before — a 2^20 batch over 4096 keys sent half its tuples to the one-lane kernel).
 * Passing a non-zero min_batch sets both thresholds (and the variant schemes'); SBV_GROUP_MIN_BATCH_DEFAULT restores the built-in
 * ones together with the built-in min_count and max_groups (a non-zero min_count / max_groups in the same call still applies);
 * 0 for a numeric argument keeps its current value.  Env: SBV_GROUP=0
 * disables, SBV_GROUP_MIN_BATCH=<n>.  What IS remembered between calls is the key-table cache. */
#define SBV_GROUP_MIN_BATCH_DEFAULT ((size_t)-1)
int sbv_p256_set_grouping(int enabled, size_t min_batch, uint32_t min_count, uint32_t max_groups);

/* Persistent key-table cache.  The comb a grouped batch builds for a key is a pure function of the key's 64 bytes, and
 * SmartBFT's signer sets are stable for whole epochs (pkg/types/types.go:25-29; reconfiguration:
 * pkg/consensus/consensus.go:185-252), so the tables are kept in HBM (270 KiB per key) and later batches only build the
 * tables of keys they have not met: a "warm" batch skips the doubling chains and the table kernels altogether.  Verdicts
 * cannot depend on it (a slot is found by comparing all 64 key bytes and holds exactly what the batch would have built).
 * enabled: default 1; switching it off also empties it.  capacity: keys kept (default 16384; 0 = unchanged); when full,
 * further keys are simply rebuilt per batch.  stats: out[0] = cached keys, out[1] / out[2] = groups of the last grouped
 * batch that hit / missed, out[3] = capacity.  bench.py measures its headline with the cache OFF (every step cold). */
int sbv_p256_key_cache(int enabled, uint32_t capacity);
int sbv_p256_key_cache_stats(uint32_t out[4]);
/* The same cache for every signature scheme of the library (round 4).  Consenter and client traffic on the secp256k1 and
 * Ed25519 variants is the same handful of keys forever (internal/bft/view.go:631, 834), so their grouped steps keep their
 * per-key combs too — each scheme in a pool of its OWN: a slot is found by the key bytes, and the same 64 bytes can be a
 * point of both ECDSA curves (a shared table would let a crafted key be verified against the other curve's comb).
 * scheme = SBV_SCHEME_*; defaults: on, 16384 (P-256) / 1024 (secp256k1: 270 KiB per key) / 1024 (Ed25519: 384 KiB per key)
 * keys.  With a scheme's cache on, its batches take the grouped step from 64 tuples (sbv_p256_set_grouping's min_batch).
 * sbv_p256_key_cache(e, c) == sbv_key_cache(SBV_SCHEME_P256, e, c). */
#define SBV_SCHEME_P256 0
#define SBV_SCHEME_SECP256K1 1
#define SBV_SCHEME_ED25519 2
int sbv_key_cache(int scheme, int enabled, uint32_t capacity);
int sbv_key_cache_stats(int scheme, uint32_t out[4]);

/* Same, on device-resident buffers, asynchronous on `hip_stream` (a hipStream_t; NULL = the
 * default stream).  d_tuples: n*160 bytes, 16-byte aligned.  d_bitmap: ceil(n/8) bytes.
 * The caller synchronises the stream.  Used by bench.py / multi-GPU shards. */
int sbv_p256_verify_batch_dev(const void* d_tuples, size_t n, void* d_bitmap, void* hip_stream);

/* ---- registered public keys ---------------------------------------------------------------------
 * SmartBFT's consenter set (and an application's client set) is a registry: types.Signature.ID
 * selects the key (pkg/types/types.go:25-29), it is not carried per signature.  Registering a key
 * builds a fixed-base comb for it (33 x 128 affine multiples, 270 KiB of HBM per key) once, after
 * which a verification against that key needs NO point doublings: R = u1*G + u2*Q is 13 + 32.2 = 45.2 mixed
 * additions on average (13 from the 20-bit comb of G, 32 key-comb windows + the rarely needed carry window; ~5x fewer field
 * multiplications than the generic form).  Batches of at most 32 signatures take a one-launch latency form (stage A of the
 * call on the calling CPU thread with one inversion, records and verdicts in mapped host memory, 16 lanes per signature), up to 32768 the 8-lanes-per-signature kernel.  Verdicts are identical to
 * the generic entry points: a key that crypto/ecdsa would refuse (coordinate >= p, off curve) still
 * gets a slot, flagged invalid, and every signature against it is rejected.
 *   keys: m x 64 bytes (Qx|Qy big-endian).  slots_out[i] = slot of keys[i] (equal keys share a slot). */
int sbv_p256_register_keys(const uint8_t* keys, size_t m, uint32_t* slots_out);
int sbv_p256_key_count(void);
int sbv_p256_clear_keys(void);
/* Wide combs for the consenters' keys (round 4).  The consenters of a cluster are a handful of keys that sign every vote of every
 * decision for a whole epoch (pkg/types/types.go:25-29; internal/bft/view.go:531-541 collects 2f+1 of their commit signatures per
 * decision, view.go:631 and :834 verify them; reconfiguration replaces the set: pkg/consensus/consensus.go:185-252), and HBM holds
 * 288 GB: sbv_p256_widen_keys gives the named registered slots a second, `bits`-wide comb (the layout of the comb of G), after
 * which u2*Q is ceil(257 / bits) additions instead of 32.2 — 16 bits: 16 additions, 35.7 MB per key; 20 bits: 13 additions,
 * 436 MB per key.  A wavefront whose signatures all belong to wide slots takes the wide
 * combs, any other the 8-bit combs every key keeps; verdicts are identical.  Slots that are wide already are skipped, slots
 * beyond `max_keys` wide ones stay narrow (no error); an unregistered slot is SBV_EINVAL.  On a device fault the call returns
 * SBV_EDEVICE / SBV_ENOMEM and leaves the registry as it was: the named slots keep their 8-bit combs and may be named again.
 * sbv_p256_wide_keys sets the width and the cap for every device.  Default: bits = SBV_WIDE_BITS_AUTO, 64 keys — 20-bit combs
 * while at most 16 slots are wide (a 16-node cluster: 7 GB), 16-bit combs beyond (64 keys: 2.3 GB); crossing the line rebuilds
 * what was there, on the device, in milliseconds.  An explicit width (10..20) is kept whatever the count; bits = 0 switches the
 * feature off and frees the tables; another width rebuilds the combs of the slots already widened.  Env: SBV_KEYED_WIDE_BITS
 * (0 = off, 1 = auto), SBV_KEYED_WIDE_MAX.  stats: out[0] = wide slots, out[1] = bits, out[2] = max_keys, out[3] = KiB per key. */
#define SBV_WIDE_BITS_AUTO 1
int sbv_p256_wide_keys(int bits, uint32_t max_keys);
int sbv_p256_widen_keys(const uint32_t* slots, size_t m);
int sbv_p256_wide_key_stats(uint32_t out[4]);
/* The combs are built on the device (consensus_amd/csrc/p256_widetab29.h: two launches for all the keys of a call: 0.01 s for 16
 * keys at 16 bits, 0.03 s at 20 bits, measured); SBV_KEYED_WIDE_HOST=1 selects the host builder.  Diagnostics: sbv_p256_wide_selfcheck(slot) = 1 when the
 * device-resident comb of `slot` equals the host builder's output byte for byte, 0 when it differs. */
int sbv_p256_wide_selfcheck(uint32_t slot);
/* rsh: n x 96 bytes (r|s|hash, big-endian), slots: n key slots.  Takes over VerifyConsenterSig
 * (view.go:631, 834), VerifySignature (viewchanger.go:598...) and decision replay for registered
 * signers.  An out-of-range slot is a reject, not an error. */
int sbv_p256_verify_batch_keyed(const uint8_t* rsh, const uint32_t* slots, size_t n, uint8_t* accept_bitmap);
int sbv_p256_verify_batch_keyed_dev(const void* d_rsh, const void* d_slots, size_t n, void* d_bitmap, void* hip_stream);

/* Batch signing: the batch form of api.Signer.Sign / SignProposal (pkg/api/dependencies.go:46-52;
 * examples and tests sign with crypto/ecdsa over SHA-256).  Signature i = ECDSA-P256(key[key_index[i]], digests[i]) with the
 * deterministic nonce of RFC 6979 (HMAC-SHA256), bit-identical to consensus_amd/host's Signer; key_index == NULL means key
 * i % n_keys.  keys: n_keys x 32 bytes (private scalar, big-endian); digests: n x 32 bytes (SHA-256 of the message);
 * sigs: n x 64 bytes r | s (big-endian; DER-encode for types.Signature.Value); ok[i] = 0 when key i is not in [1, N-1] or the
 * index is out of range (sigs[i] is then all zero).  NOT constant-time (secret-indexed table lookups in HBM): for test
 * traffic and trusted single-tenant hosts, see consensus_amd/csrc/p256_sign.h. */
int sbv_p256_sign_batch(const uint8_t* keys, uint32_t n_keys, const uint32_t* key_index, const uint8_t* digests, size_t n,
                        uint8_t* sigs, uint8_t* ok);
int sbv_p256_sign_batch_dev(const void* d_keys, uint32_t n_keys, const void* d_key_index, const void* d_digests, size_t n,
                            void* d_sigs, void* d_ok, void* hip_stream);

/* Message front end on the device: SHA-256 of each message and the strict DER
 * parse of each signature run as a kernel in front of the registered-key verification, so the host
 * only concatenates bytes.  msg i = msgs[msg_offsets[i] .. msg_offsets[i+1]), signature i (ASN.1 DER,
 * types.Signature.Value) = sigs[sig_offsets[i] .. sig_offsets[i+1]), signer = slots[i].
 * Equivalent to crypto/ecdsa.VerifyASN1(key[slots[i]], sha256(msg i), sig i) for every i. */
int sbv_p256_verify_msgs_keyed(const uint8_t* msgs, const uint64_t* msg_offsets, const uint8_t* sigs,
                               const uint64_t* sig_offsets, const uint32_t* slots, size_t n, uint8_t* accept_bitmap);

/* ---- secp256k1 variant ("other curves": api.Verifier / api.Signer are curve-agnostic,
 * pkg/api/dependencies.go:46-71) --------------------------------------------------------------------------------------
 * The same 160-byte tuples r | s | hash | Qx | Qy and the same rules as sbv_p256_verify_batch (1 <= r, s <= n - 1, key
 * coordinates < p and on the curve, hash = leftmost 32 bytes reduced mod n, R = infinity rejected, accept iff
 * R.x mod n == r; no low-S rule at this layer) over y^2 = x^3 + 7, p = 2^256 - 2^32 - 977.  One lane per signature
 * (consensus_amd/csrc/k256_core.h); the 35.7 MB comb of G is built on the first call of a process. */
int sbv_secp256k1_verify_batch(const uint8_t* tuples, size_t n, uint8_t* accept_bitmap);
int sbv_secp256k1_verify_batch_dev(const void* d_tuples, size_t n, void* d_bitmap, void* hip_stream);

/* ---- Ed25519 variant (BASELINE.json configs[4]) ----------------------------------------------------
 * Semantics of Go crypto/ed25519.Verify(pk, msg, sig) (crypto/internal/edwards25519): S canonical,
 * A decoded with Go's leniency (non-canonical y accepted, no small-order rejection), cofactorless
 * [S]B = R + [k]A checked by re-encoding R byte-wise.
 * Tuple, 128 bytes little-endian as on the wire:  R (32) | S (32) | public key (32) | k (32), where
 * k = SHA-512(R || pk || msg) mod L is produced by sbv_ed25519_make_tuples (host) or, for the raw-message
 * entry sbv_ed25519_verify_msgs, on the device.  A tuple with k >= L or S >= L is rejected.  len(sig) != 64 is
 * the caller's reject (encode S = 2^256 - 1). */
#define SBV_ED25519_TUPLE_BYTES 128
int sbv_ed25519_verify_batch(const uint8_t* tuples, size_t n, uint8_t* accept_bitmap);
int sbv_ed25519_verify_batch_dev(const void* d_tuples, size_t n, void* d_bitmap, void* hip_stream);
/* Builds the tuples: sigs n x 64, pks n x 32, msgs packed (offsets[i]..offsets[i+1]) -> tuples n x 128. */
int sbv_ed25519_make_tuples(const uint8_t* sigs, const uint8_t* pks, const uint8_t* msgs, const uint64_t* offsets,
                            size_t n, uint8_t* tuples_out);
/* Raw-message entry: the same inputs as sbv_ed25519_make_tuples, hashed on the device (SHA-512 + reduction mod L per
 * lane, consensus_amd/csrc/sha512_dev.h) and verified in the same call; n <= 2^21.  What a Verifier's
 * VerifyProposal / decision replay hands over when signatures are Ed25519 (internal/bft/view.go:555). */
int sbv_ed25519_verify_msgs(const uint8_t* sigs, const uint8_t* pks, const uint8_t* msgs, const uint64_t* msg_offsets,
                            size_t n, uint8_t* accept_bitmap);

/* Strict DER parse of an ECDSA-Sig-Value with Go x/crypto/cryptobyte rules
 * (crypto/ecdsa.parseSignature): out = r | s, 32 bytes each, big-endian, zero padded.
 * Returns SBV_OK or SBV_EPARSE (then out is all zero, which every verify rejects). */
int sbv_p256_parse_der(const uint8_t* der, size_t len, uint8_t out_rs[64]);

/* SHA-256 of n messages packed back to back (offsets[i]..offsets[i+1]) -> n*32 bytes.
 * Host implementation used to build tuples (hash = SHA-256(Signature.Msg)).  On x86-64 CPUs with the SHA extensions the
 * compression function runs on them (as Go's crypto/sha256 does); sbv_sha256_uses_cpu_extensions() says which one is in use
 * (SBV_SHA_PORTABLE=1 in the environment keeps the portable loop). */
int sbv_sha256_batch(const uint8_t* msgs, const uint64_t* offsets, size_t n, uint8_t* out_hashes);
int sbv_sha256_uses_cpu_extensions(void);

typedef struct sbv_timing {
    double h2d_us;      /* host -> device copy of the tuples   (host-pointer entry only) */
    double prep_us;     /* stage A kernel (range checks, s^-1, u1, u2)                   */
    double verify_us;   /* stage B kernel (u1*G + u2*Q, final comparison)                */
    double d2h_us;      /* bitmap device -> host                                          */
    double total_us;    /* wall clock of the whole call                                   */
    uint64_t n;         /* tuples in the call                                             */
} sbv_timing;
/* Timing of the most recent sbv_p256_verify_batch call made by any thread. */
int sbv_last_timing(sbv_timing* out);

/* Per-kernel timing of the device-pointer entry (bench.py's roofline leg).  While enabled,
 * every sbv_p256_verify_batch_dev call records HIP events around its two kernels ON THE
 * CALLER'S STREAM; sbv_profile_read waits for them and returns the sums (microseconds) and the
 * number of stage-B launches since the previous read.
 * on = 1: an event triple per call (before stage A, after stage A, after stage B) AND a pair around every launch of the
 * dominant kernel; on = 2: the dominant-kernel pairs only (what bench.py keeps inside its timed region: every recorded event
 * is a packet between two kernels of the step); on = 0: off.  Process-wide: applies to every initialised device and to
 * devices initialised later. */
int sbv_profile_enable(int on);
int sbv_profile_read(double* prep_us, double* verify_us, uint64_t* launches);
/* Same window as sbv_profile_read (call it BEFORE sbv_profile_read, which resets): summed duration and number
 * of launches of the dominant stage-B kernel alone — k_verify_keyed_q (one launch per chunk of key-comb windows; k_ed_qphase /
 * k_k256_qphase for the grouped steps of the device-pointer entries of the other two schemes)
 * when the batch was grouped, else k_p256_verify. */
int sbv_profile_read_dominant(double* dominant_us, uint64_t* dominant_launches);
/* Device-side counters of the most recent grouped batch: out[0] = key groups, out[1] = tuples verified through
 * the per-batch key tables, out[2] = tuples verified by the generic kernel, out[3] = ungrouped tuples rejected
 * for their public key alone (pointFromAffine: coordinate >= p or off the curve).  Synchronises the device. */
int sbv_p256_last_group_stats(uint32_t out[4]);
/* Table classes of the most recent grouped P-256 batch (round 5; consensus_amd/csrc/p256_group.h).  Every grouped key gets its ROWS —
 * the babies b * 2^(8j) Q and giants 16 a * 2^(8j) Q of every window: a comb with 4-bit windows, two additions per window — which pay
 * from ~4 signatures per key; the FILL that completes the 8-bit comb (three quarters of a table's cost, one addition per window) is
 * spent on keys that sign at least 256 tuples of the batch (SBV_FULL_TABLE_MIN), also later: a key cached with rows only is upgraded
 * by the first batch in which it is hot.  out[0] = groups verified from a full table, out[1] = groups whose table was filled in this
 * batch, out[2] = grouped tuples served by the rows-only pass.  Verdicts never depend on the class.  Synchronises the device.
 * K arbitrary clients per proposal: internal/bft/view.go:553-559. */
int sbv_p256_last_table_classes(uint32_t out[3]);
/* Pools of the grouped P-256 step on the default device (round 6).  The defaults — a key-table cache of 16 384 keys and 65 536 groups per
 * batch: 25 GB of combs — are sized for a 288 GB MI355X; a device that cannot hold them (a shared or smaller part, one of several
 * contexts folded onto one GPU by SBV_LOGICAL_DEVICES) gets halved pools instead of SBV_ENOMEM, and a batch whose pools cannot be
 * allocated at all is verified by the one-lane kernel: a Verifier must stay live (internal/bft/view.go:387-392 deposes a leader on a
 * VerifyProposal error).  out[0] = cached keys the pool holds, out[1] = groups per batch, out[2] = 1 when either is smaller than asked
 * for, out[3] = grouped batches that fell back to the one-lane kernel for lack of memory, out[4] = combs of the hot-key pool,
 * out[5] = contexts sharing this GPU.  out[0..1] are 0 before the first grouped batch.  Rates change with the pools, verdicts never. */
int sbv_p256_pool_stats(uint32_t out[6]);
/* Diagnostics (tests / tools): every promoted hot-key comb of context `device` compared with the host builder, and the consistency of the
 * pool's bookkeeping.  out[0] promoted slots, [1] combs that differ, [2] the first such comb, [3] its first differing entry, [4] how many of
 * its entries differ, [5] index / owner inconsistencies, [6] combs claimed by more than one slot, [7] combs handed out so far.  Slow. */
int sbv_debug_hot_check(int device, uint32_t out[8]);
/* Hot keys (round 5): wide combs in the GENERIC path.  A key that arrives inside tuples — a client key of VerifyProposal
 * (internal/bft/view.go:553-559), a consenter of a replica that registered nothing — and keeps being hit is promoted: once the
 * key-table cache has verified `min_hits` tuples against its slot, a 16-bit comb (35.7 MB; up to `max_keys` of them, default 1024 =
 * 36.5 GB of the 288, less if the device lacks the room) is built on the device behind a batch's verdicts, at most 64 keys per batch,
 * from base points the slot's 8-bit table already holds.  Later batches verify its tuples in 13 + 17 comb additions instead of
 * 13 + 32, in one launch that needs no table of the batch.  Needs the key-table cache (on by default); verdicts never depend on it.
 * max_keys = 0 switches it off; min_hits = 0 keeps the current value (default 4096).  Env: SBV_HOT_KEYS, SBV_HOT_MIN_HITS.
 * stats: out[0] = promoted keys, out[1] = pool capacity, out[2] = tuples the wide pass served in the last grouped batch,
 * out[3] = min_hits.  sbv_p256_hot_selfcheck(i) = 1 when promoted comb i equals the host builder's output for its key. */
int sbv_p256_hot_keys(uint32_t max_keys, uint32_t min_hits);
int sbv_p256_hot_key_stats(uint32_t out[4]);
int sbv_p256_hot_selfcheck(uint32_t index);
/* The same for the Ed25519 variant (round 6): a cache slot of that scheme whose count passes `min_hits` gets a 16-bit comb of -A
 * (16 windows x 32 768 affine-Niels entries at a 128-byte pitch = 64 MiB), built on the device from base points the slot's 8-bit comb
 * holds; later batches add [k](-A) for its tuples in 16 additions instead of 32, in one launch that needs no table of the batch.  Same
 * bookkeeping as above (counts per slot, decay, eviction with hysteresis).  Default: up to 1024 keys (64 GiB of the 288; fewer if the
 * device lacks the room) from 4096 hits on; max_keys = 0 switches it off; env SBV_ED_HOT_KEYS, SBV_ED_HOT_MIN_HITS.  A comb costs
 * about 80 us of device time to build, behind a batch's verdicts, and saves about 0.5 ns per tuple verified from it afterwards: it
 * pays for signers that stay (consenters), which is what the decaying count selects.  Verdicts never depend on it.  stats /
 * selfcheck as for P-256. */
int sbv_ed25519_hot_keys(uint32_t max_keys, uint32_t min_hits);
int sbv_ed25519_hot_key_stats(uint32_t out[4]);
int sbv_ed25519_hot_selfcheck(uint32_t index);

/* Page-locked host memory for the host-pointer entries.  Handing pageable memory to a 100 MB batch makes the HIP
 * runtime pin (or bounce) it inside the call — measured at 25 ms for a 550 000-signature replay batch whose kernels
 * take 5 ms.  A caller that lays its batch out in sbv_host_alloc memory gets a plain DMA.  NULL on failure (or
 * before sbv_init); sbv_host_free accepts NULL. */
void* sbv_host_alloc(size_t bytes);
void sbv_host_free(void* p);

/* ---- every GPU of the node behind one process ---------------------------------------------------------------------
 * The reference injects ONE Verifier into a replica process (pkg/consensus/consensus.go:35, 107); that process drives all
 * the GPUs of its node.  sbv_init(d) may be called for several devices (the first one stays the default of the
 * single-device entry points above); sbv_init_all() initialises every visible gfx950 device and returns how many.
 *
 * sbv_p256_verify_batch_sharded splits a host batch into contiguous shards, one per device (BASELINE.json: "client-request
 * signatures queued in the RequestPool and the 2f+1 consenter Commit signatures collected per proposal are sharded across
 * the 8 GPUs of one node"), each a multiple of lcm(512, 8 * group) tuples: whole bitmap bytes per device and — with
 * group = signatures per proposal (11 at N = 16: internal/bft/util.go:183-187) — whole proposals, so that each device
 * also emits the per-proposal quorum bit (>= quorum accepted signatures by DISTINCT keys, the rule of
 * internal/bft/viewchanger.go:681-727).  When more than one device took part, one in-place ncclAllGather of the bitmap
 * shards (RCCL over xGMI, uint8, per-device streams) leaves the full bitmap on every device and one D2H returns it (with
 * fewer shards than devices the idle ranks of the node-wide communicator join with an unused slot); a batch smaller than
 * twice the per-device minimum is NOT split — it goes whole to one device, round-robin, with no collective.  The minimum is
 * chosen per batch (sbv_shard_min_for: 2^16 tuples when a sample of the batch shows a handful of signers — consenter commit
 * signatures, configs[3]: 550 000 tuples over 16 keys span 8 devices, 68 750 each — and 2^17 otherwise; SBV_SHARD_MIN
 * overrides both).
 * Each device takes its shard in pieces through two upload slots: the host -> device copy of piece i + 1 runs on a copy stream
 * beside the kernels of piece i (pieces of 2^18 tuples while the key-table cache is on — the first piece builds the signers'
 * combs, the later ones find them — and whole launches when it is off; SBV_SHARD_PIECE overrides the size).
 *   group = 0: plain tuples.  quorum_bitmap may be NULL; otherwise ceil((n / group) / 8) bytes, bit p = proposal p.
 *   info (optional) reports what was done.  sbv_shard_plan is the pure split (testable without a GPU):
 *   first[0..shards], returns shards.  sbv_p256_verify_batch_on runs a whole batch on one chosen device. */
typedef struct sbv_shard_info {
    int devices;              /* devices available to the call                              */
    int shards;               /* shards the batch was split into (1 = not split)            */
    int mode;                 /* 0 = one device, 1 = RCCL all-gather, 2 = per-device D2H, 3 = key-affine + RCCL all-reduce, 4 = key-affine + host OR */
    size_t tuples_per_shard;
    double h2d_us;            /* slowest device: the copies' own durations, summed over its pieces           */
    double kernels_us;        /* slowest device: end of its first copy -> end of its last kernel (stage A + B,
                                 quorum bits; waits for later copies included)                                */
    double gather_us;         /* all-gather + final D2H                                      */
    double total_us;
} sbv_shard_info;
/* SBV_LOGICAL_DEVICES=G (environment, read at initialisation; round 6): sbv_init accepts G device indices and sbv_init_all creates G
 * contexts, context i on HIP device i % (visible devices), every pool budget divided by the contexts that share a GPU.  One MI355X then
 * runs the sharded entries exactly as a G-GPU node would — G shards from G host threads, shard offsets > 0, idle contexts, the bitmap
 * gathered through the host (an RCCL communicator needs one rank per physical device) — which is how the one-GPU test tier covers them.
 * sbv_init_all initialises its devices in parallel, one host thread each. */
int sbv_init_all(void);
int sbv_initialised_devices(int* out, int max);
size_t sbv_shard_plan(size_t n, int devices, size_t group, size_t min_per_device, size_t* first);
/* The per-device minimum the sharded entry plans THIS host batch with (pure host code, no device needed): the public keys of
 * 256 evenly spaced tuples are compared; <= 32 distinct ones = few signers.  Reference: the traffic it tells apart is
 * ValidateLastDecision / decision replay (internal/bft/viewchanger.go:681-727: N consenters) against VerifyProposal's K
 * client requests (internal/bft/view.go:553-559). */
size_t sbv_shard_min_for(const uint8_t* tuples, size_t n, size_t group);
int sbv_p256_verify_batch_sharded(const uint8_t* tuples, size_t n, size_t group, uint32_t quorum, uint8_t* accept_bitmap,
                                  uint8_t* quorum_bitmap, sbv_shard_info* info);
int sbv_p256_verify_batch_on(int device, const uint8_t* tuples, size_t n, uint8_t* accept_bitmap);
/* The registered-key form of the sharded entry (round 5): configs[3] as BASELINE.json words it — "11 consenter sigs x 50k proposals
 * sharded over 8 GPUs" — on the consenters' resident combs.  rsh: n x 96 bytes r | s | hash, slots: n key slots (100 B per signature
 * over PCIe instead of 160).  The key registry is process-wide: sbv_p256_register_keys / sbv_p256_widen_keys replicate a key's 8-bit
 * comb and a consenter's wide comb onto every initialised device (a device initialised later, or one whose replication failed, is
 * brought up to date in front of the call; if that fails the call returns < 0 — a stale device never answers "invalid"), so a slot
 * means the same key wherever its shard lands.  Plan, pieces (2^17 signatures; SBV_SHARD_PIECE_KEYED), collective and info as for
 * sbv_p256_verify_batch_sharded with the few-signers minimum (2^16 per device); quorum bit p = proposal p carries >= quorum accepted
 * signatures by DISTINCT slots (equal keys share a slot), the rule of internal/bft/viewchanger.go:681-727 for the signatures
 * internal/bft/view.go:531-541 collects.  Also the pipelined host-pointer entry for large registered-key batches on ONE device
 * (sbv_init only): the upload of piece i + 1 runs beside the kernels of piece i.  An out-of-range slot is a reject, not an error. */
int sbv_p256_verify_batch_keyed_sharded(const uint8_t* rsh, const uint32_t* slots, size_t n, size_t group, uint32_t quorum,
                                        uint8_t* accept_bitmap, uint8_t* quorum_bitmap, sbv_shard_info* info);
/* ... and its raw-messages form: sbv_p256_verify_msgs_keyed's inputs (messages and DER signatures packed back to back, their offset
 * tables, key slots) through the same plan — SHA-256 and the strict DER parse run on every device over its share, piece by piece
 * beside the uploads, so a replaying replica's host pass only lays bytes out (decision replay: internal/bft/controller.go:587-633;
 * the signatures of a decision: pkg/types/types.go:31-34).  No 2^21 limit: pieces are launches.  The offset tables need not start
 * at 0 (msgs / sigs are the bases they refer to: a slice of a larger batch's tables is a valid argument).  Equivalent to
 * crypto/ecdsa.VerifyASN1(key[slots[i]], sha256(msg i), sig i) for every i; quorum bits as above. */
int sbv_p256_verify_msgs_keyed_sharded(const uint8_t* msgs, const uint64_t* msg_offsets, const uint8_t* sigs, const uint64_t* sig_offsets,
                                       const uint32_t* slots, size_t n, size_t group, uint32_t quorum, uint8_t* accept_bitmap,
                                       uint8_t* quorum_bitmap, sbv_shard_info* info);
/* Key-affine partition (the other way to spread a batch: by signer instead of by position).  A contiguous split hands every
 * device signatures of every signer, so every device builds every key's tables — the part of a cold step that does not shrink
 * with the shard.  With sbv_shard_mode(1, parts) the sharded entry partitions by a hash of the public key: part p (on device
 * p % devices; parts = 0 means one per device, more parts than devices run one after another — how a single GPU rehearses an
 * 8-GPU partition) verifies the tuples of "its" keys only.  Every participating device receives the WHOLE batch over its own
 * PCIe link (the price of not bucketing on the host); the per-device bitmaps have disjoint bits and are combined with one
 * in-place ncclAllReduce(sum, uint8) (info->mode 3) or an OR on the host (mode 4); quorum bits are computed on the first device
 * from the combined bitmap.  Env: SBV_SHARD_MODE=keys, SBV_SHARD_PARTS=<n>.  Same verdicts as every other entry.
 * sbv_p256_verify_batch_dev_part is the device-resident building block: part `part` of `parts` of n tuples already in HBM
 * (default device), asynchronous on hip_stream except for ONE internal synchronisation (the member count sizes the launches);
 * d_bitmap_words: ceil(n / 32) 32-bit words, 4-byte aligned, receives this part's verdict bits (all other bits 0);
 * *part_tuples (optional) = how many tuples the part held.  Call sites: the same as sbv_p256_verify_batch_sharded
 * (VerifyProposal's K request signatures, decision replay: internal/bft/view.go:553-559, controller.go:587-633). */
int sbv_shard_mode(int by_key, unsigned parts);
int sbv_p256_verify_batch_dev_part(const void* d_tuples, size_t n, uint32_t part, uint32_t parts, void* d_bitmap_words,
                                   void* hip_stream, size_t* part_tuples);

/* Human-readable description of the calling thread's last failing call ("" if none). */
const char* sbv_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* SBV_H_ */

// verifier.h — C++ mirror of the reference's plugin pair for this path:
//   api.Verifier  pkg/api/dependencies.go:54-71   (same method names, argument meaning, error behaviour)
//   api.Signer    pkg/api/dependencies.go:46-52
// written in C++ because the reference is compiled code and this image has no Go toolchain; the Go
// cgo adapter a maintainer would add is in INTEGRATION.md and has exactly this structure.
//
// Error behaviour: Go's `error == nil` <=> Status::OK.  Status::INVALID is the Go `error` the
// protocol acts on (vote dropped: internal/bft/view.go:839-842; proposal rejected and leader accused:
// view.go:387-392; request not pooled: controller.go:239-245).  Status::UNAVAILABLE means the
// backend could not answer (device fault, library missing): the Go adapter then verifies with stock
// crypto/ecdsa — this C++ mirror has no CPU verifier on purpose and surfaces the condition instead
// of ever turning it into INVALID.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>

#include "formats.h"

namespace sbvhost {

struct Status {
    enum Code { OK = 0, INVALID = 1, UNAVAILABLE = 2 } code = OK;
    std::string msg;
    bool ok() const { return code == OK; }
    static Status Ok() { return Status(); }
    static Status Invalid(const std::string& m) { Status s; s.code = INVALID; s.msg = m; return s; }
    static Status Unavailable(const std::string& m) { Status s; s.code = UNAVAILABLE; s.msg = m; return s; }
};

// What sits below the seam: n tuples -> accept bitmap; returns 0 or a negative infrastructure error
// (include/sbv.h convention).  The product backend is libsbv.so; tests may inject a stand-in, the
// way the reference's own tests inject mocks.VerifierMock (internal/bft/mocks/verifier_mock.go).
class Backend {
 public:
    virtual ~Backend() {}
    virtual int verify(const uint8_t* tuples, size_t n, uint8_t* bitmap) = 0;
    // Optional registered-key form (include/sbv.h: sbv_p256_register_keys / verify_batch_keyed).
    // register_key returns a slot >= 0, or -1 when the backend has no key registry (then callers
    // use verify() with the key carried in the tuple).
    virtual long register_key(const uint8_t q[64]) { (void)q; return -1; }
    // The slot belongs to a consenter: a key that signs every vote of every decision of the epoch (internal/bft/view.go:531-541,
    // 631, 834).  The product backend gives it a wide comb (include/sbv.h: sbv_p256_widen_keys); a no-op elsewhere.
    virtual void widen_key(long slot) { (void)slot; }
    virtual uint64_t keyed_batches() { return 0; }      // test hook: how many verify_keyed batches ran
    virtual uint64_t widened_keys() { return 0; }       // test hook: how many distinct slots widen_key was given
    // Page-locked staging memory (include/sbv.h: sbv_host_alloc); nullptr = none available, the caller uses the heap
    virtual void* host_alloc(size_t bytes) { (void)bytes; return nullptr; }
    virtual void host_free(void* p) { (void)p; }
    // Ed25519 variant (include/sbv.h: sbv_ed25519_verify_batch): n tuples of 128 bytes R|S|A|k.  -2 when unsupported.
    virtual int verify_ed25519(const uint8_t* tuples128, size_t n, uint8_t* bitmap) { (void)tuples128; (void)n; (void)bitmap; return -2; }
    // secp256k1 variant (include/sbv.h: sbv_secp256k1_verify_batch): n tuples of 160 bytes, same layout as verify().  -2 when unsupported.
    virtual int verify_k256(const uint8_t* tuples, size_t n, uint8_t* bitmap) { (void)tuples; (void)n; (void)bitmap; return -2; }
    virtual int verify_keyed(const uint8_t* rsh, const uint32_t* slots, size_t n, uint8_t* bitmap) {
        (void)rsh; (void)slots; (void)n; (void)bitmap; return -2;
    }
    // Optional device front end (sbv_p256_verify_msgs_keyed): raw messages + DER signatures + slots.
    // Returns -2 when unsupported.
    virtual int verify_msgs_keyed(const uint8_t* msgs, const uint64_t* moff, const uint8_t* sigs, const uint64_t* soff,
                                  const uint32_t* slots, size_t n, uint8_t* bitmap) {
        (void)msgs; (void)moff; (void)sigs; (void)soff; (void)slots; (void)n; (void)bitmap; return -2;
    }
};
std::shared_ptr<Backend> make_sbv_backend(int device);     // device >= 0: sbv_init(device); device < 0: sbv_init_all() + the sharded entry
typedef int (*backend_fn)(const uint8_t* tuples, size_t n, uint8_t* bitmap, void* user);
std::shared_ptr<Backend> make_callback_backend(backend_fn fn, void* user, bool with_key_registry = false);

// A lock for critical sections of a few dozen instructions that N-1 threads hit in the same microsecond (a burst of commit
// votes: view.go:537-541).  Under that pattern std::mutex parks the losers in the kernel — one futex wake-up (5-10 us) per
// hand-over, 15-24 us until the last of 15 votes was queued (profiles/r04/m2_trace_r04h.txt) — a spinning lock hands over in
// ~100 ns.  BasicLockable: works with std::lock_guard / std::unique_lock.
class SpinLock {
 public:
    void lock() {
        // test-and-test-and-set: the waiters spin on a shared (read-only) copy of the line and only the one that sees it free
        // tries the exchange, so a hand-over is one line transfer instead of a storm of them
        while (f_.exchange(true, std::memory_order_acquire)) {
            while (f_.load(std::memory_order_relaxed)) {
#if defined(__x86_64__) || defined(__i386__)
                __builtin_ia32_pause();
#endif
            }
        }
    }
    void unlock() { f_.store(false, std::memory_order_release); }

 private:
    std::atomic<bool> f_{false};
};

struct CoalescerStats {
    uint64_t calls = 0;        // single-signature submissions
    uint64_t batches = 0;      // backend invocations
    uint64_t max_batch = 0;
};

// Collects concurrent single-tuple submissions into one backend batch: the first submitter of a burst becomes its
// leader (no dispatcher thread to wake), waits up to `max_wait` for more (or until `max_batch`), and ships them in one
// call.  This is what turns the <= N-1 goroutines of View.processCommits (view.go:537-541) into one
// GPU micro-batch.  submit_many() bypasses the wait (a VerifyProposal already is a batch).
class Coalescer {
 public:
    Coalescer(std::shared_ptr<Backend> be, size_t max_batch, std::chrono::microseconds max_wait);
    ~Coalescer();
    // 1 = accept, 0 = reject, <0 = backend error
    // slot >= 0: the signer's key is registered with the backend (tuple[96..160) is then ignored)
    // ed25519 = true: `tuple` holds a 128-byte R|S|A|k tuple instead (one Verifier = one scheme, so a burst never mixes)
    // err (optional): on a backend error, the library's error text as the DISPATCHER thread saw it (sbv_last_error() is per
    // thread, and the failing call ran there, not on the submitter)
    int submit(const uint8_t tuple[160], long slot = -1, bool ed25519 = false, bool k256 = false, std::string* err = nullptr);
    int submit_many_ed25519(const uint8_t* tuples128, size_t n, uint8_t* bitmap);
    int submit_many_k256(const uint8_t* tuples, size_t n, uint8_t* bitmap);
    int submit_many(const uint8_t* tuples, size_t n, uint8_t* bitmap);
    int submit_many_keyed(const uint8_t* rsh, const uint32_t* slots, size_t n, uint8_t* bitmap);
    // raw messages + DER signatures + slots through the backend's front end (-2 when it has none)
    int submit_many_msgs_keyed(const uint8_t* msgs, const uint64_t* moff, const uint8_t* sigs, const uint64_t* soff, const uint32_t* slots, size_t n,
                               uint8_t* bitmap);
    Backend& backend() { return *be_; }
    CoalescerStats stats();
    // How many submissions a burst is expected to bring (N-1 commit votes: view.go:537-541); the dispatcher ships as soon as
    // that many are queued instead of sitting out the window.  0 = unknown.
    void set_burst_hint(size_t n) { burst_hint_.store(n, std::memory_order_relaxed); }

 private:
    struct Job { uint8_t tuple[160]; long slot = -1; bool ed25519 = false; bool k256 = false; int result = -100; std::string err; std::atomic<bool> done{false}; double t_push = 0; };
    void serve_as_leader(const std::atomic<bool>* own_done);
    std::shared_ptr<Backend> be_;
    size_t max_batch_;
    std::chrono::microseconds max_wait_;
    SpinLock mu_;                           // the queue, the leader flag, the statistics
    std::mutex sleep_mu_;                   // only for submitters that gave up spinning (cv_done_)
    std::condition_variable cv_done_;
    std::atomic<int> sleepers_{0};
    bool leader_ = false;                   // under mu_: some submitter is serving the queue
    std::atomic<bool> leader_flag_{false};  // the same, readable by the spinning submitters without the lock
    std::deque<Job*> q_;
    std::atomic<size_t> qn_{0};            // q_.size(), readable without the lock by the spinning dispatcher
    std::atomic<size_t> burst_hint_{0};
    CoalescerStats st_;
};

// Signature scheme of a Verifier / Signer pair.  P256: ECDSA over SHA-256, DER signatures, 64-byte Qx|Qy keys (Go
// crypto/ecdsa.VerifyASN1).  ED25519: the BASELINE.json configs[4] variant — 64-byte R|S signatures, 32-byte keys
// (Go crypto/ed25519.Verify); registered-key slots do not apply (the device groups by key inside each batch).
// SECP256K1: ECDSA over SHA-256 with DER signatures and 64-byte keys exactly like P256, on the other curve
// (include/sbv.h: sbv_secp256k1_verify_batch); no registered-key slots yet (every tuple carries its key).
enum class Scheme { P256 = 0, ED25519 = 1, SECP256K1 = 2 };

struct VerifierOptions {
    Scheme scheme = Scheme::P256;
    size_t coalesce_max = 4096;
    std::chrono::microseconds coalesce_wait{50};
    bool cache_verified = true;     // commit sigs of sequence s reappear at s+1 (view.go:376, 630)
    // false: client keys are NOT given comb slots on the device; request signatures then travel as generic tuples with the
    // key inline and the device groups them by key inside each batch (what an open client population gets: a replica
    // cannot pre-register keys it has never seen).  Consenter keys are always registered.
    bool device_client_keys = true;
};

class Verifier {
 public:
    Verifier(std::shared_ptr<Backend> be, const VerifierOptions& opt = VerifierOptions());
    ~Verifier();

    // key registry (id -> public key: 64 bytes Qx|Qy, or the first 32 bytes = A_enc under Scheme::ED25519); the kernel
    // re-validates every key
    void RegisterConsenter(uint64_t id, const uint8_t* q);
    void RegisterClient(const std::string& client_id, const uint8_t* q);
    void SetVerificationSequence(uint64_t s);
    void SetDeviceClientKeys(bool on) { std::lock_guard<SpinLock> lk(mu_); opt_.device_client_keys = on; }     // applies to clients registered afterwards

    // ---- api.Verifier ---------------------------------------------------------------------------
    Status VerifyProposal(const Proposal& proposal, std::vector<RequestInfo>* requests);
    Status VerifyRequest(const bytes& val, RequestInfo* info);
    Status VerifyConsenterSig(const Signature& signature, const Proposal& prop, bytes* aux);
    Status VerifySignature(const Signature& signature);
    uint64_t VerificationSequence();
    std::vector<RequestInfo> RequestsFromProposal(const Proposal& proposal);
    bytes AuxiliaryData(const bytes& msg);
    // ---- api.RequestInspector (pkg/api/dependencies.go:80-83) ------------------------------------
    RequestInfo RequestID(const bytes& req);

    // Batch form used by decision replay / sync (pkg/types/types.go:31-34): all signatures of many
    // decisions in one backend call; out[i] = 1 accept / 0 reject.
    Status VerifyConsenterSigBatch(const std::vector<Signature>& sigs, const std::vector<const Proposal*>& props,
                                   std::vector<uint8_t>* out);
    CoalescerStats stats() { return co_.stats(); }
    Scheme scheme() const { return opt_.scheme; }

 private:
    bool consenter_key(uint64_t id, uint8_t q[64], long* slot = nullptr);
    bool client_key(const std::string& id, uint8_t q[64], long* slot = nullptr);
    void make_tuple(const uint8_t q[64], const bytes& msg, const bytes& sig_der, uint8_t out[160]);
    static void make_tuple_ed25519(const uint8_t a_enc[32], const bytes& msg, const bytes& sig, uint8_t out[128]);
    bool ed() const { return opt_.scheme == Scheme::ED25519; }
    bool k256() const { return opt_.scheme == Scheme::SECP256K1; }
    size_t key_bytes() const { return ed() ? 32 : 64; }
    Status verify_one(const uint8_t q[64], const bytes& msg, const bytes& sig, long slot = -1);
    std::string cache_key(const uint8_t q[64], const bytes& msg, const bytes& sig) const;
    SpinLock mu_;
    std::map<uint64_t, bytes> consenters_;
    std::map<uint64_t, long> consenter_slot_;
    std::map<std::string, long> client_slot_;       // backend key slot per client, -1 without a registry
    std::map<std::string, bytes> clients_;
    uint64_t seq_ = 0;
    VerifierOptions opt_;
    Coalescer co_;
    SpinLock cache_mu_;
    std::unordered_map<std::string, bool> cache_;
    // Grow-only staging arrays of the raw-messages batch path, page-locked when the backend offers it: handing
    // pageable memory to a ~100 MB batch costs more in the runtime's pinning than the kernels take.
    struct Staging {
        void* p = nullptr; size_t cap = 0; bool pinned = false;
    };
    void* staging(Staging& s, size_t bytes);
    void staging_release(Staging& s);
    std::mutex staging_mu_;
    Staging st_msgs_, st_sigs_, st_moff_, st_soff_, st_slots_;
    // Proposal.Digest() once per Proposal object (formats.h: ProposalDigestSlot).  digest_of: the digest, computed here by the
    // first caller (concurrent first callers wait for it instead of hashing the same megabytes N - 1 times) or awaited from the
    // prefetch.  digest_prefetch: VerifyProposal hands ASN.1 + SHA-256 of the proposal to a worker thread, so that it runs
    // beside the backend call and the prepare round (a K = 10 000 proposal is 1.7 MB: ~1.2 ms of marshalling + SHA-256 that the
    // first commit vote used to pay); the worker reads the caller's object only until it holds the marshalled bytes, and
    // VerifyProposal does not return before that (the caller may drop the proposal afterwards).
    bytes digest_of(const Proposal& p);
    void digest_prefetch(const Proposal& p, std::shared_ptr<ProposalDigestSlot>* slot_out);
    void digest_worker();
    struct DigestJob { const Proposal* p; std::shared_ptr<ProposalDigestSlot> slot; };
    std::mutex dw_mu_;
    std::condition_variable dw_cv_;
    std::vector<DigestJob> dw_jobs_;
    std::thread dw_thread_;
    bool dw_stop_ = false;
};

// api.Signer for one node (pkg/api/dependencies.go:46-52)
class Signer {
 public:
    // private_key: the P-256 / secp256k1 scalar d, or the RFC 8032 seed under Scheme::ED25519
    Signer(uint64_t id, const uint8_t private_key[32], Scheme scheme = Scheme::P256);
    uint64_t id() const { return id_; }
    const uint8_t* public_key() const { return q_; }                        // 64 bytes Qx|Qy, or 32 bytes A_enc (+ 32 zero bytes)
    bytes Sign(const bytes& msg);                                           // DER over SHA-256(msg), or the 64-byte Ed25519 R|S
    Signature SignProposal(const Proposal& proposal, const bytes& auxiliary_input);

 private:
    uint64_t id_;
    Scheme scheme_;
    uint8_t d_[32], q_[64];
};

// f = (n-1)/3, q = ceil((n+f+1)/2)   internal/bft/util.go:183-187
void compute_quorum(uint64_t n, int* q, int* f);

}  // namespace sbvhost

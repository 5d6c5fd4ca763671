// formats.h — value types crossing the api.Verifier seam and the signed-data formats the
// application must fix (the reference defines none for signed data: SURVEY.md Appendix C).
//
//   Proposal / Signature / RequestInfo mirror pkg/types/types.go:18-29, 41-48.
//   proposal_digest() restates Proposal.Digest() (types.go:50-69): SHA-256 over Go's
//   encoding/asn1 DER of SEQUENCE{OCTET STRING Payload, OCTET STRING Header, OCTET STRING
//   Metadata, INTEGER VerificationSequence}, hex encoded.
#pragma once
#include <stdint.h>

#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

namespace sbvhost {

typedef std::string bytes;

// Proposal.Digest() of one Proposal OBJECT, computed at most once (Verifier::digest_of / digest_prefetch).  The reference
// recomputes it three times per sequence (internal/bft/view.go:435, 443, 524) and every VerifyConsenterSig must bind its
// message to it; it passes the SAME proposal value to VerifyProposal (view.go:555) and, a round trip later, to every
// VerifyConsenterSig of that sequence (view.go:834).  The memo therefore travels with the object — no table of proposals, no
// payload comparison, no copy kept — as a lazily computed field does.
//
// It cannot go stale: the four fields of a Proposal are PRIVATE, every mutator drops the object's slot, a copy
// shares the slot of its source (equal contents at that moment; a later mutation of either object drops only that object's
// slot), and a moved-from object loses its slot together with its contents.  A caller that reuses one Proposal variable across
// sequences therefore gets a fresh digest after every change, as the reference's per-call Digest() would give it.
struct ProposalDigestSlot {
    std::mutex mu;
    std::condition_variable cv;
    bool ready = false;       // digest is final (under mu, for the waiters)
    bool released = false;    // the worker no longer reads the caller's Proposal (it holds the marshalled bytes)
    std::atomic<bool> ready_flag{false};   // the same as `ready`, readable without the lock: the N-1 votes of a burst ask at the same instant
    bytes digest;             // immutable once ready
};

class Proposal {                  // pkg/types/types.go:18-23
 public:
    Proposal() = default;
    Proposal(bytes payload, bytes header, bytes metadata, int64_t verification_sequence = 0)
        : payload_(std::move(payload)), header_(std::move(header)), metadata_(std::move(metadata)), verification_sequence_(verification_sequence) {}
    Proposal(const Proposal& o) : payload_(o.payload_), header_(o.header_), metadata_(o.metadata_), verification_sequence_(o.verification_sequence_), slot_(o.digest_slot()) {}
    Proposal(Proposal&& o) noexcept : payload_(std::move(o.payload_)), header_(std::move(o.header_)), metadata_(std::move(o.metadata_)),
                                      verification_sequence_(o.verification_sequence_), slot_(o.take_slot()) {}
    Proposal& operator=(const Proposal& o) {
        if (this != &o) { payload_ = o.payload_; header_ = o.header_; metadata_ = o.metadata_; verification_sequence_ = o.verification_sequence_; set_slot(o.digest_slot()); }
        return *this;
    }
    Proposal& operator=(Proposal&& o) noexcept {
        if (this != &o) { payload_ = std::move(o.payload_); header_ = std::move(o.header_); metadata_ = std::move(o.metadata_); verification_sequence_ = o.verification_sequence_; set_slot(o.take_slot()); }
        return *this;
    }
    const bytes& payload() const { return payload_; }
    const bytes& header() const { return header_; }
    const bytes& metadata() const { return metadata_; }
    int64_t verification_sequence() const { return verification_sequence_; }
    // every mutator drops this object's digest slot: the next digest_of() hashes the new contents
    void set_payload(bytes v) { payload_ = std::move(v); set_slot(nullptr); }
    void set_header(bytes v) { header_ = std::move(v); set_slot(nullptr); }
    void set_metadata(bytes v) { metadata_ = std::move(v); set_slot(nullptr); }
    void set_verification_sequence(int64_t v) { verification_sequence_ = v; set_slot(nullptr); }
    // The digest slot of this object (see ProposalDigestSlot; never part of the value).  Guarded by a spin flag of its own: the
    // free-function atomics on shared_ptr take a pooled pthread mutex, and 15 votes asking at once queued on it for ~1 us each.
    std::shared_ptr<ProposalDigestSlot> digest_slot() const {
        lock();
        std::shared_ptr<ProposalDigestSlot> s = slot_;
        unlock();
        return s;
    }
    // installs `fresh` when no slot is there yet; returns the slot that is there afterwards, *installed = whether it is `fresh`
    std::shared_ptr<ProposalDigestSlot> digest_slot_install(const std::shared_ptr<ProposalDigestSlot>& fresh, bool* installed) const {
        lock();
        *installed = !slot_;
        if (!slot_) slot_ = fresh;
        std::shared_ptr<ProposalDigestSlot> s = slot_;
        unlock();
        return s;
    }

 private:
    void set_slot(std::shared_ptr<ProposalDigestSlot> s) const { lock(); slot_.swap(s); unlock(); }
    std::shared_ptr<ProposalDigestSlot> take_slot() const { std::shared_ptr<ProposalDigestSlot> s; lock(); slot_.swap(s); unlock(); return s; }
    void lock() const { while (slot_lock_.test_and_set(std::memory_order_acquire)) {} }
    void unlock() const { slot_lock_.clear(std::memory_order_release); }
    bytes payload_, header_, metadata_;
    int64_t verification_sequence_ = 0;
    mutable std::atomic_flag slot_lock_ = ATOMIC_FLAG_INIT;
    mutable std::shared_ptr<ProposalDigestSlot> slot_;
};
struct Signature {                // pkg/types/types.go:25-29
    uint64_t id = 0;
    bytes value;                  // ASN.1 DER ECDSA-Sig-Value
    bytes msg;                    // signed bytes; hash = SHA-256(msg)
};
struct RequestInfo {              // pkg/types/types.go:41-48
    std::string client_id, id;
    bool operator==(const RequestInfo& o) const { return client_id == o.client_id && id == o.id; }
};

bytes asn1_marshal_proposal(const Proposal& p);     // Go asn1.Marshal(Proposal{...})
bytes proposal_digest_raw(const Proposal& p);       // 32-byte SHA-256 of the above
std::string proposal_digest(const Proposal& p);     // hex, == Proposal.Digest()

// CommitSignaturesDigest (internal/bft/util.go:564-595): SHA-256 over Go's asn1.Marshal of the signature list
// (IntDoubleBytes); 32 raw bytes, or empty for an empty list (Go returns nil).  It is what a ViewData's
// last-decision signatures are summarised to in the view-change path.
bytes asn1_marshal_commit_signatures(const std::vector<Signature>& sigs);
bytes commit_signatures_digest(const std::vector<Signature>& sigs);

// ---- client request:  u16 len|ClientID  u16 len|ID  u32 len|payload  u16 len|sig  (big-endian lengths)
struct Request {
    std::string client_id, id;
    bytes payload;
    bytes sig;        // DER
    bytes signed_part;  // everything before the signature length field
};
bytes request_unsigned(const std::string& client_id, const std::string& id, const bytes& payload);
bytes request_encode(const bytes& unsigned_part, const bytes& sig_der);
bool request_parse(const bytes& raw, Request* out);

// The same parse without copies: offsets into the buffer the request lives in (a proposal's payload).  request_parse_view
// accepts exactly what request_parse accepts.
struct RequestView {
    size_t client_off = 0, client_len = 0, id_off = 0, id_len = 0;
    size_t signed_off = 0, signed_len = 0;      // everything before the signature length field
    size_t sig_off = 0, sig_len = 0;
};
bool request_parse_view(const bytes& buf, size_t off, size_t len, RequestView* out);

// ---- proposal payload:  u32 count  (u32 len | request)*
bytes payload_encode(const std::vector<bytes>& requests);
bool payload_split(const bytes& payload, std::vector<bytes>* out);
// (offset, length) of every request inside `payload`; accepts exactly what payload_split accepts
bool payload_split_views(const bytes& payload, std::vector<std::pair<size_t, size_t>>* out);

// ---- consenter signature message:  "SBV1" | SHA-256(asn1(proposal)) | u32 len | aux
bytes consenter_msg(const Proposal& p, const bytes& aux);
bool consenter_msg_split(const bytes& msg, bytes* binding32, bytes* aux);

}  // namespace sbvhost

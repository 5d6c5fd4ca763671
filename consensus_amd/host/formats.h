// formats.h — value types crossing the api.Verifier seam and the signed-data formats the
// application must fix (the reference defines none for signed data: SURVEY.md Appendix C).
//
//   Proposal / Signature / RequestInfo mirror pkg/types/types.go:18-29, 41-48.
//   proposal_digest() restates Proposal.Digest() (types.go:50-69): SHA-256 over Go's
//   encoding/asn1 DER of SEQUENCE{OCTET STRING Payload, OCTET STRING Header, OCTET STRING
//   Metadata, INTEGER VerificationSequence}, hex encoded.
#pragma once
#include <stdint.h>

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
// payload comparison, no copy kept — as a lazily computed field does.  Contract: a Proposal is immutable once a Verifier has
// seen it (as in the reference, where these are protobuf-decoded bytes nobody writes to); copies share the slot.
struct ProposalDigestSlot {
    std::mutex mu;
    std::condition_variable cv;
    bool ready = false;       // digest is final
    bool released = false;    // the worker no longer reads the caller's Proposal (it holds the marshalled bytes)
    bytes digest;
};

struct Proposal {                 // pkg/types/types.go:18-23
    bytes payload, header, metadata;
    int64_t verification_sequence = 0;
    mutable std::shared_ptr<ProposalDigestSlot> digest_slot;   // see ProposalDigestSlot; never part of the value
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

#include "formats.h"

#include "p256_host.h"

namespace sbvhost {

namespace {

void der_len(bytes& out, size_t n) {
    if (n < 128) { out.push_back((char)n); return; }
    uint8_t tmp[8]; int k = 0;
    while (n) { tmp[k++] = (uint8_t)n; n >>= 8; }
    out.push_back((char)(0x80 | k));
    while (k) out.push_back((char)tmp[--k]);
}
void der_octets(bytes& out, const bytes& v) { out.push_back(0x04); der_len(out, v.size()); out += v; }
void der_int64(bytes& out, int64_t v) {          // minimal two's complement, as encoding/asn1 does
    uint8_t b[8];
    for (int i = 0; i < 8; ++i) b[i] = (uint8_t)((uint64_t)v >> (56 - 8 * i));
    int off = 0;
    while (off < 7 && ((b[off] == 0x00 && !(b[off + 1] & 0x80)) || (b[off] == 0xff && (b[off + 1] & 0x80)))) ++off;
    out.push_back(0x02);
    out.push_back((char)(8 - off));
    out.append((const char*)b + off, 8 - off);
}
void put_u16(bytes& o, size_t v) { o.push_back((char)(v >> 8)); o.push_back((char)v); }
void put_u32(bytes& o, size_t v) { o.push_back((char)(v >> 24)); o.push_back((char)(v >> 16)); o.push_back((char)(v >> 8)); o.push_back((char)v); }
bool get_u16(const bytes& b, size_t& pos, size_t& v) {
    if (pos + 2 > b.size()) return false;
    v = ((size_t)(uint8_t)b[pos] << 8) | (uint8_t)b[pos + 1]; pos += 2; return true;
}
bool get_u32(const bytes& b, size_t& pos, size_t& v) {
    if (pos + 4 > b.size()) return false;
    v = ((size_t)(uint8_t)b[pos] << 24) | ((size_t)(uint8_t)b[pos + 1] << 16) | ((size_t)(uint8_t)b[pos + 2] << 8) | (uint8_t)b[pos + 3];
    pos += 4; return true;
}
bool get_bytes(const bytes& b, size_t& pos, size_t n, bytes& out) {
    if (n > b.size() || pos > b.size() - n) return false;
    out.assign(b, pos, n); pos += n; return true;
}
const char kHex[] = "0123456789abcdef";

}  // namespace

bytes asn1_marshal_proposal(const Proposal& p) {
    bytes body;
    der_octets(body, p.payload());
    der_octets(body, p.header());
    der_octets(body, p.metadata());
    der_int64(body, p.verification_sequence());
    bytes out;
    out.push_back(0x30);
    der_len(out, body.size());
    return out + body;
}
bytes proposal_digest_raw(const Proposal& p) { return sha256(asn1_marshal_proposal(p)); }
std::string proposal_digest(const Proposal& p) {
    const bytes d = proposal_digest_raw(p);
    std::string hex;
    for (unsigned char c : d) { hex.push_back(kHex[c >> 4]); hex.push_back(kHex[c & 15]); }
    return hex;
}

// CommitSignaturesDigest (internal/bft/util.go:564-595): asn1.Marshal(IntDoubleBytes{A: []IntDoubleByte{A int64; B, C []byte}})
// = SEQUENCE { SEQUENCE OF SEQUENCE { INTEGER signer, OCTET STRING value, OCTET STRING msg } }, then SHA-256.
// Go returns nil for an empty list; so does this (empty string).
bytes asn1_marshal_commit_signatures(const std::vector<Signature>& sigs) {
    bytes list;
    for (const Signature& s : sigs) {
        bytes item;
        der_int64(item, (int64_t)s.id);
        der_octets(item, s.value);
        der_octets(item, s.msg);
        list.push_back(0x30);
        der_len(list, item.size());
        list += item;
    }
    bytes inner;
    inner.push_back(0x30);
    der_len(inner, list.size());
    inner += list;
    bytes out;
    out.push_back(0x30);
    der_len(out, inner.size());
    return out + inner;
}
bytes commit_signatures_digest(const std::vector<Signature>& sigs) {
    if (sigs.empty()) return bytes();
    return sha256(asn1_marshal_commit_signatures(sigs));
}

bytes request_unsigned(const std::string& client_id, const std::string& id, const bytes& payload) {
    bytes o;
    put_u16(o, client_id.size()); o += client_id;
    put_u16(o, id.size()); o += id;
    put_u32(o, payload.size()); o += payload;
    return o;
}
bytes request_encode(const bytes& unsigned_part, const bytes& sig_der) {
    bytes o = unsigned_part;
    put_u16(o, sig_der.size());
    return o + sig_der;
}
bool request_parse(const bytes& raw, Request* out) {
    size_t pos = 0, n = 0;
    Request r;
    if (!get_u16(raw, pos, n) || !get_bytes(raw, pos, n, r.client_id)) return false;
    if (!get_u16(raw, pos, n) || !get_bytes(raw, pos, n, r.id)) return false;
    if (!get_u32(raw, pos, n) || !get_bytes(raw, pos, n, r.payload)) return false;
    r.signed_part.assign(raw, 0, pos);
    if (!get_u16(raw, pos, n) || !get_bytes(raw, pos, n, r.sig)) return false;
    if (pos != raw.size()) return false;
    *out = r;
    return true;
}

bool request_parse_view(const bytes& buf, size_t off, size_t len, RequestView* out) {
    if (len > buf.size() || off > buf.size() - len) return false;
    const uint8_t* b = (const uint8_t*)buf.data() + off;
    size_t pos = 0;
    auto u16 = [&](size_t& v) { if (pos + 2 > len) return false; v = ((size_t)b[pos] << 8) | b[pos + 1]; pos += 2; return true; };
    auto u32 = [&](size_t& v) {
        if (pos + 4 > len) return false;
        v = ((size_t)b[pos] << 24) | ((size_t)b[pos + 1] << 16) | ((size_t)b[pos + 2] << 8) | b[pos + 3];
        pos += 4; return true;
    };
    auto skip = [&](size_t n) { if (n > len - pos) return false; pos += n; return true; };
    RequestView v;
    size_t n = 0;
    if (!u16(n)) return false;
    v.client_off = off + pos; v.client_len = n;
    if (!skip(n) || !u16(n)) return false;
    v.id_off = off + pos; v.id_len = n;
    if (!skip(n) || !u32(n) || !skip(n)) return false;
    v.signed_off = off; v.signed_len = pos;
    if (!u16(n)) return false;
    v.sig_off = off + pos; v.sig_len = n;
    if (!skip(n) || pos != len) return false;
    *out = v;
    return true;
}

bytes payload_encode(const std::vector<bytes>& requests) {
    bytes o;
    put_u32(o, requests.size());
    for (const bytes& r : requests) { put_u32(o, r.size()); o += r; }
    return o;
}
bool payload_split(const bytes& payload, std::vector<bytes>* out) {
    size_t pos = 0, count = 0;
    if (!get_u32(payload, pos, count)) return false;
    if (count > payload.size()) return false;
    out->clear();
    out->reserve(count);
    for (size_t i = 0; i < count; ++i) {
        size_t n = 0; bytes r;
        if (!get_u32(payload, pos, n) || !get_bytes(payload, pos, n, r)) return false;
        out->push_back(r);
    }
    return pos == payload.size();
}

bool payload_split_views(const bytes& payload, std::vector<std::pair<size_t, size_t>>* out) {
    size_t pos = 0, count = 0;
    if (!get_u32(payload, pos, count)) return false;
    if (count > payload.size()) return false;
    out->clear();
    out->reserve(count);
    for (size_t i = 0; i < count; ++i) {
        size_t n = 0;
        if (!get_u32(payload, pos, n) || n > payload.size() || pos > payload.size() - n) return false;
        out->emplace_back(pos, n);
        pos += n;
    }
    return pos == payload.size();
}

bytes consenter_msg(const Proposal& p, const bytes& aux) {
    bytes o = "SBV1";
    o += proposal_digest_raw(p);
    put_u32(o, aux.size());
    return o + aux;
}
bool consenter_msg_split(const bytes& msg, bytes* binding32, bytes* aux) {
    if (msg.size() < 40 || msg.compare(0, 4, "SBV1") != 0) return false;
    size_t pos = 36, n = 0;
    if (!get_u32(msg, pos, n) || pos + n != msg.size()) return false;
    if (binding32) binding32->assign(msg, 4, 32);
    if (aux) aux->assign(msg, pos, n);
    return true;
}

}  // namespace sbvhost

package gpuverifier

import (
	"crypto/sha256"
	"encoding/asn1"
	"encoding/binary"
	"errors"

	bft "github.com/hyperledger-labs/SmartBFT/pkg/types"
)

// Request is a signed client request:
//
//	u16 len | ClientID | u16 len | ID | u32 len | Payload | u16 len | Sig
//
// (big-endian lengths).  Sig is the ASN.1 DER ECDSA signature, as ecdsa.SignASN1 emits it, over SHA-256 of everything
// before the signature's length field.  Same layout as consensus_amd/host/formats.cc.
type Request struct {
	ClientID string
	ID       string
	Payload  []byte
	Signed   []byte // the signed prefix of the raw request
	Sig      []byte
}

func appendField16(b, f []byte) []byte {
	var l [2]byte
	binary.BigEndian.PutUint16(l[:], uint16(len(f)))
	return append(append(b, l[:]...), f...)
}

// RequestUnsigned is the byte string a client signs.
func RequestUnsigned(clientID, id string, payload []byte) []byte {
	b := appendField16(nil, []byte(clientID))
	b = appendField16(b, []byte(id))
	var l [4]byte
	binary.BigEndian.PutUint32(l[:], uint32(len(payload)))
	return append(append(b, l[:]...), payload...)
}

// RequestEncode appends the signature to the signed part.
func RequestEncode(unsigned, sig []byte) []byte {
	return appendField16(append([]byte(nil), unsigned...), sig)
}

var errMalformed = errors.New("malformed request")

func take(b []byte, n int) ([]byte, []byte, bool) {
	if n < 0 || len(b) < n {
		return nil, nil, false
	}
	return b[:n], b[n:], true
}

// ParseRequest is strict: every length must fit and nothing may follow the signature.
func ParseRequest(raw []byte) (*Request, error) {
	rest := raw
	var f, l []byte
	var ok bool
	r := &Request{}
	if l, rest, ok = take(rest, 2); !ok {
		return nil, errMalformed
	}
	if f, rest, ok = take(rest, int(binary.BigEndian.Uint16(l))); !ok {
		return nil, errMalformed
	}
	r.ClientID = string(f)
	if l, rest, ok = take(rest, 2); !ok {
		return nil, errMalformed
	}
	if f, rest, ok = take(rest, int(binary.BigEndian.Uint16(l))); !ok {
		return nil, errMalformed
	}
	r.ID = string(f)
	if l, rest, ok = take(rest, 4); !ok {
		return nil, errMalformed
	}
	if f, rest, ok = take(rest, int(binary.BigEndian.Uint32(l))); !ok {
		return nil, errMalformed
	}
	r.Payload = f
	r.Signed = raw[:len(raw)-len(rest)]
	if l, rest, ok = take(rest, 2); !ok {
		return nil, errMalformed
	}
	if f, rest, ok = take(rest, int(binary.BigEndian.Uint16(l))); !ok || len(rest) != 0 {
		return nil, errMalformed
	}
	r.Sig = f
	return r, nil
}

// PayloadEncode packs the requests of a proposal: u32 count | (u32 len | request)*.
func PayloadEncode(reqs [][]byte) []byte {
	var l [4]byte
	binary.BigEndian.PutUint32(l[:], uint32(len(reqs)))
	b := append([]byte(nil), l[:]...)
	for _, r := range reqs {
		binary.BigEndian.PutUint32(l[:], uint32(len(r)))
		b = append(append(b, l[:]...), r...)
	}
	return b
}

// PayloadSplit is the inverse; any inconsistency is an error (a malformed proposal must be rejected).
func PayloadSplit(payload []byte) ([][]byte, error) {
	l, rest, ok := take(payload, 4)
	if !ok {
		return nil, errMalformed
	}
	n := int(binary.BigEndian.Uint32(l))
	if n > len(rest)/4 {
		return nil, errMalformed
	}
	out := make([][]byte, 0, n)
	for i := 0; i < n; i++ {
		var f []byte
		if l, rest, ok = take(rest, 4); !ok {
			return nil, errMalformed
		}
		if f, rest, ok = take(rest, int(binary.BigEndian.Uint32(l))); !ok {
			return nil, errMalformed
		}
		out = append(out, f)
	}
	if len(rest) != 0 {
		return nil, errMalformed
	}
	return out, nil
}

// proposalDigestRaw is SHA-256 over the same ASN.1 encoding Proposal.Digest() hashes (pkg/types/types.go:50-69);
// Digest() returns its hex form.
func proposalDigestRaw(p bft.Proposal) [32]byte {
	// the struct itself, so that the field order is the reference's (Payload, Header, Metadata, VerificationSequence)
	raw, err := asn1.Marshal(bft.Proposal{Payload: p.Payload, Header: p.Header, Metadata: p.Metadata, VerificationSequence: p.VerificationSequence})
	if err != nil {
		panic(err)
	}
	return sha256.Sum256(raw)
}

// ConsenterMsg is the Msg of a consenter signature: "SBV1" | digest(proposal) | u32 len | aux.  It binds the signature
// to the proposal and carries the auxiliary input verbatim, so that VerifyConsenterSig and AuxiliaryData can return it
// (internal/bft/view.go:481, 631-643, 1029, 1071).
func ConsenterMsg(digest [32]byte, aux []byte) []byte {
	b := append([]byte("SBV1"), digest[:]...)
	var l [4]byte
	binary.BigEndian.PutUint32(l[:], uint32(len(aux)))
	return append(append(b, l[:]...), aux...)
}

// ConsenterMsgSplit returns the bound digest and the auxiliary data.
func ConsenterMsgSplit(msg []byte) (digest [32]byte, aux []byte, ok bool) {
	if len(msg) < 40 || string(msg[:4]) != "SBV1" {
		return digest, nil, false
	}
	copy(digest[:], msg[4:36])
	n := int(binary.BigEndian.Uint32(msg[36:40]))
	if len(msg) != 40+n {
		return digest, nil, false
	}
	return digest, msg[40:], true
}

// p256_kernels.hip — gfx950 kernels for batch ECDSA P-256 verification outside the grouped step
// (the grouped step's kernels are in p256_group_kernels.hip).
//
//   k_p256_prep / _keyed  stage A.  One 64-lane workgroup (= one wavefront) walks T slabs of 64 tuples.  Each slab
//                         (64 x 160 B = 10 KiB, contiguous in HBM) is fetched with coalesced 16-byte loads
//                         (global_load_dwordx4, lane l reads bytes 16*l..) into LDS with a 41-dword row pitch (41 is odd ->
//                         the per-lane ds_read_b32 column walk is bank-conflict free), then every lane reads its own tuple's
//                         dwords back from LDS.  Montgomery's trick runs along the T slabs inside each lane: one
//                         division-step inversion (modinv30.h) per T signatures, all products on the carry-free scalar
//                         field (p256_sc29.h).
//   k_p256_verify         stage B, generic form: one lane per signature, per-signature affine window table in HBM,
//                         256 doublings on the carry-free field (p256_comb29.h: verify29_lane_generic).
//   k_p256_verify_keyed   registered keys: 13 + 33 comb additions per lane, no doublings.
//   k_p256_verify_keyed_coop   the same for batches <= 32768: 8 lanes per signature, butterfly of exact XYZZ additions.
//   k_p256_verify_prepared_small  batches <= 32 in ONE launch: stage A on the host (host_prep_small), records / verdicts in mapped host memory, 16 lanes per signature.
//   k_p256_sign           batch signing (RFC 6979), k_msg_frontend: SHA-256 + strict DER in front of the keyed kernels.
// The accept bits are gathered with a 64-wide ballot and written as bitmap bytes.
//
// No MFMA: this is 256-bit modular integer arithmetic (v_mad_i64_i32 into 64-bit columns, p256_fe29.h).
#include <hip/hip_runtime.h>
#include <cstring>
#include <stdlib.h>

#include <thread>
#include <vector>

#include "p256_core.h"
#include "p256_kernels.h"
#include "p256_widetab29.h"
#include "p256_sign.h"
#include "p256_comb29.h"
#include "sha256_dev.h"

namespace sbv {

constexpr int kPrepLanes = 64;


struct LdsTuple {
    const u32* row;
    __device__ __forceinline__ u32 operator[](int i) const { return row[i]; }
};

// TD = dwords per input tuple: 40 (r|s|hash|Qx|Qy) or 24 (r|s|hash, registered-key form).
template <int TD, bool HAS_Q>
__device__ __forceinline__ void prep_body(const uint8_t* __restrict__ tuples, size_t n, const Scratch& s, int T, unsigned block_off) {
    constexpr int kPitch = TD + 1;                 // odd -> conflict-free per-lane ds_read_b32 walk
    constexpr int kVec = TD / 4;                   // 16-byte elements per tuple
    __shared__ u32 lds[kPrepLanes * kPitch];
    const size_t block_first = ((size_t)blockIdx.x + block_off) * kPrepLanes * (size_t)T;
    const int lane = threadIdx.x;
    auto words = [&](int k, size_t) -> LdsTuple {
        const size_t slab = block_first + (size_t)k * kPrepLanes;      // first tuple of the slab
        __syncthreads();                                               // previous slab fully consumed
        const uint4* src = reinterpret_cast<const uint4*>(tuples + slab * (size_t)(TD * 4));
#pragma unroll
        for (int it = 0; it < kVec; ++it) {
            const int e = it * kPrepLanes + lane;                      // 16-byte element of the slab
            const int t = e / kVec, part = e - t * kVec;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (slab + (size_t)t < n) v = src[e];
            u32* dst = lds + t * kPitch + part * 4;
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
        }
        __syncthreads();
        return LdsTuple{lds + lane * kPitch};
    };
    prep_chunk29<HAS_Q>(words, n, s, block_first + (size_t)lane, (size_t)kPrepLanes, T);
}

// block_off: the launch covers workgroups [block_off, block_off + gridDim.x) of the batch
__global__ __launch_bounds__(kPrepLanes) void k_p256_prep(const uint8_t* __restrict__ tuples, size_t n,
                                                          Scratch s, int T, unsigned block_off) {
    prep_body<40, true>(tuples, n, s, T, block_off);
}
__global__ __launch_bounds__(kPrepLanes) void k_p256_prep_keyed(const uint8_t* __restrict__ rsh, size_t n,
                                                                Scratch s, int T) {
    prep_body<24, false>(rsh, n, s, T, 0u);
}

// accept bits of one wavefront -> 8 bitmap bytes (LSB-first)
__device__ __forceinline__ void finish_wave(bool accept, size_t i, size_t n, uint8_t* __restrict__ bitmap) {
    const unsigned long long m = __ballot(accept);
    const int lane = threadIdx.x & 63;
    const size_t wave_first = i - (size_t)lane;
    if (lane < 8) {
        const size_t byte = (wave_first >> 3) + (size_t)lane;
        if (byte < ((n + 7) >> 3)) bitmap[byte] = (uint8_t)(m >> (8 * lane));
    }
}

// Stage B, registered-key form: 13 + 33 mixed additions per lane from two combs (G and the key's), on the carry-free
// field (p256_comb29.h); both tables in the R = 2^261 domain.  A wavefront whose lanes all hold widened slots (the consenters':
// sbv_p256_widen_keys) takes the key's wide comb instead: 13 + 16 additions at 16 bits (550 000 signatures of 16 keys: 1.66 -> 1.25 ms,
// 1.14 ms at 20 bits; profiles/r04/ab_wide_r04k.jsonl).  A wide-only kernel at 3 waves per SIMD with this one as second pass over
// the wavefronts it left was measured in the same session and removed: 1.25 / 1.19 ms against 1.25 / 1.14 for the single kernel.
__global__ __launch_bounds__(SBV_VERIFY_BLOCK, 2) void k_p256_verify_keyed(Scratch s, size_t n, const u32* __restrict__ slots,
                                                                          u32 nkeys, const apt* __restrict__ ktab,
                                                                          const uint8_t* __restrict__ kvalid,
                                                                          gcomb g16r, widekeys wk, uint8_t* __restrict__ bitmap) {
    const size_t i = (size_t)blockIdx.x * SBV_VERIFY_BLOCK + threadIdx.x;
    // the tail lanes of the last wavefront replay tuple n - 1 (their verdicts are dropped): every lane of a wavefront takes part in
    // the wave-uniform choice between the wide and the 8-bit combs inside verify29_lane_keyed
    const size_t ii = i < n ? i : n - 1;
    const bool accept = verify29_lane_keyed(s, ii, slots[ii], nkeys, ktab, kvalid, g16r, wk) && i < n;
    finish_wave(accept, i, n, bitmap);
}

// Registered-key form, SBV_COOP_LANES lanes per signature (p256_core.h): the latency kernel for small batches.
// A wavefront holds 8 signatures = exactly one byte of the bitmap.
__global__ __launch_bounds__(SBV_VERIFY_BLOCK, 2) void k_p256_verify_keyed_coop(Scratch s, size_t n, const u32* __restrict__ slots,
                                                                            u32 nkeys, const apt* __restrict__ ktab,
                                                                            const uint8_t* __restrict__ kvalid,
                                                                            gcomb g16, widekeys wk, uint8_t* __restrict__ bitmap) {
    const size_t lane_g = (size_t)blockIdx.x * SBV_VERIFY_BLOCK + threadIdx.x;
    const size_t i = lane_g / SBV_COOP_LANES;
    const int sub = (int)(lane_g % SBV_COOP_LANES);
    const bool active = i < n;
    xyzz R;
    pt29_set_inf(R);
    bool ok = false;
    u32 slot = active ? slots[i] : 0u;
    const bool in_range = slot < nkeys;
    if (!in_range) slot = 0;
    const u32 widx = active ? widekeys_index(wk, slot) : 0u;
    const bool wide = wave_all(widx != SBV_WIDE_NONE);               // wave-uniform: all 8 signatures of the wavefront hold wide slots
    if (active) {
        ok = s.ok[i] != 0 && in_range && kvalid[slot] != 0;
        u256 u1, u2;
        soa_load(u1, s.u1, s.cap, i);
        soa_load(u2, s.u2, s.cap, i);
        const gcomb kw = {wk.tab + (size_t)(wide ? widx : 0u) * wk.stride, wk.bits, wk.windows};
        keyed29_partial_lane(R, u1, u2, ktab + (size_t)slot * (SBV_GTAB_WINDOWS * SBV_GTAB_PER_WINDOW), g16, sub, SBV_COOP_LANES, wide, kw);
    }
    // butterfly: after log2(lanes) exchanges every lane of the group holds the whole sum
    SBV_NOUNROLL
    for (int off = SBV_COOP_LANES / 2; off >= 1; off >>= 1) {
        xyzz P;
        SBV_UNROLL
        for (int l = 0; l < 9; ++l) {
            P.X.v[l] = __shfl_xor(R.X.v[l], off, 64);
            P.Y.v[l] = __shfl_xor(R.Y.v[l], off, 64);
            P.ZZ.v[l] = __shfl_xor(R.ZZ.v[l], off, 64);
            P.ZZZ.v[l] = __shfl_xor(R.ZZZ.v[l], off, 64);
        }
        P.inf = __shfl_xor(R.inf ? 1 : 0, off, 64) != 0;
        pt29_add(R, P);
    }
    bool accept = false;
    if (active && sub == 0) {
        u256 r;
        soa_load(r, s.r, s.cap, i);
        accept = ok && pt29_rx_matches(R, r);
    }
    const unsigned long long m = __ballot(accept);          // bits 0, 8, ..., 56
    if ((threadIdx.x & 63) == 0 && active) {                // lane 0 is sub 0 of the wavefront's first signature
        u32 byte = 0;
        SBV_UNROLL
        for (int g = 0; g < 8; ++g) byte |= (u32)((m >> (8 * g)) & 1ull) << g;
        bitmap[i >> 3] = (uint8_t)byte;
    }
}

// The latency form in ONE launch (n <= SBV_SMALL_MAX: a commit quorum, a handful of serial single verifications), stage A on
// the HOST.  A commit quorum is 15 signatures: their stage A — one modular inversion shared by Montgomery's trick and four
// multiplications mod N each — is ~7 us of one CPU core (host_prep_small below: the same prep_chunk29 the stage-A kernel runs,
// compiled for the host), against a 600-step division chain of ~40 us on ONE lane of a 2.4 GHz SIMD that 63 other lanes wait
// for.  The device keeps what is parallel: the 13 + 33 comb terms of u1 G + u2 Q, SBV_SMALL_LANES = 16 lanes per signature
// (3 terms per lane, then a 4-level butterfly of exact XYZZ additions: 7 dependent additions instead of the 9 of the 8-lane
// form).  The records come from the caller's page-locked buffer (mapped into the device's address space: no staging copy),
// every verdict goes back as a byte in mapped host memory followed by a system-scope counter the host polls — no copy back,
// no stream synchronisation on the way out.  Measured (profiles/r04/latency_small_r04d.jsonl, kernel_stats_latency_small_r04f.csv):
// a 15-signature call 136 -> 58 us, a lone signature 111 -> 54 us against round 3's form with stage A in lane 0 of each group
// (k_p256_verify_keyed_small, removed); the kernel itself 85-100 -> 38-43 us.
// in: n records of 24 words r | u1 | u2 (plain 256-bit integers, least significant word first), then at word
// SBV_SMALL_MAX * 24 the n key slots; a slot >= nkeys (the host writes 0xFFFFFFFF for a signature that failed the range checks)
// is a reject.
#define SBV_SMALL_LANES 16
__global__ __launch_bounds__(SBV_VERIFY_BLOCK, 2) void k_p256_verify_prepared_small(const u32* __restrict__ in, u32 n, u32 nkeys,
                                                                                 const apt* __restrict__ ktab, const uint8_t* __restrict__ kvalid,
                                                                                 gcomb g16, widekeys wk, uint8_t* __restrict__ out, u32* __restrict__ done) {
    constexpr int kSigs = SBV_VERIFY_BLOCK / SBV_SMALL_LANES;         // signatures per workgroup
    __shared__ u32 rec[kSigs * 24];
    __shared__ u32 slot_s[kSigs];
    __shared__ u32 verdict_s[kSigs];
    const u32 first = blockIdx.x * kSigs;
    const u32 here = n - first < (u32)kSigs ? n - first : (u32)kSigs;
    {   // the input lives in HOST memory: 16-byte loads, one per lane (whole PCIe bursts), into LDS
        const uint4* src = reinterpret_cast<const uint4*>(in + (size_t)first * 24);
        for (u32 e = threadIdx.x; e < here * 6; e += SBV_VERIFY_BLOCK) {
            const uint4 v = src[e];
            rec[4 * e] = v.x; rec[4 * e + 1] = v.y; rec[4 * e + 2] = v.z; rec[4 * e + 3] = v.w;
        }
        if (threadIdx.x < here) slot_s[threadIdx.x] = in[SBV_SMALL_MAX * 24 + first + threadIdx.x];
        if (threadIdx.x < kSigs) verdict_s[threadIdx.x] = 0;
    }
    __syncthreads();
    const u32 g = threadIdx.x / SBV_SMALL_LANES;
    const int sub = (int)(threadIdx.x % SBV_SMALL_LANES);
    const bool active = g < here;
    u256 r, u1, u2;
    SBV_UNROLL
    for (int l = 0; l < 8; ++l) { r.v[l] = 0; u1.v[l] = 0; u2.v[l] = 0; }
    xyzz R;
    pt29_set_inf(R);
    bool ok = false;
    u32 slot = active ? slot_s[g] : 0u;
    // a commit quorum is signed by consenters only: with their wide combs (p256_comb29.h: widekeys) the 13 + 16 terms are two per
    // lane instead of three — 6 dependent additions instead of 7.  Wave-uniform: the 4 signatures of the wavefront all hold wide slots.
    const bool in_range = slot < nkeys;           // a rejected record carries slot 0xFFFFFFFF
    if (!in_range) slot = 0;
    const u32 widx = active ? widekeys_index(wk, slot) : 0u;
    const bool wide = wave_all(widx != SBV_WIDE_NONE);
    if (active) {
        const u32* t = rec + g * 24;                                  // every lane of the group reads the same words: LDS broadcasts
        SBV_UNROLL
        for (int l = 0; l < 8; ++l) { r.v[l] = t[l]; u1.v[l] = t[8 + l]; u2.v[l] = t[16 + l]; }
        ok = in_range && kvalid[slot] != 0;
        const gcomb kw = {wk.tab + (size_t)(wide ? widx : 0u) * wk.stride, wk.bits, wk.windows};
        keyed29_partial_lane(R, u1, u2, ktab + (size_t)slot * (SBV_GTAB_WINDOWS * SBV_GTAB_PER_WINDOW), g16, sub, SBV_SMALL_LANES, wide, kw);
    }
    SBV_NOUNROLL
    for (int off = SBV_SMALL_LANES / 2; off >= 1; off >>= 1) {
        xyzz P;
        SBV_UNROLL
        for (int l = 0; l < 9; ++l) {
            P.X.v[l] = __shfl_xor(R.X.v[l], off, 64);
            P.Y.v[l] = __shfl_xor(R.Y.v[l], off, 64);
            P.ZZ.v[l] = __shfl_xor(R.ZZ.v[l], off, 64);
            P.ZZZ.v[l] = __shfl_xor(R.ZZZ.v[l], off, 64);
        }
        P.inf = __shfl_xor(R.inf ? 1 : 0, off, 64) != 0;
        pt29_add(R, P);
    }
    if (active && sub == 0) verdict_s[g] = ok && pt29_rx_matches(R, r) ? 1u : 0u;
    __syncthreads();
    if (threadIdx.x < kSigs / 4) {                                     // 4 lanes, one 32-bit store each: 4 verdict bytes
        const u32 w = verdict_s[4 * threadIdx.x] | (verdict_s[4 * threadIdx.x + 1] << 8) | (verdict_s[4 * threadIdx.x + 2] << 16) |
                      (verdict_s[4 * threadIdx.x + 3] << 24);
        reinterpret_cast<u32*>(out)[first / 4 + threadIdx.x] = w;       // first is a multiple of 16
        __threadfence_system();
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        __hip_atomic_fetch_add(done, here, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

hipError_t launch_p256_verify_prepared_small(const void* d_in, size_t n, u32 nkeys, const apt* d_ktab, const uint8_t* d_kvalid, const gcomb& d_gtab,
                                             const widekeys& wk, uint8_t* d_out, u32* d_done, hipStream_t stream) {
    if (n == 0 || n > SBV_SMALL_MAX) return hipErrorInvalidValue;
    const size_t lanes = n * SBV_SMALL_LANES;
    hipLaunchKernelGGL(k_p256_verify_prepared_small, dim3((unsigned)((lanes + SBV_VERIFY_BLOCK - 1) / SBV_VERIFY_BLOCK)), dim3(SBV_VERIFY_BLOCK), 0, stream,
                       static_cast<const u32*>(d_in), (u32)n, nkeys, d_ktab, d_kvalid, d_gtab, wk, d_out, d_done);
    return hipGetLastError();
}

// Host half of the latency form: stage A of n <= SBV_SMALL_MAX records r | s | hash (96 bytes each, big-endian, as the C-ABI
// takes them) -> rec: n x 24 words r | u1 | u2 and slot_out[i] = slots[i], or 0xFFFFFFFF when r or s is out of range.  It IS
// the stage-A lane of the kernels (p256_core.h: prep_chunk29<false>, one chunk of n tuples: ONE inversion for the whole
// quorum), compiled for the host; product code, not a fallback: the verdict still comes from the comb kernel.
namespace {
struct HostRecWords {
    const uint8_t* base;
    struct W {
        const uint8_t* p;
        u32 operator[](int i) const { u32 v; memcpy(&v, p + 4 * i, 4); return v; }
    };
    W operator()(int, size_t idx) const { return W{base + 96 * idx}; }
};
}  // namespace
void host_prep_small(const uint8_t* rsh, const u32* slots, size_t n, u32* rec, u32* slot_out) {
    constexpr size_t cap = SBV_SMALL_MAX;
    if (n > cap) n = cap;
    u32 pr[8 * cap], pu1[8 * cap], pu2[8 * cap], psm[8 * cap];
    uint8_t ok[cap];
    Scratch s{pr, pu1, pu2, nullptr, nullptr, psm, ok, cap};
    prep_chunk29<false>(HostRecWords{rsh}, n, s, 0, 1, (int)n);
    for (size_t i = 0; i < n; ++i) {
        u256 t;
        soa_load(t, s.r, cap, i);
        memcpy(rec + 24 * i, t.v, 32);
        soa_load(t, s.u1, cap, i);
        memcpy(rec + 24 * i + 8, t.v, 32);
        soa_load(t, s.u2, cap, i);
        memcpy(rec + 24 * i + 16, t.v, 32);
        slot_out[i] = ok[i] ? slots[i] : 0xFFFFFFFFu;
    }
}

// Generic form (public key in the tuple), carry-free field: per-signature AFFINE window table in HBM (p256_comb29.h:
// verify29_lane_generic), Jacobian accumulator with fused reductions.  qtab: SBV_QTAB29_WORDS words per lane.
__global__ __launch_bounds__(SBV_VERIFY_BLOCK, 2) void k_p256_verify(Scratch s, size_t n, u32* __restrict__ qtab, gcomb gc,
                                                                      uint8_t* __restrict__ bitmap) {
    const size_t i = (size_t)blockIdx.x * SBV_VERIFY_BLOCK + threadIdx.x;
    bool accept = false;
    if (i < n) accept = verify29_lane_generic(s, i, qtab + i * (size_t)SBV_QTAB29_WORDS, gc);
    finish_wave(accept, i, n, bitmap);
}

// Batch signing (p256_sign.h): lane i signs digest i with private key key_index[i] (or i % n_keys).
// keys / digests / sigs are byte strings as on the wire (big-endian 32-byte integers); ok[i] = 1 when a signature was produced.
__global__ __launch_bounds__(SBV_VERIFY_BLOCK, 2) void k_p256_sign(const u32* __restrict__ keys, u32 n_keys, const u32* __restrict__ key_index,
                                                                    const u32* __restrict__ digests, size_t n, gcomb gc,
                                                                    u32* __restrict__ sigs, uint8_t* __restrict__ ok) {
    const size_t i = (size_t)blockIdx.x * SBV_VERIFY_BLOCK + threadIdx.x;
    if (i >= n) return;
    u32 kidx = key_index ? key_index[i] : (u32)(i % n_keys);
    const bool known = kidx < n_keys;
    if (!known) kidx = 0;
    u32 d[8], h[8], rs[16];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        d[k] = __builtin_bswap32(keys[(size_t)kidx * 8 + k]);
        h[k] = __builtin_bswap32(digests[i * 8 + k]);
    }
    const bool good = sign29_lane(d, h, gc, rs) && known;
#pragma unroll
    for (int k = 0; k < 16; ++k) sigs[i * 16 + k] = good ? __builtin_bswap32(rs[k]) : 0u;
    ok[i] = good ? 1 : 0;
}

hipError_t launch_p256_sign(const uint8_t* d_keys, u32 n_keys, const u32* d_key_index, const uint8_t* d_digests, size_t n,
                            const gcomb& d_gcomb, uint8_t* d_sigs, uint8_t* d_ok, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_p256_sign, dim3((unsigned)((n + SBV_VERIFY_BLOCK - 1) / SBV_VERIFY_BLOCK)), dim3(SBV_VERIFY_BLOCK), 0, stream,
                       reinterpret_cast<const u32*>(d_keys), n_keys, d_key_index, reinterpret_cast<const u32*>(d_digests), n, d_gcomb,
                       reinterpret_cast<u32*>(d_sigs), d_ok);
    return hipGetLastError();
}

// Message front end: lane i hashes message i (SHA-256) and parses DER signature i,
// emitting the 96-byte r|s|hash record the registered-key stage A consumes.  Input is variable length,
// so lanes read their own byte ranges (offset tables); the 96-byte records are written back contiguous.
// mbase / sbase: the offset-table values of the first byte held in msgs / sigs (a PIECE of a larger batch carries the batch's own
// offsets: sbv_p256_verify_msgs_keyed_sharded uploads slices of the caller's tables as they are)
// mbytes / sbytes: bytes of msgs / sigs that were uploaded.  Every lane checks ITS OWN slice of the offset tables against them
// (monotone, inside the upload): an entry that is not turns into an empty message and an empty signature — a DER failure, r = s = 0,
// rejected — instead of a read outside the staging buffers.  (The host used to scan both tables with 8 threads before every call:
// 0.2-0.3 ms of thread start-up per call, profiles/r05; it now checks the piece boundaries only.)
__global__ __launch_bounds__(256) void k_msg_frontend(const uint8_t* __restrict__ msgs, const u64* __restrict__ moff,
                                                      const uint8_t* __restrict__ sigs, const u64* __restrict__ soff,
                                                      size_t n, u32* __restrict__ rsh, u64 mbase, u64 sbase, u64 mbytes, u64 sbytes) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    u64 m0 = moff[i] - mbase, m1 = moff[i + 1] - mbase, s0 = soff[i] - sbase, s1 = soff[i + 1] - sbase;
    if (!(m0 <= m1 && m1 <= mbytes && s0 <= s1 && s1 <= sbytes)) { m0 = m1 = 0; s0 = s1 = 0; }      // unsigned: an offset below the base wraps far above the bound
    u32 rec[24];
    msg_frontend_lane(msgs + m0, (size_t)(m1 - m0), sigs + s0, (size_t)(s1 - s0), rec);
#pragma unroll
    for (int k = 0; k < 24; ++k) rsh[i * 24 + k] = rec[k];
}

hipError_t launch_msg_frontend(const uint8_t* d_msgs, const u64* d_moff, const uint8_t* d_sigs, const u64* d_soff, size_t n,
                               u32* d_rsh, hipStream_t stream, u64 mbase, u64 sbase, u64 mbytes, u64 sbytes) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_msg_frontend, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_msgs, d_moff, d_sigs, d_soff, n, d_rsh, mbase, sbase, mbytes, sbytes);
    return hipGetLastError();
}

int prep_chunk_T(size_t n) {
    // Tuples per lane for Montgomery's trick.  The inversion (division steps, ~40 multiplications' worth) is
    // cheap enough that long chunks no longer pay: small batches take one inversion per tuple (shortest chain),
    // large ones amortise it over up to 8 and still put >= 2 wavefronts on every SIMD.
    static const int cap = [] { const char* e = getenv("SBV_PREP_T"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= 32 ? v : 8; }();
    size_t t = (n + 131071) / 131072;
    if (t < 1) t = 1;
    if (t > (size_t)cap) t = (size_t)cap;
    return (int)t;
}

hipError_t launch_p256_prep(const uint8_t* d_tuples, size_t n, const Scratch& s, hipStream_t stream, bool keyed) {
    if (n == 0) return hipSuccess;
    const int T = prep_chunk_T(n);
    const size_t per_block = (size_t)kPrepLanes * T;
    const unsigned grid = (unsigned)((n + per_block - 1) / per_block);
    if (keyed) hipLaunchKernelGGL(k_p256_prep_keyed, dim3(grid), dim3(kPrepLanes), 0, stream, d_tuples, n, s, T);
    else hipLaunchKernelGGL(k_p256_prep, dim3(grid), dim3(kPrepLanes), 0, stream, d_tuples, n, s, T, 0u);
    return hipGetLastError();
}
// workgroups [block_lo, block_hi) of stage A for the generic tuple form; prep_blocks() = how many there are, each
// covering prep_block_tuples() consecutive tuples
size_t prep_block_tuples(size_t n) { return (size_t)kPrepLanes * prep_chunk_T(n); }
hipError_t launch_p256_prep_blocks(const uint8_t* d_tuples, size_t n, const Scratch& s, hipStream_t stream, unsigned block_lo, unsigned block_hi) {
    if (n == 0 || block_hi <= block_lo) return hipSuccess;
    hipLaunchKernelGGL(k_p256_prep, dim3(block_hi - block_lo), dim3(kPrepLanes), 0, stream, d_tuples, n, s, prep_chunk_T(n), block_lo);
    return hipGetLastError();
}

// Batches up to this size take the lanes-per-signature kernel: 32768 x 8 lanes = 4 wavefronts per SIMD, still
// latency-bound (12 additions deep against 50 for the one-lane kernel at half a wavefront per SIMD).
// SBV_COOP_MAX=0 switches it off.
static size_t coop_max_batch() {
    static const size_t v = [] { const char* e = getenv("SBV_COOP_MAX"); return e ? (size_t)strtoull(e, nullptr, 10) : (size_t)32768; }();
    return v;
}

hipError_t launch_p256_verify_keyed(const Scratch& s, size_t n, const u32* d_slots, u32 nkeys, const apt* d_ktab,
                                    const uint8_t* d_kvalid, const gcomb& d_gtab, const widekeys& wk, uint8_t* d_bitmap, uint8_t* d_rerun,
                                    hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (n <= coop_max_batch()) {      // small batch: latency matters, lanes are plentiful
        const size_t lanes = n * SBV_COOP_LANES;
        hipLaunchKernelGGL(k_p256_verify_keyed_coop, dim3((unsigned)((lanes + SBV_VERIFY_BLOCK - 1) / SBV_VERIFY_BLOCK)),
                           dim3(SBV_VERIFY_BLOCK), 0, stream, s, n, d_slots, nkeys, d_ktab, d_kvalid, d_gtab, wk, d_bitmap);
        return hipGetLastError();
    }
    const unsigned grid = (unsigned)((n + SBV_VERIFY_BLOCK - 1) / SBV_VERIFY_BLOCK);
    (void)d_rerun;
    hipLaunchKernelGGL(k_p256_verify_keyed, dim3(grid), dim3(SBV_VERIFY_BLOCK), 0, stream, s, n, d_slots, nkeys, d_ktab, d_kvalid, d_gtab, wk, d_bitmap);
    return hipGetLastError();
}

hipError_t launch_p256_verify(const Scratch& s, size_t n, u32* d_qtab, const gcomb& d_gcomb, uint8_t* d_bitmap, uint8_t* d_rerun,
                              hipStream_t stream) {
    if (n == 0) return hipSuccess;
    (void)d_rerun;
    const unsigned grid = (unsigned)((n + SBV_VERIFY_BLOCK - 1) / SBV_VERIFY_BLOCK);
    hipLaunchKernelGGL(k_p256_verify, dim3(grid), dim3(SBV_VERIFY_BLOCK), 0, stream, s, n, d_qtab, d_gcomb, d_bitmap);
    return hipGetLastError();
}

void host_build_gtable(apt* out) { build_gtable(out); }

void host_build_gcomb(int bits, apt* out) {
    const int windows = (257 + bits - 1) / bits;
    std::vector<std::thread> th;
    for (int j = 0; j < windows; ++j) th.emplace_back([=] { build_gcomb_window(bits, j, out + ((size_t)j << (bits - 1))); });
    for (auto& t : th) t.join();
}

void host_build_g16(apt* out) {
    std::vector<std::thread> th;
    for (int j = 0; j < SBV_G16_WINDOWS; ++j) th.emplace_back([j, out] { build_g16_window(j, out + (size_t)j * SBV_G16_PER_WINDOW); });
    for (auto& t : th) t.join();
}

void host_convert_table_r261(const apt* in, apt* out, size_t count) {
    size_t nt = std::thread::hardware_concurrency();
    if (nt == 0) nt = 1;
    if (nt > 32) nt = 32;
    std::vector<std::thread> th;
    for (size_t t = 0; t < nt; ++t)
        th.emplace_back([=] {
            for (size_t k = count * t / nt; k < count * (t + 1) / nt; ++k) apt_to_r261(out[k], in[k]);
        });
    for (auto& x : th) x.join();
}

// comb of a registered key for the carry-free kernels (R = 2^261 domain)
bool host_build_key_table(const uint8_t q[64], apt* out) {
    u256 x, y;
    from_be32(x, q);
    from_be32(y, q + 32);
    if (!key_is_valid(x, y)) return false;
    build_comb_table(x, y, out);
    for (size_t k = 0; k < (size_t)SBV_KEYTAB_ENTRIES; ++k) { apt t; apt_to_r261(t, out[k]); out[k] = t; }
    return true;
}


// ---- wide combs built on the device (p256_widetab29.h) -----------------------------------------------------------------------
// bases: per key 2 * windows affine points (B_0 .. B_{W-1}, then C_0 .. C_{W-1}), canonical R = 2^261 words; tab: the wide comb
// pool; widx[k]: index of key k's comb in it.
__global__ __launch_bounds__(64) void k_widetab_chains(const apt* __restrict__ bases, const u32* __restrict__ widx, u32 nkeys, widebuild w, size_t stride,
                                                      u32* __restrict__ tmp, apt* __restrict__ tab) {
    const u32 lane = blockIdx.x * 64 + threadIdx.x;
    const u32 per_key = (u32)w.windows * 2u;
    if (lane >= nkeys * per_key) return;
    const u32 k = lane / per_key, j = (lane % per_key) >> 1, role = lane & 1u;
    const apt* kb = bases + (size_t)k * per_key;
    widetab_chain_role(w, kb + j, kb + w.windows + j, (int)role, tmp + (size_t)lane * widebuild_chain_len(w) * SBV_WIDETAB_REC_WORDS,
                       tab + (size_t)widx[k] * stride + (size_t)j * w.per_window);
}
__global__ __launch_bounds__(256) void k_widetab_fill(const u32* __restrict__ widx, u32 nkeys, widebuild w, size_t stride, apt* __restrict__ tab) {
    const size_t lane = (size_t)blockIdx.x * 256 + threadIdx.x;
    const u32 chunks = widebuild_fill_chunks(w);
    const size_t per_window = (size_t)(w.giants - 1) * chunks;
    const size_t per_key = per_window * (size_t)w.windows;
    if (lane >= per_key * nkeys) return;
    const u32 k = (u32)(lane / per_key);
    const size_t r = lane % per_key;
    const u32 j = (u32)(r / per_window);
    const size_t q = r % per_window;
    const u32 g = 1u + (u32)(q / chunks), c = (u32)(q % chunks);
    widetab_fill_lane(w, g, 1u + c * SBV_WIDETAB_T, tab + (size_t)widx[k] * stride + (size_t)j * w.per_window);
}
// d_bases / d_widx: device copies for `nkeys` keys; d_tmp: nkeys * windows * 2 * widebuild_chain_len * SBV_WIDETAB_REC_WORDS words
hipError_t launch_widetab_build(const apt* d_bases, const u32* d_widx, u32 nkeys, int bits, u32* d_tmp, apt* d_tab, hipStream_t stream) {
    if (nkeys == 0) return hipSuccess;
    const widebuild w = widebuild_make(bits);
    const size_t stride = gcomb_entries(bits);
    const u32 chain_lanes = nkeys * (u32)w.windows * 2u;
    hipLaunchKernelGGL(k_widetab_chains, dim3((chain_lanes + 63) / 64), dim3(64), 0, stream, d_bases, d_widx, nkeys, w, stride, d_tmp, d_tab);
    if (w.giants > 1) {
        const size_t fill_lanes = (size_t)nkeys * w.windows * (w.giants - 1) * widebuild_fill_chunks(w);
        hipLaunchKernelGGL(k_widetab_fill, dim3((unsigned)((fill_lanes + 255) / 256)), dim3(256), 0, stream, d_widx, nkeys, w, stride, d_tab);
    }
    return hipGetLastError();
}
size_t widetab_tmp_words(u32 nkeys, int bits) {
    const widebuild w = widebuild_make(bits);
    return (size_t)nkeys * w.windows * 2 * widebuild_chain_len(w) * SBV_WIDETAB_REC_WORDS;
}
// the host half: the 2 * windows base points of a key (B_j, then C_j), canonical R = 2^261 words; false = not a point of the curve
bool host_wide_bases(const uint8_t q[64], int bits, apt* out) {
    u256 x, y;
    from_be32(x, q);
    from_be32(y, q + 32);
    if (!key_is_valid(x, y)) return false;
    const widebuild w = widebuild_make(bits);
    comb_bases_bc(x, y, bits, w.hb, w.windows, out, out + w.windows);
    for (int k = 0; k < 2 * w.windows; ++k) { apt c; apt_to_r261(c, out[k]); out[k] = c; }
    return true;
}

// `bits`-wide comb of a registered key (p256_comb29.h: widekeys) for the carry-free kernels: gcomb_entries(bits) entries, window j
// at out + (j << (bits-1)).  One host thread per window (`threads` > 0 caps them); false = not a point of the curve.
bool host_build_wide_key_table(const uint8_t q[64], int bits, apt* out, int threads) {
    u256 x, y;
    from_be32(x, q);
    from_be32(y, q + 32);
    if (!key_is_valid(x, y)) return false;
    const int windows = (257 + bits - 1) / bits;
    const size_t per = (size_t)1 << (bits - 1);
    int nt = threads > 0 && threads < windows ? threads : windows;
    std::vector<std::thread> th;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([=] {
            for (int j = t; j < windows; j += nt) {
                apt* row = out + (size_t)j * per;
                build_comb_window_of(x, y, bits, j, row);
                for (size_t k = 0; k < per; ++k) { apt c; apt_to_r261(c, row[k]); row[k] = c; }
            }
        });
    for (auto& t : th) t.join();
    return true;
}

}  // namespace sbv

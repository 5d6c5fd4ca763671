// p256_kernels.h — launch interface between the C-ABI layer (sbv_api.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "p256_core.h"
#include "p256_comb29.h"
#include "p256_group.h"
#include "p256_widetab29.h"

#define SBV_TUPLE_BYTES 160
#define SBV_VERIFY_BLOCK 256

namespace sbv {

int prep_chunk_T(size_t n);
hipError_t launch_p256_prep(const uint8_t* d_tuples, size_t n, const Scratch& s, hipStream_t stream, bool keyed = false);
size_t prep_block_tuples(size_t n);
hipError_t launch_p256_prep_blocks(const uint8_t* d_tuples, size_t n, const Scratch& s, hipStream_t stream, unsigned block_lo, unsigned block_hi);
hipError_t launch_p256_verify_keyed(const Scratch& s, size_t n, const u32* d_slots, u32 nkeys, const apt* d_ktab,
                                    const uint8_t* d_kvalid, const gcomb& d_gcomb, const widekeys& wk, uint8_t* d_bitmap, uint8_t* d_rerun,
                                    hipStream_t stream);
// latency form of small registered-key batches in one launch (p256_kernels.hip: host_prep_small + k_p256_verify_prepared_small).
// d_in: the device view of a page-locked buffer holding n records r | u1 | u2 (24 words each, written by host_prep_small) and,
// at byte SBV_SMALL_MAX * 96, n u32 key slots; d_out: one verdict byte per signature; d_done: system-scope counter, += the
// workgroup's signatures when their verdicts are visible.  Up to two workgroups of 16 signatures.
#define SBV_SMALL_MAX 32
hipError_t launch_p256_verify_prepared_small(const void* d_in, size_t n, u32 nkeys, const apt* d_ktab, const uint8_t* d_kvalid, const gcomb& d_gcomb,
                                             const widekeys& wk, uint8_t* d_out, u32* d_done, hipStream_t stream);
void host_prep_small(const uint8_t* rsh, const u32* slots, size_t n, u32* rec, u32* slot_out);
void host_build_gcomb(int bits, apt* out);   // `bits`-wide comb of G, 8 x 32 Montgomery domain: gcomb_entries(bits) entries
// mbase / sbase: value of the offset tables at the first byte of d_msgs / d_sigs (0 for a whole batch; a piece keeps the batch's offsets)
// mbytes / sbytes: bytes uploaded behind d_msgs / d_sigs; a lane whose offsets are not monotone or leave them gets an empty message and signature (reject)
hipError_t launch_msg_frontend(const uint8_t* d_msgs, const u64* d_moff, const uint8_t* d_sigs, const u64* d_soff, size_t n,
                               u32* d_rsh, hipStream_t stream, u64 mbase, u64 sbase, u64 mbytes, u64 sbytes);
// comb table (33 x 128 affine multiples, R = 2^261 domain) of a registered key; false if the key is not a valid curve point
bool host_build_key_table(const uint8_t q[64], apt* out);
// `bits`-wide comb of a registered key (p256_comb29.h: widekeys), gcomb_entries(bits) entries, R = 2^261 domain
bool host_build_wide_key_table(const uint8_t q[64], int bits, apt* out, int threads);
// the same comb built on the device (p256_widetab29.h): host_wide_bases gives a key's 2 * windows base points, launch_widetab_build
// fills the combs of `nkeys` keys (d_widx[k] = index of key k's comb in d_tab); scratch words: widetab_tmp_words(nkeys, bits)
bool host_wide_bases(const uint8_t q[64], int bits, apt* out);
size_t widetab_tmp_words(u32 nkeys, int bits);
hipError_t launch_widetab_build(const apt* d_bases, const u32* d_widx, u32 nkeys, int bits, u32* d_tmp, apt* d_tab, hipStream_t stream);
#define SBV_KEYTAB_ENTRIES (SBV_GTAB_WINDOWS * SBV_GTAB_PER_WINDOW)
// d_rerun: ceil(n/64) bytes of per-wavefront flags (fast pass -> exact pass)
hipError_t launch_p256_verify(const Scratch& s, size_t n, u32* d_qtab, const gcomb& d_gcomb, uint8_t* d_bitmap, uint8_t* d_rerun,
                              hipStream_t stream);
// batch signing (p256_sign.h): keys n_keys x 32 B, key_index n x u32 or nullptr (i % n_keys), digests n x 32 B -> sigs n x 64 B (r | s), ok n B
hipError_t launch_p256_sign(const uint8_t* d_keys, u32 n_keys, const u32* d_key_index, const uint8_t* d_digests, size_t n,
                            const gcomb& d_gcomb, uint8_t* d_sigs, uint8_t* d_ok, hipStream_t stream);
// device buffers of the in-step key grouping (p256_group.h); owned by the context
struct GroupBuffers {
    u32* ht = nullptr; u32 ht_mask = 0;
    u32 *rep = nullptr, *cnt = nullptr, *slot_of = nullptr, *group_rep = nullptr, *counters = nullptr, *grp_idx = nullptr,
        *ung_idx = nullptr, *slots = nullptr;
    u32* jbases = nullptr;      // (Ed25519 grouped step) [max_groups][33] window bases
    u32* bases = nullptr;       // [max_groups][33][2] chain records of 36 words: B_j = 2^(8j) Q and 16 B_j, modified Jacobian (p256_keytab29.h)
    u32* jstate = nullptr;      // [max_groups][27] the doubling chain between chunks of windows
    apt* ktab = nullptr; uint8_t* kvalid = nullptr; u32* tmp = nullptr; uint8_t* acc = nullptr;
    apt* ntab = nullptr;        // [kc.cap + max_groups][33 x 16] compact rows (babies + giants) of every table slot: what the rows-only pass reads (p256_keytab29.h)
    KeyCache kc = {};           // persistent key-table cache: slots [0, kc.cap) of ktab / kvalid; [kc.cap, kc.cap + max_groups) = per batch
    u32* tslot = nullptr;       // [max_groups] table slot of each group of the current batch
    uint8_t* cold = nullptr;    // [max_groups] 1 = the group's tables are built in this batch
    u32* gacc = nullptr;        // [36][scratch cap] u1*G per tuple (XYZZ, 9-limb coordinates), then the running sum of the Q phase
    u32* gcount = nullptr;      // [2][max_groups] key-sorted list: exact group sizes, then the scatter cursors (p256_group.h)
    u32* grp_of = nullptr;      // [cap] group of lane L of the key-sorted list
    u32* ung_cand = nullptr;    // [cap] key-sorted step: ungrouped candidates before the key check
    u32* rec = nullptr;         // [scratch cap][SBV_REC_WORDS] stage A's per-tuple records (Scratch::rec) for the key-sorted list
    u32 max_groups = 0, min_count = 0;
    u32 seed = 0;               // key of the grouping hash table (GroupState::seed): random per context
    int sample_shift = -1;      // >= 0: count every 2^shift-th tuple against min_count (the P-256 default); -1: group_set_threshold's rule (explicit thresholds, the variants)
    // table classes (p256_group.h, round 5): [max_groups] this batch's groups — one addition per window allowed / run the fill now;
    // [kc.cap + max_groups] what a table slot holds across batches; the per-key signature count from which a full table pays
    uint8_t *full = nullptr, *needfill = nullptr, *kfull = nullptr;
    u32 full_min = 256;
    // hot keys (p256_group.h, round 5): wide combs of promoted cache slots and their bookkeeping; wtab == nullptr = off
    apt* wtab = nullptr;          // [wide_cap][gcomb_entries(SBV_HOT_BITS)]
    u32 *kwide = nullptr, *khits = nullptr;     // [kc.cap]
    u32* hot = nullptr;           // [4] wide combs handed out | promotions of this batch | lanes of the wide pass | spare
    u32* plist = nullptr;         // [2 x SBV_PROMOTE_MAX] (slot, wide index)
    u32* wowner = nullptr;        // [wide_cap] the cache slot that owns comb w (round 6: evictions), SBV_WIDE_NONE before its first owner
    u32* elist = nullptr;         // [SBV_PROMOTE_MAX] slots that found the pool full in this batch: the eviction candidates
    u32 hot_tick = 0;             // grouped batches with hot keys on so far: every SBV_HOT_DECAY_EVERY-th halves the hit counters
    apt* pbases = nullptr;        // [SBV_PROMOTE_MAX][2 x 17] base points of the promotions under construction
    u32* ptmp = nullptr;          // the builder's chain scratch for SBV_PROMOTE_MAX keys
    uint8_t* wide = nullptr;      // [max_groups] this batch's groups that may take the wide pass
    u32 wide_cap = 0, promote_min = 4096;
    size_t cap = 0;
    size_t gacc_cap = 0;        // the scratch capacity gacc was sized for
};
// A scheme's comb pool with its persistent key-table cache (secp256k1; the Ed25519 step keeps the same three things in its
// EdGroupBuffers, the P-256 step in GroupBuffers): table slots [0, kc.cap) are kept between batches, [kc.cap, kc.cap + max_groups)
// are rebuilt per batch.  One pool per scheme: a slot is found by the key BYTES, and the same bytes can be a point of two curves.
struct KeyPool {
    void* ktab = nullptr;       // (kc.cap + max_groups) x SBV_KEYTAB_ENTRIES 64-byte affine entries
    uint8_t* kvalid = nullptr;  // (kc.cap + max_groups)
    KeyCache kc = {};
    u32 max_groups = 0;
};
// streams and events of the grouped step; owned by the context.  `chunks` (1..SBV_GROUP_MAX_CHUNKS) = how
// many pieces the 33 key-comb windows are built and consumed in.
#define SBV_GROUP_MAX_CHUNKS 4
struct GroupSync {
    hipStream_t side_a = nullptr;   // insert, assign, key-cache lookup, the doubling chains
    hipStream_t side_b = nullptr;   // split / sort, rows + fill of the even chunks
    hipStream_t side_t = nullptr;   // P-256, secp256k1: not owned — rows + fill of the odd chunks when tstreams = 2 (the context's own stream while the caller's runs the step)
    int tstreams = 2;               // SBV_GROUP_TSTREAMS (1, 2): 1 = every chunk's rows + fill queue up on side_b.  2: rows of chunk 1 start when ITS chain ends, not when fill of chunk 0 does — cold 2^18 2.04 -> 1.70 ms, 2^17 2.54 -> 2.23, 2^20 unchanged (profiles/r03/ab_sched_r03m.jsonl).  The Ed25519 step keeps one table stream (measured in round 4: 4.55 -> 4.67 ms with two, profiles/r04/ab_ed_tstreams_r04a.jsonl)
    hipEvent_t ev_fork = nullptr, ev_assign = nullptr, ev_split = nullptr, ev_generic = nullptr;
    hipEvent_t ev_cache = nullptr, ev_class = nullptr;       // P-256: table slots assigned (side_a) / table classes decided (side_b)
    hipEvent_t ev_narrow = nullptr;                          // P-256: the rows-only pass (side_a) is done
    hipEvent_t ev_promote = nullptr;                         // P-256: this batch's promotions are selected (stream): side_b builds them
    hipEvent_t ev_wide = nullptr;                            // P-256: the wide pass (side_a) is done
    hipEvent_t ev_promoted = nullptr;                        // P-256: ... and published (side_b): the next batch's side_a waits for it
    hipEvent_t ev_bases[SBV_GROUP_MAX_CHUNKS] = {}, ev_tables[SBV_GROUP_MAX_CHUNKS] = {};
    int chunks = 1;
    int sorted = 1;                 // key-sorted grouped list + XCD-aware Q phase (SBV_GROUP_SORT=0: the split's compaction order; the form the step falls back to when a batch has more groups than one LDS histogram holds)
    size_t coop_max = (size_t)1 << 15;     // P-256: batches up to this size finish in ONE launch of 8 lanes per grouped tuple (k_group_coop; SBV_GROUP_COOP_MAX, 0 = off).  Measured in round 4 (profiles/r04/ab_coop_r04a.jsonl): warm 2^10 0.30 -> 0.19 ms, 2^12 0.34 -> 0.22, 2^14 0.37 -> 0.27, 2^15 0.39 -> 0.33
    size_t gsplit_min = (size_t)1 << 19;   // P-256: batches from this size run the G phase as its own 3-waves-per-SIMD kernel (SBV_GPHASE_SPLIT_MIN; 0 = never)
};
// Enqueues stage A AND stage B of a grouped batch.  ev_fork must have been recorded on `stream` first.  after_prep
// (optional) is recorded on `stream` once stage A is ordered before it.  prof (optional): 2 * chunks
// events, a pair around every Q-phase launch; *prof_pairs = the number of pairs used.
// d_g16: 16-bit comb of G in the 8 x 32 Montgomery domain (generic kernel); d_g16r: the same points for the carry-free field
hipError_t launch_p256_verify_grouped(const uint8_t* d_tuples, const Scratch& s, size_t n, const GroupBuffers& b, u32* d_qtab,
                                      const apt* d_g16, const gcomb& d_g16r, uint8_t* d_bitmap, hipStream_t stream, const GroupSync& y,
                                      hipEvent_t after_prep = nullptr, hipEvent_t* prof = nullptr, int* prof_pairs = nullptr);
void host_convert_table_r261(const apt* in, apt* out, size_t count);   // 8 x 32 Montgomery entries -> R = 2^261 domain (host threads)
void host_build_gtable(apt* out);   // 33 x 128 affine multiples of G (8-bit comb; host signer, key tables)
void host_build_g16(apt* out);      // 17 x 32768 affine multiples of G (16-bit comb used by the verify kernels)
#define SBV_G16_ENTRIES ((size_t)SBV_G16_WINDOWS * SBV_G16_PER_WINDOW)

// secp256k1 variant (k256_kernels.hip): stage A + stage B for n generic tuples; d_qtab = the per-lane strips of the P-256 generic
// kernel (SBV_QTAB29_WORDS words per lane); d_gtab = the 17 x 32768-entry comb of G built by host_build_k256_gtable
struct kapt;
hipError_t launch_k256_verify(const uint8_t* d_tuples, size_t n, const Scratch& s, u32* d_qtab, const kapt* d_gtab, uint8_t* d_bitmap,
                              hipStream_t stream);
void host_build_k256_gtable(kapt* out);
// grouped step on this curve (k256_group_kernels.hip): stage A + stage B; ev_fork recorded on `stream` by the caller
// d_gtab: the 16-bit comb (the generic lanes of the ungrouped list), d_gcomb: the `gcomb_bits`-wide comb of the G phase
hipError_t launch_k256_verify_grouped(const uint8_t* d_tuples, const Scratch& s, size_t n, const GroupBuffers& b, const KeyPool& kp, u32* d_qtab,
                                      const kapt* d_gtab, const kapt* d_gcomb, int gcomb_bits, uint8_t* d_bitmap, hipStream_t stream, const GroupSync& y,
                                      hipEvent_t* prof = nullptr, int* prof_pairs = nullptr);      // prof: 4 events, a pair around each of the two k_k256_qphase launches
void host_build_k256_gcomb(int bits, kapt* out);     // ceil(257 / bits) << (bits - 1) entries
#define SBV_K256_GTABLE_ENTRIES ((size_t)17 * 32768)

}  // namespace sbv

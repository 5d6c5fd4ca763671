#pragma once
#include <hip/hip_runtime.h>

#include "ed25519_core.h"
#include "p256_kernels.h"

namespace sbv {
hipError_t launch_ed25519_verify(const uint8_t* d_tuples, size_t n, u32* d_qtab, const aniels* d_btab, uint8_t* d_bitmap,
                                 hipStream_t stream);
// device buffers of the Ed25519 grouped step that the P-256 GroupBuffers do not already provide
struct EdGroupBuffers {
    aniels* ktab = nullptr;       // [kc.cap + max_groups][32 x 128] combs of -A: slots [0, kc.cap) = this scheme's persistent key-table
                                  // cache (round 4), [kc.cap, kc.cap + max_groups) = per batch
    uint8_t* okb = nullptr;       // [cap] S < L && k < L
    u32* ungxy = nullptr;         // [cap][20] X | Y of the ungrouped list's keys, left by the key check for the quad form of the one-lane kernel
    uint8_t* kvalid = nullptr;    // [kc.cap + max_groups] 1 = the slot's key decompressed.  NOT GroupBuffers::kvalid: bytes [0, kc.cap) of
                                  // that array belong to the P-256 key-table cache and outlive the batch — an Ed25519 batch writing
                                  // its own group verdicts there invalidated cached P-256 keys (found in round 3 by test order)
    KeyCache kc = {};             // keyed by the 32 key bytes (padded to the cache's 16 words)
    size_t cap = 0;
    u32 max_groups = 0;
    // hot keys of this scheme (ed25519_group.h, round 6): 16-bit combs of -A for promoted cache slots and the bookkeeping the P-256
    // step keeps in GroupBuffers (same meaning, same kernels: group_kernels_common.h); wtab == nullptr = off
    uint8_t* wtab = nullptr;      // [wide_cap] combs of SBV_ED_HOT_COMB_BYTES (16 x 32 768 entries at a 128-byte pitch)
    u32 *kwide = nullptr, *khits = nullptr;     // [kc.cap]
    u32* hot = nullptr;           // [4] combs handed out | promotions of this batch | lanes of the wide pass | eviction candidates
    u32* plist = nullptr;         // [2 x SBV_PROMOTE_MAX] (slot, comb)
    u32* wowner = nullptr;        // [wide_cap]
    u32* elist = nullptr;         // [SBV_PROMOTE_MAX]
    u32* ptmp = nullptr;          // the builder's scratch: SBV_ED_HOT_TMP_WORDS per resident lane
    uint8_t* wide = nullptr;      // [max_groups] this batch's groups whose slot owns a comb
    u32 wide_cap = 0, promote_min = 4096, hot_tick = 0;
};
// Grouped step (ed25519_group.h).  `b` supplies the grouping arrays, jbases, tmp, acc and gacc (32 words per
// tuple, stride b.gacc_cap); ev_fork of `y` must have been recorded on `stream` before the call.
hipError_t launch_ed25519_verify_grouped(const uint8_t* d_tuples, size_t n, const GroupBuffers& b, const EdGroupBuffers& eb,
                                         u32* d_qtab, const aniels* d_btab, const edcomb& bcomb, uint8_t* d_bitmap, hipStream_t stream,
                                         const GroupSync& y, hipEvent_t* prof = nullptr, int* prof_pairs = nullptr);   // prof: 2 * chunks events, a pair around every k_ed_qphase launch
// message front end: raw signatures / keys / messages -> 128-byte tuples on the device (sha512_dev.h)
hipError_t launch_ed_msg_frontend(const uint8_t* d_sigs, const uint8_t* d_pks, const uint8_t* d_msgs, const u64* d_moff, size_t n,
                                  u32* d_tuples, hipStream_t stream);
#define SBV_ED_KEYTAB_ENTRIES_PER_KEY 4096   // 32 windows x 128 entries (ed25519_group.h)
void host_build_ed_b16(aniels* out);      // 16 x 32768 affine-Niels multiples of B: the comb of the one-lane kernel
void host_build_ed_bcomb(int bits, aniels* out);   // edcomb_entries(bits) entries: the grouped step's comb (SBV_ED_B_BITS, default 20), one host thread per window
}  // namespace sbv

// emul.cc — CPU TEST TIER ONLY: lane-by-lane emulation of the device algorithm.
//
// Compiles the very headers the gfx950 kernels are built from (consensus_amd/csrc/p256_*.h)
// with g++ and runs stage A (prep_chunk, with the same thread -> tuple mapping as the
// kernel) and stage B (verify_lane) sequentially, so the build container (no GPU) can diff
// the device algorithm against the oracle and the golden vectors.  It is NOT part of
// libsbv.so, is never shipped and is not a fallback: the product fails loudly without a GPU.
#include <stdlib.h>
#include <algorithm>
#include <string.h>
#include <thread>
#include <string>
#include <vector>

#include "../../consensus_amd/csrc/p256_core.h"
#include "p256_legacy_m32.h"
#include "../../consensus_amd/csrc/ed25519_core.h"
#include "../../consensus_amd/csrc/ed25519_group.h"
#include "../../consensus_amd/csrc/sha512_dev.h"
#include "../../consensus_amd/csrc/sha256_dev.h"
#include "../../consensus_amd/csrc/p256_group.h"
#include "../../consensus_amd/csrc/p256_widetab29.h"
#include "../../consensus_amd/csrc/p256_pt29.h"
#include "../../consensus_amd/csrc/p256_keytab29.h"
#include "../../consensus_amd/csrc/p256_sign.h"
#include "../../consensus_amd/csrc/k256_core.h"
#include "../../consensus_amd/csrc/k256_group.h"

using namespace sbv;

static unsigned long g_sticky_reruns = 0, g_fast_mismatches = 0;
static apt* g_gtab = nullptr;
static const apt* gtab() {          // 8-bit comb (host signer / registered-key tables use this shape)
    if (!g_gtab) {
        g_gtab = (apt*)aligned_alloc(64, sizeof(apt) * SBV_GTAB_WINDOWS * SBV_GTAB_PER_WINDOW);
        build_gtable(g_gtab);
    }
    return g_gtab;
}
static apt* g_g16 = nullptr;
static const apt* g16tab() {        // 16-bit comb for G, as the verify kernels use
    if (!g_g16) {
        g_g16 = (apt*)aligned_alloc(64, sizeof(apt) * (size_t)SBV_G16_WINDOWS * SBV_G16_PER_WINDOW);
        std::vector<std::thread> th;
        for (int j = 0; j < SBV_G16_WINDOWS; ++j) th.emplace_back([j] { build_g16_window(j, g_g16 + (size_t)j * SBV_G16_PER_WINDOW); });
        for (auto& t : th) t.join();
    }
    return g_g16;
}

static apt* g_gc_tab = nullptr;
static int g_gc_bits = 16;
// the comb of G for the carry-free kernels (R = 2^261 domain); sbve_set_gcomb_bits picks the window width
static gcomb g16rtab() {
    if (!g_gc_tab) {
        const size_t count = gcomb_entries(g_gc_bits);
        g_gc_tab = (apt*)aligned_alloc(64, sizeof(apt) * count);
        apt* tmp = (apt*)aligned_alloc(64, sizeof(apt) * count);
        const int windows = (257 + g_gc_bits - 1) / g_gc_bits, bits = g_gc_bits;
        std::vector<std::thread> th;
        for (int j = 0; j < windows; ++j) th.emplace_back([=] { build_gcomb_window(bits, j, tmp + ((size_t)j << (bits - 1))); });
        for (auto& t : th) t.join();
        th.clear();
        apt* dst = g_gc_tab;
        for (int t = 0; t < 8; ++t)
            th.emplace_back([=] { for (size_t k = count * t / 8; k < count * (t + 1) / 8; ++k) apt_to_r261(dst[k], tmp[k]); });
        for (auto& t : th) t.join();
        free(tmp);
    }
    return gcomb_make(g_gc_tab, g_gc_bits);
}

struct HostWords {
    const uint8_t* tuples;
    size_t stride = 160;
    struct W {
        const uint8_t* p;
        u32 operator[](int i) const { u32 v; memcpy(&v, p + 4 * i, 4); return v; }
    };
    W operator()(int, size_t idx) const { return W{tuples + stride * idx}; }
};

extern "C" {

void sbve_set_gcomb_bits(int bits) { if (bits != g_gc_bits) { free(g_gc_tab); g_gc_tab = nullptr; g_gc_bits = bits; } }

// block = threads per emulated workgroup, T = tuples per thread (chunk for Montgomery's trick)
void sbve_p256_verify_batch(const uint8_t* tuples, size_t n, uint8_t* bitmap, int block, int T) {
    size_t cap = (n + 63) & ~(size_t)63;
    if (cap == 0) cap = 64;
    std::vector<u32> r(8 * cap), u1(8 * cap), u2(8 * cap), qx(8 * cap), qy(8 * cap), sm(8 * cap);
    std::vector<uint8_t> ok(cap, 0);
    Scratch s{r.data(), u1.data(), u2.data(), qx.data(), qy.data(), sm.data(), ok.data(), cap};
    const size_t per_block = (size_t)block * T;
    const size_t nblocks = (n + per_block - 1) / per_block;
    HostWords hw{tuples, 160};
    for (size_t b = 0; b < nblocks; ++b)
        for (int t = 0; t < block; ++t) prep_chunk29<true>(hw, n, s, b * per_block + t, (size_t)block, T);
    memset(bitmap, 0, (n + 7) / 8);
    u32* qtab = (u32*)aligned_alloc(16, SBV_QTAB29_WORDS * 4);
    for (size_t i = 0; i < n; ++i)
        if (verify29_lane_generic(s, i, qtab, g16rtab())) bitmap[i >> 3] |= (uint8_t)(1u << (i & 7));
    free(qtab);
}
// the 8 x 32 generic lane (p256_core.h: verify_lane, fast + exact passes) kept as a second implementation to diff against
void sbve_p256_verify_batch_v0(const uint8_t* tuples, size_t n, uint8_t* bitmap, int block, int T) {
    size_t cap = (n + 63) & ~(size_t)63;
    if (cap == 0) cap = 64;
    std::vector<u32> r(8 * cap), u1(8 * cap), u2(8 * cap), qx(8 * cap), qy(8 * cap), sm(8 * cap);
    std::vector<uint8_t> ok(cap, 0);
    Scratch s{r.data(), u1.data(), u2.data(), qx.data(), qy.data(), sm.data(), ok.data(), cap};
    const size_t per_block = (size_t)block * T;
    const size_t nblocks = (n + per_block - 1) / per_block;
    HostWords hw{tuples, 160};
    for (size_t b = 0; b < nblocks; ++b)
        for (int t = 0; t < block; ++t) prep_chunk<true>(hw, n, s, b * per_block + t, (size_t)block, T);
    memset(bitmap, 0, (n + 7) / 8);
    u32* qtab = (u32*)aligned_alloc(16, SBV_QTAB_ENTRIES * 40 * 4);
    for (size_t i = 0; i < n; ++i) {
        // exactly the old kernel's two-pass scheme: fast pass, exact re-run when the sticky word fired
        u32 sticky = 0;
        bool acc = verify_lane<true>(s, i, qtab, g16tab(), &sticky);
        const bool exact = verify_lane<false>(s, i, qtab, g16tab());
        if (sticky == 0xFFFFFFFFu) { ++g_sticky_reruns; acc = exact; }
        else if (acc != exact) ++g_fast_mismatches;      // must never happen: the test asserts it stays 0
        if (acc) bitmap[i >> 3] |= (uint8_t)(1u << (i & 7));
    }
    free(qtab);
}
unsigned long sbve_sticky_reruns() { return g_sticky_reruns; }
unsigned long sbve_fast_mismatches() { return g_fast_mismatches; }
// fast conditional subtraction, standalone: returns the sticky word
u32 sbve_fe_add_fast(const u32* a, const u32* b, u32* out) { fe x, y, z; memcpy(&x, a, 32); memcpy(&y, b, 32); u32 st = 0; fe_add<true>(z, x, y, &st); memcpy(out, &z, 32); return st; }
u32 sbve_fe_mul_fast(const u32* a, const u32* b, u32* out) { fe x, y, z; memcpy(&x, a, 32); memcpy(&y, b, 32); u32 st = 0; fe_mul<true>(z, x, y, &st); memcpy(out, &z, 32); return st; }

static unsigned long g_coop_disagreements = 0, g_small_disagreements = 0;     // lanes of a cooperating group that ended with different points / stage-A forms that disagreed
static int g_group_chunks = 3, g_group_sort = 1, g_group_coop = 0;
void sbve_set_group_sort(int on) { g_group_sort = on ? 1 : 0; }
static unsigned long g_sort_violations = 0;
// the key-sorted list's invariant: every group's lanes form ONE run, and the runs come in the order of group_sort_scan_seq (p256_group.h)
struct RunOrder {
    std::vector<u32> pos;            // position of group k in the scan's order
    explicit RunOrder(u32 groups) : pos(groups ? groups : 1, 0) {
        const u32 rows = group_sort_rows(groups), P = group_sort_positions(groups);
        for (u32 p = 0; p < P; ++p) { const u32 k = group_sort_group_at(p, rows); if (k < groups) pos[k] = p; }
    }
    bool ok(u32 prev, u32 cur) const { return cur < pos.size() && (prev == SBV_GROUP_NONE || (prev < pos.size() && pos[prev] <= pos[cur])); }
};
unsigned long sbve_group_sort_violations() { return g_sort_violations; }
// the order of the groups' runs in the key-sorted list (p256_group.h: group_sort_group_at): out[p] = group at position p of
// group_sort_positions(groups), SBV_GROUP_NONE for an empty position; returns the number of positions
u32 sbve_group_sort_order(u32 groups, u32* out) {
    const u32 rows = group_sort_rows(groups), P = group_sort_positions(groups);
    for (u32 p = 0; p < P; ++p) { const u32 k = group_sort_group_at(p, rows); out[p] = k < groups ? k : SBV_GROUP_NONE; }
    return P;
}
// persistent key-table cache of the emulated grouped step (sbve_key_cache resets it)
static KeyCache g_kc = {};
static apt* g_kc_ktab = nullptr;
static apt* g_kc_ntab = nullptr;            // compact rows of the cache's slots (GroupBuffers::ntab)
static std::vector<uint8_t> g_kc_full;       // kfull of the emulated P-256 key-table cache's slots (p256_group.h: table classes)
static std::vector<uint8_t> g_kc_valid;
static std::vector<u32> g_kc_ht, g_kc_keys, g_kc_count;
// hot keys (p256_group.h): the wide-comb pool of the emulated cache, its per-slot index and hit counters, hot[] as on the device
static u32 g_hot_cap = 0, g_hot_min = 4096;
static apt* g_hot_wtab = nullptr;
static std::vector<u32> g_hot_kwide, g_hot_khits, g_hot_wowner;
static u32 g_hot[4] = {0, 0, 0, 0};
static u32 g_hot_tick = 0;          // GroupBuffers::hot_tick: the clock of the decay
static u32 g_hot_evictions = 0;     // combs that changed owner so far (test statistics)
static void hot_reset() {
    g_hot_kwide.assign(g_kc.cap ? g_kc.cap : 1, SBV_WIDE_NONE); g_hot_khits.assign(g_kc.cap ? g_kc.cap : 1, 0); memset(g_hot, 0, sizeof g_hot);
    g_hot_wowner.assign(g_hot_cap ? g_hot_cap : 1, SBV_WIDE_NONE); g_hot_tick = 0; g_hot_evictions = 0;
}
void sbve_hot_keys(u32 cap, u32 min_hits) {       // after sbve_key_cache (which forgets the promotions, like the library)
    free(g_hot_wtab);
    g_hot_wtab = nullptr;
    g_hot_cap = cap;
    if (min_hits) g_hot_min = min_hits;
    if (cap) {
        g_hot_wtab = (apt*)aligned_alloc(64, (size_t)cap * gcomb_entries(SBV_HOT_BITS) * sizeof(apt));
        memset((void*)g_hot_wtab, 0xA5, (size_t)cap * gcomb_entries(SBV_HOT_BITS) * sizeof(apt));        // an entry nobody wrote must not look like a point
    }
    hot_reset();
}
// hit counter of the cache slot that holds `key` (64 bytes), or 0xFFFFFFFF; out[0] = evictions so far, out[1] = decay clock
u32 sbve_hot_hits_of_key(const uint8_t* key) {
    for (size_t sl = 0; sl < g_hot_khits.size() && sl < g_kc.cap; ++sl)
        if (memcmp(&g_kc_keys[sl * 16], key, 64) == 0) return g_hot_khits[sl];
    return 0xFFFFFFFFu;
}
u32 sbve_hot_wide_of_key(const uint8_t* key) {
    for (size_t sl = 0; sl < g_hot_kwide.size() && sl < g_kc.cap; ++sl)
        if (memcmp(&g_kc_keys[sl * 16], key, 64) == 0) return g_hot_kwide[sl];
    return 0xFFFFFFFEu;
}
void sbve_hot_life(u32 out[2]) { out[0] = g_hot_evictions; out[1] = g_hot_tick; }
void sbve_hot_stats(u32 out[4]) { out[0] = g_hot[0] < g_hot_cap ? g_hot[0] : g_hot_cap; out[1] = g_hot_cap; out[2] = g_hot[2]; out[3] = g_hot_min; }
// promoted comb `index` against the host builder (build_comb_window_of + apt_to_r261), as sbv_p256_hot_selfcheck compares: every entry
// of windows 0..15, the babies of the top window.  Number of differing entries, or (size_t)-1 if nobody owns the index.
size_t sbve_hot_comb_mismatches(u32 index) {
    size_t slot = g_hot_kwide.size();
    for (size_t i = 0; i < g_hot_kwide.size(); ++i) if (g_hot_kwide[i] == index) slot = i;
    if (slot == g_hot_kwide.size() || !g_hot_wtab) return (size_t)-1;
    u256 x, y;
    from_be32(x, (const uint8_t*)&g_kc_keys[slot * 16]);
    from_be32(y, (const uint8_t*)&g_kc_keys[slot * 16] + 32);
    const widebuild w = widebuild_make(SBV_HOT_BITS);
    const apt* tab = g_hot_wtab + (size_t)index * gcomb_entries(SBV_HOT_BITS);
    std::vector<apt> want(w.per_window);
    size_t bad = 0;
    for (int j = 0; j < w.windows; ++j) {
        build_comb_window_of(x, y, SBV_HOT_BITS, j, want.data());
        const size_t lim = j + 1 < w.windows ? w.per_window : w.babies - 1;
        for (size_t e = 0; e < lim; ++e) {
            apt t;
            apt_to_r261(t, want[e]);
            if (memcmp(&t, &tab[(size_t)j * w.per_window + e], sizeof(apt)) != 0) ++bad;
        }
    }
    return bad;
}
void sbve_key_cache(int enabled, u32 cap) {
    free(g_kc_ktab);
    g_kc_ktab = nullptr;
    free(g_kc_ntab);
    g_kc_ntab = nullptr;
    if (cap) g_kc_ntab = (apt*)aligned_alloc(64, (size_t)cap * SBV_NTAB_ENTRIES * sizeof(apt));
    size_t ht = 16;
    while (ht < 4 * (size_t)(cap ? cap : 1)) ht *= 2;
    g_kc_ht.assign(ht, 0); g_kc_keys.assign((size_t)(cap ? cap : 1) * 16, 0); g_kc_count.assign(4, 0); g_kc_valid.assign(cap ? cap : 1, 0);
    if (cap) {
        g_kc_ktab = (apt*)aligned_alloc(64, (size_t)cap * SBV_GTAB_WINDOWS * SBV_GTAB_PER_WINDOW * sizeof(apt));
        memset((void*)g_kc_ktab, 0xA5, (size_t)cap * SBV_GTAB_WINDOWS * SBV_GTAB_PER_WINDOW * sizeof(apt));     // an entry nobody wrote must not look like a point (nor like the table an earlier cache held there)
    }
    g_kc.ht = g_kc_ht.data(); g_kc.ht_mask = (u32)(ht - 1); g_kc.keys = g_kc_keys.data(); g_kc.count = g_kc_count.data();
    g_kc.cap = cap; g_kc.enabled = enabled && cap ? 1u : 0u;
    g_kc_full.assign(cap ? cap : 1, 0);
    hot_reset();
}
void sbve_key_cache_stats(u32 out[3]) { out[0] = g_kc.count ? g_kc_count[0] : 0; out[1] = g_kc.count ? g_kc_count[1] : 0; out[2] = g_kc.count ? g_kc_count[2] : 0; }
// the other two schemes' caches (sbv_key_cache(SBV_SCHEME_SECP256K1 / SBV_SCHEME_ED25519)): scheme 1 = secp256k1, 2 = Ed25519
struct EmulKeyCache {
    KeyCache kc = {};
    void* pool = nullptr;          // cap x per-key table bytes
    size_t key_bytes = 0;
    std::vector<uint8_t> valid;
    std::vector<u32> ht, keys, count;
    void reset(int enabled, u32 cap, size_t bytes_per_key) {
        free(pool);
        pool = nullptr;
        key_bytes = bytes_per_key;
        size_t h = 16;
        while (h < 4 * (size_t)(cap ? cap : 1)) h *= 2;
        ht.assign(h, 0); keys.assign((size_t)(cap ? cap : 1) * 16, 0); count.assign(4, 0); valid.assign(cap ? cap : 1, 0);
        if (cap) { pool = aligned_alloc(64, (size_t)cap * bytes_per_key); memset(pool, 0xA5, (size_t)cap * bytes_per_key); }
        kc.ht = ht.data(); kc.ht_mask = (u32)(h - 1); kc.keys = keys.data(); kc.count = count.data();
        kc.cap = cap; kc.enabled = enabled && cap ? 1u : 0u;
    }
};
static EmulKeyCache g_kc_k256, g_kc_ed;
void sbve_scheme_key_cache(int scheme, int enabled, u32 cap) {
    if (scheme == 1) g_kc_k256.reset(enabled, cap, (size_t)SBV_GTAB_WINDOWS * SBV_GTAB_PER_WINDOW * sizeof(kapt));
    else if (scheme == 2) g_kc_ed.reset(enabled, cap, (size_t)SBV_ED_KEYTAB_ENTRIES * sizeof(aniels));
    else sbve_key_cache(enabled, cap);
}
void sbve_scheme_key_cache_stats(int scheme, u32 out[3]) {
    if (scheme == 0) { sbve_key_cache_stats(out); return; }
    const EmulKeyCache& e = scheme == 1 ? g_kc_k256 : g_kc_ed;
    for (int i = 0; i < 3; ++i) out[i] = e.count.size() ? e.count[i] : 0;
}
void sbve_set_group_chunks(int c) { g_group_chunks = c < 1 ? 1 : (c > 4 ? 4 : c); }
static u32 g_full_min = 256;       // GroupBuffers::full_min: uses of a key in a batch from which its table is filled (p256_group.h: table classes)
void sbve_set_full_table_min(u32 v) { g_full_min = v; }
static u32 g_last_classes[3] = {0, 0, 0};    // last grouped P-256 batch: groups with a full table, groups filled in this batch, lanes served by the narrow pass
void sbve_last_table_classes(u32 out[3]) { for (int i = 0; i < 3; ++i) out[i] = g_last_classes[i]; }
static u32 g_hash_seed = 0;        // GroupState::seed of the emulated grouped steps (the library draws a random one per context)
void sbve_set_hash_seed(u32 s) { g_hash_seed = s; }
static void emul_window_rows_fill(const u32* recs, bool top, u32* tmp, apt* row);
static void emul_window_rows(const u32* recs, bool top, u32* tmp, apt* row, apt* crow = nullptr);
static void emul_window_fill(bool top, u32* tmp, apt* row);
static bool g_fill_shared = true;       // SBV_FILL_SHARED: one inversion per window in the fill (k_keytab29_fill_shared)
void sbve_set_fill_shared(int on) { g_fill_shared = on != 0; }
void sbve_set_group_coop(int on) { g_group_coop = on != 0; }     // k_group_coop instead of the G phase + Q launches (key-sorted step only)
// grouped form: generic tuples, keys grouped inside the call (p256_group.h), emulated sequentially.
// stats_out[0..3] = groups, grouped tuples, ungrouped tuples, ungrouped tuples rejected for their key.
void sbve_p256_verify_batch_grouped(const uint8_t* tuples, size_t n, uint8_t* bitmap, u32 min_count, u32 max_groups,
                                    u32 ht_bits, u32* stats_out) {
    size_t cap = (n + 63) & ~(size_t)63;
    if (cap == 0) cap = 64;
    std::vector<u32> r(8 * cap), u1(8 * cap), u2(8 * cap), qx(8 * cap), qy(8 * cap), sm(8 * cap);
    std::vector<uint8_t> ok(cap, 0);
    Scratch s{r.data(), u1.data(), u2.data(), qx.data(), qy.data(), sm.data(), ok.data(), cap};
    std::vector<u32> rec(g_group_sort ? cap * SBV_REC_WORDS + 4 : 4, 0xDEADBEEFu);
    u32* rec_al = rec.data();
    while ((uintptr_t)rec_al & 15) ++rec_al;
    if (g_group_sort) s.rec = rec_al;
    HostWords hw{tuples, 160};
    const int T = 4;
    const size_t per_block = (size_t)64 * T, nblocks = (n + per_block - 1) / per_block;
    for (size_t b = 0; b < nblocks; ++b)
        for (int t = 0; t < 64; ++t) prep_chunk29<true>(hw, n, s, b * per_block + t, 64, T);
    if (g_group_sort) {                  // nothing may read the limb-major planes in the key-sorted step: make a stray read visible
        for (auto* v : {&r, &u1, &u2, &qx, &qy, &sm}) std::fill(v->begin(), v->end(), 0xDBDBDBDBu);
    }
    std::vector<u32> ht((size_t)1 << ht_bits, 0), rep(cap), cnt(cap, 0), slot_of(cap), group_rep(max_groups ? max_groups : 1), counters(SBV_GROUP_COUNTERS, 0), ung_cand(cap),
        grp_idx(cap, 0xFFFFFFFFu), ung_idx(cap), slots(cap), gcount(2 * (size_t)(max_groups ? max_groups : 1), 0), grp_of(cap, 0xFFFFFFFFu);
    GroupState g{};
    g.seed = g_hash_seed;
    g.gcount = gcount.data(); g.gcursor = gcount.data() + (max_groups ? max_groups : 1); g.grp_of = grp_of.data(); g.ung_cand = ung_cand.data(); g.sorted = (u32)g_group_sort;
    g.ht = ht.data(); g.ht_mask = (u32)(((size_t)1 << ht_bits) - 1); g.rep = rep.data(); g.cnt = cnt.data(); g.slot_of = slot_of.data();
    g.group_rep = group_rep.data(); g.counters = counters.data(); g.grp_idx = grp_idx.data(); g.ung_idx = ung_idx.data();
    g.slots = slots.data(); g.max_groups = max_groups;
    group_set_threshold(g, min_count);
    for (size_t i = 0; i < n; ++i) group_insert_lane(tuples, i, g);
    for (size_t i = 0; i < n; ++i) group_assign_lane(tuples, i, g, g_kc);      // cached keys are grouped whatever their count
    std::vector<uint8_t> accb(cap, 0xEE);
    if (g.sorted) {                      // two passes (k_group_classify, k_group_keycheck); candidates visited backwards, the order is free
        for (size_t i = 0; i < n; ++i) group_classify_lane(i, g);
        for (size_t L = counters[4]; L-- > 0;) group_keycheck_lane(tuples, L, g, accb.data());
    } else {
        for (size_t i = 0; i < n; ++i) group_split_lane(tuples, i, g, accb.data());
    }
    const u32 ngroups = counters[0] < max_groups ? counters[0] : max_groups;
    if (g.sorted) {                      // counting sort of the grouped tuples by key (k_group_sort_count / _scan / _scatter)
        for (size_t i = 0; i < n; ++i) group_sort_count_lane(i, g);
        group_sort_scan_seq(g, ngroups);
        // the device scatters tile by tile in any order: walk the tuples backwards so that the emulated order differs from
        // both the tuple order and the device's — verdicts may not depend on it
        for (size_t i = n; i-- > 0;) group_sort_scatter_lane(i, g);
        // invariants of the sorted list (the test reads the violation counter): a permutation of exactly the grouped tuples,
        // one run per group in the scan's order (RunOrder), grp_of consistent with the per-tuple group
        size_t grouped = 0;
        for (size_t i = 0; i < n; ++i) grouped += slots[i] != SBV_GROUP_NONE;
        if (grouped != counters[1]) ++g_sort_violations;
        std::vector<uint8_t> seen(cap, 0);
        const RunOrder run_order(ngroups);
        for (u32 L = 0; L < counters[1]; ++L) {
            const u32 t = grp_idx[L];
            if (t >= n || seen[t] || slots[t] != grp_of[L] || !run_order.ok(L ? grp_of[L - 1] : SBV_GROUP_NONE, grp_of[L])) { ++g_sort_violations; continue; }
            seen[t] = 1;
        }
    }
    // G phase: for every tuple, or (key-sorted list) for the grouped ones at their sorted position
    std::vector<u32> gacc(SBV_GACC29_WORDS * cap);
    if (g_group_coop && g.sorted) {}          // the coop launch below does the G phase's job too
    else if (g.sorted) for (u32 L = 0; L < counters[1]; ++L) gphase29_lane_sorted(s, grp_idx[L], L, g16rtab(), gacc.data());
    else for (size_t i = 0; i < n; ++i) gphase29_lane(s, i, g16rtab(), gacc.data());
    // key tables and the Q phase, in `chunks` pieces like the device pipeline
    const size_t ng1 = ngroups ? ngroups : 1;
    u32* bases = (u32*)aligned_alloc(64, ng1 * SBV_GTAB_WINDOWS * SBV_KT29_POINTS_PER_WINDOW * SBV_KT29_REC_WORDS * sizeof(u32));
    memset(bases, 0xA5, ng1 * SBV_GTAB_WINDOWS * SBV_KT29_POINTS_PER_WINDOW * SBV_KT29_REC_WORDS * sizeof(u32));
    std::vector<u32> jstate(ng1 * SBV_KT29_STATE_WORDS);
    const size_t per_key = (size_t)SBV_GTAB_WINDOWS * SBV_GTAB_PER_WINDOW;
    apt* ktab = (apt*)aligned_alloc(64, ng1 * per_key * sizeof(apt));                     // the per-batch area
    memset(ktab, 0xA5, ng1 * per_key * sizeof(apt));      // an entry nobody wrote must not look like a point
    std::vector<uint8_t> kvalid(ng1, 0);
    std::vector<u32> tmpa(7 * SBV_KT29_FILL_TMP_WORDS);
    // persistent key-table cache (p256_group.h): the two phases of k_key_cache_assign, sequentially
    std::vector<u32> tslot(ng1, SBV_GROUP_NONE);
    std::vector<uint8_t> cold(ng1, 1);
    KeyCache kc = g_kc;
    if (kc.enabled) kc.count[1] = kc.count[2] = 0;
    for (u32 k = 0; k < ngroups; ++k) key_cache_phase_lookup<160, 96, 16>(tuples, g, kc, k, tslot.data(), cold.data());
    for (u32 k = 0; k < ngroups; ++k) key_cache_phase_insert<160, 96, 16>(tuples, g, kc, k, tslot.data());
    auto table_of = [&](u32 k) -> apt* { return tslot[k] < kc.cap ? g_kc_ktab + (size_t)tslot[k] * per_key : ktab + (size_t)k * per_key; };
    apt* ntab = (apt*)aligned_alloc(64, ng1 * (size_t)SBV_NTAB_ENTRIES * sizeof(apt));      // compact rows of the per-batch slots
    memset((void*)ntab, 0xA5, ng1 * (size_t)SBV_NTAB_ENTRIES * sizeof(apt));
    auto ntable_of = [&](u32 k) -> apt* { return tslot[k] < kc.cap ? g_kc_ntab + (size_t)tslot[k] * SBV_NTAB_ENTRIES : ntab + (size_t)k * SBV_NTAB_ENTRIES; };
    auto valid_of = [&](u32 k) -> uint8_t* { return tslot[k] < kc.cap ? &g_kc_valid[tslot[k]] : &kvalid[k]; };
    // table classes (p256_group.h; k_group_table_class): kfull = [cache slots | this batch's per-batch slots], as on the device
    const u32 table_slots = kc.cap + (u32)ng1;
    if (g_kc_full.size() < kc.cap) g_kc_full.assign(kc.cap, 0);
    std::vector<uint8_t> kfull(table_slots, 0), full(ng1, 0), needfill(ng1, 0);
    memcpy(kfull.data(), g_kc_full.data(), kc.cap);
    const u32 full_min = g_group_coop && g.sorted ? 0u : g_full_min;      // the coop launch reads any entry of a row
    for (u32 k = 0; k < ngroups; ++k) group_table_class_lane(k, g, tslot.data(), cold.data(), kfull.data(), table_slots, full_min, full.data(), needfill.data());
    g_last_classes[0] = g_last_classes[1] = g_last_classes[2] = 0;
    for (u32 k = 0; k < ngroups; ++k) { g_last_classes[0] += full[k]; g_last_classes[1] += needfill[k]; }
    // hot keys (k_group_table_class's second half): hits of the slots, and which groups may take the wide pass
    const bool hot_on = g_hot_wtab && kc.enabled && g.sorted;
    std::vector<uint8_t> wide(ng1, 0);
    if (hot_on) {
        if (g_hot_kwide.size() < kc.cap) hot_reset();
        g_hot[1] = g_hot[2] = g_hot[3] = 0;
        ++g_hot_tick;
        for (u32 k = 0; k < ngroups; ++k) group_hot_class_lane(k, g, tslot.data(), cold.data(), kc.cap, g_hot_kwide.data(), g_hot_khits.data(), wide.data());
    }
    const widekeys wk = hot_on ? widekeys_make(g_hot_wtab, g_hot_kwide.data(), SBV_HOT_BITS) : widekeys_none();
    // which lanes of the grouped list the chunk launches serve (wavefronts of 64 whose lanes ALL hold full tables) and which the narrow pass
    // (group_wave_class: SBV_Q_NONE = every lane dead, SBV_Q_WIDE, SBV_Q_FULL = the chunks' launches, SBV_Q_NARROW = rows-only pass)
    std::vector<uint8_t> wave_cls((counters[1] + 63) / 64 + 1, SBV_Q_NONE);
    auto decide_waves = [&] {            // the kernels decide at Q time: the first chunk of the chain has judged every cold key by then
        std::vector<uint8_t> all_dead(wave_cls.size(), 1), all_w(wave_cls.size(), 1), all_f(wave_cls.size(), 1);
        for (u32 L = 0; L < counters[1]; ++L) {
            const u32 grp = g.sorted ? grp_of[L] : slots[grp_idx[L]];
            const bool dead = !(grp < ngroups) || !*valid_of(grp);        // no slot, or a key that is no point: never drags its wavefront to the narrow pass
            if (dead) continue;
            all_dead[L / 64] = 0;
            const bool w = wide[grp] != 0;
            if (!w) all_w[L / 64] = 0;
            if (!full[grp]) all_f[L / 64] = 0;               // a promoted key's 8-bit table may be rows only: in a mixed wavefront it counts as what it is
        }
        for (size_t wv = 0; wv < wave_cls.size(); ++wv) wave_cls[wv] = (uint8_t)group_wave_class(all_dead[wv] != 0, all_w[wv] != 0, all_f[wv] != 0);      // the kernels' own rule (p256_group.h)
    };
    memset(bitmap, 0, (n + 7) / 8);
    const int chunks = g_group_chunks;
    // entries the fill never wrote must not be read by ANY pass: poison them in the warm rows-only tables before the first launch (the
    // cold ones sit in memory that was poisoned when it was allocated), so that a stray read shows as a wrong verdict
    auto poison_unfilled = [&](u32 k) {
        apt* tab = table_of(k);
        for (int j = 0; j < SBV_GTAB_WINDOWS - 1; ++j)
            for (int m = 1; m <= SBV_GTAB_PER_WINDOW; ++m)
                if (m > 8 && (m & 15) != 0) memset((void*)(tab + (size_t)j * SBV_GTAB_PER_WINDOW + m - 1), 0xA5, sizeof(apt));
    };
    if (!(g_group_coop && g.sorted))
        for (u32 k = 0; k < ngroups; ++k)
            if (!cold[k] && !full[k] && tslot[k] < table_slots) poison_unfilled(k);
    for (int c = 0; c < chunks; ++c) {
        const int j_first = SBV_GTAB_WINDOWS * c / chunks, j_end = SBV_GTAB_WINDOWS * (c + 1) / chunks;
        for (u32 k = 0; k < ngroups; ++k)
            if (cold[k] && (j_first == 0 || *valid_of(k))) {               // k_keytab29_chain: the four lanes of the key's quad in lockstep (an invalid key stops after its check)
                keychain_quad_host q;
                keychain29_run(q, tuples, k, g, jstate.data(), bases, valid_of(k), j_first, j_end - 1, 0x11u);
            }
        for (u32 k = 0; k < ngroups; ++k)
            for (int j = j_first; j < j_end && cold[k] && *valid_of(k); ++j) {        // k_keytab29_rows: every cold group whose key is a point
                const size_t w = (size_t)k * SBV_GTAB_WINDOWS + j;
                apt* row = table_of(k) + (size_t)j * SBV_GTAB_PER_WINDOW;
                emul_window_rows(bases + w * (SBV_KT29_POINTS_PER_WINDOW * SBV_KT29_REC_WORDS), j == SBV_GTAB_WINDOWS - 1, tmpa.data(), row,
                                 ntable_of(k) + (size_t)j * SBV_NTAB_PER_WINDOW);
            }
        for (u32 k = 0; k < ngroups; ++k)
            for (int j = j_first; j < j_end && needfill[k] && *valid_of(k); ++j)      // k_keytab29_fill_sym: the groups that earn a full table (cold, or a cached narrow one: upgrade)
                emul_window_fill(j == SBV_GTAB_WINDOWS - 1, tmpa.data(), table_of(k) + (size_t)j * SBV_GTAB_PER_WINDOW);
        const bool last = c + 1 == chunks;
        if (c == 0) decide_waves();
        if (g_group_coop && g.sorted) {
            if (!last) continue;
            // k_group_coop: SBV_COOP_LANES partial sums per grouped tuple (comb of G + the key's table), xor-butterfly of exact
            // XYZZ additions; every lane of a group must end with the same point
            for (u32 L = 0; L < counters[1]; ++L) {
                const u32 t = grp_idx[L];
                const u32 grp = grp_of[L];
                const u32 table_slots = kc.cap + (u32)ng1;
                u32 slot = grp < ngroups ? tslot[grp] : SBV_GROUP_NONE;
                bool okk = slot < table_slots;
                const apt* tab = okk ? table_of(grp) : ktab;
                okk = okk && *valid_of(grp < ngroups ? grp : 0) != 0 && s.rec[(size_t)t * SBV_REC_WORDS + SBV_REC_OK] != 0;
                u256 a, bb, rr;
                rec_load256(a, s.rec, t, SBV_REC_U1);
                rec_load256(bb, s.rec, t, SBV_REC_U2);
                rec_load256(rr, s.rec, t, SBV_REC_R);
                xyzz part[SBV_COOP_LANES];
                for (int sub = 0; sub < SBV_COOP_LANES; ++sub) keyed29_partial_lane(part[sub], a, bb, tab, g16rtab(), sub);
                for (int off = SBV_COOP_LANES / 2; off >= 1; off >>= 1) {
                    xyzz nxt[SBV_COOP_LANES];
                    for (int sub = 0; sub < SBV_COOP_LANES; ++sub) { nxt[sub] = part[sub]; pt29_add(nxt[sub], part[sub ^ off]); }
                    for (int sub = 0; sub < SBV_COOP_LANES; ++sub) part[sub] = nxt[sub];
                }
                const bool v = okk && pt29_rx_matches(part[0], rr);
                for (int sub = 1; sub < SBV_COOP_LANES; ++sub)
                    if ((okk && pt29_rx_matches(part[sub], rr)) != v) g_coop_disagreements++;
                if (v) bitmap[t >> 3] |= (uint8_t)(1u << (t & 7));
            }
            continue;
        }
        for (u32 L = 0; L < counters[1]; ++L) {                       // k_verify_keyed_q<false>: the wavefronts whose lanes all hold full tables
            if (wave_cls[L / 64] != SBV_Q_FULL) continue;
            const u32 t = grp_idx[L];
            const u32 grp = g.sorted ? grp_of[L] : slots[t];
            if (grp < ngroups && !*valid_of(grp)) continue;           // a key that is no point has no table: rejected (the device computes on whatever the slot holds and drops the result)
            bool v;
            if (g.sorted) v = grp < ngroups ? qphase29_lane_sorted(s, t, L, 0, 1, table_of(grp), valid_of(grp), gacc.data(), j_first, j_end, last)
                                            : qphase29_lane_sorted(s, t, L, SBV_GROUP_NONE, 1, ktab, kvalid.data(), gacc.data(), j_first, j_end, last);
            else v = grp < ngroups ? qphase29_lane(s, t, 0, 1, table_of(grp), valid_of(grp), gacc.data(), j_first, j_end, last)
                                   : qphase29_lane(s, t, SBV_GROUP_NONE, 1, ktab, kvalid.data(), gacc.data(), j_first, j_end, last);
            if (last && v) bitmap[t >> 3] |= (uint8_t)(1u << (t & 7));
        }
    }
    if (!(g_group_coop && g.sorted)) {
        // k_verify_keyed_q<true>: every other wavefront, all 33 windows from babies and giants (two additions per window).  Entries the
        // fill never wrote must not be read: poison them first (0xA5 is what a fresh per-batch table holds; a cached table's unfilled
        // entries are poisoned here too), so that a stray read shows as a wrong verdict.
        for (u32 k = 0; k < ngroups; ++k) {
            if (full[k]) continue;
            apt* tab = table_of(k);
            for (int j = 0; j < SBV_GTAB_WINDOWS - 1; ++j)
                for (int m = 1; m <= SBV_GTAB_PER_WINDOW; ++m)
                    if (m > 8 && (m & 15) != 0) memset((void*)(tab + (size_t)j * SBV_GTAB_PER_WINDOW + m - 1), 0xA5, sizeof(apt));
        }
        for (u32 L = 0; L < counters[1]; ++L) {
            if (wave_cls[L / 64] != SBV_Q_NARROW) continue;
            ++g_last_classes[2];
            const u32 t = grp_idx[L];
            const u32 grp = g.sorted ? grp_of[L] : slots[t];
            if (grp < ngroups && !*valid_of(grp)) continue;
            bool v;
            if (g.sorted) v = grp < ngroups ? qphase29_lane_sorted<true>(s, t, L, 0, 1, ntable_of(grp), valid_of(grp), gacc.data(), 0, SBV_GTAB_WINDOWS, true)
                                            : qphase29_lane_sorted<true>(s, t, L, SBV_GROUP_NONE, 1, ntab, kvalid.data(), gacc.data(), 0, SBV_GTAB_WINDOWS, true);
            else v = grp < ngroups ? qphase29_lane<true>(s, t, 0, 1, ntable_of(grp), valid_of(grp), gacc.data(), 0, SBV_GTAB_WINDOWS, true)
                                   : qphase29_lane<true>(s, t, SBV_GROUP_NONE, 1, ntab, kvalid.data(), gacc.data(), 0, SBV_GTAB_WINDOWS, true);
            if (v) bitmap[t >> 3] |= (uint8_t)(1u << (t & 7));
        }
    }
    if (hot_on && !(g_group_coop && g.sorted)) {
        // k_verify_keyed_q<SBV_Q_WIDE>: the wavefronts whose live lanes all own a wide comb — u2 * Q from the 16-bit comb of the slot
        for (u32 L = 0; L < counters[1]; ++L) {
            if (wave_cls[L / 64] != SBV_Q_WIDE) continue;
            ++g_hot[2];
            const u32 t = grp_idx[L];
            const u32 grp = grp_of[L];
            const u32 ts = grp < ngroups ? tslot[grp] : SBV_GROUP_NONE;
            const bool dead = !(ts < table_slots) || !*valid_of(grp);
            u256 u2, rr;
            rec_load256(u2, s.rec, t, SBV_REC_U2);
            rec_load256(rr, s.rec, t, SBV_REC_R);
            xyzz R;
            gacc29_load(R, gacc.data(), s.cap, L);
            wide_qphase29_point(R, u2, wk, dead ? 0u : wk.idx[ts]);
            if (!dead && s.rec[(size_t)t * SBV_REC_WORDS + SBV_REC_OK] != 0 && pt29_rx_matches(R, rr)) bitmap[t >> 3] |= (uint8_t)(1u << (t & 7));
        }
    }
    // k_group_table_mark: what the cache slots hold from now on
    for (u32 k = 0; k < ngroups; ++k) group_table_mark_lane(k, tslot.data(), cold.data(), full.data(), needfill.data(), table_slots, kfull.data());
    memcpy(g_kc_full.data(), kfull.data(), kc.cap);
    if (hot_on) {
        // promotions: k_promote_select -> _bases -> _chains -> _fill -> _publish, lane by lane (groups visited backwards: the order is free)
        // the life cycle first (k_hot_decay): the clock sweep
        if (g_hot_tick % SBV_HOT_DECAY_EVERY == SBV_HOT_DECAY_EVERY - 1) for (u32 sl = 0; sl < kc.cap; ++sl) hot_decay_lane(sl, g_hot_khits.data());
        std::vector<u32> plist(2 * SBV_PROMOTE_MAX, 0xDEADBEEFu), elist(SBV_PROMOTE_MAX, 0xDEADBEEFu);
        for (u32 k = ngroups; k-- > 0;)
            group_promote_select_lane(k, tslot.data(), g_kc_valid.data(), kc.cap, g_hot_kwide.data(), g_hot_khits.data(), g_hot_min, g_hot_cap, g_hot, plist.data(), elist.data());
        {   // k_promote_evict: one agent, the same scan / merge / commit functions (p256_group.h); 3 "lanes" so that the merge runs too
            const u32 ncand = g_hot[3] < SBV_PROMOTE_MAX ? g_hot[3] : SBV_PROMOTE_MAX;
            u32 entries = g_hot[1] < SBV_PROMOTE_MAX ? g_hot[1] : SBV_PROMOTE_MAX;
            std::vector<u32> taken((g_hot_cap + 31) / 32 + 1, 0);
            for (u32 c = 0; c < ncand; ++c) {
                u32 bh = 0xFFFFFFFFu, bw = 0xFFFFFFFFu;
                for (u32 lane = 0; lane < 3; ++lane) {
                    u32 h, w;
                    hot_evict_scan(g_hot_khits.data(), g_hot_wowner.data(), taken.data(), g_hot_cap, kc.cap, lane, 3u, h, w);
                    if (hot_evict_better(h, w, bh, bw)) { bh = h; bw = w; }
                }
                const u32 before = entries;
                entries = hot_evict_commit(elist[c], bh, bw, g_hot_khits.data(), g_hot_kwide.data(), g_hot_wowner.data(), taken.data(), entries, plist.data());
                g_hot_evictions += entries - before;
            }
            if (ncand) g_hot[1] = entries;
        }
        const u32 live = g_hot[1] < SBV_PROMOTE_MAX ? g_hot[1] : SBV_PROMOTE_MAX;
        const widebuild w = widebuild_make(SBV_HOT_BITS);
        const size_t stride = gcomb_entries(SBV_HOT_BITS);
        std::vector<apt> pb((size_t)SBV_PROMOTE_MAX * 2 * w.windows);
        memset((void*)pb.data(), 0xA5, pb.size() * sizeof(apt));
        for (u32 i = 0; i < live; ++i)
            for (u32 e = 0; e < 2u * w.windows; ++e) promote_base_lane(i, e, plist.data(), g_kc_ktab, pb.data());
        std::vector<u32> tmpw((size_t)widebuild_chain_len(w) * SBV_WIDETAB_REC_WORDS);
        const u32 fchunks = widebuild_fill_chunks(w);
        for (u32 i = 0; i < live; ++i) {
            if (plist[2 * i] == 0xFFFFFFFFu) continue;
            const apt* kb = pb.data() + (size_t)i * 2 * w.windows;
            apt* comb = g_hot_wtab + (size_t)plist[2 * i + 1] * stride;
            for (int j = 0; j < w.windows; ++j)
                for (int role = 1; role >= 0; --role) widetab_chain_role(w, kb + j, kb + w.windows + j, role, tmpw.data(), comb + (size_t)j * w.per_window);
            for (int j = 0; j < w.windows; ++j)
                for (u32 gi = 1; gi < w.giants; ++gi)
                    for (u32 c = fchunks; c-- > 0;) widetab_fill_lane(w, gi, 1u + c * SBV_WIDETAB_T, comb + (size_t)j * w.per_window);
        }
        for (u32 i = 0; i < live; ++i)
            if (plist[2 * i] != 0xFFFFFFFFu) { g_hot_kwide[plist[2 * i]] = plist[2 * i + 1]; g_hot_wowner[plist[2 * i + 1]] = plist[2 * i]; }
    }
    u32* qtab = (u32*)aligned_alloc(16, SBV_QTAB29_WORDS * 4);
    for (u32 L = 0; L < counters[2]; ++L) {
        const u32 t = ung_idx[L];
        const bool v = s.rec ? verify29_lane_generic_rec(s, tuples, t, qtab, g16rtab()) : verify29_lane_generic(s, t, qtab, g16rtab());
        if (v) bitmap[t >> 3] |= (uint8_t)(1u << (t & 7));
    }
    free(qtab); free(ktab); free(ntab); free(bases);
    if (stats_out) { stats_out[0] = ngroups; stats_out[1] = counters[1]; stats_out[2] = counters[2]; stats_out[3] = counters[3]; }
}

// rows + fill of one (key, window) the way the launcher does it: k_keytab29_rows (two lanes), then k_keytab29_fill_sym (lane
// a - 1 fills both sides of giant 16 a; the lanes run in descending order here so that a lane that wrongly depended on
// another's output would show)
static void emul_window_rows(const u32* recs, bool top, u32* tmp, apt* row, apt* crow) {
    for (int which = 0; which < 2; ++which) {
        if (which == 1 && top) continue;
        keytab29_rows_lane(recs, which, top, tmp, row, crow);
    }
}
static void emul_window_fill(bool top, u32* tmp, apt* row) {
    if (top) return;
    if (!g_fill_shared) {
        for (int a = 8; a >= 1; --a) keytab29_fill_sym_lane(a, tmp + (size_t)(a - 1) * (SBV_KT29_FILL_TMP_WORDS / 2), row);
        return;
    }
    // k_keytab29_fill_shared: the eight lanes' products, ONE inversion for the window, the eight walks back
    u32 w[16 * 9];
    for (int a = 8; a >= 1; --a) {
        fe29 acc;
        keytab29_fill_sym_acc(a, tmp + (size_t)(a - 1) * (SBV_KT29_FILL_TMP_WORDS / 2), row, acc);
        f29_store_raw(w + (a - 1) * 9, acc);
    }
    keytab29_fill_group_inverses(w);
    for (int a = 1; a <= 8; ++a) {
        fe29 inv;
        f29_load_raw(inv, w + (a - 1) * 9);
        keytab29_fill_sym_finish(a, tmp + (size_t)(a - 1) * (SBV_KT29_FILL_TMP_WORDS / 2), row, inv);
    }
}
static void emul_window_rows_fill(const u32* recs, bool top, u32* tmp, apt* row) {
    emul_window_rows(recs, top, tmp, row, nullptr);
    emul_window_fill(top, tmp, row);
}

// n doublings of the affine point (x, y) (plain words) through the quad-cooperative chain of the table builder
// (p256_keytab29.h: keychain29_dbl, four lanes in lockstep) -> affine plain x, y.  Returns 0 when the four lanes disagree or
// the carried T differs from -3 Z^4, 1 otherwise.
int sbve_keychain_dbl(const u32* x8, const u32* y8, int n, u32* outx, u32* outy) {
    u256 px, py;
    memcpy(&px, x8, 32); memcpy(&py, y8, 32);
    fe29 x, y;
    f29_from_plain(x, px);
    f29_from_plain(y, py);
    keychain_quad_host q;
    for (int i = 0; i < 4; ++i) keychain29_start(q.s[i], x, y);
    for (int d = 0; d < n; ++d) keychain29_dbl(q);
    int ok = 1;
    for (int i = 1; i < 4; ++i) ok &= memcmp(&q.s[i], &q.s[0], sizeof(kchain)) == 0;
    const kchain& s = q.s[0];
    fe29 z2, z4, t, zi, zi2, zi3, ax, ay, one1 = f29_zero();
    one1.v[0] = 1;
    f29_sqr(z2, s.Z); f29_sqr(z4, z2);
    f29_add(t, z4, z4); f29_add(t, t, z4); f29_add(t, t, s.T);      // 3 Z^4 + T == 0 ?
    ok &= f29_is_zero(t) ? 1 : 0;
    f29_inv(zi, s.Z);
    f29_sqr(zi2, zi); f29_mul(zi3, zi2, zi);
    f29_mul(ax, s.X, zi2); f29_mul(ay, s.Y, zi3);
    f29_mul(ax, ax, one1); f29_mul(ay, ay, one1);                   // out of the Montgomery domain
    fe29 c;
    f29_canon(c, ax); f29_pack(outx, c);
    f29_canon(c, ay); f29_pack(outy, c);
    return ok;
}

// The whole per-batch table of ONE key as the grouped step builds it (k_keytab29_chain -> k_keytab29_rows -> k_keytab29_fill_sym,
// in `chunks` pieces): table = 33 x 128 entries of 16 words (x | y canonical words of the R = 2^261 domain).  key64 = Qx | Qy
// big-endian.  Returns the key's pointFromAffine verdict.
int sbve_keytab_build(const uint8_t* key64, int chunks, u32* table) {
    std::vector<uint8_t> tup(160, 0);
    memcpy(tup.data() + 96, key64, 64);
    u32 rep0 = 0, counters[SBV_GROUP_COUNTERS] = {1};
    GroupState g{};
    g.seed = g_hash_seed;
    g.group_rep = &rep0; g.counters = counters; g.max_groups = 1;
    std::vector<u32> bases((size_t)SBV_GTAB_WINDOWS * SBV_KT29_POINTS_PER_WINDOW * SBV_KT29_REC_WORDS, 0xA5A5A5A5u), jstate(SBV_KT29_STATE_WORDS), tmpa(7 * SBV_KT29_FILL_TMP_WORDS);
    uint8_t valid = 0xEE;
    apt* ktab = reinterpret_cast<apt*>(table);
    for (int c = 0; c < chunks; ++c) {
        const int j_first = SBV_GTAB_WINDOWS * c / chunks, j_end = SBV_GTAB_WINDOWS * (c + 1) / chunks;
        keychain_quad_host q;
        if (c > 0 && !valid) break;          // as the kernels: a key pointFromAffine refuses gets no table (the chain stops after its check, rows / fill skip the slot)
        keychain29_run(q, tup.data(), 0, g, jstate.data(), bases.data(), &valid, j_first, j_end - 1, 0x11u);
        if (!valid) break;
        for (int j = j_first; j < j_end; ++j) {
            apt* row = ktab + (size_t)j * SBV_GTAB_PER_WINDOW;
            emul_window_rows_fill(bases.data() + (size_t)j * (SBV_KT29_POINTS_PER_WINDOW * SBV_KT29_REC_WORDS), j == SBV_GTAB_WINDOWS - 1, tmpa.data(), row);
        }
    }
    return valid;
}

static bool g_keyed_coop = false, g_keyed_prepared = false;
// wide combs of the first registered slots (p256_comb29.h: widekeys; libsbv: sbv_p256_wide_keys): 0 keys = off
static int g_keyed_wide_bits = 16;
static u32 g_keyed_wide_n = 0;
static std::vector<u32> g_prep_rec, g_prep_slot;
// the host half of the prepared latency form, as consensus_amd/csrc/p256_kernels.hip: host_prep_small does it
struct EmulRecWords {
    const uint8_t* base;
    struct W {
        const uint8_t* p;
        u32 operator[](int i) const { u32 v; memcpy(&v, p + 4 * i, 4); return v; }
    };
    W operator()(int, size_t idx) const { return W{base + 96 * idx}; }
};
static void emul_host_prep_small(const uint8_t* rsh, const u32* slots, size_t n, u32* rec, u32* slot_out) {
    const size_t cap = 32;
    u32 pr[8 * 32], pu1[8 * 32], pu2[8 * 32], psm[8 * 32];
    uint8_t ok[32];
    Scratch s{pr, pu1, pu2, nullptr, nullptr, psm, ok, cap};
    prep_chunk29<false>(EmulRecWords{rsh}, n, s, 0, 1, (int)n);
    for (size_t i = 0; i < n; ++i) {
        u256 t;
        soa_load(t, s.r, cap, i); memcpy(rec + 24 * i, t.v, 32);
        soa_load(t, s.u1, cap, i); memcpy(rec + 24 * i + 8, t.v, 32);
        soa_load(t, s.u2, cap, i); memcpy(rec + 24 * i + 16, t.v, 32);
        slot_out[i] = ok[i] ? slots[i] : 0xFFFFFFFFu;
    }
}
// registered-key form: rsh = n x 96 B (r|s|hash), slots[i] indexes keys (nkeys x 64 B, Qx|Qy)
void sbve_p256_verify_batch_keyed(const uint8_t* rsh, const u32* slots, size_t n, const uint8_t* keys, u32 nkeys,
                                  uint8_t* bitmap, int block, int T) {
    size_t cap = (n + 63) & ~(size_t)63;
    if (cap == 0) cap = 64;
    std::vector<u32> r(8 * cap), u1(8 * cap), u2(8 * cap), sm(8 * cap);
    std::vector<uint8_t> ok(cap, 0);
    Scratch s{r.data(), u1.data(), u2.data(), nullptr, nullptr, sm.data(), ok.data(), cap};
    const size_t per_block = (size_t)block * T;
    const size_t nblocks = (n + per_block - 1) / per_block;
    HostWords hw{rsh, 96};
    for (size_t b = 0; b < nblocks; ++b)
        for (int t = 0; t < block; ++t) prep_chunk29<false>(hw, n, s, b * per_block + t, (size_t)block, T);
    const size_t per_key = (size_t)SBV_GTAB_WINDOWS * SBV_GTAB_PER_WINDOW;
    // the registry of one test is the same from call to call: its tables are kept (what sbv_p256_register_keys does once)
    static std::string reg_keys, wide_keys;
    static std::vector<apt> ktab, wtab;
    static std::vector<uint8_t> kvalid;
    static int wide_bits_built = 0;
    const std::string keys_now((const char*)keys, 64 * (size_t)nkeys);
    if (keys_now != reg_keys || ktab.empty()) {
        reg_keys = keys_now;
        wide_keys.clear();
        ktab.assign(per_key * (nkeys ? nkeys : 1), apt{});
        kvalid.assign(nkeys ? nkeys : 1, 0);
        for (u32 k = 0; k < nkeys; ++k) {
            u256 x, y;
            from_be32(x, keys + 64 * k);
            from_be32(y, keys + 64 * k + 32);
            kvalid[k] = key_is_valid(x, y) ? 1 : 0;
            if (kvalid[k]) {
                build_comb_table(x, y, &ktab[per_key * k]);
                for (size_t e = 0; e < per_key; ++e) { apt t; apt_to_r261(t, ktab[per_key * k + e]); ktab[per_key * k + e] = t; }
            }
        }
    }
    // wide combs of slots [0, nw): what extend_wide_keys (sbv_api.hip) builds at registration
    const u32 nw = g_keyed_wide_n < nkeys ? g_keyed_wide_n : nkeys;
    const size_t wstride = gcomb_entries(g_keyed_wide_bits);
    const std::string wide_now((const char*)keys, 64 * (size_t)nw);
    if (wide_now != wide_keys || wide_bits_built != g_keyed_wide_bits || wtab.empty()) {
        wide_keys = wide_now;
        wide_bits_built = g_keyed_wide_bits;
        wtab.assign(nw ? nw * wstride : 1, apt{});
        for (u32 k = 0; k < nw; ++k) {
            if (!kvalid[k]) continue;                      // zeros, as the library leaves them: kvalid rejects whatever the lanes add
            u256 x, y;
            from_be32(x, keys + 64 * k);
            from_be32(y, keys + 64 * k + 32);
            const int windows = (257 + g_keyed_wide_bits - 1) / g_keyed_wide_bits;
            for (int j = 0; j < windows; ++j) {
                apt* row = &wtab[k * wstride + ((size_t)j << (g_keyed_wide_bits - 1))];
                build_comb_window_of(x, y, g_keyed_wide_bits, j, row);
                for (size_t e = 0; e < ((size_t)1 << (g_keyed_wide_bits - 1)); ++e) { apt t; apt_to_r261(t, row[e]); row[e] = t; }
            }
        }
    }
    std::vector<u32> widx(nkeys ? nkeys : 1, SBV_WIDE_NONE);       // d_kwidx: the library maps any subset of the slots; here the first nw
    for (u32 k = 0; k < nw; ++k) widx[k] = k;
    const widekeys wk = nw ? widekeys_make(wtab.data(), widx.data(), g_keyed_wide_bits) : widekeys_none();
    memset(bitmap, 0, (n + 7) / 8);
    for (size_t i = 0; i < n; ++i) {
        bool accept;
        if (!g_keyed_coop) {
            accept = verify29_lane_keyed(s, i, slots[i], nkeys, ktab.data(), kvalid.data(), g16rtab(), wk);
        } else {
            // k_p256_verify_keyed_coop: SBV_COOP_LANES partial sums, xor-butterfly of exact XYZZ additions
            u32 slot = slots[i];
            bool okk = s.ok[i] != 0 && slot < nkeys;
            if (slot >= nkeys) slot = 0;
            okk = okk && kvalid[slot] != 0;
            u256 a, b, rr;
            soa_load(a, s.u1, s.cap, i);
            soa_load(b, s.u2, s.cap, i);
            soa_load(rr, s.r, s.cap, i);
            if (g_keyed_prepared) {    // k_p256_verify_prepared_small: stage A by the host half (one chunk of all n records, ONE inversion)
                if (n <= 32) {
                    if (i == 0) {
                        g_prep_rec.assign(24 * n, 0); g_prep_slot.assign(n, 0);
                        emul_host_prep_small(rsh, slots, n, g_prep_rec.data(), g_prep_slot.data());
                    }
                    u256 r2, a2, b2;
                    memcpy(r2.v, &g_prep_rec[24 * i], 32); memcpy(a2.v, &g_prep_rec[24 * i + 8], 32); memcpy(b2.v, &g_prep_rec[24 * i + 16], 32);
                    const bool ok1 = g_prep_slot[i] != 0xFFFFFFFFu;
                    if (ok1 != (s.ok[i] != 0) || !eq256(rr, r2) || (ok1 && (!eq256(a, a2) || !eq256(b, b2)))) g_small_disagreements++;
                    a = a2; b = b2; rr = r2;
                    okk = g_prep_slot[i] < nkeys && kvalid[slot] != 0;
                }
            }
            const int L = g_keyed_prepared ? 16 : SBV_COOP_LANES;      // SBV_SMALL_LANES of the prepared form
            xyzz part[16];
            // the kernels decide per wavefront (wave_all over its 8 / 4 signatures); a signature alone is the strictest case of it
            const u32 wi = widekeys_index(wk, slot);            // slot is clamped already; a rejected record keeps ok = false whatever its comb
            const bool wide = wi != SBV_WIDE_NONE;
            const gcomb kw = {wk.tab + (size_t)(wide ? wi : 0u) * wk.stride, wk.bits, wk.windows};
            for (int sub = 0; sub < L; ++sub) keyed29_partial_lane(part[sub], a, b, &ktab[per_key * slot], g16rtab(), sub, L, wide, kw);
            for (int off = L / 2; off >= 1; off >>= 1) {
                xyzz nxt[16];
                for (int sub = 0; sub < L; ++sub) { nxt[sub] = part[sub]; pt29_add(nxt[sub], part[sub ^ off]); }
                for (int sub = 0; sub < L; ++sub) part[sub] = nxt[sub];
            }
            accept = okk && pt29_rx_matches(part[0], rr);
            for (int sub = 1; sub < L; ++sub)           // every lane of the group must hold the same point
                if ((okk && pt29_rx_matches(part[sub], rr)) != accept) g_coop_disagreements++;
        }
        if (accept) bitmap[i >> 3] |= (uint8_t)(1u << (i & 7));
    }
}
void sbve_set_keyed_coop(int on) { g_keyed_coop = on != 0; g_keyed_prepared = on == 3; }   // 1: 8 lanes per signature (k_p256_verify_keyed_coop); 3: the one-launch latency form (stage A by the host half, 16 lanes per signature)
// The device-side builder of a wide comb (p256_widetab29.h: k_widetab_chains + k_widetab_fill), lane by lane, against the host
// builder (build_comb_window_of + apt_to_r261: what host_build_wide_key_table does).  Returns the number of differing entries
// (0 = byte for byte equal), or (size_t)-1 when the key is no point.
size_t sbve_widetab_build_mismatches(const uint8_t* key64, int bits) {
    u256 x, y;
    from_be32(x, key64);
    from_be32(y, key64 + 32);
    if (!key_is_valid(x, y)) return (size_t)-1;
    const widebuild w = widebuild_make(bits);
    const size_t stride = gcomb_entries(bits);
    std::vector<apt> bases(2 * (size_t)w.windows);
    comb_bases_bc(x, y, bits, w.hb, w.windows, bases.data(), bases.data() + w.windows);
    for (auto& b : bases) { apt c; apt_to_r261(c, b); b = c; }
    std::vector<apt> tab(stride);
    memset((void*)tab.data(), 0xA5, stride * sizeof(apt));                 // an entry nobody wrote must not look like a point
    std::vector<u32> tmp((size_t)widebuild_chain_len(w) * SBV_WIDETAB_REC_WORDS);
    for (int j = 0; j < w.windows; ++j)
        for (int role = 0; role < 2; ++role)
            widetab_chain_role(w, &bases[j], &bases[w.windows + j], role, tmp.data(), &tab[(size_t)j * w.per_window]);
    const u32 chunks = widebuild_fill_chunks(w);
    for (int j = 0; j < w.windows; ++j)
        for (u32 g = 1; g < w.giants; ++g)
            for (u32 c = chunks; c-- > 0;)                                  // any order: lanes are independent
                widetab_fill_lane(w, g, 1u + c * SBV_WIDETAB_T, &tab[(size_t)j * w.per_window]);
    std::vector<apt> want(w.per_window);
    size_t bad = 0;
    for (int j = 0; j < w.windows; ++j) {
        build_comb_window_of(x, y, bits, j, want.data());
        for (size_t e = 0; e < w.per_window; ++e) {
            apt t;
            apt_to_r261(t, want[e]);
            if (memcmp(&t, &tab[(size_t)j * w.per_window + e], sizeof(apt)) != 0) ++bad;
        }
    }
    return bad;
}

void sbve_set_keyed_wide(int bits, unsigned n_wide) { if (bits >= 8 && bits <= 20) g_keyed_wide_bits = bits; g_keyed_wide_n = n_wide; }
unsigned long sbve_small_disagreements() { return g_small_disagreements; }
unsigned long sbve_coop_disagreements() { return g_coop_disagreements; }

// device message front end (SHA-256 + strict DER) emulated: -> 96-byte r|s|hash record
void sbve_msg_frontend(const uint8_t* msg, size_t mlen, const uint8_t* der, size_t dlen, uint8_t out96[96]) {
    u32 w[24];
    msg_frontend_lane(msg, mlen, der, dlen, w);
    memcpy(out96, w, 96);
}

// ---- Ed25519 -----------------------------------------------------------------------------------------
static aniels* g_btab = nullptr;
static const aniels* btab() {
    if (!g_btab) {
        g_btab = (aniels*)aligned_alloc(64, sizeof(aniels) * SBV_ED_B16_ENTRIES);
        std::vector<std::thread> th;
        for (int j = 0; j < SBV_ED_B16_WINDOWS; ++j) th.emplace_back([j] { build_ed_b16_window(j, g_btab + (size_t)j * SBV_ED_B16_PER_WINDOW); });
        for (auto& t : th) t.join();
    }
    return g_btab;
}
static unsigned long g_ed_chain_mismatches = 0;      // quad-lane base chain against the one-lane chain: bases or key verdicts that differ
unsigned long sbve_ed_chain_mismatches() { return g_ed_chain_mismatches; }
static unsigned long g_ed_quad_mismatches = 0;       // quad form of the one-lane kernel against the one-lane kernel: verdicts that differ (any of the four lanes)
unsigned long sbve_ed_quad_mismatches() { return g_ed_quad_mismatches; }
// the grouped step's comb of B at another width (ed25519_group.h: edcomb; libsbv: SBV_ED_B_BITS, default 20).  The emulator's default
// stays 16 (the one-lane table, no second build); tests switch to 12 / 13 / 19 / 20 bits
static aniels* g_ed_bcomb = nullptr;
static int g_ed_bbits = 16, g_ed_bbits_built = 0;
void sbve_set_ed_b_bits(int bits) { if (bits >= 8 && bits <= 20) g_ed_bbits = bits; }
static edcomb ed_bcomb() {
    if (g_ed_bbits == 16) return edcomb_make(btab(), 16);
    if (g_ed_bbits_built != g_ed_bbits) {
        free(g_ed_bcomb);
        g_ed_bcomb = (aniels*)aligned_alloc(64, sizeof(aniels) * edcomb_entries(g_ed_bbits));
        std::vector<std::thread> th;
        const int bits = g_ed_bbits;
        for (int j = 0; j < edcomb_windows(bits); ++j) th.emplace_back([bits, j] { build_ed_b_window(bits, j, g_ed_bcomb + ((size_t)j << (bits - 1))); });
        for (auto& t : th) t.join();
        g_ed_bbits_built = bits;
    }
    return edcomb_make(g_ed_bcomb, g_ed_bbits);
}
// [S]B through the grouped step's comb walk (ed25519_group.h: ed_add_sB_comb) at the width set by sbve_set_ed_b_bits -> encode(R)
void sbve_ed_comb_mul(const u32* s8, uint8_t out32[32]) {
    u256 S;
    memcpy(&S, s8, 32);
    ept R;
    ed_set_ident(R);
    ed_add_sB_comb(R, S, ed_bcomb());
    fe25 zi, x, y;
    fe25_inv(zi, R.Z);
    fe25_mul(x, R.X, zi);
    fe25_mul(y, R.Y, zi);
    u256 yw;
    fe25_freeze(yw, y);
    yw.v[7] |= (fe25_is_negative(x) ? 1u : 0u) << 31;
    memcpy(out32, &yw, 32);
}
struct EdWords {
    const uint8_t* p;
    u32 operator[](int i) const { u32 v; memcpy(&v, p + 4 * i, 4); return v; }
};
void sbve_ed25519_verify_batch(const uint8_t* tuples, size_t n, uint8_t* bitmap) {
    memset(bitmap, 0, (n + 7) / 8);
    u32* qtab = (u32*)aligned_alloc(16, SBV_ED_QTAB_ENTRIES * SBV_ED_PT_WORDS * 4);
    for (size_t i = 0; i < n; ++i)
        if (ed25519_verify_lane(EdWords{tuples + 128 * i}, qtab, btab())) bitmap[i >> 3] |= (uint8_t)(1u << (i & 7));
    free(qtab);
}
// hot keys of the Ed25519 scheme (ed25519_group.h; EdGroupBuffers' pool): the library's state, emulated
static uint8_t* g_ed_hot_wtab = nullptr;
static std::vector<u32> g_ed_hot_kwide, g_ed_hot_khits, g_ed_hot_wowner;
static u32 g_ed_hot[4] = {0, 0, 0, 0};
static u32 g_ed_hot_cap = 0, g_ed_hot_min = 4096, g_ed_hot_tick = 0, g_ed_hot_evictions = 0;
void sbve_ed_hot_keys(u32 cap, u32 min_hits) {       // after sbve_scheme_key_cache(2, ...): a new cache forgets the promotions, like the library
    free(g_ed_hot_wtab);
    g_ed_hot_wtab = nullptr;
    g_ed_hot_cap = cap;
    if (min_hits) g_ed_hot_min = min_hits;
    if (cap) {
        g_ed_hot_wtab = (uint8_t*)aligned_alloc(128, (size_t)cap * SBV_ED_HOT_COMB_BYTES);
        memset(g_ed_hot_wtab, 0xA5, (size_t)cap * SBV_ED_HOT_COMB_BYTES);
    }
    const u32 K = g_kc_ed.kc.cap ? g_kc_ed.kc.cap : 1;
    g_ed_hot_kwide.assign(K, SBV_WIDE_NONE); g_ed_hot_khits.assign(K, 0); g_ed_hot_wowner.assign(cap ? cap : 1, SBV_WIDE_NONE);
    memset(g_ed_hot, 0, sizeof g_ed_hot);
    g_ed_hot_tick = 0; g_ed_hot_evictions = 0;
}
// out[0] = promoted keys, [1] = pool capacity, [2] = tuples the wide pass served in the last batch, [3] = min_hits, [4] = evictions so far, [5] = the decay's clock
void sbve_ed_hot_stats(u32 out[6]) {
    out[0] = g_ed_hot[0] < g_ed_hot_cap ? g_ed_hot[0] : g_ed_hot_cap; out[1] = g_ed_hot_cap; out[2] = g_ed_hot[2]; out[3] = g_ed_hot_min;
    out[4] = g_ed_hot_evictions; out[5] = g_ed_hot_tick;
}
static size_t ed_hot_slot_of_key(const uint8_t* key32) {
    for (size_t sl = 0; sl < g_kc_ed.kc.cap; ++sl)
        if (memcmp(&g_kc_ed.keys[sl * 16], key32, 32) == 0) return sl;
    return (size_t)-1;
}
u32 sbve_ed_hot_hits_of_key(const uint8_t* key32) { const size_t sl = ed_hot_slot_of_key(key32); return sl == (size_t)-1 || sl >= g_ed_hot_khits.size() ? 0xFFFFFFFFu : g_ed_hot_khits[sl]; }
u32 sbve_ed_hot_wide_of_key(const uint8_t* key32) { const size_t sl = ed_hot_slot_of_key(key32); return sl == (size_t)-1 || sl >= g_ed_hot_kwide.size() ? 0xFFFFFFFEu : g_ed_hot_kwide[sl]; }
// promoted comb `index` against the host builder of combs (ed25519_core.h: build_ed_window_of) on -A, as sbv_ed25519_hot_selfcheck
// compares: the number of differing entries, or (size_t)-1 if nobody owns the index
size_t sbve_ed_hot_comb_mismatches(u32 index) {
    size_t slot = g_ed_hot_kwide.size();
    for (size_t i = 0; i < g_ed_hot_kwide.size(); ++i) if (g_ed_hot_kwide[i] == index) slot = i;
    if (slot == g_ed_hot_kwide.size() || !g_ed_hot_wtab) return (size_t)-1;
    ept A;
    if (!ed_decompress(A, &g_kc_ed.keys[slot * 16])) return (size_t)-2;
    fe25_neg(A.X, A.X);
    fe25_neg(A.T, A.T);
    std::vector<aniels> want((size_t)SBV_ED_HOT_WINDOWS * SBV_ED_HOT_PER_WINDOW);
    std::vector<std::thread> th;
    for (int j = 0; j < SBV_ED_HOT_WINDOWS; ++j) th.emplace_back([&, j] { build_ed_window_of(A, SBV_ED_HOT_BITS, j, want.data() + (size_t)j * SBV_ED_HOT_PER_WINDOW); });
    for (auto& t : th) t.join();
    const uint8_t* comb = g_ed_hot_wtab + (size_t)index * SBV_ED_HOT_COMB_BYTES;
    size_t bad = 0;
    for (size_t e = 0; e < want.size(); ++e) if (memcmp(comb + e * SBV_ED_HOT_PITCH, &want[e], sizeof(aniels)) != 0) ++bad;
    return bad;
}
// grouped form (ed25519_group.h), emulated sequentially with the device pipeline's chunking.
// stats_out[0..3] = groups, grouped tuples, ungrouped tuples, ungrouped tuples rejected for their key.
void sbve_ed25519_verify_batch_grouped(const uint8_t* tuples_in, size_t n, uint8_t* bitmap, u32 min_count, u32 max_groups,
                                       u32 ht_bits, int chunks, int parts, u32* stats_out) {
    size_t cap = (n + 63) & ~(size_t)63;
    if (cap == 0) cap = 64;
    uint8_t* tuples = (uint8_t*)aligned_alloc(16, cap * 128);           // the kernels read 16-byte vectors
    memcpy(tuples, tuples_in, n * 128);
    std::vector<u32> ht((size_t)1 << ht_bits, 0), rep(cap), cnt(cap, 0), slot_of(cap), group_rep(max_groups ? max_groups : 1), counters(SBV_GROUP_COUNTERS, 0),
        grp_idx(cap, 0xFFFFFFFFu), ung_idx(cap), ung_cand(cap), slots(cap), gcount(2 * (size_t)(max_groups ? max_groups : 1), 0), grp_of(cap, 0xFFFFFFFFu);
    GroupState g{};
    g.seed = g_hash_seed;
    g.gcount = gcount.data(); g.gcursor = gcount.data() + (max_groups ? max_groups : 1); g.grp_of = grp_of.data(); g.ung_cand = ung_cand.data(); g.sorted = (u32)g_group_sort;
    g.ht = ht.data(); g.ht_mask = (u32)(((size_t)1 << ht_bits) - 1); g.rep = rep.data(); g.cnt = cnt.data(); g.slot_of = slot_of.data();
    g.group_rep = group_rep.data(); g.counters = counters.data(); g.grp_idx = grp_idx.data(); g.ung_idx = ung_idx.data();
    g.slots = slots.data(); g.max_groups = max_groups;
    group_set_threshold(g, min_count);
    std::vector<uint8_t> accb(cap, 0xEE), okb(cap, 0);
    std::vector<u32> ungxy(cap * SBV_ED_UNGXY_WORDS, 0xDEADBEEFu);
    KeyCache kc = g_kc_ed.kc;            // this scheme's persistent key-table cache (off unless sbve_scheme_key_cache(2, ...) switched it on)
    for (size_t i = 0; i < n; ++i) ed_group_insert_lane(tuples, i, g);
    for (size_t i = 0; i < n; ++i) group_assign_lane_t<128, 64, 8>(tuples, i, g, kc);      // cached keys are grouped whatever their count
    const u32 ngroups = counters[0] < max_groups ? counters[0] : max_groups;
    if (g.sorted) {                      // key-sorted list: classify, then the counting sort of p256_group.h (scatter walked backwards)
        for (size_t i = 0; i < n; ++i) ed_group_classify_lane(i, g);
        for (u32 L = 0; L < counters[4]; ++L) ed_group_keycheck_lane(tuples, L, g, accb.data(), ungxy.data());     // k_ed_keycheck: candidates whose key is no point leave here
        for (size_t i = 0; i < n; ++i) group_sort_count_lane(i, g);
        group_sort_scan_seq(g, ngroups);
        for (size_t i = n; i-- > 0;) group_sort_scatter_lane(i, g);
        const RunOrder run_order(ngroups);
        for (u32 L = 0; L < counters[1]; ++L)
            if (grp_idx[L] >= n || slots[grp_idx[L]] != grp_of[L] || !run_order.ok(L ? grp_of[L - 1] : SBV_GROUP_NONE, grp_of[L])) ++g_sort_violations;
    } else {
        for (size_t i = 0; i < n; ++i) ed_group_split_lane(i, g);
    }
    const bool tm = g.sorted != 0;       // tuple-major accumulator records
    u32* gacc = (u32*)aligned_alloc(16, (size_t)SBV_ED_GACC_WORDS * cap * 4);
    const edcomb bc = ed_bcomb();
    for (size_t i = 0; i < n; ++i) ed_gphase_lane(tuples, i, bc, gacc, cap, okb.data(), tm);
    const size_t ng1 = ngroups ? ngroups : 1;
    u32* jbases = (u32*)aligned_alloc(16, ng1 * SBV_ED_KEY_WINDOWS * SBV_ED_JBASE_DWORDS * 4);
    aniels* ktab = (aniels*)aligned_alloc(64, ng1 * SBV_ED_KEYTAB_ENTRIES * sizeof(aniels));
    memset(ktab, 0xA5, ng1 * SBV_ED_KEYTAB_ENTRIES * sizeof(aniels));
    std::vector<uint8_t> kvalid(ng1, 0);
    u32* tmpa = (u32*)aligned_alloc(16, SBV_ED_KEY_PER_WINDOW * SBV_ED_WINDOW_TMP_WORDS * 4);
    // the two phases of the key-table cache (k_key_cache_lookup_t / _insert_t), group by group
    std::vector<u32> tslot(ng1, SBV_GROUP_NONE);
    std::vector<uint8_t> cold(ng1, 1);
    if (kc.enabled) kc.count[1] = kc.count[2] = 0;
    for (u32 k = 0; k < ngroups; ++k) key_cache_phase_lookup<128, 64, 8>(tuples, g, kc, k, tslot.data(), cold.data());
    for (u32 k = 0; k < ngroups; ++k) key_cache_phase_insert<128, 64, 8>(tuples, g, kc, k, tslot.data());
    auto table_of = [&](u32 k) -> aniels* { return tslot[k] < kc.cap ? (aniels*)g_kc_ed.pool + (size_t)tslot[k] * SBV_ED_KEYTAB_ENTRIES : ktab + (size_t)k * SBV_ED_KEYTAB_ENTRIES; };
    auto valid_of = [&](u32 k) -> uint8_t* { return tslot[k] < kc.cap ? &g_kc_ed.valid[tslot[k]] : &kvalid[k]; };
    memset(bitmap, 0, (n + 7) / 8);
    // hot keys (k_ed_hot_class; the wavefronts of the wide pass: ed_wave_is_wide) — with the cache on, a pool and the key-sorted list
    const bool hot_on = g_ed_hot_cap && g_ed_hot_wtab && kc.enabled && g.sorted && g_ed_hot_kwide.size() >= kc.cap;
    std::vector<uint8_t> wide(ng1, 0), wave_wide((counters[1] + 63) / 64 + 1, 0);
    if (hot_on) {
        ++g_ed_hot_tick;
        g_ed_hot[1] = g_ed_hot[2] = g_ed_hot[3] = 0;
        for (u32 k = 0; k < ngroups; ++k) group_hot_class_lane(k, g, tslot.data(), cold.data(), kc.cap, g_ed_hot_kwide.data(), g_ed_hot_khits.data(), wide.data());
        for (u32 w = 0; w * 64 < counters[1]; ++w) {
            bool all = true;
            for (u32 L = w * 64; L < counters[1] && L < (w + 1) * 64; ++L) all = all && grp_of[L] < ngroups && wide[grp_of[L]] != 0;
            wave_wide[w] = all ? 1 : 0;
        }
        // k_ed_qphase_wide: [k](-A) from the slot's 16-bit comb, before any table of this batch exists
        for (u32 L = 0; L < counters[1]; ++L) {
            if (!wave_wide[L / 64]) continue;
            ++g_ed_hot[2];
            const u32 t = grp_idx[L];
            const u32 grp = grp_of[L];
            const u32 slot = tslot[grp];
            const bool v = ed_qphase_wide_lane(tuples, t, g_kc_ed.valid[slot] != 0, g_ed_hot_wtab + (size_t)g_ed_hot_kwide[slot] * SBV_ED_HOT_COMB_BYTES, gacc, okb.data());
            accb[t] = v ? SBV_ED_PENDING : 0;
        }
    }
    for (int c = 0; c < chunks; ++c) {
        const int j_first = SBV_ED_KEY_WINDOWS * c / chunks, j_end = SBV_ED_KEY_WINDOWS * (c + 1) / chunks;
        for (u32 k = 0; k < ngroups; ++k)
            if (cold[k]) {
                // k_ed_keytab_bases: the four lanes of the key's quad in lockstep (edchain_run) — and, as the reference of the byte-for-byte
                // claim, the one-lane chain on a copy of the same state (the chain of a later chunk continues from the recorded base)
                u32* mine = jbases + (size_t)k * (SBV_ED_KEY_WINDOWS * SBV_ED_JBASE_DWORDS);
                std::vector<u32> ref(mine, mine + SBV_ED_KEY_WINDOWS * SBV_ED_JBASE_DWORDS);
                uint8_t vref = 0xEE;
                ed_keytab_bases_lane(tuples, k, g, ref.data() - (size_t)k * (SBV_ED_KEY_WINDOWS * SBV_ED_JBASE_DWORDS), &vref, j_first, j_end - 1);
                edchain_quad_host q;
                edchain_run(q, tuples, k, g, jbases, valid_of(k), j_first, j_end - 1);
                if (memcmp(ref.data() + (size_t)j_first * SBV_ED_JBASE_DWORDS, mine + (size_t)j_first * SBV_ED_JBASE_DWORDS, (size_t)(j_end - j_first) * SBV_ED_JBASE_DWORDS * 4) != 0 ||
                    (j_first == 0 && vref != *valid_of(k)))
                    ++g_ed_chain_mismatches;
            }
        for (u32 k = 0; k < ngroups; ++k)
            for (int j = j_first; j < j_end && cold[k]; ++j)
                for (int part = 0; part < parts; ++part) {
                    const size_t w = (size_t)k * SBV_ED_KEY_WINDOWS + j;
                    ed_keytab_window_lane(jbases + w * SBV_ED_JBASE_DWORDS, part, parts, tmpa, table_of(k) + (size_t)j * SBV_ED_KEY_PER_WINDOW);
                }
        const bool last = c + 1 == chunks;
        for (u32 L = 0; L < counters[1]; ++L) {
            if (wave_wide[L / 64]) continue;                             // the wide pass's wavefront
            const u32 t = grp_idx[L];
            const u32 grp = g.sorted ? grp_of[L] : slots[t];
            const bool v = grp < ngroups ? ed_qphase_lane(tuples, t, 0, 1, table_of(grp), valid_of(grp), gacc, cap, okb.data(), j_first, j_end, last, tm)
                                         : ed_qphase_lane(tuples, t, SBV_GROUP_NONE, 1, ktab, kvalid.data(), gacc, cap, okb.data(), j_first, j_end, last, tm);
            if (last) accb[t] = v ? SBV_ED_PENDING : 0;
        }
    }
    if (hot_on) {
        // the tail: k_hot_decay, k_promote_select, k_promote_evict (the functions of p256_group.h on this scheme's arrays, groups visited
        // backwards: the order is free), k_ed_promote_window, k_promote_publish
        if (g_ed_hot_tick % SBV_HOT_DECAY_EVERY == SBV_HOT_DECAY_EVERY - 1) for (u32 sl = 0; sl < kc.cap; ++sl) hot_decay_lane(sl, g_ed_hot_khits.data());
        std::vector<u32> plist(2 * SBV_PROMOTE_MAX, 0xDEADBEEFu), elist(SBV_PROMOTE_MAX, 0xDEADBEEFu);
        for (u32 k = ngroups; k-- > 0;)
            group_promote_select_lane(k, tslot.data(), g_kc_ed.valid.data(), kc.cap, g_ed_hot_kwide.data(), g_ed_hot_khits.data(), g_ed_hot_min, g_ed_hot_cap, g_ed_hot,
                                      plist.data(), elist.data());
        {
            const u32 ncand = g_ed_hot[3] < SBV_PROMOTE_MAX ? g_ed_hot[3] : SBV_PROMOTE_MAX;
            u32 entries = g_ed_hot[1] < SBV_PROMOTE_MAX ? g_ed_hot[1] : SBV_PROMOTE_MAX;
            std::vector<u32> taken((g_ed_hot_cap + 31) / 32 + 1, 0);
            for (u32 c = 0; c < ncand; ++c) {
                u32 bh = 0xFFFFFFFFu, bw = 0xFFFFFFFFu;
                for (u32 lane = 0; lane < 3; ++lane) {
                    u32 h, w;
                    hot_evict_scan(g_ed_hot_khits.data(), g_ed_hot_wowner.data(), taken.data(), g_ed_hot_cap, kc.cap, lane, 3u, h, w);
                    if (hot_evict_better(h, w, bh, bw)) { bh = h; bw = w; }
                }
                const u32 before = entries;
                entries = hot_evict_commit(elist[c], bh, bw, g_ed_hot_khits.data(), g_ed_hot_kwide.data(), g_ed_hot_wowner.data(), taken.data(), entries, plist.data());
                g_ed_hot_evictions += entries - before;
            }
            if (ncand) g_ed_hot[1] = entries;
        }
        const u32 live = g_ed_hot[1] < SBV_PROMOTE_MAX ? g_ed_hot[1] : SBV_PROMOTE_MAX;
        for (u32 i = 0; i < live; ++i) {
            if (plist[2 * i] == 0xFFFFFFFFu) continue;
            const aniels* key_tab = (const aniels*)g_kc_ed.pool + (size_t)plist[2 * i] * SBV_ED_KEYTAB_ENTRIES;
            uint8_t* comb = g_ed_hot_wtab + (size_t)plist[2 * i + 1] * SBV_ED_HOT_COMB_BYTES;
            std::vector<std::thread> th;                                 // lanes are independent: one host thread per window
            for (u32 j = 0; j < SBV_ED_HOT_WINDOWS; ++j)
                th.emplace_back([=] {
                    std::vector<u32> tmpj(SBV_ED_HOT_TMP_WORDS);
                    for (u32 part = SBV_ED_HOT_PARTS; part-- > 0;) ed_widetab_lane(key_tab, j, part, tmpj.data(), comb);
                });
            for (auto& t : th) t.join();
        }
        for (u32 i = 0; i < live; ++i)
            if (plist[2 * i] != 0xFFFFFFFFu) { g_ed_hot_kwide[plist[2 * i]] = plist[2 * i + 1]; g_ed_hot_wowner[plist[2 * i + 1]] = plist[2 * i]; }
    }
    // k_ed_finish: the pending tuples' encodings, one inversion per SBV_ED_FINISH_T consecutive tuples
    for (u32 L = 0; L < counters[2]; ++L) accb[ung_idx[L]] = 0;         // the ungrouped list's verdicts come from the one-lane kernel below
    for (size_t i0 = 0; i0 < n; i0 += SBV_ED_FINISH_T) ed_finish_lane(tuples, n, i0, gacc, cap, accb.data(), tm);
    for (u32 L = 0; L < counters[1]; ++L) {
        const u32 t = grp_idx[L];
        if (accb[t] == 1) bitmap[t >> 3] |= (uint8_t)(1u << (t & 7));
        else if (accb[t] != 0) ++g_sort_violations;                      // a marker the finish pass left behind
    }
    u32* qtab = (u32*)aligned_alloc(16, SBV_ED_QTAB_ENTRIES * SBV_ED_PT_WORDS * 4);
    for (u32 L = 0; L < counters[2]; ++L) {
        const u32 t = ung_idx[L];
        const bool v1 = ed25519_verify_lane(EdGlobalTuple{ed_tuple_words(tuples, t)}, qtab, btab());
        bool v = v1;
        if (g.sorted) {
            // k_ed_generic_list on the key the key check left behind (no second square root) ...
            v = ed25519_verify_lane(EdGlobalTuple{ed_tuple_words(tuples, t)}, qtab, btab(), ungxy.data() + (size_t)L * SBV_ED_UNGXY_WORDS);
            if (v != v1) ++g_ed_quad_mismatches;
            // ... and (SBV_ED_UNGROUPED_QUAD=1)
            // k_ed_generic_quad: the four lanes of the tuple's quad in lockstep, on the key the key check left behind — and the one-lane
            // kernel's verdict beside it
            edchain_quad_host q;
            bool v4[4];
            ed25519_verify_quad(q, EdGlobalTuple{ed_tuple_words(tuples, t)}, ungxy.data() + (size_t)L * SBV_ED_UNGXY_WORDS, qtab, btab(), v4);
            if (v4[0] != v1 || v4[1] != v1 || v4[2] != v1 || v4[3] != v1) ++g_ed_quad_mismatches;
            if (L & 1) v = v4[L & 3];
        }
        if (v) bitmap[t >> 3] |= (uint8_t)(1u << (t & 7));
    }
    free(qtab); free(tmpa); free(ktab); free(jbases); free(tuples); free(gacc);
    if (stats_out) { stats_out[0] = ngroups; stats_out[1] = counters[1]; stats_out[2] = counters[2]; stats_out[3] = counters[3]; }
}
// Ed25519 message front end (sha512_dev.h): 512-bit little-endian x -> x mod L; sig | pk | msg -> 128-byte tuple
void sbve_mod_l_512(const u32* x16, u32* out8) { mod_l_512(x16, out8); }
void sbve_ed_msg_frontend(const uint8_t* sig64, const uint8_t* a32, const uint8_t* msg, size_t mlen, uint8_t out128[128]) {
    u32 w[32];
    ed_msg_frontend_lane(sig64, a32, msg, mlen, w);
    memcpy(out128, w, 128);
}
// carry-free GF(2^255-19) (ed25519_fe.h): ten raw signed limbs in and out; freeze -> 8 canonical words
static fe25 f25in(const int32_t* a) { fe25 x; memcpy(&x, a, 40); return x; }
void sbve_fe25_mul(const int32_t* a, const int32_t* b, int32_t* out) { fe25 z; fe25_mul(z, f25in(a), f25in(b)); memcpy(out, &z, 40); }
void sbve_fe25_sqr(const int32_t* a, int32_t* out) { fe25 z; fe25_sqr(z, f25in(a)); memcpy(out, &z, 40); }
void sbve_fe25_carry(const int32_t* a, int32_t* out) { fe25 z; fe25_carry(z, f25in(a)); memcpy(out, &z, 40); }
void sbve_fe25_inv(const int32_t* a, int32_t* out) { fe25 z; fe25_inv(z, f25in(a)); memcpy(out, &z, 40); }
void sbve_fe25_inv_gcd(const int32_t* a, int32_t* out) { fe25 z; fe25_inv_gcd(z, f25in(a)); memcpy(out, &z, 40); }
void sbve_fe25_freeze(const int32_t* a, u32* out8) { u256 w; fe25_freeze(w, f25in(a)); memcpy(out8, &w, 32); }
void sbve_fe25_from_words(const u32* w8, int32_t* out) { fe25 z; fe25_from_words(z, w8); memcpy(out, &z, 40); }
void sbve_fe25_consts(int32_t* out30) { fe25 d = fe25_d(), d2 = fe25_2d(), s = fe25_sqrtm1(); memcpy(out30, &d, 40); memcpy(out30 + 10, &d2, 40); memcpy(out30 + 20, &s, 40); }

// ---- secp256k1 (k256_fe.h, k256_sc.h, k256_core.h) ---------------------------------------------------------------------
static kapt* g_k256_gtab = nullptr;
static const kapt* k256_gtab() {
    if (!g_k256_gtab) {
        g_k256_gtab = (kapt*)aligned_alloc(64, sizeof(kapt) * SBV_K256_G_ENTRIES);
        std::vector<std::thread> th;
        for (int j = 0; j < SBV_K256_G_WINDOWS; ++j)
            th.emplace_back([j] { k256_build_g_window(j, g_k256_gtab + (size_t)j * SBV_K256_G_PER_WINDOW, SBV_K256_G_PER_WINDOW); });
        for (auto& t : th) t.join();
    }
    return g_k256_gtab;
}
static kfe kfe_in(const u32* w8) { u256 w; memcpy(&w, w8, 32); kfe r; kfe_from_words(r, w); return r; }
static void kfe_out(u32* w8, const kfe& a) { u256 w; kfe_to_words(w, a); memcpy(w8, &w, 32); }
// op: 0 mul, 1 sqr, 2 add, 3 sub, 4 inv, 5 a*3 - b*8 (kfe_lin), 6 neg; words in (any 256-bit value), canonical words out
void sbve_kfe_op(int op, const u32* a, const u32* b, u32* out) {
    const kfe x = kfe_in(a), y = kfe_in(b);
    kfe z = kfe_zero();
    if (op == 0) kfe_mul(z, x, y);
    else if (op == 1) kfe_sqr(z, x);
    else if (op == 2) kfe_add(z, x, y);
    else if (op == 3) kfe_sub(z, x, y);
    else if (op == 4) kfe_inv(z, x);
    else if (op == 5) kfe_lin(z, x, 3, y, 8);
    else if (op == 6) kfe_cneg(z, x, true);
    kfe_out(out, z);
}
// raw limbs in (any signed 32-bit values that satisfy the product bound), canonical words of a * b (or a^2 when b is null) out:
// the extremes of the operand contract, which kfe_in can never produce
void sbve_kfe_mul_raw(const int32_t* a9, const int32_t* b9, u32* out) {
    kfe x, y, z;
    memcpy(&x, a9, 36);
    if (b9) { memcpy(&y, b9, 36); kfe_mul(z, x, y); } else kfe_sqr(z, x);
    kfe_out(out, z);
}
// a chain of operations on unnormalised intermediates: ((a - b) * (a + b) - a^2 + b^2) must be 0, and is_zero must say so
int sbve_kfe_chain_is_zero(const u32* a, const u32* b) {
    const kfe x = kfe_in(a), y = kfe_in(b);
    kfe d, s2, p, xx, yy, t;
    kfe_sub(d, x, y); kfe_add(s2, x, y); kfe_mul(p, d, s2); kfe_sqr(xx, x); kfe_sqr(yy, y);
    kfe_sub(t, p, xx); kfe_add(t, t, yy);
    return kfe_is_zero(t) ? 1 : 0;
}
void sbve_ksc_mul(const u32* a, const u32* b, u32* out) { u256 x, y, z; memcpy(&x, a, 32); memcpy(&y, b, 32); ksc_mul(z, x, y); memcpy(out, &z, 32); }
void sbve_ksc_inv(const u32* a, u32* out) { u256 x, z; memcpy(&x, a, 32); ksc_inv(z, x); memcpy(out, &z, 32); }
void sbve_ksc_reduce512(const u32* x16, u32* out) { u256 z; ksc_reduce512(z, x16); memcpy(out, &z, 32); }
// entry (window j, multiple m) of the comb of G as 16 words x | y
void sbve_k256_g_entry(int j, int m, u32* out16) { memcpy(out16, k256_gtab() + (size_t)j * SBV_K256_G_PER_WINDOW + (m - 1), 64); }
// k * (x, y) + l * G on the device point layer (double-and-add with kpt_dbl / kpt_madd): affine words out, returns 0 for infinity
int sbve_k256_mul2(const u32* k8, const u32* x8, const u32* y8, const u32* l8, u32* out16) {
    const kfe x = kfe_in(x8), y = kfe_in(y8);
    kfe gx, gy;
    kfe_from_words(gx, k256_gx_words());
    kfe_from_words(gy, k256_gy_words());
    kjpt R;
    kpt_set_inf(R);
    for (int bit = 255; bit >= 0; --bit) {
        kpt_dbl(R, R);
        kpt_madd(R, R, x, y, false, ((k8[bit >> 5] >> (bit & 31)) & 1) == 0);
        kpt_madd(R, R, gx, gy, false, ((l8[bit >> 5] >> (bit & 31)) & 1) == 0);
    }
    if (R.inf) return 0;
    kfe zi, zi2, zi3, ax, ay;
    kfe_inv(zi, R.Z); kfe_sqr(zi2, zi); kfe_mul(zi3, zi2, zi); kfe_mul(ax, R.X, zi2); kfe_mul(ay, R.Y, zi3);
    kfe_out(out16, ax); kfe_out(out16 + 8, ay);
    return 1;
}
// the whole path: stage A (k256_prep_lane) + stage B (k256_verify_lane), lane by lane
// GLV decomposition of a scalar (k256_sc.h: ksc_split_lambda): out = k1[8] | k2[8] | neg1 | neg2
void sbve_k256_split_lambda(const u32* k8, u32* out18) {
    u256 k, k1, k2;
    for (int i = 0; i < 8; ++i) k.v[i] = k8[i];
    bool n1, n2;
    ksc_split_lambda(k1, n1, k2, n2, k);
    for (int i = 0; i < 8; ++i) { out18[i] = k1.v[i]; out18[8 + i] = k2.v[i]; }
    out18[16] = n1 ? 1u : 0u;
    out18[17] = n2 ? 1u : 0u;
}
void sbve_k256_verify_batch(const uint8_t* tuples, size_t n, uint8_t* bitmap) {
    size_t cap = (n + 63) & ~(size_t)63;
    if (cap == 0) cap = 64;
    std::vector<u32> r(8 * cap), u1(8 * cap), u2(8 * cap), qx(8 * cap), qy(8 * cap), sm(8 * cap);
    std::vector<uint8_t> ok(cap, 0);
    Scratch s{r.data(), u1.data(), u2.data(), qx.data(), qy.data(), sm.data(), ok.data(), cap};
    HostWords hw{tuples, 160};
    for (size_t i = 0; i < n; ++i) k256_prep_lane(hw(0, i), i, s);
    memset(bitmap, 0, (n + 7) / 8);
    const kapt* gt = k256_gtab();
    const unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            u32* qtab = (u32*)aligned_alloc(16, SBV_K256_QTAB_WORDS * 4);
            const size_t lo = (n * t / nt) & ~(size_t)7, hi = t + 1 == nt ? n : (n * (t + 1) / nt) & ~(size_t)7;     // whole bitmap bytes per thread
            for (size_t i = lo; i < hi; ++i)
                if (k256_verify_lane(s, i, qtab, gt)) bitmap[i >> 3] |= (uint8_t)(1u << (i & 7));
            free(qtab);
        });
    for (auto& t : th) t.join();
}

// ---- unit hooks (plain little-endian limb arrays) ---------------------------------------------------
// u1 * G through the grouped step's comb walk (k256_group.h: k256_gphase_point) with a `bits`-wide comb built here
// (k256_build_g_window_bits, as the host builds the 20-bit one) -> affine x | y words; returns 0 for infinity
int sbve_k256_gcomb_mul(const u32* u1w, int bits, u32* out16) {
    static std::vector<kapt> tab;
    static int tab_bits = 0;
    const int windows = (257 + bits - 1) / bits;
    if (tab_bits != bits) {
        tab.assign((size_t)windows << (bits - 1), kapt{});
        std::vector<std::thread> th;
        for (int j = 0; j < windows; ++j) th.emplace_back([=] { k256_build_g_window_bits(bits, j, tab.data() + ((size_t)j << (bits - 1)), 1 << (bits - 1)); });
        for (auto& t : th) t.join();
        tab_bits = bits;
    }
    u256 u1;
    memcpy(&u1, u1w, 32);
    kjpt R;
    kpt_set_inf(R);
    k256_gphase_point(R, u1, kgcomb_make(tab.data(), bits));
    if (R.inf) return 0;
    kfe zi, zi2, zi3, x, y;
    kfe_inv(zi, R.Z);
    kfe_sqr(zi2, zi); kfe_mul(zi3, zi2, zi);
    kfe_mul(x, R.X, zi2); kfe_mul(y, R.Y, zi3);
    kapt e;
    kapt_store(&e, x, y);
    memcpy(out16, &e, 64);
    return 1;
}

// modinv30.h: the variable-time division steps against the constant-time ones and the modulus (0 = P-256 p, 1 = P-256 n,
// 2 = 2^255 - 19, 3 = secp256k1 p, 4 = secp256k1 n); out_var / out_ct = x^-1 mod m; returns 1 when both agree
int sbve_modinv30_both(const u32* x8, int which, u32* out_var, u32* out_ct) {
    const modinfo30 mi = which == 0 ? modinfo30_p256() : which == 1 ? modinfo30_p256_order() : which == 2 ? modinfo30_25519()
                       : which == 3 ? modinfo30_k256_p() : modinfo30_k256_n();
    u256 x, a, b;
    memcpy(&x, x8, 32);
    modinv30(a, x, mi);
    modinv30_ct(b, x, mi);
    memcpy(out_var, &a, 32); memcpy(out_ct, &b, 32);
    return memcmp(&a, &b, 32) == 0;
}

// secp256k1 grouped step (k256_group.h, k256_group_kernels.hip) emulated sequentially: stage A with records, grouping, key check
// of the ungrouped candidates, counting sort, G phase over the sorted list, the per-batch combs (quad chain in lockstep, rows,
// fill) and the Q phase in `chunks` pieces, the one-lane kernel over the ungrouped list.  stats_out as for the P-256 form.
static int g_k256_prep_T = 1;
void sbve_set_k256_prep_t(int t) { g_k256_prep_T = t >= 1 && t <= 8 ? t : 1; }   // GroupSync::k256_prep_t
void sbve_k256_verify_batch_grouped(const uint8_t* tuples, size_t n, uint8_t* bitmap, u32 min_count, u32 max_groups, u32 ht_bits, int chunks,
                                    u32* stats_out) {
    size_t cap = (n + 63) & ~(size_t)63;
    if (cap == 0) cap = 64;
    std::vector<u32> r(8 * cap), u1(8 * cap), u2(8 * cap), qx(8 * cap), qy(8 * cap), sm(8 * cap), rec(cap * SBV_REC_WORDS + 4, 0xDEADBEEFu);
    std::vector<uint8_t> ok(cap, 0);
    Scratch s{r.data(), u1.data(), u2.data(), qx.data(), qy.data(), sm.data(), ok.data(), cap};
    u32* rec_al = rec.data();
    while ((uintptr_t)rec_al & 15) ++rec_al;
    s.rec = rec_al;
    HostWords hw{tuples, 160};
    if (g_k256_prep_T > 1) {                    // k_k256_prep_chunk: workgroups of 64 lanes, T tuples per lane, one inversion per lane
        const size_t per_block = (size_t)64 * g_k256_prep_T;
        auto words = [&](size_t idx) { return hw(0, idx); };
        for (size_t b = 0; b * per_block < n; ++b)
            for (int t = 0; t < 64; ++t) k256_prep_chunk(words, n, s, b * per_block + t, (size_t)64, g_k256_prep_T);
    } else {
        for (size_t i = 0; i < n; ++i) k256_prep_lane(hw(0, i), i, s);
    }
    const u32 G = max_groups ? max_groups : 1;
    std::vector<u32> ht((size_t)1 << ht_bits, 0), rep(cap), cnt(cap, 0), slot_of(cap), group_rep(G), counters(SBV_GROUP_COUNTERS, 0), ung_cand(cap),
        grp_idx(cap, 0xFFFFFFFFu), ung_idx(cap), slots(cap), gcount(2 * (size_t)G, 0), grp_of(cap, 0xFFFFFFFFu);
    GroupState g{};
    g.seed = g_hash_seed;
    g.gcount = gcount.data(); g.gcursor = gcount.data() + G; g.grp_of = grp_of.data(); g.ung_cand = ung_cand.data(); g.sorted = 1;
    g.ht = ht.data(); g.ht_mask = (u32)(((size_t)1 << ht_bits) - 1); g.rep = rep.data(); g.cnt = cnt.data(); g.slot_of = slot_of.data();
    g.group_rep = group_rep.data(); g.counters = counters.data(); g.grp_idx = grp_idx.data(); g.ung_idx = ung_idx.data();
    g.slots = slots.data(); g.max_groups = max_groups;
    group_set_threshold(g, min_count);
    KeyCache kc = g_kc_k256.kc;          // this curve's persistent key-table cache (off unless sbve_scheme_key_cache(1, ...) switched it on)
    for (size_t i = 0; i < n; ++i) group_insert_lane(tuples, i, g);
    for (size_t i = 0; i < n; ++i) group_assign_lane(tuples, i, g, kc);          // cached keys are grouped whatever their count
    std::vector<uint8_t> accb(cap, 0xEE);
    for (size_t i = 0; i < n; ++i) group_classify_lane(i, g);
    for (size_t L = counters[4]; L-- > 0;) k256_keycheck_lane(tuples, L, g, accb.data());
    const u32 ngroups = counters[0] < max_groups ? counters[0] : max_groups;
    for (size_t i = 0; i < n; ++i) group_sort_count_lane(i, g);
    group_sort_scan_seq(g, ngroups);
    for (size_t i = n; i-- > 0;) group_sort_scatter_lane(i, g);
    std::vector<u32> gacc((size_t)SBV_K256_GACC_WORDS * cap);
    for (u32 L = 0; L < counters[1]; ++L) k256_gphase_lane_sorted(s, grp_idx[L], L, kgcomb_make(k256_gtab(), 16), gacc.data());
    const size_t ng1 = ngroups ? ngroups : 1, per_key = (size_t)SBV_GTAB_WINDOWS * SBV_GTAB_PER_WINDOW;
    std::vector<u32> bases(ng1 * SBV_GTAB_WINDOWS * SBV_K256_BASES_STRIDE, 0xA5A5A5A5u), jstate(ng1 * SBV_K256_STATE_WORDS), tmpa(SBV_K256_WINDOW_TMP);
    kapt* ktab = (kapt*)aligned_alloc(64, ng1 * per_key * sizeof(kapt));
    memset(ktab, 0xA5, ng1 * per_key * sizeof(kapt));
    std::vector<uint8_t> kvalid(ng1, 0);
    // the two phases of the key-table cache (k_key_cache_lookup_t / _insert_t), group by group
    std::vector<u32> tslot(ng1, SBV_GROUP_NONE);
    std::vector<uint8_t> cold(ng1, 1);
    if (kc.enabled) kc.count[1] = kc.count[2] = 0;
    for (u32 k = 0; k < ngroups; ++k) key_cache_phase_lookup<160, 96, 16>(tuples, g, kc, k, tslot.data(), cold.data());
    for (u32 k = 0; k < ngroups; ++k) key_cache_phase_insert<160, 96, 16>(tuples, g, kc, k, tslot.data());
    auto table_of = [&](u32 k) -> kapt* { return tslot[k] < kc.cap ? (kapt*)g_kc_k256.pool + (size_t)tslot[k] * per_key : ktab + (size_t)k * per_key; };
    auto valid_of = [&](u32 k) -> uint8_t* { return tslot[k] < kc.cap ? &g_kc_k256.valid[tslot[k]] : &kvalid[k]; };
    memset(bitmap, 0, (n + 7) / 8);
    for (int c = 0; c < chunks; ++c) {
        const int j_first = SBV_GTAB_WINDOWS * c / chunks, j_end = SBV_GTAB_WINDOWS * (c + 1) / chunks;
        for (u32 k = 0; k < ngroups; ++k) {
            if (!cold[k]) continue;              // a cached key: its comb is already in the pool
            k256_quad_host q;
            k256_chain_run(q, tuples, k, g, jstate.data(), bases.data(), valid_of(k), j_first, j_end - 1);
            for (int j = j_first; j < j_end; ++j) {
                const size_t w = (size_t)k * SBV_GTAB_WINDOWS + j;
                kapt* row = table_of(k) + (size_t)j * SBV_GTAB_PER_WINDOW;
                for (int which = 0; which < 2; ++which) {
                    if (which == 1 && j == SBV_GTAB_WINDOWS - 1) continue;
                    k256_rows_lane(bases.data() + w * SBV_K256_BASES_STRIDE, which, j == SBV_GTAB_WINDOWS - 1, tmpa.data(), row);
                }
                if (j == SBV_GTAB_WINDOWS - 1) continue;
                for (int a = 1; a <= 7; ++a) k256_fill_lane(a, tmpa.data(), row);
            }
        }
        const bool last = c + 1 == chunks;
        for (u32 L = 0; L < counters[1]; ++L) {
            const u32 t = grp_idx[L], grp = grp_of[L];
            const bool v = grp < ngroups ? k256_qphase_lane_sorted(s, t, L, 0, 1, table_of(grp), valid_of(grp), gacc.data(), j_first, j_end, last)
                                         : k256_qphase_lane_sorted(s, t, L, SBV_GROUP_NONE, 1, ktab, kvalid.data(), gacc.data(), j_first, j_end, last);
            if (last && v) bitmap[t >> 3] |= (uint8_t)(1u << (t & 7));
        }
    }
    u32* qtab = (u32*)aligned_alloc(16, SBV_QTAB29_WORDS * 4);
    for (u32 L = 0; L < counters[2]; ++L) {
        const u32 t = ung_idx[L];
        if (k256_verify_lane(s, t, qtab, k256_gtab())) bitmap[t >> 3] |= (uint8_t)(1u << (t & 7));
    }
    free(qtab); free(ktab);
    if (stats_out) { stats_out[0] = ngroups; stats_out[1] = counters[1]; stats_out[2] = counters[2]; stats_out[3] = counters[3]; }
}

void sbve_fe_mul(const u32* a, const u32* b, u32* out) { fe x, y, z; memcpy(&x, a, 32); memcpy(&y, b, 32); fe_mul(z, x, y); memcpy(out, &z, 32); }
void sbve_fe_sqr(const u32* a, u32* out) { fe x, z; memcpy(&x, a, 32); fe_sqr(z, x); memcpy(out, &z, 32); }
void sbve_fe_add(const u32* a, const u32* b, u32* out) { fe x, y, z; memcpy(&x, a, 32); memcpy(&y, b, 32); fe_add(z, x, y); memcpy(out, &z, 32); }
void sbve_fe_sub(const u32* a, const u32* b, u32* out) { fe x, y, z; memcpy(&x, a, 32); memcpy(&y, b, 32); fe_sub(z, x, y); memcpy(out, &z, 32); }
void sbve_fe_inv(const u32* a, u32* out) { fe x, z; memcpy(&x, a, 32); fe_inv(z, x); memcpy(out, &z, 32); }
void sbve_mul_wide(const u32* a, const u32* b, u32* out16) { mul_wide(out16, a, b); }
void sbve_sqr_wide(const u32* a, u32* out16) { sqr_wide(out16, a); }
void sbve_mont_reduce(const u32* t16, u32* out) { fe z; fe_mont_reduce(z, t16); memcpy(out, &z, 32); }
// scalar field on the carry-free representation (p256_sc29.h): raw limbs in / out
void sbve_s29_mul(const i32* a, const i32* b, i32* out) { fe29 x, y, z; memcpy(&x, a, 36); memcpy(&y, b, 36); s29_mul(z, x, y); memcpy(out, &z, 36); }
void sbve_s29_canon(const i32* a, i32* out) { fe29 x, z; memcpy(&x, a, 36); s29_canon(z, x); memcpy(out, &z, 36); }
void sbve_s29_inv(const i32* a, i32* out) { fe29 x, z; memcpy(&x, a, 36); s29_inv(z, x); memcpy(out, &z, 36); }
void sbve_sc_mul(const u32* a, const u32* b, u32* out) { sc x, y, z; memcpy(&x, a, 32); memcpy(&y, b, 32); sc_mul(z, x, y); memcpy(out, &z, 32); }
void sbve_sc_inv(const u32* a, u32* out) { sc x, z; memcpy(&x, a, 32); sc_inv(z, x); memcpy(out, &z, 32); }
// division-step inversions (modinv30.h): Montgomery in/out wrappers and the plain-integer core (which: 0 = p, 1 = N, 2 = 2^255-19)
void sbve_fe_inv_gcd(const u32* a, u32* out) { fe x, z; memcpy(&x, a, 32); fe_inv_gcd(z, x); memcpy(out, &z, 32); }
void sbve_sc_inv_gcd(const u32* a, u32* out) { sc x, z; memcpy(&x, a, 32); sc_inv_gcd(z, x); memcpy(out, &z, 32); }
void sbve_modinv30(int which, const u32* a, u32* out) {
    u256 x, z; memcpy(&x, a, 32);
    modinv30(z, x, which == 0 ? modinfo30_p256() : which == 1 ? modinfo30_p256_order() : modinfo30_25519());
    memcpy(out, &z, 32);
}
// affine Montgomery-form G-table entry (j, k): 16 dwords
void sbve_g16_entry(int j, int k, u32* out16) { memcpy(out16, &g16tab()[(size_t)j * SBV_G16_PER_WINDOW + (k - 1)], 64); }
void sbve_gtab_entry(int j, int k, u32* out16) { memcpy(out16, &gtab()[(size_t)j * SBV_GTAB_PER_WINDOW + (k - 1)], 64); }

// ---- carry-free field (p256_fe29.h) and XYZZ point layer (p256_pt29.h): raw signed 29-bit limbs in / out ----------
void sbve_f29_mul(const i32* a, const i32* b, i32* out) { fe29 x, y, z; memcpy(&x, a, 36); memcpy(&y, b, 36); f29_mul(z, x, y); memcpy(out, &z, 36); }
void sbve_f29_sqr(const i32* a, i32* out) { fe29 x, z; memcpy(&x, a, 36); f29_sqr(z, x); memcpy(out, &z, 36); }
// hot-path forms: 32-bit-multiplier reduction, fused a*b - c*d, fused a^2 - v
void sbve_f29_mulx(const i32* a, const i32* b, i32* out) { fe29 x, y, z; memcpy(&x, a, 36); memcpy(&y, b, 36); f29_mulx(z, x, y); memcpy(out, &z, 36); }
void sbve_f29_sqrx(const i32* a, i32* out) { fe29 x, z; memcpy(&x, a, 36); f29_sqrx(z, x); memcpy(out, &z, 36); }
void sbve_f29_mul_sub_mul(const i32* a, const i32* b, const i32* c, const i32* d, i32* out) {
    fe29 x, y, u, v, z; memcpy(&x, a, 36); memcpy(&y, b, 36); memcpy(&u, c, 36); memcpy(&v, d, 36);
    fe29 nu; f29_neg(nu, u);
    f29_cols t; f29_cols_zero(t); f29_cols_mul(t, x, y); f29_cols_mul(t, nu, v); f29_reduce_x(z, t); f29_red_q(z);
    memcpy(out, &z, 36);
}
void sbve_f29_sqr_sub_val(const i32* a, const i32* v, i32* out) {
    fe29 x, w, z; memcpy(&x, a, 36); memcpy(&w, v, 36);
    f29_cols t; f29_cols_zero(t); f29_cols_sqr(t, x); f29_cols_sub_val(t, w); f29_reduce_x(z, t); f29_red_q(z);
    memcpy(out, &z, 36);
}
void sbve_f29_canon(const i32* a, i32* out) { fe29 x, z; memcpy(&x, a, 36); f29_canon(z, x); memcpy(out, &z, 36); }
void sbve_f29_norm(const i32* a, i32* out) { fe29 x, z; memcpy(&x, a, 36); f29_norm(z, x); memcpy(out, &z, 36); }
void sbve_f29_norm_red(const i32* a, i32* out) { fe29 x, z; memcpy(&x, a, 36); f29_norm_red(z, x); memcpy(out, &z, 36); }
int sbve_f29_is_zero(const i32* a) { fe29 x; memcpy(&x, a, 36); return f29_is_zero(x) ? 1 : 0; }
int sbve_f29_maybe_zero(const i32* a) { fe29 x; memcpy(&x, a, 36); return f29_maybe_zero(x) ? 1 : 0; }
void sbve_f29_unpack(const u32* w, i32* out) { fe29 z; f29_unpack(z, w); memcpy(out, &z, 36); }
void sbve_f29_pack(const i32* c, u32* w) { fe29 x; memcpy(&x, c, 36); f29_pack(w, x); }
void sbve_f29_from_fe(const u32* w, i32* out) { fe x; memcpy(&x, w, 32); fe29 z; f29_from_fe(z, x); memcpy(out, &z, 36); }
void sbve_f29_to_fe(const i32* a, u32* w) { fe29 x; memcpy(&x, a, 36); fe z; f29_to_fe(z, x); memcpy(w, &z, 32); }
void sbve_f29_from_plain(const u32* w, i32* out) { u256 x; memcpy(&x, w, 32); fe29 z; f29_from_plain(z, x); memcpy(out, &z, 36); }
// sum of n affine points given as stored table entries (16 words each: canonical x | y of the R = 2^261 domain), each
// optionally negated; out = X, Y, ZZ, ZZZ (36 limbs); returns the infinity flag
int sbve_pt29_sum(const u32* entries, const uint8_t* negs, size_t n, i32* out) {
    xyzz R;
    pt29_set_inf(R);
    for (size_t i = 0; i < n; ++i) {
        apt29 q;
        apt29_load(q, entries + 16 * i);
        pt29_madd(R, q, negs[i] != 0);
    }
    memcpy(out, &R.X, 36); memcpy(out + 9, &R.Y, 36); memcpy(out + 18, &R.ZZ, 36); memcpy(out + 27, &R.ZZZ, 36);
    return R.inf ? 1 : 0;
}
int sbve_pt29_rx_matches(const i32* xyzz36, int inf, const u32* r) {
    xyzz R;
    memcpy(&R.X, xyzz36, 36); memcpy(&R.Y, xyzz36 + 9, 36); memcpy(&R.ZZ, xyzz36 + 18, 36); memcpy(&R.ZZZ, xyzz36 + 27, 36);
    R.inf = inf != 0;
    u256 rr; memcpy(&rr, r, 32);
    return pt29_rx_matches(R, rr) ? 1 : 0;
}

// batch signing (p256_sign.h) lane by lane: the source the k_p256_sign kernel runs
void sbve_p256_sign_batch(const uint8_t* keys, uint32_t n_keys, const uint32_t* key_index, const uint8_t* digests, size_t n,
                          uint8_t* sigs, uint8_t* ok) {
    const gcomb gc = g16rtab();
    for (size_t i = 0; i < n; ++i) {
        uint32_t kidx = key_index ? key_index[i] : (uint32_t)(i % n_keys);
        const bool known = kidx < n_keys;
        if (!known) kidx = 0;
        u32 d[8], h[8], rs[16];
        for (int k = 0; k < 8; ++k) {
            const uint8_t* a = keys + (size_t)kidx * 32 + 4 * k;
            const uint8_t* b = digests + i * 32 + 4 * k;
            d[k] = ((u32)a[0] << 24) | ((u32)a[1] << 16) | ((u32)a[2] << 8) | a[3];
            h[k] = ((u32)b[0] << 24) | ((u32)b[1] << 16) | ((u32)b[2] << 8) | b[3];
        }
        const bool good = sign29_lane(d, h, gc, rs) && known;
        for (int k = 0; k < 16; ++k) {
            const u32 w = good ? rs[k] : 0u;
            sigs[i * 64 + 4 * k] = (uint8_t)(w >> 24); sigs[i * 64 + 4 * k + 1] = (uint8_t)(w >> 16);
            sigs[i * 64 + 4 * k + 2] = (uint8_t)(w >> 8); sigs[i * 64 + 4 * k + 3] = (uint8_t)w;
        }
        ok[i] = good ? 1 : 0;
    }
}

}  // extern "C"
